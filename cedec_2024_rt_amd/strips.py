"""Row-strip multi-GPU driver of the frame (new — the reference is single-GPU).

The image is cut into contiguous strips of STORAGE rows, one per rank. Only
`spatial_resampling` reads other pixels (examples/10_restir_di/10_restir_di.cu:323-340); its
neighbour offset is bounded by 30/1.96 * sqrt(2*23*ln 2) = 86.43 px, so an 87-row reservoir halo
exchanged with rank +-1 before each spatial pass makes the N-rank frame bit-identical to the
1-rank frame (RNG is keyed by global pixel coordinates). The exchange is point-to-point
(RCCL send/recv over xGMI with backend "nccl"; "gloo" in the CPU tests) — no collective sits on
the critical path.

`StripFrame` holds the partition + exchange logic and is backend-agnostic: the product backend is
`HipStripBackend` (C-ABI context per rank); the CPU tests drive the same logic with a checker
backend over gloo.
"""
import math

HALO_ROWS = 87  # ceil(86.43)

RT_RES_0, RT_RES_1, RT_RES_TEMPORAL = 0, 1, 2


def halo_bound(radius=30.0):
    """Exact bound of |neighbour offset|: rv0 >= 2^-23 when non-zero."""
    return radius / 1.96 * math.sqrt(2.0 * 23.0 * math.log(2.0))


def partition_rows(height, n_ranks, halo=HALO_ROWS):
    """Contiguous, near-equal strips; every strip must be at least `halo` rows tall so that a
    halo never reaches beyond the adjacent rank."""
    base, extra = divmod(height, n_ranks)
    bounds, a = [], 0
    for r in range(n_ranks):
        b = a + base + (1 if r < extra else 0)
        bounds.append((a, b))
        a = b
    if n_ranks > 1 and min(b - a for a, b in bounds) < halo:
        raise ValueError(f"{height} rows over {n_ranks} ranks gives strips thinner than the {halo}-row halo")
    return bounds


def exchange_plan(bounds, rank, halo=HALO_ROWS):
    """[(peer, send_row0, send_n, recv_row0, recv_n)] for one halo exchange of `rank`."""
    a, b = bounds[rank]
    plan = []
    if rank > 0:
        pa, pb = bounds[rank - 1]
        n = min(halo, b - a)
        m = min(halo, pb - pa)
        plan.append((rank - 1, a, n, a - m, m))
    if rank + 1 < len(bounds):
        na, nb = bounds[rank + 1]
        n = min(halo, b - a)
        m = min(halo, nb - na)
        plan.append((rank + 1, b - n, n, b, m))
    return plan


class StripFrame:
    """One rank's frame loop: examples/10_restir_di/10_restir_di.cpp:257-379 on its strip, with a
    halo exchange in front of every spatial pass.

    backend needs: raycast(), generate_candidate(frame, dst), temporal_resampling(frame, prev, inout),
    save_temporal_reservoir(src, dst), spatial_resampling(frame, k, src, dst), resolve(res),
    tone_mapping(), clear(), passes (int), halo_export(res, row0, n) -> tensor,
    halo_import(res, row0, n, tensor), halo_empty(n) -> tensor.
    """

    def __init__(self, backend, bounds, rank, dist=None, halo=HALO_ROWS):
        self.b, self.bounds, self.rank, self.dist, self.halo = backend, bounds, rank, dist, halo
        self.plan = exchange_plan(bounds, rank, halo)

    def exchange(self, res):
        if not self.plan:
            return
        d = self.dist
        ops, recvs = [], []
        for peer, s0, sn, r0, rn in self.plan:
            t_send = self.b.halo_export(res, s0, sn)
            t_recv = self.b.halo_empty(rn)
            ops.append(d.P2POp(d.isend, t_send, peer))
            ops.append(d.P2POp(d.irecv, t_recv, peer))
            recvs.append((r0, rn, t_recv))
        for w in d.batch_isend_irecv(ops):
            w.wait()
        for r0, rn, t in recvs:
            self.b.halo_import(res, r0, rn, t)

    def frame(self, frame, clear_first=False):
        b = self.b
        if getattr(b, "staged", False):
            # fused stages of the C-ABI (generate+temporal in one kernel, rotating buffers)
            b.frame_stage(frame, 0, clear_first)
            for k in range(b.passes):
                self.exchange(b.frame_stage_input(k + 1))
                b.frame_stage(frame, k + 1)
            b.frame_stage(frame, b.passes + 1)
            return None
        if clear_first:
            b.clear()
        b.raycast()
        b.generate_candidate(frame, RT_RES_0)
        b.temporal_resampling(frame, RT_RES_TEMPORAL, RT_RES_0)
        b.save_temporal_reservoir(RT_RES_0, RT_RES_TEMPORAL)
        src, dst = RT_RES_0, RT_RES_1
        for k in range(b.passes):
            if k != 0:
                src, dst = dst, src
            self.exchange(src)
            b.spatial_resampling(frame, k, src, dst)
        b.resolve(dst)
        b.tone_mapping()
        return dst


class HipStripBackend:
    """Product backend: one C-ABI context on this rank's GPU, work enqueued on torch's current
    stream so that RCCL's stream dependencies order packs, sends, receives and unpacks."""

    def __init__(self, renderer, device, host_staging=False):
        import torch

        self.r, self.torch, self.device = renderer, torch, device
        self.passes = int(renderer.options()["spatial_resampling_passes"][0])
        # host_staging: the transport cannot move device memory (gloo; used to exercise this path
        # with several ranks on ONE GPU, where RCCL refuses duplicate devices)
        self.host_staging = host_staging
        self.staged = True  # use rt_frame_stage

    def __getattr__(self, name):  # kernel entry points pass straight through
        return getattr(self.r, name)

    def _dev_empty(self, n_rows):
        return self.torch.empty(self.r.halo_bytes(n_rows), dtype=self.torch.uint8, device=self.device)

    def halo_empty(self, n_rows):
        if self.host_staging:
            return self.torch.empty(self.r.halo_bytes(n_rows), dtype=self.torch.uint8)
        return self._dev_empty(n_rows)

    def halo_export(self, res, row0, n_rows):
        t = self._dev_empty(n_rows)
        self.r.halo_pack(res, row0, n_rows, t.data_ptr())
        return t.cpu() if self.host_staging else t

    def halo_import(self, res, row0, n_rows, t):
        if self.host_staging:
            t = t.to(self.device)
        self.r.halo_unpack(res, row0, n_rows, t.data_ptr())
        if self.host_staging:
            self.torch.cuda.current_stream().synchronize()  # t is freed on return


def make_hip_strip(width, height, rank, world, triangles, eye, center, options, device_index=None):
    """Create this rank's context (+ StripFrame) on the current torch CUDA device."""
    import torch
    import torch.distributed as dist

    from . import api

    dev = torch.cuda.current_device() if device_index is None else device_index
    bounds = partition_rows(height, world)
    a, b = bounds[rank]
    stream = torch.cuda.current_stream().cuda_stream
    r = api.Renderer(width, height, device=dev, rows=(a, b), halo=HALO_ROWS if world > 1 else 0, stream=stream)
    r.set_scene(triangles)
    r.lookat(eye, center)
    r.set_options(options)
    host_staging = world > 1 and dist.get_backend() == "gloo"
    be = HipStripBackend(r, torch.device("cuda", dev), host_staging=host_staging)
    return r, StripFrame(be, bounds, rank, dist if world > 1 else None)


def run_frame_local(renderers, bounds, frame, device, halo=HALO_ROWS, staged=False):
    """Single-process variant for tests: several strip contexts on ONE GPU, halos moved through a
    device staging tensor with explicit stream syncs (no torch.distributed)."""
    import torch

    passes = int(renderers[0].options()["spatial_resampling_passes"][0])
    if staged:
        for r in renderers:
            r.frame_stage(frame, 0)
        for k in range(passes):
            for rank, r in enumerate(renderers):
                for peer, s0, sn, r0, rn in exchange_plan(bounds, rank, halo):
                    t = torch.empty(renderers[peer].halo_bytes(rn), dtype=torch.uint8, device=device)
                    renderers[peer].halo_pack(renderers[peer].frame_stage_input(k + 1), r0, rn, t.data_ptr())
                    renderers[peer].sync()
                    r.halo_unpack(r.frame_stage_input(k + 1), r0, rn, t.data_ptr())
                    r.sync()
            for r in renderers:
                r.frame_stage(frame, k + 1)
        for r in renderers:
            r.frame_stage(frame, passes + 1)
            r.sync()
        return None
    for r in renderers:
        r.raycast()
        r.generate_candidate(frame, RT_RES_0)
        r.temporal_resampling(frame, RT_RES_TEMPORAL, RT_RES_0)
        r.save_temporal_reservoir(RT_RES_0, RT_RES_TEMPORAL)
    src, dst = RT_RES_0, RT_RES_1
    for k in range(passes):
        if k != 0:
            src, dst = dst, src
        for rank, r in enumerate(renderers):
            for peer, s0, sn, r0, rn in exchange_plan(bounds, rank, halo):
                # what `rank` receives is what `peer` owns: rows [r0, r0+rn)
                t = torch.empty(renderers[peer].halo_bytes(rn), dtype=torch.uint8, device=device)
                renderers[peer].halo_pack(src, r0, rn, t.data_ptr())
                renderers[peer].sync()
                r.halo_unpack(src, r0, rn, t.data_ptr())
                r.sync()
        for r in renderers:
            r.spatial_resampling(frame, k, src, dst)
    for r in renderers:
        r.resolve(dst)
        r.tone_mapping()
        r.sync()
    return dst
