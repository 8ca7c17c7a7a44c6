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
import os

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


def row_bands(bounds, rank, halo=HALO_ROWS):
    """(boundary, interior): owned row ranges whose results a neighbour needs (computed and sent
    first) and the rest (computed while the halos travel)."""
    a, b = bounds[rank]
    cuts = []
    if rank > 0:
        cuts.append((a, min(a + halo, b)))
    if rank + 1 < len(bounds):
        cuts.append((max(b - halo, a), b))
    if not cuts:
        return [], [(a, b)]
    cuts.sort()
    boundary = [cuts[0]]
    for r0, r1 in cuts[1:]:
        if r0 <= boundary[-1][1]:
            boundary[-1] = (boundary[-1][0], max(boundary[-1][1], r1))
        else:
            boundary.append((r0, r1))
    interior, cur = [], a
    for r0, r1 in boundary:
        if cur < r0:
            interior.append((cur, r0))
        cur = r1
    if cur < b:
        interior.append((cur, b))
    return boundary, interior


class DistTransport:
    """torch.distributed point-to-point (RCCL with backend "nccl", gloo in the CPU tests)."""

    def __init__(self, dist):
        self.d = dist

    def post(self, rank, items):
        """items: [(peer, send_tensor, recv_tensor)] -> handle"""
        d = self.d
        ops = []
        for peer, ts, tr in items:
            ops.append(d.P2POp(d.isend, ts, peer))
            ops.append(d.P2POp(d.irecv, tr, peer))
        return d.batch_isend_irecv(ops) if ops else []

    def finish(self, rank, handle, items):
        for w in handle:
            w.wait()


class LocalTransport:
    """In-process mailbox for several strip contexts driven in lock-step (tests on one GPU)."""

    def __init__(self):
        self.box = {}
        self.sent = {}

    def post(self, rank, items):
        # messages of successive stages between the same pair are told apart by a sequence number
        seqs = []
        for peer, ts, tr in items:
            n = self.sent.get((rank, peer), 0)
            self.sent[(rank, peer)] = n + 1
            self.box[(rank, peer, n)] = ts
            seqs.append(n)
        return seqs

    def finish(self, rank, handle, items):
        for (peer, ts, tr), n in zip(items, handle):
            tr.copy_(self.box.pop((peer, rank, n)))
        if items and items[0][2].is_cuda:
            # the copies ran on torch's stream, the contexts unpack on their own streams
            import torch

            torch.cuda.synchronize()


class StripFrame:
    """One rank's frame loop: examples/10_restir_di/10_restir_di.cpp:257-379 on its strip.

    Stage 0 = [clear,] raycast, generate_candidate, temporal_resampling; stage k = spatial pass k-1;
    stage passes+1 = resolve + tone_mapping. Each producing stage computes the boundary rows first,
    posts the halo exchange of what it just wrote, then computes the interior rows while the halos
    travel; the next stage waits for them. Backend protocol: passes, stage_begin(frame, s, clear),
    stage_run(frame, s, r0, r1), stage_end(frame, s), stage_output(s) -> buffer handle,
    halo_export(buf, row0, n) -> tensor, halo_import(buf, row0, n, tensor), halo_empty(n) -> tensor.
    """

    def __init__(self, backend, bounds, rank, transport=None, halo=HALO_ROWS, sparse=None):
        self.b, self.bounds, self.rank, self.t, self.halo = backend, bounds, rank, transport, halo
        self.plan = exchange_plan(bounds, rank, halo) if transport is not None else []
        self.boundary, self.interior = row_bands(bounds, rank, halo) if self.plan else ([], [bounds[rank]])
        # sparse halos: only the records the neighbour's RNG will gather travel (needs backend support)
        self.sparse = bool(getattr(backend, "sparse_supported", False)) if sparse is None else bool(sparse)
        self.need, self.give, self.need_cnt, self.give_cnt = {}, {}, {}, {}

    def _post(self, buf, k=0):
        items, recvs = [], []
        for peer, s0, sn, r0, rn in self.plan:
            if self.sparse:
                ts = self.b.pack_sparse(buf, s0, sn, self.give[peer][k], self.give_cnt[peer][k])
                tr = self.b.list_empty(self.need_cnt[peer][k])
            else:
                ts = self.b.halo_export(buf, s0, sn)
                tr = self.b.halo_empty(rn)
            items.append((peer, ts, tr))
            recvs.append((peer, r0, rn, tr))
        return (self.t.post(self.rank, items), items, recvs, buf, k)

    def _finish(self, pending):
        handle, items, recvs, buf, k = pending
        self.t.finish(self.rank, handle, items)
        for peer, r0, rn, tr in recvs:
            if self.sparse:
                self.b.unpack_sparse(buf, r0, rn, self.need[peer][k], tr)
            else:
                self.b.halo_import(buf, r0, rn, tr)

    # ---- sparse halo set-up of one frame (generator: yields once, after posting the bitmaps)
    def _sparse_setup_gen(self, frame):
        b, P = self.b, self.b.passes
        # 1. shaded flags of the neighbours' boundary rows (static for the frame)
        items = [(peer, b.flags_export(s0, sn), b.flags_empty(rn)) for peer, s0, sn, r0, rn in self.plan]
        h = self.t.post(self.rank, items)
        yield "flags"
        self.t.finish(self.rank, h, items)
        for (peer, s0, sn, r0, rn), (_, _, tr) in zip(self.plan, items):
            b.flags_import(r0, rn, tr)
        # 2. what I will gather from each neighbour, per pass (RNG replay on the device)
        items = []
        for peer, s0, sn, r0, rn in self.plan:
            side = 0 if peer < self.rank else 1
            self.need[peer] = b.mark_all(frame, P, side)
            ts = b.bitmap_message(self.need[peer], rn)
            tr = b.bitmap_message_empty(P, sn)
            items.append((peer, ts, tr))
        self.need_cnt = {peer: b.bitmap_counts(self.need[peer]) for peer in self.need}  # host sync
        h = self.t.post(self.rank, items)
        yield "bitmaps"
        self._bitmap_pending = (h, items)

    def _sparse_setup_finish(self):
        b, P = self.b, self.b.passes
        h, items = self._bitmap_pending
        self.t.finish(self.rank, h, items)
        for (peer, s0, sn, r0, rn), (_, _, tr) in zip(self.plan, items):
            self.give[peer] = b.bitmap_split(tr, P, sn)  # scanned on the device
            self.give_cnt[peer] = b.bitmap_counts(self.give[peer])  # host sync

    def frame_gen(self, frame, clear_first=False):
        """Generator: yields after every posted exchange (lets a single-process driver interleave
        several ranks); `frame` just exhausts it."""
        b, P = self.b, self.b.passes
        pending = None
        sparse = self.sparse and self.plan and P > 0
        two_lanes = bool(self.plan) and hasattr(b, "stage_run_async") and not os.environ.get("RT_ONE_LANE")
        for s in range(0, P + 1):
            b.stage_begin(frame, s, clear_first if s == 0 else False)
            part = 0
            if s == 0 and sparse:
                a0, e0 = self.bounds[self.rank]
                b.stage_run_part(frame, 0, 1, a0, e0)  # [clear,] raycast of all owned rows
                part = 2  # only generate_candidate(+temporal) is left
                if two_lanes:
                    b.stage_fork()  # the second lane needs the G-buffer
            if two_lanes:
                # interior rows never read halo rows (they are >= halo rows away from the neighbours):
                # they start at once on the second lane and fill the GPU while the main
                # lane waits for halos, computes the boundary rows and packs them
                for r0, r1 in self.interior:
                    b.stage_run_async(frame, s, part, r0, r1)
            if pending is not None:
                self._finish(pending)
                pending = None
            if s == 0 and sparse:
                for tag in self._sparse_setup_gen(frame):
                    yield tag
            for r0, r1 in self.boundary:
                b.stage_run_part(frame, s, part, r0, r1) if part else b.stage_run(frame, s, r0, r1)
            if s < P and self.plan:
                if s == 0 and sparse:
                    self._sparse_setup_finish()
                pending = self._post(b.stage_output(s), s)
                yield s
            if not two_lanes:
                for r0, r1 in self.interior:
                    b.stage_run_part(frame, s, part, r0, r1) if part else b.stage_run(frame, s, r0, r1)
            b.stage_end(frame, s)
        if pending is not None:  # (P == 0: nothing was posted)
            self._finish(pending)
        a, e = self.bounds[self.rank]
        b.stage_begin(frame, P + 1, False)
        b.stage_run(frame, P + 1, a, e)
        b.stage_end(frame, P + 1)

    def frame(self, frame, clear_first=False):
        for _ in self.frame_gen(frame, clear_first):
            pass


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

    def stage_begin(self, frame, s, clear_first=False):
        self.r.frame_stage_begin(frame, s, clear_first)

    def stage_run(self, frame, s, r0, r1):
        self.r.frame_stage_run(frame, s, r0, r1)

    def stage_end(self, frame, s):
        self.r.frame_stage_end(s)

    def stage_output(self, s):
        return self.r.frame_stage_output(s)

    def stage_run_part(self, frame, s, part, r0, r1):
        self.r.frame_stage_run_part(frame, s, part, r0, r1)

    def stage_fork(self):
        self.r.frame_stage_fork()

    def stage_run_async(self, frame, s, part, r0, r1):
        self.r.frame_stage_run_async(frame, s, part, r0, r1)

    # ---- sparse halos (receiver-marked bitmaps, see include/restir_rt.h)
    sparse_supported = True

    def _to_wire(self, t):
        return t.cpu() if self.host_staging else t

    def _from_wire(self, t):
        return t.to(self.device) if self.host_staging else t

    def _wire_empty(self, n, dtype):
        return self.torch.empty(n, dtype=dtype, device=None if self.host_staging else self.device)

    def flags_empty(self, n_rows):
        return self._wire_empty(self.r.halo_flags_bytes(n_rows), self.torch.uint8)

    def flags_export(self, row0, n_rows):
        t = self.torch.empty(self.r.halo_flags_bytes(n_rows), dtype=self.torch.uint8, device=self.device)
        self.r.halo_flags_pack(row0, n_rows, t.data_ptr())
        return self._to_wire(t)

    def flags_import(self, row0, n_rows, t):
        t = self._from_wire(t)
        self.r.halo_flags_unpack(row0, n_rows, t.data_ptr())
        self._keep = t

    def mark_all(self, frame, passes, side):
        """need-bitmaps of all spatial passes of the frame for the neighbour on `side` (one launch)"""
        r0, n = self._side_region(side)
        words = self.r.halo_bitmap_words(n)
        t = self.torch.empty(passes * words, dtype=self.torch.int32, device=self.device)
        self.r.halo_mark(frame, 0, passes, side, t.data_ptr())
        return [t[k * words: (k + 1) * words] for k in range(passes)]

    def _side_region(self, side):
        a, b = self.r.rows
        l0, l1 = self.r.local_row0, self.r.local_row0 + self.r.local_rows
        return (l0, a - l0) if side == 0 else (b, l1 - b)

    def bitmap_counts(self, bitmaps):
        return [int(x) for x in self.torch.stack([bm[0] for bm in bitmaps]).cpu().tolist()]

    def bitmap_message(self, bitmaps, n_rows):
        nw = (self.r.halo_bitmap_words(n_rows) - 1) // 2
        return self._to_wire(self.torch.cat([bm[: 1 + nw] for bm in bitmaps]))

    def bitmap_message_empty(self, passes, n_rows):
        nw = (self.r.halo_bitmap_words(n_rows) - 1) // 2
        return self._wire_empty(passes * (1 + nw), self.torch.int32)

    def bitmap_split(self, msg, passes, n_rows):
        msg = self._from_wire(msg)
        words = self.r.halo_bitmap_words(n_rows)
        nw = (words - 1) // 2
        t = self.torch.zeros((passes, words), dtype=self.torch.int32, device=self.device)
        t[:, : 1 + nw] = msg.view(passes, 1 + nw)
        self.r.halo_scan(n_rows, passes, t.data_ptr())
        return [t[k] for k in range(passes)]

    def list_empty(self, count):
        return self._wire_empty(max(count, 1) * 80, self.torch.uint8)

    def pack_sparse(self, res, row0, n_rows, bitmap, count):
        t = self.torch.empty(max(count, 1) * 80, dtype=self.torch.uint8, device=self.device)
        self.r.halo_pack_sparse(res, row0, n_rows, bitmap.data_ptr(), t.data_ptr())
        return self._to_wire(t)

    def unpack_sparse(self, res, row0, n_rows, bitmap, t):
        t = self._from_wire(t)
        self.r.halo_unpack_sparse(res, row0, n_rows, bitmap.data_ptr(), t.data_ptr())
        self._keep2 = t
        if self.host_staging:
            self.torch.cuda.current_stream().synchronize()

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
    return r, StripFrame(be, bounds, rank, DistTransport(dist) if world > 1 else None)


def run_frame_local(renderers, bounds, frame, device, halo=HALO_ROWS, sparse=False, stats=None):
    """Single-process variant for tests: several strip contexts on ONE GPU driven in lock-step
    through the same StripFrame logic, halos moved through an in-process mailbox."""
    import torch

    class _Sync(HipStripBackend):
        # contexts have their own streams here: make every hand-off visible before the peer reads it
        def halo_export(self, res, row0, n_rows):
            t = super().halo_export(res, row0, n_rows)
            self.r.sync()
            return t

        def halo_import(self, res, row0, n_rows, t):
            super().halo_import(res, row0, n_rows, t)
            self.r.sync()

        def flags_export(self, row0, n_rows):
            t = super().flags_export(row0, n_rows)
            self.r.sync()
            return t

        def flags_import(self, row0, n_rows, t):
            super().flags_import(row0, n_rows, t)
            self.r.sync()

        def mark_all(self, frame, passes, side):
            t = super().mark_all(frame, passes, side)
            self.r.sync()
            return t

        def bitmap_split(self, msg, passes, n_rows):
            out = super().bitmap_split(msg, passes, n_rows)
            self.r.sync()
            return out

        def pack_sparse(self, res, row0, n_rows, bitmap, count):
            t = super().pack_sparse(res, row0, n_rows, bitmap, count)
            self.r.sync()
            if stats is not None:
                stats.append((count, n_rows * self.r.W))
            return t

        def unpack_sparse(self, res, row0, n_rows, bitmap, t):
            super().unpack_sparse(res, row0, n_rows, bitmap, t)
            self.r.sync()

    tr = LocalTransport()
    # as in the product path, the contexts enqueue on torch's current stream: the torch ops of the
    # backend (message assembly, zero-fills, copies) and the contexts' kernels are then ordered
    for r in renderers:
        r.sync()
        r.set_stream(torch.cuda.current_stream().cuda_stream)
    frames = [StripFrame(_Sync(r, device), bounds, k, tr, halo, sparse=sparse) for k, r in enumerate(renderers)]
    gens = [f.frame_gen(frame) for f in frames]
    live = list(range(len(gens)))
    while live:
        for k in list(live):
            try:
                next(gens[k])
            except StopIteration:
                live.remove(k)
    for r in renderers:
        r.sync()
        r.set_stream_own()
