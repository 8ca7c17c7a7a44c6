"""Multi-rank row-strip logic (cedec_2024_rt_amd/strips.py) on CPU: world_size 2 and 3 over
`gloo`, the per-rank compute done by the oracle. Each rank only ever sees its own rows + the
87-row halos it received; everything else in its buffers is poisoned. The assembled N-rank
image must equal the 1-rank image bit for bit (SURVEY.md §8e determinism check)."""
import os
import socket

import numpy as np
import pytest


def test_halo_bound_and_partition():
    from cedec_2024_rt_amd import strips

    assert 86.4 < strips.halo_bound(30.0) < 86.5 and strips.HALO_ROWS == 87
    assert strips.partition_rows(1080, 8) == [(i * 135, (i + 1) * 135) for i in range(8)]
    b = strips.partition_rows(2160, 8)
    assert b[0] == (0, 270) and b[-1] == (1890, 2160)
    assert strips.partition_rows(10, 1) == [(0, 10)]
    with pytest.raises(ValueError):
        strips.partition_rows(1080, 16)  # 67-row strips < 87-row halo
    plan = strips.exchange_plan(b, 3)
    assert plan == [(2, 810, 87, 723, 87), (4, 993, 87, 1080, 87)]
    assert strips.exchange_plan(b, 0) == [(1, 183, 87, 270, 87)]
    # boundary rows (needed by a neighbour) are computed first, the interior while halos travel
    assert strips.row_bands(b, 3) == ([(810, 897), (993, 1080)], [(897, 993)])
    assert strips.row_bands(b, 0) == ([(183, 270)], [(0, 183)])
    assert strips.row_bands(strips.partition_rows(1080, 8), 3) == ([(405, 540)], [])  # 135-row strips: all boundary
    assert strips.row_bands([(0, 100)], 0) == ([], [(0, 100)])


class OracleBackend:
    """Checker backend for StripFrame: oracle kernels restricted to this rank's rows."""

    def __init__(self, ob, tris, W, H, rows, halo, eye, center, opt):
        import torch

        self.torch, self.ob, self.W, self.H = torch, ob, W, H
        self.a, self.b = rows
        self.l0, self.l1 = max(0, self.a - halo), min(H, self.b + halo)
        self.sc = ob.Scene(tris, use_bvh=True)
        self.rg = ob.raygen_lookat(eye, center, (0, 1, 0), np.float32(np.pi) / np.float32(4), W, H)
        self.eye = np.asarray(eye, np.float32)
        self.opt = opt
        self.passes = int(opt["spatial_resampling_passes"][0])
        st = ob.new_state(W, H)
        self.vis, self.accum, self.pixels = st["vis"], st["accum"], st["pixels"]
        self.res = [st["r0"], st["r1"], st["temporal"]]
        self.staged_temporal = []
        self.src, self.dst = 0, 1
        # poison everything this rank must never read: rows outside [l0, l1)
        for r in self.res:
            v = r.view(np.uint8).reshape(H, -1)
            v[: self.l0] = 0xFF
            v[self.l1:] = 0xFF
        self.vis["index"][: self.l0 * W] = -12345
        self.vis["index"][self.l1 * W:] = -12345

    # ---- StripFrame backend protocol (stages of the frame, restricted to row ranges)
    def stage_begin(self, frame, s, clear_first=False):
        if s == 0:
            if clear_first:
                self.ob.clear(self.accum, self.W, self.H)
            # visibility of the halo rows is recomputed locally (the oracle's spatial pass reads the
            # neighbour's Visibility; the HIP path keeps that bit inside the exchanged record)
            self.sc.raycast(self.W, self.H, self.rg, self.vis, rows=(self.l0, self.l1))
            self.src, self.dst = 0, 1
        elif 2 <= s <= self.passes:
            self.src, self.dst = self.dst, self.src

    def stage_run(self, frame, s, r0, r1):
        assert self.a <= r0 < r1 <= self.b
        W, H, rows = self.W, self.H, (r0, r1)
        if s == 0:
            self.sc.generate_candidate(W, H, frame, self.vis, self.eye, self.opt, self.res[0], rows=rows)
            self.sc.temporal_resampling(W, H, frame, self.vis, self.eye, self.opt, self.res[2], self.res[0], rows=rows)
            sl = slice(r0 * W, r1 * W)
            self.staged_temporal.append(sl)  # save_temporal_reservoir after ALL rows merged with the old history
        elif s <= self.passes:
            self.sc.spatial_resampling(W, H, frame, s - 1, self.vis, self.eye, self.opt, self.res[self.src], self.res[self.dst], rows=rows)
        else:
            self.sc.resolve(self.accum, W, H, self.vis, self.eye, self.opt, self.res[self.dst], rows=rows)

    # the two-lane protocol of StripFrame: interior rows are computed BEFORE the halos of the stage are
    # imported (on the device: on a second stream); here simply earlier, which checks the claim that
    # interior rows never read a halo row (the halo rows still hold the previous stage's data / poison)
    def stage_fork(self):
        pass

    def stage_run_async(self, frame, s, part, r0, r1):
        assert part == 0
        self.stage_run(frame, s, r0, r1)

    def stage_end(self, frame, s):
        if s == 0:
            for sl in self.staged_temporal:
                self.res[2][sl] = self.res[0][sl]
            self.staged_temporal = []

    def stage_output(self, s):
        return 0 if s == 0 else self.dst

    def halo_empty(self, n):
        return self.torch.empty(n * self.W * 76, dtype=self.torch.uint8)

    def halo_export(self, res, row0, n):
        assert self.a <= row0 and row0 + n <= self.b, "a rank may only export rows it owns"
        v = self.res[res][row0 * self.W:(row0 + n) * self.W]
        return self.torch.from_numpy(v.view(np.uint8).copy())

    def halo_import(self, res, row0, n, t):
        assert self.l0 <= row0 and row0 + n <= self.l1 and (row0 + n <= self.a or row0 >= self.b)
        self.res[res][row0 * self.W:(row0 + n) * self.W] = np.frombuffer(t.numpy().tobytes(), dtype=self.ob.RESERVOIR)


def _worker(rank, world, port, W, H, frames, q, native=False):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist

    from cedec_2024_rt_amd import scenes, strips
    from oracle import binding as ob

    torch.set_num_threads(1)
    ob.set_threads(2)
    ob.set_math_mode(ob.MATH_PORTABLE)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        tris = scenes.make_quad_room()
        eye, center = (0.5, 2.5, 6.0), (0.0, 1.5, -1.0)
        opt = ob.bench_options()
        bounds = strips.partition_rows(H, world)
        if native:
            # the partition and the row bands of the NATIVE strip driver (rt_mg_partition with a row cost ->
            # irregular strip heights, rt_mg_bands), driving the same exchange logic
            from cedec_2024_rt_amd import api

            cost = np.full(H, 40, np.uint32)
            cost[: H // 3] = 3  # a cheap band (sky-like rows): the first strip grows
            bounds = api.mg_partition(H, world, strips.HALO_ROWS, cost)
            assert bounds != strips.partition_rows(H, world) and all(e - a >= strips.HALO_ROWS for a, e in bounds)
        be = OracleBackend(ob, tris, W, H, bounds[rank], strips.HALO_ROWS, eye, center, opt)
        sf = strips.StripFrame(be, bounds, rank, strips.DistTransport(dist))
        if native:
            assert (sf.boundary, sf.interior) == api.mg_bands(bounds, rank)
        # single-rank truth
        sc = ob.Scene(tris, use_bvh=True)
        rg = ob.raygen_lookat(eye, center, (0, 1, 0), np.float32(np.pi) / np.float32(4), W, H)
        st = ob.new_state(W, H)
        a, b = bounds[rank]
        for f in range(1, frames + 1):
            sf.frame(f)
            sc.frame(W, H, f, rg, np.asarray(eye, np.float32), opt, st)
            mine = be.accum.reshape(H, W, 4)[a:b]
            want = st["accum"].reshape(H, W, 4)[a:b]
            if not np.array_equal(mine.view(np.uint32), want.view(np.uint32)):
                q.put((rank, f"frame {f}: {(mine != want).any(axis=2).sum()} pixels differ"))
                return
        # gather the strips on rank 0 and compare the assembled image
        parts = [None] * world
        dist.gather_object(be.accum.reshape(H, W, 4)[a:b].copy(), parts if rank == 0 else None, dst=0)
        if rank == 0:
            img = np.concatenate(parts, axis=0)
            ok = np.array_equal(img.view(np.uint32), st["accum"].reshape(H, W, 4).view(np.uint32))
            q.put((rank, "ok" if ok else "assembled image differs"))
        else:
            q.put((rank, "ok"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,native", [(2, 200, False), (3, 270, False), (3, 330, True)])
def test_strips_bit_identical_over_gloo(world, H, native):
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 40, H, 2, q, native)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    results = {}
    while not q.empty():
        r, msg = q.get()
        results[r] = msg
    for p in procs:
        assert p.exitcode == 0, f"worker exit {p.exitcode}: {results}"
    assert results == {r: "ok" for r in range(world)}, results
