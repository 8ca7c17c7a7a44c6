"""Native multi-GPU strip driver (cedec_2024_rt_amd/csrc/strip_mg.cpp, C-ABI rt_mg_*).

CPU (`-m "not gpu"`): the partition / band logic against the Python model of round 1 and a brute-force
optimum for the cost-weighted partition. GPU: N strip contexts of ONE process driven in lock-step
through the LOCAL transport (the development boxes have one GPU; RCCL refuses two ranks on one
device) must reproduce the single-context frame bit for bit — warm frames (halo plan prepared one frame
ahead, no host wait), cold frames (first frame, camera move, option change, frame-number jump), dense
and sparse halos, one and two lanes, BASELINE sizes 1920x1080 and 3840x2160 over 8 strips. The RCCL
code path itself is exercised by a one-rank communicator sending to itself.
"""
import itertools
import os

import numpy as np
import pytest


def _api():
    from cedec_2024_rt_amd import api

    return api


def test_partition_matches_python_model_and_bands():
    api = _api()
    from cedec_2024_rt_amd import strips

    for H, N in ((1080, 1), (1080, 2), (1080, 4), (1080, 8), (2160, 8), (1083, 5), (435, 5), (100, 1)):
        b = api.mg_partition(H, N)
        assert b == strips.partition_rows(H, N)
        for r in range(N):
            assert api.mg_bands(b, r) == strips.row_bands(b, r)
    with pytest.raises(ValueError):
        api.mg_partition(1080, 16)  # 67-row strips < 87-row halo
    # irregular (cost-weighted) bounds: bands still follow the halo rule
    b = [(0, 493), (493, 689), (689, 885), (885, 1080)]
    assert api.mg_bands(b, 0) == ([(406, 493)], [(0, 406)])
    assert api.mg_bands(b, 1) == ([(493, 580), (602, 689)], [(580, 602)])
    assert api.mg_bands(b, 3) == ([(885, 972)], [(972, 1080)])


def test_cost_weighted_partition_is_optimal():
    """rt_mg_partition(row_cost) minimises the most expensive strip subject to >= halo rows per strip:
    checked against brute force over all admissible cuts on small problems."""
    api = _api()
    rng = np.random.default_rng(7)
    for trial in range(40):
        H, N, halo = int(rng.integers(12, 40)), int(rng.integers(2, 5)), int(rng.integers(1, 4))
        if H // N < halo:
            continue
        cost = rng.integers(0, 50, H).astype(np.uint32)
        if trial % 3 == 0:
            cost[: H // 3] = 0  # a sky band
        b = api.mg_partition(H, N, halo, cost)
        assert b[0][0] == 0 and b[-1][1] == H and all(e - a >= halo for a, e in b)
        assert all(b[i][1] == b[i + 1][0] for i in range(N - 1))
        c1 = cost.astype(np.int64) + 1
        got = max(int(c1[a:e].sum()) for a, e in b)
        best = None
        for cuts in itertools.combinations(range(1, H), N - 1):
            edges = (0,) + cuts + (H,)
            if min(edges[i + 1] - edges[i] for i in range(N)) < halo:
                continue
            worst = max(int(c1[edges[i]:edges[i + 1]].sum()) for i in range(N))
            best = worst if best is None else min(best, worst)
        assert got == best, (H, N, halo, b, got, best)
    # a sky band of cheap rows makes the first strip taller at the benchmark size
    cost = np.full(1080, 1920, np.uint32)
    cost[:300] = 0
    b = api.mg_partition(1080, 4, 87, cost)
    assert b[0][1] - b[0][0] > 400 and all(e - a >= 87 for a, e in b)


def test_library_exports_strip_driver_symbols():
    api = _api()
    L = api.load_library()
    for sym in ("rt_mg_partition", "rt_mg_bands", "rt_mg_unique_id", "rt_mg_create", "rt_mg_frame", "rt_mg_frame_begin",
                "rt_mg_frame_step", "rt_mg_destroy", "rt_mg_get_stats", "rt_mg_selftest_rccl", "rt_lane", "rt_res_region"):
        assert hasattr(L, sym), sym


# ------------------------------------------------------------------------------------------ GPU
def _eq_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


class _Rig:
    """A single full-frame context and N strip contexts + native drivers over the LOCAL transport."""

    def __init__(self, api, tris, W, H, n, eye, at, optkw, flags=0, bounds=None):
        from cedec_2024_rt_amd.types import bench_options

        self.api, self.W, self.H = api, W, H
        self.bounds = bounds or api.mg_partition(H, n)
        self.opt = bench_options(**optkw)

        def make(rows=None, halo=0):
            r = api.Renderer(W, H, rows=rows, halo=halo)
            r.set_scene(tris)
            r.lookat(eye, at)
            r.set_options(self.opt)
            return r

        self.full = make()
        self.ctxs = [make(rows=b, halo=87) for b in self.bounds]
        self.hub = api.MgHub(len(self.bounds), renderer=self.ctxs[0])
        self.mgs = [api.MultiGpu(c, k, self.bounds, transport=api.RT_MG_TRANSPORT_LOCAL, hub=self.hub, flags=flags)
                    for k, c in enumerate(self.ctxs)]

    def everyone(self):
        return [self.full] + self.ctxs

    def frame(self, frame, clear_first=False):
        self.full.frame(frame, clear_first)
        self.api.mg_frame_lockstep(self.mgs, frame, clear_first)

    def check(self, what):
        api, W, H = self.api, self.W, self.H
        ref = self.full.download(api.RT_BUF_ACCUMULATION).reshape(H, W, 4)
        refpx = self.full.download(api.RT_BUF_PIXELS).reshape(H, W, 4)
        for c, (a, b) in zip(self.ctxs, self.bounds):
            acc = c.download(api.RT_BUF_ACCUMULATION).reshape(c.local_rows, W, 4)[a - c.local_row0: b - c.local_row0]
            assert _eq_bits(acc, ref[a:b]), f"{what}: rows {a}:{b}: {int((acc != ref[a:b]).any(axis=2).sum())} pixels differ"
            px = c.download(api.RT_BUF_PIXELS).reshape(c.local_rows, W, 4)[a - c.local_row0: b - c.local_row0]
            assert np.array_equal(px, refpx[a:b]), f"{what}: pixels of rows {a}:{b}"

    def check_history(self, what):
        """temporal history of the owned rows (what the next frame starts from)"""
        api, W, H = self.api, self.W, self.H
        ref = self.full.download(api.RT_BUF_RES_TEMPORAL).reshape(H, W)
        for c, (a, b) in zip(self.ctxs, self.bounds):
            mine = c.download(api.RT_BUF_RES_TEMPORAL).reshape(c.local_rows, W)[a - c.local_row0: b - c.local_row0]
            assert _eq_bits(mine, ref[a:b]), f"{what}: temporal history of rows {a}:{b}"

    def close(self):
        for m in self.mgs:
            m.close()
        self.hub.close()
        for c in self.everyone():
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,H,flags,optkw", [
    (2, 240, 0, dict()),
    (3, 300, 0, dict()),
    (3, 600, 0, dict()),                                         # strips with interior rows: two lanes
    (3, 600, 2, dict()),                                         # one lane
    (2, 400, 1, dict()),                                         # dense halos, in place
    (3, 600, 1, dict(spatial_resampling_passes=2)),
    (2, 240, 0, dict(spatial_resampling_passes=1)),              # the only exchange carries the next plan
    (2, 240, 0, dict(spatial_resampling_passes=0)),              # nothing to exchange
    (3, 330, 0, dict(use_shadowed_target_function=1, spatial_resampling_passes=2)),
    (2, 240, 0, dict(use_spatial_resampling=0)),
    (4, 700, 0, dict(accumulate=1, use_visibility_reuse=0, spatial_resampling_sample_count=3)),
    (3, 600, 4, dict()),                                         # sparse halos packed / unpacked by launches of their own (r02 form)
    (3, 330, 4, dict(use_shadowed_target_function=1, spatial_resampling_passes=2)),
    (3, 600, 4 | 2, dict()),
])
def test_native_strips_match_single_context(n, H, flags, optkw):
    """Frames 1..6 with a camera move before frame 4 (cold frame + clear) and a frame-number jump after
    frame 5: accumulation, pixels and temporal history of every strip == the single context, bit for bit."""
    api = _api()
    from cedec_2024_rt_amd import scenes

    tris = scenes.make_quad_room()
    W = 96
    rig = _Rig(api, tris, W, H, n, (0.5, 2.5, 6.0), (0.0, 1.5, -1.0), optkw, flags)
    seq = [1, 2, 3, 4, 5, 9]
    for i, frame in enumerate(seq):
        clear = False
        if frame == 4:
            for r in rig.everyone():
                r.orbit(35.0, -12.0)
                assert r.camera_updated()
            clear = True
        rig.frame(frame, clear)
        rig.check(f"frame {frame}")
        rig.check_history(f"frame {frame}")
    st = rig.mgs[0].stats()
    assert st["frames"] == len(seq)
    sparse = not (flags & 1) and rig.opt["spatial_resampling_passes"][0] > 0 and rig.opt["use_spatial_resampling"][0]
    if sparse:
        # cold: frame 1 (no plan), frame 4 (camera moved), frame 9 (not the frame the plan was made for)
        assert st["cold_frames"] == 3, st
        assert st["records_sent"] > 0
    else:
        assert st["cold_frames"] == 0
    assert sum(c.ray_count()[0] for c in rig.ctxs) == rig.full.ray_count()[0]
    rig.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("RT_MG_SEEDS", "6")))))
def test_native_strips_random(seed):
    """Random image sizes, 2-5 strips of irregular heights, random option sets and driver flags (dense / one lane / halo
    records packed by launches of their own or by the spatial passes), the blocks scene or the room, a camera move and an
    option change somewhere in the sequence: every frame of the native driver == the single context."""
    api = _api()
    from cedec_2024_rt_amd import scenes
    from cedec_2024_rt_amd.types import bench_options

    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(2, 6))
    cuts = np.sort(rng.integers(0, 120, size=n))
    H = int(87 * n + cuts.sum() + rng.integers(0, 60))
    extra = H - 87 * n
    # irregular strip heights, each >= 87 rows
    parts = rng.multinomial(extra, np.ones(n) / n)
    edges = np.concatenate([[0], np.cumsum(87 + parts)])
    bounds = [(int(edges[i]), int(edges[i + 1])) for i in range(n)]
    W = int(rng.integers(40, 200))
    flags = int(rng.choice([0, 0, 0, 1, 2, 4, 6]))

    def options():
        return dict(use_temporal_resampling=int(rng.integers(0, 2)), use_visibility_reuse=int(rng.integers(0, 2)),
                    use_shadowed_target_function=int(rng.integers(0, 3) == 0), ris_sample_count=int(rng.integers(1, 9)),
                    spatial_resampling_passes=int(rng.integers(1, 4)), spatial_resampling_sample_count=int(rng.integers(1, 6)),
                    accumulate=int(rng.integers(0, 2)))

    if rng.integers(0, 2):
        tris, eye, at = scenes.make_quad_room(), (0.5 + float(rng.normal()) * 0.4, 2.5, 6.0), (0.0, 1.5, -1.0)
    else:
        tris, eye, at = scenes.make_blocks_restir(), scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT
    optkw = options()
    rig = _Rig(api, tris, W, H, n, eye, at, optkw, flags, bounds=bounds)
    move_at, change_at = int(rng.integers(2, 7)), int(rng.integers(2, 7))
    for frame in range(1, 8):
        clear = False
        if frame == move_at:
            for r in rig.everyone():
                r.orbit(float(20 + 10 * seed), -9.0)
            clear = True
        if frame == change_at:
            new = bench_options(**options())
            for r in rig.everyone():
                r.set_options(new)
        rig.frame(frame, clear)
        what = f"seed {seed}: {n} strips {bounds} of {W}x{H}, flags {flags}, frame {frame} (move at {move_at}, options at {change_at})"
        rig.check(what)
        rig.check_history(what)
    assert sum(c.ray_count()[0] for c in rig.ctxs) == rig.full.ray_count()[0]
    rig.close()


@pytest.mark.gpu
def test_native_strips_option_change_and_irregular_bounds():
    """Cost-weighted (irregular) strip heights and an option change between frames (new plan)."""
    api = _api()
    from cedec_2024_rt_amd import scenes
    from cedec_2024_rt_amd.types import bench_options

    tris = scenes.make_quad_room()
    W, H = 80, 520
    bounds = [(0, 200), (200, 300), (300, 520)]
    rig = _Rig(api, tris, W, H, 3, (0.5, 2.5, 6.0), (0.0, 1.5, -1.0), dict(), 0, bounds=bounds)
    for frame in (1, 2, 3):
        rig.frame(frame)
        rig.check(f"frame {frame}")
    new = bench_options(spatial_resampling_passes=2, spatial_resampling_radius=20.0)
    for r in rig.everyone():
        r.set_options(new)
    for frame in (4, 5):
        rig.frame(frame)
        rig.check(f"frame {frame} after the option change")
    assert rig.mgs[1].stats()["cold_frames"] == 2
    rig.close()


@pytest.mark.gpu
def test_halo_mark_quick_reject_marks_the_same_records():
    """rt_tuning keys 18 and 19: the quick reject of k_halo_mark (rows far from a neighbour's region test the pass's first draws
    against a bound on the neighbour distance before replaying log / sqrt / sincos) marks exactly the records of the full
    replay: need-bitmaps, counts and prefix words of both sides, three passes, four frames, a 270-row strip of the
    benchmark frame and a 135-row one (bands that meet)."""
    import ctypes as C

    import torch

    api = _api()
    from cedec_2024_rt_amd import scenes
    from cedec_2024_rt_amd.types import bench_options

    L = api.load_library()
    tris = scenes.make_blocks_restir()
    W, H = 1920, 1080
    for n in (4, 8):
        bounds = api.mg_partition(H, n)
        ctxs = []
        for k in (0, 1, 2):
            c = api.Renderer(W, H, rows=bounds[k], halo=87)
            c.set_scene(tris)
            c.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
            c.set_options(bench_options())
            c.raycast()
            ctxs.append(c)
        mid = ctxs[1]
        a, b = bounds[1]
        buf = torch.zeros(87 * W + 64, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for src, row0 in ((ctxs[0], a - 87), (ctxs[2], b)):
            src.halo_flags_pack(row0, 87, buf.data_ptr())
            src.sync()
            mid.halo_flags_unpack(row0, 87, buf.data_ptr())
            mid.sync()
        words = mid.halo_bitmap_words(87)
        marked = 0
        for frame in (1, 2, 3, 4):
            got = []
            for quick, window in ((0, 0), (1, 0), (0, 1), (1, 1)):  # key 19 (r04): marks collected in an LDS window per workgroup
                mid.tuning(18, quick)
                mid.tuning(19, window)
                bm = torch.full((2, 3 * words), -1, dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()  # torch fills on its own stream; the context marks on its non-blocking stream
                rc = L.rt_halo_mark_sides(mid.h, frame, 0, 3, C.c_void_p(bm[0].data_ptr()), C.c_void_p(bm[1].data_ptr()))
                assert rc == 0, mid.last_error() if hasattr(mid, "last_error") else rc
                mid.sync()
                got.append(bm.cpu().numpy().copy())
            for k in (1, 2, 3):
                assert np.array_equal(got[0], got[k]), f"{n} strips, frame {frame}, variant {k}: {(got[0] != got[k]).sum()} words differ"
            marked += int(got[0][0, 0]) + int(got[0][1, 0])
        assert marked > 10000, marked
        for c in ctxs:
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", [(1920, 1080), (3840, 2160)])  # BASELINE configs #4 and #5, the 8-GPU partition
def test_native_strips_full_size(W, H):
    api = _api()
    from cedec_2024_rt_amd import scenes

    tris = scenes.make_blocks_restir()
    rig = _Rig(api, tris, W, H, 8, scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT, dict())
    for frame in (1, 2, 3):
        rig.frame(frame)
        rig.check(f"{W}x{H} frame {frame}")
    st = [m.stats() for m in rig.mgs]
    assert all(s["cold_frames"] == 1 for s in st)
    dense = 87 * W * 3 * 3  # records of a dense exchange per side, 3 passes, 3 frames
    mid = st[3]
    frac = mid["records_sent"] / (2 * dense)
    print(f"{W}x{H}: sparse halos move {frac:.3f} of the dense band; host {mid['host_ns'] / mid['frames'] / 1e3:.0f} us/frame (lock-step, 8 ranks in one process)")
    assert 0.05 < frac < 0.5
    assert sum(c.ray_count()[0] for c in rig.ctxs) == rig.full.ray_count()[0]
    rig.close()


@pytest.mark.gpu
def test_rccl_path_one_rank_selftest():
    """dlopen(RCCL) + ncclCommInitRank + grouped ncclSend/ncclRecv on a stream, on a communicator of one rank."""
    api = _api()
    L = api.load_library()
    assert len(api.mg_unique_id()) == 128
    for nbytes in (80, 4096, 5 * 1024 * 1024):
        rc = L.rt_mg_selftest_rccl(nbytes)
        assert rc == 0, (nbytes, rc, L.rt_mg_load_error())


@pytest.mark.gpu
def test_native_single_rank_driver_equals_rt_frame():
    """world = 1: rt_mg_frame is rt_frame."""
    api = _api()
    from cedec_2024_rt_amd import scenes
    from cedec_2024_rt_amd.types import bench_options

    tris = scenes.make_quad_room()
    W, H = 64, 48
    a, b = api.Renderer(W, H), api.Renderer(W, H)
    for r in (a, b):
        r.set_scene(tris)
        r.lookat((0.5, 2.5, 6.0), (0.0, 1.5, -1.0))
        r.set_options(bench_options())
    mg = api.MultiGpu(b, 0, [(0, H)])
    for frame in (1, 2, 3):
        a.frame(frame)
        mg.frame(frame)
        assert _eq_bits(a.download(api.RT_BUF_ACCUMULATION), b.download(api.RT_BUF_ACCUMULATION))
    mg.close()
    a.close()
    b.close()


@pytest.mark.gpu
def test_multi_process_host_app_shm_equals_single_process(tmp_path):
    """`restir_app --ranks 3 --shm`: three PROCESSES (forked before any HIP call) sharing this box's one GPU, strips
    exchanged through the host-staged shared-memory transport, cost-weighted strip heights, every rank writing its rows
    into one PFM — byte-identical to the single-process image of the same frames. (The RCCL transport needs one GPU per
    rank; this runs the same process-level code with a transport that works on one.)"""
    import subprocess

    from cedec_2024_rt_amd import scenes

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    app = os.path.join(root, "app", "restir_app")
    if not os.path.exists(app):
        pytest.skip("app/restir_app not built")
    tris = scenes.make_quad_room()
    tpath = str(tmp_path / "scene.tris")
    tris.tofile(tpath)
    common = ["--tris", tpath, "--size", "160", "330", "--frames", "5", "--eye", "0.5", "2.5", "6.0", "--lookat", "0.0", "1.5", "-1.0"]
    one, many = str(tmp_path / "one.pfm"), str(tmp_path / "many.pfm")
    subprocess.check_call([app] + common + ["--pfm", one], stdout=subprocess.DEVNULL, timeout=120)
    a = open(one, "rb").read()
    # cost-weighted heights (round 2's shaded-pixel model), explicit irregular bounds (a measured cut), equal rows (the default)
    for extra in (["--cost-strips"], ["--bounds", "0,95,230,330"], []):
        out = subprocess.check_output([app] + common + ["--pfm", many, "--ranks", "3", "--shm"] + extra, timeout=300).decode()
        assert "3 ranks:" in out and out.count("cold frame") == 3, out
        if extra[:1] == ["--bounds"]:
            assert "rows [0,95)" in out and "rows [95,230)" in out and "rows [230,330)" in out, out
        b = open(many, "rb").read()
        assert len(a) == len(b) and a == b, extra


def test_cached_strip_cuts_are_valid_partitions():
    """profiles/strip_cuts.json (written by tools/strip_overhead.py, read by bench.py instead of start-up balance rounds): every
    entry is a partition of the image rows into N strips of at least the 87-row halo, keyed by size, N and the scene's hash."""
    import json
    import re

    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "strip_cuts.json")
    if not os.path.exists(p):
        pytest.skip("no cached cuts committed")
    cuts = json.load(open(p))
    assert cuts
    for key, c in cuts.items():
        m = re.fullmatch(r"(\d+)x(\d+):(\d+):([0-9a-f]{16})", key)
        assert m, key
        H, n = int(m.group(2)), int(m.group(3))
        e = c["bounds"]
        assert len(e) == n + 1 and e[0] == 0 and e[-1] == H, key
        assert all(e[i + 1] - e[i] >= 87 for i in range(n)), key
        assert c["max_ms"] <= c["equal_rows_max_ms"] * 1.05, key  # a cut that is worse than equal rows should not be cached
