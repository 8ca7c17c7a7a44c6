"""Round-5 GPU tests: every new launch form against the ORACLE (not against another setting of the product) on the frames
bench.py times, plus form-against-form checks of buffers the oracle has no opinion on.

* rt_tuning 20: resolve tone-maps its own pixel (common/kernels/common.cu:30-74 behind 10_restir_di.cu:451-458);
* rt_tuning 23: the LAST spatial pass + resolve in one kernel (10_restir_di.cu:256-459 for one pixel), with and without the
  pass's own stores, whole frames and three LOCAL strips;
* rt_tuning 22: the look-ahead stage 0 free of the main stream and of the latest resolve (three G-buffer sets, five reservoir
  buffers), forced on a whole-frame context (strips have it by default: tests/test_mg_native.py, test_gpu_round4.py);
* rt_tuning 21: the halo marks with and without the cached shaded-bit rows: same bitmaps, also across a camera move;
* rt_tuning 0-3 = 2 ... 7: the tile orders that interleave the XCDs (results never depend on the order); 8 = 4: the pass in
  one-wavefront workgroups;
* rt_tuning 25: stage 0 of the staged frame as ONE launch (raycast inside the candidates' kernel, 10_restir_di.cu:9-135);
* rt_tuning 24: raycast at half density — 32 primary rays and 32 rayless helper lanes per wavefront (10_restir_di.cu:9-34);
* RT_MG_TRANSPORT_WIRE_MODEL moves what MIRROR moves and holds the stream for the modelled time.
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FOVY = np.float32(np.pi) / np.float32(4)


def _eq_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def _res_bad(a, b, mask):
    bad = []
    for f in a.dtype.names:
        if f == "pad":
            continue
        if not _eq_bits(np.ascontiguousarray(a[f][mask]), np.ascontiguousarray(b[f][mask])):
            bad.append(f)
    return bad


@pytest.fixture(scope="module")
def api():
    from cedec_2024_rt_amd import api as _api

    return _api


@pytest.fixture(scope="module")
def scenes():
    from cedec_2024_rt_amd import scenes as s

    return s


def _oracle_frames(oracle, tris, W, H, eye, at, frames, **optkw):
    oracle.set_math_mode(oracle.MATH_PORTABLE)
    sc = oracle.Scene(tris, use_bvh=True)
    rg = oracle.raygen_lookat(eye, at, (0, 1, 0), FOVY, W, H)
    st = oracle.new_state(W, H)
    opt = oracle.bench_options(**optkw)
    eyev = np.asarray(eye, np.float32)
    for f in range(1, frames + 1):
        sc.frame(W, H, f, rg, eyev, opt, st, None)
    shaded = (st["vis"]["index"] >= 0) & ~np.isin(st["vis"]["index"], sc.lights)
    return st, shaded


@pytest.mark.parametrize("W,H,frames,tuning,optkw", [
    (480, 270, 12, {23: 1}, {}),                  # last pass + resolve fused, the pass's records stored
    (480, 270, 12, {23: 2}, {}),                  # ... records kept in registers (A/B form)
    (480, 270, 12, {23: 1, 20: 0}, {}),           # ... with the reference's separate tone_mapping launch
    (480, 270, 12, {23: 1, 14: 0, 17: 0}, {}),    # ... frames back to back on one stream (the headline's form)
    (480, 270, 7, {23: 1}, {"accumulate": 1}),    # ... accumulating (resolve reads the buffer it adds to)
    (480, 270, 5, {23: 1}, {"spatial_resampling_passes": 1}),
    (480, 270, 5, {23: 1}, {"spatial_resampling_passes": 2}),
    (480, 270, 12, {22: 1}, {}),                  # free-running look-ahead on a whole frame
    (480, 270, 12, {22: 1, 23: 1}, {}),
    (480, 270, 12, {20: 0}, {}),                  # the reference's two tail launches
    (1920, 1080, 6, {23: 1, 22: 1}, {}),          # the benchmark's own size
    (480, 270, 6, {24: 1}, {}),                   # half-density raycast (32 rays + 32 helper lanes per wavefront)
    (1920, 1080, 3, {24: 1, 22: 1}, {}),
    (480, 270, 5, {0: 2, 1: 3, 2: 2, 3: 3}, {}),  # r05 tile orders: tile rows interleaved over the XCDs, row- / column-major
    (480, 270, 5, {0: 4, 1: 5, 2: 4, 3: 5}, {}),  # ... tile b on XCD b % 8, row by row / in stripes of 32 tiles
    (480, 270, 5, {0: 6, 1: 7, 2: 6, 3: 7}, {}),  # ... runs of 4 / 16 tiles per XCD
    (500, 277, 4, {0: 5, 1: 6, 2: 7, 3: 2}, {}),  # ... on an image whose sides are no multiple of the tile
    (480, 270, 5, {0: 0, 1: 1, 2: 0, 3: 1}, {}),  # the band orders of r01-r04
    (480, 270, 4, {2: 7}, {"use_shadowed_target_function": 1}),
    (480, 270, 5, {8: 4}, {}),                    # the pass as one-wavefront workgroups on 8 x 8 tiles (A/B form)
    (1920, 1080, 3, {0: 7, 1: 4, 2: 5, 3: 6}, {}),
    (480, 270, 8, {25: 0}, {}),                   # stage 0 as two launches (r01-r04) ...
    (480, 270, 8, {25: 1}, {}),                   # ... and as one: primary ray, then candidates + temporal merge (the default)
    (480, 270, 8, {25: 1, 14: 0, 17: 0}, {}),     # ... frames back to back on one stream (the headline's form)
    (500, 277, 5, {25: 1, 22: 1}, {}),
    (480, 270, 6, {25: 1}, {"accumulate": 1}),
    (480, 270, 4, {25: 1}, {"use_temporal_resampling": 0}),   # no temporal merge: the one-launch form does not apply, two launches run
    (480, 270, 3, {25: 1}, {"use_shadowed_target_function": 1}),
    (1920, 1080, 4, {25: 1}, {}),
])
def test_new_launch_forms_vs_oracle(api, oracle, scenes, W, H, frames, tuning, optkw):
    """blocks_restir, bench options, frames enqueued back to back with no sync in between; accumulation, pixels and the temporal
    history of the last frame == the oracle's"""
    from cedec_2024_rt_amd.types import bench_options

    tris = scenes.make_blocks_restir()
    eye, at = scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT
    r = api.Renderer(W, H, exp=bool({23, 24} & set(tuning)) or tuning.get(8) == 4)  # A/B forms: librestir_rt_exp.so
    for k, v in tuning.items():
        r.tuning(k, v)
    r.set_scene(tris)
    r.lookat(eye, at)
    r.set_options(bench_options(**optkw))
    for f in range(1, frames + 1):
        r.frame(f)
    st, shaded = _oracle_frames(oracle, tris, W, H, eye, at, frames, **optkw)
    acc = r.download(api.RT_BUF_ACCUMULATION)
    ref = st["accum"].reshape(acc.shape)
    nbad = int((acc.view(np.uint32) != ref.view(np.uint32)).any(axis=1).sum())
    assert nbad == 0, f"frame {frames}: {nbad} pixels differ from the oracle"
    assert np.array_equal(r.download(api.RT_BUF_PIXELS).reshape(H, W, 4), st["pixels"])
    bad = _res_bad(r.download(api.RT_BUF_RES_TEMPORAL), st["temporal"], shaded)
    assert not bad, f"temporal history after frame {frames}: {bad}"
    r.close()


def test_fused_final_pass_writes_the_reference_buffers(api, scenes):
    """rt_tuning 23 = 1 keeps every buffer a caller can download identical to the two-kernel form: the final reservoirs the last
    pass writes (RT_BUF_RES_0 / RT_BUF_RES_1 after rt_frame), frame after frame"""
    from cedec_2024_rt_amd.types import bench_options

    W, H = 320, 200
    tris = scenes.make_blocks_restir()
    rs = []
    for fused in (0, 1):
        r = api.Renderer(W, H, exp=True)
        r.tuning(23, fused)
        r.set_scene(tris)
        r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
        r.set_options(bench_options())
        rs.append(r)
    for f in range(1, 6):
        finals = [r.frame(f) for r in rs]
        assert finals[0] == finals[1]
        for buf in (api.RT_BUF_RES_0, api.RT_BUF_RES_1, api.RT_BUF_RES_TEMPORAL, api.RT_BUF_ACCUMULATION, api.RT_BUF_PIXELS):
            a, b = rs[0].download(buf), rs[1].download(buf)
            if a.dtype.names:
                sh = rs[0].download(api.RT_BUF_RES_TEMPORAL)["M"] > 0
                assert not _res_bad(a, b, sh), (f, buf)
            else:
                assert _eq_bits(a, b), (f, buf)
    for r in rs:
        r.close()


def test_fused_final_pass_on_three_local_strips_vs_oracle(api, oracle, scenes):
    """rt_tuning 23 = 1 on strip contexts: the last pass (boundary rows on the main stream, interior rows on the second lane,
    halo records from the exchange lists) resolves its own rows"""
    from cedec_2024_rt_amd.types import bench_options

    W, H, frames = 480, 270, 8
    tris = scenes.make_blocks_restir()
    eye, at = scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT
    bounds = api.mg_partition(H, 3)
    ctxs = []
    for b in bounds:
        c = api.Renderer(W, H, rows=b, halo=87, exp=True)
        c.tuning(23, 1)
        c.set_scene(tris)
        c.lookat(eye, at)
        c.set_options(bench_options())
        ctxs.append(c)
    hub = api.MgHub(3, renderer=ctxs[0])
    mgs = [api.MultiGpu(c, k, bounds, transport=api.RT_MG_TRANSPORT_LOCAL, hub=hub) for k, c in enumerate(ctxs)]
    for f in range(1, frames + 1):
        api.mg_frame_lockstep(mgs, f)
    st, _ = _oracle_frames(oracle, tris, W, H, eye, at, frames)
    ref = st["accum"].reshape(H, W, 4)
    for c, (a, b) in zip(ctxs, bounds):
        acc = c.download(api.RT_BUF_ACCUMULATION).reshape(c.local_rows, W, 4)[a - c.local_row0: b - c.local_row0]
        assert _eq_bits(acc, ref[a:b]), f"rows {a}:{b}: {int((acc != ref[a:b]).any(axis=2).sum())} pixels differ from the oracle"
        px = c.download(api.RT_BUF_PIXELS).reshape(c.local_rows, W, 4)[a - c.local_row0: b - c.local_row0]
        assert np.array_equal(px, st["pixels"][a:b])
    for m in mgs:
        m.close()
    hub.close()
    for c in ctxs:
        c.close()


def test_one_launch_stage0_on_three_local_strips_vs_oracle(api, oracle, scenes):
    """rt_tuning 25 = 1 forced on strip contexts (auto: whole frames only): the look-ahead's stage 0 of every strip is one launch; the
    marks of the halo plans then read a G-buffer the candidates' kernel wrote"""
    from cedec_2024_rt_amd.types import bench_options

    W, H, frames = 480, 270, 9
    tris = scenes.make_blocks_restir()
    eye, at = scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT
    bounds = api.mg_partition(H, 3)
    ctxs = []
    for b in bounds:
        c = api.Renderer(W, H, rows=b, halo=87)
        c.tuning(25, 1)
        c.set_scene(tris)
        c.lookat(eye, at)
        c.set_options(bench_options())
        ctxs.append(c)
    hub = api.MgHub(3, renderer=ctxs[0])
    mgs = [api.MultiGpu(c, k, bounds, transport=api.RT_MG_TRANSPORT_LOCAL, hub=hub) for k, c in enumerate(ctxs)]
    for f in range(1, frames + 1):
        api.mg_frame_lockstep(mgs, f)
    st, _ = _oracle_frames(oracle, tris, W, H, eye, at, frames)
    ref = st["accum"].reshape(H, W, 4)
    for c, (a, b) in zip(ctxs, bounds):
        acc = c.download(api.RT_BUF_ACCUMULATION).reshape(c.local_rows, W, 4)[a - c.local_row0: b - c.local_row0]
        assert _eq_bits(acc, ref[a:b]), f"rows {a}:{b}: {int((acc != ref[a:b]).any(axis=2).sum())} pixels differ from the oracle"
        vis = c.download(api.RT_BUF_VISIBILITY).reshape(c.local_rows, W)[a - c.local_row0: b - c.local_row0]
        assert np.array_equal(vis["index"], st["vis"]["index"].reshape(H, W)[a:b])
    for m in mgs:
        m.close()
    hub.close()
    for c in ctxs:
        c.close()


def test_permuted_dispatch_order_vs_oracle(api, oracle, scenes):
    """rt_exp_tile_perm (experiments library, tools/tile_lpt.py): the workgroups of stage 0 and of resolve in a random XCD-keeping order
    — every tile still runs once, the frames are the oracle's; a permutation that moves a workgroup to another XCD is refused"""
    import ctypes as C

    from cedec_2024_rt_amd.types import bench_options

    W, H, frames = 480, 270, 6
    tris = scenes.make_blocks_restir()
    eye, at = scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT
    r = api.Renderer(W, H, exp=True)
    fn = r.L.rt_exp_tile_perm
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    fn.restype = C.c_int
    tx, ty = (W + 7) // 8, (H + 7) // 8
    grid = max((tx * ty + 127) // 128 * 128, 8 * ((ty + 7) // 8) * tx)
    rng = np.random.default_rng(5)
    for kernel in (0, 1, 3):
        perm = np.arange(grid, dtype=np.uint32)
        for x in range(8):
            idx = np.arange(x, grid, 8)
            perm[idx] = rng.permutation(idx)
        assert fn(r.h, kernel, perm.ctypes.data, grid) == 0
    bad = np.arange(grid, dtype=np.uint32)
    bad[[0, 1]] = bad[[1, 0]]
    assert fn(r.h, 2, bad.ctypes.data, grid) != 0 and fn(r.h, 1, bad.ctypes.data, grid) != 0  # the spatial pass has no such hook; XCD changed
    for kernel in (1,):  # the refused call removed kernel 1's order: set it again
        perm = np.arange(grid, dtype=np.uint32)
        for x in range(8):
            idx = np.arange(x, grid, 8)
            perm[idx] = rng.permutation(idx)
        assert fn(r.h, kernel, perm.ctypes.data, grid) == 0
    r.set_scene(tris)
    r.lookat(eye, at)
    r.set_options(bench_options())
    for f in range(1, frames + 1):
        r.frame(f)
    st, shaded = _oracle_frames(oracle, tris, W, H, eye, at, frames)
    acc = r.download(api.RT_BUF_ACCUMULATION)
    assert _eq_bits(acc, st["accum"].reshape(acc.shape))
    assert not _res_bad(r.download(api.RT_BUF_RES_TEMPORAL), st["temporal"], shaded)
    r.close()


def test_cached_mark_rows_give_the_same_plans(api, scenes):
    """rt_tuning 21: the shaded-bit rows of rt_halo_mark_sides built once per epoch == rebuilt in front of every mark, for several
    frames and across a camera move (the cache must not survive the epoch)"""
    import torch

    from cedec_2024_rt_amd.types import bench_options

    W, H = 480, 270
    tris = scenes.make_blocks_restir()
    bounds = api.mg_partition(H, 3)
    rank = 1
    out = []
    for cache in (1, 0):
        c = api.Renderer(W, H, rows=bounds[rank], halo=87)
        c.tuning(21, cache)
        c.set_scene(tris)
        c.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
        c.set_options(bench_options())
        words = c.halo_bitmap_words(87)
        fb = c.halo_flags_bytes(87)
        got = []
        for move in range(2):
            if move:
                c.orbit(40.0, -15.0)
            c.raycast()
            # the neighbours' flags: this rank's own boundary rows stand in for them (any deterministic content will do)
            flags = torch.zeros(fb, dtype=torch.uint8, device="cuda")
            for side, (src, dst) in enumerate(((bounds[rank][0], bounds[rank][0] - 87), (bounds[rank][1] - 87, bounds[rank][1]))):
                c.halo_flags_pack(src, 87, flags.data_ptr())
                c.sync()
                c.halo_flags_unpack(dst, 87, flags.data_ptr())
            for frame in (3, 4, 5):
                bm = torch.zeros((2, 3, words), dtype=torch.int32, device="cuda")
                c._ck(c.L.rt_halo_mark_sides(c.h, frame, 0, 3, bm[0].data_ptr(), bm[1].data_ptr()))
                c.sync()
                got.append(bm.cpu().numpy().copy())
        out.append(got)
        c.close()
    for a, b in zip(*out):
        assert np.array_equal(a, b)
    assert any(int(a[:, :, 0].sum()) > 0 for a in out[0])  # something was marked at all


def test_wire_model_transport_moves_what_mirror_moves_and_takes_its_time(api, scenes):
    """RT_MG_TRANSPORT_WIRE_MODEL = RCCL_SELF + a dependent delay per exchange: same images as MIRROR (a rank receives what it
    sent), the modelled wire time is reported, and with a slow modelled link the frames really take that long"""
    import os

    from cedec_2024_rt_amd.types import bench_options

    W, H = 480, 270
    tris = scenes.make_blocks_restir()
    bounds = api.mg_partition(H, 3)
    imgs, stats, wall = {}, {}, {}
    for name, T, env in (("mirror", api.RT_MG_TRANSPORT_MIRROR, {}), ("wire", api.RT_MG_TRANSPORT_WIRE_MODEL, {}),
                         ("slow", api.RT_MG_TRANSPORT_WIRE_MODEL, {"RT_MG_WIRE_GBS": "0.5", "RT_MG_WIRE_LAT_US": "200"})):
        old = {k: os.environ.get(k) for k in ("RT_MG_WIRE_GBS", "RT_MG_WIRE_LAT_US")}
        os.environ.update(env)
        try:
            c = api.Renderer(W, H, rows=bounds[1], halo=87)
            c.set_scene(tris)
            c.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
            c.set_options(bench_options())
            mg = api.MultiGpu(c, 1, bounds, transport=T)
            for f in range(1, 4):
                mg.frame(f)
            c.sync()
            mg.reset_stats()
            t0 = time.perf_counter()
            for f in range(4, 10):
                mg.frame(f)
            c.sync()
            wall[name] = (time.perf_counter() - t0) / 6
            stats[name] = mg.stats()
            imgs[name] = c.download(api.RT_BUF_ACCUMULATION)
            mg.close()
            c.close()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    assert _eq_bits(imgs["mirror"], imgs["wire"]) and _eq_bits(imgs["mirror"], imgs["slow"])
    assert stats["mirror"]["wire_ns"] == 0 and stats["wire"]["wire_ns"] > 0
    assert stats["wire"]["bytes_sent"] == stats["mirror"]["bytes_sent"]
    # three exchanges per frame, each held for >= 200 us + bytes / 0.5 GB/s: the stream cannot be faster than the model
    per_frame_model = stats["slow"]["wire_ns"] / 6 * 1e-9
    assert per_frame_model > 3 * 200e-6
    assert wall["slow"] >= 0.8 * per_frame_model, (wall, per_frame_model)  # a sanity bound, not a measurement: the host clock against the GPU's


def _log_uniform(rng, n, e_lo, e_hi, signed=False):
    """float32 values with exponents uniform in [e_lo, e_hi) and random mantissas, a quarter of them extreme"""
    e = rng.integers(e_lo + 127, e_hi + 127, size=n, dtype=np.uint32)
    m = rng.integers(0, 1 << 23, size=n, dtype=np.uint32)
    k = n // 4
    m[:k // 3] = 0
    m[k // 3: 2 * k // 3] = (1 << 23) - 1
    m[2 * k // 3: k] = 1 << rng.integers(0, 23, size=k - 2 * k // 3, dtype=np.uint32)
    bits = (e << 23) | m
    if signed:
        bits |= rng.integers(0, 2, size=n, dtype=np.uint32) << 31
    return bits.view(np.float32)


def test_division_free_accept_is_the_division(api):
    """reservoir_accept (rt_device.h, r05) == `u < weight / w_sum` (common/reservoir.hpp:22-37) for EVERY operand triple: decided
    from u * w_sum against weight where the two are more than 2^-20 |weight| apart, by the division itself otherwise. Device
    self-check against IEEE binary32 division recomputed on the host (numpy) and on the device, over: triples as the RIS loop has
    them (w_sum >= weight > 0, u in [0, 1)), triples built to land within a few ulps of equality (the only place the two forms
    could differ), u = 0, zeros, subnormals, huge and tiny magnitudes, infinities, NaNs, negative weights and sums."""
    r = api.Renderer(8, 8)
    rng = np.random.default_rng(505)
    fast_total = 0
    for batch in range(5):
        n = 3_000_000
        u = ((rng.integers(0, 1 << 23, size=n, dtype=np.uint32) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)).astype(np.float32)  # PCG::uniformf's grid
        W = _log_uniform(rng, n, -40, 40)
        S = (W * (np.float32(1.0) + _log_uniform(rng, n, -10, 12))).astype(np.float32)  # w_sum = weight + earlier weights
        if batch == 1:  # near equality: S = W / u up to a few ulps, so that u * S is within rounding distance of W
            with np.errstate(all="ignore"):
                S = (W / np.maximum(u, np.float32(2.0 ** -23))).astype(np.float32)
            S = (S.view(np.uint32) + rng.integers(-3, 4, size=n).astype(np.uint32)).view(np.float32)
        if batch == 2:  # arbitrary magnitudes and signs, u anywhere in [0, 1)
            W = _log_uniform(rng, n, -126, 120, signed=True)
            S = _log_uniform(rng, n, -126, 120, signed=True)
        if batch == 3:  # special values in every position
            sp = np.float32([0.0, -0.0, 1e-45, -1e-45, 1e-38, 1.0, np.inf, -np.inf, np.nan, 3.4e38, 2.0 ** -120, 2.0 ** -100])
            k = len(sp)
            g = np.array(np.meshgrid(np.float32([0.0, 2.0 ** -23, 0.25, 0.5, 1.0 - 2.0 ** -23]), sp, sp)).reshape(3, -1)
            u[: g.shape[1]], W[: g.shape[1]], S[: g.shape[1]] = g[0], g[1], g[2]
            u[g.shape[1]: g.shape[1] + 200000] = 0.0  # u = 0: "is the quotient non-zero", incl. quotients that underflow
            W[g.shape[1]: g.shape[1] + 100000] = _log_uniform(rng, 100000, -126, -60)
            S[g.shape[1]: g.shape[1] + 100000] = _log_uniform(rng, 100000, 20, 120)
        if batch == 4:  # equality exactly and one ulp around it
            S = _log_uniform(rng, n, -30, 30)
            W = (u * S).astype(np.float32)
            W = (W.view(np.uint32) + rng.integers(-2, 3, size=n).astype(np.uint32)).view(np.float32)
        x = np.stack([u, W, S], axis=1).astype(np.float32)
        got = r.math_eval(36, x)
        dev = r.math_eval(37, x)
        with np.errstate(all="ignore"):
            want = (x[:, 0] < (x[:, 1] / x[:, 2]).astype(np.float32))
        assert np.array_equal(dev != 0, want), "device IEEE division differs from the host's"  # the yardstick itself
        acc = (got.astype(np.int32) & 1) != 0
        bad = acc != want
        assert not bad.any(), f"batch {batch}: reservoir_accept differs from the division for {int(bad.sum())} triples, e.g. {x[bad][:4]}"
        fast = got >= 2.0
        fast_total += int(fast.sum())
        if batch == 0:
            assert fast.mean() > 0.99  # the RIS loop's operands: the division is the rare path
        if batch in (1, 4):
            assert (~fast).mean() > 0.2  # and the near-equality batches really reach it
    assert fast_total > 5_000_000
    r.close()


def test_single_guard_ris_weight_is_the_nested_form(api):
    """ris_weight (frame_kernels.h, r05: the three shared-reciprocal forms of a RIS candidate behind ONE range test) ==
    target_unshadowed(...) / pdf with the compiler's divisions, bit for bit: surface / light pairs as a frame has them and pairs
    built to leave every range (coincident and nearly coincident points, axis-aligned offsets, huge coordinates, zero / tiny /
    huge luminance and pdf)."""
    r = api.Renderer(8, 8)
    rng = np.random.default_rng(606)
    n = 3_000_000
    sp = (rng.random((n, 3), dtype=np.float32) * 60 - 30).astype(np.float32)
    lp = (rng.random((n, 3), dtype=np.float32) * 60 - 30).astype(np.float32)
    k = n // 10
    lp[:k, 0] = sp[:k, 0]
    lp[k:2 * k] = sp[k:2 * k] + (rng.random((k, 3), dtype=np.float32) * np.float32(1e-4)).astype(np.float32)
    lp[2 * k:3 * k] = sp[2 * k:3 * k]
    lp[3 * k:4 * k, 1] = (sp[3 * k:4 * k, 1] + np.float32(1e-30)).astype(np.float32)
    sp[4 * k:5 * k] *= np.float32(1e18)
    nrm = rng.normal(size=(2, n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
    lum = _log_uniform(rng, n, -20, 10)
    pdf = _log_uniform(rng, n, -20, 20)
    lum[5 * k:6 * k] = _log_uniform(rng, k, -126, 100)
    pdf[6 * k:7 * k] = _log_uniform(rng, k, -100, 100)
    lum[7 * k:7 * k + 1000] = 0.0
    x = np.concatenate([sp, nrm[0], lp, nrm[1], lum[:, None], pdf[:, None]], axis=1).astype(np.float32)
    out = r.math_eval(38, x).view(np.uint32)
    both_nan = (sp == lp).all(axis=1) | ~np.isfinite(sp).all(axis=1)
    bad = (out != 0) & ~both_nan
    # a NaN on both sides may differ in payload only
    assert not bad.any(), f"ris_weight: {int(bad.sum())} of {n} differ from the nested-guard form, e.g. {x[bad][:2]}"
    r.close()
