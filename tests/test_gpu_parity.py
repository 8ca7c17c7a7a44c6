"""GPU parity tests proper: the HIP path (through the C-ABI) against the oracle, bit for bit.

Integer / index / decision data must be identical; float data is compared as raw bits (the
portable math + -ffp-contract=off contract, DESIGN.md "Parity") — no tolerance anywhere except
the north-star's 1e-4 relative L2 that is asserted on top for the radiance.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FOVY = np.float32(np.pi) / np.float32(4)


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint8)


def _eq_bits(a, b):
    return np.array_equal(_bits(a), _bits(b))


def _res_fields_equal(a, b, mask=None):
    """Compare reservoir arrays field by field (padding excluded); returns list of bad fields."""
    bad = []
    for f in a.dtype.names:
        if f == "pad":
            continue
        x, y = a[f], b[f]
        if mask is not None:
            x, y = x[mask], y[mask]
        if not _eq_bits(np.ascontiguousarray(x), np.ascontiguousarray(y)):
            bad.append((f, int((np.ascontiguousarray(x).reshape(len(x), -1) != np.ascontiguousarray(y).reshape(len(y), -1)).any(axis=1).sum())))
    return bad


def _rel_l2(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


@pytest.fixture(scope="module")
def api():
    from cedec_2024_rt_amd import api as _api

    return _api


@pytest.fixture(scope="module")
def scenes():
    from cedec_2024_rt_amd import scenes as s

    return s


@pytest.fixture(scope="module")
def golden_scenes(golden_dir):
    import os

    return np.load(os.path.join(golden_dir, "scenes.npz"))


def test_portable_math_and_ieee_on_device(api, oracle):
    """Device portable_math.h == host portable_math.h; device fp32 divide / sqrt are IEEE
    (correctly rounded, denormals kept)."""
    oracle.set_math_mode(oracle.MATH_PORTABLE)
    r = api.Renderer(8, 8)
    rng = np.random.default_rng(5)
    n = 200000
    u = rng.random(n, dtype=np.float32)
    cases = {
        "logf": np.concatenate([u, (u * 2.0 ** -20).astype(np.float32), np.float32([0.0, 1.0, 2.0 ** -23, 1e-38, 1e-45])]),
        "cosf": np.concatenate([(u * 6.2831855).astype(np.float32), (u * 40 - 20).astype(np.float32)]),
        "sinf": np.concatenate([(u * 6.2831855).astype(np.float32), (u * 40 - 20).astype(np.float32)]),
        "sincos_sin": np.concatenate([(u * 6.2831855).astype(np.float32), (u * 40 - 20).astype(np.float32)]),
        "sincos_cos": np.concatenate([(u * 6.2831855).astype(np.float32), (u * 40 - 20).astype(np.float32)]),
        "expf": np.concatenate([(-u * 100).astype(np.float32), (u * 4 - 2).astype(np.float32), np.float32([-103.0, -104.5, 0.0, 88.0, 89.0])]),
        "pow8": np.concatenate([u, np.float32([0.0, 1.0, 1.0000001])]),
        "pow_gamma": np.concatenate([u, (u * 8).astype(np.float32), np.float32([0.0, 1.0])]),
        "sqrt": np.concatenate([u, (u * 1e-38).astype(np.float32), (u * 1e30).astype(np.float32)]),
    }
    for name, x in cases.items():
        fid = oracle.FN[name][0]
        dev = r.math_eval(fid, x)
        host = oracle.fn_bulk(name, x).reshape(-1)
        assert _eq_bits(dev, host), f"{name}: {(dev.view(np.uint32) != host.view(np.uint32)).sum()} of {len(x)} differ"
    a = np.concatenate([u, (u * 1e-30).astype(np.float32), (u * 1e-39).astype(np.float32)])
    b = np.concatenate([rng.random(n, dtype=np.float32) + np.float32(1e-3), (rng.random(n, dtype=np.float32) * 1e10).astype(np.float32) + np.float32(1), rng.random(n, dtype=np.float32) + np.float32(0.5)])
    x = np.stack([a, b], axis=1).astype(np.float32)
    dev = r.math_eval(26, x)
    host = (a / b).astype(np.float32)
    assert _eq_bits(dev, host), "fp32 division differs from IEEE"
    # float -> int the way every kernel and the oracle do it (rt_device.h f2i_sat = v_cvt_i32_f32): NaN -> 0, saturating, toward zero
    x = np.concatenate([(u * 4000 - 2000).astype(np.float32), (u * 1e10 - 5e9).astype(np.float32),
                        np.float32([np.nan, np.inf, -np.inf, 2147483520.0, 2147483648.0, -2147483648.0, -2147483904.0, 3e9, -3e9, -0.0, 0.99999994, -0.99999994,
                                    1e-45, -1e-45])])
    dev = r.math_eval(30, x).view(np.int32)
    x64 = x.astype(np.float64)
    host = np.where(np.isnan(x64), 0, np.clip(np.trunc(np.nan_to_num(x64, nan=0.0, posinf=3e9, neginf=-3e9)), -2147483648, 2147483647)).astype(np.int64)
    assert (dev.astype(np.int64) == host).all(), "float -> int conversion differs"
    r.close()


def _random_rays(rng, n, lo, hi):
    o = (rng.random((n, 3), dtype=np.float32) * (hi - lo) + lo).astype(np.float32)
    d = (rng.random((n, 3), dtype=np.float32) * 2 - 1).astype(np.float32)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = o
    rays[:, 3:6] = d
    rays[:, 6] = 0.0
    rays[:, 7] = 3.402823466e38
    # some axis-parallel and some short (shadow-style) rays
    rays[: n // 16, 3] = 0.0
    rays[n // 16: n // 8, 4] = 0.0
    rays[n // 8: n // 4, 7] = 0.99
    return rays


def test_lbvh_equals_brute_force(api, oracle, scenes, golden_scenes):
    """LBVH closest hit == brute force over all triangles (t, u, v, index incl. the tie rule)."""
    rng = np.random.default_rng(11)
    for name, tris in (("cornellbox1", golden_scenes["cornellbox1"]), ("cornellbox2", golden_scenes["cornellbox2"]),
                       ("quad_room", scenes.make_quad_room())):
        v = tris["v"].reshape(-1, 3)
        lo, hi = v.min(0), v.max(0)
        rays = _random_rays(rng, 60000 if len(tris) < 1000 else 20000, lo - 0.5, hi + 0.5)
        sc = oracle.Scene(tris, use_bvh=False)
        ref = sc.trace_closest(rays, force_brute=True)
        for builder in (0, 1, 2, 3):  # 0: device LBVH (Morton + Karras), 1: host binned SAH, 2: device PLOC, 3: device binned SAH
            r = api.Renderer(8, 8, exp=True)  # builders 0-2 and walk modes 1-3: librestir_rt_exp.so
            r.tuning(5, builder)
            r.set_scene(tris)
            # 0: 4-wide quantised BVH + LDS stack (production), 1: binary tree + stackless trail,
            # 2: persistent ray queue with lane refill (closest hit), 7 (r06): four lanes per ray, one child box each (closest_quad)
            for mode in (0, 1, 2, 7):
                r.trace_mode(mode)
                dev = r.trace_closest(rays)
                assert _eq_bits(dev, ref), f"{name} builder {builder} mode {mode}: {(dev.view(np.uint32) != ref.view(np.uint32)).any(axis=1).sum()} rays differ"
            # shadow-ray semantics (3: persistent queue, 4: the frame kernels' any-hit walk): occluded <=> a closest hit exists
            for mode in (3, 4):
                r.trace_mode(mode)
                occ = r.trace_closest(rays)[:, 3].view(np.int32) >= 0
                assert (occ == (ref[:, 3].view(np.int32) >= 0)).all(), f"{name} builder {builder} any-hit mode {mode}"
            # 5: the work-sharing any-hit walk (idle lanes take over half of a busy lane's stack), also with a third
            # of the lanes holding no ray at all (tmax < 0: they only help)
            occ, passes, steals = r.trace_occluded_ws(rays)
            assert (occ == (ref[:, 3].view(np.int32) >= 0)).all(), f"{name} builder {builder} work-sharing walk"
            holes = rays.copy()
            holes[::3, 7] = -1.0
            occ_h, _, _ = r.trace_occluded_ws(holes)
            want = ref[:, 3].view(np.int32) >= 0
            want[::3] = False
            assert (occ_h == want).all(), f"{name} builder {builder} work-sharing walk with idle lanes"
            if name == "quad_room":
                assert steals.sum() > 0, "no lane ever took over work: the walk under test is not exercised"
            r.close()
        assert (ref[:, 3].view(np.int32) >= 0).mean() > 0.2


def test_deep_traversal_stack_spills_past_lds(api, oracle):
    """A deck of 1.1M parallel cards: a ray along the deck hits every box of the tree, so the
    front-to-back walk holds ~3 entries per level and runs past the 24 LDS stack entries into the
    scratch half (the wave-uniform `deep` path of trace_wide). Results == brute force."""
    from cedec_2024_rt_amd.types import TRIANGLE as TRIANGLE_DTYPE

    n = 1100000
    tris = np.zeros(n, TRIANGLE_DTYPE)
    z = (np.arange(n, dtype=np.float32) * np.float32(0.004)).astype(np.float32)
    tris["v"][:, 0] = np.stack([np.zeros(n), np.zeros(n), z], 1)
    tris["v"][:, 1] = np.stack([np.full(n, 2.0), np.zeros(n), z], 1)
    tris["v"][:, 2] = np.stack([np.zeros(n), np.full(n, 2.0), z], 1)
    tris["color"] = 0.5
    rng = np.random.default_rng(5)
    m = 192
    rays = np.zeros((m, 8), np.float32)
    rays[:, 0:2] = rng.random((m, 2), dtype=np.float32) * 0.9 + 0.02
    rays[:, 2] = np.where(np.arange(m) % 2 == 0, -1.0, z[-1] + 1.0)
    rays[:, 3:6] = rng.normal(size=(m, 3)).astype(np.float32) * 0.002
    rays[:, 5] = np.where(np.arange(m) % 2 == 0, 1.0, -1.0)
    rays[:, 6] = 0.0
    rays[:, 7] = np.where(np.arange(m) % 3 == 0, 3.0e38, rng.random(m, dtype=np.float32) * 4000.0)
    rays[::7, 6] = 500.0  # tmin inside the deck: the nearest cards are skipped
    sc = oracle.Scene(tris, use_bvh=False)
    ref = sc.trace_closest(rays, force_brute=True)
    assert (ref[:, 3].view(np.int32) >= 0).mean() > 0.5
    for builder in (0, 1, 2, 3):
        r = api.Renderer(8, 8, exp=True)  # builders 0-2 and walk modes 1-3: librestir_rt_exp.so
        r.tuning(5, builder)
        r.set_scene(tris)
        assert 3 * (r.bvh_info()["wide_height"] - 1) >= 24 + 6, r.bvh_info()
        for mode in (0, 2, 7):  # 7: closest_quad keeps its whole (inner-records-only) stack in LDS
            r.trace_mode(mode)
            dev = r.trace_closest(rays)
            assert _eq_bits(dev, ref), f"builder {builder} mode {mode}"
        for mode in (3, 4):
            r.trace_mode(mode)
            occ = r.trace_closest(rays)[:, 3].view(np.int32) >= 0
            assert (occ == (ref[:, 3].view(np.int32) >= 0)).all(), f"builder {builder} any-hit mode {mode}"
        occ, _, _ = r.trace_occluded_ws(rays)  # a stolen stack window never reaches into the spilled (per-lane) part
        assert (occ == (ref[:, 3].view(np.int32) >= 0)).all(), f"builder {builder} work-sharing walk"
        r.close()


def test_lbvh_blocks_scene_vs_oracle_bvh(api, oracle, scenes):
    """On the benchmark stand-in (212k triangles) the oracle's own BVH (itself == brute force on
    the small scenes, tests/test_oracle_bvh.py) and the device LBVH agree on 200k rays."""
    tris = scenes.make_blocks_restir()
    rng = np.random.default_rng(3)
    rays = _random_rays(rng, 200000, np.float32([-20, 0, -10]), np.float32([40, 40, 60]))
    sc = oracle.Scene(tris, use_bvh=True)
    ref = sc.trace_closest(rays)
    for builder in (0, 1, 2, 3):
        r = api.Renderer(8, 8, exp=True)  # builders 0-2 and walk modes 1-3: librestir_rt_exp.so
        r.tuning(5, builder)
        r.set_scene(tris)
        info = r.scene_info()
        print(f"builder {builder}: rt_scene_set {r.build_ms():.1f} ms, {r.bvh_info()}, binary height {info['bvh_height']}")
        assert info["triangles"] == len(tris) and info["lights"] == len(scenes.light_indices(tris))
        for mode in (0, 1, 2, 7):
            r.trace_mode(mode)
            dev = r.trace_closest(rays)
            assert _eq_bits(dev, ref), f"builder {builder} mode {mode}: {(dev.view(np.uint32) != ref.view(np.uint32)).any(axis=1).sum()} rays differ"
        occ, passes, steals = r.trace_occluded_ws(rays)
        assert (occ == (ref[:, 3].view(np.int32) >= 0)).all(), f"builder {builder} work-sharing walk"
        assert steals.sum() > 1000, steals.sum()
        r.trace_mode(0)
        # a 2000-ray subset against true brute force
        sub = rays[:2000]
        assert _eq_bits(r.trace_closest(sub), sc.trace_closest(sub, force_brute=True))
        r.close()


def _setup(api, oracle, tris, W, H, eye, center, **optkw):
    oracle.set_math_mode(oracle.MATH_PORTABLE)
    r = api.Renderer(W, H)
    r.set_scene(tris)
    r.lookat(eye, center)
    from cedec_2024_rt_amd.types import bench_options

    r.set_options(bench_options(**optkw))
    sc = oracle.Scene(tris, use_bvh=True)
    rg = oracle.raygen_lookat(eye, center, (0, 1, 0), FOVY, W, H)
    assert rg.tobytes() == r.raygen().tobytes()
    return r, sc, rg, oracle.bench_options(**optkw), np.asarray(eye, np.float32)


@pytest.mark.parametrize("scene_name", ["cornellbox1", "quad_room", "cornellbox2", "quad_room_shadowed"])
def test_kernel_by_kernel_parity(api, oracle, scenes, golden_scenes, scene_name):
    """Every reference kernel, one at a time through its own entry point, two frames
    (`_shadowed`: with use_shadowed_target_function, README key 3)."""
    optkw = dict(use_shadowed_target_function=1) if scene_name.endswith("_shadowed") else {}
    scene_name = scene_name.replace("_shadowed", "")
    if scene_name == "quad_room":
        tris, eye, center = scenes.make_quad_room(), (0.5, 2.5, 6.0), (0.0, 1.5, -1.0)
    elif scene_name == "cornellbox1":
        tris, eye, center = golden_scenes["cornellbox1"], scenes.DEFAULT_EYE, scenes.DEFAULT_LOOKAT
    else:
        tris, eye, center = golden_scenes["cornellbox2"], scenes.CORNELLBOX_EYE, scenes.CORNELLBOX_LOOKAT
    W, H = 96, 54
    r, sc, rg, opt, eyev = _setup(api, oracle, tris, W, H, eye, center, **optkw)
    st = oracle.new_state(W, H)
    lights = set(sc.lights.tolist())
    for frame in (1, 2):
        # raycast
        r.raycast()
        sc.raycast(W, H, rg, st["vis"])
        vis = r.download(api.RT_BUF_VISIBILITY)
        for f in ("uv", "index"):
            assert _eq_bits(vis[f], st["vis"][f]), f"raycast {f} frame {frame}"
        shaded = (st["vis"]["index"] >= 0) & ~np.isin(st["vis"]["index"], list(lights))
        assert shaded.mean() > 0.1
        # generate_candidate
        r.generate_candidate(frame, api.RT_RES_0)
        sc.generate_candidate(W, H, frame, st["vis"], eyev, opt, st["r0"])
        got = r.download(api.RT_BUF_RES_0)
        assert not _res_fields_equal(got, st["r0"]), f"generate_candidate frame {frame}: {_res_fields_equal(got, st['r0'])}"
        # temporal_resampling
        r.temporal_resampling(frame, api.RT_RES_TEMPORAL, api.RT_RES_0)
        sc.temporal_resampling(W, H, frame, st["vis"], eyev, opt, st["temporal"], st["r0"])
        got = r.download(api.RT_BUF_RES_0)
        assert not _res_fields_equal(got, st["r0"]), f"temporal frame {frame}: {_res_fields_equal(got, st['r0'])}"
        # save_temporal_reservoir
        r.save_temporal_reservoir(api.RT_RES_0, api.RT_RES_TEMPORAL)
        oracle.save_temporal_reservoir(W, H, st["r0"], st["temporal"])
        assert not _res_fields_equal(r.download(api.RT_BUF_RES_TEMPORAL), st["temporal"])
        # spatial x3 (the reference writes only shaded pixels of `out`)
        src, dst = api.RT_RES_0, api.RT_RES_1
        osrc, odst = st["r0"], st["r1"]
        for k in range(3):
            if k:
                src, dst = dst, src
                osrc, odst = odst, osrc
            r.spatial_resampling(frame, k, src, dst)
            sc.spatial_resampling(W, H, frame, k, st["vis"], eyev, opt, osrc, odst)
            got = r.download(api.RT_BUF_RES_0 + dst)
            bad = _res_fields_equal(got, odst, mask=shaded)
            assert not bad, f"spatial pass {k} frame {frame}: {bad}"
        # resolve + tone mapping
        r.resolve(dst)
        sc.resolve(st["accum"], W, H, st["vis"], eyev, opt, odst)
        acc = r.download(api.RT_BUF_ACCUMULATION)
        assert _eq_bits(acc, st["accum"].reshape(acc.shape)), f"resolve frame {frame}"
        assert _rel_l2(acc[:, :3], st["accum"].reshape(acc.shape)[:, :3]) <= 1e-4
        r.tone_mapping()
        px = r.download(api.RT_BUF_PIXELS).reshape(H, W, 4)
        opx = oracle.tone_mapping(st["accum"], W, H)
        assert np.array_equal(px, opx), "tone_mapping"
    r.close()


@pytest.mark.parametrize("optkw", [
    dict(),
    dict(use_temporal_resampling=0),
    dict(use_spatial_resampling=0),
    dict(use_visibility_reuse=0),
    dict(accumulate=1),
    dict(spatial_resampling_passes=2),
    dict(spatial_resampling_passes=1, spatial_resampling_sample_count=3, ris_sample_count=8),
    dict(use_shadowed_target_function=1),
    # shadowed target: the batched shadow walks (<= 6 distinct rays per pixel and pass) ...
    dict(use_shadowed_target_function=1, use_visibility_reuse=0),
    dict(use_shadowed_target_function=1, use_temporal_resampling=0, spatial_resampling_sample_count=3),
    dict(use_shadowed_target_function=1, use_spatial_resampling=0),
    # ... and the one-ray-at-a-time form kept for more than 5 neighbours
    dict(use_shadowed_target_function=1, spatial_resampling_sample_count=7, spatial_resampling_passes=2),
    dict(spatial_resampling_sample_count=7, spatial_resampling_passes=1),
])
def test_fused_frame_equals_oracle_frame(api, oracle, scenes, optkw):
    """rt_frame (fused generate+temporal, rotating buffers) == the reference's 8-launch frame,
    over 3 frames and for every option toggle of the example (keys 1,2,3,4,A)."""
    tris = scenes.make_quad_room()
    W, H = 80, 45
    r, sc, rg, opt, eyev = _setup(api, oracle, tris, W, H, (0.5, 2.5, 6.0), (0.0, 1.5, -1.0), **optkw)
    st = oracle.new_state(W, H)
    for frame in (1, 2, 3):
        final = r.frame(frame)
        cnt = oracle.new_counters()
        sc.frame(W, H, frame, rg, eyev, opt, st, cnt)
        acc = r.download(api.RT_BUF_ACCUMULATION)
        assert _eq_bits(acc, st["accum"].reshape(acc.shape)), f"{optkw} frame {frame}: {(acc != st['accum'].reshape(acc.shape)).any(axis=1).sum()} pixels"
        # rays = raytrace() calls of the reference for this frame, whatever the build really walks
        assert r.ray_count()[0] == int(cnt["rays"][0]), f"{optkw} frame {frame}: ray count"
        # temporal history handed to the next frame
        shaded = (st["vis"]["index"] >= 0) & ~np.isin(st["vis"]["index"], sc.lights)
        bad = _res_fields_equal(r.download(api.RT_BUF_RES_TEMPORAL), st["temporal"], mask=shaded)
        assert not bad, f"{optkw} temporal history frame {frame}: {bad}"
        passes = int(opt["spatial_resampling_passes"][0])
        assert final == (api.RT_RES_1 if passes % 2 == 1 else api.RT_RES_0)
        px = r.download(api.RT_BUF_PIXELS).reshape(H, W, 4)
        assert np.array_equal(px, st["pixels"])
    rays, shaded_n = r.ray_count()
    assert shaded_n == int(shaded.sum())
    r.close()


@pytest.mark.parametrize("W,H", [(480, 270), (1920, 1080), (3840, 2160)])  # quarter res, BASELINE configs #4 and #5
def test_frame_parity_blocks_restir_quarter_res(api, oracle, scenes, W, H):
    """The benchmark stand-in at 480x270 and at the BASELINE size 1920x1080 (the bench.py workload:
    benchmark options, 2 frames): bit-identical radiance, reservoirs and pixels against the oracle,
    identical ray count (rays counted by the CPU restatement)."""
    tris = scenes.make_blocks_restir()
    r, sc, rg, opt, eyev = _setup(api, oracle, tris, W, H, scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
    st = oracle.new_state(W, H)
    for frame in (1, 2):
        cnt = oracle.new_counters()
        r.frame(frame)
        sc.frame(W, H, frame, rg, eyev, opt, st, cnt)
        acc = r.download(api.RT_BUF_ACCUMULATION)
        ref = st["accum"].reshape(acc.shape)
        nbad = int((acc.view(np.uint32) != ref.view(np.uint32)).any(axis=1).sum())
        assert nbad == 0, f"frame {frame}: {nbad} pixels differ, rel-L2 {_rel_l2(acc[:, :3], ref[:, :3])}"
        rays, shaded_n = r.ray_count()
        assert rays == int(cnt["rays"][0]) and shaded_n == int(cnt["shaded_pixels"][0])
    shaded = (st["vis"]["index"] >= 0) & ~np.isin(st["vis"]["index"], sc.lights)
    bad = _res_fields_equal(r.download(api.RT_BUF_RES_TEMPORAL), st["temporal"], mask=shaded)
    assert not bad, f"temporal history: {bad}"
    assert np.array_equal(r.download(api.RT_BUF_PIXELS).reshape(H, W, 4), st["pixels"])
    r.close()


@pytest.mark.parametrize("W,H,frames,optkw", [
    (480, 270, 3, dict(use_shadowed_target_function=1)),
    (480, 270, 3, dict(use_shadowed_target_function=1, use_visibility_reuse=0)),
    (1920, 1080, 1, dict(use_shadowed_target_function=1)),   # the size the shadowed-mode ms / Mray/s are quoted on
])
def test_shadowed_target_on_the_bench_scene(api, oracle, scenes, W, H, frames, optkw):
    """README key 3 (common/reservoir.hpp:52-57, 10_restir_di.cu:195-199,346-350) on the blocks_restir stand-in —
    the deep tree (29 levels, 380 k records) the batched work-sharing walks are quoted on: radiance, temporal history,
    pixels and the reference's ray count against the oracle."""
    tris = scenes.make_blocks_restir()
    r, sc, rg, opt, eyev = _setup(api, oracle, tris, W, H, scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT, **optkw)
    st = oracle.new_state(W, H)
    for frame in range(1, frames + 1):
        cnt = oracle.new_counters()
        r.frame(frame)
        sc.frame(W, H, frame, rg, eyev, opt, st, cnt)
        acc = r.download(api.RT_BUF_ACCUMULATION)
        ref = st["accum"].reshape(acc.shape)
        nbad = int((acc.view(np.uint32) != ref.view(np.uint32)).any(axis=1).sum())
        assert nbad == 0, f"{optkw} frame {frame}: {nbad} pixels differ, rel-L2 {_rel_l2(acc[:, :3], ref[:, :3])}"
        rays, shaded_n = r.ray_count()
        assert rays == int(cnt["rays"][0]) and shaded_n == int(cnt["shaded_pixels"][0])
    shaded = (st["vis"]["index"] >= 0) & ~np.isin(st["vis"]["index"], sc.lights)
    bad = _res_fields_equal(r.download(api.RT_BUF_RES_TEMPORAL), st["temporal"], mask=shaded)
    assert not bad, f"temporal history: {bad}"
    assert np.array_equal(r.download(api.RT_BUF_PIXELS).reshape(H, W, 4), st["pixels"])
    r.close()


def test_shadowed_target_three_strips_on_the_bench_scene(api, oracle, scenes):
    """Three 90-row strips of a 480x270 shadowed-target frame of the bench scene == the oracle's frame (and so the
    single context's, by the test above), two frames."""
    import torch

    from cedec_2024_rt_amd import strips
    from cedec_2024_rt_amd.types import bench_options

    W, H = 480, 270
    optkw = dict(use_shadowed_target_function=1)
    tris = scenes.make_blocks_restir()
    oracle.set_math_mode(oracle.MATH_PORTABLE)
    sc = oracle.Scene(tris, use_bvh=True)
    rg = oracle.raygen_lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT, (0, 1, 0), FOVY, W, H)
    eyev = np.asarray(scenes.BLOCKS_RESTIR_EYE, np.float32)
    st = oracle.new_state(W, H)
    bounds = strips.partition_rows(H, 3)
    ctxs = []
    for b in bounds:
        c = api.Renderer(W, H, rows=b, halo=strips.HALO_ROWS)
        c.set_scene(tris)
        c.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
        c.set_options(bench_options(**optkw))
        ctxs.append(c)
    for frame in (1, 2):
        cnt = oracle.new_counters()
        sc.frame(W, H, frame, rg, eyev, oracle.bench_options(**optkw), st, cnt)
        ref = st["accum"].reshape(H, W, 4)
        strips.run_frame_local(ctxs, bounds, frame, torch.device("cuda:0"), sparse=bool(frame % 2))
        for c, (a, b) in zip(ctxs, bounds):
            acc = c.download(api.RT_BUF_ACCUMULATION).reshape(c.local_rows, W, 4)
            assert _eq_bits(acc[a - c.local_row0: b - c.local_row0], ref[a:b]), f"rows {a}:{b} frame {frame}"
        assert sum(c.ray_count()[0] for c in ctxs) == int(cnt["rays"][0])
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("n_strips,H,sparse", [(2, 240, False), (3, 300, False), (2, 240, True), (3, 300, True)])
def test_strip_contexts_match_full_frame(api, oracle, scenes, n_strips, H, sparse):
    """Two row-strip contexts on one GPU with an 87-row halo exchanged through the C-ABI halo
    calls reproduce the single-context frame bit for bit."""
    import ctypes as C

    tris = scenes.make_quad_room()
    W = 64
    from cedec_2024_rt_amd.types import bench_options
    from cedec_2024_rt_amd import strips

    full = api.Renderer(W, H)
    full.set_scene(tris)
    full.lookat((0.5, 2.5, 6.0), (0.0, 1.5, -1.0))
    full.set_options(bench_options())
    ctxs = []
    bounds = strips.partition_rows(H, n_strips)
    for (a, b) in bounds:
        c = api.Renderer(W, H, rows=(a, b), halo=strips.HALO_ROWS)
        c.set_scene(tris)
        c.lookat((0.5, 2.5, 6.0), (0.0, 1.5, -1.0))
        c.set_options(bench_options())
        ctxs.append(c)
    lib = api.load_library()
    import torch

    for frame in (1, 2):
        full.frame(frame)
        ref = full.download(api.RT_BUF_ACCUMULATION).reshape(H, W, 4)
        strips.run_frame_local(ctxs, bounds, frame, torch.device("cuda:0"), sparse=sparse)
        for c, (a, b) in zip(ctxs, bounds):
            acc = c.download(api.RT_BUF_ACCUMULATION).reshape(c.local_rows, W, 4)
            mine = acc[a - c.local_row0: b - c.local_row0]
            assert _eq_bits(mine, ref[a:b]), f"strip rows {a}:{b} frame {frame}"
    for c in ctxs:
        c.close()
    full.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("RT_STRIP_SEEDS", "8")))))
def test_random_strip_partitions_match_single_context(api, scenes, seed):
    """Random image sizes, 2-5 strips, random option sets, dense or sparse halos: the strip contexts driven
    through StripFrame (two lanes where a strip has interior rows) reproduce the single context, 2 frames."""
    import torch

    from cedec_2024_rt_amd import strips
    from cedec_2024_rt_amd.types import bench_options

    rng = np.random.default_rng(500 + seed)
    n_strips = int(rng.integers(2, 6))
    H = int(rng.integers(n_strips * strips.HALO_ROWS, n_strips * 260))
    W = int(rng.integers(33, 150))
    sparse = bool(rng.integers(0, 2))
    optkw = dict(use_temporal_resampling=int(rng.integers(0, 2)), use_visibility_reuse=int(rng.integers(0, 2)),
                 use_shadowed_target_function=int(rng.integers(0, 4) == 0), ris_sample_count=int(rng.integers(1, 9)),
                 spatial_resampling_passes=int(rng.integers(1, 4)), spatial_resampling_sample_count=int(rng.integers(1, 7)),
                 accumulate=int(rng.integers(0, 2)))
    tris = scenes.make_quad_room()
    eye, at = (0.5 + float(rng.normal()) * 0.5, 2.5, 6.0), (0.0, 1.5 + float(rng.normal()) * 0.3, -1.0)

    def make(rows=None, halo=0):
        c = api.Renderer(W, H, rows=rows, halo=halo)
        c.set_scene(tris)
        c.lookat(eye, at)
        c.set_options(bench_options(**optkw))
        return c

    full = make()
    bounds = strips.partition_rows(H, n_strips)
    ctxs = [make(rows=b, halo=strips.HALO_ROWS) for b in bounds]
    for frame in (1, 2):
        full.frame(frame)
        ref = full.download(api.RT_BUF_ACCUMULATION).reshape(H, W, 4)
        refpx = full.download(api.RT_BUF_PIXELS).reshape(H, W, 4)
        strips.run_frame_local(ctxs, bounds, frame, torch.device("cuda:0"), sparse=sparse)
        for c, (a, b) in zip(ctxs, bounds):
            acc = c.download(api.RT_BUF_ACCUMULATION).reshape(c.local_rows, W, 4)
            assert _eq_bits(acc[a - c.local_row0: b - c.local_row0], ref[a:b]), f"seed {seed}: {n_strips} strips of {W}x{H}, sparse={sparse}, {optkw}, rows {a}:{b}, frame {frame}"
            px = c.download(api.RT_BUF_PIXELS).reshape(c.local_rows, W, 4)
            assert np.array_equal(px[a - c.local_row0: b - c.local_row0], refpx[a:b])
    assert sum(c.ray_count()[0] for c in ctxs) == full.ray_count()[0]
    for c in ctxs + [full]:
        c.close()


@pytest.mark.parametrize("example,scene_name,W,H,frames,optkw", [
    (7, "cornellbox2", 512, 512, 4, dict(accumulate=1)),                       # BASELINE config #2
    (7, "quad_room", 96, 54, 2, dict(accumulate=1, sky_color=(0.3, 0.4, 0.5))),
    (9, "quad_room", 96, 54, 2, dict(accumulate=1)),
    (9, "quad_room", 64, 36, 1, dict(use_shadowed_target_function=1, ris_sample_count=8)),
    (9, "blocks", 320, 180, 1, dict()),                                        # config #3 at quarter res
    (9, "blocks", 1280, 720, 1, dict()),                                       # BASELINE config #3 at full size (scene as 09_ris.cpp:149 loads it)
    (9, "blocks_pt", 1280, 720, 1, dict()),                                    # BASELINE config #3 as its text names it: blocks_pt + the 07_pt camera
    (7, "blocks_pt", 480, 270, 2, dict(accumulate=1)),
    (8, "blocks_pt", 480, 270, 1, dict()),
    (9, "blocks_pt", 320, 180, 1, dict(use_shadowed_target_function=1, ris_sample_count=8, max_depth=3)),
    (8, "quad_room", 96, 54, 2, dict(accumulate=1)),                           # 08_nee
    (8, "cornellbox2", 256, 256, 2, dict(accumulate=1, max_depth=3)),
    (8, "blocks", 320, 180, 1, dict()),
    # SURVEY 8f rank 2 names 1920x1080: the sizes profiles/*_config_table.json quotes a time for
    (7, "blocks_pt", 1920, 1080, 1, dict()),
    (8, "blocks_pt", 1920, 1080, 1, dict()),
    (9, "blocks_pt", 1920, 1080, 1, dict()),
])
def test_path_tracers_07_and_09(api, oracle, scenes, golden_scenes, example, scene_name, W, H, frames, optkw):
    """Configs #2/#3 and 08_nee: the `path_trace` kernels of 07_pt, 08_nee and 09_ris, bit-identical
    radiance and the same number of raytrace() calls as the oracle."""
    from cedec_2024_rt_amd.types import default_options

    if scene_name == "cornellbox2":
        tris, eye, center = golden_scenes["cornellbox2"], scenes.CORNELLBOX_EYE, scenes.CORNELLBOX_LOOKAT
    elif scene_name == "blocks":
        tris, eye, center = scenes.make_blocks_restir(), scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT
    elif scene_name == "blocks_pt":
        tris, eye, center = scenes.make_blocks_pt(), scenes.BLOCKS_PT_EYE, scenes.BLOCKS_PT_LOOKAT
    else:
        tris, eye, center = scenes.make_quad_room(), (0.5, 2.5, 6.0), (0.0, 1.5, -1.0)
    oracle.set_math_mode(oracle.MATH_PORTABLE)
    r = api.Renderer(W, H)
    r.set_scene(tris)
    r.lookat(eye, center)
    r.set_options(default_options(**optkw))
    sc = oracle.Scene(tris, use_bvh=True)
    rg = oracle.raygen_lookat(eye, center, (0, 1, 0), FOVY, W, H)
    opt = oracle.default_options(**optkw)
    acc = np.zeros((W * H, 4), np.float32)
    r.clear()
    for frame in range(1, frames + 1):
        cnt = oracle.new_counters()
        r.tuning(6, frame % 2)  # alternate: one launch per frame / wavefront (one launch per bounce, compaction)
        r.path_trace(example, frame)
        sc.path_trace(example, W, H, frame, rg, opt, acc, cnt=cnt)
        got = r.download(api.RT_BUF_ACCUMULATION)
        nbad = int((got.view(np.uint32) != acc.view(np.uint32)).any(axis=1).sum())
        assert nbad == 0, f"frame {frame} (wavefront={frame % 2}): {nbad} pixels differ, rel-L2 {_rel_l2(got[:, :3], acc[:, :3])}"
        assert r.path_trace_rays() == int(cnt["rays"][0])
    assert acc[:, :3].max() > 0
    r.tone_mapping()
    assert np.array_equal(r.download(api.RT_BUF_PIXELS).reshape(H, W, 4), oracle.tone_mapping(acc, W, H))
    r.close()


def test_edge_cases_and_error_codes(api, oracle, scenes):
    """Odd image sizes (not multiples of the 32x8 tile), all-sky frames, scenes without lights,
    a single triangle, zero spatial passes, and the error behaviour of the C-ABI (codes, no aborts)."""
    from cedec_2024_rt_amd.types import bench_options, default_options

    oracle.set_math_mode(oracle.MATH_PORTABLE)
    tris = scenes.make_quad_room()
    eye, center = (0.5, 2.5, 6.0), (0.0, 1.5, -1.0)
    # 1. odd sizes, 1 and 0 spatial passes
    for (W, H, kw) in ((37, 23, dict()), (33, 9, dict(spatial_resampling_passes=1)), (65, 17, dict(spatial_resampling_passes=0))):
        r, sc, rg, opt, eyev = _setup(api, oracle, tris, W, H, eye, center, **kw)
        st = oracle.new_state(W, H)
        for frame in (1, 2):
            r.frame(frame)
            sc.frame(W, H, frame, rg, eyev, opt, st)
            acc = r.download(api.RT_BUF_ACCUMULATION)
            if kw.get("spatial_resampling_passes", 3) > 0:  # with 0 passes the reference resolves a stale buffer
                assert _eq_bits(acc, st["accum"].reshape(acc.shape)), (W, H, kw, frame)
        r.close()
    # 2. camera looking away: every pixel is sky
    r, sc, rg, opt, eyev = _setup(api, oracle, tris, 40, 24, (0.0, 50.0, 0.0), (0.0, 100.0, 0.0))
    r.frame(1)
    acc = r.download(api.RT_BUF_ACCUMULATION)
    assert np.all(acc[:, :3] == 0) and np.all(acc[:, 3] == 1)
    assert r.ray_count() == (40 * 24, 0)
    r.close()
    # 3. a scene without emissive triangles: raycast works, generate_candidate reports an error code
    dark = tris.copy()
    dark["emissive"] = 0
    r = api.Renderer(32, 16)
    r.set_scene(dark)
    r.lookat(eye, center)
    r.set_options(bench_options())
    r.raycast()
    assert (r.download(api.RT_BUF_VISIBILITY)["index"] >= 0).any()
    with pytest.raises(api.RtError, match="no emissive"):
        r.generate_candidate(1)
    with pytest.raises(api.RtError):
        r.frame(1)
    # 4. single triangle / empty scene
    one = tris[:1].copy()
    r.set_scene(one)
    h = r.trace_closest(np.float32([[-3.0, 5.0, -3.0, 0, -1, 0, 0, 1e30], [50, 5, 50, 0, -1, 0, 0, 1e30]]))
    ref = oracle.Scene(one, use_bvh=False).trace_closest(np.float32([[-3.0, 5.0, -3.0, 0, -1, 0, 0, 1e30], [50, 5, 50, 0, -1, 0, 0, 1e30]]), force_brute=True)
    assert _eq_bits(h, ref)
    r.set_scene(tris[:0])
    r.raycast()
    assert (r.download(api.RT_BUF_VISIBILITY)["index"] == -1).all()
    # 5. argument / state errors come back as codes with a message
    with pytest.raises(api.RtError):
        r.spatial_resampling(1, 0, api.RT_RES_0, api.RT_RES_0)  # in == out
    with pytest.raises(api.RtError):
        r.upload(api.RT_BUF_RES_0, np.zeros(3, dtype=np.uint8))  # size mismatch
    with pytest.raises(api.RtError):
        r.halo_pack(api.RT_RES_0, 0, 10_000, 0)  # rows outside the context
    r.close()
    with pytest.raises(api.RtError):
        api.Renderer(0, 10)
    r2 = api.Renderer(16, 16)
    with pytest.raises(api.RtError, match="scene and camera"):
        r2.raycast()
    r2.close()
    # 6. a strip whose halo cannot cover the spatial radius is refused, not rendered wrongly
    r3 = api.Renderer(32, 200, rows=(0, 100), halo=40)
    r3.set_scene(tris)
    r3.lookat(eye, center)
    r3.set_options(bench_options())
    r3.raycast()
    r3.generate_candidate(1)
    with pytest.raises(api.RtError, match="too small"):
        r3.spatial_resampling(1, 0, api.RT_RES_0, api.RT_RES_1)
    r3.set_options(bench_options(spatial_resampling_radius=10.0))  # 29 rows are enough now
    r3.spatial_resampling(1, 0, api.RT_RES_0, api.RT_RES_1)
    import torch
    t = torch.empty(r3.halo_bitmap_words(40) * 4, dtype=torch.uint8, device="cuda")
    with pytest.raises(api.RtError, match="shaded flags"):
        r3.halo_mark(1, 0, 3, 1, t.data_ptr())  # neighbour flags not exchanged since the raycast
    # the staged frame's two lanes: misuse is reported, a split stage equals the one-call stage
    r3.set_options(bench_options(spatial_resampling_radius=10.0, spatial_resampling_passes=1))
    with pytest.raises(api.RtError, match="expected stage"):
        r3.frame_stage_run(2, 1, 0, 100)
    r3.frame_stage_begin(2, 0, False)
    r3.frame_stage_run_async(2, 0, 0, 40, 100)  # rows 40..99 on the second stream ...
    with pytest.raises(api.RtError, match="second lane"):
        r3.frame_stage_fork()
    r3.frame_stage_run(2, 0, 0, 40)             # ... rows 0..39 on the main stream
    r3.frame_stage_end(0)
    r3.frame_stage_begin(2, 1, False)
    r3.frame_stage_run(2, 1, 0, 100)
    r3.frame_stage_end(1)
    r3.frame_stage_begin(2, 2, False)
    r3.frame_stage_run(2, 2, 0, 100)
    r3.frame_stage_end(2)
    split = r3.download(api.RT_BUF_ACCUMULATION).copy()
    r4 = api.Renderer(32, 200, rows=(0, 100), halo=40)
    r4.set_scene(tris)
    r4.lookat(eye, center)
    r4.set_options(bench_options(spatial_resampling_radius=10.0, spatial_resampling_passes=1))
    for st in range(3):
        r4.frame_stage(2, st, False)
    assert _eq_bits(split, r4.download(api.RT_BUF_ACCUMULATION))
    r3.close()
    r4.close()


def test_round3_entry_points(api, scenes):
    """rt_tuning_get reads back what rt_tuning set (and the defaults), rt_build_id names the build, rt_side_stream hands out
    the tail stream, rt_halo_fuse_set refuses to be called outside a spatial stage or for a side without a neighbour."""
    import ctypes as C

    L = api.load_library(exp=True)
    r = api.Renderer(64, 48, exp=True)
    if not os.environ.get("RT_TUNING"):  # the defaults (soak runs force other settings through the environment)
        assert r.tuning_get(5) == 3 and "SAH" in r.bvh_builder()      # default builder: the device SAH build
        assert r.tuning_get(14) == -1 and r.tuning_get(17) == -1 and r.tuning_get(13) == 1
        assert r.tuning_get(0) == -1 and r.tuning_get(2) == -1 and r.tuning_get(16) == -1 and r.tuning_get(19) == 1  # r04 autos
    for key, val in ((5, 1), (14, 2), (17, 1), (9, 5), (13, 0)):
        r.tuning(key, val)
        assert r.tuning_get(key) == val
    with pytest.raises(api.RtError):
        r.tuning(5, 4)
    with pytest.raises(api.RtError):
        r.tuning_get(99)
    # the PRODUCT library carries none of the A/B forms and says so
    if not os.environ.get("RT_LIB_PATH") and not os.environ.get("RT_EXPERIMENTS"):
        prod = api.Renderer(64, 48)
        for key, val in ((5, 0), (5, 1), (5, 2), (8, 0), (8, 1), (8, 3), (8, 4), (9, 4), (9, 5), (10, 8), (11, 1), (12, 1), (15, 1), (16, 2), (23, 1), (24, 1)):
            with pytest.raises(api.RtError, match="librestir_rt_exp"):
                prod.tuning(key, val)
        for key, val in ((5, 3), (8, 2), (9, -1), (9, 6), (11, 0), (13, 0), (14, 1), (16, 1), (20, 0), (21, 0), (22, 1), (23, 0), (24, 0), (25, 0), (25, 1)):
            prod.tuning(key, val)
        for mode in (1, 2, 3, 7):
            with pytest.raises(api.RtError, match="librestir_rt_exp"):
                prod.trace_mode(mode)
        prod.close()
    assert len(api.build_id(exp=False)) == 16 and r.build_id().startswith(api.build_id(exp=False))
    s = C.c_void_p()
    assert L.rt_side_stream(r.h, 0, C.byref(s)) == 0 and s.value
    assert L.rt_side_stream(r.h, 1, C.byref(s)) != 0
    r.set_scene(scenes.make_quad_room())
    r.lookat((0.5, 2.5, 6.0), (0.0, 1.5, -1.0))
    fuse = (C.c_void_p * 8)()
    assert L.rt_halo_fuse_set(r.h, None) == 0                         # NULL clears
    assert L.rt_halo_fuse_set(r.h, fuse) == 3                         # RT_ERR_STATE: no spatial stage is running
    r.close()
    # a strip without a neighbour below: side 0 must stay empty
    top = api.Renderer(64, 200, rows=(0, 100), halo=87)
    top.set_scene(scenes.make_quad_room())
    top.lookat((0.5, 2.5, 6.0), (0.0, 1.5, -1.0))
    from cedec_2024_rt_amd.types import bench_options

    top.set_options(bench_options())
    for st in (0,):
        assert L.rt_frame_stage(top.h, 1, st, 0) == 0
    assert L.rt_frame_stage_begin(top.h, 1, 1, 0) == 0
    dummy = (C.c_uint32 * 4)()
    fuse[0] = C.addressof(dummy)   # need_bitmap[0]: the strip below — there is none
    fuse[2] = C.addressof(dummy)   # recv_list[0]
    assert L.rt_halo_fuse_set(top.h, fuse) == 1                       # RT_ERR_ARG
    top.close()


def test_interactive_camera_and_accumulation_reset(api, oracle, scenes):
    """CameraControl (common/misc.hpp:108-224) as C-ABI calls: orbit keeps the distance, zoom scales it,
    pan moves eye and look-at together; a moved camera raises `updated`, which the frame loop turns into
    a clear of the accumulation buffer (10_restir_di.cpp:257-267)."""
    from cedec_2024_rt_amd.types import bench_options

    tris = scenes.make_quad_room()
    W, H = 64, 36
    r = api.Renderer(W, H)
    r.set_scene(tris)
    eye0, at0 = np.float32([0.5, 2.5, 6.0]), np.float32([0.0, 1.5, -1.0])
    r.lookat(eye0, at0)
    r.set_options(bench_options(accumulate=1))
    assert not r.camera_updated()
    r.clear()
    for f in (1, 2, 3):
        r.frame(f, clear_first=r.camera_updated())
    vis = r.download(api.RT_BUF_VISIBILITY)
    shaded = (vis["index"] >= 0) & ~np.isin(vis["index"], scenes.light_indices(tris))
    w = r.download(api.RT_BUF_ACCUMULATION)[:, 3]
    # shaded pixels accumulate; sky / emissive pixels are assigned {.., 1} every frame (10_restir_di.cu:408-425)
    assert shaded.any() and np.all(w[shaded] == 3.0) and np.all(w[~shaded] == 1.0)
    # orbit: distance to the look-at point is preserved, look-at unchanged
    r.orbit(40.0, -25.0)
    e, a = r.camera_pose()
    assert np.allclose(a, at0) and abs(np.linalg.norm(e - a) - np.linalg.norm(eye0 - at0)) < 1e-4 and not np.allclose(e, eye0)
    # the pose is the oracle's restatement of misc.hpp:147-181 (itself pinned to the reference's CameraControl,
    # tests/test_oracle_golden.py::test_camera_control_vs_reference_fixture), bit for bit
    oe, oa, upd = oracle.camera_control(eye0, at0, 0, 40.0, -25.0)
    assert upd and e.tobytes() == oe.tobytes() and a.tobytes() == oa.tobytes()
    assert r.camera_updated() and not r.camera_updated()
    # the frame after a camera move starts a new accumulation
    r.frame(4, clear_first=True)
    assert np.all(r.download(api.RT_BUF_ACCUMULATION)[:, 3] == 1.0)
    # and equals a fresh renderer's first accumulated frame at that pose (same frame index)
    r2 = api.Renderer(W, H)
    r2.set_scene(tris)
    r2.lookat(e, a)
    r2.set_options(bench_options(accumulate=1))
    assert r2.raygen().tobytes() == r.raygen().tobytes()
    # zoom / pan
    d0 = np.linalg.norm(e - a)
    r.zoom(100.0)
    e2, a2 = r.camera_pose()
    assert abs(np.linalg.norm(e2 - a2) - d0 * (1 - 0.002 * 100.0)) < 1e-4 and np.allclose(a2, a)
    r.pan(10.0, 5.0)
    e3, a3 = r.camera_pose()
    assert np.allclose(e3 - e2, a3 - a2, atol=1e-6) and np.linalg.norm(e3 - e2) > 0
    r.close()
    r2.close()


@pytest.mark.parametrize("W,H,optkw", [(480, 270, dict()), (333, 190, dict(spatial_resampling_sample_count=3, use_visibility_reuse=0)),
                                       (1920, 1080, dict())])
def test_lds_staged_spatial_variant_is_bit_identical(api, scenes, W, H, optkw):
    """rt_tuning key 8 = 1 (the unshadowed spatial pass with the tile's shaded-bit window staged in LDS) and 8 = 2 (the
    wavefront fetches its 64 neighbour records together, four lanes per record, through LDS) make the same
    decisions as the gather kernel: accumulation, pixels and all three reservoir buffers identical over 3 frames."""
    from cedec_2024_rt_amd.types import bench_options

    tris = scenes.make_blocks_restir() if W == 1920 else scenes.make_quad_room()
    eye, at = (scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT) if W == 1920 else ((0.5, 2.5, 6.0), (0.0, 1.5, -1.0))
    rs = []
    for variant in (0, 1, 2, 3):  # 3 (r04): the cooperative fetch software-pipelined over the staged shaded-bit window
        r = api.Renderer(W, H, exp=True)
        r.set_scene(tris)
        r.lookat(eye, at)
        r.set_options(bench_options(**optkw))
        r.tuning(8, variant)
        rs.append(r)
    for frame in (1, 2, 3):
        for r in rs:
            r.frame(frame)
        for buf in (api.RT_BUF_ACCUMULATION, api.RT_BUF_PIXELS, api.RT_BUF_RES_0, api.RT_BUF_RES_1, api.RT_BUF_RES_TEMPORAL):
            for r in rs[1:]:
                assert _eq_bits(rs[0].download(buf), r.download(buf)), (frame, buf)
    for r in rs:
        r.close()


def test_candidate_kernel_variants_are_bit_identical(api, scenes):
    """rt_tuning key 11 (visibility-reuse rays only for candidates that survive the temporal merge, through a compacted
    queue) and key 12 (software-pipelined RIS loop): same accumulation, pixels and reservoirs as the default kernel over
    4 frames incl. the first (no history: every ray is walked) — and the queue really drops rays once history exists."""
    from cedec_2024_rt_amd.types import bench_options

    tris = scenes.make_quad_room()
    W, H = 320, 180
    rs = []
    for k11, k12 in ((0, 0), (1, 0), (0, 1), (1, 1)):
        r = api.Renderer(W, H, exp=True)
        r.set_scene(tris)
        r.lookat((0.5, 2.5, 6.0), (0.0, 1.5, -1.0))
        r.set_options(bench_options())
        r.tuning(11, k11)
        r.tuning(12, k12)
        rs.append(r)
    shaded = None
    for frame in (1, 2, 3, 4):
        for r in rs:
            r.frame(frame)
        for buf in (api.RT_BUF_ACCUMULATION, api.RT_BUF_PIXELS, api.RT_BUF_RES_0, api.RT_BUF_RES_1, api.RT_BUF_RES_TEMPORAL):
            ref = rs[0].download(buf)
            for r in rs[1:]:
                assert _eq_bits(ref, r.download(buf)), (frame, buf)
        walked = rs[1].visibility_rays_walked()
        shaded = rs[0].ray_count()[1]
        assert walked == shaded if frame == 1 else 0 < walked < shaded, (frame, walked, shaded)
    for r in rs:
        r.close()


def test_work_sharing_shadow_walk_is_bit_identical(api, scenes):
    """rt_tuning key 13: generate_candidate / resolve with the work-sharing shadow-ray walk (default) and with the
    one-lane-one-ray walk give the same accumulation, pixels and reservoirs over 4 frames of the bench scene at quarter
    resolution, whole frame and as a strip context."""
    from cedec_2024_rt_amd.types import bench_options

    tris = scenes.make_blocks_restir()
    W, H = 480, 270
    for rows, halo in ((None, 0), ((90, 180), 90)):
        rs = []
        for ws in (0, 1):
            r = api.Renderer(W, H, rows=rows, halo=halo)
            r.set_scene(tris)
            r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
            r.set_options(bench_options())
            r.tuning(13, ws)
            rs.append(r)
        for frame in (1, 2, 3, 4):
            for r in rs:
                if rows is None:
                    r.frame(frame)
                else:
                    for k in range(5):
                        r.frame_stage(frame, k, False)
            for buf in (api.RT_BUF_ACCUMULATION, api.RT_BUF_PIXELS, api.RT_BUF_RES_0, api.RT_BUF_RES_1, api.RT_BUF_RES_TEMPORAL):
                assert _eq_bits(rs[0].download(buf), rs[1].download(buf)), (rows, frame, buf)
        for r in rs:
            r.close()


def test_work_sharing_closest_hit_walk_is_bit_identical(api, scenes):
    """rt_tuning key 16 (primary rays walked with closest_ws; evaluated, off by default): the Visibility buffer (u, v,
    index per pixel) of the bench scene and of a scene with coplanar overlaps equals the default raycast, bit for bit."""
    for tris, eye, at in ((scenes.make_blocks_restir(), scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT),
                          (scenes.make_quad_room(), (0.5, 2.5, 6.0), (0.0, 1.5, -1.0))):
        out = []
        for ws in (0, 1):
            r = api.Renderer(640, 360)
            r.set_scene(tris)
            r.lookat(eye, at)
            r.tuning(16, ws)
            r.raycast()
            out.append(r.download(api.RT_BUF_VISIBILITY))
            r.close()
        assert _eq_bits(out[0], out[1])
        assert (out[0]["index"] >= 0).mean() > 0.3


def test_resolve_as_a_stream_is_bit_identical(api, scenes):
    """rt_tuning key 15 (persistent wavefronts that refill finished lanes with the next pixels; evaluated, off by default):
    same accumulation and pixels as the default resolve, with and without accumulation, whole frame and strip."""
    from cedec_2024_rt_amd.types import bench_options

    tris = scenes.make_blocks_restir()
    W, H = 480, 270
    for rows, halo, acc in ((None, 0, 0), (None, 0, 1), ((90, 200), 70, 0)):
        rs = []
        for stream in (0, 1):
            r = api.Renderer(W, H, rows=rows, halo=halo, exp=True)
            r.set_scene(tris)
            r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
            r.set_options(bench_options(accumulate=acc, spatial_resampling_radius=20.0))
            r.tuning(15, stream)
            rs.append(r)
        for frame in (1, 2, 3):
            for r in rs:
                if rows is None:
                    r.frame(frame)
                else:
                    for k in range(5):
                        r.frame_stage(frame, k, False)
            for buf in (api.RT_BUF_ACCUMULATION, api.RT_BUF_PIXELS):
                assert _eq_bits(rs[0].download(buf), rs[1].download(buf)), (rows, acc, frame, buf)
        for r in rs:
            r.close()


def test_next_frame_raycast_overlap_is_bit_identical(api, scenes):
    """rt_tuning key 14: with the next frame's primary rays traced beside the current frame (and thrown away when the
    camera moves, the options change or the per-kernel API is used in between) every buffer equals the run that traces
    them at the start of each frame."""
    from cedec_2024_rt_amd.types import bench_options

    tris = scenes.make_quad_room()
    W, H = 320, 180
    rs = []
    # key 14 = 2 (r03): the next frame's generate_candidate + temporal_resampling too (pipelined stage 0);
    # key 17 = 1 (r03): resolve + tone_mapping on their own stream, the next frame does not wait for them
    for spec, tail in ((0, 0), (1, 0), (2, 0), (2, 1), (0, 1)):
        r = api.Renderer(W, H)
        r.set_scene(tris)
        r.lookat((0.5, 2.5, 6.0), (0.0, 1.5, -1.0))
        r.set_options(bench_options())
        r.tuning(14, spec)
        r.tuning(17, tail)
        rs.append(r)
    frame = 0
    steps = ("still", "still", "orbit", "still", "options", "still", "kernels", "still", "still", "upload", "still", "jump", "still",
             "temporal_off", "still", "shadowed", "still", "still", "accumulate", "still", "still", "orbit", "still")
    for step in steps:
        frame += 3 if step == "jump" else 1  # a frame number the pipelined stage 0 was not made for
        saved = rs[0].download(api.RT_BUF_RES_TEMPORAL) if step == "upload" else None
        for r in rs:
            if step == "orbit":
                r.orbit(37.0, -11.0)
            if step == "options":
                r.set_options(bench_options(spatial_resampling_passes=2))
            if step == "temporal_off":
                r.set_options(bench_options(use_temporal_resampling=0))
            if step == "shadowed":
                r.set_options(bench_options(use_shadowed_target_function=1))
            if step == "accumulate":
                r.set_options(bench_options(accumulate=1, spatial_resampling_passes=2))
            if step == "kernels":  # the per-kernel API between two frames: its raycast writes the current G-buffer,
                r.raycast()        # its generate_candidate a reservoir buffer the next frame's history could be
                r.generate_candidate(frame, api.RT_RES_TEMPORAL)
            if step == "upload":   # fixture injection into the temporal history between two frames
                h = saved.copy()
                h["M"] = np.minimum(h["M"], 3)
                r.upload(api.RT_BUF_RES_TEMPORAL, h)
            r.frame(frame, clear_first=(step in ("orbit", "options")))
        for buf in (api.RT_BUF_VISIBILITY, api.RT_BUF_ACCUMULATION, api.RT_BUF_PIXELS, api.RT_BUF_RES_0, api.RT_BUF_RES_1, api.RT_BUF_RES_TEMPORAL):
            for other in rs[1:]:
                assert _eq_bits(rs[0].download(buf), other.download(buf)), (frame, step, buf)
    for r in rs:
        r.close()


def test_camera_api_equals_the_references_camera_control(api, scenes, golden_dir):
    """rt_camera_orbit / _zoom / _pan + the RayGenerator they re-derive == the REFERENCE'S CameraControl
    (common/misc.hpp:108-224) + RayGenerator::lookat over the committed drag sequences (3 x 120 events, produced by
    the reference's own code: tests/golden/ref_camera.npz), bit for bit."""
    g = np.load(os.path.join(golden_dir, "ref_camera.npz"))
    fovy = g["fovy"][0]
    tris = scenes.make_quad_room()
    for i in range(3):
        start, events, want = g[f"pose{i}_start"], g[f"pose{i}_events"], g[f"pose{i}_out"]
        r = api.Renderer(int(start[6]), int(start[7]))
        r.set_scene(tris)
        r.lookat(start[:3], start[3:6], fovy=fovy)
        for k, (b, dx, dy) in enumerate(events):
            (r.orbit, r.zoom, r.pan)[int(b)](*((float(dx), float(dy)) if int(b) != 1 else (float(dy),)))
            e, a = r.camera_pose()
            got = [int(v) for v in e.view(np.uint32)] + [int(v) for v in a.view(np.uint32)] + [int(r.camera_updated())]
            got += [int(v) for v in r.raygen().view(np.uint32).reshape(-1)]
            assert got == [int(v) for v in want[k]], f"pose {i}, event {k} (button {int(b)}, dx {dx}, dy {dy})"
        r.close()


def test_frame_sequence_across_a_camera_move(api, oracle, scenes):
    """The interactive loop of 10_restir_di.cpp:231-383 with a camera drag in the middle: frames 1-2, an orbit
    (is_updated() -> `clear`, :257-267), frames 3-4 — which merge temporal history gathered at the OLD pose at
    the same pixel (no reprojection, 10_restir_di.cu:178). Accumulation, pixels and the temporal reservoirs of
    every frame == the oracle driven through the same sequence, bit for bit."""
    from cedec_2024_rt_amd.types import bench_options

    oracle.set_math_mode(oracle.MATH_PORTABLE)
    tris = scenes.make_quad_room()
    W, H = 96, 54
    eye, at = np.float32([0.5, 2.5, 6.0]), np.float32([0.0, 1.5, -1.0])
    fovy = np.float32(np.pi) / np.float32(4)
    for optkw in (dict(accumulate=1), dict(accumulate=1, use_shadowed_target_function=1, spatial_resampling_passes=2)):
        r = api.Renderer(W, H)
        r.set_scene(tris)
        r.lookat(eye, at)
        r.set_options(bench_options(**optkw))
        r.clear()
        sc = oracle.Scene(tris, use_bvh=True)
        oopt = oracle.bench_options(**optkw)
        st = oracle.new_state(W, H)
        oe, oa = eye.copy(), at.copy()
        for frame in (1, 2, 3, 4, 5):
            if frame == 3:
                r.orbit(55.0, -20.0)
                oe, oa, _ = oracle.camera_control(oe, oa, 0, 55.0, -20.0)
            if frame == 5:
                r.pan(-30.0, 12.0)
                r.zoom(40.0)
                oe, oa, _ = oracle.camera_control(oe, oa, 2, -30.0, 12.0)
                oe, oa, _ = oracle.camera_control(oe, oa, 1, 0.0, 40.0)
            moved = r.camera_updated()
            assert moved == (frame in (3, 5))
            rg = oracle.raygen_lookat(oe, oa, (0, 1, 0), fovy, W, H)
            assert rg.tobytes() == r.raygen().tobytes()
            if moved:
                oracle.clear(st["accum"], W, H)
            r.frame(frame, clear_first=moved)
            sc.frame(W, H, frame, rg, oe, oopt, st)
            acc = r.download(api.RT_BUF_ACCUMULATION)
            assert _eq_bits(acc, st["accum"].reshape(acc.shape)), f"{optkw} frame {frame}: accumulation"
            assert np.array_equal(r.download(api.RT_BUF_PIXELS).reshape(H, W, 4), st["pixels"])
            hist = r.download(api.RT_BUF_RES_TEMPORAL)
            vis = r.download(api.RT_BUF_VISIBILITY)
            shaded = (vis["index"] >= 0) & ~np.isin(vis["index"], scenes.light_indices(tris))
            assert not _res_fields_equal(hist, st["temporal"].reshape(hist.shape), mask=shaded), f"{optkw} frame {frame}: temporal history"
        r.close()


@pytest.mark.parametrize("W,H", [(1920, 1080), (3840, 2160)])  # BASELINE configs #4 and #5
def test_full_size_properties_1080p(api, scenes, W, H):
    """At BASELINE.json's full sizes (blocks_restir stand-in, 1920x1080 and 3840x2160, bench options) the oracle is too
    slow for a per-pixel check in a test, so size-independent properties are asserted instead:
    (1) the fused rt_frame == the reference's kernel-by-kernel launch sequence, bit for bit;
    (2) 8 row strips (135 / 270 rows, 87-row halos, the 8-GPU partition) == the single-context frame;
    (3) ray count = pixels + 2 x shaded pixels; the spatial pass's algorithmic bytes are bounded by
        16 + 152 + 5 x 92 per pixel and are the same for every pass input that shares the G-buffer."""
    import torch

    from cedec_2024_rt_amd import strips
    from cedec_2024_rt_amd.types import bench_options

    tris = scenes.make_blocks_restir()
    eye, at = scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT

    def make(rows=None, halo=0):
        r = api.Renderer(W, H, rows=rows, halo=halo)
        r.set_scene(tris)
        r.lookat(eye, at)
        r.set_options(bench_options())
        return r

    fused, bykernel = make(), make()
    bounds = strips.partition_rows(H, 8)
    ctxs = [make(rows=b, halo=strips.HALO_ROWS) for b in bounds]
    for frame in (1, 2):
        fused.frame(frame)
        bykernel.frame_by_kernels(frame)
        a = fused.download(api.RT_BUF_ACCUMULATION)
        b = bykernel.download(api.RT_BUF_ACCUMULATION)
        assert _eq_bits(a, b), f"fused vs kernel-by-kernel, frame {frame}: {(a != b).any(axis=1).sum()} pixels"
        assert np.array_equal(fused.download(api.RT_BUF_PIXELS), bykernel.download(api.RT_BUF_PIXELS))
        stats = []
        strips.run_frame_local(ctxs, bounds, frame, torch.device("cuda:0"), sparse=(frame == 2), stats=stats)
        if frame == 2:  # sparse halos: same image from a fraction of the records
            sent, dense = sum(s[0] for s in stats), sum(s[1] for s in stats)
            assert 0.05 < sent / dense < 0.5, (sent, dense)
            print(f"sparse halos: {sent} of {dense} records = {sent / dense:.3f}")
        full = a.reshape(H, W, 4)
        for c, (r0, r1) in zip(ctxs, bounds):
            mine = c.download(api.RT_BUF_ACCUMULATION).reshape(c.local_rows, W, 4)[r0 - c.local_row0: r1 - c.local_row0]
            assert _eq_bits(mine, full[r0:r1]), f"strip rows {r0}:{r1}, frame {frame}"
    rays, shaded = fused.ray_count()
    assert rays == W * H + 2 * shaded and 0.5 < shaded / (W * H) <= 1.0
    assert sum(c.ray_count()[0] for c in ctxs) == rays
    nbytes = [fused.spatial_bytes(3, k, api.RT_RES_0)[0] for k in range(3)]
    assert all(16 * W * H + 152 * shaded < n <= 16 * W * H + (152 + 5 * 92) * shaded for n in nbytes)
    assert fused.spatial_bytes(3, 0, api.RT_RES_0) == fused.spatial_bytes(3, 0, api.RT_RES_1)
    for c in ctxs + [fused, bykernel]:
        c.close()


def test_upload_download_round_trip_is_lossless(api, oracle, scenes):
    """The internal 64-B record + 16-B radiance layout holds every field of the reference's 76-B
    Reservoir: arbitrary reservoirs (random bits in the float fields, incl. NaN/inf/denormals,
    0 <= M < 2^30) survive rt_upload -> rt_download unchanged; visibility / accumulation too."""
    W, H = 40, 24
    tris = scenes.make_quad_room()
    r = api.Renderer(W, H)
    r.set_scene(tris)
    r.lookat((0.5, 2.5, 6.0), (0.0, 1.5, -1.0))
    sc = oracle.Scene(tris, use_bvh=True)
    rg = oracle.raygen_lookat((0.5, 2.5, 6.0), (0.0, 1.5, -1.0), (0, 1, 0), FOVY, W, H)
    vis = sc.raycast(W, H, rg)
    r.upload(api.RT_BUF_VISIBILITY, vis)
    got = r.download(api.RT_BUF_VISIBILITY)
    assert _eq_bits(got["uv"], vis["uv"]) and np.array_equal(got["index"], vis["index"])
    # the G-buffer rebuilt from the uploaded visibility is the one raycast would have produced
    r.generate_candidate(3, api.RT_RES_0)
    want = sc.generate_candidate(W, H, 3, vis, np.float32([0.5, 2.5, 6.0]), oracle.default_options())
    assert not _res_fields_equal(r.download(api.RT_BUF_RES_0), want)
    rng = np.random.default_rng(8)
    res = np.zeros(W * H, dtype=oracle.RESERVOIR)
    for f in ("origin_position", "origin_normal", "hit_position", "hit_normal", "radiance"):
        res[f] = rng.integers(0, 2 ** 32, size=(W * H, 3), dtype=np.uint64).astype(np.uint32).view(np.float32)
    for f in ("w_sum", "ucw"):
        res[f] = rng.integers(0, 2 ** 32, size=W * H, dtype=np.uint64).astype(np.uint32).view(np.float32)
    res["visibility"] = rng.integers(0, 2, size=W * H)
    res["M"] = rng.integers(0, 2 ** 30, size=W * H)
    for buf in (api.RT_BUF_RES_0, api.RT_BUF_RES_1, api.RT_BUF_RES_TEMPORAL):
        r.upload(buf, res)
        back = r.download(buf)
        assert not _res_fields_equal(back, res), _res_fields_equal(back, res)
    acc = rng.random((W * H, 4), dtype=np.float32)
    r.upload(api.RT_BUF_ACCUMULATION, acc)
    assert _eq_bits(r.download(api.RT_BUF_ACCUMULATION), acc)
    r.close()


def test_headless_host_app_images(api, scenes, tmp_path):
    """app/restir_app (the C++ mirror of the example's main(), written against the C-ABI only) renders
    the same pixels as the ctypes host, and its PPM / PNG (key S of the example) / PFM files decode to
    the downloaded buffers."""
    import struct
    import subprocess
    import zlib

    from cedec_2024_rt_amd.types import bench_options

    app = os.path.join(ROOT, "app", "restir_app")
    assert os.path.exists(app), "app/restir_app not built: run __graft_entry__.build()"
    W, H, frames = 112, 63, 3
    tris = scenes.make_quad_room()
    tp = tmp_path / "q.tris"
    tris.tofile(tp)
    eye, at = (0.5, 2.5, 6.0), (0.0, 1.5, -1.0)
    out = subprocess.run([app, "--tris", str(tp), "--size", str(W), str(H), "--frames", str(frames),
                          "--eye", *map(str, eye), "--lookat", *map(str, at),
                          "--ppm", str(tmp_path / "o.ppm"), "--png", str(tmp_path / "o.png"), "--pfm", str(tmp_path / "o.pfm")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    r = api.Renderer(W, H)
    r.set_scene(tris)
    r.lookat(eye, at)
    r.set_options(bench_options())
    r.clear()
    for f in range(1, frames + 1):
        r.frame(f)
    px = r.download(api.RT_BUF_PIXELS).reshape(H, W, 4)[::-1, :, :3]  # top row first
    acc = r.download(api.RT_BUF_ACCUMULATION).reshape(H, W, 4)
    assert f"rays/frame: {r.ray_count()[0]}" in out.stdout
    r.close()
    # PPM
    raw = (tmp_path / "o.ppm").read_bytes()
    head = f"P6\n{W} {H}\n255\n".encode()
    assert raw.startswith(head)
    assert np.array_equal(np.frombuffer(raw[len(head):], np.uint8).reshape(H, W, 3), px)
    # PNG: signature, IHDR, one IDAT (zlib stream), IEND; CRCs checked
    raw = (tmp_path / "o.png").read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body)
        chunks.append((typ, body))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", chunks[0][1]) == (W, H, 8, 2, 0, 0, 0)
    rows = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(H, 1 + 3 * W)
    assert not rows[:, 0].any()
    assert np.array_equal(rows[:, 1:].reshape(H, W, 3), px)
    # PFM: bottom-up little-endian RGB = accumulation / spp
    raw = (tmp_path / "o.pfm").read_bytes()
    head = f"PF\n{W} {H}\n-1.0\n".encode()
    assert raw.startswith(head)
    rgb = np.frombuffer(raw[len(head):], np.float32).reshape(H, W, 3)
    assert np.array_equal(rgb.view(np.uint32), (acc[:, :, :3] / acc[:, :, 3:4]).astype(np.float32).view(np.uint32))


_SOUP_STATS = []


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("RT_SOUP_SEEDS", "10")))))
def test_random_triangle_soups_differential(api, oracle, seed):
    """Differential test on random triangle soups (sliver and degenerate triangles, coplanar overlaps,
    zero-area lights, lights facing away, camera inside the soup) with random option sets: the fused
    frame on the device == the oracle's frame, bit for bit, two frames."""
    from cedec_2024_rt_amd.types import TRIANGLE, bench_options

    rng = np.random.default_rng(1000 + seed)
    big = int(os.environ.get("RT_SOUP_SCALE", "1"))  # ad-hoc soak runs: more triangles, larger images
    n = int(rng.integers(8, 400 * big * big))
    tris = np.zeros(n, TRIANGLE)
    c = rng.normal(size=(n, 1, 3)).astype(np.float32) * np.float32(3.0)
    size = np.float32(10.0) ** rng.uniform(-2.0, 0.7, size=(n, 1, 1)).astype(np.float32)
    tris["v"] = (c + rng.normal(size=(n, 3, 3)).astype(np.float32) * size).astype(np.float32)
    # special shapes: degenerate (two equal vertices), collinear, exact duplicates, axis-aligned quads' halves
    k = n // 8
    tris["v"][:k, 1] = tris["v"][:k, 0]
    tris["v"][k:2 * k, 2] = (tris["v"][k:2 * k, 0] + (tris["v"][k:2 * k, 1] - tris["v"][k:2 * k, 0]) * np.float32(0.5)).astype(np.float32)
    tris["v"][2 * k:3 * k] = tris["v"][3 * k:4 * k]
    tris["v"][4 * k:5 * k, :, 1] = np.float32(-2.0)  # a coplanar patch (floor-like, many exact t ties)
    tris["color"] = rng.random((n, 3), dtype=np.float32)
    lights = rng.random(n) < 0.3
    lights[0] = True
    tris["emissive"][lights] = (rng.random((int(lights.sum()), 3), dtype=np.float32) * np.float32(20.0)).astype(np.float32)
    optkw = dict(
        use_temporal_resampling=int(rng.integers(0, 2)), use_spatial_resampling=int(rng.integers(0, 2)),
        use_visibility_reuse=int(rng.integers(0, 2)), use_shadowed_target_function=int(rng.integers(0, 2)),
        ris_sample_count=int(rng.integers(1, 12)), spatial_resampling_passes=int(rng.integers(0, 4)),
        spatial_resampling_sample_count=int(rng.integers(1, 7)), spatial_resampling_radius=float(rng.uniform(2.0, 30.0)),
        accumulate=int(rng.integers(0, 2)))
    W, H = int(rng.integers(20, 90 * big)), int(rng.integers(12, 60 * big))
    eye = tuple(float(v) for v in rng.normal(size=3) * 6.0)
    at = tuple(float(v) for v in rng.normal(size=3))
    r, sc, rg, opt, eyev = _setup(api, oracle, tris, W, H, eye, at, **optkw)
    st = oracle.new_state(W, H)
    for frame in (1, 2):
        cnt = oracle.new_counters()
        r.frame(frame)
        sc.frame(W, H, frame, rg, eyev, opt, st, cnt)
        vis = r.download(api.RT_BUF_VISIBILITY)
        assert _eq_bits(vis, st["vis"].reshape(vis.shape)), f"seed {seed} {optkw}: visibility frame {frame}"
        acc = r.download(api.RT_BUF_ACCUMULATION)
        ref = st["accum"].reshape(acc.shape)
        bad = (acc.view(np.uint32) != ref.view(np.uint32)).any(axis=1)
        assert not bad.any(), f"seed {seed} {optkw} frame {frame}: {int(bad.sum())} pixels, first {np.flatnonzero(bad)[:5]}: {acc[bad][:3]} vs {ref[bad][:3]}"
        assert r.ray_count()[0] == int(cnt["rays"][0])
    assert np.array_equal(r.download(api.RT_BUF_PIXELS).reshape(H, W, 4), st["pixels"])
    _SOUP_STATS.append((seed, int(r.ray_count()[1]), W * H))
    # the path tracers of 07_pt / 08_nee / 09_ris on the same soup (one launch and wavefront form)
    from cedec_2024_rt_amd.types import default_options

    ptkw = dict(max_depth=int(rng.integers(1, 7)), ris_sample_count=int(rng.integers(1, 9)),
                use_shadowed_target_function=int(rng.integers(0, 2)), sky_color=tuple(float(v) for v in rng.random(3)))
    r.set_options(default_options(**ptkw))
    popt = oracle.default_options(**ptkw)
    for example in (7, 8, 9):
        for mode in (0, 1):
            acc = np.zeros((W * H, 4), np.float32)
            cnt = oracle.new_counters()
            r.tuning(6, mode)
            r.path_trace(example, 3)
            sc.path_trace(example, W, H, 3, rg, popt, acc, cnt=cnt)
            got = r.download(api.RT_BUF_ACCUMULATION)
            assert _eq_bits(got, acc), f"seed {seed} example {example} mode {mode} {ptkw}: {(got.view(np.uint32) != acc.view(np.uint32)).any(axis=1).sum()} pixels"
            assert r.path_trace_rays() == int(cnt["rays"][0])
    r.close()


def test_random_triangle_soups_are_not_trivial():
    """(runs after the soups, same module order) most of them put shaded pixels on screen"""
    if not _SOUP_STATS:
        pytest.skip("soup tests did not run")
    frac = [s / n for _, s, n in _SOUP_STATS]
    assert np.mean(np.array(frac) > 0.05) > 0.5, _SOUP_STATS
