"""The oracle in MATH_LIBM mode reproduces, bit for bit, what the REFERENCE'S OWN code produced
(fixtures written by tests/golden/make_golden.py from oracle/_ref). Where oracle/_ref is present
(build container) the comparison is also made live on fresh random inputs."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def fn_gold(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_functions.npz"))


@pytest.fixture(scope="module")
def k_gold(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_kernels.npz"))


@pytest.fixture(scope="module")
def gscenes(golden_dir):
    return np.load(os.path.join(golden_dir, "scenes.npz"))


@pytest.fixture(autouse=True)
def _libm(oracle):
    oracle.set_math_mode(oracle.MATH_LIBM)
    yield
    oracle.set_math_mode(oracle.MATH_PORTABLE)


FNS = ["warp_unit_triangle", "sample_hemisphere", "sample_2d_gaussian", "geometry_term", "intersect_ray_triangle",
       "luminance", "normal_rejection", "depth_rejection", "triangle_props", "aces", "surface_ray", "tangent_world",
       # the arithmetic of resolve (10_restir_di.cu:433-458) through the reference's own functions and operators, the shadow ray's
       # answer V given (the kernel itself needs HIPRT): pins the expression order of the oracle's o_resolve
       "resolve_arithmetic"]


@pytest.mark.parametrize("name", FNS)
def test_functions_vs_reference_fixture(oracle, fn_gold, name):
    x, want = fn_gold[name + "_in"], fn_gold[name + "_out"]
    got = oracle.fn_bulk(name, x)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("name", FNS)
def test_functions_vs_reference_live(oracle, name):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built here")
    rng = np.random.default_rng(99)
    nin = oracle.FN[name][1]
    x = (rng.random((3000, nin), dtype=np.float32) * 6 - 3).astype(np.float32)
    if name == "intersect_ray_triangle":
        x[:, 6], x[:, 7] = 0.0, 1e30
    if name in ("sample_hemisphere", "sample_2d_gaussian", "warp_unit_triangle"):
        x = rng.random((3000, nin), dtype=np.float32)
    if name == "aces":
        x = np.abs(x)
    if name == "resolve_arithmetic":
        import importlib.util

        spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
        mg = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mg)
        x = mg.resolve_inputs(rng, 3000)
    got, want = oracle.fn_bulk(name, x), oracle.ref_fn(name, x)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def _fields_equal(a, b, mask=None):
    for f in a.dtype.names:
        if f == "pad":
            continue
        x, y = (a[f], b[f]) if mask is None else (a[f][mask], b[f][mask])
        if not np.array_equal(np.ascontiguousarray(x).view(np.uint8), np.ascontiguousarray(y).view(np.uint8)):
            return f
    return None


def test_camera(oracle, k_gold):
    cam, W, H = k_gold["cam"], int(k_gold["W"]), int(k_gold["H"])
    rg = oracle.raygen_lookat(cam[0:3], cam[3:6], cam[6:9], cam[9], W, H)
    assert rg.tobytes() == k_gold["raygen"].tobytes()
    import ctypes as C

    for (u, v), want in zip(k_gold["cam_uv"], k_gold["cam_rays"]):
        ro, rd = np.zeros(3, np.float32), np.zeros(3, np.float32)
        oracle.lib().o_raygen_shoot(rg.ctypes.data_as(C.c_void_p), C.c_float(u), C.c_float(v),
                                    ro.ctypes.data_as(C.c_void_p), rd.ctypes.data_as(C.c_void_p))
        assert np.array_equal(np.concatenate([ro, rd]).view(np.uint32), want.view(np.uint32))


def test_config1_04_ao_kernel(oracle, k_gold, gscenes):
    """Config #1: the 04_ao kernel (examples/04_ao/04_ao.cu:31-88), brute force and via the BVH."""
    W, H = int(k_gold["W"]), int(k_gold["H"])
    for use_bvh in (False, True):
        sc = oracle.Scene(gscenes["cornellbox1"], use_bvh=use_bvh)
        px = sc.ao_04(W, H, k_gold["raygen"])
        assert np.array_equal(px, k_gold["ao04_pixels"]), f"use_bvh={use_bvh}"
    assert (k_gold["ao04_pixels"][..., 0] != 32).mean() > 0.1


def test_restir_kernels_vs_reference_fixture(oracle, k_gold, gscenes):
    """generate_candidate, temporal_resampling, 3 x spatial_resampling, tone_mapping."""
    W, H = int(k_gold["W"]), int(k_gold["H"])
    tris = gscenes["cornellbox1"]
    sc = oracle.Scene(tris, use_bvh=True)
    vis, opt, eye = k_gold["vis"], k_gold["options"], k_gold["eye"]
    assert np.array_equal(sc.raycast(W, H, k_gold["raygen"])["index"], vis["index"])
    shaded = (vis["index"] >= 0) & ~np.isin(vis["index"], sc.lights)
    assert shaded.sum() > 100
    g1 = sc.generate_candidate(W, H, 1, vis, eye, opt)
    g2 = sc.generate_candidate(W, H, 2, vis, eye, opt)
    assert _fields_equal(g1, k_gold["gen_frame1"]) is None
    assert _fields_equal(g2, k_gold["gen_frame2"]) is None
    t = g2.copy()
    sc.temporal_resampling(W, H, 2, vis, eye, opt, g1, t)
    assert _fields_equal(t, k_gold["temporal_frame2"]) is None
    assert (t["M"][shaded] > 32).any()
    rin = t
    for p in range(3):
        out = sc.spatial_resampling(W, H, 2, p, vis, eye, opt, rin)
        # the reference leaves non-shaded pixels of `out` unwritten (10_restir_di.cu:275-287)
        assert _fields_equal(out, k_gold[f"spatial_frame2_pass{p}"], mask=shaded) is None, f"pass {p}"
        rin = out
    px = oracle.tone_mapping(k_gold["tone_accum"], W, H)
    assert np.array_equal(px, k_gold["tone_pixels"])


def test_restir_kernels_vs_reference_live(oracle, gscenes):
    """Fresh run of the reference's kernels on cornellbox2 (3470 triangles), another camera."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built here")
    from cedec_2024_rt_amd import scenes

    tris = gscenes["cornellbox2"]
    W, H = 40, 30
    sc = oracle.Scene(tris, use_bvh=True)
    fovy = np.float32(np.pi) / np.float32(4)
    rg = oracle.raygen_lookat(scenes.CORNELLBOX_EYE, scenes.CORNELLBOX_LOOKAT, (0, 1, 0), fovy, W, H)
    eye = np.asarray(scenes.CORNELLBOX_EYE, np.float32)
    vis = sc.raycast(W, H, rg)
    opt = oracle.bench_options(use_visibility_reuse=0, spatial_resampling_radius=12.0)
    shaded = (vis["index"] >= 0) & ~np.isin(vis["index"], sc.lights)
    kw = dict(W=W, H=H, tris=tris, vis=vis, options=opt, eye=eye)
    res = {}
    for fr in (5, 6):
        res[fr] = sc.generate_candidate(W, H, fr, vis, eye, opt)
        o = oracle.ref_run("generate_candidate", frame=fr, lights=sc.lights, **kw)
        assert _fields_equal(res[fr], np.frombuffer(o["res"], dtype=oracle.RESERVOIR)) is None
    t = res[6].copy()
    sc.temporal_resampling(W, H, 6, vis, eye, opt, res[5], t)
    o = oracle.ref_run("temporal_resampling", frame=6, prev=res[5], res=res[6], **kw)
    assert _fields_equal(t, np.frombuffer(o["res"], dtype=oracle.RESERVOIR)) is None
    rin = t
    for p in range(2):
        out = sc.spatial_resampling(W, H, 6, p, vis, eye, opt, rin)
        o = oracle.ref_run("spatial_resampling", frame=6, res=rin, **{"pass": p}, **kw)
        assert _fields_equal(out, np.frombuffer(o["res"], dtype=oracle.RESERVOIR), mask=shaded) is None
        rin = out


# ---------------------------------------------------------------- round 2: interactive camera, config #1 at 256x256
def _camera_rows(oracle, start, events, fovy):
    """the oracle's CameraControl + RayGenerator::lookat over a drag sequence, as uint32 bit patterns"""
    e, a = start[:3].astype(np.float32), start[3:6].astype(np.float32)
    W, H = int(start[6]), int(start[7])
    rows = []
    for b, dx, dy in events:
        e, a, upd = oracle.camera_control(e, a, int(b), dx, dy)
        rg = oracle.raygen_lookat(e, a, (0, 1, 0), fovy, W, H)
        rows.append([int(v) for v in e.view(np.uint32)] + [int(v) for v in a.view(np.uint32)] + [int(upd)]
                    + [int(v) for v in rg.view(np.uint32).reshape(-1)])
    return np.array(rows, dtype=np.uint64)


def test_camera_control_vs_reference_fixture(oracle, golden_dir):
    """CameraControl::cursorPosCallback (common/misc.hpp:129-205) + RayGenerator::lookat (common/camera.hpp:11-25):
    the oracle's restatement == the reference's own code over 3 x 120 random orbit / zoom / pan drags, incl. the
    pole clamp and the zoom floor, bit for bit (eye, look-at, updated flag, raygen)."""
    g = np.load(os.path.join(golden_dir, "ref_camera.npz"))
    fovy = g["fovy"][0]
    for i in range(3):
        got = _camera_rows(oracle, g[f"pose{i}_start"], g[f"pose{i}_events"], fovy)
        want = g[f"pose{i}_out"]
        assert got.shape == want.shape and np.array_equal(got, want), f"pose {i}: first bad event {np.flatnonzero((got != want).any(axis=1))[:3]}"
    # the two special events really exercised their branches
    ev = g["pose0_events"]
    assert ev[5][2] == 5000.0 and ev[6][2] == 1e6


def test_camera_control_vs_reference_live(oracle):
    if not os.path.exists(oracle.REF_CAMERA_BIN):
        pytest.skip("oracle/_ref/ref_camera not built here")
    rng = np.random.default_rng(5)
    ev = np.zeros((300, 3), np.float32)
    ev[:, 0] = rng.integers(0, 3, 300)
    ev[:, 1:] = (rng.normal(size=(300, 2)) * 120).astype(np.float32)
    start = np.array([3.0, 4.0, -5.0, 0.5, 0.25, 1.0, 640, 360], np.float32)
    fovy = np.float32(np.pi) / np.float32(4)
    want = oracle.ref_camera_run(start[:3], start[3:6], 640, 360, fovy, [(int(b), float(x), float(y)) for b, x, y in ev])
    assert np.array_equal(_camera_rows(oracle, start, ev, fovy), want)


def test_config1_04_ao_at_256x256(oracle, golden_dir, gscenes):
    """BASELINE config #1 at its real size: 04_ao kernelMain (examples/04_ao/04_ao.cu:31-88) on cornellbox1.obj,
    256x256, default camera — the oracle's host loop == the reference's own kernel run on the host, every byte."""
    g = np.load(os.path.join(golden_dir, "ref_ao04_256.npz"))
    W, H = int(g["W"]), int(g["H"])
    assert (W, H) == (256, 256)
    sc = oracle.Scene(gscenes["cornellbox1"], use_bvh=False)  # brute force, as 04_ao.cu:8-29
    from cedec_2024_rt_amd import scenes

    rg = oracle.raygen_lookat(scenes.DEFAULT_EYE, scenes.DEFAULT_LOOKAT, (0, 1, 0), np.float32(np.pi) / np.float32(4), W, H)
    assert rg.tobytes() == g["raygen"].tobytes()
    px = sc.ao_04(W, H, rg)
    assert np.array_equal(np.asarray(px).reshape(H, W, 4), g["pixels"])
    # and the oracle's BVH gives the same image as its brute force (the intersection definition)
    sc2 = oracle.Scene(gscenes["cornellbox1"], use_bvh=True)
    assert np.array_equal(np.asarray(sc2.ao_04(W, H, rg)).reshape(H, W, 4), g["pixels"])
