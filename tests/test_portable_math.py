"""portable_math.h: accuracy against glibc (the functions replace device libm, DESIGN.md
"Parity") and the effect of the substitution on a whole frame."""
import numpy as np


def _ulp_err(a, b):
    """error of a against reference b in ulps of b"""
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    ulp = np.spacing(np.abs(b).astype(np.float32)).astype(np.float64)
    return np.abs(a64 - b64) / ulp


def _both(oracle, name, x):
    oracle.set_math_mode(oracle.MATH_PORTABLE)
    p = oracle.fn_bulk(name, x).ravel()
    oracle.set_math_mode(oracle.MATH_LIBM)
    l = oracle.fn_bulk(name, x).ravel()
    oracle.set_math_mode(oracle.MATH_PORTABLE)
    return p, l


def test_accuracy_vs_glibc(oracle):
    rng = np.random.default_rng(0)
    u = rng.random(400000, dtype=np.float32)
    u = u[u > 0]
    p, l = _both(oracle, "logf", u)
    assert _ulp_err(p, l).max() <= 1.0
    for name in ("cosf", "sinf"):
        x = (rng.random(400000, dtype=np.float32) * 12.566371 - 3.1415927).astype(np.float32)
        p, l = _both(oracle, name, x)
        big = np.abs(l) > 1e-3   # near the zeros compare absolutely
        assert _ulp_err(p[big], l[big]).max() <= 1.0
        assert np.abs(p[~big].astype(np.float64) - l[~big]).max() < 1e-9
    x = (-rng.random(400000, dtype=np.float32) * 90).astype(np.float32)
    p, l = _both(oracle, "expf", x)
    assert _ulp_err(p, l).max() <= 1.0
    p, l = _both(oracle, "pow8", u)
    assert _ulp_err(p, l).max() <= 6.0   # three squarings: 3.5 ulp relative, binade edges add a factor
    p, l = _both(oracle, "pow_gamma", (u * 4).astype(np.float32))
    assert _ulp_err(p, l).max() <= 16.0  # display-only path (tone mapping)


def test_fused_sincos_equals_the_two_functions(oracle):
    """pm_sincosf (what the device kernels call) == (pm_sinf, pm_cosf) bit for bit: in [0, 2 pi) where
    the spatial pass uses it, over a wide range, and on the special values."""
    oracle.set_math_mode(oracle.MATH_PORTABLE)
    rng = np.random.default_rng(7)
    x = np.concatenate([
        (rng.random(500000, dtype=np.float32) * np.float32(6.2831855)).astype(np.float32),
        ((rng.random(300000, dtype=np.float32) - np.float32(0.5)) * np.float32(2000.0)).astype(np.float32),
        np.float32([0.0, -0.0, 1e-30, -1e-30, 1.5707964, 3.1415927, 4.712389, 6.2831855, 5e8, 6e8, -7e8, np.inf, -np.inf, np.nan]),
        (np.arange(-64, 65, dtype=np.float32) * np.float32(np.pi / 4)).astype(np.float32),
    ])
    f = oracle.fn_bulk
    s1, c1 = f("sinf", x).reshape(-1), f("cosf", x).reshape(-1)
    s2, c2 = f("sincos_sin", x).reshape(-1), f("sincos_cos", x).reshape(-1)
    assert np.array_equal(s1.view(np.uint32), s2.view(np.uint32))
    assert np.array_equal(c1.view(np.uint32), c2.view(np.uint32))


def test_special_values(oracle):
    oracle.set_math_mode(oracle.MATH_PORTABLE)
    f = oracle.fn_bulk
    assert np.isneginf(f("logf", [0.0])[0, 0]) and f("logf", [1.0])[0, 0] == 0.0 and np.isnan(f("logf", [-1.0])[0, 0])
    assert f("expf", [0.0])[0, 0] == 1.0 and f("expf", [-200.0])[0, 0] == 0.0 and np.isposinf(f("expf", [100.0])[0, 0])
    assert f("cosf", [0.0])[0, 0] == 1.0 and f("sinf", [0.0])[0, 0] == 0.0
    assert f("pow_gamma", [0.0])[0, 0] == 0.0 and f("pow_gamma", [1.0])[0, 0] == 1.0
    g = f("sample_2d_gaussian", [[0.0, 0.3]])[0]   # rv0 = 0 -> radius = +inf
    assert not np.isfinite(g).any()


def test_portable_vs_libm_frame_within_the_north_star_tolerance(oracle):
    """The link between what the GPU computes (portable transcendental functions) and what the reference computes
    (glibc: MATH_LIBM == oracle/_ref bit for bit): two full frames of the BENCHMARK workload (blocks_restir stand-in,
    1920x1080, bench options) in both modes. The two may flip a handful of discrete reservoir decisions (a <= 1 ulp
    difference in log/exp/sin/cos landing on the other side of a comparison); the radiance must agree within the
    north star's 1e-4 relative L2 (BASELINE.json), at the benchmark's own size."""
    from cedec_2024_rt_amd import scenes

    tris = scenes.make_blocks_restir()
    W, H = 1920, 1080
    eye, center = scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT
    rg = oracle.raygen_lookat(eye, center, (0, 1, 0), np.float32(np.pi) / np.float32(4), W, H)
    acc = {}
    for mode in (oracle.MATH_LIBM, oracle.MATH_PORTABLE):
        oracle.set_math_mode(mode)
        sc = oracle.Scene(tris, use_bvh=True)
        st = oracle.new_state(W, H)
        per_frame = []
        for fr in (1, 2):
            sc.frame(W, H, fr, rg, np.asarray(eye, np.float32), oracle.bench_options(), st, tone_map=False)
            per_frame.append(st["accum"].copy())
        acc[mode] = per_frame
    oracle.set_math_mode(oracle.MATH_PORTABLE)
    for fr in (0, 1):
        a, b = acc[oracle.MATH_LIBM][fr].astype(np.float64), acc[oracle.MATH_PORTABLE][fr].astype(np.float64)
        flipped = int((acc[oracle.MATH_LIBM][fr] != acc[oracle.MATH_PORTABLE][fr]).any(axis=1).sum())
        rel = float(np.sqrt(((a[:, :3] - b[:, :3]) ** 2).sum()) / np.sqrt((a[:, :3] ** 2).sum()))
        print(f"frame {fr + 1}: {flipped} of {W * H} pixels differ between libm and portable math, rel-L2 {rel:.3e}")
        assert rel <= 1e-4, (fr, flipped, rel)
        assert flipped < 2000  # reported above; the gate is the radiance tolerance


def test_logf_expf_exhaustive_vs_double_precision(tmp_path):
    """portable_math.h pinned independently of itself. The oracle takes its transcendental functions from the
    header the GPU uses, so a defect there would be invisible to GPU-vs-oracle; tools/logexp_exhaustive.c evaluates pm_logf on
    EVERY binary32 in (0, 1] and (1, 4] and pm_expf on EVERY binary32 in [-104, 0] and [0, 0.1] (3.2e9 arguments: the ranges of the
    Box-Muller radius, the depth rejection heuristics and the tone-mapping power) against glibc's binary64 log / exp: <= 1 ulp of
    the exact value everywhere (measured: 0.83 / 0.83 / 0.91 / 0.56 ulp; profiles/r05_logexp_exhaustive.txt). ~8 s on 8 cores."""
    import os
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "logexp_exhaustive")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-o", exe, os.path.join(root, "tools", "logexp_exhaustive.c"), "-lm"])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=1200)
    print(p.stdout)
    assert p.returncode == 0, p.stdout + p.stderr
    errs = [float(v) for v in re.findall(r"max error ([0-9.]+) ulp", p.stdout)]
    counts = [int(v) for v in re.findall(r"(\d+) arguments", p.stdout)]
    assert len(errs) == 4 and max(errs) <= 1.0 and "strided" not in p.stdout
    assert counts[0] == 0x3f800000 and counts[2] > 1_100_000_000  # every binary32 of (0, 1]; of [-104, -0]


def test_steady_state_drift_report_is_within_the_tolerance():
    """the libm-vs-portable link over a 30-frame 1080p sequence is REPORTED (tools/portable_drift.py ->
    profiles/r05_portable_drift.json; two minutes of CPU, not re-run here); this only checks that the committed report is what the
    documents quote: every frame's rel-L2 far inside the 1e-4 contract, no growth with the frame number (the history is saved
    BEFORE the spatial passes, 10_restir_di.cpp:314-321, so the passes' flipped decisions never feed back), zero differing
    histories."""
    import json
    import os

    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_portable_drift.json")
    d = json.load(open(p))
    fr = d["frames"]
    assert len(fr) >= 30 and fr[0]["frame"] == 1
    assert d["max_rel_l2"] == max(f["rel_l2"] for f in fr) <= 1e-5
    assert all(f["hist"] == 0 for f in fr)
    assert max(f["flipped"] for f in fr) < 2500
