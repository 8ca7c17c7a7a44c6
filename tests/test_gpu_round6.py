"""Round-6 GPU tests.

* rt_timing brackets the ONE-launch stage 0 the headline frames run (rt_tuning 25): the timed frame is still the
  oracle's frame (examples/10_restir_di/10_restir_di.cu:9-135 behind 10_restir_di.cpp:257-311), rt_stage0_one_launch says which form
  ran, and the per-kernel entries add up to the frame;
* the measurement libraries with the device libm the reference gets from hiprtc (ocml: common/reservoir.hpp:61-95,
  common/kernels/common.cu:58-61) stay within the north star's 1e-4 relative L2 of the product per frame;
* the four-lanes-per-ray closest-hit walk (common/raytrace.hpp:18-43 for launches that cannot fill the GPU with one lane per ray).
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FOVY = np.float32(np.pi) / np.float32(4)


def _eq_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


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
        sc.frame(W, H, f, rg, eyev, opt, st)
    return st


def test_timed_frames_run_the_one_launch_stage0_and_match_the_oracle(api, oracle, scenes):
    from cedec_2024_rt_amd.types import bench_options

    W, H, frames = 480, 270, 6
    tris = scenes.make_blocks_restir()
    eye, at = scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT
    r = api.Renderer(W, H)
    r.set_scene(tris)
    r.lookat(eye, at)
    r.set_options(bench_options())
    r.tuning(16, 0)  # a 480 x 270 launch is "about one generation of wavefronts": auto would give its primary rays the strips' work-sharing walk, two launches
    r.tuning(13, 1)  # the defaults this test is about, whatever RT_TUNING (soak runs) has set on the context
    r.tuning(25, -1)
    r.timing_enable(True)
    forms = []
    for f in range(1, frames + 1):
        if f == 4:
            r.tuning(25, 0)  # the reference's two kernels under the same events
        r.frame(f)
        t = r.timing()
        forms.append(r.stage0_one_launch())
        parts = sum(t[k] for k in ("clear", "raycast", "generate_candidate", "spatial0", "spatial1", "spatial2", "resolve", "tone_mapping"))
        assert abs(parts - t["frame"]) <= 0.02 * t["frame"] + 0.005, (parts, t)
        if forms[-1]:
            assert t["raycast"] < 0.25 * t["generate_candidate"], t  # the empty bracket where the raycast launch would be
    assert forms == [True, True, True, False, False, False], forms
    st = _oracle_frames(oracle, tris, W, H, eye, at, frames)
    acc = r.download(api.RT_BUF_ACCUMULATION)
    assert _eq_bits(acc, st["accum"].reshape(acc.shape)), int((acc != st["accum"].reshape(acc.shape)).any(axis=1).sum())
    r.close()


def test_ocml_device_libm_stays_inside_the_north_star_tolerance(api, scenes):
    """what the reference computes when hiprtc compiles it for this GPU (ocml's log / exp / sin / cos / pow; with and without the
    default FMA contraction) against the product, 12 frames at 480 x 270: a few pixels flip, the frame stays within 1e-4 rel-L2"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ocml_drift

    for p in (api.OCML_LIB_PATH, api.OCML_FMA_LIB_PATH):
        assert os.path.exists(p), f"{p} not built: __graft_entry__.build() (make -C cedec_2024_rt_amd/csrc ocml)"
    rows, ids = ocml_drift.run(api, scenes, 480, 270, 12, verbose=False)
    assert ids["ocml"].endswith("-ocml") and ids["ocml_fma"].endswith("-ocml-fma") and "-" not in ids["product"].replace("-exp", "")
    worst = max(r["ocml"]["rel_l2"] for r in rows)
    assert worst <= 1e-4, (worst, [r["ocml"] for r in rows])
    assert max(r["ocml"]["hist"] for r in rows) == 0  # the history is saved before the passes that call the functions: nothing feeds back
    # With hipcc's default FMA contraction on top (what hiprtc gives the reference's kernels) EVERY product-and-add of the frame rounds
    # once instead of twice: reservoir weights differ in their last bits at most pixels and some selections flip. That is compiler-
    # flag drift of the reference against ITSELF (its own kernels under -ffp-contract=off vs fast), reported by tools/ocml_drift.py
    # (profiles/r06_ocml_drift.json), not a property of the transcendental functions: bounded loosely here, gated nowhere.
    assert max(r["ocml_fma"]["rel_l2"] for r in rows) <= 0.1
    # not vacuous: the measurement library really evaluates other functions — ocml's logf / expf / sinf differ from
    # portable_math.h's in the last bit for some of 200 000 arguments of the renderer's ranges
    rng = np.random.default_rng(6)
    a, b = api.Renderer(16, 16), api.Renderer(16, 16, lib_path=api.OCML_LIB_PATH)
    differ = 0
    for fn, x in ((20, rng.uniform(2.0 ** -23, 1.0, 200000)), (23, rng.uniform(-60.0, 0.0, 200000)), (22, rng.uniform(0.0, 2 * np.pi, 200000))):
        ya, yb = a.math_eval(fn, x.astype(np.float32)), b.math_eval(fn, x.astype(np.float32))
        differ += int((ya.view(np.uint32) != yb.view(np.uint32)).sum())
        assert np.allclose(ya, yb, rtol=1e-6, atol=1e-7)
    a.close()
    b.close()
    assert differ > 0


def test_halo_marks_one_workgroup_per_tile_and_pass_give_the_same_plans(api, scenes):
    """rt_tuning 26 (r06): k_halo_mark as one workgroup per (tile, pass) == one workgroup per tile replaying the passes in series
    (r02-r05), with and without the LDS window (key 19), marks of 1, 2 and 3 passes per call, both sides"""
    import torch

    from cedec_2024_rt_amd.types import bench_options

    W, H = 480, 270
    tris = scenes.make_blocks_restir()
    bounds = api.mg_partition(H, 3)
    rank = 1
    out = {}
    for split, window in ((1, 1), (0, 1), (1, 0), (0, 0)):
        c = api.Renderer(W, H, rows=bounds[rank], halo=87)
        c.tuning(26, split)
        c.tuning(19, window)
        assert c.tuning_get(26) == split  # default 0 (measured: no gain)
        c.set_scene(tris)
        c.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
        c.set_options(bench_options())
        words = c.halo_bitmap_words(87)
        fb = c.halo_flags_bytes(87)
        c.raycast()
        flags = torch.zeros(fb, dtype=torch.uint8, device="cuda")
        for src, dst in ((bounds[rank][0], bounds[rank][0] - 87), (bounds[rank][1] - 87, bounds[rank][1])):
            c.halo_flags_pack(src, 87, flags.data_ptr())
            c.sync()
            c.halo_flags_unpack(dst, 87, flags.data_ptr())
        got = []
        for frame, first, n in ((3, 0, 3), (4, 0, 3), (5, 1, 2), (6, 2, 1)):
            bm = torch.zeros((2, 3, words), dtype=torch.int32, device="cuda")
            c._ck(c.L.rt_halo_mark_sides(c.h, frame, first, n, bm[0].data_ptr(), bm[1].data_ptr()))
            c.sync()
            got.append(bm.cpu().numpy()[:, :n].copy())
        out[(split, window)] = got
        c.close()
    ref = out[(0, 0)]
    assert any(int(a[:, :, 0].sum()) > 0 for a in ref)  # something was marked at all
    for key, got in out.items():
        for a, b in zip(ref, got):
            assert np.array_equal(a, b), key


def test_four_lanes_per_primary_ray_writes_the_same_gbuffer(api, oracle, scenes):
    """rt_tuning 16 = 2 (k_raycast_quad, bvh.h closest_quad): rt_raycast with 16 rays per wavefront, each lane one child box of the
    4-wide record == the one-lane-per-ray walks, Visibility records byte for byte (whole image, a strip with a ragged width, a tiny
    image), and whole strip frames through rt_frame_stage against the oracle (10_restir_di.cu:9-34 behind common/raytrace.hpp:18-43)"""
    from cedec_2024_rt_amd.types import bench_options

    tris = scenes.make_blocks_restir()
    eye, at = scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT
    for W, H, rows in ((480, 270, None), (1918, 1080, (405, 540)), (37, 19, None)):
        vis = {}
        for mode in (0, 1, 2):
            r = api.Renderer(W, H, rows=rows, halo=87 if rows else 0, exp=True)  # 16 = 2: librestir_rt_exp.so
            r.tuning(16, mode)
            r.set_scene(tris)
            r.lookat(eye, at)
            r.set_options(bench_options())
            r.raycast()
            v = r.download(api.RT_BUF_VISIBILITY)
            if rows:
                v = v.reshape(r.local_rows, W)[rows[0] - r.local_row0: rows[1] - r.local_row0]
            vis[mode] = np.ascontiguousarray(v).copy()
            r.close()
        assert _eq_bits(vis[2], vis[0]) and _eq_bits(vis[1], vis[0]), (W, H, rows)
        assert (vis[0]["index"] >= 0).mean() > 0.5
    # whole frames: three LOCAL strips, every strip's primary rays through the quad walk, against the oracle
    W, H, frames = 480, 270, 5
    bounds = api.mg_partition(H, 3)
    ctxs = []
    for b in bounds:
        c = api.Renderer(W, H, rows=b, halo=87, exp=True)
        c.tuning(16, 2)
        c.set_scene(tris)
        c.lookat(eye, at)
        c.set_options(bench_options())
        ctxs.append(c)
    hub = api.MgHub(3, renderer=ctxs[0])
    mgs = [api.MultiGpu(c, k, bounds, transport=api.RT_MG_TRANSPORT_LOCAL, hub=hub) for k, c in enumerate(ctxs)]
    for f in range(1, frames + 1):
        api.mg_frame_lockstep(mgs, f)
    st = _oracle_frames(oracle, tris, W, H, eye, at, frames)
    ref = st["accum"].reshape(H, W, 4)
    for c, (a, b) in zip(ctxs, bounds):
        acc = c.download(api.RT_BUF_ACCUMULATION).reshape(c.local_rows, W, 4)[a - c.local_row0: b - c.local_row0]
        assert _eq_bits(acc, ref[a:b]), f"rows {a}:{b}: {int((acc != ref[a:b]).any(axis=2).sum())} pixels differ from the oracle"
    for m in mgs:
        m.close()
    hub.close()
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_look_ahead_soak_camera_moves_key22_toggles_and_per_kernel_calls_between_strip_frames(api, scenes, seed):
    """the free-running look-ahead (rt_tuning 22, the strips' default) rests on invariants about which G-buffer set and
    which reservoir buffer stage 0 of frame f+1 may overwrite; rt_frame_stage now checks them where it writes. This soak drives three
    LOCAL strips and a single context through 24 frames with, at random between frames: camera moves, frame-number jumps, key 22
    toggled (on / off / auto) on every context, key 14 toggled, and per-kernel entry points called on every context (raycast alone;
    raycast + generate_candidate into a named buffer; resolve + tone_mapping of the last frame again) — every call that must make a
    queued look-ahead stale. Every frame of every strip == the single context, accumulation, pixels and temporal history."""
    import test_mg_native as tm

    rng = np.random.default_rng(2200 + seed)
    tris = scenes.make_quad_room()
    W, H, n = 96, 330, 3
    rig = tm._Rig(api, tris, W, H, n, (0.5, 2.5, 6.0), (0.0, 1.5, -1.0), dict(), 0)
    frame = 0
    for step in range(24):
        frame += 1 if rng.integers(0, 6) else int(rng.integers(2, 5))  # now and then a jump: the plan and the look-ahead are for another frame
        clear = False
        ev = int(rng.integers(0, 8))
        if ev == 0:
            dx, dy = float(rng.uniform(-40, 40)), float(rng.uniform(-15, 15))
            for r in rig.everyone():
                r.orbit(dx, dy)
            clear = True
        elif ev == 1:
            v = int(rng.choice([-1, 0, 1]))
            for r in rig.everyone():
                r.tuning(22, v)
        elif ev == 2:
            v = int(rng.choice([-1, 0, 1, 2]))
            for r in rig.everyone():
                r.tuning(14, v)
        elif ev == 3:
            for r in rig.everyone():
                r.raycast()
        elif ev == 4 and frame > 1:
            for r in rig.everyone():
                r.raycast()
                r.generate_candidate(frame, api.RT_RES_1)  # a reservoir buffer changes outside the staged frame
        elif ev == 5 and frame > 1:
            for r in rig.everyone():
                r.resolve(api.RT_RES_1)
                r.tone_mapping()
        rig.frame(frame, clear)
        what = f"seed {seed} step {step} frame {frame} event {ev}"
        rig.check(what)
        rig.check_history(what)
    rig.close()


def test_mirror_wire_transport_moves_what_mirror_moves_and_takes_the_wires_time(api, scenes):
    """RT_MG_TRANSPORT_MIRROR_WIRE (r06) = MIRROR + the modelled link as a dependent delay: the same images and bytes as MIRROR, the same
    modelled time as WIRE_MODEL asks for (it charges the wire without RCCL's self-send in front of it), and with a slow modelled link
    the frames really take that long"""
    import time

    from cedec_2024_rt_amd.types import bench_options

    W, H = 480, 270
    tris = scenes.make_blocks_restir()
    bounds = api.mg_partition(H, 3)
    imgs, stats, wall = {}, {}, {}
    for name, T, env in (("mirror", api.RT_MG_TRANSPORT_MIRROR, {}), ("wire_model", api.RT_MG_TRANSPORT_WIRE_MODEL, {}),
                         ("mirror_wire", api.RT_MG_TRANSPORT_MIRROR_WIRE, {}),
                         ("slow", api.RT_MG_TRANSPORT_MIRROR_WIRE, {"RT_MG_WIRE_GBS": "0.5", "RT_MG_WIRE_LAT_US": "200"})):
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
    assert _eq_bits(imgs["mirror"], imgs["mirror_wire"]) and _eq_bits(imgs["mirror"], imgs["slow"])
    assert stats["mirror"]["wire_ns"] == 0 and stats["mirror_wire"]["wire_ns"] == stats["wire_model"]["wire_ns"] > 0
    assert stats["mirror_wire"]["bytes_sent"] == stats["mirror"]["bytes_sent"]
    per_frame_model = stats["slow"]["wire_ns"] / 6 * 1e-9
    assert per_frame_model > 3 * 200e-6
    assert wall["slow"] >= 0.8 * per_frame_model, (wall, per_frame_model)  # a sanity bound: the host clock against the GPU's
