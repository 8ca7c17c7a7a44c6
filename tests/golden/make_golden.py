"""Generates the committed golden fixtures from the REFERENCE'S OWN code.

Run in the build container only (needs /root/reference and oracle/_ref/ref_kernels, i.e.
`make -C oracle`):   python tests/golden/make_golden.py

Outputs (data only — inputs and expected outputs, no reference source):
  scenes.npz        triangle arrays of assets/cornellbox1.obj / cornellbox2.obj as the reference's
                    loader orders them (own OBJ reader, counts pinned in tests), MIT-licensed assets
  ref_kat.json      integer / struct-size known answers printed by the reference's functions
  ref_functions.npz random inputs + outputs of the reference's inline functions (glibc math)
  ref_kernels.npz   inputs + outputs of the reference's kernels run on the host at 48x27:
                    04_ao kernelMain, generate_candidate (no visibility reuse: that needs HIPRT),
                    temporal_resampling, 3 x spatial_resampling, tone_mapping
  ref_ao04_256.npz  BASELINE config #1 at its real size: 04_ao kernelMain, cornellbox1.obj, 256x256 (RGBA8)
  ref_camera.npz    the reference's CameraControl (common/misc.hpp:108-224) + RayGenerator::lookat over
                    sequences of random orbit / zoom / pan drags: eye, look-at, updated flag, raygen bits
  ref_tinyobj.npz   triangle arrays the reference's vendored tinyobjloader + common/loader.hpp:25-64 loop
                    produce for assets/cornellbox1.obj and assets/blocks_ao.obj
  assets/           those two OBJ/MTL pairs themselves (MIT-licensed data files of the reference, (c) 2024
                    Kenta Eto, see assets/ATTRIBUTION): the INPUTS of the OBJ-reader test on the GPU box
The oracle (MATH_LIBM mode) must reproduce every output bit for bit: tests/test_oracle_golden.py.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from cedec_2024_rt_amd import scenes  # noqa: E402
from oracle import binding as ob  # noqa: E402

REF = "/root/reference"


def resolve_inputs(rng, n):
    """operands of resolve's arithmetic (10_restir_di.cu:433-458) as a frame has them: Kd in [0,1], surface / light positions a few
    units apart, unit normals, radiance up to 120 (assets/blocks_restir.mtl Ke), ucw from tiny to large, V in {0, 1} (and a few other
    values: the expression multiplies by it), both accumulate settings, previous accumulation values, plus zeros / coincident points"""
    x = np.zeros((n, 25), np.float32)
    x[:, 0:3] = rng.random((n, 3), dtype=np.float32)
    x[:, 3:6] = (rng.random((n, 3), dtype=np.float32) * 40 - 20).astype(np.float32)
    nn = rng.standard_normal((n, 3)).astype(np.float32)
    x[:, 6:9] = nn / np.linalg.norm(nn, axis=1, keepdims=True).astype(np.float32)
    x[:, 9:12] = x[:, 3:6] + (rng.standard_normal((n, 3)) * np.float32(6)).astype(np.float32)
    nn = rng.standard_normal((n, 3)).astype(np.float32)
    x[:, 12:15] = nn / np.linalg.norm(nn, axis=1, keepdims=True).astype(np.float32)
    x[:, 15:18] = (rng.random((n, 3), dtype=np.float32) * np.float32(120)).astype(np.float32)
    x[:, 18] = np.exp(rng.random(n) * 30 - 20).astype(np.float32)
    x[:, 19] = (rng.random(n) < 0.7).astype(np.float32)
    x[::17, 19] = rng.random(len(x[::17]), dtype=np.float32)
    x[:, 20] = (rng.random(n) < 0.5).astype(np.float32)
    x[:, 21:24] = (rng.random((n, 3), dtype=np.float32) * np.float32(50)).astype(np.float32)
    x[:, 24] = np.floor(rng.random(n) * 64).astype(np.float32)
    x[::29, 9:12] = x[::29, 3:6]      # light sample on the surface point: G = 0/0
    x[::31, 18] = 0.0                 # ucw = 0 (p_hat = 0)
    x[::37, 15:18] = 0.0
    return x


def main():
    assert ob.have_ref(), "oracle/_ref/ref_kernels missing: make -C oracle"
    c1 = scenes.load_obj(os.path.join(REF, "assets/cornellbox1.obj"))
    c2 = scenes.load_obj(os.path.join(REF, "assets/cornellbox2.obj"))
    assert len(c1) == 36 and len(c2) == 3470
    np.savez_compressed(os.path.join(HERE, "scenes.npz"), cornellbox1=c1, cornellbox2=c2)

    with open(os.path.join(HERE, "ref_kat.json"), "w") as f:
        json.dump(ob.ref_run("kat"), f, indent=1)

    rng = np.random.default_rng(20240823)
    fn = {}
    for name, (fid, nin, nout) in ob.FN.items():
        if fid >= 20:
            continue
        if name == "resolve_arithmetic":  # (r05) a generator of its own: the fixtures of rounds 1-4 keep their random streams
            x = resolve_inputs(np.random.default_rng(20251003), 4000)
            fn[name + "_in"] = x
            fn[name + "_out"] = ob.ref_fn(name, x)
            continue
        x = rng.random((2000, nin), dtype=np.float32)
        if name == "intersect_ray_triangle":
            x = (x * 2 - 1).astype(np.float32)
            x[:, 6] = 0.0
            x[:, 7] = 1e30
        elif name in ("geometry_term", "normal_rejection", "depth_rejection", "triangle_props", "surface_ray", "tangent_world"):
            x = (x * 4 - 2).astype(np.float32)
        elif name == "aces":
            x = (x * 8).astype(np.float32)
        fn[name + "_in"] = x
        fn[name + "_out"] = ob.ref_fn(name, x)
    np.savez_compressed(os.path.join(HERE, "ref_functions.npz"), **fn)
    if "--only-functions" in sys.argv:
        return

    # kernels on cornellbox1, default camera (common/misc.hpp:217-218), 48x27
    W, H = 48, 27
    ob.set_math_mode(ob.MATH_LIBM)
    fovy = np.float32(np.pi) / np.float32(4)
    cam = np.array(list(scenes.DEFAULT_EYE) + list(scenes.DEFAULT_LOOKAT) + [0, 1, 0, fovy], dtype=np.float32)
    uv = rng.random((64, 2), dtype=np.float32)
    o = ob.ref_run("camera", cam=cam, W=W, H=H, uv=uv)
    rg = np.frombuffer(o["raygen"], dtype=ob.RAYGEN).copy()
    rays = np.frombuffer(o["rays"], dtype=np.float32).reshape(-1, 6).copy()
    eye = np.asarray(scenes.DEFAULT_EYE, dtype=np.float32)

    k = dict(W=W, H=H, raygen=rg, cam=cam, cam_uv=uv, cam_rays=rays, eye=eye)
    o = ob.ref_run("ao04", W=W, H=H, tris=c1, raygen=rg)
    k["ao04_pixels"] = np.frombuffer(o["pixels"], dtype=np.uint8).reshape(H, W, 4).copy()

    # visibility buffer: by the pinned definition (brute force of core.hpp:91-136); the reference
    # cannot produce it without HIPRT. It is an INPUT of the fixtures below.
    sc = ob.Scene(c1, use_bvh=False)
    vis = sc.raycast(W, H, rg)
    k["vis"] = vis
    opt = ob.bench_options(use_visibility_reuse=0)
    k["options"] = opt
    lights = sc.lights
    o = ob.ref_run("generate_candidate", W=W, H=H, frame=1, tris=c1, vis=vis, options=opt, eye=eye, lights=lights)
    gen1 = np.frombuffer(o["res"], dtype=ob.RESERVOIR).copy()
    o = ob.ref_run("generate_candidate", W=W, H=H, frame=2, tris=c1, vis=vis, options=opt, eye=eye, lights=lights)
    gen2 = np.frombuffer(o["res"], dtype=ob.RESERVOIR).copy()
    gen1["pad"] = 0
    gen2["pad"] = 0
    k["gen_frame1"], k["gen_frame2"] = gen1, gen2
    # frame 2: temporal with frame-1 candidates as history
    o = ob.ref_run("temporal_resampling", W=W, H=H, frame=2, tris=c1, vis=vis, options=opt, eye=eye, prev=gen1, res=gen2)
    tmp = np.frombuffer(o["res"], dtype=ob.RESERVOIR).copy()
    k["temporal_frame2"] = tmp
    rin = tmp
    for p in range(3):
        o = ob.ref_run("spatial_resampling", W=W, H=H, frame=2, tris=c1, vis=vis, options=opt, eye=eye, res=rin, **{"pass": p})
        out = np.frombuffer(o["res"], dtype=ob.RESERVOIR).copy()
        k[f"spatial_frame2_pass{p}"] = out
        rin = out
    acc = rng.random((W * H, 4), dtype=np.float32) * np.float32(4.0)
    acc[:, 3] = np.float32(1.0) + np.floor(acc[:, 3])
    o = ob.ref_run("tone_mapping", W=W, H=H, accum=acc)
    k["tone_accum"] = acc
    k["tone_pixels"] = np.frombuffer(o["pixels"], dtype=np.uint8).reshape(H, W, 4).copy()
    np.savez_compressed(os.path.join(HERE, "ref_kernels.npz"), **k)

    # BASELINE config #1 at its real size (256x256): the reference's 04_ao kernel on cornellbox1, default camera
    W2, H2 = 256, 256
    cam2 = np.array(list(scenes.DEFAULT_EYE) + list(scenes.DEFAULT_LOOKAT) + [0, 1, 0, fovy], dtype=np.float32)
    o = ob.ref_run("camera", cam=cam2, W=W2, H=H2, uv=uv)
    rg2 = np.frombuffer(o["raygen"], dtype=ob.RAYGEN).copy()
    o = ob.ref_run("ao04", W=W2, H=H2, tris=c1, raygen=rg2)
    np.savez_compressed(os.path.join(HERE, "ref_ao04_256.npz"), W=W2, H=H2, raygen=rg2,
                        pixels=np.frombuffer(o["pixels"], dtype=np.uint8).reshape(H2, W2, 4).copy())

    # interactive camera: random drag sequences from three start poses
    cam = {}
    poses = [(scenes.DEFAULT_EYE, scenes.DEFAULT_LOOKAT, 1920, 1080), (scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT, 1920, 1080),
             ((0.5, 2.5, 6.0), (0.0, 1.5, -1.0), 96, 54)]
    for i, (e0, a0, w, h) in enumerate(poses):
        ev = np.zeros((120, 3), dtype=np.float32)
        ev[:, 0] = rng.integers(0, 3, 120)
        ev[:, 1:] = (rng.normal(size=(120, 2)) * 80.0).astype(np.float32)
        ev[5] = (0, 0.0, 5000.0)   # orbit past the pole: the clamp of misc.hpp:172 keeps the old elevation
        ev[6] = (1, 0.0, 1e6)      # zoom through the look-at point: the 0.01 floor of :186
        out = ob.ref_camera_run(e0, a0, w, h, fovy, [(int(b), float(dx), float(dy)) for b, dx, dy in ev])
        cam[f"pose{i}_start"] = np.array(list(e0) + list(a0) + [w, h], dtype=np.float32)
        cam[f"pose{i}_events"] = ev
        cam[f"pose{i}_out"] = out
    cam["fovy"] = np.array([fovy], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "ref_camera.npz"), **cam)

    # OBJ ingest: the reference's tinyobj path on two shipped scenes + the scene files themselves
    import shutil
    import subprocess
    import tempfile

    os.makedirs(os.path.join(HERE, "assets"), exist_ok=True)
    tiny = {}
    for name, count in (("cornellbox1", 36), ("blocks_ao", 3034)):
        for ext in ("obj", "mtl"):
            shutil.copyfile(os.path.join(REF, "assets", f"{name}.{ext}"), os.path.join(HERE, "assets", f"{name}.{ext}"))
        with tempfile.TemporaryDirectory() as d:
            outp = os.path.join(d, "t.tris")
            subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "ref_tinyobj"), os.path.join(REF, "assets", name + ".obj"),
                                   os.path.join(REF, "assets") + "/", outp], stdout=subprocess.DEVNULL)
            t = np.fromfile(outp, dtype=ob.TRIANGLE)
        assert len(t) == count
        tiny[name] = t
    np.savez_compressed(os.path.join(HERE, "ref_tinyobj.npz"), **tiny)
    with open(os.path.join(HERE, "assets", "ATTRIBUTION"), "w") as f:
        f.write("cornellbox1.obj/.mtl and blocks_ao.obj/.mtl are data files of yumcyaWiz/CEDEC-2024-RT (assets/), MIT License,\n"
                "Copyright (c) 2024 Kenta Eto. They are test INPUTS here (OBJ reader parity on machines without the reference checkout).\n")
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
