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
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
