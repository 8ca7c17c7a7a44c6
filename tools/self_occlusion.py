"""Where do the shadow rays of the benchmark frame end? (CPU, oracle only: a measurement that motivated the
self-occlusion pre-test of csrc/bvh.h, self_occluded.) For the visibility-reuse rays of generate_candidate ("temporal":
surface point -> the candidate that survived) and the rays of resolve ("r1": surface point -> the final sample) of frame
3 of the 480x270 benchmark workload: share occluded, share whose CLOSEST hit is the triangle the ray starts from, share
heading below their own surface (n . d < 0), and the quantiles of the closest occluder's t.

  python tools/self_occlusion.py > profiles/r03_self_occlusion.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import scenes  # noqa: E402
from oracle import binding as ob  # noqa: E402

ob.set_math_mode(ob.MATH_PORTABLE)
tris = scenes.make_blocks_restir()
W, H = 480, 270
sc = ob.Scene(tris, use_bvh=True)
rg = ob.raygen_lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT, (0, 1, 0), np.float32(np.pi) / np.float32(4), W, H)
eye = np.asarray(scenes.BLOCKS_RESTIR_EYE, np.float32)
for sh in (0, 1):
    opt = ob.bench_options(use_shadowed_target_function=sh)
    st = ob.new_state(W, H)
    for f in (1, 2, 3):
        sc.frame(W, H, f, rg, eye, opt, st, None)
    vis = st["vis"]["index"].reshape(H, W)
    shaded = (vis >= 0) & ~np.isin(vis, sc.lights)
    for name in ("temporal", "r1"):
        r = st[name].reshape(H, W)
        d = r["hit_position"] - r["origin_position"]
        nd = (d * r["origin_normal"]).sum(-1)
        rays = np.zeros((H * W, 8), np.float32)
        rays[:, :3] = (r["origin_position"] + np.float32(0.001) * r["origin_normal"]).reshape(-1, 3)
        rays[:, 3:6] = d.reshape(-1, 3)
        rays[:, 7] = 0.99
        hits = sc.trace_closest(rays)
        idx = hits[:, 3].view(np.int32).reshape(H, W)
        t = hits[:, 0].reshape(H, W)
        occ = (idx >= 0) & shaded
        own = (idx == vis) & shaded
        below = (nd < 0) & shaded
        n = shaded.sum()
        q = np.quantile(t[occ], [0.1, 0.25, 0.5, 0.75, 0.9])
        print(f"use_shadowed_target_function={sh} {name:8s}: occluded {occ.sum() / n:.3f}; closest hit = own triangle {own.sum() / n:.3f}; "
              f"heading below the own surface {below.sum() / n:.3f} (of those occluded {(below & occ).sum() / below.sum():.3f}, by the own triangle "
              f"{(below & own).sum() / below.sum():.3f}); closest occluder t quantiles 10/25/50/75/90 %: " + " ".join(f"{v:.2e}" for v in q))
