"""GPU. Frame and tracing-kernel times against the BVH pre-split factor (rt_bvh_config: large triangles are cut into fragments no
longer than factor x the median triangle extent before the SAH build; results never depend on it)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes  # noqa: E402
from cedec_2024_rt_amd.types import bench_options  # noqa: E402

W, H = 1920, 1080
tris = scenes.make_blocks_restir()
names = ("clear", "raycast", "generate_candidate", "spatial0", "spatial1", "spatial2", "resolve", "tone_mapping", "frame")
for sf in [float(v) for v in (sys.argv[1:] or ["0", "3", "5", "7", "10", "14", "20", "40"])]:
    r = api.Renderer(W, H)
    r.bvh_config(sf)
    r.set_scene(tris)
    r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
    r.set_options(bench_options())
    r.timing_enable(True)
    rows = []
    for f in range(1, 41):
        r.frame(f)
        t = r.timing()
        if f > 8:
            rows.append([t[k] for k in names])
    m = np.array(rows).mean(axis=0)
    print("split factor %5.1f: build %.1f ms, %s; raycast %.4f generate %.4f resolve %.4f frame %.4f" % (sf, r.build_ms(), r.bvh_info(), m[1], m[2], m[6], m[8]), flush=True)
    r.close()
