"""GPU, experiments library. rt_raycast alone on row strips: the whole-frame walk, the work-sharing walk (strips' default), the
HALF-DENSITY form (rt_tuning key 24: 32 rays + 32 helper lanes per wavefront, twice the wavefronts) and (r06) FOUR LANES PER RAY
(rt_tuning key 16 = 2: 16 rays per wavefront, each lane one child box, four times the wavefronts) — time per launch, and the
Visibility buffer of each form against the default's, byte for byte.

  python tools/half_raycast.py [WxH:rows ...]      default: 1920x1080:135 1920x1080:270 1920x1080:1080 3840x2160:270
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes  # noqa: E402
from cedec_2024_rt_amd.types import bench_options  # noqa: E402

FORMS = (("plain walk", {16: 0, 24: 0}), ("work-sharing walk", {16: 1, 24: 0}), ("half density + helpers", {16: 1, 24: 1}),
         ("four lanes per ray (r06, key 16 = 2)", {16: 2, 24: 0}))
if os.environ.get("RAYCAST_FORMS"):  # e.g. RAYCAST_FORMS=0,1,3 with RT_LIB_PATH=<a product-library variant> (it refuses the experiments' key 24)
    FORMS = tuple(FORMS[int(i)] for i in os.environ["RAYCAST_FORMS"].split(","))
tris = scenes.make_blocks_restir()
for case in sys.argv[1:] or ["1920x1080:135", "1920x1080:270", "1920x1080:1080", "3840x2160:270"]:
    size, rows = case.split(":")
    W, H = (int(v) for v in size.split("x"))
    rows = int(rows)
    a = (H - rows) // 2
    r = api.Renderer(W, H, rows=(a, a + rows), halo=0, exp=True)
    r.set_scene(tris)
    r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
    r.set_options(bench_options())
    ref = None
    line = []
    for name, keys in FORMS:
        for k, v in keys.items():
            if k != 24 or v != 0 or not os.environ.get("RAYCAST_FORMS"):
                r.tuning(k, v)
        r.clear()
        for _ in range(20):
            r.raycast()
        r.sync()
        t0 = time.perf_counter()
        n = 400
        for _ in range(n):
            r.raycast()
        r.sync()
        us = (time.perf_counter() - t0) / n * 1e6
        vis = r.download(api.RT_BUF_VISIBILITY).tobytes()
        if ref is None:
            ref = vis
        same = vis == ref
        line.append("%s %.1f us%s" % (name, us, "" if same else " (DIFFERS)"))
        if not same:
            print("MISMATCH", case, name, flush=True)
    print("%s, %d rows (%d wavefronts of 64 rays): %s" % (size, rows, W * rows // 64, "; ".join(line)), flush=True)
    r.close()
