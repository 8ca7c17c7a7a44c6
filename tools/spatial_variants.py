"""A/B of the unshadowed spatial_resampling pass on the benchmark frame (1920x1080): the default gather kernel
(explicit bound of 5 wavefronts per SIMD), the same with round 1's 32 KB dummy-LDS throttle on top (rt_tuning key 4),
and the LDS-staged variant (rt_tuning key 8 = 1). Per-pass HIP-event times over 60 frames.
  python tools/spatial_variants.py [variant]     variant: only run `gather` | `gather+lds32k` | `lds` (for rocprofv3)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cedec_2024_rt_amd import api, scenes  # noqa: E402
from cedec_2024_rt_amd.types import bench_options  # noqa: E402

W, H = 1920, 1080
tris = scenes.make_blocks_restir()
only = sys.argv[1] if len(sys.argv) > 1 else None
out = {}
CASES = [("gather", 0, 0, -1), ("lds", 0, 1, -1), ("coop", 0, 2, -1), ("pipe", 0, 3, -1)]
if only == "r04":
    only = None
    CASES = [("coop w6", 0, 2, 6), ("coop w5", 0, 2, 5)] + [("pipe w%d" % w, 0, 3, w) for w in (0, 6, 5, 4)]
elif not only:
    CASES = [("gather w%d" % w, 0, 0, w) for w in (0, 6, 5, 4)] + [("gather w5 +lds32k", 32768, 0, 5), ("gather w0 +lds32k", 32768, 0, 0)] + \
            [("lds w%d" % w, 0, 1, w) for w in (0, 6, 5, 4)] + [("lds w0 +lds32k", 32768, 1, 0)] + \
            [("coop w%d" % w, 0, 2, w) for w in (0, 6, 5, 4)] + [("pipe w%d" % w, 0, 3, w) for w in (0, 6, 5, 4)]
for name, key4, key8, key9 in CASES:
    if only and only != name:
        continue
    r = api.Renderer(W, H, exp=True)  # the A/B forms live in librestir_rt_exp.so
    r.set_scene(tris)
    r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
    r.set_options(bench_options())
    r.tuning(4, key4)
    r.tuning(8, key8)
    r.tuning(9, key9)
    for f in range(1, 6):
        r.frame(f)
    r.timing_enable(True)
    rows = []
    for f in range(6, 66):
        r.frame(f)
        t = r.timing()
        rows.append([t["spatial0"], t["spatial1"], t["spatial2"], t["frame"]])
    rows = np.array(rows)
    out[name] = dict(spatial_ms_per_pass=round(float(rows[:, :3].mean()), 4), spatial_ms_median=round(float(np.median(rows[:, :3])), 4),
                     frame_ms=round(float(np.median(rows[:, 3])), 4))
    r.close()
print(json.dumps(out))
