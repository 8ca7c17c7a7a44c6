"""The four BVH builders on the benchmark scene (211 916 triangles): rt_scene_set wall time (upload + tables +
build, synchronised), tree statistics, and the frame time each tree gives (1920x1080, bench options, HIP events).
  python tools/bvh_builders.py"""
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
out = {}
names = {0: "device LBVH (Karras) + host pre-split/collapse", 1: "host binned SAH", 2: "device: pre-split, PLOC, host SAH sweep over the last 8192 clusters, collapse",
         3: "device: pre-split, top-down binned SAH (32 bins), collapse"}
CASES = [(1, None), (3, None), (0, None), (2, 16)] if not os.environ.get("BVH_PLOC_SWEEP") else [(2, rad) for rad in (8, 16, 32, 64, 128)]
for builder, rad in CASES:
    r = api.Renderer(W, H, exp=True)  # the A/B forms live in librestir_rt_exp.so
    r.tuning(5, builder)
    if rad is not None:
        r.tuning(10, rad)
    builds = []
    for _ in range(4):
        r.set_scene(tris)
        builds.append(r.build_ms())
    r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
    r.set_options(bench_options())
    for f in range(1, 6):
        r.frame(f)
    r.timing_enable(True)
    rows = []
    for f in range(6, 56):
        r.frame(f)
        t = r.timing()
        rows.append([t["raycast"], t["generate_candidate"], t["resolve"], t["frame"]])
    rows = np.array(rows)
    info = r.bvh_info()
    key = names[builder] + (f", PLOC radius {rad}" if rad is not None else "")
    out[key] = dict(build_ms_first=round(builds[0], 2), build_ms=round(float(np.median(builds[1:])), 2), references=info["references"],
                               wide_records=info["wide_records"], wide_height=info["wide_height"], binary_height=r.scene_info()["bvh_height"],
                               raycast_ms=round(float(np.median(rows[:, 0])), 4), generate_ms=round(float(np.median(rows[:, 1])), 4),
                               resolve_ms=round(float(np.median(rows[:, 2])), 4), frame_ms=round(float(np.median(rows[:, 3])), 4))
    print(json.dumps({key: out[key]}), flush=True)
    r.close()
