"""N frames of the benchmark workload with use_shadowed_target_function = 1 (README key 3), kernels back to
back on one stream: the program tools/profile_round.sh puts under rocprofv3 for the shadowed-mode kernel trace and counters.

  python tools/shadowed_frames.py [frames]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes  # noqa: E402
from cedec_2024_rt_amd.types import bench_options  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, H = 1920, 1080
r = api.Renderer(W, H)
r.set_scene(scenes.make_blocks_restir())
r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
r.set_options(bench_options(use_shadowed_target_function=1))
r.tuning(14, 0)  # per-kernel numbers: no overlap between frames
for f in range(1, n + 1):
    r.frame(f)
r.sync()
rays, shaded = r.ray_count()
print(f"{n} shadowed frames, {rays} reference rays per frame, {shaded} shaded pixels")
r.close()
