"""N frames of config #4 at 1080p (for rocprofv3 runs of experiments): python tools/experiments/run_frames.py [frames]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
r = api.Renderer(1920, 1080)
r.set_scene(scenes.make_blocks_restir()); r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT); r.set_options(bench_options())
for f in range(1, n + 1):
    r.frame(f)
r.sync()
