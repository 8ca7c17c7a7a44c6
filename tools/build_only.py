"""rt_scene_set of the benchmark scene with one builder, a few times (the program for a kernel trace of the BVH build):
  rocprofv3 --kernel-trace --stats -d DIR -- python3 tools/build_only.py 3"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes  # noqa: E402

builder = int(sys.argv[1]) if len(sys.argv) > 1 else 3
r = api.Renderer(64, 64, exp=True)  # builders 0-2: librestir_rt_exp.so
r.tuning(5, builder)
tris = scenes.make_blocks_restir()
for _ in range(5):
    r.set_scene(tris)
    print(f"builder {builder}: rt_scene_set {r.build_ms():.2f} ms, {r.bvh_info()}")
r.close()
