"""Dev tool: frame time vs the triangle pre-split factor (rt_bvh_config) on the bench frame."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options
W, H = 1920, 1080
tris = scenes.make_blocks_restir()
for f in [float(a) for a in sys.argv[1:]] or [0, 3, 4, 6, 8, 12, 16]:
    r = api.Renderer(W, H)
    r.bvh_config(f)
    r.set_scene(tris)
    r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
    r.set_options(bench_options())
    r.timing_enable(True)
    acc = None
    for fr in range(1, 24):
        r.frame(fr); t = r.timing()
        if fr > 3: acc = {k: acc[k] + v for k, v in t.items()} if acc else dict(t)
    print(f, r.bvh_info(), json.dumps({k: round(v / 20, 4) for k, v in acc.items() if k in ("raycast", "generate_candidate", "resolve", "frame")}), flush=True)
    r.close()
