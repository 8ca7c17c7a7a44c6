"""Dev tool: spatial kernel time vs tile order / occupancy limit."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options
W, H = 1920, 1080
r = api.Renderer(W, H)
r.set_scene(scenes.make_blocks_restir())
r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
r.set_options(bench_options())
r.timing_enable(True)
for mode in (0, 1):
    for lds in (0, 20480, 32768, 40960, 53248, 81920):
        r.tuning(2, mode); r.tuning(4, lds)
        acc = None
        for fr in range(1, 10):
            r.frame(fr); t = r.timing()
            if fr > 3: acc = {k: acc[k] + v for k, v in t.items()} if acc else dict(t)
        print(json.dumps(dict(mode=mode, lds=lds, wg_per_cu=(160 * 1024 // lds if lds else 8), **{k: round(v / 6, 4) for k, v in acc.items() if k in ("raycast", "generate_candidate", "spatial0", "spatial1", "spatial2", "resolve", "frame")})), flush=True)
