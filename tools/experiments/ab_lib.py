"""Dev tool: A/B two builds of librestir_rt (env RT_LIB selects the .so) on the bench frame."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes
if os.environ.get("RT_LIB"):
    api.LIB_PATH = os.path.join(ROOT, "cedec_2024_rt_amd", os.environ["RT_LIB"])
from cedec_2024_rt_amd.types import bench_options
W, H = 1920, 1080
r = api.Renderer(W, H)
r.set_scene(scenes.make_blocks_restir())
r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
r.set_options(bench_options())
for kv in os.environ.get("RT_TUNE", "").split(","):
    if kv: r.tuning(*[int(v) for v in kv.split("=")])
r.timing_enable(True)
acc = None
for fr in range(1, 24):
    r.frame(fr); t = r.timing()
    if fr > 3: acc = {k: acc[k] + v for k, v in t.items()} if acc else dict(t)
print(os.environ.get("RT_LIB", "default"), json.dumps({k: round(v / 20, 4) for k, v in acc.items()}))
