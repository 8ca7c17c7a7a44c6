"""Dev tool: per-kernel times with use_shadowed_target_function on (README key 3)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options
if os.environ.get("RT_LIB"):
    api.LIB_PATH = os.path.join(ROOT, "cedec_2024_rt_amd", os.environ["RT_LIB"])
W, H = 1920, 1080
r = api.Renderer(W, H)
r.set_scene(scenes.make_blocks_restir())
r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
r.timing_enable(True)
for name, kw in (("unshadowed", {}), ("shadowed", dict(use_shadowed_target_function=1)), ("shadowed, no vis reuse", dict(use_shadowed_target_function=1, use_visibility_reuse=0))):
    o = bench_options()
    for k, v in kw.items(): o[k] = v
    r.set_options(o); r.clear()
    acc = None
    for fr in range(1, 10):
        r.frame(fr); t = r.timing()
        if fr > 3: acc = {k: acc[k] + v for k, v in t.items()} if acc else dict(t)
    rays, shaded = r.ray_count()
    ms = acc["frame"] / 6
    print("%-24s" % name, json.dumps({k: round(v / 6, 3) for k, v in acc.items()}), "rays %d  %.0f Mray/s" % (rays, rays / ms / 1e3), flush=True)
