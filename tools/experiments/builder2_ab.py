"""Builder 2 (all-device PLOC) against the default host SAH: build time and kernel times of config #4 at 1080p."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options
tris = scenes.make_blocks_restir()
for builder in (1, 2):
    r = api.Renderer(1920, 1080)
    r.tuning(5, builder)
    r.set_scene(tris); r.set_scene(tris)
    r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT); r.set_options(bench_options())
    r.timing_enable(True)
    acc = None
    for f in range(1, 35):
        r.frame(f); r.sync()
        if f > 4:
            t = r.timing(); acc = t if acc is None else {k: acc[k] + t[k] for k in t}
    print(os.environ.get("RT_LIB_PATH", "default")[-16:], "builder", builder, "build_ms %.1f" % r.build_ms(), {k: round(acc[k] / 30, 4) for k in ("raycast", "generate_candidate", "resolve", "frame")}, r.bvh_info())
    r.close()
