import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import default_options
if os.environ.get("RT_LIB"): api.LIB_PATH = os.path.join(os.path.dirname(api.__file__), os.environ["RT_LIB"])
r = api.Renderer(1920, 1080); r.set_scene(scenes.make_blocks_restir()); r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT); r.set_options(default_options())
out = []
for ex in (8, 9):
    r.path_trace(ex, 1); r.sync(); t0 = time.perf_counter()
    for k in range(2, 12): r.path_trace(ex, k)
    r.sync(); out.append("ex%d %.3f ms" % (ex, (time.perf_counter() - t0) / 10 * 1e3))
print(os.environ.get("RT_LIB"), out)
