"""Dev tool: path tracers (07_pt, 08_nee) vs the triangle pre-split factor, on cornellbox2 and the blocks stand-in."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import default_options
g = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
for name, tris, W, H, eye, at in (("cornellbox2", g["cornellbox2"], 512, 512, scenes.CORNELLBOX_EYE, scenes.CORNELLBOX_LOOKAT), ("blocks", scenes.make_blocks_restir(), 1920, 1080, scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)):
    for f in (8.0, 10.0, 12.0):
        r = api.Renderer(W, H); r.bvh_config(f); r.set_scene(tris); r.lookat(eye, at); r.set_options(default_options(accumulate=1)); r.clear()
        for ex in (7, 8):
            r.path_trace(ex, 1); r.sync(); t0 = time.perf_counter()
            for k in range(2, 12): r.path_trace(ex, k)
            r.sync(); print(name, "split", f, "example", ex, round((time.perf_counter() - t0) / 10 * 1e3, 4), "ms", flush=True)
        r.close()
