"""Dev tool: path tracers as one launch per frame vs wavefront (launch per bounce, ballot compaction)."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import default_options
g = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
blocks = scenes.make_blocks_restir()
cases = [("07_pt cornellbox2 512^2", 7, g["cornellbox2"], 512, 512, scenes.CORNELLBOX_EYE, scenes.CORNELLBOX_LOOKAT),
         ("07_pt blocks 1080p", 7, blocks, 1920, 1080, scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT),
         ("07_pt blocks 1080p cam2", 7, blocks, 1920, 1080, (-14.853887, 27.826096, -46.717506), (-3.813484, 11.610589, 4.625816)),
         ("09_ris blocks 720p", 9, blocks, 1280, 720, scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)]
for name, ex, tris, W, H, eye, at in cases:
    r = api.Renderer(W, H); r.set_scene(tris); r.lookat(eye, at); r.set_options(default_options())
    res = {}
    for mode in (0, 1):
        r.tuning(6, mode)
        for f in range(1, 4): r.path_trace(ex, f)
        r.sync(); t0 = time.perf_counter()
        for f in range(4, 12): r.path_trace(ex, f)
        r.sync(); res[mode] = (time.perf_counter() - t0) / 8 * 1e3
    rays = r.path_trace_rays()
    print(json.dumps(dict(case=name, megakernel_ms=round(res[0], 3), wavefront_ms=round(res[1], 3), rays=rays, rays_per_px=round(rays / (W * H), 2))), flush=True)
    r.close()
