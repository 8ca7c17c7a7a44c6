"""Dev tool: frame time on other cameras / scenes (is the headline view representative?)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options
W, H = 1920, 1080
g = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
blocks = scenes.make_blocks_restir()
cases = [("blocks cam1", blocks, scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT),
         ("blocks cam2 (10_restir_di.cpp:192-193)", blocks, (-14.853887, 27.826096, -46.717506), (-3.813484, 11.610589, 4.625816)),
         ("blocks cam 07_pt (07_pt.cpp:139-140)", blocks, (5.983407, 13.970583, -28.553869), (-5.354514, 4.815835, -2.047728)),
         ("blocks top-down", blocks, (5.0, 70.0, 20.0), (5.0, 0.0, 21.0)),
         ("cornellbox2", g["cornellbox2"], scenes.CORNELLBOX_EYE, scenes.CORNELLBOX_LOOKAT),
         ("cornellbox1 default cam", g["cornellbox1"], scenes.DEFAULT_EYE, scenes.DEFAULT_LOOKAT)]
for name, tris, eye, at in cases:
    r = api.Renderer(W, H); r.set_scene(tris); r.lookat(eye, at); r.set_options(bench_options()); r.timing_enable(True)
    acc = None
    for fr in range(1, 14):
        r.frame(fr); t = r.timing()
        if fr > 3: acc = {k: acc[k] + v for k, v in t.items()} if acc else dict(t)
    rays, shaded = r.ray_count()
    ms = acc["frame"] / 10
    print(json.dumps(dict(case=name, shaded_frac=round(shaded / (W * H), 3), ms=round(ms, 3), mray_s=round(rays / ms / 1e3),
                          kernels={k: round(v / 10, 3) for k, v in acc.items() if k in ("raycast", "generate_candidate", "spatial0", "resolve")})), flush=True)
    r.close()
