"""rt_frame (config #4, whole frame on one GPU) with the work-sharing shadow-ray walk off / on / auto (rt_tuning key 13):
HIP-event time per kernel."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options

tris = scenes.make_blocks_restir()
out = {}
for W, H in ((1920, 1080), (3840, 2160)):
    r = api.Renderer(W, H)
    r.set_scene(tris); r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT); r.set_options(bench_options())
    r.timing_enable(True)
    for ws in (0, 1, 0, 1):
        r.tuning(13, ws)
        acc = [0.0] * 9
        for f in range(1, 45):
            r.frame(f, clear_first=(f == 1)); r.sync()
            if f > 4:
                acc = [x + y for x, y in zip(acc, r.timing().values())]
        n = 40
        out.setdefault("%dx%d ws%d" % (W, H, ws), []).append(dict(zip(("clear", "raycast", "generate", "sp0", "sp1", "sp2", "resolve", "tone", "frame"), [round(x / n, 4) for x in acc])))
    r.close()
print(json.dumps(out, indent=1))
