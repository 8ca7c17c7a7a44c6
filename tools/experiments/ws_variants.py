"""One A/B line per library build: whole-frame kernel times at 1080p (rt_frame, HIP events) and a 135-row strip's
generate / resolve stage times (host-timed, synchronised). RT_LIB_PATH selects the build."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options

tris = scenes.make_blocks_restir()
r = api.Renderer(1920, 1080)
r.set_scene(tris); r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT); r.set_options(bench_options())
r.timing_enable(True)
acc = None
for f in range(1, 45):
    r.frame(f, clear_first=(f == 1)); r.sync()
    if f > 4:
        t = r.timing()
        acc = t if acc is None else {k: acc[k] + t[k] for k in t}
out = {"full": {k: round(acc[k] / 40, 4) for k in ("generate_candidate", "resolve", "frame")}}
r.close()
a = (1080 - 135) // 2
r = api.Renderer(1920, 1080, rows=(a, a + 135), halo=90)
r.set_scene(tris); r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT); r.set_options(bench_options())
def stage(f, k, clear=False):
    r.sync(); t0 = time.perf_counter(); r.frame_stage(f, k, clear); r.sync()
    return (time.perf_counter() - t0) * 1e3
acc = [0.0] * 5
for f in range(1, 45):
    t = [stage(f, k, f == 1 and k == 0) for k in range(5)]
    if f > 4:
        acc = [x + y for x, y in zip(acc, t)]
out["strip135"] = dict(raycast_generate=round(acc[0] / 40, 4), resolve_tone=round(acc[4] / 40, 4))
print(os.environ.get("RT_LIB_PATH", "default"), json.dumps(out))
