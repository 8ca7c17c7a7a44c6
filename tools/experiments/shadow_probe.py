"""Dev tool: traversal work of the resolve-pass shadow rays (closest-hit stats as a proxy)."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options
W, H = 1920, 1080
tris = scenes.make_blocks_restir()
r = api.Renderer(W, H)
if len(sys.argv) > 1: r.bvh_config(float(sys.argv[1]))
r.set_scene(tris)
r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
r.trace_mode(int(os.environ.get("TRACE_MODE", "0")))
r.set_options(bench_options())
for fr in (1, 2, 3):
    final = r.frame(fr)
res = r.download(api.RT_BUF_RES_0 + final)
vis = r.download(api.RT_BUF_VISIBILITY)
li = scenes.light_indices(tris)
shaded = (vis["index"] >= 0) & ~np.isin(vis["index"], li)
rs = res[shaded]
rays = np.zeros((len(rs), 8), np.float32)
rays[:, :3] = rs["origin_position"] + np.float32(0.001) * rs["origin_normal"]
rays[:, 3:6] = rs["hit_position"] - rs["origin_position"]
rays[:, 7] = 0.99
print("shadow rays", len(rays), "visible frac", float(rs["visibility"].mean()))
st = r.trace_stats(rays)
n = st[:, 0].astype(np.int64); t = st[:, 1].astype(np.int64)
print("nodes mean %.1f p50 %d p99 %d max %d | tris mean %.2f p99 %d max %d" % (n.mean(), np.percentile(n, 50), np.percentile(n, 99), n.max(), t.mean(), np.percentile(t, 99), t.max()))
wm = n[: len(n) // 64 * 64].reshape(-1, 64)
print("per-wave: max mean %.1f, mean-of-mean %.1f, utilisation %.2f" % (wm.max(1).mean(), wm.mean(1).mean(), wm.mean() / wm.max(1).mean()))
t0 = time.time(); r.trace_closest(rays); print("trace_closest wall %.2f ms (incl. copies)" % ((time.time() - t0) * 1e3))
