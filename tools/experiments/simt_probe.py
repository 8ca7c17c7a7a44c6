"""Dev tool: SIMT efficiency of the shadow-ray traversal: per-lane steps vs passes the wavefront ran."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options
W, H = 1920, 1080
if os.environ.get("RT_LIB"):
    api.LIB_PATH = os.path.join(ROOT, "cedec_2024_rt_amd", os.environ["RT_LIB"])
tris = scenes.make_blocks_restir()
r = api.Renderer(W, H)
r.set_scene(tris)
r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
r.set_options(bench_options())
for fr in (1, 2, 3):
    final = r.frame(fr)
res = r.download(api.RT_BUF_RES_0 + final).reshape(H, W)
vis = r.download(api.RT_BUF_VISIBILITY).reshape(H, W)
li = scenes.light_indices(tris)
shaded = (vis["index"] >= 0) & ~np.isin(vis["index"], li)
ty, tx = H // 8, W // 32
tiled = lambda a: a.reshape(ty, 8, tx, 32).transpose(0, 2, 1, 3).reshape(ty * tx * 256)
rs = tiled(res); sh = tiled(shaded)
rays = np.zeros((len(rs), 8), np.float32)
rays[:, :3] = rs["origin_position"] + np.float32(0.001) * rs["origin_normal"]
rays[:, 3:6] = rs["hit_position"] - rs["origin_position"]
rays[:, 7] = np.where(sh, 0.99, -1.0).astype(np.float32)
for mode, name in ((4, "any-hit (shadow)"), (0, "closest")):
    r.trace_mode(mode)
    st = r.trace_stats(rays).astype(np.int64); wp = r.last_wave_passes.astype(np.int64)
    n_in, n_lf = st[:, 0], st[:, 1]
    w_in = wp[:, 0].reshape(-1, 64).max(1); w_lf = wp[:, 1].reshape(-1, 64).max(1)
    lanes = 64 * len(w_in)
    print("%s: per ray inner %.2f leaf %.2f | per wave inner passes %.1f leaf passes %.1f | lane use inner %.2f leaf %.2f" % (
        name, n_in.sum() / sh.sum(), n_lf.sum() / sh.sum(), w_in.mean(), w_lf.mean(),
        n_in.sum() / (64.0 * w_in.sum()), n_lf.sum() / (64.0 * max(w_lf.sum(), 1))), flush=True)
    hits = r.trace_closest(rays); print("   kernel %.3f ms" % r.trace_time())

