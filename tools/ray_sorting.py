"""Would sorting the shadow rays of a pixel block by their target help the walk? The resolve rays of config #4 (1080p,
frame 3) as a list in 8x8-tile order, walked by the list kernels (rt_trace_mode 6 = one lane per ray, 5 = work-sharing),
as they are and re-ordered inside blocks of 256 / 1024 consecutive rays (4 / 16 tiles) by target."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options

W, H = 1920, 1080
tris = scenes.make_blocks_restir()
r = api.Renderer(W, H)
r.set_scene(tris); r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT); r.set_options(bench_options())
for fr in (1, 2, 3):
    final = r.frame(fr)
res = r.download(api.RT_BUF_RES_0 + final).reshape(H, W)
vis = r.download(api.RT_BUF_VISIBILITY).reshape(H, W)
li = scenes.light_indices(tris)
shaded = (vis["index"] >= 0) & ~np.isin(vis["index"], li)
ty, tx = H // 8, W // 8
tiled = lambda a: a.reshape(ty, 8, tx, 8).transpose(0, 2, 1, 3).reshape(ty * tx * 64)
rs, sh = tiled(res), tiled(shaded)
rays = np.zeros((len(rs), 8), np.float32)
rays[:, :3] = rs["origin_position"] + np.float32(0.001) * rs["origin_normal"]
rays[:, 3:6] = rs["hit_position"] - rs["origin_position"]
rays[:, 7] = np.where(sh, 0.99, -1.0).astype(np.float32)
hp = rs["hit_position"].astype(np.float64)
key_target = (np.floor(hp[:, 0] * 4) * 73856093 + np.floor(hp[:, 1] * 4) * 19349663 + np.floor(hp[:, 2] * 4) * 83492791).astype(np.int64)
d = rays[:, 3:6] / np.maximum(np.linalg.norm(rays[:, 3:6], axis=1, keepdims=True), 1e-20)
key_dir = (np.floor((d[:, 0] + 1) * 4) * 64 + np.floor((d[:, 1] + 1) * 4) * 8 + np.floor((d[:, 2] + 1) * 4)).astype(np.int64)


def timed(name, lst):
    out = []
    for mode in (6, 5):
        r.trace_mode(mode)
        ts = []
        for _ in range(4):
            r.trace_closest(lst)
            ts.append(r.trace_time())
        out.append(min(ts))
    print("%-52s one lane per ray %.3f ms, work-sharing %.3f ms" % (name, out[0], out[1]), flush=True)


uniq = np.array([len(np.unique(key_target[i:i + 64][sh[i:i + 64]])) for i in range(0, len(rs), 64 * 50)])
print("distinct targets per 8x8 tile (sampled): mean %.1f" % uniq.mean())
timed("tile order (as the frame kernels walk them)", rays)
for blk in (256, 1024):
    for kname, key in (("target cell", key_target), ("direction octant bins", key_dir)):
        order = np.arange(len(rays)).reshape(-1, blk)
        k = np.where(sh, key, np.int64(1) << 62).reshape(-1, blk)  # lanes without a ray last
        idx = np.take_along_axis(order, np.argsort(k, axis=1, kind="stable"), axis=1).reshape(-1)
        timed("blocks of %d rays sorted by %s" % (blk, kname), rays[idx])
