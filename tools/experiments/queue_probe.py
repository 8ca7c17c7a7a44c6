"""Dev tool: shadow-ray batch traced by (0) one-thread-per-ray wide kernel vs (2/3) persistent queue."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options
W, H = 1920, 1080
tris = scenes.make_blocks_restir()
r = api.Renderer(W, H)
r.set_scene(tris)
r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
r.set_options(bench_options())
for fr in (1, 2, 3):
    final = r.frame(fr)
res = r.download(api.RT_BUF_RES_0 + final)
vis = r.download(api.RT_BUF_VISIBILITY)
shaded = (vis["index"] >= 0) & ~np.isin(vis["index"], scenes.light_indices(tris))
rs = res[shaded]
rays = np.zeros((len(rs), 8), np.float32)
rays[:, :3] = rs["origin_position"] + np.float32(0.001) * rs["origin_normal"]
rays[:, 3:6] = rs["hit_position"] - rs["origin_position"]
rays[:, 7] = 0.99
ref = None
for mode in (0, 2, 3):
    r.trace_mode(mode)
    for k in range(2):
        h = r.trace_closest(rays)
    if mode == 0: ref = h
    if mode == 2: print("queue closest == plain:", np.array_equal(h.view(np.uint32), ref.view(np.uint32)))
    if mode == 3: print("queue any-hit agrees on occlusion:", np.array_equal(h[:, 0] > 0, ref[:, 3].view(np.int32) >= 0))
