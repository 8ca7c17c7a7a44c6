"""Dev tool: how much would re-grouping shadow rays inside a workgroup buy? Takes the resolve-pass
shadow rays of a bench frame in kernel order (32x8 tiles = one 256-thread workgroup each) and
times the any-hit traversal (trace mode 4) for several lane assignments."""
import sys, os, json
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
res = r.download(api.RT_BUF_RES_0 + final).reshape(H, W)
vis = r.download(api.RT_BUF_VISIBILITY).reshape(H, W)
li = scenes.light_indices(tris)
shaded = (vis["index"] >= 0) & ~np.isin(vis["index"], li)
# kernel order: tiles of 32x8, row-major tiles (XCD banding ignored)
ty, tx = H // 8, W // 32
def tiled(a):
    return a.reshape(ty, 8, tx, 32).transpose(0, 2, 1, 3).reshape(ty * tx, 256)
rs = tiled(res); sh = tiled(shaded)
rays = np.zeros((ty * tx, 256, 8), np.float32)
rays[..., :3] = rs["origin_position"] + np.float32(0.001) * rs["origin_normal"]
rays[..., 3:6] = rs["hit_position"] - rs["origin_position"]
rays[..., 7] = np.where(sh, 0.99, -1.0).astype(np.float32)  # tmax < tmin: no work for sky/emissive lanes
def morton(p, bits=10):
    lo = p.reshape(-1, 3).min(0); ext = p.reshape(-1, 3).max(0) - lo
    q = np.clip(((p - lo) / ext * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(q.shape[:-1], np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[..., a] >> b) & 1) << (3 * b + a)
    return code
r.trace_mode(4)
def run(name, order):
    rr = np.take_along_axis(rays, order[..., None], axis=1).reshape(-1, 8)
    best = 1e9
    for _ in range(3):
        hits = r.trace_closest(rr); best = min(best, r.trace_time())
    print("%-28s %.3f ms" % (name, best), flush=True)
ident = np.broadcast_to(np.arange(256), (ty * tx, 256)).copy()
run("tile order", ident)
key_light = morton(rs["hit_position"]) + np.where(sh, 0, 1 << 40)
run("by light position", np.argsort(key_light, axis=1, kind="stable"))
d = rays[..., 3:6]; octant = (d[..., 0] < 0) * 1 + (d[..., 1] < 0) * 2 + (d[..., 2] < 0) * 4
run("by direction octant", np.argsort(octant + np.where(sh, 0, 8), axis=1, kind="stable"))
dn = d / np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-20)
run("by direction morton", np.argsort(morton(dn) + np.where(sh, 0, 1 << 40), axis=1, kind="stable"))
run("compact only (shaded first)", np.argsort(np.where(sh, 0, 1), axis=1, kind="stable"))
rng = np.random.default_rng(1)
run("random within tile", np.argsort(rng.random((ty * tx, 256)), axis=1))
# upper bound for length-based regrouping: sort each tile's rays by their ACTUAL any-hit step count
st = r.trace_stats(rays.reshape(-1, 8)).astype(np.int64).reshape(ty * tx, 256, 2)
steps = st[..., 0] + st[..., 1]
run("by actual step count", np.argsort(steps, axis=1, kind="stable"))
# a realistic predictor: the step count of the same pixel's ray one frame earlier
res_prev = res
final = r.frame(4)
res4 = r.download(api.RT_BUF_RES_0 + final).reshape(H, W)
rs4 = tiled(res4)
rays4 = np.zeros_like(rays)
rays4[..., :3] = rs4["origin_position"] + np.float32(0.001) * rs4["origin_normal"]
rays4[..., 3:6] = rs4["hit_position"] - rs4["origin_position"]
rays4[..., 7] = np.where(sh, 0.99, -1.0).astype(np.float32)
rays_keep = rays
rays = rays4
r.trace_mode(4)
run("frame 4, tile order", ident)
run("frame 4, by frame-3 step count", np.argsort(steps, axis=1, kind="stable"))
st4 = r.trace_stats(rays.reshape(-1, 8)).astype(np.int64).reshape(ty * tx, 256, 2)
run("frame 4, by own step count", np.argsort(st4[..., 0] + st4[..., 1], axis=1, kind="stable"))
print("corr(frame3 steps, frame4 steps) = %.3f" % np.corrcoef(steps[sh].ravel(), (st4[..., 0] + st4[..., 1])[sh].ravel())[0, 1])
