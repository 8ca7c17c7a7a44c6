"""How long is the slowest wavefront? Per-ray BVH steps and per-wavefront passes (inner + leaf) of the frame's
shadow rays and primary rays at 1920x1080: mean, percentiles and maximum. A launch that fits the GPU in one round
(a strip of a multi-GPU frame) lasts as long as its slowest wavefront, so the tail of this distribution — not
the mean — bounds strong scaling.
  python tools/wave_tail.py > profiles/rNN_wave_tail.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cedec_2024_rt_amd import api, scenes  # noqa: E402
from cedec_2024_rt_amd.types import bench_options  # noqa: E402

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
ty, tx = H // 8, W // 8
tiled = lambda a: a.reshape(ty, 8, tx, 8).transpose(0, 2, 1, 3).reshape(ty * tx * 64)  # 8x8 tiles = one wavefront each
rs, sh = tiled(res), tiled(shaded)
shadow = np.zeros((len(rs), 8), np.float32)
shadow[:, :3] = rs["origin_position"] + np.float32(0.001) * rs["origin_normal"]
shadow[:, 3:6] = rs["hit_position"] - rs["origin_position"]
shadow[:, 7] = np.where(sh, 0.99, -1.0).astype(np.float32)
rg = r.raygen()
# primary rays as raycast shoots them (pixel order of the 8x8 tiles)
ys, xs = np.mgrid[0:H, 0:W]
yi = H - 1 - ys
o = rg["origin"][0].astype(np.float64); rt_, up = rg["right"][0].astype(np.float64), rg["up"][0].astype(np.float64)
fwd = np.cross(up, rt_); fwd /= np.linalg.norm(fwd)
u, v = xs / W, yi / H
to = o + fwd + (-rt_ + 2 * rt_ * u[..., None]) + (up - 2 * up * v[..., None])
d = to - o
d /= np.linalg.norm(d, axis=2, keepdims=True)
prim = np.zeros((H * W, 8), np.float32)
prim[:, :3] = o.astype(np.float32)
prim[:, 3:6] = tiled(d.astype(np.float32).reshape(H, W, 3)[..., 0]).reshape(-1, 1) * 0  # placeholder, filled below
for k in range(3):
    prim[:, 3 + k] = tiled(d[..., k].astype(np.float32))
prim[:, 7] = 3.0e38


def report(name, mode, rays, live):
    r.trace_mode(mode)
    st = r.trace_stats(rays).astype(np.int64)
    wp = r.last_wave_passes.astype(np.int64)
    steps = (st[:, 0] + st[:, 1])[live]
    wave = (wp[:, 0] + wp[:, 1]).reshape(-1, 64).max(1)
    wave = wave[wave > 0]
    q = lambda a: "mean %.1f p50 %d p90 %d p99 %d p99.9 %d max %d" % (a.mean(), *np.percentile(a, [50, 90, 99, 99.9]).astype(int), a.max())
    print(f"{name}: per-ray steps  {q(steps)}")
    print(f"{name}: per-wave passes {q(wave)}   (slowest / mean = {wave.max() / wave.mean():.1f})", flush=True)


report("shadow rays (any hit)", 4, shadow, sh)
report("primary rays (closest)", 0, prim, np.ones(len(prim), bool))

# the work-sharing walk (occluded_ws): passes per wavefront and steals
r.trace_mode(5)
st = r.trace_stats(shadow)  # masked to 16 bits by the binding: use the raw call
import ctypes as C
raw = np.zeros((len(shadow), 2), dtype=np.uint32)
rr = np.ascontiguousarray(shadow, dtype=np.float32)
r._ck(r.L.rt_trace_stats(r.h, rr.ctypes.data_as(C.c_void_p), len(rr), raw.ctypes.data_as(C.c_void_p)))
wave = (raw[:, 0] & 0x7fff).reshape(-1, 64).max(1).astype(np.int64)
leafp = ((raw[:, 0] >> 15) & 0xff).reshape(-1, 64).max(1).astype(np.int64)
tests = ((raw[:, 0] >> 23) & 0xff).astype(np.int64)
steals = (raw[:, 1] & 0xffff).astype(np.int64)
steps = (raw[:, 1] >> 16).astype(np.int64)
keep = wave > 0
print("work-sharing shadow rays: of %.1f passes per wavefront %.1f are leaf passes (triangle tests per lane %.2f, i.e. %.0f %% of a leaf pass's lanes busy) and %.1f record visits (%.0f %% of their lanes busy)" % (
    wave[keep].mean(), leafp[keep].mean(), tests.reshape(-1, 64).sum(1)[keep].mean() / 64, 100 * tests.reshape(-1, 64).sum(1)[keep].mean() / 64 / max(leafp[keep].mean(), 1e-9),
    (wave - leafp)[keep].mean(), 100 * (raw[:, 1] >> 16).astype(np.int64).reshape(-1, 64).sum(1)[keep].mean() / 64 / (wave - leafp)[keep].mean()), flush=True)
wave = wave[keep]
print("work-sharing shadow rays: per-wave passes mean %.1f p50 %d p90 %d p99 %d max %d; steals per wavefront %.1f; inner records per lane mean %.1f (sum over a wave / 64 = %.1f)" % (
    wave.mean(), *np.percentile(wave, [50, 90, 99]).astype(int), wave.max(), steals.reshape(-1, 64).sum(1).mean(), steps[sh].mean(), steps.reshape(-1, 64).sum(1).mean() / 64), flush=True)

# device time of the same shadow rays from a list: persistent lane-refill queue (3), 256-thread workgroups one lane per
# ray (4), one-wavefront workgroups with the work-sharing walk (5) and without (6)
for mode, name in ((3, "persistent queue, lanes refilled with new rays"), (4, "one lane per ray, 256-thread workgroups"),
                   (6, "one lane per ray, one-wavefront workgroups"), (5, "work-sharing walk, one-wavefront workgroups")):
    r.trace_mode(mode)
    ts = []
    for _ in range(5):
        r.trace_closest(shadow)
        ts.append(r.trace_time())
    print("shadow rays from a list, mode %d (%s): %.3f ms" % (mode, name, min(ts)), flush=True)
