"""GPU, experiments library. What would longest-first dispatch gain? One frame's per-wavefront durations of the one-launch stage 0
(kernel 1) and of resolve (kernel 3) are measured with rt_exp_wave_clock in the default order; rt_exp_tile_perm then re-orders the
workgroups of each XCD — every tile longest first, or groups of G neighbouring workgroups by their slowest member (keeps neighbours
together) — and un-pipelined frames are timed (wall clock over 60 frames, alternating with the default order). Results never change.

  python tools/tile_lpt.py [WxH]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes  # noqa: E402
from cedec_2024_rt_amd.types import bench_options  # noqa: E402

W, H = (int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (1920, 1080)
r = api.Renderer(W, H, exp=True)
r.set_scene(scenes.make_blocks_restir())
r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
r.set_options(bench_options())
r.tuning(14, 0)
r.tuning(17, 0)
clock = r.L.rt_exp_wave_clock
clock.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
clock.restype = C.c_int
perm_fn = r.L.rt_exp_tile_perm
perm_fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
perm_fn.restype = C.c_int
tx, ty = (W + 7) // 8, (H + 7) // 8
grid = max((tx * ty + 127) // 128 * 128, 8 * ((ty + 7) // 8) * tx)
frame = [0]


def frames(n):
    for _ in range(n):
        frame[0] += 1
        r.frame(frame[0])


def timed(n=60):
    frames(5)
    r.sync()
    t0 = time.perf_counter()
    frames(n)
    r.sync()
    return (time.perf_counter() - t0) / n * 1e3


def durations(kernel):
    assert clock(r.h, kernel, 0, None, 0) == 0
    frames(1)
    r.sync()
    buf = np.zeros(2 * grid, np.uint64)
    assert clock(r.h, kernel, 0, buf.ctypes.data, 2 * grid) == 0
    assert clock(r.h, -1, 0, None, 0) == 0
    t0 = (buf[0::2] & np.uint64(0xFFFFFFFFFF)).astype(np.int64)
    t1 = (buf[1::2] & np.uint64(0xFFFFFFFFFF)).astype(np.int64)
    return np.where(buf[0::2] > 0, t1 - t0, 0) / 100.0  # us; 0 for workgroups without a tile


def order(d, group):
    """perm[new b] = old b, per XCD: groups of `group` consecutive slots, the group with the slowest member first"""
    perm = np.arange(grid, dtype=np.uint32)
    for x in range(8):
        old = np.arange(x, grid, 8)
        n = len(old) // group * group
        g = d[old[:n]].reshape(-1, group).max(axis=1)
        o = np.argsort(-g, kind="stable")
        new = np.concatenate([old[:n].reshape(-1, group)[o].ravel(), old[n:]])
        perm[old] = new
    return perm


frames(12)
r.sync()
print("%dx%d, un-pipelined frames, ms per frame (wall clock over 60 frames)" % (W, H), flush=True)
d = {k: durations(k) for k in (1, 3)}
for k in (1, 3):
    dd = d[k][d[k] > 0]
    print("kernel %d: %d wavefronts, duration us mean %.1f p99 %.1f max %.1f" % (k, len(dd), dd.mean(), np.percentile(dd, 99), dd.max()), flush=True)
for label, group in (("default order", 0), ("every tile longest first", 1), ("groups of 4", 4), ("groups of 16", 16), ("groups of 64", 64), ("default order", 0)):
    for k in (1, 3):
        if group:
            p = order(d[k], group)
            assert perm_fn(r.h, k, p.ctypes.data, grid) == 0, r.L.rt_last_error(r.h)
        else:
            assert perm_fn(r.h, k, None, 0) == 0
    print("%-26s %s" % (label, "  ".join("%.4f" % timed() for _ in range(3))), flush=True)
for k in (1, 3):
    perm_fn(r.h, k, None, 0)
# each kernel alone, groups of 16
for k, name in ((1, "stage 0 only"), (3, "resolve only")):
    p = order(d[k], 16)
    assert perm_fn(r.h, k, p.ctypes.data, grid) == 0
    print("groups of 16, %-13s %s" % (name, "  ".join("%.4f" % timed() for _ in range(3))), flush=True)
    perm_fn(r.h, k, None, 0)
print("%-26s %s" % ("default order", "  ".join("%.4f" % timed() for _ in range(3))), flush=True)
r.close()
