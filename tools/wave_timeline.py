"""GPU, experiments library. When does every wavefront of a frame kernel start and end, and where? rt_exp_wave_clock notes the
constant 100-MHz clock at a wavefront's first and last instruction and its hardware id (XCD, SE, CU, SIMD); this tool turns one
launch of raycast / generate_candidate / spatial_resampling / resolve (frames back to back on one stream, frame 12) into: launch
span, wavefront-time / span = mean wavefronts in flight, the in-flight count over 20 slices of the span, when each XCD ran out of
work, and the span a perfectly packed launch would need (wavefront-time / peak in flight).

  python tools/wave_timeline.py [WxH] [rows=A:B] [rt_tuning k=v ...]     e.g.  python tools/wave_timeline.py 1920x1080 0=0 1=0 3=0
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes  # noqa: E402
from cedec_2024_rt_amd.types import bench_options  # noqa: E402

args = sys.argv[1:]
W, H = (int(v) for v in args.pop(0).split("x")) if args and "x" in args[0] else (1920, 1080)
rows = None
if args and args[0].startswith("rows="):  # a strip (87 halo rows that nobody fills: timing only)
    a, b = args.pop(0)[5:].split(":")
    rows = (int(a), int(b))
r = api.Renderer(W, H, rows=rows, halo=87 if rows else 0, exp=True)
r.set_scene(scenes.make_blocks_restir())
r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
r.set_options(bench_options())
r.tuning(14, 0)
r.tuning(17, 0)
for kv in args:
    k, v = kv.split("=")
    r.tuning(int(k), int(v))
fn = r.L.rt_exp_wave_clock
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
fn.restype = C.c_int
def frame(f):
    if rows:  # the staged frame of a strip, its halo rows never filled (timing only)
        for stage in range(5):
            r.frame_stage(f, stage)
    else:
        r.frame(f)


for f in range(1, 12):
    frame(f)
r.sync()
print("%dx%d%s, rt_tuning %s" % (W, H, " rows %d:%d" % rows if rows else "", " ".join(args) or "defaults"))
for name, kernel, pas in (("raycast", 0, 0), ("generate_candidate", 1, 0), ("spatial_resampling pass 1", 2, 1), ("resolve", 3, 0)):
    assert fn(r.h, kernel, pas, None, 0) == 0
    frame(12)
    r.sync()
    n = 16 * (W // 8 + 2) * (H // 8 + 2) + 4096  # k_raycast_quad: four workgroups per 8 x 8 tile
    buf = np.zeros(n, np.uint64)
    assert fn(r.h, kernel, pas, buf.ctypes.data, n) == 0
    assert fn(r.h, -1, 0, None, 0) == 0
    t0, e = buf[0::2], buf[1::2]
    ok = t0 > 0
    if not ok.any():
        print("\n%s: no launch of its own in this frame (whole frames trace the primary rays inside generate_candidate: rt_tuning 25)" % name)
        continue
    t0, e = t0[ok].astype(np.int64), e[ok]
    t1 = (e & np.uint64(0xFFFFFFFFFF)).astype(np.int64)
    hw = (e >> np.uint64(40)).astype(np.int64)
    t0 &= 0xFFFFFFFFFF
    xcd, se, cu, simd = (hw >> 12) & 15, (hw >> 9) & 7, (hw >> 4) & 31, (hw >> 2) & 3
    base = t0.min()
    t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0  # microseconds
    span = t1.max()
    dur = t1 - t0
    edges = np.linspace(0, span, 21)
    mid = (edges[:-1] + edges[1:]) / 2
    inflight = [(int(((t0 <= m) & (t1 > m)).sum())) for m in mid]
    peak = max(inflight)
    print("\n%s: %d wavefronts, span %.1f us, wavefront-time %.0f us = %.0f in flight on average (peak of the slices %d); packed at the peak: %.1f us"
          % (name, len(t0), span, dur.sum(), dur.sum() / span, peak, dur.sum() / peak))
    print("  wavefront duration us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f; last start at %.1f us" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max(), t0.max()))
    print("  in flight per 5 %% slice: " + " ".join(str(v) for v in inflight))
    print("  XCD: last wavefront ends at us / wavefronts / wavefront-time us: " + "  ".join("%d: %.0f / %d / %.0f" % (k, t1[xcd == k].max(), (xcd == k).sum(), dur[xcd == k].sum()) for k in sorted(set(xcd.tolist()))))
    if os.environ.get("WAVE_TIMELINE_DUMP"):
        idx = np.nonzero(ok)[0]
        np.savez_compressed(os.path.join(os.environ["WAVE_TIMELINE_DUMP"], "waves_%dx%d_k%d.npz" % (W, H, kernel)), wave=idx.astype(np.int32),
                            t0=t0.astype(np.float32), t1=t1.astype(np.float32), xcd=xcd.astype(np.int8), cu=(se * 16 + cu).astype(np.int8))
    cus = {}
    for k in set((xcd * 64 + se * 16 + cu).tolist()):
        m = (xcd * 64 + se * 16 + cu) == k
        cus[k] = dur[m].sum()
    v = np.array(list(cus.values()))
    print("  compute units seen %d; wavefront-time per CU us: min %.0f mean %.0f max %.0f" % (len(v), v.min(), v.mean(), v.max()))
r.close()
