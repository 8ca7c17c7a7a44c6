"""CPU. The steady-state drift of the libm -> portable-math substitution: the link between what the reference
computes (glibc transcendental functions: the oracle in MATH_LIBM mode == oracle/_ref bit for bit) and what the GPU computes
(csrc/portable_math.h: the oracle in MATH_PORTABLE mode == the GPU bit for bit) is GATED on two frames (tests/test_portable_math.py:
rel-L2 <= 1e-4). With the temporal history capped at M = 640 (20 x 32 candidates, 10_restir_di.cu:185-187) a flipped reservoir
decision lives on in the history of later frames, so this tool REPORTS — it gates nothing — the same two numbers per frame over a
long sequence of the benchmark workload (blocks_restir stand-in, 1920x1080, bench options, static camera):

  flipped  pixels whose accumulation value differs in any bit between the two modes
  rel_l2   || rgb_libm - rgb_portable ||_2 / || rgb_libm ||_2 of the frame (the north star's measure)
  hist     pixels whose temporal history (the reservoir carried into the next frame) differs in any field

  python tools/portable_drift.py [--frames 30] [--size 1920x1080] [--out profiles/r05_portable_drift.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from cedec_2024_rt_amd import scenes
    from oracle import binding as ob

    W, H = (int(v) for v in args.size.split("x"))
    tris = scenes.make_blocks_restir()
    eye, center = scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT
    rg = ob.raygen_lookat(eye, center, (0, 1, 0), np.float32(np.pi) / np.float32(4), W, H)
    eyev = np.asarray(eye, np.float32)
    ob.set_threads(ob.effective_cpus())
    sc = {}
    st = {}
    for mode in (ob.MATH_LIBM, ob.MATH_PORTABLE):
        ob.set_math_mode(mode)
        sc[mode] = ob.Scene(tris, use_bvh=True)
        st[mode] = ob.new_state(W, H)
    rows = []
    t0 = time.time()
    for f in range(1, args.frames + 1):
        for mode in (ob.MATH_LIBM, ob.MATH_PORTABLE):
            ob.set_math_mode(mode)
            sc[mode].frame(W, H, f, rg, eyev, ob.bench_options(), st[mode], tone_map=False)
        a, b = st[ob.MATH_LIBM]["accum"], st[ob.MATH_PORTABLE]["accum"]
        flipped = int((a.view(np.uint32) != b.view(np.uint32)).any(axis=1).sum())
        a64, b64 = a.astype(np.float64)[:, :3], b.astype(np.float64)[:, :3]
        rel = float(np.sqrt(((a64 - b64) ** 2).sum()) / np.sqrt((a64 ** 2).sum()))
        ha, hb = st[ob.MATH_LIBM]["temporal"], st[ob.MATH_PORTABLE]["temporal"]
        hist = np.zeros(ha.shape[0], bool)
        for name in ha.dtype.names:
            if name == "pad":
                continue
            x, y = np.ascontiguousarray(ha[name]), np.ascontiguousarray(hb[name])
            d = x.view(np.uint8).reshape(x.shape[0], -1) != y.view(np.uint8).reshape(y.shape[0], -1)
            hist |= d.any(axis=1)
        rows.append(dict(frame=f, flipped=flipped, rel_l2=rel, hist=int(hist.sum())))
        print(f"frame {f:3d}: {flipped:6d} of {W * H} pixels differ, rel-L2 {rel:.3e}, {int(hist.sum()):6d} histories differ   ({time.time() - t0:.0f} s)", flush=True)
    ob.set_math_mode(ob.MATH_PORTABLE)
    out = dict(workload=f"blocks_restir stand-in {W}x{H}, bench options, static camera, frames 1..{args.frames}",
               what="oracle MATH_LIBM (== the reference's sources compiled in place) against oracle MATH_PORTABLE (== the GPU), per frame; reported, not gated",
               tolerance_north_star=1e-4, max_rel_l2=max(r["rel_l2"] for r in rows), max_flipped=max(r["flipped"] for r in rows),
               last_10_mean_rel_l2=float(np.mean([r["rel_l2"] for r in rows[-10:]])), last_10_mean_flipped=float(np.mean([r["flipped"] for r in rows[-10:]])),
               frames=rows)
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "frames"}))


if __name__ == "__main__":
    main()
