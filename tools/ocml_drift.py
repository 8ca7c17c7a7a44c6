"""GPU. Drift between the product's portable transcendental functions and the DEVICE libm the reference gets.

The reference's kernels are compiled by hiprtc at run time (common/shader.hpp:107-175): on an AMD GPU their `log / cos / sin`
(common/reservoir.hpp:89-95), `expf / powf` (:61-75) and the tone mapper's `powf` (common/kernels/common.cu:58-61) are ocml's, and
hiprtc's default -ffp-contract contracts a*b+c into FMAs. The parity chain of DESIGN.md section 2 links the GPU to the oracle bit for bit
(csrc/portable_math.h on both sides) and the oracle to GLIBC's libm (profiles/r05_portable_drift.json); this tool measures the
remaining link on the MI355X itself: the same frames rendered by
    product   librestir_rt.so                 portable_math.h, -ffp-contract=off          (== the oracle, bit for bit)
    ocml      librestir_rt_ocml.so            ocml's functions, -ffp-contract=off          (csrc/Makefile target `ocml`)
    ocml_fma  librestir_rt_ocml_fma.so        ocml's functions, -ffp-contract=fast         (the flags the reference's kernels get)
and, per frame and variant against the product:
    flipped  pixels whose accumulation value differs in any bit
    rel_l2   || rgb_variant - rgb_product ||_2 / || rgb_product ||_2        (the north star's measure; gate <= 1e-4)
    hist     pixels whose temporal history (the reservoir carried into the next frame) differs in any byte
    pixels8  RGBA8 pixels that differ after tone mapping (what a screenshot shows)
Reported, not gated here (tests/test_gpu_round6.py gates rel_l2 <= 1e-4 at a small size).

  python tools/ocml_drift.py [--frames 30] [--size 1920x1080] [--out profiles/r06_ocml_drift.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compare(api, ref, var):
    a, b = ref.download(api.RT_BUF_ACCUMULATION), var.download(api.RT_BUF_ACCUMULATION)
    flipped = int((a.view(np.uint32) != b.view(np.uint32)).any(axis=1).sum())
    a64, b64 = a.astype(np.float64)[:, :3], b.astype(np.float64)[:, :3]
    rel = float(np.sqrt(((a64 - b64) ** 2).sum()) / np.sqrt((a64 ** 2).sum()))
    ha, hb = ref.download(api.RT_BUF_RES_TEMPORAL), var.download(api.RT_BUF_RES_TEMPORAL)
    shaded = (ha["M"] > 0) | (hb["M"] > 0)  # the reference stores nothing for sky / emissive pixels
    hist = np.zeros(ha.shape[0], bool)
    for name in ha.dtype.names:
        if name == "pad":
            continue
        x, y = np.ascontiguousarray(ha[name]), np.ascontiguousarray(hb[name])
        hist |= (x.view(np.uint8).reshape(x.shape[0], -1) != y.view(np.uint8).reshape(y.shape[0], -1)).any(axis=1)
    pa, pb = ref.download(api.RT_BUF_PIXELS), var.download(api.RT_BUF_PIXELS)
    return dict(flipped=flipped, rel_l2=rel, hist=int((hist & shaded).sum()), pixels8=int((pa != pb).any(axis=1).sum()))


def run(api, scenes, W, H, frames, verbose=True):
    from cedec_2024_rt_amd.types import bench_options

    tris = scenes.make_blocks_restir()
    eye, center = scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT
    libs = {"product": None, "ocml": api.OCML_LIB_PATH, "ocml_fma": api.OCML_FMA_LIB_PATH}
    ctx = {}
    for name, path in libs.items():
        if path and not os.path.exists(path):
            raise SystemExit(f"{path} not built: make -C cedec_2024_rt_amd/csrc ocml")
        r = api.Renderer(W, H, lib_path=path)
        r.set_scene(tris)
        r.lookat(eye, center)
        r.set_options(bench_options())
        ctx[name] = r
    rows = []
    for f in range(1, frames + 1):
        for r in ctx.values():
            r.frame(f)
        row = dict(frame=f)
        for name in ("ocml", "ocml_fma"):
            row[name] = compare(api, ctx["product"], ctx[name])
        rows.append(row)
        if verbose:
            print("frame %3d: " % f + "; ".join("%s %6d px, rel-L2 %.3e, %6d histories, %5d RGBA8" % (n, row[n]["flipped"], row[n]["rel_l2"], row[n]["hist"], row[n]["pixels8"])
                                                for n in ("ocml", "ocml_fma")), flush=True)
    ids = {n: r.build_id() for n, r in ctx.items()}
    for r in ctx.values():
        r.close()
    return rows, ids


def summary(rows, name):
    rs = [r[name] for r in rows]
    return dict(max_rel_l2=max(r["rel_l2"] for r in rs), max_flipped=max(r["flipped"] for r in rs), max_hist=max(r["hist"] for r in rs),
                last_10_mean_rel_l2=float(np.mean([r["rel_l2"] for r in rs[-10:]])), last_10_mean_flipped=float(np.mean([r["flipped"] for r in rs[-10:]])),
                last_10_mean_hist=float(np.mean([r["hist"] for r in rs[-10:]])), max_pixels8=max(r["pixels8"] for r in rs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from cedec_2024_rt_amd import api, scenes

    W, H = (int(v) for v in args.size.split("x"))
    rows, ids = run(api, scenes, W, H, args.frames)
    out = dict(workload=f"blocks_restir stand-in {W}x{H}, bench options, static camera, frames 1..{args.frames}, on the MI355X",
               what="librestir_rt.so (portable_math.h == the oracle bit for bit) against the same sources with ocml's device libm (ocml) and "
                    "with ocml + default FMA contraction (ocml_fma: the flags hiprtc gives the reference's kernels); reported, not gated",
               tolerance_north_star=1e-4, build_ids=ids, ocml=summary(rows, "ocml"), ocml_fma=summary(rows, "ocml_fma"), frames=rows)
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "frames"}))


if __name__ == "__main__":
    main()
