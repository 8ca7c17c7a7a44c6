"""GPU. A/B of BUILDS of the library on the frame bench.py times (config #4, 1920x1080, frames back to back on one stream): the variants
are loaded side by side in ONE process (api.Renderer(lib_path=...)) and measured alternately, `reps` rounds, so that the box's state
is the same for all of them. Per variant: wall-clock ms per un-pipelined frame, ms per pipelined frame, HIP-event time per kernel
(stage 0 as the one launch the headline runs, and as its two kernels), optionally at 3840x2160 too.

  tools/build_variants.sh old "-DRT_NO_DEFER_BARY=1"            # -> gpurun_variants/lib_old.so
  python tools/lib_ab.py new=cedec_2024_rt_amd/librestir_rt.so old=gpurun_variants/lib_old.so [--reps 3] [--4k] [--tuning 16=1,13=0]
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
    ap.add_argument("libs", nargs="+", help="name=path ...")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--4k", dest="four_k", action="store_true")
    ap.add_argument("--tuning", default="")
    ap.add_argument("--rows", default=None, help="A:B = a strip context of those storage rows (per-kernel entry points only)")
    args = ap.parse_args()
    from cedec_2024_rt_amd import api, scenes
    from cedec_2024_rt_amd.types import bench_options

    tris = scenes.make_blocks_restir()
    sizes = [(1920, 1080)] + ([(3840, 2160)] if args.four_k else [])
    for W, H in sizes:
        ctx = {}
        for spec in args.libs:
            name, path = spec.split("=", 1)
            r = api.Renderer(W, H, lib_path=os.path.abspath(path))
            for kv in filter(None, args.tuning.split(",")):
                k, v = kv.split("=")
                r.tuning(int(k), int(v))
            r.set_scene(tris)
            r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
            r.set_options(bench_options())
            ctx[name] = r
        res = {n: [] for n in ctx}
        frame = {n: 0 for n in ctx}

        def run(n, count, sync_each=False):
            r = ctx[n]
            for _ in range(count):
                frame[n] += 1
                r.frame(frame[n])
                if sync_each:
                    r.sync()

        for rep in range(args.reps):
            for n, r in ctx.items():
                row = {}
                for label, t14, t17 in (("unpipelined", 0, 0), ("pipelined", -1, -1)):
                    r.timing_enable(False)
                    r.tuning(14, t14)
                    r.tuning(17, t17)
                    run(n, 6)
                    r.sync()
                    t0 = time.perf_counter()
                    run(n, args.frames)
                    r.sync()
                    row[label] = round((time.perf_counter() - t0) / args.frames * 1e3, 4)
                r.tuning(14, 0)
                r.tuning(17, 0)
                r.timing_enable(True)
                for form, t25 in (("one", -1), ("two", 0)):
                    r.tuning(25, t25)
                    acc = []
                    for _ in range(24):
                        frame[n] += 1
                        r.frame(frame[n])
                        t = r.timing()
                        acc.append([t[k] for k in ("raycast", "generate_candidate", "spatial0", "spatial1", "spatial2", "resolve", "frame")])
                    m = np.array(acc[4:]).mean(axis=0)
                    if form == "one":
                        row.update(stage0=round(float(m[0] + m[1]), 4), spatial=round(float(m[2:5].mean()), 4), resolve=round(float(m[5]), 4), frame_events=round(float(m[6]), 4))
                    else:
                        row.update(raycast=round(float(m[0]), 4), generate=round(float(m[1]), 4))
                r.tuning(25, -1)
                r.timing_enable(False)
                res[n].append(row)
                print(f"{W}x{H} rep {rep} {n}: {json.dumps(row)}", flush=True)
        print(json.dumps({"size": f"{W}x{H}", "median": {n: {k: round(float(np.median([x[k] for x in rows])), 4) for k in rows[0]} for n, rows in res.items()},
                          "build_ids": {n: r.build_id() for n, r in ctx.items()}}), flush=True)
        for r in ctx.values():
            r.close()


if __name__ == "__main__":
    main()
