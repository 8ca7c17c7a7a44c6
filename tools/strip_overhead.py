"""What a rank of an N-strip frame costs when it never waits for a neighbour: the native strip
driver (rt_mg_*, csrc/strip_mg.cpp) with the MIRROR transport (a rank receives the bytes it sent, by a
device copy), alone on the GPU. Same launches, same pack/unpack work and message sizes as a real
exchange; only the xGMI hop and the peers' skew are missing. Reported per N and image size:

  ms_per_frame    wall time per frame of the rank in a steady loop (GPU-bound)
  host_us         host time spent enqueueing one frame (rt_mg_stats.host_ns / frames)
  plan_wait_us    host time waiting for the next frame's halo plan (0 = the plan was ready)
  speedup_bound   single-GPU ms / this rank's ms: the scaling the compute side allows

The N-GPU job runs at the pace of its SLOWEST rank. The table cases time the middle rank of an equal partition (comparable
with r02 / r03_c); the "all ranks" cases time EVERY rank, for equal rows and for the rows bench.py cuts at start-up
(measured cost per strip, averaged over the rounds, fed back into rt_mg_partition, 4 rounds), and report the maximum.

  python tools/strip_overhead.py [--out profiles/r02_strip_overhead.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cedec_2024_rt_amd import api, scenes  # noqa: E402
from cedec_2024_rt_amd.types import bench_options  # noqa: E402


TRANSPORTS = {"mirror": api.RT_MG_TRANSPORT_MIRROR, "rccl_self": api.RT_MG_TRANSPORT_RCCL_SELF}
XGMI_GBS = 153.0  # one xGMI link, one direction (MI355X_MICROARCH.md): a strip talks to each neighbour over a link of its own


def measure(W, H, N, flags, tris, frames=40, warm=6, rank=None, bounds=None, transport="mirror"):
    bounds = bounds or api.mg_partition(H, N)
    rank = N // 2 if rank is None else rank
    a, b = bounds[rank]
    r = api.Renderer(W, H, rows=(a, b), halo=87 if N > 1 else 0)
    r.set_scene(tris)
    r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
    r.set_options(bench_options())
    mg = api.MultiGpu(r, rank, bounds, transport=TRANSPORTS[transport], flags=flags)
    f = 0
    for _ in range(warm):
        f += 1
        mg.frame(f)
    r.sync()
    mg.reset_stats()
    t0 = time.perf_counter()
    for _ in range(frames):
        f += 1
        mg.frame(f)
    t_host = time.perf_counter() - t0
    r.sync()
    wall = time.perf_counter() - t0
    st = mg.stats()
    out = dict(rows=b - a, ms_per_frame=round(wall / frames * 1e3, 4), gpu_event_ms_per_frame=round(st["gpu_ns_per_frame"] / 1e6, 4), host_us=round(st["host_ns"] / frames / 1e3, 1),
               host_loop_us=round(t_host / frames * 1e6, 1), plan_wait_us=round(st["plan_wait_ns"] / frames / 1e3, 1),
               cold_frames=st["cold_frames"], MB_sent_per_frame=round(st["bytes_sent"] / frames / 1e6, 3),
               messages_per_frame=st["messages"] / frames, transport=transport)
    # what neither transport contains: the wire. Bytes per frame and SIDE over one xGMI link, summed over the frame's exchanges
    # (3 on the chain; the two sides travel on different links at the same time) - a stated addend, not part of ms_per_frame
    sides = (1 if rank in (0, N - 1) else 2) if N > 1 else 0
    out["xgmi_wire_us_per_frame"] = round(st["bytes_sent"] / frames / max(sides, 1) / (XGMI_GBS * 1e9) * 1e6, 1) if sides else 0.0
    mg.close()
    r.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--churn", type=int, default=0, help="create and destroy this many strip contexts + drivers first (stream -> hardware queue mapping after a host has re-created its contexts, as bench.py does for cost-weighted strips)")
    ap.add_argument("--only", default=None, help="WxH:N:sparse|dense|onelane|separate[:rank], e.g. 3840x2160:8:sparse (for a kernel trace of one case)")
    ap.add_argument("--transport", default="mirror", choices=sorted(TRANSPORTS), help="mirror: one copy launch per exchange; rccl_self: the real grouped ncclSend/ncclRecv (to self)")
    ap.add_argument("--quick", action="store_true", help="N = 8 only, both sizes, every rank with equal rows (no balance rounds)")
    args = ap.parse_args()
    tris = scenes.make_blocks_restir()
    res = {}
    for _ in range(args.churn):
        measure(1920, 1080, 8, 0, tris, frames=3, warm=1)
    T = args.transport
    if args.only:
        parts = args.only.split(":")
        wh, n, mode = parts[:3]
        w, h = (int(v) for v in wh.split("x"))
        flags = {"sparse": 0, "dense": api.RT_MG_DENSE, "onelane": api.RT_MG_ONE_LANE, "separate": api.RT_MG_SEPARATE_PACK}[mode]
        print(json.dumps({args.only: measure(w, h, int(n), flags, tris, transport=T, rank=int(parts[3]) if len(parts) > 3 else None)}), flush=True)
        return
    if args.quick:
        for (W, H) in ((1920, 1080), (3840, 2160)):
            single = measure(W, H, 1, 0, tris, transport=T)["ms_per_frame"]
            bounds = api.mg_partition(H, 8)
            rows = [measure(W, H, 8, 0, tris, rank=k, bounds=bounds, transport=T) for k in range(8)]
            t = [m["ms_per_frame"] for m in rows]
            row = dict(transport=T, single_ms=single, ms=t, max_ms=max(t), speedup_bound=round(single / max(t), 2),
                       xgmi_wire_us_per_frame=max(m["xgmi_wire_us_per_frame"] for m in rows), MB_sent_per_frame=max(m["MB_sent_per_frame"] for m in rows))
            res[f"{W}x{H} N=8 all ranks, equal rows"] = row
            print(json.dumps({f"{W}x{H} N=8 all ranks, equal rows": row}), flush=True)
        if args.out:
            json.dump(res, open(args.out, "w"), indent=1)
        return
    for (W, H) in ((1920, 1080), (3840, 2160)):
        single = None
        for N in (1, 2, 4, 8):
            for name, flags in (("sparse", 0), ("dense", api.RT_MG_DENSE), ("sparse, one lane", api.RT_MG_ONE_LANE)) if N > 1 else (("single", 0),):
                m = measure(W, H, N, flags, tris, transport=T)
                if N == 1:
                    single = m["ms_per_frame"]
                m["speedup_bound"] = round(single / m["ms_per_frame"], 2)
                res[f"{W}x{H} N={N} {name}"] = m
                print(json.dumps({f"{W}x{H} N={N} {name}": m}), flush=True)
    # every rank, slowest counts: equal rows, then bench.py's start-up balancing
    import numpy as np
    for (W, H) in ((1920, 1080), (3840, 2160)):
        single = res[f"{W}x{H} N=1 single"]["ms_per_frame"]
        for N in (2, 4, 8):
            bounds = api.mg_partition(H, N)
            est = []
            for it in range(5):  # 4 balance rounds as bench.py does them, then the cut they arrive at
                # reported rows (equal rows, the final cut): the lower of two runs per rank - a rank's time here is one process on a
                # shared box, its noise (clock dips, a neighbour's job) only ever adds; the balance rounds in between use one run
                reps = 2 if (it == 0 or it == 4 or N == 2) else 1
                t = [min(measure(W, H, N, 0, tris, frames=40, rank=k, bounds=bounds, transport=T)["ms_per_frame"] for _ in range(reps)) for k in range(N)]
                row = dict(transport=T, rows=[b - a for a, b in bounds], ms=t, max_ms=max(t), speedup_bound=round(single / max(t), 2))
                if it == 0:
                    res[f"{W}x{H} N={N} all ranks, equal rows"] = row
                    print(json.dumps({f"{W}x{H} N={N} all ranks, equal rows": row}), flush=True)
                if it == 4 or N == 2:
                    res[f"{W}x{H} N={N} all ranks, rows cut by measured cost"] = row
                    print(json.dumps({f"{W}x{H} N={N} all ranks, rows cut by measured cost": row}), flush=True)
                    break
                cost = np.zeros(H)
                for (a, b), ms in zip(bounds, t):
                    cost[a:b] = ms / (b - a)
                est.append(cost)
                cost = np.mean(est, axis=0)
                bounds = api.mg_partition(H, N, 87, np.maximum(1, cost / cost.max() * 60000).astype(np.uint32))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
