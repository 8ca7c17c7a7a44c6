"""What a rank of an N-strip frame costs when it never waits for a neighbour: the native strip driver (rt_mg_*,
csrc/strip_mg.cpp) with one rank ALONE on the GPU, its neighbours replaced by a transport that hands it back what it sent:

  --transport mirror      one copy launch per exchange (rt_copy_parts): the compute side only (r02 / r03 tables)
  --transport rccl_self   the real grouped ncclSend/ncclRecv of the RCCL transport, true message sizes, to the rank itself on
                          a one-rank communicator (r04): RCCL's launch + copy kernel are on the chain; the xGMI wire is not
                          (xgmi_wire_us_per_frame = bytes per frame and side / 153 GB/s: a stated addend)
  --transport wire_model  rccl_self + the wire ON the chain (r05): every exchange completes no earlier than
                          max over the neighbours (bytes to / from it) / RT_MG_WIRE_GBS (153) + RT_MG_WIRE_LAT_US (5) after its
                          data was ready on the stream (a dependent delay on the GPU, csrc/strip_mg.cpp post()). THE bound.

  --transport mirror_wire mirror + the same dependent delay (r06): the wire alone, without RCCL's one-rank self-send (which moves a 4K
                          exchange at ~85 GB/s, slower than the link it stands in for): the other bracket of the bound

Same launches, same pack/unpack work and message sizes as a real exchange; the peers' skew is missing. Every measurement
runs in a PROCESS OF ITS OWN (HIP maps streams onto hardware queues by creation history: a process that has created and
destroyed contexts before measures something else — 1.55 instead of 1.07 ms at 4K with RCCL's own streams in the mix,
profiles/r04_strip_notes.txt). The N-GPU job runs at the pace of its SLOWEST rank: every rank is timed (lower of two runs),
for equal rows and for rows re-cut by measured cost per strip (4 rounds, as bench.py's start-up does); the final cuts are
written to profiles/strip_cuts.json, which bench.py uses instead of re-measuring.

  python tools/strip_overhead.py --transport rccl_self --out profiles/r04_strip_overhead_rccl_self.json [--sizes 1920x1080] [--ns 8]
  python tools/strip_overhead.py --only 3840x2160:8:sparse[:rank] --transport mirror      (one case, this process: for rocprofv3)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

XGMI_GBS = 153.0  # one xGMI link, one direction (MI355X_MICROARCH.md): a strip talks to each neighbour over a link of its own
FLAGS = {"sparse": 0, "dense": 1, "onelane": 2, "separate": 4}
CUTS = os.path.join(ROOT, "profiles", "strip_cuts.json")


def measure(W, H, N, flags, tris, frames=40, warm=6, rank=None, bounds=None, transport="mirror"):
    """one rank, in THIS process"""
    from cedec_2024_rt_amd import api, scenes
    from cedec_2024_rt_amd.types import bench_options

    T = {"mirror": api.RT_MG_TRANSPORT_MIRROR, "rccl_self": api.RT_MG_TRANSPORT_RCCL_SELF, "wire_model": api.RT_MG_TRANSPORT_WIRE_MODEL,
         "mirror_wire": api.RT_MG_TRANSPORT_MIRROR_WIRE}[transport]
    bounds = bounds or api.mg_partition(H, N)
    rank = N // 2 if rank is None else rank
    a, b = bounds[rank]
    r = api.Renderer(W, H, rows=(a, b), halo=87 if N > 1 else 0)
    r.set_scene(tris)
    r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
    r.set_options(bench_options())
    mg = api.MultiGpu(r, rank, bounds, transport=T, flags=flags)
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
               messages_per_frame=st["messages"] / frames, transport=transport,
               wire_model_us_per_frame=round(st["wire_ns"] / frames / 1e3, 1))
    if transport == "wire_model":
        out["wire_model"] = dict(GBs=float(os.environ.get("RT_MG_WIRE_GBS", "153")), latency_us=float(os.environ.get("RT_MG_WIRE_LAT_US", "5")))
    sides = (1 if rank in (0, N - 1) else 2) if N > 1 else 0
    out["xgmi_wire_us_per_frame"] = round(st["bytes_sent"] / frames / max(sides, 1) / (XGMI_GBS * 1e9) * 1e6, 1) if sides else 0.0
    mg.close()
    r.close()
    return out


def fresh(W, H, N, mode, rank, bounds, transport):
    """the same in a process of its own; returns the measurement dict"""
    cmd = [sys.executable, os.path.abspath(__file__), "--only", f"{W}x{H}:{N}:{mode}:{rank}", "--transport", transport]
    if bounds is not None:
        cmd += ["--bounds", ",".join(str(v) for v in [bounds[0][0]] + [e for _, e in bounds])]
    for attempt in range(2):
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        for line in p.stdout.splitlines():
            if line.startswith("{"):
                return list(json.loads(line).values())[0]
    raise RuntimeError(f"measurement failed: {' '.join(cmd)}\n{p.stderr[-2000:]}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="WxH:N:sparse|dense|onelane|separate[:rank] — one case in this process")
    ap.add_argument("--bounds", default=None, help="strip edges for --only: 0,a,b,...,H")
    ap.add_argument("--transport", default="mirror", choices=("mirror", "rccl_self", "wire_model", "mirror_wire"))
    ap.add_argument("--sizes", default="1920x1080,3840x2160")
    ap.add_argument("--ns", default="2,4,8")
    ap.add_argument("--rounds", type=int, default=4, help="balance rounds for N = 8 (0: equal rows only)")
    args = ap.parse_args()
    T = args.transport
    if args.only:
        from cedec_2024_rt_amd import scenes

        parts = args.only.split(":")
        w, h = (int(v) for v in parts[0].split("x"))
        n = int(parts[1])
        bounds = None
        if args.bounds:
            e = [int(v) for v in args.bounds.split(",")]
            bounds = [(e[i], e[i + 1]) for i in range(len(e) - 1)]
        m = measure(w, h, n, FLAGS[parts[2]], scenes.make_blocks_restir(), transport=T, rank=int(parts[3]) if len(parts) > 3 else None, bounds=bounds)
        print(json.dumps({args.only: m}), flush=True)
        return

    import numpy as np

    from cedec_2024_rt_amd import api, scenes

    res = {"transport": T, "note": "one rank alone on the GPU, a process per measurement, lower of two runs per rank where two were made; "
                                   "speedup_bound = single-context frame of the same run / slowest rank"}
    cuts = {}
    if os.path.exists(CUTS):
        cuts = json.load(open(CUTS))
    sha = scenes.scene_sha256(scenes.make_blocks_restir())[:16]
    for wh in args.sizes.split(","):
        W, H = (int(v) for v in wh.split("x"))
        single = min(fresh(W, H, 1, "sparse", 0, None, T)["ms_per_frame"] for _ in range(2))
        res[f"{W}x{H} N=1 single"] = dict(ms_per_frame=single)
        print(json.dumps({f"{W}x{H} N=1": single}), flush=True)
        for N in (int(v) for v in args.ns.split(",")):
            bounds = api.mg_partition(H, N)
            est = []
            rounds = args.rounds if N == 8 else 0
            for it in range(rounds + 1):
                final = it == rounds
                reps = 2 if (it == 0 or final) else 1
                runs = [[fresh(W, H, N, "sparse", k, bounds, T) for _ in range(reps)] for k in range(N)]
                t = [min(m["ms_per_frame"] for m in rk) for rk in runs]
                row = dict(rows=[b - a for a, b in bounds], ms=t, max_ms=max(t), speedup_bound=round(single / max(t), 2),
                           MB_sent_per_frame=max(rk[0]["MB_sent_per_frame"] for rk in runs),
                           xgmi_wire_us_per_frame=max(rk[0]["xgmi_wire_us_per_frame"] for rk in runs),
                           host_us=max(rk[0]["host_us"] for rk in runs),
                           wire_model_us_per_frame=max(rk[0].get("wire_model_us_per_frame", 0.0) for rk in runs))
                name = "equal rows" if it == 0 else ("rows cut by measured cost" if final else f"balance round {it}")
                res[f"{W}x{H} N={N} all ranks, {name}"] = row
                print(json.dumps({f"{W}x{H} N={N} all ranks, {name}": row}), flush=True)
                if final:
                    if rounds:
                        cuts[f"{W}x{H}:{N}:{sha}"] = dict(bounds=[bounds[0][0]] + [e for _, e in bounds], transport=T, max_ms=max(t), build_id=api.build_id(),
                                                          equal_rows_max_ms=res[f"{W}x{H} N={N} all ranks, equal rows"]["max_ms"])
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
    if cuts:
        with open(CUTS, "w") as f:
            json.dump(cuts, f, indent=1)


if __name__ == "__main__":
    main()
