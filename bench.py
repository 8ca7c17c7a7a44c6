#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: Mray/s (+ ms/frame) of the ReSTIR DI frame.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1. Called plainly (no RANK / WORLD_SIZE in the environment) with --gpus N > 1, this process starts
N FRESH child processes of itself — one per GPU, rendezvous on 127.0.0.1 — BEFORE it makes any GPU call, relays rank 0's
single JSON line and exits non-zero with a `"value": null` line if any child fails (launch_ranks below). It never re-execs.

A step = one frame = the timed region of the reference (examples/10_restir_di/10_restir_di.cpp:
254-383): raycast, generate_candidate(+temporal_resampling), 3 x spatial_resampling, resolve,
tone_mapping on the synthetic `blocks_restir` stand-in scene at 1920x1080, 1 spp, benchmark
options of SURVEY.md §8(d) (temporal + spatial reuse on, static camera). Inputs are resident in
HBM before the timed region. Rays are counted as BASELINE.md §3 defines (one raytrace() call =
one ray): N primary + 2 per shaded pixel. RT_SCENE_OBJ=/path/blocks_restir.obj renders a user-supplied
OBJ instead of the stand-in (flagged in config.scene).

N > 1: the frame is cut into N row strips (strong scaling: total work fixed), one process per GPU, driven
by the NATIVE strip driver of librestir_rt.so (csrc/strip_mg.cpp): sparse 87-row reservoir halos
exchanged with RCCL send/recv over xGMI before each spatial pass, no host wait in a steady frame.
torch.distributed only distributes the RCCL unique id and reduces the timings.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     — spatial_resampling: SURVEY §8(d) algorithmic bytes per launch / HIP-event time (`frac`, the
                 contract figure) next to the counter-measured HBM traffic of the same binary
                 (`hbm_frac_measured`; null when profiles/ holds no PMC pass of this library build)
  cpu_baseline — the oracle (CPU restatement) timed on this host's cores, N=1 only.
"""
import argparse
import hashlib
import os

# HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4). The frame uses five streams that must run
# beside each other (main, pipelined stage 0, tail, second lane, RCCL): with 4 queues two of them share one and which two
# depends on creation order (profiles/r03_hw_queue_mapping.txt: 0.40 or 0.60 ms per frame at 1080p in 8 strips). The
# runtime reads the variable when it initialises, i.e. before torch touches the GPU: set here, first thing.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import json
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 1920, 1080
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HALO = 87
# strip heights: rounds of "every rank times its own strip alone, the strips are re-cut by measured cost per row" (make_strip)
BALANCE_ROUNDS = 4


def cpu_baseline(tris, eye, center, frames=2):
    """The oracle timed on the host cores: `frames` full 1080p frames, all OpenMP threads."""
    import numpy as np

    from oracle import binding as ob

    ob.set_math_mode(ob.MATH_PORTABLE)
    threads = ob.effective_cpus()  # cgroup CPU quota, not the 256 hardware threads the box shows
    ob.set_threads(threads)
    hw = os.cpu_count()
    sc = ob.Scene(tris, use_bvh=True)
    rg = ob.raygen_lookat(eye, center, (0, 1, 0), np.float32(np.pi) / np.float32(4), W, H)
    opt = ob.bench_options()
    st = ob.new_state(W, H)
    cnt = ob.new_counters()
    eyev = np.asarray(eye, np.float32)
    # untimed warm-up of the thread pool / page faults on a small frame
    sc.frame(64, 36, 1, ob.raygen_lookat(eye, center, (0, 1, 0), np.float32(np.pi) / np.float32(4), 64, 36), eyev, opt,
             ob.new_state(64, 36), None, tone_map=True)
    t0 = time.perf_counter()
    for f in range(1, frames + 1):
        sc.frame(W, H, f, rg, eyev, opt, st, cnt)
    dt = time.perf_counter() - t0
    rays = int(cnt["rays"][0])
    return dict(value=rays / dt / 1e6, unit="Mray/s", cores=threads, kind="port",
                sample=f"{frames} full frames of the same workload ({W}x{H}, frames 1..{frames}), "
                       f"oracle/restir_oracle.c + its CPU BVH, OpenMP {threads} threads (= the cgroup CPU quota; "
                       f"the host shows {hw} hardware threads), {dt:.2f} s",
                ms_per_frame=dt / frames * 1e3), st


def lib_sha256():
    from cedec_2024_rt_amd import api

    h = hashlib.sha256()
    # the file the contexts of this process run (RT_EXPERIMENTS=1 / RT_LIB_PATH select another one)
    path = os.environ.get("RT_LIB_PATH", api.EXP_LIB_PATH if os.environ.get("RT_EXPERIMENTS") else api.LIB_PATH)
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def committed_pmc(sha, build_id):
    """Counter data of THIS library build, if a PMC pass of it was committed. tools/profile_round.sh writes next to the
    counters the library's rt_build_id (hash of sources + flags: survives a clean rebuild of the same sources) and the
    SHA-256 of the .so file (differs between rebuilds). Nothing is reported for another build's counters."""
    out = dict(traffic=None, source=None, valu=None)
    p = os.path.join(ROOT, "profiles", "spatial_pmc_latest.json")
    if not os.path.exists(p):
        return out
    try:
        with open(p) as f:
            d = json.load(f)
    except Exception:
        return out
    same = (d.get("build_id") not in (None, "unknown") and d.get("build_id") == build_id) or d.get("lib_sha256") == sha
    src = {"round": d.get("round"), "build_id": d.get("build_id"), "lib_sha256": d.get("lib_sha256"), "matches_this_build": bool(same)}
    out["source"] = src
    if src["matches_this_build"]:
        out["traffic"] = d.get("hbm_bytes_per_launch")
        out["valu"] = d.get("valu_issue_frac")
    return out


class Watchdog:
    """Progress watchdog of the multi-GPU run: the RCCL transport of the native strip
    driver meets real neighbours for the first time in an unattended run. Every phase of the run ticks; if a rank makes no
    progress for `limit` seconds (a grouped send/recv that never completes, a barrier a dead peer never reaches), the rank
    says on stderr where it stalled, rank 0 prints ONE diagnostic JSON line (value null, "error"), and the process leaves
    with os._exit(3) — no destructor may wait on a stuck stream, and a process that touched the GPU is never re-exec'ed."""

    instance = None

    def __init__(self, json_fd, rank, world, args):
        import threading

        Watchdog.instance = self
        self.json_fd, self.rank, self.world, self.args = json_fd, rank, world, args
        self.where, self.limit, self.t = "start-up", 600.0, time.monotonic()
        self.done = False
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def tick(self, where, limit=None):
        self.where, self.t = where, time.monotonic()
        if limit is not None:
            self.limit = float(limit)

    def stop(self):
        self.done = True

    def _run(self):
        while not self.done:
            time.sleep(0.5)
            idle = time.monotonic() - self.t
            if not self.done and idle > self.limit:
                self.fail(f"bench.py watchdog: rank {self.rank} of {self.world} made no progress for {idle:.0f} s in phase '{self.where}'")

    def fail(self, msg):
        self.done = True
        sys.stderr.write(msg + "\n")
        sys.stderr.flush()
        if self.rank == 0:
            os.write(self.json_fd, (json.dumps(null_line(self.args, self.world, msg)) + "\n").encode())
        os._exit(3)


class _PythonStrips:
    """bench.py's view of the round-1 Python strip schedule (fallback only): the rt_mg methods the loop uses"""

    def __init__(self, sf):
        self.sf = sf

    def frame(self, frame, clear_first=False):
        self.sf.frame(frame, clear_first)

    def reset_stats(self):
        pass

    def stats(self):
        return dict(frames=0, cold_frames=0, host_ns=0, plan_wait_ns=0, bytes_sent=0, messages=0, records_sent=0, gpu_ns_per_frame=0)

    def close(self):
        pass


NULL_LINE = {"metric": "Mray/s", "value": None, "unit": "Mray/s", "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
             "vs_baseline": None, "dtype": "f32", "data": "synthetic",
             "config": {"workload": "10_restir_di blocks_restir stand-in ReSTIR DI (run aborted)"}}


def null_line(args, world, msg):
    return dict(NULL_LINE, n_gpus=world, steps=args.steps, warmup=args.warmup, error=msg)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: this process — which has made NO GPU call and has not
    even imported torch — starts N children of this script, one per GPU (RANK = LOCAL_RANK = 0..N-1, WORLD_SIZE = N, rendezvous
    on 127.0.0.1 at a free port), exactly the environment torch.distributed.run would give them. Rank 0's stdout is read here
    and its ONE JSON line is relayed; the other ranks' stdout goes to stderr. If a child exits non-zero the others get
    BENCH_CHILD_GRACE_S seconds to finish by themselves (every rank has the progress watchdog; rank 0 prints the diagnostic
    line), then the ones THIS process started are terminated by PID. Exit code: 0 only if every child returned 0 and the
    line carries a value."""
    import socket
    import subprocess
    import threading

    n = args.gpus
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile

    rdv_dir = tempfile.mkdtemp(prefix="bench_rdv_")  # the children's file store lives here (no port race); removed below
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), BENCH_LAUNCHED_BY_PARENT="1",
                BENCH_RENDEZVOUS_FILE=os.path.join(rdv_dir, "store"))
    me = os.path.abspath(__file__)
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
        procs.append(subprocess.Popen([sys.executable, me] + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      stderr=None, cwd=os.getcwd()))
    lines = []

    def pump():
        for raw in procs[0].stdout:
            lines.append(raw.decode(errors="replace").rstrip("\n"))

    th = threading.Thread(target=pump, daemon=True)
    th.start()
    limit = float(os.environ.get("BENCH_LAUNCH_TIMEOUT_S", "3000"))
    grace = float(os.environ.get("BENCH_CHILD_GRACE_S", "90"))
    t0 = time.monotonic()
    first_bad = None
    why = None
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        now = time.monotonic()
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad and first_bad is None:
            first_bad = now
            why = "rank %d exited with code %d" % bad[0]
        if (first_bad is not None and now - first_bad > grace) or now - t0 > limit:
            why = why or "no result after %.0f s" % limit
            for p in procs:  # exactly the processes started above, by PID
                if p.poll() is None:
                    p.terminate()
            time.sleep(3.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    for p in procs:
        p.wait()
    th.join(timeout=10.0)
    import shutil

    shutil.rmtree(rdv_dir, ignore_errors=True)
    rcs = [p.returncode for p in procs]
    out = None
    for ln in lines:
        try:
            d = json.loads(ln)
        except Exception:
            sys.stderr.write(ln + "\n")
            continue
        if isinstance(d, dict) and "metric" in d:
            out = d
    ok = all(rc == 0 for rc in rcs) and out is not None and out.get("value") is not None
    if out is None or (not ok and out.get("value") is not None):
        out = null_line(args, n, why or "child return codes %s, no JSON line from rank 0" % rcs)
    if not ok:
        out.setdefault("error", why or "child return codes %s" % rcs)
        out["child_return_codes"] = rcs
    out["launched_by"] = "bench.py itself: %d fresh child processes, started before any GPU call" % n
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()
    raise SystemExit(0 if ok else 1)


def main():
    try:
        _main()
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001 — a failed rank must not leave the others waiting at a barrier for ever
        import traceback

        traceback.print_exc()
        wd = Watchdog.instance
        if wd is not None:
            wd.fail(f"rank {wd.rank} of {wd.world} failed in phase '{wd.where}': {type(e).__name__}: {e}")
        raise


def _main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--width", type=int, default=W)
    ap.add_argument("--height", type=int, default=H)
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        launch_ranks(args, sys.argv[1:])  # does not return

    # the contract is ONE JSON line on stdout: libraries that print banners to fd 1 (RCCL's version block,
    # gloo's connection messages) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch

    from cedec_2024_rt_amd import api, scenes
    from cedec_2024_rt_amd.types import bench_options

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    wd = Watchdog(json_fd, rank, world, args)
    frame_limit = float(os.environ.get("BENCH_WATCHDOG_S", "60"))  # seconds without a completed step
    stall_at = tuple(int(v) for v in os.environ["BENCH_TEST_STALL"].split(":")) if os.environ.get("BENCH_TEST_STALL") else None
    def die(msg):
        """a start-up condition no rank can recover from: rank 0 still prints the ONE JSON line (value null), exit code 2"""
        wd.stop()
        sys.stderr.write("bench.py: " + msg + "\n")
        if rank == 0:
            os.write(json_fd, (json.dumps(null_line(args, world, msg)) + "\n").encode())
        raise SystemExit(2)

    if world != args.gpus:
        die(f"--gpus {args.gpus} but WORLD_SIZE={world}: start `python bench.py --gpus N` plainly (it launches its own ranks) or under "
            "torch.distributed.run with --nproc-per-node N")
    if not torch.cuda.is_available():
        die("bench.py needs a GPU: the HIP path has no CPU fallback")
    # development aid for 1-GPU boxes (RCCL refuses two ranks on one device): BENCH_DEV_MIRROR=1 runs the N ranks on
    # GPU 0 with the MIRROR transport (every rank receives what it sent) and a gloo control plane. It exercises this
    # script's N > 1 branch and the native driver's launch sequence; its images and timings are NOT a multi-GPU result.
    dev_mirror = bool(os.environ.get("BENCH_DEV_MIRROR")) and world > 1
    # BENCH_DEV_SHM=1: the same N processes on GPU 0, but with the SHM transport (host-staged through shared memory):
    # exact images, so BENCH_VERIFY=1 can compare the assembled N-rank frame with a single context. Slow by design.
    dev_shm = bool(os.environ.get("BENCH_DEV_SHM")) and world > 1
    if dev_shm:
        dev_mirror = True  # same control plane (gloo) and device placement
    if dev_mirror:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        die(f"rank {rank} wants GPU {local_rank} and this node shows {torch.cuda.device_count()} (BENCH_DEV_SHM=1 / BENCH_DEV_MIRROR=1 run "
            "the N ranks on GPU 0 as a protocol check)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if dev_mirror else dev
    dist = None
    if world > 1:
        import torch.distributed as dist

        # ranks started by bench.py itself rendezvous through a file of the parent's private directory (a port picked by
        # bind(0) + close can be taken by someone else before rank 0 binds it); under torch.distributed.run the launcher's TCP store
        rdv = os.environ.get("BENCH_RENDEZVOUS_FILE")
        kw = dict(init_method="file://" + rdv, rank=rank, world_size=world) if rdv else {}
        if dev_mirror:
            dist.init_process_group(backend="gloo", **kw)
        else:
            dist.init_process_group(backend="nccl", device_id=dev, **kw)  # RCCL; only barriers, the id broadcast and reductions

    width, height = args.width, args.height
    scene_desc = {"generator": "scenes.make_blocks_restir (seed 2024)", "stand_in": True}
    obj = os.environ.get("RT_SCENE_OBJ")
    if obj:
        tris = scenes.load_obj(obj)  # a user-supplied blocks_restir.obj (missing from the reference checkout)
        scene_desc = {"obj": os.path.basename(obj), "stand_in": False}
    else:
        tris = scenes.make_blocks_restir()
    eye, center = scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT
    opt = bench_options()
    K, Wm = args.steps, args.warmup

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_sum(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    def make_renderer(w, h, rows=None):
        t0 = time.perf_counter()
        r = api.Renderer(w, h, device=local_rank, rows=rows, halo=HALO if world > 1 else 0)
        r.set_scene(tris)
        build_ms = r.build_ms()  # rt_scene_set alone: upload + tables + BVH build, synchronised (not context creation)
        r.lookat(eye, center)
        r.set_options(opt)
        return r, build_ms

    def make_strip(w, h):
        """This rank's strip context + native driver. Strip heights are cut by MEASURED cost: the job runs at the pace of
        its slowest rank, and what a strip costs is far from proportional to its rows or its shaded pixels (fixed work per
        frame, halo work per boundary; a shaded-pixel model made the slowest strip 7 % slower than equal rows at 4K,
        profiles/r03_strip_balance.jsonl). So each rank times a few frames of its own strip with the MIRROR transport (same
        launches and message sizes, no neighbour involved), the times are all-gathered, every rank re-cuts the rows by
        piecewise-constant cost per row averaged over the rounds so far (rt_mg_partition, >= 87 rows each), BALANCE_ROUNDS
        times. Start-up work, outside the timed region; the strips are in config.strips."""
        uid = [api.mg_unique_id() if rank == 0 and not dev_mirror else None]
        if dev_shm:
            uid = [f"rtmg_{os.getpid()}_{time.time_ns():x}_{w}x{h}" if rank == 0 else None]  # per-run nonce: never a stale segment
        dist.broadcast_object_list(uid, src=0)
        bounds = api.mg_partition(h, world, HALO)
        part = "equal rows"
        # the cut tools/strip_overhead.py arrived at for this (size, N, scene) — measured once, committed (profiles/strip_cuts.json):
        # no start-up rounds (4 x (context + BVH build + 46 frames) per rank before the run). BENCH_REBALANCE=1
        # measures anyway.
        cached = None
        try:
            with open(os.path.join(ROOT, "profiles", "strip_cuts.json")) as f:
                cached = json.load(f).get(f"{w}x{h}:{world}:{scenes.scene_sha256(tris)[:16]}")
        except Exception:
            cached = None
        if cached and not os.environ.get("BENCH_REBALANCE") and not os.environ.get("BENCH_EQUAL_STRIPS"):
            e = cached["bounds"]
            worse = cached.get("equal_rows_max_ms", 0.0) > 0.0 and cached.get("max_ms", 0.0) >= cached["equal_rows_max_ms"]
            if worse:
                # the measured cut did not beat equal rows (differences of 1-2 % are the boxes' spread): equal rows, and say so
                part = "equal rows (the cut cached in profiles/strip_cuts.json measured %.3f ms against %.3f for equal rows)" % (cached["max_ms"], cached["equal_rows_max_ms"])
            elif len(e) == world + 1 and e[0] == 0 and e[-1] == h and all(e[i + 1] - e[i] >= HALO for i in range(world)):
                bounds = [(int(e[i]), int(e[i + 1])) for i in range(world)]
                # the cut is a load-balance heuristic measured with one library build: it stays usable with another, but never silently
                same = cached.get("build_id") == api.build_id()  # exp=None: the library this process's contexts load
                part = "rows cut by measured cost per strip (cached: profiles/strip_cuts.json, slowest strip alone %.3f -> %.3f ms, measured with %s)" % (
                    cached.get("equal_rows_max_ms", 0.0), cached.get("max_ms", 0.0),
                    "this library build" if same else "ANOTHER library build %s: BENCH_REBALANCE=1 re-measures" % cached.get("build_id", "(unrecorded)"))
            else:
                cached = None
        # ranks that share one GPU (the dev transports) cannot time their strips: equal rows there (BENCH_FORCE_BALANCE runs the
        # rounds anyway, to exercise this code on a one-GPU box)
        if not cached and not os.environ.get("BENCH_EQUAL_STRIPS") and ((not dev_shm and not dev_mirror) or os.environ.get("BENCH_FORCE_BALANCE")):
            est, slowest = [], []
            for it in range(BALANCE_ROUNDS):
                wd.tick("strip balance round %d" % it, 600)
                rb, _ = make_renderer(w, h, bounds[rank])
                mgb = api.MultiGpu(rb, rank, bounds, transport=api.RT_MG_TRANSPORT_MIRROR)
                for f in range(1, 7):
                    mgb.frame(f)
                rb.sync()
                t0 = time.perf_counter()
                for f in range(7, 47):
                    mgb.frame(f)
                rb.sync()
                mine = (time.perf_counter() - t0) / 40 * 1e3
                mgb.close()
                rb.close()
                allt = [None] * world
                dist.all_gather_object(allt, mine)
                slowest.append(max(allt))
                cost = np.zeros(h)
                for (a, b), ms in zip(bounds, allt):
                    cost[a:b] = ms / (b - a)
                est.append(cost)
                cost = np.mean(est, axis=0)  # the estimates of all rounds so far, averaged per row: one round alone is +-1 % noisy
                bounds = api.mg_partition(h, world, HALO, np.maximum(1, cost / cost.max() * 60000).astype(np.uint32))
            part = "rows cut by measured cost per strip (%d rounds; slowest strip alone, equal rows -> last measured cut: %.3f -> %.3f ms)" % (
                BALANCE_ROUNDS, slowest[0], slowest[-1])
        r, build_ms = make_renderer(w, h, bounds[rank])
        def python_strips(why):
            """every rank TOGETHER: the round-1 schedule (Python StripFrame over torch.distributed send/recv: same HIP
            kernels, same images, a slower host loop)"""
            from cedec_2024_rt_amd import strips

            if rank == 0:
                sys.stderr.write("native strip driver not used (%s): Python strip schedule over torch.distributed\n" % why)
            r.set_stream(torch.cuda.current_stream().cuda_stream)  # torch's stream orders packs, sends and unpacks
            be = strips.HipStripBackend(r, dev, host_staging=dev_mirror)  # gloo (dev mode) moves host memory only
            return _PythonStrips(strips.StripFrame(be, bounds, rank, transport=strips.DistTransport(dist), halo=HALO, sparse=True))

        if os.environ.get("BENCH_FORCE_FALLBACK"):
            mg = python_strips("BENCH_FORCE_FALLBACK")
            part += ", FALLBACK: Python StripFrame over torch.distributed"
        elif dev_shm:
            mg = api.MultiGpu(r, rank, bounds, transport=api.RT_MG_TRANSPORT_SHM, shm_name=uid[0])
        elif dev_mirror:
            mg = api.MultiGpu(r, rank, bounds, transport=api.RT_MG_TRANSPORT_MIRROR, unique_id=uid[0])
        else:
            # The RCCL transport of the native driver has only ever run on one rank (the development boxes have one
            # GPU). If its communicator cannot be created on this node, every rank falls back together and the line
            # says so, rather than losing the multi-GPU measurement.
            mg, err = None, None
            try:
                mg = api.MultiGpu(r, rank, bounds, transport=api.RT_MG_TRANSPORT_RCCL, unique_id=uid[0])
            except Exception as e:  # noqa: BLE001
                err = f"rank {rank}: {e}"
            errs = [None] * world
            dist.all_gather_object(errs, err)
            if any(errs):
                if mg is not None:
                    mg.close()
                if rank == 0:  # every rank's own failure: WHICH RCCL call failed, its code and ncclGetLastError's text (strip_mg.cpp MG_NCCL)
                    for e in errs:
                        if e:
                            sys.stderr.write("native RCCL strip driver: %s\n" % e)
                mg = python_strips(next(e for e in errs if e))
                part += ", FALLBACK: Python StripFrame over torch.distributed (native RCCL driver failed to initialise)"
        return r, mg, bounds, build_ms, part

    def timed(r, mg, step, frame, w, h, steps, warm):
        """`warm` untimed frames, then EXACTLY `steps` frames between barrier + synchronize on both sides; max over ranks"""
        for _ in range(warm):
            frame += 1
            wd.tick(f"{w}x{h} warm-up frame {frame}", frame_limit)
            step(frame)
        wd.tick(f"{w}x{h} synchronise after the warm-up", frame_limit)
        r.sync()
        torch.cuda.synchronize()
        rays = reduce_sum(r.ray_count()[0])
        if mg is not None:
            mg.reset_stats()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            frame += 1
            wd.tick(f"{w}x{h} timed frame {frame}")  # one store per frame; the limit stays the frame limit
            if stall_at == (rank, frame):  # BENCH_TEST_STALL="rank:frame": this rank stops here (watchdog self-test)
                time.sleep(1e6)
            step(frame)
        wd.tick(f"{w}x{h} barrier at the end of the timed region")
        r.sync()  # the context's own streams (non-blocking streams are not covered by the null-stream synchronize)
        barrier()
        dt = reduce_max(time.perf_counter() - t0)
        wd.tick(f"{w}x{h} after the timed region", 600)
        return frame, rays, dt

    def run(w, h, steps, warm, pipelined=True):
        """warm-up + timed region on the current N GPUs; returns the rank-0 summary pieces. pipelined=False (one GPU): the frames
        run back to back on ONE stream, every kernel of frame f before any kernel of frame f+1 — the reference's timed region
        (10_restir_di.cpp:254-383) repeated, SURVEY 8(d)'s frame time; pipelined=True: stage 0 of frame f+1 beside the passes of
        frame f (rt_tuning 14, 17 at their defaults)."""
        if world == 1:
            r, build_ms = make_renderer(w, h)
            if not pipelined:
                r.tuning(14, 0)
                r.tuning(17, 0)
            step, mg, bounds, part = r.frame, None, [(0, h)], "single GPU"
        else:
            r, mg, bounds, build_ms, part = make_strip(w, h)
            step = mg.frame
        frame, rays, dt = timed(r, mg, step, 0, w, h, steps, warm)
        return dict(r=r, mg=mg, frame=frame, rays=rays, dt=dt, bounds=bounds, build_ms=build_ms, part=part, builder=r.bvh_builder())

    wd.tick("scene + context set-up", 600)
    # N = 1: the headline is SURVEY 8(d)'s frame — the reference's timed region, frames back to back; the
    # throughput of pipelined frames is measured right after it on the same context and reported as `value_pipelined`.
    # N > 1: the native strip driver's frame loop (always pipelined: its look-ahead is what hides the halo exchanges).
    R = run(width, height, K, Wm, pipelined=(world > 1))
    r, mg, frame, total_rays, elapsed = R["r"], R["mg"], R["frame"], R["rays"], R["dt"]
    info = r.scene_info()
    pipelined_dt = None
    # profiling runs (tools/profile_round.sh: RT_TUNING=14=0,17=0) keep every kernel of the process un-overlapped: no pipelined region
    no_pipelined = bool(os.environ.get("BENCH_NO_PIPELINED")) or any(kv.split("=")[0] in ("14", "17") for kv in filter(None, os.environ.get("RT_TUNING", "").split(",")))
    if world == 1 and not no_pipelined:
        r.tuning(14, -1)
        r.tuning(17, -1)
        frame, _, pipelined_dt = timed(r, None, r.frame, frame, width, height, K, Wm)
        r.tuning(14, 0)  # the per-kernel event loop, the PCIe-inclusive loop and the walk counters below: un-overlapped frames again
        r.tuning(17, 0)

    single_ref = None
    spatial_ms = per_kernel = algo_bytes = pcie_ms = event_median = two_launch = None
    verified = mg_stats = walks = verified_seq = None
    if world == 1:
        # algorithmic bytes of the three spatial launches of one timed frame (RNG replay, untimed): pass k of
        # frame f reads the buffer the previous pass wrote; the count only depends on the shaded bits
        fmid = Wm + 1 + K // 2
        algo_bytes = float(np.mean([r.spatial_bytes(fmid, k, api.RT_RES_0)[0] for k in range(3)]))
        # per-kernel HIP events on the context's stream over K more frames: per-launch average of the
        # roofline kernel, and the GPU-event median of the frame (SURVEY §8d's definition of the frame time)
        names = ("clear", "raycast", "generate_candidate", "spatial0", "spatial1", "spatial2", "resolve", "tone_mapping", "frame")
        r.timing_enable(True)

        def timed_rows(n):
            nonlocal frame
            rows_, one = [], True
            for _ in range(n):
                frame += 1
                r.frame(frame)
                t = r.timing()
                one = one and r.stage0_one_launch()
                rows_.append([t[k] for k in names])
            return np.array(rows_), one

        # r06: the events bracket the launches the HEADLINE frames run — stage 0 as ONE launch (rt_tuning 25)
        # is one entry, `stage0` (the bracket where the raycast launch would be is empty and is added to it, so that the entries
        # sum to `frame`); the two kernels of rt_raycast / rt_generate_candidate are timed right after it (`kernel_ms_two_launch_stage0`)
        rows, one_launch = timed_rows(max(K, 50))
        mean = rows.mean(axis=0)
        if one_launch:
            per_kernel = {"clear": round(float(mean[0]), 4), "stage0": round(float(mean[1] + mean[2]), 4)}
        else:
            per_kernel = {"clear": round(float(mean[0]), 4), "raycast": round(float(mean[1]), 4), "generate_candidate": round(float(mean[2]), 4)}
        per_kernel.update({k: round(float(x), 4) for k, x in zip(names[3:], mean[3:])})
        per_kernel["stage0_form"] = ("one launch: primary ray + candidates + temporal merge (k_generate_candidate<..., RAYCAST>), as in the frames `value` times"
                                     if one_launch else "two launches: k_raycast, k_generate_candidate")
        per_kernel["sum_over_frame"] = round(float(sum(mean[:8]) / mean[8]), 4)
        spatial_ms = float(rows[:, 3:6].mean())
        event_median = float(np.median(rows[:, 8]))
        two_launch = None
        if one_launch:
            r.tuning(25, 0)
            rows2, _ = timed_rows(max(K // 2, 20))
            r.tuning(25, -1)
            m2 = rows2.mean(axis=0)
            two_launch = {"raycast": round(float(m2[1]), 4), "generate_candidate": round(float(m2[2]), 4), "frame": round(float(m2[8]), 4),
                          "note": "rt_tuning 25 = 0: the reference's two kernels back to back, what rt_raycast / rt_generate_candidate launch"}
        # PCIe-inclusive variant (never `value`): the reference copies the RGBA8 image to the host and
        # synchronises every frame (10_restir_di.cpp:386-389)
        r.timing_enable(False)
        n_pcie = max(5, K // 2)
        barrier()
        tp = time.perf_counter()
        for _ in range(n_pcie):
            frame += 1
            r.frame(frame)
            r.download(api.RT_BUF_PIXELS)
        pcie_ms = (time.perf_counter() - tp) / n_pcie * 1e3
        # BVH walks really performed: the headline counts the REFERENCE's raytrace() calls; the build
        # settles some shadow rays with a one-triangle test and skips rays whose answer an earlier kernel of the frame holds
        # or nobody can observe. Counted by the kernels themselves over 4 more frames (untimed; a few atomics per wavefront).
        r.walk_stats_enable(True)
        n_walk = 4
        for _ in range(n_walk):
            frame += 1
            r.frame(frame)
        ws = r.walk_stats()
        r.walk_stats_enable(False)
        # frames are pipelined: stage 0 of the frame after the last one has run too -> per-frame averages over the launches counted
        n_stage0 = ws["raycast"]["reference_rays"] / float(width * height)
        per = {k: {kk: (vv / (n_stage0 if k in ("raycast", "generate_candidate") else n_walk)) for kk, vv in c.items()} for k, c in ws.items()}
        walks = {"per_kernel": {k: {kk: int(round(vv)) for kk, vv in c.items()} for k, c in per.items()},
                 "reference_rays": int(round(sum(c["reference_rays"] for c in per.values()))),
                 "walked": int(round(sum(c["walked"] for c in per.values()))),
                 "settled_by_self_test": int(round(sum(c["self_test"] for c in per.values()))),
                 "not_evaluated": int(round(sum(c["not_evaluated"] for c in per.values())))}
        # the timed path against the reference's launch sequence: a second context renders the same frame
        # numbers one C-ABI call per reference kernel, synchronously (10_restir_di.cpp:270-379); the accumulation buffer and the
        # temporal history of the last frame must be identical, bit for bit. Outside every timed region.
        if os.environ.get("BENCH_VERIFY", "1") not in ("", "0"):
            wd.tick("verification against the kernel sequence", 600)
            acc_fast = r.download(api.RT_BUF_ACCUMULATION)
            hist_fast = r.download(api.RT_BUF_RES_TEMPORAL)
            seq = api.Renderer(width, height, device=local_rank)
            seq.set_scene(tris)
            seq.lookat(eye, center)
            seq.set_options(opt)
            for f in range(1, frame + 1):
                seq.frame_by_kernels(f)
            acc_seq = seq.download(api.RT_BUF_ACCUMULATION)
            hist_seq = seq.download(api.RT_BUF_RES_TEMPORAL)
            shaded_mask = hist_seq["M"] > 0
            verified_seq = bool(np.array_equal(acc_fast.view(np.uint32), acc_seq.view(np.uint32))
                                and all(np.array_equal(np.ascontiguousarray(hist_fast[f_][shaded_mask]).view(np.uint8),
                                                       np.ascontiguousarray(hist_seq[f_][shaded_mask]).view(np.uint8))
                                        for f_ in hist_seq.dtype.names if f_ != "pad"))
            seq.close()
    else:
        st = mg.stats()
        mine = dict(rank=rank, rows=list(R["bounds"][rank]), host_us_per_frame=round(st["host_ns"] / K / 1e3, 1),
                    plan_wait_us_per_frame=round(st["plan_wait_ns"] / K / 1e3, 1), cold_frames=st["cold_frames"],
                    MB_sent_per_frame=round(st["bytes_sent"] / K / 1e6, 3))
        allst = [None] * world
        dist.all_gather_object(allst, mine)
        mg_stats = allst
        # N > 1 on real GPUs: the assembled image is compared with a single context by default (the RCCL transport has
        # only ever run on one rank) — outside the timed region. BENCH_VERIFY=0 skips it; the MIRROR
        # development transport cannot be verified (a rank receives what it sent).
        want_verify = os.environ.get("BENCH_VERIFY", "0" if (dev_mirror and not dev_shm) else "1") not in ("", "0")
        if want_verify:
            # development aid: the assembled N-rank image of the last frame must equal, bit for bit,
            # what a single full-frame context renders for the same frame sequence
            a, b = R["bounds"][rank]
            part = r.download(api.RT_BUF_ACCUMULATION).reshape(r.local_rows, width, 4)[a - r.local_row0: b - r.local_row0].copy()
            parts = [None] * world
            dist.gather_object(part, parts if rank == 0 else None, dst=0)
            if rank == 0:
                full = api.Renderer(width, height, device=local_rank)
                full.set_scene(tris)
                full.lookat(eye, center)
                full.set_options(opt)
                for f in range(1, frame + 1):
                    full.frame(f)
                ref = full.download(api.RT_BUF_ACCUMULATION).reshape(height, width, 4)
                verified = bool(np.array_equal(np.concatenate(parts, axis=0).view(np.uint32), ref.view(np.uint32)))
                full.close()
        # The single-GPU frame of THIS node, measured in this very run on rank 0's GPU while the other ranks wait (outside the timed
        # region): the N = 1 line's `value` is the un-pipelined frame of SURVEY 8(d) and an N > 1 `value` is the strip driver's
        # pipelined throughput, so a ratio of the two lines flatters the strips by the pipelining gain (~11 %). The line therefore
        # carries both single-GPU frame times and the two ratios itself. BENCH_NO_SINGLE=1 skips it.
        if not os.environ.get("BENCH_NO_SINGLE") and (dev_shm or not dev_mirror):
            wd.tick("single-GPU reference frames on rank 0", 600)
            single = None
            if rank == 0:
                one = api.Renderer(width, height, device=local_rank)
                one.set_scene(tris)
                one.lookat(eye, center)
                one.set_options(opt)
                res1 = {}
                for name, t14, t17 in (("unpipelined", 0, 0), ("pipelined", -1, -1)):
                    one.tuning(14, t14)
                    one.tuning(17, t17)
                    for f in range(1, Wm + 3):
                        one.frame(f)
                    one.sync()
                    t1 = time.perf_counter()
                    for f in range(Wm + 3, Wm + 3 + max(K, 20)):
                        one.frame(f)
                    one.sync()
                    res1[name] = (time.perf_counter() - t1) / max(K, 20) * 1e3
                one.close()
                single = res1
            dist.barrier()
            single_ref = single
        else:
            single_ref = None
    lib_build_id = r.build_id()  # of the library this context runs
    if mg is not None:
        mg.close()
    r.close()

    # Secondary line, reported next to the headline and never instead of it: the 3840x2160 frame of BASELINE
    # config #5 on the same N GPUs (4x the pixels: a strip is then large enough to amortise the per-kernel
    # latency floor that bounds strong scaling of the 2.3 ms 1080p frame).
    also_4k = None
    if (width, height) == (W, H) and not os.environ.get("BENCH_NO_4K"):
        R4 = run(3840, 2160, 10, 3, pipelined=(world > 1))
        also_4k = {"workload": "same scene and options at 3840x2160 (the frame of BASELINE config #5)", "steps": 10, "warmup": 3,
                   "frames": "pipelined (native strip driver)" if world > 1 else "back to back on one stream, as the headline",
                   "ms_per_step": R4["dt"] / 10 * 1e3, "value": R4["rays"] * 10 / R4["dt"] / 1e6, "unit": "Mray/s",
                   "rays_per_frame": R4["rays"], "strips": [list(b) for b in R4["bounds"]] if world > 1 else None}
        if R4["mg"] is not None:
            R4["mg"].close()
        R4["r"].close()

    if rank == 0:
        ms = elapsed / K * 1e3
        scene_desc.update({"triangles": info["triangles"], "lights": info["lights"], "bvh_height": info["bvh_height"],
                           "sha256": scenes.scene_sha256(tris)[:16]})
        out = {
            "metric": "Mray/s", "value": total_rays * K / elapsed / 1e6, "unit": "Mray/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"10_restir_di blocks_restir {'stand-in' if scene_desc['stand_in'] else 'user OBJ'} {width}x{height} 1spp ReSTIR DI "
                            "(temporal+spatial reuse, 3 spatial passes, visibility reuse, unshadowed target)",
                "scene": scene_desc,
                "bvh_builder": R["builder"], "build_ms": round(R["build_ms"], 1),
                "rt_tuning_env": os.environ.get("RT_TUNING") or None,
                "rays_per_frame": total_rays,
                # strips (N > 1): every timed frame launches exactly one raycast, the one of the NEXT frame, on a second stream
                # beside this frame's passes (rt_tuning key 14; rt_sync at the end of the timed region waits for it)
                # every timed frame launches each kernel of a frame exactly once; stage 0 of frame f+1 (primary rays, candidates +
                # temporal merge: they depend on frame f only through the history frame f's stage 0 wrote) runs on a second
                # stream beside the spatial passes and resolve of frame f (rt_tuning key 14; the sync at the end of the
                # timed region waits for it). kernel_ms / gpu_event_median_ms below are of frames run back to back on one stream.
                "frame_pipeline": ("stage 0 (raycast, generate_candidate + temporal) of frame f+1 beside the spatial passes and resolve of frame f"
                                   "; resolve + tone mapping of frame f beside the first halo exchange of frame f+1; halo records read from / "
                                   "written to the exchange lists by the spatial passes") if world > 1 else
                                  "none in `value`: every kernel of frame f before any kernel of frame f+1, one stream (rt_tuning 14 = 0, 17 = 0); "
                                  "`value_pipelined` overlaps stage 0 of frame f+1 with the passes and resolve of frame f",
                "parallelism": (f"row strips x{world}, {R['part']}, sparse 87-row halos over "
                                + ("torch.distributed send/recv (Python schedule" + (", gloo on ONE GPU: not a scaling number)" if dev_mirror else ")")
                                   if "FALLBACK" in R["part"] else
                                   "the SHM transport on ONE GPU (BENCH_DEV_SHM: protocol check, not a scaling number) (native driver)" if dev_shm else
                                   "self-loopback on ONE GPU (BENCH_DEV_MIRROR: per-rank overhead, not a scaling number) (native driver)" if dev_mirror else
                                   "RCCL send/recv (native driver)")) if world > 1 else "single GPU",
            },
            # what the parity chain cannot pin to the reference: HIPRT's device code is a missing binary
            # (DESIGN.md section 2); everything else is bit-exact against the reference's own sources
            "parity_unpinned": ["raytrace()/HIPRT: intersection pinned by definition (brute force of common/core.hpp:91-136)"],
        }
        if dev_shm:
            out["dev_shm"] = "N ranks on ONE GPU with the host-staged SHM transport: exact images, not a multi-GPU measurement"
        elif dev_mirror:
            out["dev_mirror"] = "N ranks on ONE GPU with the MIRROR transport: script/driver smoke run, not a multi-GPU measurement"
        if world > 1:
            out["value_definition"] = ("throughput of the native strip driver's pipelined frames; the N = 1 line's `value` is un-pipelined "
                                       "frames (SURVEY 8d) and its `value_pipelined` is the figure this one is comparable with")
            if single_ref is not None:
                out["single_gpu_on_this_node"] = {
                    "ms_per_frame_unpipelined": single_ref["unpipelined"], "ms_per_frame_pipelined": single_ref["pipelined"],
                    "speedup_vs_unpipelined": single_ref["unpipelined"] / ms, "speedup_vs_pipelined": single_ref["pipelined"] / ms,
                    "note": "a whole-frame context on rank 0's GPU, timed in this run while the other ranks waited: the N = 1 line's `value` "
                            "is the un-pipelined frame, this line's `value` is pipelined throughput — compare like with like"}
            out["config"]["strips"] = [list(b) for b in R["bounds"]]
            out["strip_driver"] = mg_stats
            if verified is not None:
                out["verified_vs_single_context"] = verified
        if also_4k is not None:
            out["also_3840x2160"] = also_4k
        if world == 1:
            sha = lib_sha256()
            pmc = committed_pmc(sha, lib_build_id)
            contract_ach = algo_bytes / (spatial_ms * 1e-3) / 1e9
            traffic = pmc["traffic"]
            measured_ach = (traffic / (spatial_ms * 1e-3) / 1e9) if traffic else None
            out["roofline"] = {
                "kernel": "k_spatial_coop (spatial_resampling)",
                # r04: `achieved` / `frac` are what the HBM counters of THIS build say the kernel moves per
                # launch (FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes; tools/profile_round.sh)
                # over the launch duration measured live here with HIP events; null when no counter pass of this build is
                # committed (rt_build_id mismatch). The SURVEY 8(d) figure — algorithmic bytes in REFERENCE record sizes
                # (16 + 76 B per neighbour; the kernel gathers one 64-B record, mostly from L2) / time / peak — is
                # `contract_achieved` / `contract_frac`; it exceeds what the kernel moves and can pass 1.
                "bound": "valu" if measured_ach is not None else "hbm", "bound_contract": "hbm",
                "achieved": measured_ach if measured_ach is not None else contract_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (measured_ach if measured_ach is not None else contract_ach) / HBM_PEAK_GBS,
                "frac_kind": "HBM counter traffic / launch time / peak" if measured_ach is not None else "CONTRACT figure (no counter pass of this build is committed)",
                "traffic": traffic,
                "contract_achieved": contract_ach, "contract_frac": contract_ach / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": algo_bytes, "ms_per_launch": spatial_ms,
                "traffic_source": pmc["source"],
                "limiter": "the CU's gather path plus vector-ALU issue, not HBM bandwidth: tools/gather_ceiling.hip runs the pass's access shape without "
                           "the pass (profiles/r06_gather_ceiling.jsonl) — the gathers alone 0.057 ms, with the pass's streamed reads and stores 0.079, "
                           "with the pass's arithmetic as 352 vector FMAs per round 0.127-0.129 ms = this kernel, in every issue order incl. 2-3 rounds "
                           "in flight (DESIGN.md section 5)",
                "ceiling_microbenchmark_ms": {"gathers_only": 0.057, "gathers_and_streams": 0.079, "with_the_pass_arithmetic": 0.129,
                                              "source": "profiles/r06_gather_ceiling.jsonl (tools/gather_ceiling.hip, one MI355X box of this pool)"},
                "valu_issue_frac_weighted": pmc["valu"],
            }
            out["lib_sha256"] = sha
            out["build_id"] = lib_build_id
            out["kernel_ms"] = per_kernel
            if two_launch is not None:
                out["kernel_ms_two_launch_stage0"] = two_launch
            out["value_definition"] = ("un-pipelined frames of SURVEY 8(d): K frames back to back on one stream, wall clock; since round 5. Rounds 1-4 "
                                       "reported the pipelined throughput here: compare across rounds through `value_pipelined`")
            out["gpu_event_median_ms"] = event_median
            # SURVEY 8(d)'s frame time = GPU-event median of >= 50 un-overlapped frames
            out["value_gpu_event_median"] = {"value": total_rays / event_median / 1e3, "unit": "Mray/s", "ms_per_frame": event_median,
                                             "note": "rays / GPU-event median of frames run back to back on one stream with per-kernel events (SURVEY 8d's "
                                                     "definition of the frame time); `value` is the same frames without the events, wall clock over K frames"}
            out["value_pipelined"] = None if pipelined_dt is None else {
                "value": total_rays * K / pipelined_dt / 1e6, "unit": "Mray/s", "ms_per_step": pipelined_dt / K * 1e3, "steps": K, "warmup": Wm,
                "note": "throughput of pipelined frames (stage 0 of frame f+1 beside the passes of frame f; every kernel still runs "
                        "once per frame): the headline of rounds 1-4, and what the N > 1 lines are comparable with"}
            if walks is not None:
                out["bvh_walks_per_frame"] = walks
                out["Mwalk_per_s"] = walks["walked"] / ms / 1e3
                out["value_counts"] = "raytrace() calls of the REFERENCE per frame (SURVEY 8d); BVH traversals actually performed: bvh_walks_per_frame / Mwalk_per_s"
            if verified_seq is not None:
                out["verified_vs_kernel_sequence"] = verified_seq
            out["pcie_inclusive"] = {"ms_per_frame": pcie_ms, "value": total_rays / pcie_ms / 1e3, "unit": "Mray/s",
                                     "note": "frame + RGBA8 read-back to pageable host memory + sync, as the reference's loop does"}
            if not args.no_cpu_baseline and (width, height) == (W, H):
                cb, _ = cpu_baseline(tris, eye, center)
                out["cpu_baseline"] = cb
            else:
                out["cpu_baseline"] = None
        sys.stdout.flush()
        wd.stop()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    wd.stop()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
