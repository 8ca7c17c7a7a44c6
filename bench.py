#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: Mray/s (+ ms/frame) of the ReSTIR DI frame.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one frame = the timed region of the reference (examples/10_restir_di/10_restir_di.cpp:
254-383): raycast, generate_candidate(+temporal_resampling), 3 x spatial_resampling, resolve,
tone_mapping on the synthetic `blocks_restir` stand-in scene at 1920x1080, 1 spp, benchmark
options of SURVEY.md §8(d) (temporal + spatial reuse on, static camera). Inputs are resident in
HBM before the timed region. Rays are counted as BASELINE.md §3 defines (one raytrace() call =
one ray): N primary + 2 per shaded pixel.

N > 1: the frame is cut into N row strips (strong scaling: total work fixed), one process per
GPU, 87-row reservoir halos exchanged with RCCL send/recv before each spatial pass.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     — spatial_resampling: SURVEY §8(d) algorithmic bytes per launch / HIP-event time
  cpu_baseline — the oracle (CPU restatement) timed on this host's cores, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 1920, 1080
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--width", type=int, default=W)
    ap.add_argument("--height", type=int, default=H)
    args = ap.parse_args()

    import numpy as np
    import torch

    from cedec_2024_rt_amd import api, scenes, strips
    from cedec_2024_rt_amd.types import bench_options

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if os.environ.get("BENCH_SHARE_GPU0"):  # test aid: several ranks on one GPU (1-GPU dev box)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        # RCCL over xGMI; BENCH_BACKEND=gloo (host-staged halos) only exists to exercise the N>1
        # path on a single-GPU development box, where RCCL rejects two ranks on one device
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    width, height = args.width, args.height
    tris = scenes.make_blocks_restir()
    eye, center = scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT
    opt = bench_options()

    K, Wm = args.steps, args.warmup

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    spatial_ms = None
    verified = None
    pcie_ms = None
    per_kernel = None
    algo_bytes = None
    if world == 1:
        r = api.Renderer(width, height, device=local_rank)
        r.set_scene(tris)
        r.lookat(eye, center)
        r.set_options(opt)
        frame = 0
        for _ in range(Wm):
            frame += 1
            r.frame(frame)
        r.sync()
        rays_per_frame, shaded = r.ray_count()
        # algorithmic bytes of the three spatial launches of one timed frame (RNG replay, untimed):
        # pass k of frame f reads the buffer the previous pass wrote; the count only depends on
        # the shaded bits, which every reservoir buffer carries identically.
        fmid = Wm + 1 + K // 2
        algo = [r.spatial_bytes(fmid, k, api.RT_RES_0)[0] for k in range(3)]
        algo_bytes = float(np.mean(algo))
        r.timing_enable(True)
        acc_ms = np.zeros(9)
        barrier()
        t0 = time.perf_counter()
        for _ in range(K):
            frame += 1
            r.frame(frame)
        barrier()
        dt = time.perf_counter() - t0
        # per-kernel HIP-event times of the last timed frame + a separate event-timed replay of K
        # frames for the per-launch average of the roofline kernel
        r.timing_enable(True)
        for _ in range(K):
            frame += 1
            r.frame(frame)
            t = r.timing()
            acc_ms += np.array([t[k] for k in ("clear", "raycast", "generate_candidate", "spatial0", "spatial1",
                                                "spatial2", "resolve", "tone_mapping", "frame")])
        acc_ms /= K
        per_kernel = dict(zip(("clear", "raycast", "generate_candidate", "spatial0", "spatial1", "spatial2",
                               "resolve", "tone_mapping", "frame"), (round(float(x), 4) for x in acc_ms)))
        spatial_ms = float(acc_ms[3:6].mean())
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
        total_rays = rays_per_frame
        info = r.scene_info()
        elapsed = dt
    else:
        torch.cuda.synchronize()
        r, sf = strips.make_hip_strip(width, height, rank, world, tris, eye, center, opt, device_index=local_rank)
        frame = 0
        for _ in range(Wm):
            frame += 1
            sf.frame(frame)
        torch.cuda.synchronize()
        my_rays, _ = r.ray_count()
        barrier()
        t0 = time.perf_counter()
        for _ in range(K):
            frame += 1
            sf.frame(frame)
        barrier()
        dt = time.perf_counter() - t0
        cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        rr = torch.tensor([my_rays], dtype=torch.int64, device=cdev)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        total_rays = int(rr.item())
        info = r.scene_info()
        verified = None
        if os.environ.get("BENCH_VERIFY"):
            # development aid: the assembled N-rank image of the last frame must equal, bit for bit,
            # what a single full-frame context renders for the same frame sequence
            a, b = strips.partition_rows(height, world)[rank]
            mine = r.download(api.RT_BUF_ACCUMULATION).reshape(r.local_rows, width, 4)[a - r.local_row0: b - r.local_row0].copy()
            parts = [None] * world
            dist.gather_object(mine, parts if rank == 0 else None, dst=0)
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

    # Secondary line, reported next to the headline and never instead of it: the 3840x2160 frame of BASELINE
    # config #5 on the same N GPUs (4x the pixels: a strip is then large enough to amortise the per-kernel
    # latency floor that bounds strong scaling of the 2.3 ms 1080p frame).
    also_4k = None
    if (width, height) == (W, H) and not os.environ.get("BENCH_NO_4K"):
        r.close()
        w4, h4, k4, wm4 = 3840, 2160, 10, 3
        if world == 1:
            r4 = api.Renderer(w4, h4, device=local_rank)
            r4.set_scene(tris)
            r4.lookat(eye, center)
            r4.set_options(opt)
            step4 = r4.frame
        else:
            r4, sf4 = strips.make_hip_strip(w4, h4, rank, world, tris, eye, center, opt, device_index=local_rank)
            step4 = sf4.frame
        f4 = 0
        for _ in range(wm4):
            f4 += 1
            step4(f4)
        torch.cuda.synchronize()
        rays4, _ = r4.ray_count()
        barrier()
        t4 = time.perf_counter()
        for _ in range(k4):
            f4 += 1
            step4(f4)
        barrier()
        dt4 = time.perf_counter() - t4
        if world > 1:
            cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
            t = torch.tensor([dt4], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt4 = float(t.item())
            rr = torch.tensor([rays4], dtype=torch.int64, device=cdev)
            dist.all_reduce(rr, op=dist.ReduceOp.SUM)
            rays4 = int(rr.item())
        also_4k = {"workload": "same scene and options at 3840x2160 (the frame of BASELINE config #5)", "steps": k4, "warmup": wm4,
                   "ms_per_step": dt4 / k4 * 1e3, "value": rays4 * k4 / dt4 / 1e6, "unit": "Mray/s", "rays_per_frame": rays4}
        r4.close()

    if rank == 0:
        ms = elapsed / K * 1e3
        out = {
            "metric": "Mray/s", "value": total_rays * K / elapsed / 1e6, "unit": "Mray/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"10_restir_di blocks_restir stand-in {width}x{height} 1spp ReSTIR DI "
                            "(temporal+spatial reuse, 3 spatial passes, visibility reuse, unshadowed target)",
                "scene": {"triangles": info["triangles"], "lights": info["lights"], "bvh_height": info["bvh_height"],
                          "sha256": scenes.scene_sha256(tris)[:16], "generator": "scenes.make_blocks_restir (seed 2024)"},
                "rays_per_frame": total_rays, "parallelism": f"row-strips x{world}" if world > 1 else "single GPU",
            },
        }
        if world > 1 and verified is not None:
            out["verified_vs_single_context"] = verified
        if also_4k is not None:
            out["also_3840x2160"] = also_4k
        if world == 1:
            ach = algo_bytes / (spatial_ms * 1e-3) / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "spatial_pmc_latest.json")
            if os.path.exists(pmc):
                try:
                    with open(pmc) as f:
                        traffic = json.load(f).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            out["roofline"] = {"kernel": "k_spatial (spatial_resampling)", "bound": "hbm", "achieved": ach,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                               "algorithmic_bytes_per_launch": algo_bytes, "ms_per_launch": spatial_ms}
            out["kernel_ms"] = per_kernel
            out["pcie_inclusive"] = {"ms_per_frame": pcie_ms, "value": total_rays / pcie_ms / 1e3, "unit": "Mray/s",
                                     "note": "frame + RGBA8 read-back to pageable host memory + sync, as the reference's loop does"}
            if not args.no_cpu_baseline and (width, height) == (W, H):
                cb, _ = cpu_baseline(tris, eye, center)
                out["cpu_baseline"] = cb
            else:
                out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
