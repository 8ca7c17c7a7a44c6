"""Dev tool: oracle (CPU baseline) frame time vs OpenMP thread count on this host."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cedec_2024_rt_amd import scenes
from oracle import binding as ob
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "omp max", ob.max_threads())
try:
    print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup cpu.max", e)
W, H = 1920, 1080
tris = scenes.make_blocks_restir()
ob.set_math_mode(ob.MATH_PORTABLE)
sc = ob.Scene(tris, use_bvh=True)
rg = ob.raygen_lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT, (0, 1, 0), np.float32(np.pi) / np.float32(4), W, H)
eye = np.asarray(scenes.BLOCKS_RESTIR_EYE, np.float32)
opt = ob.bench_options()
for th in [int(x) for x in sys.argv[1:]] or [8, 16, 32, 64, 128]:
    ob.set_threads(th)
    st = ob.new_state(W, H); cnt = ob.new_counters()
    sc.frame(W, H, 1, rg, eye, opt, st, None)
    t0 = time.perf_counter(); sc.frame(W, H, 2, rg, eye, opt, st, cnt); dt = time.perf_counter() - t0
    print(json.dumps(dict(threads=th, ms=round(dt * 1e3, 1), mray_s=round(int(cnt["rays"][0]) / dt / 1e6, 2))), flush=True)
