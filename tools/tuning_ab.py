"""A/B of one rt_tuning key on rt_frame (config #4, whole frame on one GPU): throughput (wall clock over 60 frames, one
sync at the end) and the HIP-event time per kernel.   python tools/tuning_ab.py KEY V0 V1 [...]"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options

key, values = int(sys.argv[1]), [int(v) for v in sys.argv[2:]]
tris = scenes.make_blocks_restir()
out = {}
for W, H in ((1920, 1080), (3840, 2160)):
    r = api.Renderer(W, H, exp=True)  # the A/B forms live in librestir_rt_exp.so
    r.set_scene(tris); r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT); r.set_options(bench_options())
    for v in values * 2:
        r.tuning(key, v)
        r.timing_enable(False)
        for f in range(1, 6):
            r.frame(f, clear_first=(f == 1))
        r.sync()
        n = 60 if W < 3000 else 20
        t0 = time.perf_counter()
        for f in range(6, 6 + n):
            r.frame(f)
        r.sync()
        wall = (time.perf_counter() - t0) / n * 1e3
        r.timing_enable(True)
        acc = None
        for f in range(6 + n, 26 + n):
            r.frame(f); r.sync()
            t = r.timing()
            acc = t if acc is None else {k: acc[k] + t[k] for k in t}
        out.setdefault("%dx%d key%d=%d" % (W, H, key, v), []).append(dict(ms_per_frame_wall=round(wall, 4), **{k: round(x / 20, 4) for k, x in acc.items()}))
    r.close()
for k, v in out.items():
    for e in v:
        print(k, json.dumps(e))
