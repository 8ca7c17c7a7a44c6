"""Dev tool: BVH quality / frame time vs the pre-split factor on the benchmark scene."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options

W, H = 1920, 1080
tris = scenes.make_blocks_restir()
factors = [float(x) for x in sys.argv[1:]] or [0.0, 2.0, 4.0, 8.0]
import itertools
for builder, f in itertools.product((0, 1), factors):
    r = api.Renderer(W, H)
    r.bvh_config(f)
    r.tuning(5, builder)
    t0 = time.time(); r.set_scene(tris); tb = time.time() - t0
    r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
    r.set_options(bench_options())
    r.timing_enable(True)
    for fr in range(1, 4):
        r.frame(fr)
    t = r.timing()
    # primary-ray stats on a 240x135 subsample
    rg = r.raygen()[0]
    o = rg["origin"]; right = rg["right"]; up = rg["up"]
    fw = np.cross(up, right); fw /= np.linalg.norm(fw)
    xs, ys = np.meshgrid(np.arange(0, W, 8), np.arange(0, H, 8))
    u = (xs / W).astype(np.float32).ravel(); v = (ys / H).astype(np.float32).ravel()
    to = o + fw + (-right + 2 * right * u[:, None]) + (up - 2 * up * v[:, None])
    d = to - o; d /= np.linalg.norm(d, axis=1)[:, None]
    rays = np.zeros((len(u), 8), np.float32); rays[:, :3] = o; rays[:, 3:6] = d; rays[:, 7] = 3e38
    st = r.trace_stats(rays)
    print(json.dumps(dict(builder=builder, split=f, build_s=round(tb, 3), info=r.scene_info(), bvh=r.bvh_info(),
                          nodes_mean=float(st[:, 0].mean()), nodes_p99=float(np.percentile(st[:, 0], 99)),
                          tris_mean=float(st[:, 1].mean()), ms={k: round(x, 3) for k, x in t.items()})), flush=True)
    r.close()
