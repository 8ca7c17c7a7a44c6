"""GPU, timing only (results are garbage: the passes race on purpose). Upper bound of what overlapping the three spatial passes'
ramp-downs could gain: the same three launches back to back on one stream against on three streams with no dependency."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options
for W, H in ((1920, 1080), (3840, 2160)):
    r = api.Renderer(W, H)
    r.set_scene(scenes.make_blocks_restir()); r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT); r.set_options(bench_options())
    for f in range(1, 6):
        r.frame(f)
    r.sync()
    streams = [torch.cuda.Stream() for _ in range(3)]
    jobs = ((0, api.RT_RES_0, api.RT_RES_1), (1, api.RT_RES_0, api.RT_RES_TEMPORAL), (2, api.RT_RES_0, api.RT_RES_1))
    def run(parallel, n=40):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            for k, (p, a, b) in enumerate(jobs):
                r.set_stream(streams[k if parallel else 0].cuda_stream)
                r.spatial_resampling(6, p, a, b)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    run(False, 5); run(True, 5)
    for rep in range(3):
        print("%dx%d three passes: one stream %.4f ms, three streams (no dependencies) %.4f ms" % (W, H, run(False), run(True)), flush=True)
    r.set_stream_own(); r.close()
