"""Dev tool: one middle strip of the 8-way partition alone on the GPU (transport that answers instantly): cProfile of
the host loop; run it under rocprofv3 --kernel-trace for the per-kernel times of a strip."""
import sys, os, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cedec_2024_rt_amd import api, scenes, strips
from cedec_2024_rt_amd.types import bench_options
class NullTransport:
    def post(self, rank, items): return None
    def finish(self, rank, handle, items):
        for _, _, tr in items: tr.zero_()
W, H, N = 1920, 1080, 8
tris = scenes.make_blocks_restir()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
bounds = strips.partition_rows(H, N)
a, b = bounds[4]
r = api.Renderer(W, H, device=0, rows=(a, b), halo=strips.HALO_ROWS, stream=stream)
r.set_scene(tris); r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT); r.set_options(bench_options())
f = strips.StripFrame(strips.HipStripBackend(r, dev), bounds, 4, NullTransport(), sparse=True)
for fr in range(1, 6): f.frame(fr)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for fr in range(6, 46): f.frame(fr)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
