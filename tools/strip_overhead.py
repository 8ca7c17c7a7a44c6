"""Dev tool: what one rank's frame costs in the N-strip decomposition when nothing has to wait for a
neighbour (transport = drop sends, zero-fill receives): host orchestration + the extra launches of
the halo machinery + the strip's kernels. N strip contexts live in this process; each rank's frame
loop runs to completion one after the other, so wall/N = per-rank frame time on an otherwise idle GPU."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cedec_2024_rt_amd import api, scenes, strips
from cedec_2024_rt_amd.types import bench_options

class NullTransport:
    def post(self, rank, items): return None
    def finish(self, rank, handle, items):
        for _, _, tr in items: tr.zero_()

W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
tris = scenes.make_blocks_restir()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
out = {}
for N in (1, 2, 4, 8):
    bounds = strips.partition_rows(H, N)
    for sparse in ((False,) if N == 1 else (False, True)):
        rs, fs = [], []
        for k, (a, b) in enumerate(bounds):
            r = api.Renderer(W, H, device=0, rows=(a, b), halo=strips.HALO_ROWS if N > 1 else 0, stream=stream)
            r.set_scene(tris); r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT); r.set_options(bench_options())
            rs.append(r)
            fs.append(strips.StripFrame(strips.HipStripBackend(r, dev), bounds, k, NullTransport() if N > 1 else None, sparse=sparse))
        def run(frames):
            for fr in frames:
                for f in fs: f.frame(fr)
            torch.cuda.synchronize()
        run(range(1, 4))
        t0 = time.perf_counter(); run(range(4, 24)); wall = (time.perf_counter() - t0) / 20 * 1e3
        # one rank alone on the GPU (what a rank of a real N-GPU job sees), and the host time of its loop
        mid = fs[len(fs) // 2]
        t0 = time.perf_counter()
        for fr in range(24, 44): mid.frame(fr)
        host = (time.perf_counter() - t0) / 20 * 1e3
        torch.cuda.synchronize()
        alone = (time.perf_counter() - t0) / 20 * 1e3
        out["N=%d %s" % (N, "sparse" if sparse else "dense")] = dict(ms_all_ranks=round(wall, 3), ms_per_rank=round(wall / N, 3),
                                                                      ms_middle_rank_alone=round(alone, 3), host_ms_middle_rank=round(host, 3))
        print(json.dumps({k: v for k, v in out.items() if k.startswith("N=%d" % N)}), flush=True)
        for r in rs: r.close()
