"""Every rank of an N-strip frame, one at a time, alone on the GPU (MIRROR transport, as tools/strip_overhead.py): the frame rate of
the N-GPU job is that of its SLOWEST rank, so the compute-side bound on the speed-up is single-GPU ms / max over ranks.
Partitions: equal rows, bench.py's cost model (rows weighted by their shaded pixels), and measured costs (each strip's own
time per row, fed back into rt_mg_partition for a few rounds - what bench.py does at start-up when BENCH_MEASURED_STRIPS is set).

  python tools/strip_balance.py [WxH] [N]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

from cedec_2024_rt_amd import api, scenes  # noqa: E402
from cedec_2024_rt_amd.types import bench_options  # noqa: E402
import strip_overhead  # noqa: E402

W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3840x2160").split("x"))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tris = scenes.make_blocks_restir()


def measure_all(bounds, frames=30):
    out = []
    for rank in range(N):
        m = strip_overhead.measure(W, H, N, 0, tris, frames=frames, warm=6, rank=rank, bounds=bounds)
        out.append(m["ms_per_frame"])
    return out


single = strip_overhead.measure(W, H, 1, 0, tris)["ms_per_frame"]
r = api.Renderer(W, H)
r.set_scene(tris)
r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
r.set_options(bench_options())
r.raycast()
shaded = r.row_shaded().astype(np.int64)
r.close()
res = {"single_ms": single}
equal = api.mg_partition(H, N, 87)
for name, bounds in (("equal rows", equal),) if os.environ.get("BALANCE_ONLY") else (("equal rows", equal),
                     ("shaded-pixel model 1:7 (bench.py until r03_e)", api.mg_partition(H, N, 87, (shaded * 7 + W).astype(np.uint32))),
                     ("shaded-pixel model 2:9", api.mg_partition(H, N, 87, (shaded * 9 + 2 * W).astype(np.uint32)))):
    t = measure_all(bounds)
    res[name] = dict(rows=[b - a for a, b in bounds], ms=t, max_ms=max(t), bound=round(single / max(t), 2))
    print(json.dumps({name: res[name]}), flush=True)
# measured costs: piecewise-constant cost per row from each strip's own time; the estimates of all rounds so far are averaged per row
bounds, t = equal, res["equal rows"]["ms"]
est = []
for it in range(5):
    cost = np.zeros(H)
    for (a, b), ms in zip(bounds, t):
        cost[a:b] = ms / (b - a)
    est.append(cost)
    cost = np.mean(est, axis=0)
    bounds = api.mg_partition(H, N, 87, np.maximum(1, cost / cost.max() * 60000).astype(np.uint32))
    t = measure_all(bounds, frames=40)
    name = f"measured costs (averaged over rounds), round {it + 1}"
    res[name] = dict(rows=[b - a for a, b in bounds], ms=t, max_ms=max(t), bound=round(single / max(t), 2))
    print(json.dumps({name: res[name]}), flush=True)
