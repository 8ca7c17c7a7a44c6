"""How much of a strip's frame is LATENCY (a chain of small launches and exchanges) rather than GPU work: two independent strip
contexts (two ranks of the same 8-strip partition) driven alternately from one host thread on ONE GPU, against each of them
alone. If the pair costs little more than one alone, the chains overlap: the bound on what keeping two frames of ONE strip in
flight (frame f+1's passes beside frame f's) can gain.   python tools/two_chains.py [mirror|rccl_self]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes  # noqa: E402
from cedec_2024_rt_amd.types import bench_options  # noqa: E402

T = {"mirror": api.RT_MG_TRANSPORT_MIRROR, "rccl_self": api.RT_MG_TRANSPORT_RCCL_SELF}[sys.argv[1] if len(sys.argv) > 1 else "mirror"]
tris = scenes.make_blocks_restir()


def rig(W, H, rank, N=8):
    bounds = api.mg_partition(H, N)
    r = api.Renderer(W, H, rows=bounds[rank], halo=87)
    r.set_scene(tris)
    r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
    r.set_options(bench_options())
    return r, api.MultiGpu(r, rank, bounds, transport=T)


def run(rigs, frames=40, warm=6):
    f = 0
    for _ in range(warm):
        f += 1
        for _, mg in rigs:
            mg.frame(f)
    for r, _ in rigs:
        r.sync()
    t0 = time.perf_counter()
    for _ in range(frames):
        f += 1
        for _, mg in rigs:
            mg.frame(f)
    for r, _ in rigs:
        r.sync()
    return (time.perf_counter() - t0) / frames * 1e3


out = {}
for (W, H) in ((1920, 1080), (3840, 2160)):
    a, b = rig(W, H, 3), rig(W, H, 4)
    one_a, one_b = run([a]), run([b])
    pair = run([a, b])
    out[f"{W}x{H}"] = dict(rank3_alone_ms=round(one_a, 4), rank4_alone_ms=round(one_b, 4), both_alternating_ms=round(pair, 4),
                           per_strip_frame_ms_when_paired=round(pair / 2, 4))
    for r, mg in (a, b):
        mg.close()
        r.close()
print(json.dumps(out))
