"""GPU. Where k_generate_candidate's time goes: the kernel timed (HIP events, rt_timing) on the bench workload with parts of its
work switched off through the reference's own options — ris_sample_count (the RIS loop: 32 candidates), use_visibility_reuse (the
shadow ray), use_temporal_resampling (the fused temporal merge). Not a parity run: the images differ by construction.
  python tools/generate_breakdown.py [WxH]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes  # noqa: E402
from cedec_2024_rt_amd.types import bench_options  # noqa: E402

W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1920x1080").split("x"))
tris = scenes.make_blocks_restir()
names = ("clear", "raycast", "generate_candidate", "spatial0", "spatial1", "spatial2", "resolve", "tone_mapping", "frame")
for label, kw in (("full", {}), ("ris 16", dict(ris_sample_count=16)), ("ris 1", dict(ris_sample_count=1)),
                  ("no visibility reuse", dict(use_visibility_reuse=0)), ("no temporal", dict(use_temporal_resampling=0)),
                  ("ris 1, no visibility reuse", dict(ris_sample_count=1, use_visibility_reuse=0)),
                  ("ris 1, no visibility reuse, no temporal", dict(ris_sample_count=1, use_visibility_reuse=0, use_temporal_resampling=0))):
    r = api.Renderer(W, H)
    r.set_scene(tris)
    r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
    r.set_options(bench_options(**kw))
    r.timing_enable(True)
    rows = []
    for f in range(1, 41):
        r.frame(f)
        t = r.timing()
        if f > 8:
            rows.append([t[k] for k in names])
    m = np.array(rows).mean(axis=0)
    print("%-42s generate %.4f ms  raycast %.4f  spatial %.4f  resolve %.4f  frame %.4f" % (label, m[2], m[1], m[3:6].mean(), m[6], m[8]), flush=True)
    r.close()
