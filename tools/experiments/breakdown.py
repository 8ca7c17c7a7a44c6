"""Dev tool: per-kernel times under option variations (what costs what)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options

W, H = 1920, 1080
tris = scenes.make_blocks_restir()
r = api.Renderer(W, H)
r.set_scene(tris)
r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
r.timing_enable(True)
variants = [("bench", {}), ("no_visreuse", dict(use_visibility_reuse=0)), ("ris8", dict(ris_sample_count=8)),
            ("ris1_novis", dict(ris_sample_count=1, use_visibility_reuse=0)), ("no_temporal", dict(use_temporal_resampling=0)),
            ("spatial1", dict(spatial_resampling_sample_count=1)), ("shadowed", dict(use_shadowed_target_function=1))]
for name, kw in variants:
    r.set_options(bench_options(**kw))
    acc = None
    for fr in range(1, 9):
        r.frame(fr)
        t = r.timing()
        if fr > 3:
            acc = {k: acc[k] + v for k, v in t.items()} if acc else dict(t)
    print(name, json.dumps({k: round(v / 5, 3) for k, v in acc.items()}), flush=True)
