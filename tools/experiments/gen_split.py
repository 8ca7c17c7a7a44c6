"""Dev tool: what the fused candidate kernel spends its time on (option ablations)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options
W, H = 1920, 1080
r = api.Renderer(W, H)
r.set_scene(scenes.make_blocks_restir())
r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
r.timing_enable(True)
for name, kw in (("default", {}), ("no vis reuse", dict(use_visibility_reuse=0)), ("no temporal", dict(use_temporal_resampling=0)),
                 ("neither", dict(use_visibility_reuse=0, use_temporal_resampling=0)), ("ris 16", dict(ris_sample_count=16)),
                 ("ris 1, neither", dict(ris_sample_count=1, use_visibility_reuse=0, use_temporal_resampling=0))):
    o = bench_options()
    for k, v in kw.items(): o[k] = v
    r.set_options(o); r.clear()
    acc = None
    for fr in range(1, 14):
        r.frame(fr); t = r.timing()
        if fr > 3: acc = {k: acc[k] + v for k, v in t.items()} if acc else dict(t)
    print("%-16s" % name, json.dumps({k: round(v / 10, 4) for k, v in acc.items() if k in ("generate_candidate", "resolve", "frame")}), flush=True)
