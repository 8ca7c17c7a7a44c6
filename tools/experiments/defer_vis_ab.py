"""A/B of rt_tuning key 11 (visibility-reuse rays only for candidates that survive the temporal merge) on the
benchmark frame: per-kernel HIP-event medians over 60 frames, first frame (no history: every ray is walked) apart."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options
for W, H in ((1920, 1080), (3840, 2160)):
    tris = scenes.make_blocks_restir()
    for key11, key12 in ((0, 0), (0, 1), (1, 0), (1, 1)):
        r = api.Renderer(W, H)
        r.set_scene(tris); r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT); r.set_options(bench_options())
        r.tuning(11, key11)
        r.tuning(12, key12)
        r.timing_enable(True)
        r.frame(1); first = r.timing()
        for f in range(2, 6): r.frame(f)
        rows = []
        for f in range(6, 66):
            r.frame(f); t = r.timing(); rows.append([t["generate_candidate"], t["frame"]])
        rows = np.array(rows)
        if key11:
            print("visibility rays walked in the last frame:", r.visibility_rays_walked(), "of", r.ray_count()[1], "shaded pixels", flush=True)
        print(json.dumps({f"{W}x{H} defer={key11} pipe={key12}": dict(first_frame_generate_ms=round(first["generate_candidate"], 4), first_frame_ms=round(first["frame"], 4),
              generate_ms=round(float(np.median(rows[:, 0])), 4), frame_ms=round(float(np.median(rows[:, 1])), 4))}), flush=True)
        r.close()
