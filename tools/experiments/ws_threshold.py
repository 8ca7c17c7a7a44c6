"""Where does the work-sharing shadow-ray walk (rt_tuning key 13) pay? generate + resolve times of one strip of
`rows` rows (config #4; each stage of the strip's frame timed on the host around a synchronised call, halos left
unfilled: the strip's own pixels cost the same) with the walk off / on, by strip height."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options

tris = scenes.make_blocks_restir()
out = {}
for W, H, hs in ((1920, 1080, (135, 270, 540, 1080)), (3840, 2160, (270, 540, 1080, 2160))):
    for h in hs:
        a = (H - h) // 2
        r = api.Renderer(W, H, rows=(a, a + h), halo=min(90, a))
        r.set_scene(tris); r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT); r.set_options(bench_options())
        def stage(f, k, clear=False):
            r.sync(); t0 = time.perf_counter(); r.frame_stage(f, k, clear); r.sync()
            return (time.perf_counter() - t0) * 1e3
        row = {}
        for ws in (0, 1):
            r.tuning(13, ws)
            acc = [0.0] * 5
            for f in range(1, 25):
                t = [stage(f, k, f == 1 and k == 0) for k in range(5)]
                if f > 4:
                    acc = [x + y for x, y in zip(acc, t)]
            n = 20
            row["ws%d" % ws] = dict(raycast_generate_ms=round(acc[0] / n, 4), resolve_tone_ms=round(acc[4] / n, 4), frame_ms=round(sum(acc) / n, 4))
        row["wavefronts"] = W * h // 64
        out["%dx%d rows %d" % (W, H, h)] = row
        r.close()
print(json.dumps(out, indent=1))
