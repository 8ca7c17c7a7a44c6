"""Timings of BASELINE.json configs #1-#3 and #4 at 4K (results table of BASELINE.md / DESIGN.md).
  python tools/config_table.py > profiles/rNN_config_table.json"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options, default_options
from oracle import binding as ob

g = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
out = {}
# config #1: 04_ao cornellbox1 256x256 on the host (oracle = CPU restatement of examples/04_ao/04_ao.cu)
ob.set_math_mode(ob.MATH_LIBM)
sc = ob.Scene(g["cornellbox1"], use_bvh=False)
rg = ob.raygen_lookat(scenes.DEFAULT_EYE, scenes.DEFAULT_LOOKAT, (0, 1, 0), np.float32(np.pi) / np.float32(4), 256, 256)
sc.ao_04(32, 32, ob.raygen_lookat(scenes.DEFAULT_EYE, scenes.DEFAULT_LOOKAT, (0, 1, 0), np.float32(np.pi) / np.float32(4), 32, 32))
t0 = time.perf_counter(); px = sc.ao_04(256, 256, rg); dt = time.perf_counter() - t0
hit = int((px[..., 0] != 32).sum()); rays = 256 * 256 + 64 * hit
out["config1_04_ao_cpu"] = dict(ms=dt * 1e3, rays=rays, mray_s=rays / dt / 1e6, threads=ob.max_threads(), brute_force_tris=36)
ob.set_math_mode(ob.MATH_PORTABLE)

def timed(fn, n):
    fn(); r.sync()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    r.sync()
    return (time.perf_counter() - t0) / n * 1e3

# config #2: 07_pt cornellbox2 512x512, 4 spp (frames 1..4 accumulated)
r = api.Renderer(512, 512); r.set_scene(g["cornellbox2"]); r.lookat(scenes.CORNELLBOX_EYE, scenes.CORNELLBOX_LOOKAT)
r.set_options(default_options(accumulate=1)); r.clear()
fr = [0]
def pt7():
    fr[0] += 1; r.path_trace(7, fr[0])
ms = timed(pt7, 8); rays = r.path_trace_rays()
out["config2_07_pt"] = dict(ms_per_spp=ms, ms_4spp=4 * ms, rays_per_spp=rays, mray_s=rays / ms / 1e3)
r.close()
# config #3: 09_ris 1280x720 — on the scene its text names (blocks_pt stand-in, camera of 07_pt.cpp:139-140) and on
# the scene 09_ris.cpp:149 actually loads (blocks_restir stand-in)
tris = scenes.make_blocks_restir()
tris_pt = scenes.make_blocks_pt()
for key, tt, eye, at in (("config3_09_ris_blocks_pt", tris_pt, scenes.BLOCKS_PT_EYE, scenes.BLOCKS_PT_LOOKAT),
                         ("config3_09_ris_blocks_restir", tris, scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)):
    r = api.Renderer(1280, 720); r.set_scene(tt); r.lookat(eye, at)
    r.set_options(default_options())
    fr = [0]
    def pt9():
        fr[0] += 1; r.path_trace(9, fr[0])
    ms = timed(pt9, 8); rays = r.path_trace_rays()
    out[key] = dict(ms_per_frame=ms, rays=rays, mray_s=rays / ms / 1e3)
    r.close()
# 07_pt / 08_nee / 09_ris at 1080p (the reference's img/7-9.png settings): blocks_pt stand-in
# with the 07_pt camera, and the blocks_restir stand-in for comparison with round 1
for key, tt, eye, at in (("blocks_pt", tris_pt, scenes.BLOCKS_PT_EYE, scenes.BLOCKS_PT_LOOKAT), ("blocks_restir", tris, scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)):
    r = api.Renderer(1920, 1080); r.set_scene(tt); r.lookat(eye, at)
    r.set_options(default_options())
    for ex in (7, 8, 9):
        fr = [0]
        def pt():
            fr[0] += 1; r.path_trace(ex, fr[0])
        ms = timed(pt, 8); rays = r.path_trace_rays()
        out["example_%02d_1080p_%s" % (ex, key)] = dict(ms_per_frame=ms, rays=rays, mray_s=rays / ms / 1e3)
    r.close()
# config #4 at 3840x2160 on ONE GPU (the 8-GPU config #5 is the driver's to run)
r = api.Renderer(3840, 2160); r.set_scene(tris); r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
r.set_options(bench_options())
fr = [0]
def f4k():
    fr[0] += 1; r.frame(fr[0])
for _ in range(3): f4k()
ms = timed(f4k, 20); rays, shaded = r.ray_count()
out["config4_at_4k_1gpu"] = dict(ms_per_frame=ms, rays=rays, mray_s=rays / ms / 1e3)
r.close()
print(json.dumps(out, indent=1))
