"""Timings of BASELINE.json configs #1-#3 and #4 at 4K (results table of BASELINE.md / DESIGN.md).
  python tools/config_table.py > profiles/rNN_config_table.json"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options, default_options
import re, subprocess

g = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
out = {}
# config #1: 04_ao cornellbox1 256x256 as the product's host loop (restir_app --example 4, csrc/host_path.h; no GPU call)
p = subprocess.run([os.path.join(ROOT, "app", "restir_app"), "--example", "4", "--obj", os.path.join(ROOT, "tests", "golden", "assets", "cornellbox1.obj")],
                   capture_output=True, text=True)
m = re.search(r"04_ao 256x256: ([0-9.]+) ms on (\d+) host thread\(s\), (\d+) of", p.stdout)
ms1, threads, hit = float(m.group(1)), int(m.group(2)), int(m.group(3))
rays = 256 * 256 + 64 * hit
out["config1_04_ao_host_loop"] = dict(ms=ms1, rays=rays, mray_s=rays / ms1 / 1e3, threads=threads, brute_force_tris=36)

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
# README key 3 (use_shadowed_target_function) on the benchmark frame: frames pipelined as bench.py runs them
r = api.Renderer(1920, 1080); r.set_scene(tris); r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
r.set_options(bench_options(use_shadowed_target_function=1))
fr = [0]
def fsh():
    fr[0] += 1; r.frame(fr[0])
for _ in range(3): fsh()
ms = timed(fsh, 20); rays, shaded = r.ray_count()
out["config4_shadowed_target_1080p"] = dict(ms_per_frame=ms, reference_rays=rays, mray_s_reference_rays=rays / ms / 1e3)
r.close()
print(json.dumps(out, indent=1))
