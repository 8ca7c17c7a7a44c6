"""How often does resolve's shadow ray repeat a ray whose answer the reservoir already carries? Fraction of shaded pixels
whose FINAL reservoir sample (what resolve shades) originates from the pixel itself (origin_position / origin_normal
bit-equal to the pixel's surface point), config #4 at 1080p, frame 10 of a static sequence."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cedec_2024_rt_amd import api, scenes
from cedec_2024_rt_amd.types import bench_options
W, H = 1920, 1080
r = api.Renderer(W, H)
r.set_scene(scenes.make_blocks_restir()); r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT); r.set_options(bench_options())
r.frame(1)
own = r.download(api.RT_BUF_RES_TEMPORAL)  # frame 1 has no history: every shaded pixel holds its own candidate
sp, sn, m1 = own["origin_position"].copy(), own["origin_normal"].copy(), own["M"] > 0
for f in range(2, 11):
    final = r.frame(f)
fin = r.download(api.RT_BUF_RES_0 + final)
tmp = r.download(api.RT_BUF_RES_TEMPORAL)
same = lambda a: (a["origin_position"].view(np.uint32) == sp.view(np.uint32)).all(axis=-1) & (a["origin_normal"].view(np.uint32) == sn.view(np.uint32)).all(axis=-1)
sh = m1
print("shaded pixels %d; final sample is the pixel's own: %.1f %%; post-temporal sample is the pixel's own: %.1f %%; own AND stored visibility bit set: %.1f %%" % (
    sh.sum(), 100 * same(fin)[sh].mean(), 100 * same(tmp)[sh].mean(), 100 * (same(fin) & (fin["visibility"] != 0))[sh].mean()))
