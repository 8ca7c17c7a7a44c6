"""Dev tool: distribution of traversal work for primary rays + wall time of the trace utility."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cedec_2024_rt_amd import api, scenes

W, H = 1920, 1080
tris = scenes.make_blocks_restir()
r = api.Renderer(W, H)
r.bvh_config(float(sys.argv[1]) if len(sys.argv) > 1 else 0.0)
r.set_scene(tris)
r.lookat(scenes.BLOCKS_RESTIR_EYE, scenes.BLOCKS_RESTIR_LOOKAT)
r.trace_mode(int(os.environ.get("TRACE_MODE", "0")))
rg = r.raygen()[0]
o = rg["origin"]; right = rg["right"]; up = rg["up"]
fw = np.cross(up, right); fw /= np.linalg.norm(fw)
xs, ys = np.meshgrid(np.arange(0, W, 1), np.arange(0, H, 1))
u = (xs / W).astype(np.float32).ravel(); v = (ys / H).astype(np.float32).ravel()
to = o + fw + (-right + 2 * right * u[:, None]) + (up - 2 * up * v[:, None])
d = to - o; d /= np.linalg.norm(d, axis=1)[:, None]
rays = np.zeros((len(u), 8), np.float32); rays[:, :3] = o; rays[:, 3:6] = d; rays[:, 7] = 3e38
st = r.trace_stats(rays)
n = st[:, 0].astype(np.int64)
print("nodes: mean %.1f p50 %d p99 %d p99.9 %d max %d sum %d" % (n.mean(), np.percentile(n, 50), np.percentile(n, 99), np.percentile(n, 99.9), n.max(), n.sum()))
t = st[:, 1].astype(np.int64)
print("tris: mean %.2f p99 %d max %d" % (t.mean(), np.percentile(t, 99), t.max()))
# per-wave max (64 consecutive rays)
wm = n[: len(n) // 64 * 64].reshape(-1, 64).max(axis=1)
print("per-wave max nodes: mean %.1f max %d ; sum of wave maxima %d" % (wm.mean(), wm.max(), wm.sum()))
for k in range(3):
    t0 = time.time(); h = r.trace_closest(rays); print("trace_closest wall %.1f ms" % ((time.time() - t0) * 1e3))
# shuffled rays (incoherent)
perm = np.random.default_rng(0).permutation(len(rays))
t0 = time.time(); h2 = r.trace_closest(rays[perm]); print("shuffled wall %.1f ms" % ((time.time() - t0) * 1e3))
idx = np.argsort(-n)[:5]
print("worst rays:", [(int(i % W), int(i // W), int(n[i]), int(t[i])) for i in idx])
