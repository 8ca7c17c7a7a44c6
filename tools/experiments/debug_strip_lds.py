import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cedec_2024_rt_amd import api, scenes
import test_mg_native as T
tris = scenes.make_quad_room()
for k8, k14 in ((1, 1), (0, 1), (1, 0)):
    bad = []
    for rep in range(10):
        for n, H in ((2, 240), (3, 300)):
            rig = T._Rig(api, tris, 96, H, n, (0.5, 2.5, 6.0), (0.0, 1.5, -1.0), {}, 0)
            for c in rig.ctxs: c.tuning(8, k8); c.tuning(14, k14)
            for f in (1, 2, 3, 4, 5, 9):
                clear = False
                if f == 4:
                    for r in rig.everyone(): r.orbit(35.0, -12.0); r.camera_updated()
                    clear = True
                rig.frame(f, clear)
                ref = rig.full.download(api.RT_BUF_ACCUMULATION).reshape(H, 96, 4)
                for c, (a, b) in zip(rig.ctxs, rig.bounds):
                    acc = c.download(api.RT_BUF_ACCUMULATION).reshape(c.local_rows, 96, 4)[a - c.local_row0: b - c.local_row0]
                    d = (acc.view(np.uint32) != ref[a:b].view(np.uint32)).any(axis=2)
                    if d.any():
                        rows = np.nonzero(d.any(axis=1))[0]
                        bad.append((rep, n, f, a, int(d.sum()), int(rows.min() + a), int(rows.max() + a)))
            rig.close()
    print("lds", k8, "nextraycast", k14, "failures", bad)
