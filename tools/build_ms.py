"""GPU. What `build_ms` of the bench line is (27.7 ms in the driver's line, "11 ms" in the docs): wall time of
rt_scene_set for the bench scene — upload, light tables, BVH build, synchronised — for the FIRST call of a process (code objects of the
build kernels load, scratch buffers are allocated) and for repeated calls on the same context and on a fresh one.

  python tools/build_ms.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from cedec_2024_rt_amd import api, scenes

    tris = scenes.make_blocks_restir()
    out = {"scene": "blocks_restir stand-in, %d triangles" % len(tris), "builder": None, "ms": {}}
    r = api.Renderer(1920, 1080)
    out["builder"] = r.bvh_builder()
    calls = []
    for _ in range(5):
        r.set_scene(tris)
        calls.append(round(r.build_ms(), 2))
    out["ms"]["first_context_calls_1_to_5"] = calls
    r.close()
    r2 = api.Renderer(1920, 1080)
    calls = []
    for _ in range(3):
        r2.set_scene(tris)
        calls.append(round(r2.build_ms(), 2))
    out["ms"]["second_context_calls_1_to_3"] = calls
    r2.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
