"""ctypes binding of oracle/liboracle.so and runner for oracle/_ref/ref_kernels.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py. The product package never imports this module.
"""
import ctypes as C
import os
import struct
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORACLE_LIB") or os.path.join(HERE, "liboracle.so")  # ORACLE_LIB: e.g. the sanitizer build
REF_BIN = os.path.join(HERE, "_ref", "ref_kernels")

MATH_LIBM = 0
MATH_PORTABLE = 1

# numpy views of the reference's PODs (layouts: SURVEY.md §8a, checked in tests)
TRIANGLE = np.dtype([("v", "<f4", (3, 3)), ("color", "<f4", 3), ("emissive", "<f4", 3)])
VISIBILITY = np.dtype([("uv", "<f4", 2), ("index", "<i4"), ("pad", "<i4")])
RESERVOIR = np.dtype(
    [
        ("origin_position", "<f4", 3),
        ("origin_normal", "<f4", 3),
        ("hit_position", "<f4", 3),
        ("hit_normal", "<f4", 3),
        ("radiance", "<f4", 3),
        ("visibility", "u1"),
        ("pad", "u1", 3),
        ("w_sum", "<f4"),
        ("ucw", "<f4"),
        ("M", "<i4"),
    ]
)
OPTIONS = np.dtype(
    {
        "names": [
            "accumulate", "max_depth", "sky_color", "ris_sample_count",
            "rejection_heuristics_threshold", "use_temporal_resampling", "use_spatial_resampling",
            "spatial_resampling_sample_count", "spatial_resampling_radius",
            "spatial_resampling_passes", "use_shadowed_target_function", "use_visibility_reuse",
        ],
        "formats": ["u1", "<i4", ("<f4", 3), "<i4", "<f4", "u1", "u1", "<i4", "<f4", "<i4", "u1", "u1"],
        "offsets": [0, 4, 8, 20, 24, 28, 29, 32, 36, 40, 44, 45],
        "itemsize": 48,
    }
)
RAYGEN = np.dtype([("origin", "<f4", 3), ("right", "<f4", 3), ("up", "<f4", 3)])
COUNTERS = np.dtype(
    [("rays", "<i8"), ("shaded_pixels", "<i8"), ("spatial_bytes", "<i8"),
     ("spatial_accepted", "<i8"), ("spatial_merged", "<i8")]
)
assert TRIANGLE.itemsize == 60 and VISIBILITY.itemsize == 16 and RESERVOIR.itemsize == 76
assert OPTIONS.itemsize == 48 and RAYGEN.itemsize == 36


def default_options(**kw):
    """common/options.hpp:6-22 defaults; keyword overrides."""
    o = np.zeros(1, dtype=OPTIONS)
    o["accumulate"] = 0
    o["max_depth"] = 6
    o["sky_color"] = 0.0
    o["ris_sample_count"] = 32
    o["rejection_heuristics_threshold"] = 0.2
    o["use_temporal_resampling"] = 0
    o["use_spatial_resampling"] = 0
    o["spatial_resampling_sample_count"] = 5
    o["spatial_resampling_radius"] = 30.0
    o["spatial_resampling_passes"] = 3
    o["use_shadowed_target_function"] = 0
    o["use_visibility_reuse"] = 1
    for k, v in kw.items():
        o[k] = v
    return o


def bench_options(**kw):
    """SURVEY.md §8(d) benchmark options: defaults + temporal + spatial reuse on."""
    d = dict(use_temporal_resampling=1, use_spatial_resampling=1)
    d.update(kw)
    return default_options(**d)


_lib = None


def effective_cpus():
    """CPUs this process may actually use: scheduler affinity capped by the cgroup CPU quota
    (the GPU boxes expose 256 hardware threads but grant a 16-CPU quota: 128 OpenMP threads run
    the oracle 1.6x SLOWER than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return n


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        L.o_scene_create.restype = C.c_void_p
        L.o_scene_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.o_scene_destroy.argtypes = [C.c_void_p]
        L.o_scene_num_lights.argtypes = [C.c_void_p]
        L.o_scene_lights.argtypes = [C.c_void_p, C.c_void_p]
        L.o_scene_set_bvh.argtypes = [C.c_void_p, C.c_int]
        L.o_hashPCG.restype = C.c_uint32
        L.o_hashPCG.argtypes = [C.c_uint32]
        L.o_hashPCG3.restype = C.c_uint32
        L.o_hashPCG3.argtypes = [C.c_uint32] * 3
        L.o_hashPCG4.restype = C.c_uint32
        L.o_hashPCG4.argtypes = [C.c_uint32] * 4
        L.o_pcg_sequence.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.o_raygen_lookat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int]
        L.o_raygen_shoot.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.o_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.o_fn_bulk.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        vp, ci = C.c_void_p, C.c_int
        L.o_raycast.argtypes = [vp, ci, ci, vp, vp, ci, ci, vp]
        L.o_generate_candidate.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp, ci, ci, vp]
        L.o_temporal_resampling.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp, vp, ci, ci, vp]
        L.o_save_temporal_reservoir.argtypes = [ci, ci, vp, vp, ci, ci]
        L.o_spatial_resampling.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, ci, ci, vp]
        L.o_resolve.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp, ci, ci, vp]
        L.o_clear.argtypes = [vp, ci, ci, ci, ci]
        L.o_tone_mapping.argtypes = [vp, vp, ci, ci, ci, ci]
        L.o_frame.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.o_ao_04.argtypes = [vp, vp, vp, ci, ci]
        L.o_path_trace_07.argtypes = [vp, ci, ci, ci, vp, vp, vp, ci, ci, vp]
        L.o_path_trace_08.argtypes = [vp, ci, ci, ci, vp, vp, vp, ci, ci, vp]
        L.o_path_trace_09.argtypes = [vp, ci, ci, ci, vp, vp, vp, ci, ci, vp]
        _lib = L
        L.o_set_threads(effective_cpus())
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def set_math_mode(mode):
    lib().o_set_math_mode(int(mode))


def set_threads(n):
    lib().o_set_threads(int(n))


def max_threads():
    return int(lib().o_get_max_threads())


FN = dict(warp_unit_triangle=(0, 2, 2), sample_hemisphere=(1, 3, 3), sample_2d_gaussian=(2, 2, 2),
          geometry_term=(3, 12, 1), intersect_ray_triangle=(4, 17, 4), luminance=(5, 3, 1),
          normal_rejection=(6, 6, 1), depth_rejection=(7, 9, 1), triangle_props=(8, 9, 7), aces=(9, 1, 1),
          surface_ray=(10, 16, 6), tangent_world=(11, 15, 3),
          resolve_arithmetic=(12, 25, 4),  # 10_restir_di.cu:433-458 with the shadow ray's answer V as an input
          logf=(20, 1, 1), cosf=(21, 1, 1), sinf=(22, 1, 1), expf=(23, 1, 1), pow8=(24, 1, 1),
          pow_gamma=(25, 1, 1), div=(26, 2, 1), sqrt=(27, 1, 1),
          sincos_sin=(28, 1, 1), sincos_cos=(29, 1, 1))  # portable_math.h pm_sincosf (always the portable form)


def fn_bulk(name, x):
    fid, nin, nout = FN[name]
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, nin)
    out = np.zeros((x.shape[0], nout), dtype=np.float32)
    lib().o_fn_bulk(fid, _p(x), _p(out), x.shape[0])
    return out


def raygen_lookat(eye, center, up, fovy, W, H):
    rg = np.zeros(1, dtype=RAYGEN)
    e = np.asarray(eye, dtype=np.float32)
    c = np.asarray(center, dtype=np.float32)
    u = np.asarray(up, dtype=np.float32)
    lib().o_raygen_lookat(_p(rg), _p(e), _p(c), _p(u), C.c_float(np.float32(fovy)), W, H)
    return rg


class Scene:
    """Scene handle of the oracle (triangles + light list + optional CPU BVH)."""

    def __init__(self, triangles, use_bvh=True):
        self.tris = np.ascontiguousarray(triangles, dtype=TRIANGLE)
        self.h = lib().o_scene_create(_p(self.tris), len(self.tris), int(use_bvh))

    def __del__(self):
        try:
            if self.h:
                lib().o_scene_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def lights(self):
        n = lib().o_scene_num_lights(self.h)
        out = np.zeros(n, dtype=np.uint32)
        if n:
            lib().o_scene_lights(self.h, _p(out))
        return out

    def set_bvh(self, use):
        lib().o_scene_set_bvh(self.h, int(use))

    def trace_closest(self, rays, force_brute=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        hits = np.zeros((len(rays), 4), dtype=np.float32)
        lib().o_trace_closest(self.h, _p(rays), len(rays), _p(hits), int(force_brute))
        return hits

    # --- kernels (names as in examples/10_restir_di/10_restir_di.cu) ---
    def raycast(self, W, H, raygen, vis=None, rows=None, cnt=None):
        vis = np.zeros(W * H, dtype=VISIBILITY) if vis is None else vis
        r0, r1 = rows or (0, H)
        lib().o_raycast(self.h, W, H, _p(raygen), _p(vis), r0, r1, _p(cnt))
        return vis

    def generate_candidate(self, W, H, frame, vis, eye, opt, res=None, rows=None, cnt=None):
        res = np.zeros(W * H, dtype=RESERVOIR) if res is None else res
        eye = np.asarray(eye, dtype=np.float32)
        r0, r1 = rows or (0, H)
        lib().o_generate_candidate(self.h, W, H, frame, _p(vis), _p(eye), _p(opt), _p(res), r0, r1, _p(cnt))
        return res

    def temporal_resampling(self, W, H, frame, vis, eye, opt, prev, res, rows=None, cnt=None):
        eye = np.asarray(eye, dtype=np.float32)
        r0, r1 = rows or (0, H)
        lib().o_temporal_resampling(self.h, W, H, frame, _p(vis), _p(eye), _p(opt), _p(prev), _p(res), r0, r1, _p(cnt))
        return res

    def spatial_resampling(self, W, H, frame, pas, vis, eye, opt, rin, rout=None, rows=None, cnt=None):
        rout = np.zeros(W * H, dtype=RESERVOIR) if rout is None else rout
        eye = np.asarray(eye, dtype=np.float32)
        r0, r1 = rows or (0, H)
        lib().o_spatial_resampling(self.h, W, H, frame, pas, _p(vis), _p(eye), _p(opt), _p(rin), _p(rout), r0, r1, _p(cnt))
        return rout

    def resolve(self, accum, W, H, vis, eye, opt, res, rows=None, cnt=None):
        eye = np.asarray(eye, dtype=np.float32)
        r0, r1 = rows or (0, H)
        lib().o_resolve(self.h, _p(accum), W, H, _p(vis), _p(eye), _p(opt), _p(res), r0, r1, _p(cnt))
        return accum

    def frame(self, W, H, frame, raygen, eye, opt, state, cnt=None, tone_map=True):
        """state: dict with vis, r0, r1, temporal, accum, pixels (allocated by new_state)."""
        eye = np.asarray(eye, dtype=np.float32)
        lib().o_frame(self.h, W, H, frame, _p(raygen), _p(eye), _p(opt), _p(state["vis"]), _p(state["r0"]),
                      _p(state["r1"]), _p(state["temporal"]), _p(state["accum"]),
                      _p(state["pixels"]) if tone_map else None, _p(cnt))
        return state

    def path_trace(self, example, W, H, frame, raygen, opt, accum, rows=None, cnt=None):
        """examples/07_pt (example=7), examples/08_nee (8) or examples/09_ris (9) `path_trace` kernel."""
        r0, r1 = rows or (0, H)
        fn = {7: lib().o_path_trace_07, 8: lib().o_path_trace_08, 9: lib().o_path_trace_09}[example]
        fn(self.h, W, H, frame, _p(raygen), _p(opt), _p(accum), r0, r1, _p(cnt))
        return accum

    def ao_04(self, W, H, raygen):
        px = np.zeros((H, W, 4), dtype=np.uint8)
        lib().o_ao_04(self.h, _p(px), _p(raygen), W, H)
        return px


def save_temporal_reservoir(W, H, src, dst):
    lib().o_save_temporal_reservoir(W, H, _p(src), _p(dst), 0, H)
    return dst


def clear(accum, W, H):
    lib().o_clear(_p(accum), W, H, 0, H)
    return accum


def tone_mapping(accum, W, H):
    px = np.zeros((H, W, 4), dtype=np.uint8)
    lib().o_tone_mapping(_p(px), _p(accum), W, H, 0, H)
    return px


def new_state(W, H):
    """Zero-filled buffers of examples/10_restir_di/10_restir_di.cpp:96-122 (temporal history
    defined as Reservoir{} before frame 1)."""
    return dict(
        vis=np.zeros(W * H, dtype=VISIBILITY),
        r0=np.zeros(W * H, dtype=RESERVOIR),
        r1=np.zeros(W * H, dtype=RESERVOIR),
        temporal=np.zeros(W * H, dtype=RESERVOIR),
        accum=np.zeros((W * H, 4), dtype=np.float32),
        pixels=np.zeros((H, W, 4), dtype=np.uint8),
    )


def new_counters():
    return np.zeros(1, dtype=COUNTERS)


# ----------------------------------------------------------------- _ref runner
def have_ref():
    return os.path.exists(REF_BIN)


REF_CAMERA_BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "ref_camera")


def camera_control(eye, lookat, button, dx, dy):
    """The oracle's restatement of CameraControl::cursorPosCallback (common/misc.hpp:129-205): one drag
    event with one button held. Returns (eye, lookat, updated)."""
    e = np.array(eye, dtype=np.float32)
    a = np.array(lookat, dtype=np.float32)
    L = lib()
    L.o_camera_control.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float]
    upd = L.o_camera_control(_p(e), _p(a), int(button), C.c_float(np.float32(dx)), C.c_float(np.float32(dy)))
    return e, a, bool(upd)


def ref_camera_run(eye, lookat, W, H, fovy, events):
    """The REFERENCE'S CameraControl + RayGenerator::lookat (oracle/_ref/ref_camera) over a sequence of
    (button, dx, dy) drag events. Returns per event (eye, lookat, updated, raygen[9]) as uint32 bit patterns."""
    txt = "%r %r %r %r %r %r %d %d %r\n" % (*[float(np.float32(v)) for v in eye], *[float(np.float32(v)) for v in lookat], W, H, float(np.float32(fovy)))
    txt += "".join("%d %r %r\n" % (int(b), float(np.float32(dx)), float(np.float32(dy))) for b, dx, dy in events)
    out = subprocess.check_output([REF_CAMERA_BIN], input=txt.encode()).decode().split("\n")
    rows = []
    for line in out:
        t = line.split()
        if len(t) != 16:
            continue
        v = [int(x, 16) for x in t[:6]] + [int(t[6])] + [int(x, 16) for x in t[7:]]
        rows.append(v)
    return np.array(rows, dtype=np.uint64)


def _write_blobs(path, blobs):
    with open(path, "wb") as f:
        for name, arr in blobs.items():
            data = np.ascontiguousarray(arr).tobytes()
            f.write(name.encode().ljust(32, b"\0"))
            f.write(struct.pack("<Q", len(data)))
            f.write(data)


def _read_blobs(path):
    out = {}
    with open(path, "rb") as f:
        while True:
            nm = f.read(32)
            if len(nm) < 32:
                break
            (n,) = struct.unpack("<Q", f.read(8))
            out[nm.split(b"\0")[0].decode()] = f.read(n)
    return out


def ref_run(cmd, **blobs):
    """Run one command of oracle/_ref/ref_kernels (the reference's own code)."""
    if cmd == "kat":
        import json
        return json.loads(subprocess.check_output([REF_BIN, "kat"]).decode())
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        enc = {}
        for k, v in blobs.items():
            enc[k] = np.array([v], dtype=np.int32) if isinstance(v, (int, np.integer)) else v
        _write_blobs(fin, enc)
        subprocess.check_call([REF_BIN, cmd, fin, fout])
        return _read_blobs(fout)


def ref_fn(name, x):
    fid, nin, nout = FN[name]
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, nin)
    o = ref_run("fn", fn=fid, n=x.shape[0], **{"in": x})
    return np.frombuffer(o["out"], dtype=np.float32).reshape(-1, nout).copy()
