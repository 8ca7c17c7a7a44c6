"""ctypes mirror of include/restir_rt.h + include/restir_rt_internal.h (librestir_rt.so).

`Renderer` keeps the reference example's host API — camera, options, accumulation, one call per
kernel with the reference's kernel names (examples/10_restir_di/10_restir_di.cpp:231-383) — on
top of the C-ABI. There is NO CPU fallback: if the HIP library is missing or no GPU is visible
the constructor raises.
"""
import ctypes as C
import os

import numpy as np

from .types import OPTIONS, RAYGEN, RESERVOIR, TRIANGLE, VISIBILITY, default_options

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librestir_rt.so")

RT_RES_0, RT_RES_1, RT_RES_TEMPORAL = 0, 1, 2
RT_RES_PHYS = 16
RT_BUF_VISIBILITY, RT_BUF_RES_0, RT_BUF_RES_1, RT_BUF_RES_TEMPORAL, RT_BUF_ACCUMULATION, RT_BUF_PIXELS = range(6)

# include/restir_rt.h: the reference-facing boundary (what the stub of INTEGRATION.md section 2 binds)
PUBLIC_EXPORTS = [
    "rt_create", "rt_destroy", "rt_last_error", "rt_set_stream", "rt_sync", "rt_scene_set", "rt_scene_info", "rt_camera_lookat",
    "rt_camera_set", "rt_camera_get", "rt_camera_orbit", "rt_camera_zoom", "rt_camera_pan", "rt_camera_updated",
    "rt_options_set", "rt_options_get", "rt_clear", "rt_raycast", "rt_generate_candidate", "rt_temporal_resampling",
    "rt_save_temporal_reservoir", "rt_spatial_resampling", "rt_resolve", "rt_tone_mapping", "rt_path_trace", "rt_frame",
    "rt_local_rows", "rt_download", "rt_upload", "rt_mg_partition", "rt_mg_unique_id", "rt_mg_create", "rt_mg_destroy",
    "rt_mg_last_error", "rt_mg_frame", "rt_mg_get_stats", "rt_ray_count", "rt_timing_enable", "rt_timing", "rt_build_id",
]
# include/restir_rt_internal.h: what the strip driver, the measurement tools and the parity tests use beyond it
INTERNAL_EXPORTS = [
    "rt_set_stream_own", "rt_camera_pose", "rt_path_trace_rays", "rt_frame_stage", "rt_frame_stage_input",
    "rt_frame_stage_begin", "rt_frame_stage_run", "rt_frame_stage_run_part", "rt_frame_stage_fork", "rt_frame_stage_run_async",
    "rt_frame_stage_run_ranges", "rt_frame_stage_end", "rt_frame_stage_output", "rt_halo_bytes", "rt_halo_pack",
    "rt_halo_unpack", "rt_halo_bitmap_words", "rt_halo_flags_bytes", "rt_halo_flags_pack", "rt_halo_flags_unpack",
    "rt_halo_mark", "rt_halo_scan", "rt_halo_pack_sparse", "rt_halo_unpack_sparse", "rt_halo_mark_sides",
    "rt_halo_pack_sparse_ranges", "rt_halo_unpack_sparse_ranges", "rt_halo_fuse_set", "rt_state_epoch", "rt_get_stream",
    "rt_side_stream", "rt_copy_parts", "rt_wire_delay", "rt_geometry", "rt_res_region", "rt_lane", "rt_mg_bands",
    "rt_mg_load_error", "rt_mg_hub_create", "rt_mg_hub_destroy", "rt_mg_frame_begin", "rt_mg_frame_step", "rt_mg_reset_stats",
    "rt_mg_selftest_rccl", "rt_visibility_rays_walked", "rt_walk_stats_enable", "rt_walk_stats", "rt_stage0_one_launch",
    "rt_row_shaded", "rt_spatial_bytes", "rt_trace_closest", "rt_trace_stats", "rt_bvh_config", "rt_bvh_info", "rt_build_ms",
    "rt_trace_mode", "rt_trace_time", "rt_tuning", "rt_tuning_get", "rt_math_eval",
]
EXPORTS = PUBLIC_EXPORTS + INTERNAL_EXPORTS

RT_MG_TRANSPORT_RCCL, RT_MG_TRANSPORT_LOCAL, RT_MG_TRANSPORT_MIRROR, RT_MG_TRANSPORT_SHM, RT_MG_TRANSPORT_RCCL_SELF, RT_MG_TRANSPORT_WIRE_MODEL = 0, 1, 2, 3, 4, 5
RT_MG_TRANSPORT_MIRROR_WIRE = 6
RT_MG_DENSE, RT_MG_ONE_LANE, RT_MG_SEPARATE_PACK = 1, 2, 4


class RtError(RuntimeError):
    pass


EXP_LIB_PATH = os.path.join(HERE, "librestir_rt_exp.so")
_libs = {}


OCML_LIB_PATH = os.path.join(HERE, "librestir_rt_ocml.so")          # measurement builds (csrc/Makefile `ocml`): device libm = ocml,
OCML_FMA_LIB_PATH = os.path.join(HERE, "librestir_rt_ocml_fma.so")  # ... and FMA contraction as hiprtc's default


def load_library(exp=False, path=None):
    """Load librestir_rt.so (exp=True: librestir_rt_exp.so, the same sources built with -DRT_EXPERIMENTS = the product plus the A/B
    forms that were measured and left off; the variant tests and the A/B tools use it) and declare prototypes. Raises if the
    library was not built."""
    exp = bool(exp)
    key = os.path.abspath(path) if path else exp  # path: one more build of the same sources beside the two (tools/ocml_drift.py)
    if key in _libs:
        return _libs[key]
    path = path or os.environ.get("RT_LIB_PATH", EXP_LIB_PATH if exp else LIB_PATH)  # A/B builds of the same HIP library (tools/experiments)
    if not os.path.exists(path):
        raise RtError(f"{path} not built: run __graft_entry__.build() (make -C cedec_2024_rt_amd/csrc)")
    # One HIP runtime per process: the torch wheel bundles its own libamdhip64 (same SONAME as
    # /opt/rocm's). If librestir_rt.so pulled in the system copy first and torch loaded its own
    # afterwards (torch.distributed / RCCL for the strips), the second runtime finds no GPU.
    # Importing torch first makes the loader resolve our NEEDED libamdhip64.so.7 to the copy that
    # is already mapped. Without torch installed the system runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.rt_create.argtypes = [ci, ci, ci, ci, ci, ci, C.POINTER(vp)]
    L.rt_destroy.argtypes = [vp]
    L.rt_last_error.argtypes = [vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_set_stream.argtypes = [vp, vp]
    L.rt_set_stream_own.argtypes = [vp]
    L.rt_sync.argtypes = [vp]
    L.rt_scene_set.argtypes = [vp, vp, C.c_uint32]
    L.rt_scene_info.argtypes = [vp, vp, vp, vp]
    L.rt_camera_lookat.argtypes = [vp, vp, vp, vp, cf]
    L.rt_camera_set.argtypes = [vp, vp, vp]
    L.rt_camera_get.argtypes = [vp, vp]
    L.rt_camera_orbit.argtypes = [vp, cf, cf]
    L.rt_camera_zoom.argtypes = [vp, cf]
    L.rt_camera_pan.argtypes = [vp, cf, cf]
    L.rt_camera_updated.argtypes = [vp, vp]
    L.rt_camera_pose.argtypes = [vp, vp, vp]
    L.rt_options_set.argtypes = [vp, vp]
    L.rt_options_get.argtypes = [vp, vp]
    L.rt_clear.argtypes = [vp]
    L.rt_raycast.argtypes = [vp]
    L.rt_generate_candidate.argtypes = [vp, ci, ci]
    L.rt_temporal_resampling.argtypes = [vp, ci, ci, ci]
    L.rt_save_temporal_reservoir.argtypes = [vp, ci, ci]
    L.rt_spatial_resampling.argtypes = [vp, ci, ci, ci, ci]
    L.rt_resolve.argtypes = [vp, ci]
    L.rt_tone_mapping.argtypes = [vp]
    L.rt_frame.argtypes = [vp, ci, ci, vp]
    L.rt_path_trace.argtypes = [vp, ci, ci]
    L.rt_path_trace_rays.argtypes = [vp, vp]
    L.rt_frame_stage.argtypes = [vp, ci, ci, ci]
    L.rt_frame_stage_input.argtypes = [vp, ci, vp]
    L.rt_frame_stage_begin.argtypes = [vp, ci, ci, ci]
    L.rt_frame_stage_run.argtypes = [vp, ci, ci, ci, ci]
    L.rt_frame_stage_end.argtypes = [vp, ci]
    L.rt_frame_stage_run_part.argtypes = [vp, ci, ci, ci, ci, ci]
    L.rt_frame_stage_fork.argtypes = [vp]
    L.rt_frame_stage_run_ranges.argtypes = [vp, ci, ci, ci, ci, vp, ci]
    L.rt_frame_stage_run_async.argtypes = [vp, ci, ci, ci, ci, ci]
    L.rt_halo_bitmap_words.argtypes = [vp, ci]
    L.rt_halo_bitmap_words.restype = C.c_size_t
    L.rt_halo_flags_bytes.argtypes = [vp, ci]
    L.rt_halo_flags_bytes.restype = C.c_size_t
    L.rt_halo_flags_pack.argtypes = [vp, ci, ci, vp]
    L.rt_halo_flags_unpack.argtypes = [vp, ci, ci, vp]
    L.rt_halo_mark.argtypes = [vp, ci, ci, ci, ci, vp]
    L.rt_halo_scan.argtypes = [vp, ci, ci, vp]
    L.rt_halo_pack_sparse.argtypes = [vp, ci, ci, ci, vp, vp]
    L.rt_halo_unpack_sparse.argtypes = [vp, ci, ci, ci, vp, vp]
    L.rt_halo_mark_sides.argtypes = [vp, ci, ci, ci, vp, vp]
    L.rt_halo_pack_sparse_ranges.argtypes = [vp, ci, ci, vp, vp, vp, vp]
    L.rt_halo_unpack_sparse_ranges.argtypes = [vp, ci, ci, vp, vp, vp, vp]
    L.rt_frame_stage_output.argtypes = [vp, ci, vp]
    L.rt_local_rows.argtypes = [vp, vp, vp]
    L.rt_download.argtypes = [vp, ci, vp, C.c_size_t]
    L.rt_upload.argtypes = [vp, ci, vp, C.c_size_t]
    L.rt_halo_bytes.argtypes = [vp, ci]
    L.rt_halo_bytes.restype = C.c_size_t
    L.rt_halo_pack.argtypes = [vp, ci, ci, ci, vp]
    L.rt_halo_unpack.argtypes = [vp, ci, ci, ci, vp]
    L.rt_ray_count.argtypes = [vp, vp, vp]
    L.rt_walk_stats_enable.argtypes = [vp, ci]
    L.rt_walk_stats.argtypes = [vp, vp]
    L.rt_timing_enable.argtypes = [vp, ci]
    L.rt_timing.argtypes = [vp, vp]
    L.rt_spatial_bytes.argtypes = [vp, ci, ci, ci, vp, vp]
    L.rt_trace_closest.argtypes = [vp, vp, C.c_uint32, vp]
    L.rt_math_eval.argtypes = [vp, ci, vp, C.c_uint32, vp]
    L.rt_trace_stats.argtypes = [vp, vp, C.c_uint32, vp]
    L.rt_bvh_config.argtypes = [vp, cf]
    L.rt_bvh_info.argtypes = [vp, vp, vp, vp]
    L.rt_build_ms.argtypes = [vp, vp]
    L.rt_trace_mode.argtypes = [vp, ci]
    L.rt_trace_time.argtypes = [vp, vp]
    L.rt_tuning.argtypes = [vp, ci, ci]
    L.rt_tuning_get.argtypes = [vp, ci, vp]
    L.rt_halo_fuse_set.argtypes = [vp, vp]
    L.rt_side_stream.argtypes = [vp, ci, vp]
    L.rt_copy_parts.argtypes = [vp, ci, vp, vp, vp]
    L.rt_build_id.argtypes = []
    L.rt_build_id.restype = C.c_char_p
    L.rt_row_shaded.argtypes = [vp, vp]
    L.rt_stage0_one_launch.argtypes = [vp, vp]
    L.rt_visibility_rays_walked.argtypes = [vp, vp]
    L.rt_state_epoch.argtypes = [vp, vp]
    L.rt_get_stream.argtypes = [vp, vp]
    L.rt_geometry.argtypes = [vp, vp, vp, vp, vp, vp]
    L.rt_res_region.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp]
    L.rt_lane.argtypes = [vp, ci]
    L.rt_mg_partition.argtypes = [ci, ci, ci, vp, vp]
    L.rt_mg_bands.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp]
    L.rt_mg_unique_id.argtypes = [vp]
    L.rt_mg_load_error.restype = C.c_char_p
    L.rt_mg_hub_create.argtypes = [ci, C.POINTER(vp)]
    L.rt_mg_hub_destroy.argtypes = [vp]
    L.rt_mg_create.argtypes = [vp, ci, ci, vp, ci, vp, ci, C.POINTER(vp)]
    L.rt_mg_destroy.argtypes = [vp]
    L.rt_mg_last_error.argtypes = [vp]
    L.rt_mg_last_error.restype = C.c_char_p
    L.rt_mg_frame.argtypes = [vp, ci, ci]
    L.rt_mg_frame_begin.argtypes = [vp, ci, ci]
    L.rt_mg_frame_step.argtypes = [vp, vp]
    L.rt_mg_get_stats.argtypes = [vp, vp]
    L.rt_mg_reset_stats.argtypes = [vp]
    L.rt_mg_selftest_rccl.argtypes = [C.c_size_t]
    _libs[key] = L
    return L


def _exp_default(exp):
    """exp=None: the library Renderer() would pick (RT_EXPERIMENTS=1 selects librestir_rt_exp.so for every context)"""
    return bool(os.environ.get("RT_EXPERIMENTS")) if exp is None else bool(exp)


def build_id(exp=None):
    """rt_build_id(): hash of the sources + flags the library was built from. exp=None: the library a Renderer() of this
    process loads (a context of the experiments library must not report the product's id); Renderer.build_id() is
    the id of the library THAT context runs."""
    return load_library(_exp_default(exp)).rt_build_id().decode()


# ---- native multi-GPU strip driver (csrc/strip_mg.cpp) -------------------------------------------
def mg_partition(height, world, halo=87, row_cost=None, exp=None):
    """[(a, b)] per rank: near-equal strips, or cost-weighted if row_cost (uint32 per storage row) is given."""
    L = load_library(_exp_default(exp))
    b = np.zeros(world + 1, dtype=np.int32)
    rc_ptr = None
    if row_cost is not None:
        rc_arr = np.ascontiguousarray(row_cost, dtype=np.uint32)
        assert rc_arr.size == height
        rc_ptr = _p(rc_arr)
    rc = L.rt_mg_partition(int(height), int(world), int(halo), rc_ptr, _p(b))
    if rc != 0:
        raise ValueError(f"{height} rows over {world} ranks gives strips thinner than the {halo}-row halo")
    return [(int(b[i]), int(b[i + 1])) for i in range(world)]


def mg_bands(bounds, rank, halo=87, exp=None):
    """(boundary, interior) row ranges of a strip, as lists of (row0, row1)."""
    L = load_library(_exp_default(exp))
    world = len(bounds)
    flat = np.array([bounds[0][0]] + [e for _, e in bounds], dtype=np.int32)
    bl, il = np.zeros(4, np.int32), np.zeros(4, np.int32)
    nb, ni = C.c_int(), C.c_int()
    rc = L.rt_mg_bands(_p(flat), world, int(rank), int(halo), _p(bl), C.byref(nb), _p(il), C.byref(ni))
    if rc != 0:
        raise RtError(f"rt_mg_bands -> {rc}")
    return ([(int(bl[2 * i]), int(bl[2 * i + 1])) for i in range(nb.value)],
            [(int(il[2 * i]), int(il[2 * i + 1])) for i in range(ni.value)])


def mg_unique_id(exp=None):
    """128-byte RCCL unique id (bytes); make it on one rank and hand it to the others."""
    L = load_library(_exp_default(exp))
    buf = (C.c_char * 128)()
    rc = L.rt_mg_unique_id(buf)
    if rc != 0:
        raise RtError(f"rt_mg_unique_id -> {rc}: {L.rt_mg_load_error().decode()}")
    return bytes(buf)


class MgHub:
    """Mailbox of the LOCAL transport: several strip contexts of one process on one GPU (tests)."""

    def __init__(self, world, exp=None, renderer=None):
        # the hub must come from the library whose rt_mg_create will receive it: pass the renderer, or exp
        self.L = renderer.L if renderer is not None else load_library(_exp_default(exp))
        self.h = C.c_void_p()
        if self.L.rt_mg_hub_create(int(world), C.byref(self.h)) != 0:
            raise RtError("rt_mg_hub_create failed")

    def close(self):
        if self.h:
            self.L.rt_mg_hub_destroy(self.h)
            self.h = None


class _MgStats(C.Structure):
    _fields_ = [(n, C.c_ulonglong) for n in ("frames", "cold_frames", "host_ns", "plan_wait_ns", "bytes_sent", "messages", "records_sent", "gpu_ns_per_frame", "wire_ns")]


class MultiGpu:
    """One rank of the native strip driver: rt_mg_* over this rank's strip Renderer."""

    def __init__(self, renderer, rank, bounds, transport=RT_MG_TRANSPORT_RCCL, unique_id=None, hub=None, flags=0, shm_name=None):
        self.L, self.r, self.rank, self.bounds = renderer.L, renderer, int(rank), list(bounds)
        world = len(bounds)
        flat = np.array([bounds[0][0]] + [e for _, e in bounds], dtype=np.int32)
        arg = None
        if world > 1:
            if transport == RT_MG_TRANSPORT_RCCL:
                self._id = C.create_string_buffer(bytes(unique_id), 128)
                arg = C.cast(self._id, C.c_void_p)
            elif transport == RT_MG_TRANSPORT_LOCAL:
                if hub.L is not self.L:
                    raise RtError("MgHub and Renderer come from different libraries (product / experiments): MgHub(world, renderer=r)")
                arg = hub.h
            elif transport == RT_MG_TRANSPORT_SHM:
                self._name = C.create_string_buffer(str(shm_name).encode())
                arg = C.cast(self._name, C.c_void_p)
        h = C.c_void_p()
        rc = self.L.rt_mg_create(renderer.h, self.rank, world, _p(flat), int(transport), arg, int(flags), C.byref(h))
        self.h = h
        if rc != 0:
            msg = self.L.rt_mg_last_error(h).decode() if h else "rt_mg_create failed"
            raise RtError(f"rt_mg_create -> {rc}: {msg}")

    def _ck(self, rc):
        if rc != 0:
            raise RtError(f"strip driver error {rc}: {self.L.rt_mg_last_error(self.h).decode()}")

    def frame(self, frame, clear_first=False):
        self._ck(self.L.rt_mg_frame(self.h, int(frame), int(bool(clear_first))))

    def frame_begin(self, frame, clear_first=False):
        self._ck(self.L.rt_mg_frame_begin(self.h, int(frame), int(bool(clear_first))))

    def frame_step(self):
        more = C.c_int(0)
        self._ck(self.L.rt_mg_frame_step(self.h, C.byref(more)))
        return bool(more.value)

    def stats(self):
        st = _MgStats()
        self._ck(self.L.rt_mg_get_stats(self.h, C.byref(st)))
        return {n: int(getattr(st, n)) for n, _ in _MgStats._fields_}

    def reset_stats(self):
        self._ck(self.L.rt_mg_reset_stats(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_mg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mg_frame_lockstep(ranks, frame, clear_first=False):
    """Drive several LOCAL-transport ranks of one process through one frame in lock-step."""
    for m in ranks:
        m.frame_begin(frame, clear_first)
    live = list(ranks)
    while live:
        live = [m for m in live if m.frame_step()]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


_BUF_DTYPE = {
    RT_BUF_VISIBILITY: VISIBILITY, RT_BUF_RES_0: RESERVOIR, RT_BUF_RES_1: RESERVOIR,
    RT_BUF_RES_TEMPORAL: RESERVOIR, RT_BUF_ACCUMULATION: np.dtype(("<f4", 4)), RT_BUF_PIXELS: np.dtype(("u1", 4)),
}


class Renderer:
    """One HIP context = one GPU = one row strip of the image (the whole image by default)."""

    def __init__(self, width, height, device=0, rows=None, halo=0, stream=None, exp=False, lib_path=None):
        self.L = load_library(exp or bool(os.environ.get("RT_EXPERIMENTS")), path=lib_path)  # exp: the library with the A/B forms (rt_tuning's experiment keys)
        self.W, self.H = int(width), int(height)
        r0, r1 = rows if rows is not None else (0, self.H)
        self.rows = (int(r0), int(r1))
        self.halo = int(halo)
        h = C.c_void_p()
        rc = self.L.rt_create(int(device), self.W, self.H, int(r0), int(r1), int(halo), C.byref(h))
        self.h = h
        if rc != 0:
            msg = self.L.rt_last_error(h).decode() if h else "rt_create failed"
            raise RtError(f"rt_create -> {rc}: {msg} (no GPU visible? the HIP path has no CPU fallback)")
        if stream is not None:
            self._ck(self.L.rt_set_stream(self.h, C.c_void_p(int(stream))))
        # A/B runs of the tools: RT_TUNING="8=0,14=0" applies rt_tuning keys to every context (results never depend on them)
        for kv in filter(None, os.environ.get("RT_TUNING", "").split(",")):
            k, v = kv.split("=")
            self._ck(self.L.rt_tuning(self.h, int(k), int(v)))
        a, b = C.c_int(), C.c_int()
        self._ck(self.L.rt_local_rows(self.h, C.byref(a), C.byref(b)))
        self.local_row0, self.local_rows = a.value, b.value

    def _ck(self, rc):
        if rc != 0:
            raise RtError(f"librestir_rt error {rc}: {self.L.rt_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene / camera / options
    def set_scene(self, triangles):
        t = np.ascontiguousarray(triangles, dtype=TRIANGLE)
        self._ck(self.L.rt_scene_set(self.h, _p(t), len(t)))

    def scene_info(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._ck(self.L.rt_scene_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(triangles=a.value, lights=b.value, bvh_height=c.value)

    def lookat(self, eye, center, up=(0.0, 1.0, 0.0), fovy=None):
        fovy = np.float32(np.pi) / np.float32(4.0) if fovy is None else np.float32(fovy)  # 10_restir_di.cpp:242
        e, c, u = (np.asarray(v, dtype=np.float32) for v in (eye, center, up))
        self._ck(self.L.rt_camera_lookat(self.h, _p(e), _p(c), _p(u), C.c_float(fovy)))

    # CameraControl of common/misc.hpp:108-224 (mouse drags as calls)
    def orbit(self, dx, dy):
        self._ck(self.L.rt_camera_orbit(self.h, C.c_float(dx), C.c_float(dy)))

    def zoom(self, dy):
        self._ck(self.L.rt_camera_zoom(self.h, C.c_float(dy)))

    def pan(self, dx, dy):
        self._ck(self.L.rt_camera_pan(self.h, C.c_float(dx), C.c_float(dy)))

    def camera_updated(self):
        f = C.c_int(0)
        self._ck(self.L.rt_camera_updated(self.h, C.byref(f)))
        return bool(f.value)

    def camera_pose(self):
        e, a = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self._ck(self.L.rt_camera_pose(self.h, _p(e), _p(a)))
        return e, a

    def set_raygen(self, raygen, eye):
        rg = np.ascontiguousarray(raygen, dtype=RAYGEN)
        e = np.asarray(eye, dtype=np.float32)
        self._ck(self.L.rt_camera_set(self.h, _p(rg), _p(e)))

    def raygen(self):
        rg = np.zeros(1, dtype=RAYGEN)
        self._ck(self.L.rt_camera_get(self.h, _p(rg)))
        return rg

    def set_options(self, options=None, **kw):
        o = default_options() if options is None else np.array(options, dtype=OPTIONS, copy=True)
        for k, v in kw.items():
            o[k] = v
        self._ck(self.L.rt_options_set(self.h, _p(o)))

    def options(self):
        o = np.zeros(1, dtype=OPTIONS)
        self._ck(self.L.rt_options_get(self.h, _p(o)))
        return o

    # ---- kernels, named as in examples/10_restir_di/10_restir_di.cu
    def clear(self):
        self._ck(self.L.rt_clear(self.h))

    def raycast(self):
        self._ck(self.L.rt_raycast(self.h))

    def generate_candidate(self, frame, dst=RT_RES_0):
        self._ck(self.L.rt_generate_candidate(self.h, frame, dst))

    def temporal_resampling(self, frame, prev=RT_RES_TEMPORAL, inout=RT_RES_0):
        self._ck(self.L.rt_temporal_resampling(self.h, frame, prev, inout))

    def save_temporal_reservoir(self, src=RT_RES_0, dst=RT_RES_TEMPORAL):
        self._ck(self.L.rt_save_temporal_reservoir(self.h, src, dst))

    def spatial_resampling(self, frame, pas, src, dst):
        self._ck(self.L.rt_spatial_resampling(self.h, frame, pas, src, dst))

    def resolve(self, res):
        self._ck(self.L.rt_resolve(self.h, res))

    def tone_mapping(self):
        self._ck(self.L.rt_tone_mapping(self.h))

    def frame_by_kernels(self, frame, clear_first=False):
        """The launch sequence of 10_restir_di.cpp:257-379, kernel by kernel."""
        if clear_first:
            self.clear()
        self.raycast()
        self.generate_candidate(frame, RT_RES_0)
        self.temporal_resampling(frame, RT_RES_TEMPORAL, RT_RES_0)
        self.save_temporal_reservoir(RT_RES_0, RT_RES_TEMPORAL)
        src, dst = RT_RES_0, RT_RES_1
        passes = int(self.options()["spatial_resampling_passes"][0])
        for k in range(passes):
            if k != 0:
                src, dst = dst, src
            self.spatial_resampling(frame, k, src, dst)
        self.resolve(dst)
        self.tone_mapping()
        return dst

    def frame(self, frame, clear_first=False):
        """Fused fast path (rt_frame). Returns the logical buffer resolve read."""
        out = C.c_int(-1)
        self._ck(self.L.rt_frame(self.h, int(frame), int(bool(clear_first)), C.byref(out)))
        return out.value

    def path_trace(self, example, frame):
        """`path_trace` of examples/07_pt (example=7) or examples/09_ris (example=9)."""
        self._ck(self.L.rt_path_trace(self.h, int(example), int(frame)))

    def path_trace_rays(self):
        a = C.c_uint64()
        self._ck(self.L.rt_path_trace_rays(self.h, C.byref(a)))
        return a.value

    def frame_stage(self, frame, stage, clear_first=False):
        self._ck(self.L.rt_frame_stage(self.h, int(frame), int(stage), int(bool(clear_first))))

    def frame_stage_begin(self, frame, stage, clear_first=False):
        self._ck(self.L.rt_frame_stage_begin(self.h, int(frame), int(stage), int(bool(clear_first))))

    def frame_stage_run(self, frame, stage, row0, row1):
        self._ck(self.L.rt_frame_stage_run(self.h, int(frame), int(stage), int(row0), int(row1)))

    def frame_stage_run_part(self, frame, stage, part, row0, row1):
        self._ck(self.L.rt_frame_stage_run_part(self.h, int(frame), int(stage), int(part), int(row0), int(row1)))

    def set_stream(self, stream):
        """Enqueue on the caller's HIP stream (an int handle; 0/None = HIP's null stream)."""
        self._ck(self.L.rt_set_stream(self.h, C.c_void_p(int(stream or 0))))

    def set_stream_own(self):
        self._ck(self.L.rt_set_stream_own(self.h))

    def frame_stage_fork(self):
        self._ck(self.L.rt_frame_stage_fork(self.h))

    def frame_stage_run_async(self, frame, stage, part, row0, row1):
        self._ck(self.L.rt_frame_stage_run_async(self.h, int(frame), int(stage), int(part), int(row0), int(row1)))

    def halo_bitmap_words(self, n_rows):
        return int(self.L.rt_halo_bitmap_words(self.h, int(n_rows)))

    def halo_flags_bytes(self, n_rows):
        return int(self.L.rt_halo_flags_bytes(self.h, int(n_rows)))

    def halo_flags_pack(self, row0, n_rows, ptr):
        self._ck(self.L.rt_halo_flags_pack(self.h, int(row0), int(n_rows), C.c_void_p(int(ptr))))

    def halo_flags_unpack(self, row0, n_rows, ptr):
        self._ck(self.L.rt_halo_flags_unpack(self.h, int(row0), int(n_rows), C.c_void_p(int(ptr))))

    def halo_mark(self, frame, first_pass, n_passes, side, ptr):
        self._ck(self.L.rt_halo_mark(self.h, int(frame), int(first_pass), int(n_passes), int(side), C.c_void_p(int(ptr))))

    def halo_scan(self, n_rows, n_bitmaps, ptr):
        self._ck(self.L.rt_halo_scan(self.h, int(n_rows), int(n_bitmaps), C.c_void_p(int(ptr))))

    def halo_pack_sparse(self, res, row0, n_rows, bitmap_ptr, dst_ptr):
        self._ck(self.L.rt_halo_pack_sparse(self.h, int(res), int(row0), int(n_rows), C.c_void_p(int(bitmap_ptr)), C.c_void_p(int(dst_ptr))))

    def halo_unpack_sparse(self, res, row0, n_rows, bitmap_ptr, src_ptr):
        self._ck(self.L.rt_halo_unpack_sparse(self.h, int(res), int(row0), int(n_rows), C.c_void_p(int(bitmap_ptr)), C.c_void_p(int(src_ptr))))

    def frame_stage_end(self, stage):
        self._ck(self.L.rt_frame_stage_end(self.h, int(stage)))

    def frame_stage_output(self, stage):
        p = C.c_int(-1)
        self._ck(self.L.rt_frame_stage_output(self.h, int(stage), C.byref(p)))
        return RT_RES_PHYS + p.value

    def frame_stage_input(self, stage):
        p = C.c_int(-1)
        self._ck(self.L.rt_frame_stage_input(self.h, int(stage), C.byref(p)))
        return RT_RES_PHYS + p.value

    def sync(self):
        self._ck(self.L.rt_sync(self.h))

    # ---- data movement
    def download(self, buf):
        dt = _BUF_DTYPE[buf]
        n = self.W * self.local_rows
        out = np.zeros(n, dtype=dt)
        self._ck(self.L.rt_download(self.h, buf, _p(out), out.nbytes))
        return out

    def upload(self, buf, arr):
        if buf == RT_BUF_ACCUMULATION:
            a = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1, 4)
        else:
            a = np.ascontiguousarray(arr, dtype=_BUF_DTYPE[buf])
        self._ck(self.L.rt_upload(self.h, buf, _p(a), a.nbytes))

    def halo_bytes(self, n_rows):
        return int(self.L.rt_halo_bytes(self.h, int(n_rows)))

    def halo_pack(self, res, row0, n_rows, device_ptr):
        self._ck(self.L.rt_halo_pack(self.h, res, row0, n_rows, C.c_void_p(int(device_ptr))))

    def halo_unpack(self, res, row0, n_rows, device_ptr):
        self._ck(self.L.rt_halo_unpack(self.h, res, row0, n_rows, C.c_void_p(int(device_ptr))))

    # ---- measurement / utilities
    def ray_count(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._ck(self.L.rt_ray_count(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def visibility_rays_walked(self):
        a = C.c_uint64()
        self._ck(self.L.rt_visibility_rays_walked(self.h, C.byref(a)))
        return a.value

    WALK_KERNELS = ("raycast", "generate_candidate", "spatial_resampling", "resolve")

    def walk_stats_enable(self, on=True):
        self._ck(self.L.rt_walk_stats_enable(self.h, int(bool(on))))

    def walk_stats(self):
        """per kernel: reference rays / walked through the BVH / settled by the self-occlusion test / not evaluated"""
        a = np.zeros(16, dtype=np.uint64)
        self._ck(self.L.rt_walk_stats(self.h, _p(a)))
        return {k: dict(reference_rays=int(a[4 * i]), walked=int(a[4 * i + 1]), self_test=int(a[4 * i + 2]), not_evaluated=int(a[4 * i + 3]))
                for i, k in enumerate(self.WALK_KERNELS)}

    def row_shaded(self):
        """shaded pixels per owned storage row (uint32)"""
        out = np.zeros(self.rows[1] - self.rows[0], dtype=np.uint32)
        self._ck(self.L.rt_row_shaded(self.h, _p(out)))
        return out

    def timing_enable(self, on=True):
        self._ck(self.L.rt_timing_enable(self.h, int(on)))

    def timing(self):
        ms = np.zeros(9, dtype=np.float32)
        self._ck(self.L.rt_timing(self.h, _p(ms)))
        names = ["clear", "raycast", "generate_candidate", "spatial0", "spatial1", "spatial2", "resolve",
                 "tone_mapping", "frame"]
        return dict(zip(names, (float(x) for x in ms)))

    def stage0_one_launch(self):
        """whether the last frame's stage 0 ran as ONE launch on the context's stream (rt_tuning 25): timing()['generate_candidate']
        is then that launch and timing()['raycast'] the empty event bracket in front of it"""
        v = C.c_int()
        self._ck(self.L.rt_stage0_one_launch(self.h, C.byref(v)))
        return bool(v.value)

    def build_id(self):
        """rt_build_id() of the library THIS context runs (product or experiments)"""
        return self.L.rt_build_id().decode()

    def spatial_bytes(self, frame, pas, src):
        a, b = C.c_uint64(), C.c_uint64()
        self._ck(self.L.rt_spatial_bytes(self.h, frame, pas, src, C.byref(a), C.byref(b)))
        return a.value, b.value

    def trace_closest(self, rays):
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        hits = np.zeros((len(r), 4), dtype=np.float32)
        self._ck(self.L.rt_trace_closest(self.h, _p(r), len(r), _p(hits)))
        return hits

    def trace_stats(self, rays):
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        st = np.zeros((len(r), 2), dtype=np.uint32)
        self._ck(self.L.rt_trace_stats(self.h, _p(r), len(r), _p(st)))
        self.last_wave_passes = st >> 16  # wide traversal: inner / leaf passes of the ray's wavefront
        return st & 0xFFFF

    def trace_occluded_ws(self, rays):
        """any-hit answers of the work-sharing walk (trace mode 5; rays with tmax < 0 are lanes without a ray), plus the
        passes each ray's wavefront ran and the steals of each lane"""
        rr = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        raw = np.zeros((len(rr), 2), dtype=np.uint32)
        self._ck(self.L.rt_trace_mode(self.h, 5))
        try:
            self._ck(self.L.rt_trace_stats(self.h, _p(rr), len(rr), _p(raw)))
        finally:
            self._ck(self.L.rt_trace_mode(self.h, 0))
        return (raw[:, 0] >> 31).astype(bool), raw[:, 0] & 0x7fff, raw[:, 1] & 0xffff

    def bvh_config(self, split_factor):
        self._ck(self.L.rt_bvh_config(self.h, C.c_float(split_factor)))

    def trace_mode(self, mode):
        self._ck(self.L.rt_trace_mode(self.h, int(mode)))

    def trace_time(self):
        ms = C.c_float()
        self._ck(self.L.rt_trace_time(self.h, C.byref(ms)))
        return ms.value

    def tuning(self, key, value):
        self._ck(self.L.rt_tuning(self.h, int(key), int(value)))

    def tuning_get(self, key):
        v = C.c_int()
        self._ck(self.L.rt_tuning_get(self.h, int(key), C.byref(v)))
        return v.value

    BVH_BUILDERS = {0: "device Morton + Karras LBVH, host pre-split + 4-wide collapse",
                    1: "host binned SAH + 4-wide collapse",
                    2: "device pre-split + Morton sort + PLOC + top-level SAH sweep + 4-wide collapse",
                    3: "device pre-split + top-down binned SAH (32 bins) + 4-wide collapse, all on the GPU"}

    def bvh_builder(self):
        """name of the builder this context's rt_scene_set used / will use (rt_tuning key 5)"""
        return self.BVH_BUILDERS.get(self.tuning_get(5), "?")

    def build_ms(self):
        ms = C.c_float()
        self._ck(self.L.rt_build_ms(self.h, C.byref(ms)))
        return ms.value

    def bvh_info(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._ck(self.L.rt_bvh_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(references=a.value, wide_records=b.value, wide_height=c.value)

    def math_eval(self, fn, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.size // {26: 2, 33: 2, 35: 2, 31: 12, 32: 12, 36: 3, 37: 3, 38: 14}.get(fn, 1)
        out = np.zeros(n, dtype=np.float32)
        self._ck(self.L.rt_math_eval(self.h, int(fn), _p(x), n, _p(out)))
        return out
