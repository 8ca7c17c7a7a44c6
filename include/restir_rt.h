/*
 * restir_rt.h — C-ABI of the MI355X-native ReSTIR DI hot path (librestir_rt.so).
 *
 * Drop-in boundary. The reference has no plugin/FFI layer; its de-facto
 * boundary is the kernel-launch ABI `Shader::launch(name, args, grid, block, stream)`
 * (common/shader.hpp:179-199) used by the frame loop of
 * examples/10_restir_di/10_restir_di.cpp:257-379. This header exports one entry point per
 * reference kernel — same names, same argument meaning — plus context/scene/camera calls that
 * replace the Orochi/HIPRT setup of 10_restir_di.cpp:30-122,183-220, and a fused per-frame
 * call. Plain pointers and sizes only; every call returns 0 on success or an RT_ERR_* code
 * (rt_last_error() gives the text); nothing aborts (the reference SIGTRAPs,
 * common/shader.hpp:10-16).
 *
 * PODs are byte-for-byte the reference's (sizes checked by static_assert in the library and
 * by tests): rt_triangle = Triangle (common/core.hpp:38-43), rt_visibility = Visibility
 * (core.hpp:167-172), rt_reservoir = Reservoir (common/reservoir.hpp:5-38), rt_options =
 * Options (common/options.hpp:4-22), rt_raygen = RayGenerator (common/camera.hpp:5-9).
 *
 * This header is the reference-facing part: what the frame loop of 10_restir_di.cpp (and the stub of
 * INTEGRATION.md §2) binds — 40 entry points. What the strip driver, the measurement tools and the parity
 * tests use beyond it (frame stages, halo calls, BVH / math probes, rt_tuning) is declared in
 * restir_rt_internal.h; the same library exports both.
 *
 * Threading: a context is used by one host thread; all work is enqueued on the context's HIP
 * stream (own stream, or the caller's via rt_set_stream) and is asynchronous until
 * rt_sync / rt_download.
 */
#ifndef RESTIR_RT_H
#define RESTIR_RT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float v[3][3]; float color[3]; float emissive[3]; } rt_triangle;      /* 60 B */
typedef struct { float uv[2]; int32_t index; int32_t _pad; } rt_visibility;           /* 16 B */
typedef struct
{
    float origin_position[3], origin_normal[3], hit_position[3], hit_normal[3], radiance[3];
    uint8_t visibility;
    uint8_t _pad[3];
    float w_sum, ucw;
    int32_t M;
} rt_reservoir;                                                                        /* 76 B */
typedef struct
{
    uint8_t accumulate;
    int32_t max_depth;
    float sky_color[3];
    int32_t ris_sample_count;
    float rejection_heuristics_threshold;
    uint8_t use_temporal_resampling;
    uint8_t use_spatial_resampling;
    int32_t spatial_resampling_sample_count;
    float spatial_resampling_radius;
    int32_t spatial_resampling_passes;
    uint8_t use_shadowed_target_function;
    uint8_t use_visibility_reuse;
} rt_options;                                                                          /* 48 B */
typedef struct { float origin[3], right[3], up[3]; } rt_raygen;                        /* 36 B */

typedef struct rt_ctx rt_ctx;

enum
{
    RT_OK = 0,
    RT_ERR_ARG = 1,         /* bad argument / size mismatch */
    RT_ERR_HIP = 2,         /* a HIP call failed */
    RT_ERR_STATE = 3,       /* call order (e.g. no scene) */
    RT_ERR_BVH_DEPTH = 4,   /* BVH deeper than the traversal stack / trail word can follow */
    RT_ERR_UNSUPPORTED = 5, /* option combination not built yet */
    RT_ERR_NO_DEVICE = 6,
    RT_ERR_COMM = 7         /* RCCL could not be loaded or a collective call failed */
};

/* reservoir buffers of 10_restir_di.cpp:113-122 */
enum { RT_RES_0 = 0, RT_RES_1 = 1, RT_RES_TEMPORAL = 2 };
/* rt_download / rt_upload targets (reference layouts) */
enum
{
    RT_BUF_VISIBILITY = 0,   /* rt_visibility[W*H]            10_restir_di.cpp:108-109 */
    RT_BUF_RES_0 = 1,        /* rt_reservoir[W*H]             :113-114 */
    RT_BUF_RES_1 = 2,        /*                               :117-118 */
    RT_BUF_RES_TEMPORAL = 3, /*                               :121-122 */
    RT_BUF_ACCUMULATION = 4, /* float4[W*H] {R,G,B,spp}       :102-103 */
    RT_BUF_PIXELS = 5        /* RGBA8[W*H]                    :96-97   */
};

/* ---- context (replaces oroInitialize..oroStreamCreate + hiprtCreateContext, :30-79) ---- */
/* Full image W x H. The context owns storage rows [row_begin,row_end) of it (pass 0,H for a
 * single GPU) plus `halo` rows on each side for the spatial pass (multi-GPU row strips,
 * SURVEY.md §8e; halo >= 87 makes N-rank results bit-identical to 1 rank). Rows are STORAGE
 * rows: row r holds the reference's pixel_idx range [r*W, (r+1)*W). */
int rt_create(int device, int width, int height, int row_begin, int row_end, int halo, rt_ctx** out);
int rt_destroy(rt_ctx* ctx);
const char* rt_last_error(rt_ctx* ctx);
int rt_set_stream(rt_ctx* ctx, void* hip_stream); /* enqueue on the caller's stream; NULL = HIP's null stream (default: a non-blocking stream of the context's own) */
int rt_sync(rt_ctx* ctx);

/* ---- scene (replaces loadTrianglesFromObj + light list + buildHiprtGeometry, :184-220) ---- */
/* Uploads the triangles, extracts the emissive-triangle list in index order (:196-205) and
 * builds the BVH (pre-split of large triangles, binary tree by the host SAH builder or the device
 * LBVH kernels, collapse to the 4-wide quantised structure the kernels traverse; rt_tuning keys 5, 7). */
int rt_scene_set(rt_ctx* ctx, const rt_triangle* triangles, uint32_t count);
int rt_scene_info(rt_ctx* ctx, uint32_t* n_triangles, uint32_t* n_lights, uint32_t* bvh_height);

/* ---- camera / options (:240-251, :129) ---- */
/* RayGenerator::lookat evaluated on the host (common/camera.hpp:11-25); eye is also the
 * `eye`/cameraOrig kernel argument. */
int rt_camera_lookat(rt_ctx* ctx, const float eye[3], const float center[3], const float up[3], float fovy);
int rt_camera_set(rt_ctx* ctx, const rt_raygen* raygen, const float eye[3]);
int rt_camera_get(rt_ctx* ctx, rt_raygen* raygen);
/* the examples' interactive camera (common/misc.hpp:108-224 CameraControl), mouse drags as calls:
 * orbit = left button (dx, dy in pixels), zoom = right button, pan = middle button. They update the
 * pose set by rt_camera_lookat, re-derive the RayGenerator and raise the flag rt_camera_updated
 * returns-and-clears (CameraControl::is_updated -> `clear`, 10_restir_di.cpp:257-267). */
int rt_camera_orbit(rt_ctx* ctx, float dx, float dy);
int rt_camera_zoom(rt_ctx* ctx, float dy);
int rt_camera_pan(rt_ctx* ctx, float dx, float dy);
int rt_camera_updated(rt_ctx* ctx, int* updated);
/* Refuses (RT_ERR_UNSUPPORTED) option sets whose reservoir count M could reach 2^30 =
 * 21 * ris_sample_count * (1 + spatial_resampling_sample_count)^spatial_resampling_passes: the 64-B
 * device record keeps M in 30 bits (the reference's int has 31). spatial_resampling_passes == 0 is
 * accepted and behaves as the reference does: resolve reads reservoir_buffer1, which that frame did not write. */
int rt_options_set(rt_ctx* ctx, const rt_options* options);
int rt_options_get(rt_ctx* ctx, rt_options* options);

/* ---- one entry point per reference kernel (examples/10_restir_di/10_restir_di.cu, common/kernels/common.cu) ---- */
int rt_clear(rt_ctx* ctx);                                   /* clear               common.cu:4-17   */
int rt_raycast(rt_ctx* ctx);                                 /* raycast             .cu:9-34         */
int rt_generate_candidate(rt_ctx* ctx, int frame, int dst);  /* generate_candidate  .cu:36-135       */
int rt_temporal_resampling(rt_ctx* ctx, int frame, int prev, int inout); /* .cu:137-237              */
int rt_save_temporal_reservoir(rt_ctx* ctx, int src, int dst);           /* .cu:239-254              */
int rt_spatial_resampling(rt_ctx* ctx, int frame, int pass, int in, int out); /* .cu:256-388         */
int rt_resolve(rt_ctx* ctx, int res);                        /* resolve             .cu:390-459      */
int rt_tone_mapping(rt_ctx* ctx);                            /* tone_mapping        common.cu:30-74  */

/* ---- BASELINE configs #2 / #3 (+ 08_nee): the `path_trace` kernels of
 * examples/07_pt/07_pt.cu:11-90 (example = 7), examples/08_nee/08_nee.cu:11-140 (example = 8) and
 * examples/09_ris/09_ris.cu:11-166 (example = 9); camera, Options (max_depth,
 * sky_color, ris_sample_count, accumulate, use_shadowed_target_function) and the accumulation
 * buffer as for 10_restir_di; follow with rt_tone_mapping as 07_pt.cpp:222 does. ---- */
int rt_path_trace(rt_ctx* ctx, int example, int frame);

/* ---- one frame = 10_restir_di.cpp:257-379 (clear_first = camera.is_updated()) ----
 * Same results as the per-kernel sequence; internally generate_candidate+temporal_resampling
 * run as one kernel and save_temporal_reservoir is replaced by rotating three buffers. After
 * the call RT_RES_TEMPORAL names the temporal history and `*final_res` (if not NULL) the buffer
 * resolve read. Whole-image contexts only (row_begin = 0, row_end = H); row strips are driven by
 * rt_mg_frame below. */
int rt_frame(rt_ctx* ctx, int frame, int clear_first, int* final_res);

/* ---- host <-> device in the reference's layouts (fixture injection, result read-back) ----
 * Element counts are W * (rows held) where rows held = owned rows + halos clipped to the
 * image; `bytes` must match exactly. RT_BUF_RES_* uploads need the G-buffer of the same
 * frame (rt_raycast or an RT_BUF_VISIBILITY upload first). */
int rt_local_rows(rt_ctx* ctx, int* first_row, int* n_rows);
int rt_download(rt_ctx* ctx, int buf, void* dst, size_t bytes);
int rt_upload(rt_ctx* ctx, int buf, const void* src, size_t bytes);


/* ---- multi-GPU: the frame of 10_restir_di.cpp:257-379 on a row strip per GPU (the
 * reference is single-GPU). One process per GPU; each creates its strip context with rt_create(device,
 * W, H, bounds[rank], bounds[rank+1], 87, &ctx), sets scene / camera / options on it as usual, and
 * drives frames with rt_mg_frame instead of rt_frame. Before each spatial pass the strips exchange the
 * reservoir records their neighbours will gather (RCCL send/recv with rank +-1 over xGMI), boundary rows
 * first, interior rows on a second stream meanwhile. N-rank results are bit-identical to one context.
 * Sparse halos (default): only the records a neighbour's RNG will select travel (about 1/5 of the
 * 87-row band); which ones is known one frame ahead while the camera is static, so a steady frame
 * needs no host synchronisation; a frame after a camera / option change synchronises once. ---- */
typedef struct rt_mg rt_mg;
enum { RT_MG_TRANSPORT_RCCL = 0 /* grouped ncclSend/ncclRecv with rank +-1; the stand-in transports of tests and overhead measurements
                                   (one process, one GPU) are in restir_rt_internal.h */ };
enum { RT_MG_DENSE = 1 /* whole 87-row bands, sent from the buffers in place */, RT_MG_ONE_LANE = 2 /* no second stream */,
       RT_MG_SEPARATE_PACK = 4 /* sparse halos packed / unpacked by launches of their own (r02) instead of by the spatial passes */ };
typedef struct
{
    unsigned long long frames, cold_frames; /* cold = built its halo plan on the spot (one host wait) */
    unsigned long long host_ns;             /* host time spent inside rt_mg_frame (enqueueing) */
    unsigned long long plan_wait_ns;        /* host time waiting for the next frame's plan counts (0 in a steady loop) */
    unsigned long long bytes_sent, messages, records_sent;
    unsigned long long gpu_ns_per_frame;    /* HIP-event time between the starts of the first and the latest frame / (frames - 1) */
    unsigned long long wire_ns;             /* WIRE_MODEL: modelled link time of all exchanges (sum of the dependent delays asked for) */
} rt_mg_stats;
/* strips of >= halo rows; row_cost NULL: near-equal heights; else minimise the most expensive strip
 * (row_cost[r] = e.g. shaded pixels of storage row r). bounds: world + 1 entries. */
int rt_mg_partition(int height, int world, int halo, const uint32_t* row_cost, int* bounds);
int rt_mg_unique_id(void* id128);        /* ncclGetUniqueId: one rank makes it, the caller distributes the 128 bytes */
int rt_mg_create(rt_ctx* ctx, int rank, int world, const int* bounds, int transport, const void* unique_id_or_hub, int flags, rt_mg** out);
int rt_mg_destroy(rt_mg* mg);
const char* rt_mg_last_error(rt_mg* mg);
int rt_mg_frame(rt_mg* mg, int frame, int clear_first);
int rt_mg_get_stats(rt_mg* mg, rt_mg_stats* out);

/* ---- measurement ---- */
/* raytrace() calls the REFERENCE makes for the last frame with the current G-buffer and options
 * (the ray of BASELINE.md §3 / SURVEY §8d): N primary + per shaded pixel the visibility-reuse and
 * resolve shadow rays; with use_shadowed_target_function also the target-function rays (candidate 1,
 * temporal 2, per spatial pass 1 + one per neighbour that reaches the merge, counted by replaying
 * the passes' RNG). The build walks fewer BVH rays than that where the reference repeats a ray or
 * the answer cannot matter. shaded = hit & not emissive. */
int rt_ray_count(rt_ctx* ctx, uint64_t* rays, uint64_t* shaded_pixels);
/* time spent by the last `rt_frame` per kernel, HIP events on the context's stream.
 * ms[0..7] = clear, raycast, generate(+temporal), spatial pass 0,1,2, resolve, tone_mapping;
 * ms[8] = whole frame. Enabled by rt_timing_enable(ctx, 1). With more than 3 spatial passes the
 * passes beyond the third are attributed to ms[6] (resolve); ms[8] stays the whole frame.
 * rt_frame runs raycast and generate_candidate (+ temporal_resampling) of a whole image as ONE launch
 * by default: ms[2] is then that launch and ms[1] the empty bracket in front of it, and resolve tone-maps
 * its own pixel (ms[7] = an empty bracket). The per-kernel entry points above are always the reference's
 * kernels; restir_rt_internal.h (rt_tuning keys 25 and 20, rt_stage0_one_launch) has the switches. */
int rt_timing_enable(rt_ctx* ctx, int on);
int rt_timing(rt_ctx* ctx, float ms[9]);


/* identity of the BUILD: the first 16 hex digits of a SHA-256 over the library's sources (csrc/ *.hip *.h *.cpp,
 * include/restir_rt.h, include/restir_rt_internal.h) and compiler flags, baked in by csrc/Makefile. Unlike a hash of the .so file it is the same for
 * every clean rebuild of the same sources, so counter profiles (profiles/spatial_pmc_latest.json) can be matched to the
 * library that is benchmarked. "unknown" if the library was built without the Makefile. */
const char* rt_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* RESTIR_RT_H */
