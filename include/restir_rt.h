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
enum { RT_RES_0 = 0, RT_RES_1 = 1, RT_RES_TEMPORAL = 2, RT_RES_PHYS = 16 /* + physical index, see rt_frame_stage_input */ };
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
int rt_set_stream(rt_ctx* ctx, void* hip_stream); /* enqueue on the caller's stream; NULL = HIP's null stream */
int rt_set_stream_own(rt_ctx* ctx);                /* back to the context's own non-blocking stream (default) */
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
int rt_camera_pose(rt_ctx* ctx, float eye[3], float lookat[3]);
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
int rt_path_trace_rays(rt_ctx* ctx, uint64_t* rays); /* raytrace() calls of the last launch */

/* ---- one frame = 10_restir_di.cpp:257-379 (clear_first = camera.is_updated()) ----
 * Same results as the per-kernel sequence; internally generate_candidate+temporal_resampling
 * run as one kernel and save_temporal_reservoir is replaced by rotating three buffers. After
 * the call RT_RES_TEMPORAL names the temporal history and `*final_res` (if not NULL) the buffer
 * resolve read. Single-strip contexts only; strips drive the passes themselves and exchange
 * halos between spatial passes (rt_halo_*). */
int rt_frame(rt_ctx* ctx, int frame, int clear_first, int* final_res);
/* The same frame cut into stages for strip contexts (multi-GPU): stage 0 = [clear,] raycast,
 * generate_candidate(+temporal); stage k in 1..passes = spatial pass k-1; stage passes+1 = resolve,
 * tone_mapping, buffer renaming. Before stage k in 1..passes the caller must fill the halo rows of
 * the buffer rt_frame_stage_input(ctx, k, &p) names (use RT_RES_PHYS + p with rt_halo_pack/unpack). */
int rt_frame_stage(rt_ctx* ctx, int frame, int stage, int clear_first);
int rt_frame_stage_input(rt_ctx* ctx, int stage, int* physical_buffer);
/* Finer control for overlapping halo traffic with compute: rt_frame_stage == _begin, one _run over
 * all owned rows, _end. A caller may instead _run the boundary rows first, start sending them
 * (rt_frame_stage_output names the buffer being written), _run the interior rows, then _end. */
int rt_frame_stage_begin(rt_ctx* ctx, int frame, int stage, int clear_first);
int rt_frame_stage_run(rt_ctx* ctx, int frame, int stage, int row0, int row1);
int rt_frame_stage_run_part(rt_ctx* ctx, int frame, int stage, int part, int row0, int row1); /* stage 0: part 1 = [clear,] raycast, 2 = generate, 0 = both */
/* Second lane: the same as _run_part, on the context's second stream, beside what the
 * main stream does for this stage (interior rows next to boundary rows). It starts after all work
 * enqueued on the main stream at _begin or at the last _fork; _end joins it. The two lanes' rows must
 * be disjoint and the lane must not read halo rows that are still being received. */
int rt_frame_stage_fork(rt_ctx* ctx);
int rt_frame_stage_run_async(rt_ctx* ctx, int frame, int stage, int part, int row0, int row1);
/* the same over up to two disjoint row ranges in ONE launch per kernel (ranges: n x {row0,row1}, n <= 2;
 * lane 1 = second stream): a launch that fits the GPU in one round lasts as long as its slowest
 * wavefront, so a strip's two boundary bands are cheaper as one launch than as two */
int rt_frame_stage_run_ranges(rt_ctx* ctx, int frame, int stage, int part, int n, const int* ranges, int lane);
int rt_frame_stage_end(rt_ctx* ctx, int stage);
int rt_frame_stage_output(rt_ctx* ctx, int stage, int* physical_buffer);

/* ---- host <-> device in the reference's layouts (fixture injection, result read-back) ----
 * Element counts are W * (rows held) where rows held = owned rows + halos clipped to the
 * image; `bytes` must match exactly. RT_BUF_RES_* uploads need the G-buffer of the same
 * frame (rt_raycast or an RT_BUF_VISIBILITY upload first). */
int rt_local_rows(rt_ctx* ctx, int* first_row, int* n_rows);
int rt_download(rt_ctx* ctx, int buf, void* dst, size_t bytes);
int rt_upload(rt_ctx* ctx, int buf, const void* src, size_t bytes);

/* ---- multi-GPU halo rows: pack/unpack `n_rows` storage rows starting at
 * global row `row0` of reservoir buffer `res` to/from a caller-provided DEVICE buffer of
 * rt_halo_bytes(ctx, n_rows) bytes (exchanged by the caller, e.g. RCCL send/recv). ---- */
size_t rt_halo_bytes(rt_ctx* ctx, int n_rows);
int rt_halo_pack(rt_ctx* ctx, int res, int row0, int n_rows, void* device_dst);
int rt_halo_unpack(rt_ctx* ctx, int res, int row0, int n_rows, const void* device_src);

/* Sparse halos: a strip only needs the neighbour records its own pixels will gather, which is a pure
 * function of the RNG and the shaded bits. The RECEIVER marks them (rt_halo_mark: side 0 = strip
 * below, 1 = strip above; needs the neighbour's shaded flags in its G-buffer halo rows, exchanged
 * once per frame with rt_halo_flags_*), sends the bitmap (first 1 + (n_rows*W+31)/32 words; word 0 =
 * record count) to the owner, and the owner answers each pass with the marked records only
 * (rt_halo_pack_sparse after rt_halo_scan on the received bitmap; 80 bytes per record, bitmap order);
 * rt_halo_unpack_sparse scatters them into the halo rows. Same results as the dense calls. */
size_t rt_halo_bitmap_words(rt_ctx* ctx, int n_rows);
size_t rt_halo_flags_bytes(rt_ctx* ctx, int n_rows);
int rt_halo_flags_pack(rt_ctx* ctx, int row0, int n_rows, void* device_dst);
int rt_halo_flags_unpack(rt_ctx* ctx, int row0, int n_rows, const void* device_src);
int rt_halo_mark(rt_ctx* ctx, int frame, int first_pass, int n_passes, int side, void* device_bitmaps); /* n_passes bitmaps, back to back */
int rt_halo_scan(rt_ctx* ctx, int n_rows, int n_bitmaps, void* device_bitmaps);
int rt_halo_pack_sparse(rt_ctx* ctx, int res, int row0, int n_rows, const void* device_bitmap, void* device_dst);
int rt_halo_unpack_sparse(rt_ctx* ctx, int res, int row0, int n_rows, const void* device_bitmap, const void* device_src);
/* the same with both neighbours served by ONE launch each (the native strip driver: a strip's frame is a chain of small
 * launches): rt_halo_mark_sides marks side 0 and / or side 1 (NULL = no neighbour there; if both pointers lie in one
 * allocation, side 0 first and at most 1 MiB apart, everything from the first bitmap to the end of the last is cleared
 * by one memset); the _ranges calls pack / unpack up to two row ranges. */
int rt_halo_mark_sides(rt_ctx* ctx, int frame, int first_pass, int n_passes, void* device_bitmaps_side0, void* device_bitmaps_side1);
int rt_halo_pack_sparse_ranges(rt_ctx* ctx, int res, int n, const int* row0, const int* n_rows, const void* const* device_bitmaps, void* const* device_dsts);
int rt_halo_unpack_sparse_ranges(rt_ctx* ctx, int res, int n, const int* row0, const int* n_rows, const void* const* device_bitmaps, const void* const* device_srcs);

/* The same without pack / unpack launches (r03): the running spatial stage gathers halo records straight from the received
 * lists (per side: the need-bitmap the receiver marked + the list that arrived) and writes the records its neighbours marked
 * (give-bitmap) into the send lists as it produces them. Call between rt_frame_stage_begin(stage in 1..passes) and the _run
 * calls it applies to; rt_frame_stage_end clears it; NULL clears it. Same results as the separate calls. */
typedef struct
{
    const void* need_bitmap[2]; const void* recv_list[2]; /* side 0 = strip below, 1 = above; NULL = none */
    const void* give_bitmap[2]; void* send_list[2];
} rt_halo_fuse;
int rt_halo_fuse_set(rt_ctx* ctx, const rt_halo_fuse* fuse);

/* ---- hooks used by the native strip driver below (and usable by any other driver) ---- */
int rt_state_epoch(rt_ctx* ctx, uint64_t* epoch);   /* changes whenever camera, options, scene or an uploaded G-buffer change */
int rt_get_stream(rt_ctx* ctx, void** hip_stream);  /* the stream calls are enqueued on right now */
int rt_side_stream(rt_ctx* ctx, int which, void** hip_stream); /* which = 0: the tail stream (rt_tuning key 17); the strip driver marks its halo plans there */
/* n <= 8 device-to-device copies in ONE launch on the context's current stream: what the strip driver's stand-in transports
 * (LOCAL, MIRROR) move the parts of an exchange with, as one grouped ncclSend/ncclRecv is one launch */
int rt_copy_parts(rt_ctx* ctx, int n, const void* const* src, void* const* dst, const size_t* bytes);
/* RT_MG_TRANSPORT_WIRE_MODEL: a dependent delay on the context's current stream. phase 0 notes the GPU wall clock when the stream
 * reaches it; phase 1 holds the stream until `ns` after that note (slot 0..7). No host sleep, one sleeping wavefront. */
int rt_wire_delay(rt_ctx* ctx, int phase, int slot, unsigned long long ns);
int rt_geometry(rt_ctx* ctx, int* width, int* height, int* row_begin, int* row_end, int* halo);
/* device addresses of n_rows storage rows of a reservoir buffer: 64-B records and 16-B radiance side
 * records; dense halos travel from / into the buffers themselves */
int rt_res_region(rt_ctx* ctx, int res, int row0, int n_rows, void** rec, size_t* rec_bytes, void** rad, size_t* rad_bytes);
/* rt_lane(ctx, 1) .. rt_lane(ctx, 0): calls in between are enqueued on the context's second stream
 * (the lane of rt_frame_stage_run_async; joined by rt_frame_stage_end) */
int rt_lane(rt_ctx* ctx, int second);

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
enum { RT_MG_TRANSPORT_RCCL = 0, RT_MG_TRANSPORT_LOCAL = 1, RT_MG_TRANSPORT_MIRROR = 2 /* a rank receives what it sent: one rank alone, for overhead measurements (results are not a frame) */,
       RT_MG_TRANSPORT_SHM = 3 /* N processes of one node through a POSIX shared-memory segment (arg = its name, the same string on every
                                  rank): host-staged and blocking, for exact multi-process runs where RCCL cannot go (N ranks on ONE GPU) */,
       RT_MG_TRANSPORT_RCCL_SELF = 4 /* MIRROR with the real thing on the chain (r04): one rank alone, a ONE-rank RCCL communicator, every
                                  exchange = the grouped ncclSend/ncclRecv of the RCCL transport with the true per-side message sizes, on the
                                  stream the RCCL transport uses, addressed to the rank itself — so overhead measurements on a 1-GPU box
                                  contain RCCL's launch and copy kernel (no xGMI wire time) */,
       RT_MG_TRANSPORT_WIRE_MODEL = 5 /* RCCL_SELF + the xGMI wire (r05): an exchange completes no earlier than
                                  max over the two neighbours (bytes to / from that neighbour) / RT_MG_WIRE_GBS (default 153 GB/s, one
                                  link per neighbour) + RT_MG_WIRE_LAT_US (default 5 us) after its data was ready on the stream — a
                                  dependent delay on the exchange's stream (rt_wire_delay), not a host sleep. What tools/strip_overhead.py
                                  reports as THE bound of an N-strip frame on one-GPU boxes. */ };
enum { RT_MG_DENSE = 1 /* whole 87-row bands, sent from the buffers in place */, RT_MG_ONE_LANE = 2 /* no second stream */,
       RT_MG_SEPARATE_PACK = 4 /* sparse halos packed / unpacked by launches of their own (r02) instead of by the spatial passes */ };
typedef struct
{
    unsigned long long frames, cold_frames; /* cold = built its halo plan on the spot (one host wait) */
    unsigned long long host_ns;             /* host time spent inside rt_mg_frame_step (enqueueing) */
    unsigned long long plan_wait_ns;        /* host time waiting for the next frame's plan counts (0 in a steady loop) */
    unsigned long long bytes_sent, messages, records_sent;
    unsigned long long gpu_ns_per_frame;    /* HIP-event time between the starts of the first and the latest frame / (frames - 1) */
    unsigned long long wire_ns;             /* WIRE_MODEL: modelled link time of all exchanges (sum of the dependent delays asked for) */
} rt_mg_stats;
/* strips of >= halo rows; row_cost NULL: near-equal heights; else minimise the most expensive strip
 * (row_cost[r] = e.g. shaded pixels of storage row r). bounds: world + 1 entries. */
int rt_mg_partition(int height, int world, int halo, const uint32_t* row_cost, int* bounds);
/* rows of a strip its neighbours can reach (computed and sent first) and the rest; up to 2 ranges each */
int rt_mg_bands(const int* bounds, int world, int rank, int halo, int* boundary, int* n_boundary, int* interior, int* n_interior);
int rt_mg_unique_id(void* id128);        /* ncclGetUniqueId: one rank makes it, the caller distributes the 128 bytes */
const char* rt_mg_load_error(void);
int rt_mg_hub_create(int world, void** hub); /* LOCAL transport: mailbox of `world` contexts in ONE process (tests) */
int rt_mg_hub_destroy(void* hub);
int rt_mg_create(rt_ctx* ctx, int rank, int world, const int* bounds, int transport, const void* unique_id_or_hub, int flags, rt_mg** out);
int rt_mg_destroy(rt_mg* mg);
const char* rt_mg_last_error(rt_mg* mg);
int rt_mg_frame(rt_mg* mg, int frame, int clear_first);
/* the same frame in segments that end where an exchange was posted: LOCAL contexts are stepped in
 * lock-step (every rank's segment i before any rank's segment i+1); *more = 0 after the last one */
int rt_mg_frame_begin(rt_mg* mg, int frame, int clear_first);
int rt_mg_frame_step(rt_mg* mg, int* more);
int rt_mg_get_stats(rt_mg* mg, rt_mg_stats* out);
int rt_mg_reset_stats(rt_mg* mg);
/* one-rank RCCL communicator sending `bytes` to itself through the grouped send/recv path (1-GPU boxes) */
int rt_mg_selftest_rccl(size_t bytes);

/* ---- measurement ---- */
/* raytrace() calls the REFERENCE makes for the last frame with the current G-buffer and options
 * (the ray of BASELINE.md §3 / SURVEY §8d): N primary + per shaded pixel the visibility-reuse and
 * resolve shadow rays; with use_shadowed_target_function also the target-function rays (candidate 1,
 * temporal 2, per spatial pass 1 + one per neighbour that reaches the merge, counted by replaying
 * the passes' RNG). The build walks fewer BVH rays than that where the reference repeats a ray or
 * the answer cannot matter. shaded = hit & not emissive. */
int rt_ray_count(rt_ctx* ctx, uint64_t* rays, uint64_t* shaded_pixels);
/* visibility-reuse rays the last rt_frame actually walked (rt_tuning key 11; the reference count of rt_ray_count does
 * not change): candidates that survived the temporal merge */
int rt_visibility_rays_walked(rt_ctx* ctx, uint64_t* walked);
/* BVH walks the build really performs, next to the rays the reference traces (r04; common/raytrace.hpp:18-52 is what a
 * "ray" costs the reference: every raytrace() call is a traversal). While enabled, the frame's default kernels count per
 * kernel slot k = 0 raycast, 1 generate_candidate(+temporal_resampling), 2 spatial_resampling (shadowed target function; the
 * unshadowed pass traces nothing), 3 resolve:
 *   out[4k + 0] rays the reference traces in that kernel,  out[4k + 1] of them walked through the BVH here,
 *   out[4k + 2] settled by the one-triangle self-occlusion test (the ray starts below its own surface),
 *   out[4k + 3] not evaluated: the answer is known from the own-visibility flags of an earlier kernel of the frame, or
 *               cannot be observed (visibility-reuse ray of a candidate that lost the temporal merge, weight-0 neighbours).
 * Counters accumulate over launches since rt_walk_stats_enable(ctx, 1) (which zeroes them; both calls synchronise). Covered:
 * k_raycast, the fused work-sharing generate_candidate of rt_frame (unshadowed), k_resolve, the <= 5-neighbour shadowed
 * spatial pass; other variants (rt_tuning A/B forms, shadowed candidates) leave their slot untouched. Results and timing
 * of a frame do not depend on it beyond a few atomics per wavefront; bench.py measures with it off. */
int rt_walk_stats_enable(rt_ctx* ctx, int on);
int rt_walk_stats(rt_ctx* ctx, uint64_t out[16]);
/* shaded pixels of each owned storage row (row_end - row_begin counters): the row cost of rt_mg_partition */
int rt_row_shaded(rt_ctx* ctx, uint32_t* counts);
/* time spent by the last `rt_frame` per kernel, HIP events on the context's stream.
 * ms[0..7] = clear, raycast, generate(+temporal), spatial pass 0,1,2, resolve, tone_mapping;
 * ms[8] = whole frame. Enabled by rt_timing_enable(ctx, 1). With more than 3 spatial passes the
 * passes beyond the third are attributed to ms[6] (resolve); ms[8] stays the whole frame. */
int rt_timing_enable(rt_ctx* ctx, int on);
int rt_timing(rt_ctx* ctx, float ms[9]);

/* ALGORITHMIC bytes (reference record sizes) of the spatial_resampling launch
 * (frame, pass) reading reservoir buffer `in`; `accepted` = neighbours that passed the
 * on-screen / not-self tests. Replays the RNG; independent of reservoir contents. */
int rt_spatial_bytes(rt_ctx* ctx, int frame, int pass, int in, uint64_t* bytes, uint64_t* accepted);

/* ---- BVH utilities (parity tests: BVH traversal == brute force) ----
 * rays: n x {ox,oy,oz, dx,dy,dz, tmin,tmax}; hits: n x {t,u,v, bits(index)}; host pointers. */
int rt_trace_closest(rt_ctx* ctx, const float* rays, uint32_t n, float* hits);
/* per ray {nodes visited, triangle tests} of the same traversal (BVH quality diagnostics); for the
 * wide traversal the upper 16 bits of each word count the inner / leaf passes the ray's wavefront
 * executed while the ray was live (SIMT efficiency diagnostics) */
int rt_trace_stats(rt_ctx* ctx, const float* rays, uint32_t n, uint32_t* stats);
/* BVH build knob, call before rt_scene_set: large triangles are pre-split into fragments no
 * longer than split_factor x (median triangle extent); 0 = no pre-split. Default 10. */
int rt_bvh_config(rt_ctx* ctx, float split_factor);
/* wide_height: levels of the 4-wide tree the kernels walk; a walk holds at most 3 stack entries per level */
int rt_bvh_info(rt_ctx* ctx, uint32_t* n_references, uint32_t* n_wide_records, uint32_t* wide_height);
int rt_build_ms(rt_ctx* ctx, float* ms); /* wall time of the last rt_scene_set (upload + tables + BVH build), synchronised */
/* which traversal rt_trace_closest / rt_trace_stats exercise: 0 = 4-wide quantised BVH + LDS stack
 * (what every frame kernel uses, default), 1 = binary LBVH + stackless trail (A/B measurements),
 * 2/3 = persistent lane-refill queue (closest / any hit), 4 = mode 0 with any-hit (shadow-ray)
 * semantics: hits[i].index >= 0 iff occluded; 5 / 6 = shadow rays in one-wavefront workgroups as the frame kernels
 * walk them, with the work-sharing walk (5; rt_trace_stats then returns its pass / steal counters) or one lane per ray
 * (6); rays with tmax < 0 are lanes without a ray. rt_trace_time: device ms of the last call's kernel. */
int rt_trace_mode(rt_ctx* ctx, int mode);
int rt_trace_time(rt_ctx* ctx, float* ms);
/* Performance knobs; RESULTS NEVER DEPEND ON THEM (every key / value is tested bit for bit against the default). rt_tuning_get
 * returns the value a key holds, -1 meaning "auto" where a key has one. Keys marked [exp] select A/B forms that were measured and
 * left off: they are compiled only into librestir_rt_exp.so (csrc/Makefile, -DRT_EXPERIMENTS); the product library answers
 * RT_ERR_UNSUPPORTED for them. History and numbers of every key: docs/MEASUREMENT_LOG_*.md.
 *
 * Launch geometry
 *  0..3  workgroup -> tile order of raycast (and rt_path_trace) / generate_candidate / spatial_resampling / resolve. Workgroup b runs
 *        on XCD b % 8. 0, 1: XCD k takes ONE band of tile rows, row by row / column by column (r01-r04). r05, the XCDs
 *        interleaved: 2, 3 = XCD k takes tile rows k, k + 8, ... (row by row / column by column), 4 = tile b (row-major) on XCD
 *        b % 8, 5 = the same in stripes 32 tiles wide, 6, 7 = row-major runs of 4 / 16 tiles per XCD. -1 auto (default for all
 *        four): tracing kernels 2 on whole frames, 4 on strips (a band costs what its part of the scene costs: raycast -14 %,
 *        generate_candidate -9 %, resolve -12 %, an 8-rank 4K strip -7 %); the spatial pass 1 on whole frames and strips of
 *        400 rows or more (its +-87-px neighbour window must stay in one XCD's L2), 4 on shorter strips, 7 with the shadowed
 *        target function (4.67 -> 3.83 ms per frame). profiles/r05_tile_interleave_ab.txt.
 *  4     extra LDS bytes per unshadowed spatial workgroup (round 1's occupancy throttle; default 0).
 *  9     register budget of the unshadowed spatial pass in wavefronts per SIMD: -1 auto = 6 (default). [exp] 4, 5, 0 (= unbounded, 7).
 *  13    shadow rays of generate_candidate / resolve through the work-sharing any-hit walk: 1 always (default), 0 never, -1 only
 *        for launches of about one generation of wavefronts.
 *  16    primary rays through the work-sharing closest-hit walk: -1 auto = launches of about one generation of wavefronts, i.e.
 *        strips (default), 0 never, 1 always (a whole frame's coherent 8 x 8 tiles gain nothing: 0.311 -> 0.318 ms).
 *  24    [exp] (r05) raycast at half density: a wavefront carries 32 primary rays and 32 rayless lanes that only take work from
 *        the others' stacks, twice the wavefronts (would a strip's one-generation launch finish sooner with two lanes per ray?
 *        No: 80 -> 94 us for 135 rows, 269 -> 449 us for a whole frame; profiles/r05_half_raycast_ab.txt). Default 0.
 * Scene (before rt_scene_set)
 *  5     BVH builder: 3 = on the device: pre-split, top-down binned SAH, 4-wide collapse; the host reads counters (default;
 *        11 ms for 212 k triangles). [exp] 0 = device LBVH + host collapse (r01), 1 = host binned SAH (the tree builder 3
 *        reproduces; 230 ms), 2 = device PLOC + host SAH over the top 8 192 clusters. All feed the same walk.
 *  7     wide-BVH records emitted breadth-first before the collapse goes depth-first (default 2048; no measurable effect).
 *  10    [exp] PLOC search radius of builder 2 (default 16).
 * Frame structure
 *  6     rt_path_trace: 0 one launch per frame (the reference's shape), 1 one launch per bounce over the compacted list of live
 *        paths, 2 auto (default: per bounce for 09_ris).
 *  14    the NEXT frame's stage 0 on a stream of its own beside this frame's passes, exchanges and resolve: 0 never, 1 its
 *        primary rays (second / third G-buffer set), 2 its generate_candidate + temporal_resampling too (the reference saves the
 *        history before the passes, 10_restir_di.cpp:314-321), -1 auto = 2 (default; level 1 while rt_timing is enabled on a
 *        whole-frame context). The next frame takes the results if frame number, camera, scene, options (rt_state_epoch) and
 *        reservoir buffers are unchanged, and runs its own stage 0 otherwise. rt_sync waits for that stream too.
 *  17    resolve + tone_mapping of a staged frame on a "tail" stream of their own: -1 auto = 1 on (default), 0 off. Off while
 *        rt_timing is enabled.
 *  25    (r05) stage 0 of the staged frame as ONE launch: the candidates' kernel traces the primary ray of its pixel first
 *        (raycast needs nothing else, generate_candidate nothing but it; as two launches the second waits for the first one's
 *        ramp-down): -1 auto = whole-frame contexts (default), 0 two launches, 1 also on strips. Applies to the product's fused
 *        candidate kernel (temporal merge on, unshadowed target), not while rt_timing brackets the kernels; rt_raycast /
 *        rt_generate_candidate are always the two kernels.
 *  20    the staged frame's resolve kernel tone-maps the pixel it has just accumulated (common/kernels/common.cu:30-74 reads
 *        nothing else): 1 (default), 0 = two launches as the reference. rt_resolve / rt_tone_mapping are always the two kernels.
 *  22    (r05) the look-ahead stage 0 (key 14) of frame f+1 waits neither for the main stream (frame f took its own stage 0 from
 *        the look-ahead stream) nor for resolve(f-1): with three G-buffer sets and five reservoir buffers resolve(f-2) is the
 *        last reader of what it overwrites. -1 auto = strips (default: rank 4 of 8 at 1080p 0.306 -> 0.287 ms), 0 never (r04's
 *        dependencies), 1 always (a whole 1080p frame: 1.277 -> 1.296 ms).
 * Spatial pass
 *  8     2 = the wavefront fetches the 64 records of a round together, four lanes per 64-B record, as LDS-DMA loads that land
 *        transposed in LDS, and writes its 64 records the same way (default, the only product form). [exp] 0 = one per-lane
 *        gather per neighbour (r01), 1 = the tile's +-87-px window of shaded bits staged in LDS (r02), 3 = form 2 software-
 *        pipelined over that window (r04: 0.152 against 0.1435 ms), 4 = form 2 as one-wavefront workgroups on 8 x 8 tiles (r05:
 *        more wavefronts in flight, each slower: +1.3 %).
 *  23    [exp] (r05) the LAST spatial pass + resolve in one kernel (k_spatial_resolve): 1 = with the pass's own stores, 2 = records
 *        kept in registers (the pass's output buffer is NOT written), 0 / -1 = two kernels (default). Measured slower:
 *        profiles/r05_fused_tail_ab.txt.
 * Candidates / resolve A/B forms
 *  11    [exp] visibility-reuse rays of the fused candidate kernel only for candidates that survive the temporal merge, through a
 *        compacted queue (75 % survive in the bench scene: no gain). Default 0.
 *  12    [exp] software-pipelined RIS loop form. Default 0.
 *  15    [exp] resolve as a stream of persistent wavefronts (0.48 against 0.36 ms, r02). Default 0.
 * Strips (multi-GPU)
 *  18    rt_halo_mark: rows more than 40 rows from a neighbour's region test the pass's first draws against a bound on the
 *        neighbour distance before replaying log / sqrt / sincos: 1 (default), 0 = full replay.
 *  19    rt_halo_mark collects a workgroup's marks in an LDS bitmap of its +-87-pixel window, one global atomic per non-zero
 *        word: 1 (default; widths that are multiples of 32, reach <= 87 px, <= 3 passes per call), 0 = one atomic per mark.
 *  21    (r05) the shaded-bit rows key 19 reads are built once per camera / scene / option epoch: 1 (default), 0 = in front of
 *        every mark (r04). */
int rt_tuning(rt_ctx* ctx, int key, int value);
/* the value a key holds now (measurement records name the builder / variants that were really used) */
int rt_tuning_get(rt_ctx* ctx, int key, int* value);
/* identity of the BUILD: the first 16 hex digits of a SHA-256 over the library's sources (csrc/ *.hip *.h *.cpp,
 * include/restir_rt.h) and compiler flags, baked in by csrc/Makefile. Unlike a hash of the .so file it is the same for
 * every clean rebuild of the same sources, so counter profiles (profiles/spatial_pmc_latest.json) can be matched to the
 * library that is benchmarked. "unknown" if the library was built without the Makefile. */
const char* rt_build_id(void);
/* elementwise device evaluation of the portable math / IEEE div & sqrt (parity tests); fn ids as in
 * tests/test_portable_math.py; 31..35 (r04): the guarded shared-reciprocal divisions and square root of rt_device.h against
 * the compiler's (31 / 32: 12 floats per item = p0, n0, p1, n1; 33 / 35: 2 floats per item; results are XORs of bit patterns) */
int rt_math_eval(rt_ctx* ctx, int fn, const float* in, uint32_t n, float* out);

#ifdef __cplusplus
}
#endif
#endif /* RESTIR_RT_H */
