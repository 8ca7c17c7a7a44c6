/*
 * restir_rt_internal.h — the part of librestir_rt.so's C-ABI that is NOT the reference-facing boundary.
 *
 * include/restir_rt.h declares what a host of the reference's shape binds (context, scene, camera, options, one entry
 * point per reference kernel, rt_frame, rt_mg_*, upload / download, timing, ray count). This header declares what the
 * build's own clients use on top of it:
 *   - the native strip driver (csrc/strip_mg.cpp) and the Python checker schedule (cedec_2024_rt_amd/strips.py):
 *     rt_frame_stage_*, rt_halo_*, stream / lane / region hooks, the stand-in transports of rt_mg_create;
 *   - the measurement tools (tools/, bench.py): rt_walk_stats, rt_spatial_bytes, rt_tuning, rt_wire_delay, rt_build_ms ...;
 *   - the parity tests: rt_trace_closest (BVH == brute force), rt_math_eval (device == host bits).
 * Same conventions as restir_rt.h: plain pointers and sizes, 0 or an RT_ERR_* code, nothing aborts. Nothing here is
 * needed to render a frame; results never depend on any call of this header (rt_tuning keys are tested bit for bit).
 */
#ifndef RESTIR_RT_INTERNAL_H
#define RESTIR_RT_INTERNAL_H

#include "restir_rt.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { RT_RES_PHYS = 16 /* reservoir-buffer argument = RT_RES_PHYS + physical index, see rt_frame_stage_input */ };

/* ---- context / camera / path-trace extras ---- */
int rt_set_stream_own(rt_ctx* ctx);                /* back to the context's own non-blocking stream (default) */
int rt_camera_pose(rt_ctx* ctx, float eye[3], float lookat[3]); /* the pose the interactive camera holds now */

int rt_path_trace_rays(rt_ctx* ctx, uint64_t* rays); /* raytrace() calls of the last launch */

/* ---- rt_frame in stages ---- */
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

/* ---- strip driver: partition helper, stand-in transports, lock-step stepping, statistics ---- */
/* rows of a strip its neighbours can reach (computed and sent first) and the rest; up to 2 ranges each */
int rt_mg_bands(const int* bounds, int world, int rank, int halo, int* boundary, int* n_boundary, int* interior, int* n_interior);
/* stand-in transports of rt_mg_create (RT_MG_TRANSPORT_RCCL = 0 is the product's) */
enum { RT_MG_TRANSPORT_LOCAL = 1, RT_MG_TRANSPORT_MIRROR = 2 /* a rank receives what it sent: one rank alone, for overhead measurements (results are not a frame) */,
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
                                  reports as THE bound of an N-strip frame on one-GPU boxes. */,
       RT_MG_TRANSPORT_MIRROR_WIRE = 6 /* (r06) MIRROR + the same dependent delay: a copy launch per exchange, and the exchange completes no
                                  earlier than the modelled link allows after its data was ready. The other bracket of the bound: RCCL's
                                  one-rank self-send moves a 4K exchange at ~85 GB/s — slower than the 153-GB/s link it stands in for —
                                  so WIRE_MODEL charges a real peer transfer's time twice over; this transport charges the wire alone. */ };
const char* rt_mg_load_error(void);
int rt_mg_hub_create(int world, void** hub); /* LOCAL transport: mailbox of `world` contexts in ONE process (tests) */
int rt_mg_hub_destroy(void* hub);
/* the same frame in segments that end where an exchange was posted: LOCAL contexts are stepped in
 * lock-step (every rank's segment i before any rank's segment i+1); *more = 0 after the last one */
int rt_mg_frame_begin(rt_mg* mg, int frame, int clear_first);
int rt_mg_frame_step(rt_mg* mg, int* more);
int rt_mg_reset_stats(rt_mg* mg);
/* one-rank RCCL communicator sending `bytes` to itself through the grouped send/recv path (1-GPU boxes) */
int rt_mg_selftest_rccl(size_t bytes);

/* ---- measurement ---- */
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
/* whether stage 0 of the last staged frame (rt_frame) ran as ONE launch — primary ray + candidates + temporal merge, rt_tuning
 * key 25 — on the context's stream. rt_timing then reports that launch as ms[2] and ms[1] is the empty bracket where the
 * raycast launch would have been (bench.py: `kernel_ms.stage0`). A look-ahead stage 0 (key 14) does not count. */
int rt_stage0_one_launch(rt_ctx* ctx, int* one_launch);
/* shaded pixels of each owned storage row (row_end - row_begin counters): the row cost of rt_mg_partition */
int rt_row_shaded(rt_ctx* ctx, uint32_t* counts);
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
 * (6); rays with tmax < 0 are lanes without a ray; [exp] 7 (r06) = closest hit with FOUR LANES PER RAY, 16 rays per wavefront
 * (closest_quad: what rt_tuning key 16 = 2 gives the primary rays). rt_trace_time: device ms of the last call's kernel. */
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
 *        strips (default), 0 never, 1 always (a whole frame's coherent 8 x 8 tiles gain nothing: 0.311 -> 0.318 ms); [exp] 2 (r06) =
 *        four lanes per primary ray (k_raycast_quad: 16 rays per wavefront, each lane one child box of the 4-wide record, four
 *        times the wavefronts: 78 -> 95 us for 135 rows, 228 -> 474 us for a whole frame; profiles/r06_quad_walk_ab.txt).
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
 *        candidate kernel (temporal merge on, unshadowed target), also while rt_timing brackets the kernels (r06:
 *        rt_stage0_one_launch); rt_raycast / rt_generate_candidate are always the two kernels.
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
 *        every mark (r04).
 *  26    (r06) rt_halo_mark as one workgroup per (tile, pass) instead of one per tile that replays the passes in series: 0
 *        (default = r02-r05), 1. A strip's mark launch is half a generation of wavefronts and lasts as long as one thread's
 *        chain of 3 x 5 neighbour replays - but it is not on the frame's critical chain and its total work stays the same:
 *        measured +-1 % (profiles/r06_mark_split_ab.txt). */
int rt_tuning(rt_ctx* ctx, int key, int value);
/* the value a key holds now (measurement records name the builder / variants that were really used) */
int rt_tuning_get(rt_ctx* ctx, int key, int* value);
/* elementwise device evaluation of the portable math / IEEE div & sqrt (parity tests); fn ids as in
 * tests/test_portable_math.py; 31..35 (r04): the guarded shared-reciprocal divisions and square root of rt_device.h against
 * the compiler's (31 / 32: 12 floats per item = p0, n0, p1, n1; 33 / 35: 2 floats per item; results are XORs of bit patterns) */
int rt_math_eval(rt_ctx* ctx, int fn, const float* in, uint32_t n, float* out);
#ifdef __cplusplus
}
#endif
#endif /* RESTIR_RT_INTERNAL_H */
