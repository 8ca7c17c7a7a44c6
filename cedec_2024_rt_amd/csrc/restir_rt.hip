/*
 * restir_rt.hip — the C-ABI of include/restir_rt.h: context, scene/BVH set-up, camera, options,
 * one launch wrapper per reference kernel, the staged frame, host<->device conversion, halo calls
 * and measurement. The only translation unit of librestir_rt.so; it includes
 *   frame_kernels.h    every device kernel (32x8-pixel tiles, XCD-banded workgroup order)
 *   bvh.h              device LBVH build kernels + both traversals
 *   bvh_build_host.h   reference pre-split, SAH builder, wide-BVH collapse
 *   rt_device.h        vector math, PCG, the reference's pure functions, HBM record layouts
 * Compile with -ffp-contract=off (parity contract, rt_device.h).
 */
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "../../include/restir_rt_internal.h"
#include "bvh.h"
#include "rt_device.h"

using namespace rt;

static_assert(sizeof(rt_triangle) == 60, "Triangle layout (common/core.hpp:38-43)");
static_assert(sizeof(rt_visibility) == 16, "Visibility layout (common/core.hpp:167-172)");
static_assert(sizeof(rt_reservoir) == 76, "Reservoir layout (common/reservoir.hpp:5-38)");
static_assert(sizeof(rt_options) == 48, "Options layout (common/options.hpp:4-22)");
static_assert(offsetof(rt_options, use_shadowed_target_function) == 44, "Options layout");
static_assert(sizeof(rt_raygen) == 36, "RayGenerator layout (common/camera.hpp:5-9)");
static_assert(sizeof(BvhNode) == 64, "BVH node");

#include "frame_kernels.h"
#include "host_path.h"
#include "bvh_build_host.h"
#include "bvh_build_device.h"

/* ======================================================================= host */

struct rt_ctx
{
    int device = 0, W = 0, H = 0, row_begin = 0, row_end = 0, halo = 0;
    int lrow0 = 0, lrows = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    /* second lane of a frame stage (rt_frame_stage_run_async): interior rows run beside the boundary rows */
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_stage = nullptr, ev_aux = nullptr;
    bool aux_used = false;
    /* The raycast of frame f+1 does not depend on frame f (it needs the camera only), so rt_frame_stage launches it on a
     * stream of its own behind stage 0 of frame f, into the second G-buffer set: it runs beside the spatial passes
     * (HBM-bound) and the halo exchanges of frame f and frame f+1 starts at generate_candidate. The result is used only
     * if nothing it depends on changed meanwhile (epoch: camera, scene, options); otherwise frame f+1 traces its
     * primary rays as usual. rt_tuning key 14. */
    hipStream_t spec_stream = nullptr;
    hipEvent_t ev_spec_go = nullptr, ev_spec_done = nullptr, ev_spec_t[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    /* r05: THREE G-buffer sets (frame f reads set f mod 3): the pipelined stage 0 of frame f+1 overwrites the set of frame f-2,
     * whose last reader — resolve(f-2) — is two tails back, not the set resolve(f-1) may still be reading (with two sets every
     * look-ahead raycast waited for the previous frame's resolve: the one edge that made a strip's frame a dependency chain
     * instead of three streams of independent work, profiles/r05_strip_timelines.txt) */
    static constexpr int NGSET = 3;
    float4* d_gset[3][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}}; /* [set]{vis, g0, g1}; d_vis/d_g0/d_g1 = set gcur */
    int gcur = 0, timed_spec_set = -1;
    bool spec_valid = false, spec_timed[3] = {false, false, false};
    bool spec_outstanding = false; /* work recorded by ev_spec_done that the main stream has not waited for yet */
    uint64_t spec_epoch = 0;
    /* tags of the own-visibility flags (rt_device.h): one number per staged frame and per pipelined stage 0; the launches of
     * a staged frame carry frame_tag (cur_tag while they are enqueued), the per-kernel entry points carry 0 */
    uint32_t ownv_serial = 0, frame_tag = 0, cur_tag = 0, spec_gen_tag = 0;
    /* which G-buffer a reservoir buffer's shaded bits belong to (k_refresh_shaded): gbuf_serial counts G-buffer writes */
    uint64_t gbuf_serial = 0, rec_gserial[5] = {0, 0, 0, 0, 0};
    unsigned long long* d_walk = nullptr; /* rt_walk_stats: 4 kernel slots x 4 counters */
    bool walk_on = false;
    int tune_ws_primary = -1; /* rt_tuning key 16: primary rays with the work-sharing closest-hit walk: -1 auto (r04) = launches of at
                                 most about one generation of wavefronts (a 135-row strip: -1.5 % of its frame; whole frames +2 %), 0 never, 1 always */
    int tune_stream = 0; /* rt_tuning key 15: resolve as a stream of pixels through persistent wavefronts (A/B: slower) */
    int n_cus = 256;
    int tune_spec = -1; /* rt_tuning key 14: -1 auto = strip contexts: primary rays AND candidates of the next frame, 0 never,
                           1 = the next frame's primary rays only, 2 = primary rays and candidates */
    bool lane_saved = false, lane_timing = false; /* rt_lane */
    hipStream_t lane_main = nullptr;
    std::string err;

    int n_tris = 0, n_lights = 0, bvh_height = 0, n_refs = 0;
    /* rt_tuning: tile order per kernel {raycast, generate, spatial, resolve, other} and the spatial
     * pass's occupancy limiter (extra dynamic LDS). Defaults from the sweep in DESIGN.md §5. */
    int tune_tile_mode[5] = {-1, -1, -1, -1, 0}; /* -1 = auto (make_params): r05 — the tracing kernels take their tiles INTERLEAVED over
                                                    the XCDs (whole frames: tile rows k, k + 8, ...; strips: tile b on XCD b % 8), the
                                                    spatial pass keeps an XCD's band of tile rows, column by column (its +-87-px
                                                    neighbour window must stay in that XCD's L2); profiles/r05_tile_interleave_ab.txt */
    int tune_spatial_lds = 0;     /* rt_tuning key 4: extra dynamic LDS per unshadowed spatial workgroup (A/B of the old throttle) */
    int tune_spatial_variant = 2; /* rt_tuning key 8: 2 = k_spatial_coop (default: four lanes per record, LDS-DMA gathers, transposed
                                     stores), 0 = k_spatial_gather (one per-lane gather per neighbour), 1 = k_spatial_lds (staged
                                     shaded-bit window; falls back to 0 where it does not apply) */
    int tune_spatial_waves = -1;  /* rt_tuning key 9: register budget of the unshadowed spatial pass in wavefronts per SIMD: 4, 5, 6, 0 = what the
                                     kernel needs (7), -1 = auto: 4 for the gather kernel (its neighbour window must stay in L2: 0.184 vs 0.202 ms
                                     per pass), none for the LDS-staged kernel (0.181 ms unbounded, 0.188 at 4) — profiles/r02_spatial_variants.json */
    uint32_t* d_shaded_bits = nullptr;
    uint32_t* d_mark_bits = nullptr; /* rt_halo_mark (key 19): shaded bits of all local rows, rebuilt per mark */
    bool shaded_bits_stale = true;
    /* deferred visibility-reuse rays of the fused candidate kernel (frame_kernels.h, DEFER): one queue per lane
     * (main / second stream), its counter, and the last count that reached the host (sizes the next launch) */
    int tune_defer_vis = 0; /* rt_tuning key 11 */
    int tune_ris_pipe = 0;  /* rt_tuning key 12: software-pipelined RIS loop in the fused unshadowed candidate kernel */
    int tune_ws = 1;        /* rt_tuning key 13: work-sharing shadow-ray walk in generate / resolve: -1 auto (launches of
                               at most RT_WS_AUTO_WAVES wavefronts), 0 never, 1 always */
    uint32_t* d_visq[2] = {nullptr, nullptr};
    unsigned int* d_visq_count = nullptr; /* [2] */
    unsigned int* h_visq_count = nullptr; /* pinned [2]; 0xffffffff = unknown */
    uint64_t visq_epoch = 0;
    float last_trace_ms = 0.0f;
    int last_frame = 0; /* frame number of the last rt_frame_stage / rt_spatial_resampling (ray counting) */
    int trace_mode = 0; /* rt_trace_closest / rt_trace_stats: 0 = wide (what the frame kernels use), 1 = binary stackless */
    float bvh_split_factor = 10.0f; /* fragment length in median triangle extents; 0 = no pre-split */
    int bvh_bfs_records = 2048; /* rt_tuning key 7: records emitted breadth-first (top of the tree contiguous) */
    int bvh_builder = 3; /* 0 = device LBVH (Morton/Karras, host pre-split + collapse), 1 = host binned SAH (high quality), 2 = device: pre-split, PLOC,
                            host SAH sweep over the top, wide collapse, 3 (default, r03) = device: pre-split, top-down binned SAH (the host builder's
                            algorithm and tree), wide collapse */
    float build_ms = 0.0f; /* wall time of the last rt_scene_set */
    int ploc_radius = RT_PLOC_RADIUS; /* rt_tuning key 10 */
    float* d_tris = nullptr;
    float4* d_tv = nullptr;
    BvhNode* d_nodes = nullptr;
    float4* d_wide = nullptr;
    int n_wide = 0, wide_height = 0;
    float4* d_trimat = nullptr;
    float4* d_lights = nullptr;
    float4* d_light_ke = nullptr;

    float4 *d_vis = nullptr, *d_g0 = nullptr, *d_g1 = nullptr, *d_accum = nullptr;
    uint32_t* d_pixels = nullptr;
    /* three reservoir buffers carry the reference's names (res_map); a fourth, allocated when the pipelined stage 0 is
     * first used, receives the NEXT frame's candidates while this frame's passes still read the other three */
    /* r05: and a fifth. The buffer the previous frame's resolve reads ("quarantine") is not handed to the look-ahead candidates
     * until one more frame has passed: they get the buffer that left the roles a frame earlier (last reader: two tails back). */
    float4* d_rec[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    float4* d_rad[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int res_map[3] = {0, 1, 2};
    /* Pipelined stage 0 (rt_tuning key 14 = 2, r03): frame f+1's generate_candidate(+temporal_resampling) depends on frame
     * f only through the temporal history, and the reference saves that history right after temporal_resampling, BEFORE
     * the spatial passes (10_restir_di.cpp:314-321) — so stage 0 of frame f+1 (primary rays, candidates, temporal merge)
     * can run beside the spatial passes, halo exchanges and resolve of frame f. It is launched on spec_stream behind
     * stage 0 of frame f, reads the history frame f just wrote, writes the spare buffer, and frame f+1 takes it if it
     * is frame f+1 indeed and nothing it read has changed meanwhile (epoch: camera / scene / options / uploads;
     * res_epoch: any call that writes a reservoir buffer outside the staged frame). Same launches, same results. */
    /* Tail stream (rt_tuning key 17, r03): resolve + tone_mapping of frame f read only frame f's final reservoirs and
     * G-buffer, so a strip's main stream does not wait for them — it goes on with frame f+1's first halo exchange while
     * they run on a stream of their own. Whatever could overwrite what they read waits for ev_tail first: the next frame's
     * first spatial pass (its output buffer can be this frame's final one), the next-but-one frame's pipelined stage 0 (same
     * G-buffer set, spare reservoir buffer), and every call outside the staged frame (join_tail). */
    hipStream_t tail_stream = nullptr;
    hipEvent_t ev_tail_go = nullptr, ev_tail = nullptr;
    /* ev_resolved[0] = behind the latest staged frame's resolve (+ tone mapping) on whatever stream it ran, [1] = the one before:
     * what the pipelined stage 0 waits for before it overwrites the G-buffer set and the reservoir buffer of two frames ago */
    hipEvent_t ev_resolved[2] = {nullptr, nullptr};
    int n_resolved = 0;
    bool resolve_on_tail = false;
    bool tail_pending_main = false;
    int tail_phys = -1; /* reservoir buffer the tail in flight reads (the frame's final one) */
    int tune_tail = -1; /* -1 auto = on, 0 never, 1 always */
    int tune_mark_quick = 1; /* rt_tuning key 18: quick reject in k_halo_mark */
    int tune_mark_window = 1; /* rt_tuning key 19 (r04): k_halo_mark collects a workgroup's marks in LDS first */
    unsigned long long* d_wire = nullptr; /* rt_wire_delay: GPU clock stamps */
    int wall_khz = 100000;
    uint32_t* d_tile_perm[4] = {nullptr, nullptr, nullptr, nullptr}; /* rt_exp_tile_perm (experiments library): dispatch order of a tracing kernel */
    size_t tile_perm_n[4] = {0, 0, 0, 0};
    unsigned long long* d_wave_clock = nullptr; /* rt_exp_wave_clock (experiments library): two words per wavefront of one kernel */
    size_t wave_clock_words = 0;
    int wave_clock_kernel = -1, wave_clock_pass = 0;
    bool stage0_one_launch = false; /* the last staged frame's stage 0 on the main stream traced its primary rays in the candidates' launch */
    int tune_fuse_raycast = -1; /* rt_tuning key 25 (r05): stage 0 as ONE launch (primary ray, then candidates + temporal merge): -1 auto */
    int tune_half_raycast = 0; /* rt_tuning key 24 (r05, experiments build): half-density raycast with helper lanes */
    int tune_fuse_final = -1; /* rt_tuning key 23 (r05): last spatial pass + resolve in one kernel: -1 auto, 0 never, 1 always, 2 = A/B without the pass's stores */
    bool final_fused = false; /* the running frame's last pass has resolved its rows */
    int tune_spec_free = -1; /* rt_tuning key 22 (r05): the look-ahead stage 0 free of the main stream and of the latest resolve: -1 auto = strips */
    int tune_mark_split = 0; /* rt_tuning key 26 (r06): k_halo_mark as one workgroup per (tile, pass) instead of per tile. Measured, no gain: rank 4 of
                                8, WIRE_MODEL 0.3286 / 0.3314 (on) against 0.3310 / 0.3293 ms (off) at 1080p, 0.915 / 0.910 against 0.923 / 0.916 at 4K,
                                MIRROR 0.292 / 0.288 against 0.287 / 0.283 and 0.810 / 0.816 against 0.811 / 0.808 (profiles/r06_mark_split_ab.txt):
                                the marks are not on the frame's critical chain and their total work is the same. Default off. */
    int tune_mark_cache = 1; /* rt_tuning key 21 (r05): the shaded-bit rows of the halo marks are built once per epoch */
    uint64_t mark_bits_epoch = 0, gbuf_epoch = 0; /* epoch d_mark_bits was built under (0: not cached) / the current G-buffer was traced under */
    hipEvent_t ev_mark_bits = nullptr;
    bool mark_bits_event_valid = false, mark_bits_rebuilt = false;
    int tune_fuse_tonemap = 1; /* rt_tuning key 20 (r05): the staged frame's resolve kernel tone-maps its own pixel */
    HaloFuse fuse = {};        /* rt_halo_fuse_set: halo lists read / written by the running stage's spatial pass itself */
    int spare = 3;             /* physical buffer not named by res_map: the look-ahead candidates' */
    int quarantine = 4;        /* the buffer that left the roles at the last take (the previous frame's final one, or a free one) */
    bool spec_gen_valid = false, gen_taken = false;
    int spec_gen_frame = 0, spec_gen_hist = -1;
    uint64_t res_epoch = 1, spec_res_epoch = 0;
    int sub0 = -1, sub1 = -1; /* row sub-range of the running rt_frame_stage_run (-1: all owned rows) */
    int subb0 = 0, subb1 = 0; /* optional second row range of the same launches (rt_frame_stage_run_ranges) */
    int fX = 0, fY = 1, fZ = 2, f_in = 0, f_out = 1, f_stage = 0, f_final = RT_RES_1, f_frame = 0;
    bool f_clear = false;
    uint64_t halo_flags_epoch[2] = {0, 0}; /* epoch at which the neighbour's shaded flags were unpacked (valid while it is the current one) */
    unsigned long long* d_counter = nullptr;
    float4* d_paths[2] = {nullptr, nullptr}; /* wavefront path tracer: live-path lists (64 B per path) */
    unsigned long long* d_pt_counters = nullptr;
    int pt_wavefront = 2; /* rt_tuning key 6: 0 one launch per frame, 1 wavefront, 2 auto (wavefront for 09_ris) */
    void* d_stage = nullptr;
    size_t stage_bytes = 0;

    rt_options opt;
    rt_raygen rg;
    float eye[3] = {0, 0, 0};
    bool has_camera = false, has_scene = false, has_gbuffer = false;
    float cam_eye[3] = {8.0f, 8.0f, 8.0f}, cam_at[3] = {0.0f, 0.0f, 0.0f}, cam_fovy = 0.78539816339f; /* misc.hpp:217-218 */
    bool cam_updated = false;
    /* bumped by everything that can change the G-buffer or the spatial pass's neighbour choice (camera,
     * options, scene, rows, uploads): the strip driver keys its cached halo plans on it */
    uint64_t epoch = 1;

    bool timing = false;
    hipEvent_t ev[10] = {};
    bool ev_created = false;
    float last_ms[9] = {};
    bool last_valid = false;
};

#define RT_CHECK_CTX(ctx) \
    if (!(ctx)) return RT_ERR_ARG;
#define RT_FAIL(ctx, code, ...)                       \
    do                                                \
    {                                                 \
        char _b[512];                                 \
        snprintf(_b, sizeof(_b), __VA_ARGS__);        \
        (ctx)->err = _b;                              \
        return (code);                                \
    } while (0)
#define RT_HIP(ctx, call)                                                                      \
    do                                                                                         \
    {                                                                                          \
        hipError_t _e = (call);                                                                \
        if (_e != hipSuccess) RT_FAIL(ctx, RT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e)); \
    } while (0)

static int stream_priority(const char* env, int dflt);
static rt_options default_options()
{
    rt_options o;
    memset(&o, 0, sizeof(o));
    o.accumulate = 0; o.max_depth = 6;
    o.ris_sample_count = 32; o.rejection_heuristics_threshold = 0.2f;
    o.use_temporal_resampling = 0; o.use_spatial_resampling = 0;
    o.spatial_resampling_sample_count = 5; o.spatial_resampling_radius = 30.0f;
    o.spatial_resampling_passes = 3; o.use_shadowed_target_function = 0; o.use_visibility_reuse = 1;
    return o;
}

enum { K_RAYCAST = 0, K_GENERATE = 1, K_SPATIAL = 2, K_RESOLVE = 3, K_OTHER = 4 };
static FrameParams make_params(const rt_ctx* c, int frame, int pass, int kernel = K_OTHER)
{
    FrameParams P;
    P.W = c->W; P.H = c->H;
    P.row0 = c->sub0 >= 0 ? c->sub0 : c->row_begin;
    P.row1 = c->sub0 >= 0 ? c->sub1 : c->row_end;
    P.rowb0 = c->sub0 >= 0 ? c->subb0 : 0;
    P.rowb1 = c->sub0 >= 0 ? c->subb1 : 0;
    P.lrow0 = c->lrow0; P.lrows = c->lrows;
    P.frame = frame; P.pass = pass;
    P.eye = F3(c->eye[0], c->eye[1], c->eye[2]);
    P.rg_origin = F3(c->rg.origin[0], c->rg.origin[1], c->rg.origin[2]);
    P.rg_right = F3(c->rg.right[0], c->rg.right[1], c->rg.right[2]);
    P.rg_up = F3(c->rg.up[0], c->rg.up[1], c->rg.up[2]);
    P.n_lights = c->n_lights;
    P.accumulate = c->opt.accumulate; P.ris_sample_count = c->opt.ris_sample_count;
    P.use_temporal = c->opt.use_temporal_resampling; P.use_spatial = c->opt.use_spatial_resampling;
    P.spatial_count = c->opt.spatial_resampling_sample_count; P.vis_reuse = c->opt.use_visibility_reuse;
    P.spatial_radius = c->opt.spatial_resampling_radius;
    {
        /* auto. r01-r04 gave XCD k ONE band of tile rows in every kernel (order inside the band: r04, profiles/r04_strip_tile_modes.txt).
         * The tracing kernels' cost per tile follows the scene (sky rows, brick rows ...), so the launch ended with the dearest
         * band's XCD while the others idled: r05 interleaves their tiles over the XCDs — whole frames by tile rows (2), strips
         * tile by tile (4: balanced for any height) — raycast -14 %, generate_candidate -9 %, resolve -12 % of a whole frame; an
         * 8-rank 4K strip -7 % with its spatial launches interleaved too (a 1080p one +-0). The whole frame's spatial pass keeps
         * its bands: its neighbour window has to stay in one XCD's L2 (interleaved: 0.132 -> 0.199 ms), and cutting the bands by
         * cost instead of height gains nothing (profiles/r05_spatial_band_cuts.txt). */
        const bool whole = c->row_begin == 0 && c->row_end == c->H;
        /* the shadowed-target pass traces up to six rays per pixel: a tracing kernel first (runs of 16 tiles per XCD: 4.67 -> 3.83 ms
         * per frame; interleaved tile rows 3.91) */
        const int spatial = c->opt.use_shadowed_target_function ? 7 : (whole || c->row_end - c->row_begin >= 400 ? 1 : 4);
        /* interleaved tile rows give an XCD ceil(tile rows / 8) of them: within 2 % of an eighth for 1080p (135) and 4K (270), not
         * for 1280 x 720 (90 tile rows: 12 against 11.25) — runs of 16 tiles then (balanced for any shape; equal at 1080p) */
        const int tile_rows = (c->row_end - c->row_begin + 7) / 8;
        const int traced = !whole ? 4 : (50 * ((tile_rows + 7) / 8) * 8 <= 51 * tile_rows ? 2 : 7);
        const int autom = kernel == K_SPATIAL ? spatial : (kernel == K_OTHER ? 0 : traced);
        P.tile_mode = c->tune_tile_mode[kernel] >= 0 ? c->tune_tile_mode[kernel] : autom;
    }
    P.ownv_tag = c->cur_tag;
    P.stats = c->walk_on ? c->d_walk : nullptr;
#ifdef RT_EXPERIMENTS
    P.wave_clock = c->d_wave_clock && kernel == c->wave_clock_kernel && (kernel != K_SPATIAL || pass == c->wave_clock_pass) ? c->d_wave_clock : nullptr;
    /* whole launches over the context's own rows only (the grid the permutation was checked against) */
    const bool own_rows = c->sub0 < 0 || (c->sub0 == c->row_begin && c->sub1 == c->row_end && c->subb1 <= c->subb0);
    P.tile_perm = (kernel == K_RAYCAST || kernel == K_GENERATE || kernel == K_RESOLVE) && own_rows ? c->d_tile_perm[kernel] : nullptr;
#endif
    return P;
}
static uint32_t next_ownv_tag(rt_ctx* c)
{
    c->ownv_serial = (c->ownv_serial + 1u) & 0x3fffffffu;
    if (c->ownv_serial == 0u) c->ownv_serial = 1u;
    return c->ownv_serial;
}
static SceneView make_scene(const rt_ctx* c)
{
    SceneView S;
    S.bvh.nodes = c->d_nodes; S.bvh.tv = c->d_tv; S.bvh.n_tris = c->n_tris;
    S.wide.rec = c->d_wide; S.wide.n_tris = c->n_tris; S.wide.n_rec = c->n_wide;
    S.trimat = c->d_trimat; S.lights = c->d_lights; S.light_ke = c->d_light_ke;
    return S;
}
static size_t local_pixels(const rt_ctx* c) { return (size_t)c->W * (size_t)c->lrows; }

extern "C" {

int rt_create(int device, int width, int height, int row_begin, int row_end, int halo, rt_ctx** out)
{
    if (!out || width <= 0 || height <= 0 || row_begin < 0 || row_end > height || row_begin >= row_end || halo < 0)
        return RT_ERR_ARG;
    /* five streams of a context want a hardware queue each (see below): the HOST exports GPU_MAX_HW_QUEUES=8 before its first
     * HIP call (the Python package, bench.py and restir_app do). The library itself does not touch the
     * environment: setenv is not safe beside the host's other threads and has no effect once HIP is initialised. */
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev) return RT_ERR_NO_DEVICE;
    rt_ctx* c = new rt_ctx();
    *out = c;
    c->device = device; c->W = width; c->H = height;
    c->row_begin = row_begin; c->row_end = row_end; c->halo = halo;
    c->lrow0 = row_begin - halo < 0 ? 0 : row_begin - halo;
    const int lend = row_end + halo > height ? height : row_end + halo;
    c->lrows = lend - c->lrow0;
    /* the cooperative record fetches address a reservoir buffer by 32-bit byte offsets (frame_kernels.h, wave_gather_records_at) */
    if ((size_t)c->lrows * (size_t)width > ((size_t)1 << 26)) { delete c; *out = nullptr; return RT_ERR_UNSUPPORTED; }
    c->opt = default_options();
    memset(&c->rg, 0, sizeof(c->rg));
    RT_HIP(c, hipSetDevice(device));
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount;
    }
    /* All streams of the context are created here, in a fixed order: HIP maps streams onto a small pool of hardware queues
     * (GPU_MAX_HW_QUEUES, 4 unless the environment says otherwise) and hands a new stream the least-used one, so streams
     * created lazily, after a host has destroyed and re-created contexts, can land on the queue of the very stream they are
     * meant to run beside (measured: the pipelined stage 0 on the main stream's queue, 0.39 -> 0.51 ms per frame at 1080p
     * in 8 strips, profiles/r03_hw_queue_mapping.txt). Order of importance: main, pipelined stage 0, tail, second lane. */
    RT_HIP(c, hipStreamCreateWithPriority(&c->own_stream, hipStreamNonBlocking, stream_priority("RT_MAIN_PRIORITY", 0)));
    RT_HIP(c, hipStreamCreateWithPriority(&c->spec_stream, hipStreamNonBlocking, stream_priority("RT_SPEC_PRIORITY", 0)));
    RT_HIP(c, hipStreamCreateWithPriority(&c->tail_stream, hipStreamNonBlocking, stream_priority("RT_TAIL_PRIORITY", 0)));
    /* default priority: with the lowest priority the lane was starved in some exchanges (A/B in DESIGN.md §7) */
    RT_HIP(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    RT_HIP(c, hipEventCreateWithFlags(&c->ev_spec_go, hipEventDisableTiming));
    RT_HIP(c, hipEventCreateWithFlags(&c->ev_spec_done, hipEventDisableTiming));
    for (auto& es : c->ev_spec_t) for (auto& e : es) RT_HIP(c, hipEventCreate(&e));
    RT_HIP(c, hipEventCreateWithFlags(&c->ev_tail_go, hipEventDisableTiming));
    RT_HIP(c, hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming));
    for (auto& e : c->ev_resolved) RT_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    RT_HIP(c, hipEventCreateWithFlags(&c->ev_stage, hipEventDisableTiming));
    RT_HIP(c, hipEventCreateWithFlags(&c->ev_aux, hipEventDisableTiming));
    RT_HIP(c, hipEventCreateWithFlags(&c->ev_mark_bits, hipEventDisableTiming));
    c->stream = c->own_stream;
    const size_t n = local_pixels(c);
    RT_HIP(c, hipMalloc(&c->d_vis, n * 16));
    RT_HIP(c, hipMalloc(&c->d_g0, n * 16));
    RT_HIP(c, hipMalloc(&c->d_g1, n * 16));
    RT_HIP(c, hipMalloc(&c->d_accum, n * 16));
    RT_HIP(c, hipMalloc(&c->d_pixels, n * 4));
    for (int k = 0; k < 3; ++k)
    {
        RT_HIP(c, hipMalloc(&c->d_rec[k], n * 64));
        RT_HIP(c, hipMalloc(&c->d_rad[k], n * 16));
        /* temporal history is defined as Reservoir{} before frame 1 */
        RT_HIP(c, hipMemsetAsync(c->d_rec[k], 0, n * 64, c->stream));
        RT_HIP(c, hipMemsetAsync(c->d_rad[k], 0, n * 16, c->stream));
    }
    RT_HIP(c, hipMemsetAsync(c->d_vis, 0, n * 16, c->stream));
    RT_HIP(c, hipMemsetAsync(c->d_g0, 0, n * 16, c->stream));
    RT_HIP(c, hipMemsetAsync(c->d_g1, 0, n * 16, c->stream));
    RT_HIP(c, hipMemsetAsync(c->d_accum, 0, n * 16, c->stream));
    RT_HIP(c, hipMemsetAsync(c->d_pixels, 0, n * 4, c->stream));
    c->d_gset[0][0] = c->d_vis; c->d_gset[0][1] = c->d_g0; c->d_gset[0][2] = c->d_g1;
    RT_HIP(c, hipMalloc(&c->d_counter, 8));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

static void free_scene(rt_ctx* c)
{
    hipFree(c->d_tris); hipFree(c->d_tv); hipFree(c->d_nodes); hipFree(c->d_trimat); hipFree(c->d_lights); hipFree(c->d_light_ke); hipFree(c->d_wide);
    c->d_light_ke = nullptr;
    c->d_wide = nullptr;
    c->d_tris = nullptr; c->d_tv = nullptr; c->d_nodes = nullptr; c->d_trimat = nullptr; c->d_lights = nullptr;
    c->has_scene = false;
}

int rt_destroy(rt_ctx* c)
{
    RT_CHECK_CTX(c);
    hipSetDevice(c->device);
    if (c->stream && c->stream != c->own_stream) hipStreamSynchronize(c->stream);
    if (c->own_stream) hipStreamSynchronize(c->own_stream);
    if (c->spec_stream) hipStreamSynchronize(c->spec_stream); /* may still read the scene: before free_scene */
    if (c->tail_stream) { hipStreamSynchronize(c->tail_stream); hipStreamDestroy(c->tail_stream); }
    if (c->ev_tail_go) hipEventDestroy(c->ev_tail_go);
    if (c->ev_tail) hipEventDestroy(c->ev_tail);
    for (auto& e : c->ev_resolved) if (e) hipEventDestroy(e);
    c->spec_valid = false;
    if (c->aux_stream) { hipStreamSynchronize(c->aux_stream); hipStreamDestroy(c->aux_stream); }
    if (c->ev_stage) hipEventDestroy(c->ev_stage);
    if (c->ev_aux) hipEventDestroy(c->ev_aux);
    if (c->ev_mark_bits) hipEventDestroy(c->ev_mark_bits);
    free_scene(c);
    if (c->spec_stream) { hipStreamSynchronize(c->spec_stream); hipStreamDestroy(c->spec_stream); }
    if (c->ev_spec_go) hipEventDestroy(c->ev_spec_go);
    if (c->ev_spec_done) hipEventDestroy(c->ev_spec_done);
    for (auto& es : c->ev_spec_t) for (auto& e : es) if (e) hipEventDestroy(e);
    for (auto& gs : c->d_gset) for (auto& p : gs) hipFree(p);
    hipFree(c->d_accum); hipFree(c->d_pixels);
    for (int k = 0; k < 5; ++k) { hipFree(c->d_rec[k]); hipFree(c->d_rad[k]); }
    hipFree(c->d_shaded_bits); hipFree(c->d_mark_bits);
    hipFree(c->d_visq[0]); hipFree(c->d_visq[1]); hipFree(c->d_visq_count);
    if (c->h_visq_count) hipHostFree(c->h_visq_count);
    hipFree(c->d_walk); hipFree(c->d_wire); hipFree(c->d_wave_clock);
    for (auto& p : c->d_tile_perm) hipFree(p);
    hipFree(c->d_counter); hipFree(c->d_stage); hipFree(c->d_paths[0]); hipFree(c->d_paths[1]); hipFree(c->d_pt_counters);
    if (c->ev_created) for (auto& e : c->ev) hipEventDestroy(e);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
    return RT_OK;
}

const char* rt_last_error(rt_ctx* c) { return c ? c->err.c_str() : "null context"; }

int rt_set_stream(rt_ctx* c, void* s)
{
    RT_CHECK_CTX(c);
    /* exactly the caller's stream; NULL is HIP's null (default) stream, which is what
     * torch.cuda.current_stream() is unless the caller entered a stream context */
    c->stream = (hipStream_t)s;
    return RT_OK;
}
int rt_set_stream_own(rt_ctx* c)
{
    RT_CHECK_CTX(c);
    c->stream = c->own_stream;
    return RT_OK;
}
int rt_sync(rt_ctx* c)
{
    RT_CHECK_CTX(c);
    RT_HIP(c, hipStreamSynchronize(c->stream));
    if (c->spec_stream) RT_HIP(c, hipStreamSynchronize(c->spec_stream)); /* the next frame's raycast is work of this call too */
    if (c->tail_stream) RT_HIP(c, hipStreamSynchronize(c->tail_stream));
    c->tail_pending_main = false;
    return RT_OK;
}

static int ensure_stage(rt_ctx* c, size_t bytes)
{
    if (c->stage_bytes >= bytes) return RT_OK;
    if (c->d_stage) { RT_HIP(c, hipStreamSynchronize(c->stream)); hipFree(c->d_stage); c->d_stage = nullptr; }
    RT_HIP(c, hipMalloc(&c->d_stage, bytes));
    c->stage_bytes = bytes;
    return RT_OK;
}

static int build_wide(rt_ctx* c, const rt_triangle* tris, int n_refs)
{
    const size_t n_bin = (size_t)(n_refs > 1 ? n_refs - 1 : 1);
    std::vector<BvhNode> bin(n_bin);
    RT_HIP(c, hipMemcpyAsync(bin.data(), c->d_nodes, n_bin * sizeof(BvhNode), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    std::vector<WideRec> recs;
    c->wide_height = collapse_wide(bin, tris, recs, c->bvh_bfs_records);
    if (RT_WIDE_TOTAL_STACK >= 64 && 3 * c->wide_height + 1 > WIDE_LDS_STACK + WIDE_OVF_STACK)
        RT_FAIL(c, RT_ERR_BVH_DEPTH, "wide BVH height %d exceeds the traversal stack (%d entries)", c->wide_height,
                WIDE_LDS_STACK + WIDE_OVF_STACK);
    c->n_wide = (int)recs.size();
    RT_HIP(c, hipMalloc(&c->d_wide, recs.size() * sizeof(WideRec)));
    RT_HIP(c, hipMemcpyAsync(c->d_wide, recs.data(), recs.size() * sizeof(WideRec), hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

/* The whole build on the device (bvh_build_device.h): pre-split, Morton sort, PLOC hierarchy, wide collapse.
 * The host only reads counters back. */
#ifndef RT_PLOC_TOP
#define RT_PLOC_TOP 8192 /* clusters left when the host builds the top of the tree (builder 2): 2048 / 8192 / 32768 -> build 10.4 / 13.3 / 25.3 ms, frame +3.5 / +1.5 / +0.5 % against the host SAH tree */
#endif
static int build_bvh_device(rt_ctx* c, int n_tris)
{
    hipStream_t st = c->stream;
    std::vector<void*> tmp; /* device scratch, freed on every exit */
    auto cleanup = [&]() { for (void* p : tmp) hipFree(p); tmp.clear(); };
    auto dalloc = [&](size_t bytes) -> void* { void* p = nullptr; if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr; tmp.push_back(p); return p; };
#define BD_FAIL(code, ...) do { char _b[256]; snprintf(_b, sizeof(_b), __VA_ARGS__); c->err = _b; cleanup(); return (code); } while (0)
#define BD_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) BD_FAIL(RT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e)); } while (0)
#define BD_PTR(p) do { if (!(p)) BD_FAIL(RT_ERR_HIP, "out of device memory in the BVH build"); } while (0)
    const int gt = (n_tris + 255) / 256;
    /* 1. extents, scene bounds, median extent */
    float* d_ext = (float*)dalloc((size_t)n_tris * 4); BD_PTR(d_ext);
    float* d_ext2 = (float*)dalloc((size_t)n_tris * 4); BD_PTR(d_ext2);
    unsigned int* d_bounds = (unsigned int*)dalloc(24); BD_PTR(d_bounds);
    const unsigned int binit[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    BD_HIP(hipMemcpyAsync(d_bounds, binit, 24, hipMemcpyHostToDevice, st));
    k_tri_extents<<<gt, 256, 0, st>>>(c->d_tris, n_tris, d_ext, d_bounds);
    BD_HIP(hipGetLastError());
    size_t tb = 0;
    BD_HIP(rocprim::radix_sort_keys(nullptr, tb, d_ext, d_ext2, (size_t)n_tris, 0, 32, st));
    void* d_t0 = dalloc(tb); BD_PTR(d_t0);
    BD_HIP(rocprim::radix_sort_keys(d_t0, tb, d_ext, d_ext2, (size_t)n_tris, 0, 32, st));
    float median = 0.0f;
    unsigned int hb[6];
    BD_HIP(hipMemcpyAsync(&median, d_ext2 + n_tris / 2, 4, hipMemcpyDeviceToHost, st));
    BD_HIP(hipMemcpyAsync(hb, d_bounds, 24, hipMemcpyDeviceToHost, st));
    BD_HIP(hipStreamSynchronize(st));
    auto dec = [](unsigned int u) -> float { const unsigned int v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; float f; memcpy(&f, &v, 4); return f; };
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = dec(hb[a]); hi[a] = dec(hb[3 + a]); }
    float ext = 0.0f;
    for (int a = 0; a < 3; ++a) ext = fmaxf(ext, fmaxf(fabsf(lo[a]), fabsf(hi[a])));
    const float pad = 4e-5f * (ext > 1.0f ? ext : 1.0f); /* as the host path: hits lie within rounding distance of the triangle */
    const float3 slo = make_float3(lo[0], lo[1], lo[2]);
    const float3 sext = make_float3(fmaxf(hi[0] - lo[0], 1e-20f), fmaxf(hi[1] - lo[1], 1e-20f), fmaxf(hi[2] - lo[2], 1e-20f));
    (void)slo; (void)sext; /* Morton keys: builder 2 (experiments build) */
    /* 2. early split clipping: count, scan, emit */
    uint32_t* d_cnt = (uint32_t*)dalloc((size_t)n_tris * 4); BD_PTR(d_cnt);
    uint32_t* d_off = (uint32_t*)dalloc((size_t)n_tris * 4); BD_PTR(d_off);
    size_t sb = 0;
    BD_HIP(rocprim::exclusive_scan(nullptr, sb, d_cnt, d_off, 0u, (size_t)n_tris, rocprim::plus<uint32_t>(), st));
    void* d_t1 = dalloc(sb); BD_PTR(d_t1);
    float L = c->bvh_split_factor > 0.0f ? c->bvh_split_factor * median : 0.0f;
    const size_t budget = (size_t)n_tris * 4 + 1024;
    int n = 0;
    for (int it = 0; it < 16; ++it)
    {
        k_split_refs<false><<<gt, 256, 0, st>>>(c->d_tris, n_tris, L, pad, nullptr, d_cnt, nullptr, nullptr);
        BD_HIP(hipGetLastError());
        BD_HIP(rocprim::exclusive_scan(d_t1, sb, d_cnt, d_off, 0u, (size_t)n_tris, rocprim::plus<uint32_t>(), st));
        uint32_t last[2];
        BD_HIP(hipMemcpyAsync(&last[0], d_off + n_tris - 1, 4, hipMemcpyDeviceToHost, st));
        BD_HIP(hipMemcpyAsync(&last[1], d_cnt + n_tris - 1, 4, hipMemcpyDeviceToHost, st));
        BD_HIP(hipStreamSynchronize(st));
        n = (int)(last[0] + last[1]);
        if ((size_t)n <= budget || L <= 0.0f) break;
        L *= 1.5f;
    }
    c->n_refs = n;
    float* d_boxes = (float*)dalloc((size_t)n * 24); BD_PTR(d_boxes);
    int* d_ref_tri = (int*)dalloc((size_t)n * 4); BD_PTR(d_ref_tri);
    k_split_refs<true><<<gt, 256, 0, st>>>(c->d_tris, n_tris, L, pad, d_off, nullptr, d_boxes, d_ref_tri);
    BD_HIP(hipGetLastError());
    BD_HIP(hipMalloc(&c->d_tv, (size_t)n_tris * 48));
    k_bvh_tv<<<gt, 256, 0, st>>>(c->d_tris, n_tris, c->d_tv);
    BD_HIP(hipGetLastError());
    if (n < 2) BD_FAIL(RT_ERR_STATE, "internal: the device build needs at least two references");
    const int grid = (n + 255) / 256;
    int2* d_children = (int2*)dalloc((size_t)n * 8); BD_PTR(d_children);
    int* d_parent = (int*)dalloc((size_t)n * 4); BD_PTR(d_parent);
    float* d_node_boxes = (float*)dalloc((size_t)n * 24); BD_PTR(d_node_boxes);
    const uint32_t* d_leaf_ids = nullptr; /* leaf position -> reference, for k_bvh_emit */
    SahState* d_sah_state = nullptr;
    if (c->bvh_builder == 3)
    {
        /* 3b. top-down binned SAH on the device (bvh_build_device.h): the host enqueues levels and, from level 16 on,
         * looks every fourth level whether large nodes are left (24 bytes of counters; a balanced tree over 2^24 references
         * is done by then, deeper ones keep going up to SAH_MAX_LEVELS) */
        const int max_active = n / (SAH_SMALL + 1) + 2, max_small = n / 2 + 2;
        uint32_t* d_order[2] = {(uint32_t*)dalloc((size_t)n * 4), (uint32_t*)dalloc((size_t)n * 4)}; BD_PTR(d_order[0]); BD_PTR(d_order[1]);
        int* d_seg[2] = {(int*)dalloc((size_t)n * 4), (int*)dalloc((size_t)n * 4)}; BD_PTR(d_seg[0]); BD_PTR(d_seg[1]);
        unsigned int* d_flag = (unsigned int*)dalloc((size_t)n * 4); BD_PTR(d_flag);
        unsigned int* d_pre = (unsigned int*)dalloc((size_t)n * 4); BD_PTR(d_pre);
        SahNode* d_act[2] = {(SahNode*)dalloc((size_t)max_active * sizeof(SahNode)), (SahNode*)dalloc((size_t)max_active * sizeof(SahNode))}; BD_PTR(d_act[0]); BD_PTR(d_act[1]);
        SahSplit* d_splits = (SahSplit*)dalloc((size_t)max_active * sizeof(SahSplit)); BD_PTR(d_splits);
        const size_t bin_words = (size_t)max_active * SAH_NODE_BIN_WORDS;
        unsigned int* d_bins = (unsigned int*)dalloc(bin_words * 4); BD_PTR(d_bins);
        SahSmallRoot* d_small = (SahSmallRoot*)dalloc((size_t)max_small * sizeof(SahSmallRoot)); BD_PTR(d_small);
        d_sah_state = (SahState*)dalloc(sizeof(SahState)); BD_PTR(d_sah_state);
        size_t scan_bytes = 0;
        BD_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, d_flag, d_pre, 0u, (size_t)n, rocprim::plus<unsigned int>(), st));
        void* d_scan_tmp = dalloc(scan_bytes); BD_PTR(d_scan_tmp);
        BD_HIP(hipMemsetAsync(d_parent, 0xff, 4, st)); /* the root has no parent */
        const int big_grid = (n + SAH_BLOCK - 1) / SAH_BLOCK;
        k_sah_begin<<<1, 1, 0, st>>>(d_sah_state);
        k_sah_root_bounds<<<big_grid, SAH_BLOCK, 0, st>>>(n, d_boxes, d_order[0], d_seg[0], d_sah_state);
        k_sah_root<<<1, 1, 0, st>>>(n, d_sah_state, d_act[0], d_small);
        BD_HIP(hipGetLastError());
        constexpr int SAH_MAX_LEVELS = 4096;
        const int clear_grid = (int)std::min<size_t>((bin_words + 255) / 256, 4096);
        int levels_run = 0;
        bool large_left = n > SAH_SMALL;
        while (large_left && levels_run < SAH_MAX_LEVELS)
        {
            const int level = levels_run, r = level & 1, w = r ^ 1;
            k_sah_clear_bins<<<clear_grid, 256, 0, st>>>(level, d_sah_state, d_bins, bin_words);
            k_sah_bin<<<big_grid, SAH_BLOCK, 0, st>>>(n, level, d_sah_state, d_act[r], d_boxes, d_order[r], d_seg[r], d_bins);
            k_sah_split<<<max_active, 64, 0, st>>>(level, d_sah_state, d_act[r], d_act[w], d_bins, d_splits, d_small, d_children, d_parent, d_node_boxes, max_active, max_small);
            k_sah_medium<<<max_active, SAH_BLOCK, 0, st>>>(level, d_sah_state, d_act[r], d_act[w], d_boxes, d_order[r], d_splits, d_small, d_children, d_parent, d_node_boxes, max_active, max_small);
            k_sah_classify<<<grid, 256, 0, st>>>(n, level, d_sah_state, d_splits, d_boxes, d_order[r], d_seg[r], d_flag);
            BD_HIP(rocprim::exclusive_scan(d_scan_tmp, scan_bytes, d_flag, d_pre, 0u, (size_t)n, rocprim::plus<unsigned int>(), st));
            k_sah_scatter<<<grid, 256, 0, st>>>(n, level, d_sah_state, d_splits, d_flag, d_pre, d_order[r], d_seg[r], d_order[w], d_seg[w]);
            k_sah_next_level<<<1, 1, 0, st>>>(level, d_sah_state);
            ++levels_run;
            if (levels_run >= 16 && (levels_run & 3) == 0)
            {
                SahState probe;
                BD_HIP(hipGetLastError());
                BD_HIP(hipMemcpyAsync(&probe, d_sah_state, sizeof(probe), hipMemcpyDeviceToHost, st));
                BD_HIP(hipStreamSynchronize(st));
                large_left = probe.n_active[levels_run & 1] != 0u;
            }
        }
        BD_HIP(hipGetLastError());
        uint32_t* d_final = d_order[levels_run & 1];
        k_sah_small<<<max_small, 64, 0, st>>>(d_sah_state, d_small, d_boxes, d_final, d_children, d_parent, d_node_boxes);
        BD_HIP(hipGetLastError());
        d_leaf_ids = d_final;
    }
#ifdef RT_EXPERIMENTS
    else
    {
    /* 3. Morton order */
    uint64_t* d_keys = (uint64_t*)dalloc((size_t)n * 8); BD_PTR(d_keys);
    uint64_t* d_keys2 = (uint64_t*)dalloc((size_t)n * 8); BD_PTR(d_keys2);
    uint32_t* d_ids = (uint32_t*)dalloc((size_t)n * 4); BD_PTR(d_ids);
    uint32_t* d_ids2 = (uint32_t*)dalloc((size_t)n * 4); BD_PTR(d_ids2);
    k_bvh_keys<<<grid, 256, 0, st>>>(d_boxes, n, slo, sext, d_keys, d_ids);
    BD_HIP(hipGetLastError());
    size_t rb = 0;
    BD_HIP(rocprim::radix_sort_pairs(nullptr, rb, d_keys, d_keys2, d_ids, d_ids2, (size_t)n, 0, 64, st));
    void* d_t2 = dalloc(rb); BD_PTR(d_t2);
    BD_HIP(rocprim::radix_sort_pairs(d_t2, rb, d_keys, d_keys2, d_ids, d_ids2, (size_t)n, 0, 64, st));
    /* 4. PLOC hierarchy */
    int* d_cid[2] = {(int*)dalloc((size_t)n * 4), (int*)dalloc((size_t)n * 4)}; BD_PTR(d_cid[0]); BD_PTR(d_cid[1]);
    float* d_cbox[2] = {(float*)dalloc((size_t)n * 24), (float*)dalloc((size_t)n * 24)}; BD_PTR(d_cbox[0]); BD_PTR(d_cbox[1]);
    int* d_nn = (int*)dalloc((size_t)n * 4); BD_PTR(d_nn);
    unsigned int* d_keep = (unsigned int*)dalloc((size_t)n * 4); BD_PTR(d_keep);
    unsigned int* d_pos = (unsigned int*)dalloc((size_t)n * 4); BD_PTR(d_pos);
    PlocState* d_ps = (PlocState*)dalloc(sizeof(PlocState)); BD_PTR(d_ps);
    size_t pb = 0;
    BD_HIP(rocprim::exclusive_scan(nullptr, pb, d_keep, d_pos, 0u, (size_t)n, rocprim::plus<unsigned int>(), st));
    void* d_t3 = dalloc(pb); BD_PTR(d_t3);
    k_ploc_init<<<grid, 256, 0, st>>>(n, d_ids2, d_boxes, d_cid[0], d_cbox[0], d_ps, d_parent);
    BD_HIP(hipGetLastError());
    int cur = 0;
    bool done = n <= RT_PLOC_TOP;
    PlocState ps = {(unsigned int)n, 0u};
    for (int batch = 0; batch < 256 && !done; ++batch)
    {
        /* 12 rounds between looks at the cluster count, 2 once the top is near */
        const int rounds = ps.m > 16u * RT_PLOC_TOP ? 12 : 2;
        for (int it = 0; it < rounds; ++it)
        {
            k_ploc_nn<<<grid, 256, 0, st>>>(d_ps, d_cbox[cur], d_nn, c->ploc_radius);
            k_ploc_merge<<<grid, 256, 0, st>>>(d_ps, n, d_cid[cur], d_cbox[cur], d_nn, d_children, d_parent, d_node_boxes, d_keep);
            BD_HIP(rocprim::exclusive_scan(d_t3, pb, d_keep, d_pos, 0u, (size_t)n, rocprim::plus<unsigned int>(), st));
            k_ploc_compact<<<grid, 256, 0, st>>>(d_ps, n, d_keep, d_pos, d_cid[cur], d_cbox[cur], d_cid[cur ^ 1], d_cbox[cur ^ 1]);
            cur ^= 1;
        }
        BD_HIP(hipGetLastError());
        BD_HIP(hipMemcpyAsync(&ps, d_ps, sizeof(ps), hipMemcpyDeviceToHost, st));
        BD_HIP(hipStreamSynchronize(st));
        done = ps.m <= (unsigned int)RT_PLOC_TOP;
    }
    if (!done) BD_FAIL(RT_ERR_BVH_DEPTH, "PLOC did not converge");
    if (ps.merges + ps.m != (unsigned int)n) BD_FAIL(RT_ERR_STATE, "internal: PLOC holds %u clusters after %u merges of %d references", ps.m, ps.merges, n);
    if (ps.m > 1u)
    {
        /* The top of the tree — the levels every ray walks — by an exact top-down SAH sweep over the <= RT_PLOC_TOP
         * clusters that are left, on the host (0.2 MB down, ~2 ms of host work, 0.3 MB up): mutual-nearest-neighbour
         * merging in a +-16 window judges the last rounds poorly (frame +5.8 % against the host SAH tree, +1.5 % with
         * this, r02). Node ids 0 .. m-2 are exactly the ones PLOC has not handed out (it counts down from n-2); the
         * root is 0. */
        const int m = (int)ps.m;
        std::vector<int> h_cid((size_t)m);
        std::vector<float> h_box((size_t)m * 6);
        BD_HIP(hipMemcpyAsync(h_cid.data(), d_cid[cur], (size_t)m * 4, hipMemcpyDeviceToHost, st));
        BD_HIP(hipMemcpyAsync(h_box.data(), d_cbox[cur], (size_t)m * 24, hipMemcpyDeviceToHost, st));
        BD_HIP(hipStreamSynchronize(st));
        std::vector<int2> t_children((size_t)m - 1);
        std::vector<float> t_boxes(((size_t)m - 1) * 6);
        std::vector<int> t_parent((size_t)m - 1, -1), s_idx, s_val;
        int next_id = 0;
        auto area = [](const float* b) { const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2]; return dx * dy + dy * dz + dz * dx; };
        auto grow = [](float* b, const float* o) { for (int k = 0; k < 3; ++k) { b[k] = fminf(b[k], o[k]); b[3 + k] = fmaxf(b[3 + k], o[3 + k]); } };
        auto centroid2 = [&](int item, int ax) { return h_box[(size_t)item * 6 + ax] + h_box[(size_t)item * 6 + 3 + ax]; };
        /* exact sweep SAH with three presorted item lists (by centroid per axis) kept in step by stable partitions:
         * O(m log m) in all; a range [lo, hi) holds the same items in all three lists */
        std::vector<int> sorted[3], scratch((size_t)m);
        for (int ax = 0; ax < 3; ++ax)
        {
            sorted[ax].resize((size_t)m);
            for (int i = 0; i < m; ++i) sorted[ax][(size_t)i] = i;
            std::sort(sorted[ax].begin(), sorted[ax].end(), [&](int x, int y) {
                const float cx = centroid2(x, ax), cy = centroid2(y, ax);
                return cx < cy || (cx == cy && x < y);
            });
        }
        std::vector<unsigned char> goes_left((size_t)m);
        std::vector<float> right_area((size_t)m);
        struct Job { int lo, hi, parent, side; };
        std::vector<Job> jobs;
        jobs.push_back({0, m, -1, 0});
        while (!jobs.empty())
        {
            const Job j = jobs.back();
            jobs.pop_back();
            int ref;
            if (j.hi - j.lo == 1)
            {
                ref = h_cid[(size_t)sorted[0][(size_t)j.lo]];
                if (ref >= 0) { s_idx.push_back(ref); s_val.push_back(j.parent); }
            }
            else
            {
                const int id = next_id++;
                ref = id;
                t_parent[(size_t)id] = j.parent;
                const int cnt = j.hi - j.lo;
                float best = INFINITY;
                int best_axis = 0, best_i = cnt / 2;
                float nb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                for (int ax = 0; ax < 3; ++ax)
                {
                    const int* L = sorted[ax].data() + j.lo;
                    float rb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    for (int i = cnt - 1; i >= 1; --i) { grow(rb, &h_box[(size_t)L[i] * 6]); right_area[(size_t)i] = area(rb); }
                    float lb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    for (int i = 1; i < cnt; ++i)
                    {
                        grow(lb, &h_box[(size_t)L[i - 1] * 6]);
                        const float cost = area(lb) * (float)i + right_area[(size_t)i] * (float)(cnt - i);
                        if (cost < best) { best = cost; best_axis = ax; best_i = i; }
                    }
                    if (ax == 0) { for (int k = 0; k < 6; ++k) nb[k] = lb[k]; grow(nb, &h_box[(size_t)L[cnt - 1] * 6]); }
                }
                for (int k = 0; k < 6; ++k) t_boxes[(size_t)id * 6 + k] = nb[k];
                /* the first best_i items of the best axis go left; the other two lists follow, order preserved */
                for (int i = 0; i < cnt; ++i) goes_left[(size_t)sorted[best_axis][(size_t)(j.lo + i)]] = i < best_i ? 1 : 0;
                for (int ax = 0; ax < 3; ++ax)
                {
                    if (ax == best_axis) continue;
                    int* L = sorted[ax].data() + j.lo;
                    int nl = 0, nr = 0;
                    for (int i = 0; i < cnt; ++i)
                    {
                        if (goes_left[(size_t)L[i]]) L[nl++] = L[i];
                        else scratch[(size_t)nr++] = L[i];
                    }
                    for (int i = 0; i < nr; ++i) L[nl + i] = scratch[(size_t)i];
                }
                jobs.push_back({j.lo, j.lo + best_i, id, 0});
                jobs.push_back({j.lo + best_i, j.hi, id, 1});
            }
            if (j.parent >= 0)
            {
                if (j.side == 0) t_children[(size_t)j.parent].x = ref; else t_children[(size_t)j.parent].y = ref;
            }
        }
        if (next_id != m - 1) BD_FAIL(RT_ERR_STATE, "internal: top-level build made %d nodes for %d clusters", next_id, m);
        BD_HIP(hipMemcpyAsync(d_children, t_children.data(), ((size_t)m - 1) * 8, hipMemcpyHostToDevice, st));
        BD_HIP(hipMemcpyAsync(d_node_boxes, t_boxes.data(), ((size_t)m - 1) * 24, hipMemcpyHostToDevice, st));
        BD_HIP(hipMemcpyAsync(d_parent, t_parent.data(), ((size_t)m - 1) * 4, hipMemcpyHostToDevice, st));
        if (!s_idx.empty())
        {
            int* d_si = (int*)dalloc(s_idx.size() * 4); BD_PTR(d_si);
            int* d_sv = (int*)dalloc(s_idx.size() * 4); BD_PTR(d_sv);
            BD_HIP(hipMemcpyAsync(d_si, s_idx.data(), s_idx.size() * 4, hipMemcpyHostToDevice, st));
            BD_HIP(hipMemcpyAsync(d_sv, s_val.data(), s_idx.size() * 4, hipMemcpyHostToDevice, st));
            k_scatter_int<<<((int)s_idx.size() + 255) / 256, 256, 0, st>>>((int)s_idx.size(), d_si, d_sv, d_parent);
            BD_HIP(hipGetLastError());
        }
        BD_HIP(hipStreamSynchronize(st)); /* the host vectors go out of scope */
    }
    d_leaf_ids = d_ids2;
    } /* builder 2 */
#else
    else BD_FAIL(RT_ERR_UNSUPPORTED, "BVH builder %d is an A/B form of librestir_rt_exp.so", c->bvh_builder);
#endif
    int* d_height = (int*)dalloc(4); BD_PTR(d_height);
    BD_HIP(hipMemsetAsync(d_height, 0, 4, st));
    k_bvh_height<<<grid, 256, 0, st>>>(n, d_children, d_parent, d_height);
    BD_HIP(hipMalloc(&c->d_nodes, (size_t)(n - 1) * sizeof(BvhNode)));
    k_bvh_emit<<<grid, 256, 0, st>>>(n, d_leaf_ids, d_ref_tri, d_boxes, d_children, d_parent, d_node_boxes, c->d_nodes);
    BD_HIP(hipGetLastError());
    /* 5. wide collapse, level by level */
    const size_t rec_cap = (size_t)n * 2 + 8;
    BD_HIP(hipMalloc(&c->d_wide, rec_cap * 16 * WIDE_STRIDE));
    CollapseItem* d_q[2] = {(CollapseItem*)dalloc((size_t)n * sizeof(CollapseItem)), (CollapseItem*)dalloc((size_t)n * sizeof(CollapseItem))};
    BD_PTR(d_q[0]); BD_PTR(d_q[1]);
    CollapseState* d_cs = (CollapseState*)dalloc(sizeof(CollapseState)); BD_PTR(d_cs);
    k_collapse_init<<<1, 1, 0, st>>>(d_cs, d_q[0]);
    const int max_levels = (WIDE_LDS_STACK + WIDE_OVF_STACK - 1) / 3 + 1;
    for (int level = 0; level <= max_levels; ++level)
    {
        k_collapse_level<<<grid, 256, 0, st>>>(c->d_nodes, c->d_tris, level, d_cs, d_q[level & 1], d_q[(level + 1) & 1], (uint32_t*)c->d_wide);
        k_collapse_swap<<<1, 1, 0, st>>>(level, d_cs);
    }
    BD_HIP(hipGetLastError());
    CollapseState cs;
    int bh = 0;
    SahState sah_final;
    memset(&sah_final, 0, sizeof(sah_final));
    BD_HIP(hipMemcpyAsync(&cs, d_cs, sizeof(cs), hipMemcpyDeviceToHost, st));
    BD_HIP(hipMemcpyAsync(&bh, d_height, 4, hipMemcpyDeviceToHost, st));
    if (d_sah_state) BD_HIP(hipMemcpyAsync(&sah_final, d_sah_state, sizeof(sah_final), hipMemcpyDeviceToHost, st));
    BD_HIP(hipStreamSynchronize(st));
    if (d_sah_state)
    {
        /* the counters that came back with the heights: every level reached the small-subtree size, no table overflowed,
         * and exactly n - 1 inner nodes were made */
        if (sah_final.n_active[0] != 0u || sah_final.n_active[1] != 0u)
            BD_FAIL(RT_ERR_BVH_DEPTH, "device SAH build: large nodes left after the enqueued levels (tree deeper than expected)");
        if (sah_final.overflow != 0u || sah_final.next_node != (unsigned int)(n - 1))
            BD_FAIL(RT_ERR_STATE, "internal: device SAH build made %u inner nodes for %d references (overflow %u)", sah_final.next_node, n, sah_final.overflow);
    }
    if (cs.count[0] != 0u || cs.count[1] != 0u || 3 * (int)cs.height + 1 > WIDE_LDS_STACK + WIDE_OVF_STACK)
        BD_FAIL(RT_ERR_BVH_DEPTH, "wide BVH height %u exceeds the traversal stack (%d entries)", cs.height, WIDE_LDS_STACK + WIDE_OVF_STACK);
    if (cs.n_rec > rec_cap) BD_FAIL(RT_ERR_STATE, "internal: %u wide records for %d references", cs.n_rec, n);
    if (bh > 62) BD_FAIL(RT_ERR_BVH_DEPTH, "binary tree height %d exceeds the 63-level trail word", bh);
    c->bvh_height = bh;
    c->n_wide = (int)cs.n_rec;
    c->wide_height = (int)cs.height;
    cleanup();
#undef BD_FAIL
#undef BD_HIP
#undef BD_PTR
    return RT_OK;
}

/* LBVH build, see bvh.h */
static int build_bvh(rt_ctx* c, const rt_triangle* tris, int n_tris)
{
    if (c->bvh_builder >= 2 && n_tris >= 2) return build_bvh_device(c, n_tris);
    hipStream_t st = c->stream;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    std::vector<float> extents((size_t)n_tris);
    for (int i = 0; i < n_tris; ++i)
    {
        float tl[3] = {INFINITY, INFINITY, INFINITY}, th[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < 3; ++k)
            for (int a = 0; a < 3; ++a)
            {
                const float v = tris[i].v[k][a];
                tl[a] = fminf(tl[a], v);
                th[a] = fmaxf(th[a], v);
            }
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], tl[a]); hi[a] = fmaxf(hi[a], th[a]); }
        extents[i] = fmaxf(th[0] - tl[0], fmaxf(th[1] - tl[1], th[2] - tl[2]));
    }
    float ext = 0.0f;
    for (int a = 0; a < 3; ++a) ext = fmaxf(ext, fmaxf(fabsf(lo[a]), fabsf(hi[a])));
    /* conservative padding: the accepted hit point of core.hpp:91-136 lies within rounding
     * distance of the triangle, not exactly on it */
    const float pad = 4e-5f * (ext > 1.0f ? ext : 1.0f);
    float3 slo = make_float3(lo[0], lo[1], lo[2]);
    float3 sext = make_float3(fmaxf(hi[0] - lo[0], 1e-20f), fmaxf(hi[1] - lo[1], 1e-20f), fmaxf(hi[2] - lo[2], 1e-20f));

    /* fragment length: a few median triangle extents, relaxed until the reference count fits */
    std::vector<BvhRef> refs;
    {
        std::vector<float> e2(extents);
        std::nth_element(e2.begin(), e2.begin() + e2.size() / 2, e2.end());
        float L = c->bvh_split_factor > 0.0f ? c->bvh_split_factor * e2[e2.size() / 2] : 0.0f;
        const size_t budget = (size_t)n_tris * 4 + 1024;
        for (int it = 0; it < 16; ++it)
        {
            split_refs(tris, n_tris, L, pad, refs);
            if (refs.size() <= budget || L <= 0.0f) break;
            L *= 1.5f;
        }
    }
    const int n = (int)refs.size();
    c->n_refs = n;
#ifdef RT_EXPERIMENTS
    if (c->bvh_builder == 1 && n >= 2)
    {
        SahBuilder sb(refs);
        float rl[3], rh[3];
        sb.build(0, n, -1, 1, rl, rh);
        sb.link_siblings();
        c->bvh_height = sb.height;
        if (sb.height > 62) RT_FAIL(c, RT_ERR_BVH_DEPTH, "SAH tree height %d exceeds the 63-level trail word", sb.height);
        RT_HIP(c, hipMalloc(&c->d_tv, (size_t)n_tris * 48));
        k_bvh_tv<<<(n_tris + 255) / 256, 256, 0, st>>>(c->d_tris, n_tris, c->d_tv);
        RT_HIP(c, hipGetLastError());
        RT_HIP(c, hipMalloc(&c->d_nodes, sb.nodes.size() * sizeof(BvhNode)));
        RT_HIP(c, hipMemcpyAsync(c->d_nodes, sb.nodes.data(), sb.nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice, st));
        RT_HIP(c, hipStreamSynchronize(st));
        return build_wide(c, tris, n);
    }
#endif
    std::vector<float> h_boxes((size_t)n * 6);
    std::vector<int> h_ref_tri((size_t)n);
    for (int i = 0; i < n; ++i)
    {
        for (int k = 0; k < 3; ++k) { h_boxes[6 * (size_t)i + k] = refs[i].lo[k]; h_boxes[6 * (size_t)i + 3 + k] = refs[i].hi[k]; }
        h_ref_tri[i] = refs[i].tri;
    }

    RT_HIP(c, hipMalloc(&c->d_tv, (size_t)n_tris * 48));
    RT_HIP(c, hipMalloc(&c->d_nodes, (size_t)(n > 1 ? n - 1 : 1) * sizeof(BvhNode)));
    float* d_boxes = nullptr; float* d_node_boxes = nullptr;
    uint64_t *d_keys = nullptr, *d_keys2 = nullptr;
    uint32_t *d_ids = nullptr, *d_ids2 = nullptr;
    int2* d_children = nullptr; int *d_parent_inner = nullptr, *d_parent_leaf = nullptr, *d_level = nullptr, *d_remaining = nullptr;
    int* d_ref_tri = nullptr;
    void* d_temp = nullptr;
    auto cleanup = [&]() {
        hipFree(d_boxes); hipFree(d_node_boxes); hipFree(d_keys); hipFree(d_keys2); hipFree(d_ids); hipFree(d_ids2);
        hipFree(d_children); hipFree(d_parent_inner); hipFree(d_parent_leaf); hipFree(d_level); hipFree(d_remaining);
        hipFree(d_temp); hipFree(d_ref_tri);
    };
#define BV_HIP(call)                                                                               \
    do                                                                                             \
    {                                                                                              \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess)                                                                      \
        {                                                                                          \
            char _b[512];                                                                          \
            snprintf(_b, sizeof(_b), "%s failed: %s", #call, hipGetErrorString(_e));               \
            c->err = _b; cleanup(); return RT_ERR_HIP;                                             \
        }                                                                                          \
    } while (0)
    BV_HIP(hipMalloc(&d_boxes, (size_t)n * 24));
    BV_HIP(hipMalloc(&d_ref_tri, (size_t)n * 4));
    BV_HIP(hipMalloc(&d_node_boxes, (size_t)n * 24));
    BV_HIP(hipMalloc(&d_keys, (size_t)n * 8));
    BV_HIP(hipMalloc(&d_keys2, (size_t)n * 8));
    BV_HIP(hipMalloc(&d_ids, (size_t)n * 4));
    BV_HIP(hipMalloc(&d_ids2, (size_t)n * 4));
    BV_HIP(hipMalloc(&d_children, (size_t)n * 8));
    BV_HIP(hipMalloc(&d_parent_inner, (size_t)n * 4));
    BV_HIP(hipMalloc(&d_parent_leaf, (size_t)n * 4));
    BV_HIP(hipMalloc(&d_level, (size_t)n * 4));
    BV_HIP(hipMalloc(&d_remaining, 4));
    BV_HIP(hipMemcpyAsync(d_boxes, h_boxes.data(), (size_t)n * 24, hipMemcpyHostToDevice, st));
    BV_HIP(hipMemcpyAsync(d_ref_tri, h_ref_tri.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));

    const int grid = (n + 255) / 256;
    k_bvh_tv<<<(n_tris + 255) / 256, 256, 0, st>>>(c->d_tris, n_tris, c->d_tv);
    BV_HIP(hipGetLastError());
    k_bvh_keys<<<grid, 256, 0, st>>>(d_boxes, n, slo, sext, d_keys, d_ids);
    BV_HIP(hipGetLastError());
    size_t temp_bytes = 0;
    BV_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys, d_keys2, d_ids, d_ids2, (size_t)n, 0, 64, st));
    BV_HIP(hipMalloc(&d_temp, temp_bytes > 0 ? temp_bytes : 16));
    BV_HIP(rocprim::radix_sort_pairs(d_temp, temp_bytes, d_keys, d_keys2, d_ids, d_ids2, (size_t)n, 0, 64, st));

    if (n == 1)
    {
        /* one reference: both children are the same leaf (the second test is rejected by the tie rule) */
        BvhNode nd;
        const float* hb = h_boxes.data();
        nd.a = make_float4(hb[0], hb[1], hb[2], hb[0]);
        nd.b = make_float4(hb[3], hb[4], hb[5], hb[1]);
        nd.c = make_float4(hb[3], hb[4], hb[5], hb[2]);
        nd.d = make_int4(~h_ref_tri[0], ~h_ref_tri[0], -1, -1);
        BV_HIP(hipMemcpyAsync(c->d_nodes, &nd, sizeof(nd), hipMemcpyHostToDevice, st));
        BV_HIP(hipStreamSynchronize(st));
        c->bvh_height = 1;
        cleanup();
        return build_wide(c, tris, n);
    }

    k_bvh_hierarchy<<<grid, 256, 0, st>>>(d_keys2, n, d_children, d_parent_inner, d_parent_leaf);
    BV_HIP(hipGetLastError());
    BV_HIP(hipMemsetAsync(d_level, 0, (size_t)n * 4, st));
    int height = 0;
    for (int pass = 1; pass <= 4096; ++pass)
    {
        BV_HIP(hipMemsetAsync(d_remaining, 0, 4, st));
        k_bvh_refit_pass<<<grid, 256, 0, st>>>(n, pass, d_ids2, d_boxes, d_children, d_node_boxes, d_level, d_remaining);
        BV_HIP(hipGetLastError());
        int remaining = 0;
        BV_HIP(hipMemcpyAsync(&remaining, d_remaining, 4, hipMemcpyDeviceToHost, st));
        BV_HIP(hipStreamSynchronize(st));
        if (remaining == 0) { height = pass; break; }
    }
    if (height == 0) { c->err = "LBVH refit did not converge"; cleanup(); return RT_ERR_BVH_DEPTH; }
    c->bvh_height = height;
    if (height > 62)
    {
        char b[128];
        snprintf(b, sizeof(b), "LBVH height %d exceeds the 63-level trail word", height);
        c->err = b; cleanup();
        return RT_ERR_BVH_DEPTH;
    }
    k_bvh_emit<<<grid, 256, 0, st>>>(n, d_ids2, d_ref_tri, d_boxes, d_children, d_parent_inner, d_node_boxes, c->d_nodes);
    BV_HIP(hipGetLastError());
    BV_HIP(hipStreamSynchronize(st));
    cleanup();
#undef BV_HIP
    return build_wide(c, tris, n);
}

int rt_scene_set(rt_ctx* c, const rt_triangle* triangles, uint32_t count)
{
    RT_CHECK_CTX(c);
    if (!triangles && count) RT_FAIL(c, RT_ERR_ARG, "null triangles");
    RT_HIP(c, hipSetDevice(c->device));
    /* every stream that may still read the old scene: the caller's / own stream, the speculative next-frame raycast
     * (spec_stream) and the second lane */
    { const int rs = rt_sync(c); if (rs != RT_OK) return rs; }
    if (c->own_stream && c->own_stream != c->stream) RT_HIP(c, hipStreamSynchronize(c->own_stream));
    if (c->aux_stream) RT_HIP(c, hipStreamSynchronize(c->aux_stream));
    c->spec_valid = false;
    const auto t_build0 = std::chrono::steady_clock::now();
    free_scene(c);
    ++c->epoch;
    const int n = (int)count;
    c->n_tris = n;
    /* light list in index order (10_restir_di.cpp:196-205) */
    std::vector<uint32_t> lights;
    for (int i = 0; i < n; ++i)
        if (triangles[i].emissive[0] > 0.0f || triangles[i].emissive[1] > 0.0f || triangles[i].emissive[2] > 0.0f)
            lights.push_back((uint32_t)i);
    if (lights.size() > ((size_t)1 << 26)) RT_FAIL(c, RT_ERR_UNSUPPORTED, "more than 2^26 emissive triangles (the light table is addressed by 32-bit byte offsets)");
    c->n_lights = (int)lights.size();
    c->bvh_height = 0;
    if (n == 0) { c->has_scene = true; return RT_OK; }
    RT_HIP(c, hipMalloc(&c->d_tris, (size_t)n * 60));
    RT_HIP(c, hipMemcpyAsync(c->d_tris, triangles, (size_t)n * 60, hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipMalloc(&c->d_trimat, (size_t)n * 32));
    k_trimat<<<(n + 255) / 256, 256, 0, c->stream>>>(n, c->d_tris, c->d_trimat);
    RT_HIP(c, hipGetLastError());
    if (c->n_lights > 0)
    {
        uint32_t* d_ids = nullptr;
        RT_HIP(c, hipMalloc(&d_ids, lights.size() * 4));
        RT_HIP(c, hipMemcpyAsync(d_ids, lights.data(), lights.size() * 4, hipMemcpyHostToDevice, c->stream));
        RT_HIP(c, hipMalloc(&c->d_lights, lights.size() * 16 * RT_LIGHT_STRIDE));
        RT_HIP(c, hipMalloc(&c->d_light_ke, lights.size() * 16));
        k_light_table<<<(c->n_lights + 255) / 256, 256, 0, c->stream>>>(c->n_lights, d_ids, c->d_tris, c->d_lights, c->d_light_ke);
        RT_HIP(c, hipGetLastError());
        RT_HIP(c, hipStreamSynchronize(c->stream));
        hipFree(d_ids);
    }
    const int rc = build_bvh(c, triangles, n);
    if (rc != RT_OK) return rc;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    c->build_ms = (float)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_build0).count() * 1e-3f;
    c->has_scene = true;
    return RT_OK;
}

int rt_scene_info(rt_ctx* c, uint32_t* n_triangles, uint32_t* n_lights, uint32_t* bvh_height)
{
    RT_CHECK_CTX(c);
    if (n_triangles) *n_triangles = (uint32_t)c->n_tris;
    if (n_lights) *n_lights = (uint32_t)c->n_lights;
    if (bvh_height) *bvh_height = (uint32_t)c->bvh_height;
    return RT_OK;
}
int rt_build_ms(rt_ctx* c, float* ms)
{
    RT_CHECK_CTX(c);
    if (!ms) return RT_ERR_ARG;
    *ms = c->build_ms;
    return RT_OK;
}
int rt_bvh_info(rt_ctx* c, uint32_t* n_refs, uint32_t* n_nodes, uint32_t* wide_height)
{
    RT_CHECK_CTX(c);
    if (wide_height) *wide_height = (uint32_t)c->wide_height;
    if (n_refs) *n_refs = (uint32_t)c->n_refs;
    if (n_nodes) *n_nodes = (uint32_t)c->n_wide;
    return RT_OK;
}

/* common/camera.hpp:11-25 (host; tan of a float argument = tanf) */
int rt_camera_lookat(rt_ctx* c, const float eye[3], const float center[3], const float up[3], float fovy)
{
    RT_CHECK_CTX(c);
    if (!eye || !center || !up) RT_FAIL(c, RT_ERR_ARG, "null camera vector");
    rt_host::raygen_lookat(&c->rg, eye, center, up, fovy, c->W, c->H); /* host_path.h: shared with the host-only config #1 */
    memcpy(c->eye, eye, 12);
    memcpy(c->cam_eye, eye, 12);
    memcpy(c->cam_at, center, 12);
    c->cam_fovy = fovy;
    c->has_camera = true;
    ++c->epoch;
    return RT_OK;
}
/* ---- interactive camera of the examples (common/misc.hpp:108-224 CameraControl): the mouse
 * callbacks restated as explicit calls; the window system stays out of scope. Each call updates
 * eye / look-at, re-derives the RayGenerator (fovy and up as last set) and raises the `updated`
 * flag that the frame loop turns into a `clear` (10_restir_di.cpp:257-267). ---- */
static void camera_refresh(rt_ctx* c)
{
    const float up[3] = {0.0f, 1.0f, 0.0f};
    rt_camera_lookat(c, c->cam_eye, c->cam_at, up, c->cam_fovy);
    c->cam_updated = true;
}
/* left-button drag by (dx, dy) pixels: orbit around the look-at point (misc.hpp:147-181) */
int rt_camera_orbit(rt_ctx* c, float dx, float dy)
{
    RT_CHECK_CTX(c);
    if (!c->has_camera) RT_FAIL(c, RT_ERR_STATE, "set the camera first");
    float lx = c->cam_eye[0] - c->cam_at[0], ly = c->cam_eye[1] - c->cam_at[1], lz = c->cam_eye[2] - c->cam_at[2];
    const float r = sqrtf(lx * lx + ly * ly + lz * lz);
    const float sensitivity = 0.004f;
    {
        const float st = sinf(dx * sensitivity), ct = cosf(dx * sensitivity);
        const float nx = ct * lx - st * lz, nz = st * lx + ct * lz;
        lx = nx; lz = nz;
    }
    {
        const float xz = sqrtf(lx * lx + lz * lz);
        const float st = sinf(dy * sensitivity), ct = cosf(dy * sensitivity);
        const float new_xz = ct * xz - st * ly, new_y = st * xz + ct * ly;
        if (-r + r * 0.01f < new_y && new_y < r - r * 0.01f)
        {
            lx = lx * (new_xz / xz); lz = lz * (new_xz / xz); ly = new_y;
        }
    }
    c->cam_eye[0] = c->cam_at[0] + lx; c->cam_eye[1] = c->cam_at[1] + ly; c->cam_eye[2] = c->cam_at[2] + lz;
    camera_refresh(c);
    return RT_OK;
}
/* right-button drag: dolly towards / away from the look-at point (misc.hpp:183-190) */
int rt_camera_zoom(rt_ctx* c, float dy)
{
    RT_CHECK_CTX(c);
    if (!c->has_camera) RT_FAIL(c, RT_ERR_STATE, "set the camera first");
    const float lx = c->cam_eye[0] - c->cam_at[0], ly = c->cam_eye[1] - c->cam_at[1], lz = c->cam_eye[2] - c->cam_at[2];
    const float r = sqrtf(lx * lx + ly * ly + lz * lz);
    const float sensitivity = 0.002f;
    const float new_r = fmaxf(r - r * sensitivity * dy, 0.01f);
    const float s = new_r / r;
    c->cam_eye[0] = c->cam_at[0] + lx * s; c->cam_eye[1] = c->cam_at[1] + ly * s; c->cam_eye[2] = c->cam_at[2] + lz * s;
    camera_refresh(c);
    return RT_OK;
}
/* middle-button drag: pan eye and look-at in the view plane (misc.hpp:192-205) */
int rt_camera_pan(rt_ctx* c, float dx, float dy)
{
    RT_CHECK_CTX(c);
    if (!c->has_camera) RT_FAIL(c, RT_ERR_STATE, "set the camera first");
    const f3 eye = F3(c->cam_eye[0], c->cam_eye[1], c->cam_eye[2]), at = F3(c->cam_at[0], c->cam_at[1], c->cam_at[2]);
    const float r = length(eye - at);
    const float sensitivity = 0.001f;
    const f3 forward = normalize(at - eye);
    const f3 right = normalize(cross(forward, F3(0.0f, 1.0f, 0.0f)));
    const f3 up = cross(right, forward);
    const float amount = fmaxf(r * sensitivity, 0.01f);
    const f3 delta = (-right) * dx * amount + up * dy * amount;
    const f3 e2 = eye + delta, a2 = at + delta;
    c->cam_eye[0] = e2.x; c->cam_eye[1] = e2.y; c->cam_eye[2] = e2.z;
    c->cam_at[0] = a2.x; c->cam_at[1] = a2.y; c->cam_at[2] = a2.z;
    camera_refresh(c);
    return RT_OK;
}
/* CameraControl::is_updated(): returns the flag and clears it */
int rt_camera_updated(rt_ctx* c, int* updated)
{
    RT_CHECK_CTX(c);
    if (!updated) return RT_ERR_ARG;
    *updated = c->cam_updated ? 1 : 0;
    c->cam_updated = false;
    return RT_OK;
}
int rt_camera_pose(rt_ctx* c, float eye[3], float lookat[3])
{
    RT_CHECK_CTX(c);
    if (eye) memcpy(eye, c->cam_eye, 12);
    if (lookat) memcpy(lookat, c->cam_at, 12);
    return RT_OK;
}

int rt_camera_set(rt_ctx* c, const rt_raygen* rg, const float eye[3])
{
    RT_CHECK_CTX(c);
    if (!rg || !eye) RT_FAIL(c, RT_ERR_ARG, "null raygen/eye");
    c->rg = *rg;
    c->eye[0] = eye[0]; c->eye[1] = eye[1]; c->eye[2] = eye[2];
    c->has_camera = true;
    ++c->epoch;
    return RT_OK;
}
int rt_camera_get(rt_ctx* c, rt_raygen* rg)
{
    RT_CHECK_CTX(c);
    if (!rg) return RT_ERR_ARG;
    *rg = c->rg;
    return RT_OK;
}
int rt_options_set(rt_ctx* c, const rt_options* o)
{
    RT_CHECK_CTX(c);
    if (!o) RT_FAIL(c, RT_ERR_ARG, "null options");
    if (o->ris_sample_count < 0 || o->spatial_resampling_sample_count < 0 || o->spatial_resampling_passes < 0)
        RT_FAIL(c, RT_ERR_ARG, "negative counts in options");
    {
        /* The 64-B record keeps M in 30 bits (rt_device.h; the reference's int M has 31). The largest M a
         * frame can reach is 21 * ris (candidates + the temporal cap of 20 * ris, 10_restir_di.cu:185-187)
         * times (1 + neighbours) per spatial pass: refuse option sets that could wrap instead of wrapping. */
        double m = 21.0 * (double)(o->ris_sample_count > 0 ? o->ris_sample_count : 1);
        for (int k = 0; k < o->spatial_resampling_passes && m < 4e9; ++k) m *= 1.0 + (double)o->spatial_resampling_sample_count;
        if (m >= 1073741824.0)
            RT_FAIL(c, RT_ERR_UNSUPPORTED, "options let the reservoir count M reach %.3g >= 2^30 (ris_sample_count %d, %d neighbours, %d passes)",
                    m, o->ris_sample_count, o->spatial_resampling_sample_count, o->spatial_resampling_passes);
    }
    c->opt = *o;
    ++c->epoch;
    return RT_OK;
}
int rt_options_get(rt_ctx* c, rt_options* o)
{
    RT_CHECK_CTX(c);
    if (!o) return RT_ERR_ARG;
    *o = c->opt;
    return RT_OK;
}

#define NEED_SCENE(c)                                                               \
    if (!(c)->has_scene || !(c)->has_camera) RT_FAIL(c, RT_ERR_STATE, "scene and camera must be set first");
#define NEED_RES(c, id) \
    if ((id) < 0 || (id) > 2) RT_FAIL(c, RT_ERR_ARG, "bad reservoir buffer id %d", (id));

static int launch_grid(const rt_ctx* c)
{
    return c->sub0 >= 0 ? tile_grid(c->W, c->sub1 - c->sub0, TILE_W, TILE_H, c->subb1 - c->subb0) : tile_grid(c->W, c->row_end - c->row_begin);
}
/* The work-sharing walk evens out the per-wavefront tail (longest lane 8x the mean: profiles/r02_wave_tail.txt). A launch
 * of many rounds of wavefronts hides that tail behind the next wavefronts and only pays the bookkeeping; a launch of
 * about one round (a strip of the multi-GPU frame: 135 rows x 1920 = 4050 wavefronts on 1024 SIMDs x 4-5) ends with
 * its slowest wavefront and runs ~2x faster with it. */
#ifndef RT_WS_AUTO_WAVES
#define RT_WS_AUTO_WAVES 12288
#endif
static bool use_ws(const rt_ctx* c, int grid) { return c->tune_ws < 0 ? grid <= RT_WS_AUTO_WAVES : c->tune_ws != 0; }
#ifndef RT_WS_PRIMARY_AUTO_WAVES
#define RT_WS_PRIMARY_AUTO_WAVES 8192
#endif
static bool use_ws_primary(const rt_ctx* c, int grid) { return c->tune_ws_primary < 0 ? grid <= RT_WS_PRIMARY_AUTO_WAVES : c->tune_ws_primary != 0; }
/* grid of the tracing kernels: TRACE_BLOCK threads on TileShape<TRACE_BLOCK> tiles */
/* priority of a side stream: 0 = default, +1 = lowest, -1 = highest (the range HIP reports), overridable for A/B runs */
static int stream_priority(const char* env, int dflt)
{
    const char* e = getenv(env);
    int want = e ? atoi(e) : dflt;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return 0;
    return want > 0 ? least : (want < 0 ? greatest : 0);
}
static bool use_tail(const rt_ctx* c)
{
    if (c->timing) return false; /* rt_timing brackets the kernels with events on the main stream */
    /* auto = always (r03: strips 0.442 -> 0.373 ms per frame at 1080p in 8 strips; a whole 1080p frame 1.68 -> 1.63 ms) */
    return c->tune_tail != 0;
}
/* the stream about to be used waits for the previous frame's resolve + tone_mapping (no-op unless they are in flight) */
static int join_tail(rt_ctx* c)
{
    if (!c->tail_pending_main) return RT_OK;
    RT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_tail, 0));
    /* the second lane of a stage waits for itself; the main stream still has to */
    if (c->stream != c->aux_stream) c->tail_pending_main = false;
    return RT_OK;
}
#define JOIN_TAIL(c) do { const int _jt = join_tail(c); if (_jt != RT_OK) return _jt; } while (0)
/* a kernel of the staged frame that WRITES reservoir buffer `phys` (and reads only what the tail reads too): it waits for the
 * tail only if that is the buffer the tail reads */
static int join_tail_for(rt_ctx* c, int phys) { return (c->tail_pending_main && phys == c->tail_phys) ? join_tail(c) : RT_OK; }
/* the pipelined stage 0 (spec_stream) may still be running when a call outside the staged frame — or a frame that does not
 * take its results — starts to write what it reads (the history buffer, the other G-buffer set) or reads what it writes: the
 * stream about to be used waits for it first (cheap: that work is behind already). ADVICE r03. */
static int join_spec(rt_ctx* c)
{
    if (!c->spec_outstanding) return RT_OK;
    RT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_spec_done, 0));
    if (c->stream != c->aux_stream) c->spec_outstanding = false;
    return RT_OK;
}
#define JOIN_SPEC(c) do { const int _js = join_spec(c); if (_js != RT_OK) return _js; } while (0)
static int trace_grid(const rt_ctx* c)
{
    const int rows = c->sub0 >= 0 ? c->sub1 - c->sub0 : c->row_end - c->row_begin;
    return tile_grid(c->W, rows, TileShape<TRACE_BLOCK>::W, TileShape<TRACE_BLOCK>::H, c->sub0 >= 0 ? c->subb1 - c->subb0 : 0);
}

int rt_clear(rt_ctx* c)
{
    RT_CHECK_CTX(c);
    JOIN_TAIL(c);
    k_clear<<<launch_grid(c), BLOCK, 0, c->stream>>>(make_params(c, 0, 0), c->d_accum);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

/* the raycast launch of the context's tuning, on `st`, into the given G-buffer set */
static void launch_raycast(rt_ctx* c, hipStream_t st, float4* vis, float4* g0, float4* g1)
{
#ifdef RT_EXPERIMENTS
    if (c->tune_half_raycast) { k_raycast_half<<<2 * trace_grid(c), TRACE_BLOCK, 0, st>>>(make_scene(c), make_params(c, 0, 0, K_RAYCAST), vis, g0, g1); return; }
#endif
#ifdef RT_EXPERIMENTS
    if (c->tune_ws_primary == 2) { k_raycast_quad<<<4 * trace_grid(c), TRACE_BLOCK, 0, st>>>(make_scene(c), make_params(c, 0, 0, K_RAYCAST), vis, g0, g1); return; }
#endif
    if (use_ws_primary(c, trace_grid(c))) k_raycast<true><<<trace_grid(c), TRACE_BLOCK, 0, st>>>(make_scene(c), make_params(c, 0, 0, K_RAYCAST), vis, g0, g1);
    else k_raycast<false><<<trace_grid(c), TRACE_BLOCK, 0, st>>>(make_scene(c), make_params(c, 0, 0, K_RAYCAST), vis, g0, g1);
}
int rt_raycast(rt_ctx* c)
{
    RT_CHECK_CTX(c);
    JOIN_TAIL(c);
    JOIN_SPEC(c);
    NEED_SCENE(c);
    launch_raycast(c, c->stream, c->d_vis, c->d_g0, c->d_g1);
    RT_HIP(c, hipGetLastError());
    c->has_gbuffer = true;
    ++c->gbuf_serial;
    c->gbuf_epoch = c->epoch;
    c->shaded_bits_stale = true;
    /* halo rows of the G-buffer keep the neighbours' shaded flags: they stay valid until the camera,
     * the scene or the options change (halo_flags_epoch), which is when the neighbours' G-buffers change */
    return RT_OK;
}

/* A/B r02 (profiles/r02_next_raycast_ab.txt): a strip's frame is a chain of small launches and exchanges with idle
 * slots the raycast fills (-6 % at 1080p in 8 strips, -3 % at 4K); a whole frame gains < 1 % at 1080p (the overlapped
 * spatial pass just waits for the machine) and its per-kernel times stop being comparable, so auto means strips only */
static bool use_next_raycast(const rt_ctx* c)
{
    if (c->tune_spec >= 0) return c->tune_spec != 0;
    /* auto: strips always; whole frames too (r03: with the candidates pipelined as well a 1080p frame gains 3 %), except
     * while rt_timing brackets the kernels of a frame with events — then they run back to back on one stream */
    return (c->row_begin != 0 || c->row_end != c->H) || !c->timing;
}
/* the next frame's generate_candidate(+temporal) too. Not while per-kernel timing is on (rt_timing attributes events of the
 * main stream to kernels) and not with the deferred-visibility queue (its counters live per lane of the main frame). */
static bool use_next_generate(const rt_ctx* c)
{
    if (!use_next_raycast(c) || c->timing || c->tune_defer_vis) return false;
    return c->tune_spec < 0 || c->tune_spec == 2;
}
/* r05, rt_tuning key 22. auto: strips — their kernels are launches of one to three generations of wavefronts and what fills the GPU
 * is running several of them side by side (rank 4 of 8: 1080p 0.306 -> 0.287 ms, 4K 0.889 -> 0.870, profiles/r05_spec_free_ab.txt);
 * a whole frame's kernels fill the GPU alone and only take slots from each other (pipelined 1080p frame 1.277 -> 1.296 ms) */
static bool spec_free(const rt_ctx* c) { return c->tune_spec_free < 0 ? (c->row_begin != 0 || c->row_end != c->H) : c->tune_spec_free != 0; }
static int launch_generate(rt_ctx* c, int frame, int dst_phys, int prev_phys, bool fuse, float4* raycast_vis = nullptr);
/* r05, rt_tuning key 25: raycast + generate_candidate (+ temporal merge) of a frame in ONE launch — the candidates need the primary ray
 * of their own pixel only, and two launches on a stream cost the first one's ramp-down (its last wavefront starts at 216 of 257 us).
 * The product's fused candidate kernel only (temporal merge on, unshadowed, work-sharing shadow walk), whole owned rows.
 * auto = whole-frame contexts (a strip's marks wait for the G-buffer alone). */
static bool use_fused_stage0(const rt_ctx* c)
{
    const bool whole = c->row_begin == 0 && c->row_end == c->H;
    const int want = c->tune_fuse_raycast < 0 ? (whole ? 1 : 0) : c->tune_fuse_raycast;
    /* r06: also while rt_timing brackets the kernels — the events then bracket the ONE launch the headline frames run
     * (ms[1] = the empty bracket where the raycast launch would be, ms[2] = the launch; rt_stage0_one_launch says which
     * form the timed frame ran). rt_tuning 25 = 0 times the two kernels. */
    if (!want) return false;
#ifdef RT_EXPERIMENTS
    if (c->tune_half_raycast || c->tune_defer_vis || c->tune_ris_pipe) return false;
#endif
    const int g = trace_grid(c);
    return c->opt.use_temporal_resampling && !c->opt.use_shadowed_target_function && c->n_lights > 0 && use_ws(c, g) && !use_ws_primary(c, g);
}
/* the next frame's primary rays, behind everything enqueued on the main stream so far, on the stream of their own */
static int launch_next_raycast(rt_ctx* c, int frame)
{
    c->spec_gen_valid = false;
    if (!use_next_raycast(c)) { c->spec_valid = false; return RT_OK; }
    const size_t n = local_pixels(c);
    const int o = (c->gcur + 1) % rt_ctx::NGSET;
    bool fresh_set = false;
    if (!c->d_gset[o][0])
    {
        for (int k = 0; k < 3; ++k) RT_HIP(c, hipMalloc(&c->d_gset[o][k], n * 16));
        /* halo rows of g1 hold the neighbours' shaded flags (rt_halo_flags_unpack keeps every set current from here on) */
        for (int k = 0; k < 3; ++k) RT_HIP(c, hipMemcpyAsync(c->d_gset[o][k], c->d_gset[c->gcur][k], n * 16, hipMemcpyDeviceToDevice, c->stream));
        fresh_set = true;
    }
    /* What this stage 0 needs from the main stream. If the current frame's stage 0 ran there (a cold frame, the first frames,
     * per-kernel timing), its candidates are the history read below: behind everything enqueued on the main stream so far. If the
     * current frame TOOK its stage 0 from this very stream (steady state), nothing: the history was written here, and the set /
     * buffer overwritten below were last read two resolves ago (r05) — the look-ahead runs on, beside whatever the main and the
     * tail stream are doing. (r04 waited here for the main stream, i.e. for the previous frame's passes, and for the previous
     * frame's resolve: raycast(f+2) could not start before resolve(f) had ended.) */
    if (!c->gen_taken || fresh_set || !spec_free(c))
    {
        RT_HIP(c, hipEventRecord(c->ev_spec_go, c->stream));
        RT_HIP(c, hipStreamWaitEvent(c->spec_stream, c->ev_spec_go, 0));
    }
    /* the G-buffer set and the spare reservoir buffer written below: last read by the resolve before the latest one (the set of
     * frame f-2; the buffer that left the roles when frame f-1 was taken) — or earlier */
    /* With ONE resolve on record the free-running form waits for it too (that it need not rested on the main-stream
     * wait above being forced at that point — one frame of overlap, once, buys an invariant that does not depend on it). */
    if (c->n_resolved >= 1)
        RT_HIP(c, hipStreamWaitEvent(c->spec_stream, c->ev_resolved[(spec_free(c) && c->n_resolved >= 2) ? 1 : 0], 0));
    /* the invariants the free-running look-ahead rests on, checked where it writes (cheap integer tests, every build): the set it
     * overwrites is not the one the running frame reads, and the candidates' buffer has no role in the running frame nor is it
     * what the tail in flight reads */
    if (o == c->gcur) RT_FAIL(c, RT_ERR_STATE, "look-ahead stage 0 would overwrite the running frame's G-buffer set %d", o);
    if (use_next_generate(c) && (c->spare == c->fX || c->spare == c->fY || c->spare == c->fZ || c->spare == c->quarantine ||
                                 (spec_free(c) && c->tail_pending_main && c->spare == c->tail_phys))) /* r04's dependencies hand the tail's buffer over and wait for the latest resolve above */
        RT_FAIL(c, RT_ERR_STATE, "look-ahead candidates' buffer %d is in use (X %d Y %d Z %d quarantine %d tail %d)", c->spare, c->fX, c->fY, c->fZ, c->quarantine, c->tail_pending_main ? c->tail_phys : -1);
    const int s0 = c->sub0, s1 = c->sub1, b0 = c->subb0, b1 = c->subb1;
    c->sub0 = c->sub1 = -1; c->subb0 = c->subb1 = 0; /* all owned rows */
    if (c->timing) hipEventRecord(c->ev_spec_t[o][0], c->spec_stream);
    const bool one_launch = use_next_generate(c) && use_fused_stage0(c); /* the candidates' launch below traces the primary rays too */
    if (!one_launch) launch_raycast(c, c->spec_stream, c->d_gset[o][0], c->d_gset[o][1], c->d_gset[o][2]);
    RT_HIP(c, hipGetLastError());
    if (c->timing) hipEventRecord(c->ev_spec_t[o][1], c->spec_stream);
    c->spec_timed[o] = c->timing;
    int rc = RT_OK;
    if (use_next_generate(c) && c->n_lights > 0)
    {
        /* candidates (+ temporal merge) of frame + 1: G-buffer = the set just traced, history = the buffer this frame's
         * stage 0 wrote (c->fY), output = the spare buffer; all owned rows, on the same stream behind the raycast */
        const size_t npx = local_pixels(c);
        if (!c->d_rec[c->spare])
        {
            RT_HIP(c, hipMalloc(&c->d_rec[c->spare], npx * 64));
            RT_HIP(c, hipMalloc(&c->d_rad[c->spare], npx * 16));
            RT_HIP(c, hipMemsetAsync(c->d_rec[c->spare], 0, npx * 64, c->spec_stream));
            RT_HIP(c, hipMemsetAsync(c->d_rad[c->spare], 0, npx * 16, c->spec_stream));
        }
        hipStream_t ms = c->stream;
        float4 *g0 = c->d_g0, *g1 = c->d_g1;
        c->stream = c->spec_stream; c->d_g0 = c->d_gset[o][1]; c->d_g1 = c->d_gset[o][2];
        const uint32_t tag_now = c->cur_tag;
        c->spec_gen_tag = next_ownv_tag(c); /* the frame that takes these candidates continues under their tag */
        c->cur_tag = c->spec_gen_tag;
        rc = launch_generate(c, frame + 1, c->spare, c->fY, c->opt.use_temporal_resampling != 0, one_launch ? c->d_gset[o][0] : nullptr);
        c->cur_tag = tag_now;
        c->stream = ms; c->d_g0 = g0; c->d_g1 = g1;
        if (rc == RT_OK)
        {
            c->spec_gen_valid = true;
            c->spec_gen_frame = frame + 1;
            c->spec_gen_hist = c->fY;
            c->spec_res_epoch = c->res_epoch;
        }
    }
    c->sub0 = s0; c->sub1 = s1; c->subb0 = b0; c->subb1 = b1;
    if (rc != RT_OK) return rc;
    RT_HIP(c, hipEventRecord(c->ev_spec_done, c->spec_stream));
    c->spec_outstanding = true;
    c->spec_valid = true;
    c->spec_epoch = c->epoch;
    return RT_OK;
}
/* stage 0's raycast over all owned rows: the G-buffer traced beside the previous frame if it is still the right one */
static int raycast_or_take(rt_ctx* c, bool whole, int frame, bool may_defer = false, bool* deferred = nullptr)
{
    c->timed_spec_set = -1;
    c->gen_taken = false;
    if (whole && use_next_raycast(c) && c->spec_valid && c->spec_epoch == c->epoch)
    {
        c->gcur = (c->gcur + 1) % rt_ctx::NGSET;
        c->d_vis = c->d_gset[c->gcur][0]; c->d_g0 = c->d_gset[c->gcur][1]; c->d_g1 = c->d_gset[c->gcur][2];
        RT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_spec_done, 0));
        c->spec_outstanding = false;
        if (c->spec_gen_valid && use_next_generate(c) && c->spec_gen_frame == frame && c->spec_res_epoch == c->res_epoch &&
            c->spec_gen_hist == c->fX)
        {
            /* this frame's candidates are in the spare buffer already: it becomes Y (stage 0's generate is skipped;
             * rt_frame_stage_output / _end see the swapped roles). Of the two buffers the passes ping-pong through, the one the
             * previous frame's resolve may still be reading on the tail stream (its final buffer) becomes the new spare — the
             * pipelined stage 0 after next waits for the tail anyway — and pass 0 writes the other one, so that no spatial pass
             * of this frame has to wait for the previous frame's resolve. With no spatial pass, logical RES_1 keeps its buffer
             * (the reference resolves a buffer that frame did not write: its content must carry over). */
            const int r0 = c->fY, r1 = c->fZ;
            const int prev_final = c->res_map[c->f_final == RT_RES_1 ? RT_RES_1 : RT_RES_0];
            c->fY = c->spare;
            int freed = r0;
            if (c->opt.spatial_resampling_passes >= 1 && prev_final == r1) { c->fZ = r0; freed = r1; }
            /* r05: the freed buffer (the previous frame's final one, or one nobody reads) waits a frame in quarantine; the look-ahead
             * candidates of the NEXT frame go to the buffer freed a frame earlier. RT_TUNING 22 = 0: the freed one at once (r04). */
            if (spec_free(c)) { c->spare = c->quarantine; c->quarantine = freed; }
            else c->spare = freed;
            c->f_in = c->fY; c->f_out = c->fZ;
            c->gen_taken = true;
            c->frame_tag = c->spec_gen_tag; /* own-visibility flags of the candidates were written under this tag */
            c->cur_tag = c->frame_tag;
        }
        c->spec_gen_valid = false;
        c->spec_valid = false;
        c->has_gbuffer = true;
        ++c->gbuf_serial;
        c->gbuf_epoch = c->spec_epoch; /* == c->epoch (checked above) */
        if (c->gen_taken) c->rec_gserial[c->fY] = c->gbuf_serial; /* the candidates were made from this G-buffer set */
        c->shaded_bits_stale = true;
        if (c->spec_timed[c->gcur]) c->timed_spec_set = c->gcur;
        return RT_OK;
    }
    c->spec_valid = false;
    c->spec_gen_valid = false;
    if (may_defer)
    {
        /* rt_raycast without its launch: the candidates' kernel of this stage traces the primary rays (use_fused_stage0) */
        JOIN_TAIL(c);
        JOIN_SPEC(c);
        c->has_gbuffer = true;
        ++c->gbuf_serial;
        c->gbuf_epoch = c->epoch;
        c->shaded_bits_stale = true;
        *deferred = true;
        return RT_OK;
    }
    return rt_raycast(c);
}

static int launch_generate(rt_ctx* c, int frame, int dst_phys, int prev_phys, bool fuse, float4* raycast_vis)
{
    if (c->n_lights == 0 && c->opt.ris_sample_count > 0)
        RT_FAIL(c, RT_ERR_STATE, "scene has no emissive triangle (the reference divides by zero here)");
    const SceneView S = make_scene(c);
    const FrameParams P = make_params(c, frame, 0, K_GENERATE);
    const bool sh = c->opt.use_shadowed_target_function;
    float4 *orec = c->d_rec[dst_phys], *orad = c->d_rad[dst_phys];
    c->rec_gserial[dst_phys] = c->gbuf_serial;
    const float4 *prec = fuse ? c->d_rec[prev_phys] : nullptr, *prad = fuse ? c->d_rad[prev_phys] : nullptr;
    const int g = trace_grid(c);
#ifdef RT_EXPERIMENTS
    if (fuse && !sh && c->tune_defer_vis)
    {
        /* fused + unshadowed: the visibility-reuse rays that survive the temporal merge go through a queue */
        const int lane = (c->stream == c->aux_stream) ? 1 : 0;
        const size_t npx = local_pixels(c);
        if (!c->d_visq[0])
        {
            RT_HIP(c, hipMalloc(&c->d_visq[0], npx * 4));
            RT_HIP(c, hipMalloc(&c->d_visq[1], npx * 4));
            RT_HIP(c, hipMalloc(&c->d_visq_count, 8));
            RT_HIP(c, hipMemsetAsync(c->d_visq_count, 0, 8, c->stream));
            RT_HIP(c, hipHostMalloc(&c->h_visq_count, 8, hipHostMallocDefault));
            c->h_visq_count[0] = c->h_visq_count[1] = 0xffffffffu;
        }
        if (c->visq_epoch != c->epoch) { c->h_visq_count[0] = c->h_visq_count[1] = 0xffffffffu; c->visq_epoch = c->epoch; }
        RT_HIP(c, hipMemsetAsync(c->d_visq_count + lane, 0, 4, c->stream));
        if (c->tune_ris_pipe) k_generate_candidate<true, false, true, true><<<g, TRACE_BLOCK, 0, c->stream>>>(S, P, c->d_g0, c->d_g1, prec, prad, orec, orad, c->d_visq[lane], c->d_visq_count + lane);
        else k_generate_candidate<true, false, true><<<g, TRACE_BLOCK, 0, c->stream>>>(S, P, c->d_g0, c->d_g1, prec, prad, orec, orad, c->d_visq[lane], c->d_visq_count + lane);
        RT_HIP(c, hipGetLastError());
        /* grid of the walk: one queue entry per lane if the count is like the last one that reached the host (+50 %),
         * every pixel of the launch if that is unknown (first frame, camera / option change); the kernel's stride loop
         * covers any count */
        const unsigned int last = c->h_visq_count[lane];
        int gv = g;
        if (last != 0xffffffffu)
        {
            const unsigned long long want = ((unsigned long long)last * 3ull / 2ull + TRACE_BLOCK - 1) / TRACE_BLOCK + 64ull;
            gv = want < (unsigned long long)g ? (int)want : g;
        }
        k_candidate_visibility<<<gv, TRACE_BLOCK, 0, c->stream>>>(S, c->d_g0, c->d_g1, orec, c->d_visq[lane], c->d_visq_count + lane);
        RT_HIP(c, hipGetLastError());
        RT_HIP(c, hipMemcpyAsync(c->h_visq_count + lane, c->d_visq_count + lane, 4, hipMemcpyDeviceToHost, c->stream));
        return RT_OK;
    }
#endif
    if (fuse && sh) k_generate_candidate<true, true><<<g, TRACE_BLOCK, 0, c->stream>>>(S, P, c->d_g0, c->d_g1, prec, prad, orec, orad);
#ifdef RT_EXPERIMENTS
    else if (fuse && c->tune_ris_pipe) k_generate_candidate<true, false, false, true><<<g, TRACE_BLOCK, 0, c->stream>>>(S, P, c->d_g0, c->d_g1, prec, prad, orec, orad);
#endif
    else if (fuse && use_ws(c, g) && raycast_vis)
        k_generate_candidate<true, false, false, false, true, true><<<g, TRACE_BLOCK, 0, c->stream>>>(S, P, nullptr, nullptr, prec, prad, orec, orad, nullptr, nullptr, raycast_vis, c->d_g0, c->d_g1);
    else if (raycast_vis) RT_FAIL(c, RT_ERR_STATE, "the one-launch stage 0 needs the product's fused candidate kernel");
    else if (fuse && use_ws(c, g)) k_generate_candidate<true, false, false, false, true><<<g, TRACE_BLOCK, 0, c->stream>>>(S, P, c->d_g0, c->d_g1, prec, prad, orec, orad);
    else if (fuse) k_generate_candidate<true, false><<<g, TRACE_BLOCK, 0, c->stream>>>(S, P, c->d_g0, c->d_g1, prec, prad, orec, orad);
    else if (sh) k_generate_candidate<false, true><<<g, TRACE_BLOCK, 0, c->stream>>>(S, P, c->d_g0, c->d_g1, prec, prad, orec, orad);
    else k_generate_candidate<false, false><<<g, TRACE_BLOCK, 0, c->stream>>>(S, P, c->d_g0, c->d_g1, prec, prad, orec, orad);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int rt_generate_candidate(rt_ctx* c, int frame, int dst)
{
    RT_CHECK_CTX(c);
    JOIN_TAIL(c);
    JOIN_SPEC(c);
    NEED_SCENE(c);
    NEED_RES(c, dst);
    ++c->res_epoch; /* a reservoir buffer changes outside the staged frame: a pipelined stage 0 is stale */
    if (!c->has_gbuffer) RT_FAIL(c, RT_ERR_STATE, "no G-buffer: call rt_raycast or upload RT_BUF_VISIBILITY first");
    return launch_generate(c, frame, c->res_map[dst], 0, false);
}

int rt_temporal_resampling(rt_ctx* c, int frame, int prev, int inout)
{
    RT_CHECK_CTX(c);
    JOIN_TAIL(c);
    JOIN_SPEC(c);
    NEED_SCENE(c);
    NEED_RES(c, prev);
    NEED_RES(c, inout);
    ++c->res_epoch; /* a reservoir buffer changes outside the staged frame: a pipelined stage 0 is stale */
    if (prev == inout) RT_FAIL(c, RT_ERR_ARG, "prev and inout must differ");
    const SceneView S = make_scene(c);
    const FrameParams P = make_params(c, frame, 0);
    const int pp = c->res_map[prev], pi = c->res_map[inout];
    if (c->opt.use_shadowed_target_function)
        k_temporal<true><<<launch_grid(c), BLOCK, 0, c->stream>>>(S, P, c->d_g0, c->d_g1, c->d_rec[pp], c->d_rad[pp], c->d_rec[pi], c->d_rad[pi]);
    else
        k_temporal<false><<<launch_grid(c), BLOCK, 0, c->stream>>>(S, P, c->d_g0, c->d_g1, c->d_rec[pp], c->d_rad[pp], c->d_rec[pi], c->d_rad[pi]);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

/* examples/10_restir_di/10_restir_di.cu:239-254: a plain copy of the owned rows */
int rt_save_temporal_reservoir(rt_ctx* c, int src, int dst)
{
    RT_CHECK_CTX(c);
    JOIN_TAIL(c);
    JOIN_SPEC(c);
    NEED_RES(c, src);
    NEED_RES(c, dst);
    ++c->res_epoch; /* a reservoir buffer changes outside the staged frame: a pipelined stage 0 is stale */
    if (src == dst) return RT_OK;
    const size_t off = (size_t)(c->row_begin - c->lrow0) * c->W;
    const size_t n = (size_t)(c->row_end - c->row_begin) * c->W;
    const int ps = c->res_map[src], pd = c->res_map[dst];
    c->rec_gserial[pd] = c->rec_gserial[ps];
    RT_HIP(c, hipMemcpyAsync(c->d_rec[pd] + 4 * off, c->d_rec[ps] + 4 * off, n * 64, hipMemcpyDeviceToDevice, c->stream));
    RT_HIP(c, hipMemcpyAsync(c->d_rad[pd] + off, c->d_rad[ps] + off, n * 16, hipMemcpyDeviceToDevice, c->stream));
    return RT_OK;
}

/* rows a strip must hold beyond its own for spatial_resampling to be exact: the Gaussian offset is
 * bounded by radius/1.96 * sqrt(2*23*ln 2) because rv0 >= 2^-23 when non-zero */
static int halo_rows_needed(const rt_options& o)
{
    if (!o.use_spatial_resampling || o.spatial_resampling_sample_count <= 0) return 0;
    return (int)ceilf(o.spatial_resampling_radius / 1.96f * 5.6471f);
}
#ifndef RT_FUSE_FINAL_AUTO
#define RT_FUSE_FINAL_AUTO 0 /* what rt_tuning key 23 = -1 means for a whole-frame context, until measured */
#endif
#ifndef RT_SHADOWED_SPATIAL_LDS
#define RT_SHADOWED_SPATIAL_LDS 0 /* the walk's own LDS stack already limits the shadowed variant to 6 workgroups per CU */
#endif
/* the LDS-staged variant covers the default reach (87 px) and up to 5 neighbours. Whole-frame contexts only: the
 * kernel itself handles strips (tile origins by row range, halo rows' bits from the neighbours' flags), but there it is
 * no faster than the gather kernel (A/B r02: 0.517 / 0.517 ms per frame at 1080p in 8 strips, 1.308 / 1.289 at 4K) and
 * it adds the bitmap launch to every frame */
static bool use_lds_spatial(const rt_ctx* c)
{
    return (c->tune_spatial_variant == 1 || c->tune_spatial_variant == 3) && c->opt.use_spatial_resampling && !c->opt.use_shadowed_target_function &&
           halo_rows_needed(c->opt) <= SPL_HALO && c->opt.spatial_resampling_sample_count <= 5 &&
           c->row_begin == 0 && c->row_end == c->H;
}
/* shaded bit per pixel of all local rows, rebuilt on the context's stream when the G-buffer or the halo flags changed.
 * rt_frame_stage does this right behind the raycast, on the main stream, so that a stage's two lanes only read it. */
static int refresh_shaded_bits(rt_ctx* c)
{
    if (!c->shaded_bits_stale || !c->has_gbuffer || !use_lds_spatial(c)) return RT_OK;
    const int words = (c->W + 31) / 32;
    if (!c->d_shaded_bits) RT_HIP(c, hipMalloc(&c->d_shaded_bits, (size_t)c->lrows * words * 4));
    k_shaded_bitmap<<<dim3((c->W + 255) / 256, c->lrows), 256, 0, c->stream>>>(c->W, c->lrows, c->d_g1, c->d_shaded_bits);
    RT_HIP(c, hipGetLastError());
    c->shaded_bits_stale = false;
    return RT_OK;
}
/* register budget of the gather kernel when rt_tuning key 9 is -1. xnack-any code (until late r02) wanted 4 wavefronts per
 * SIMD to keep the neighbour window in L2 (0.184 against 0.202 ms unbounded); the xnack- build is flat across 4 / 5 / 6 /
 * unbounded (0.170 / 0.169 / 0.168 / 0.170 ms), 6 a hair ahead */
#ifndef RT_SPATIAL_GATHER_AUTO_WAVES
#define RT_SPATIAL_GATHER_AUTO_WAVES 6
#endif
#ifndef RT_SPATIAL_PIPE_AUTO_WAVES
#define RT_SPATIAL_PIPE_AUTO_WAVES 5 /* 16 more registers in flight than k_spatial_coop (the staged record parts) */
#endif
/* r05, rt_tuning key 23: the frame's last spatial pass shades its pixels itself (k_spatial_resolve). auto = whole-frame contexts;
 * the unshadowed cooperative kernel only (the shadowed pass spends its time in its own rays). */
static bool use_fused_final(const rt_ctx* c)
{
#ifndef RT_EXPERIMENTS
    return false; /* k_spatial_resolve: measured slower (profiles/r05_fused_tail_ab.txt), experiments build only */
#endif
    const bool whole = c->row_begin == 0 && c->row_end == c->H;
    const int want = c->tune_fuse_final < 0 ? (whole ? RT_FUSE_FINAL_AUTO : 0) : c->tune_fuse_final;
    return want != 0 && !c->opt.use_shadowed_target_function && c->opt.use_spatial_resampling && c->tune_spatial_variant == 2 && !c->tune_stream &&
           c->opt.spatial_resampling_passes >= 1;
}
static int launch_spatial(rt_ctx* c, int frame, int pass, int in_phys, int out_phys, bool fuse_final = false)
{
    const int need = halo_rows_needed(c->opt);
    if ((c->row_begin > 0 && c->row_begin - c->lrow0 < (need < c->row_begin ? need : c->row_begin)) ||
        (c->row_end < c->H && c->lrow0 + c->lrows - c->row_end < (need < c->H - c->row_end ? need : c->H - c->row_end)))
        RT_FAIL(c, RT_ERR_STATE, "strip halo of %d rows is too small: spatial_resampling_radius %.1f needs %d", c->halo,
                c->opt.spatial_resampling_radius, need);
    const SceneView S = make_scene(c);
    const FrameParams P = make_params(c, frame, pass, K_SPATIAL);
    const bool lds_variant = use_lds_spatial(c);
    (void)lds_variant; /* the product build has no LDS-staged form */
    c->rec_gserial[out_phys] = c->gbuf_serial;
    if (c->opt.use_shadowed_target_function)
    {
        /* the cooperative record traffic of key 8 = 2 applies to the <= 5 neighbour form (spatial_wave_shadowed) */
        if (c->tune_spatial_variant >= 2 && c->opt.use_spatial_resampling && c->opt.spatial_resampling_sample_count <= 5)
            k_spatial<true, true><<<trace_grid(c), TRACE_BLOCK, (size_t)(RT_SHADOWED_SPATIAL_LDS), c->stream>>>(S, P, c->fuse, c->d_g0, c->d_g1, c->d_rec[in_phys], c->d_rad[in_phys], c->d_rec[out_phys], c->d_rad[out_phys]);
        else
            k_spatial<true, false><<<trace_grid(c), TRACE_BLOCK, (size_t)(RT_SHADOWED_SPATIAL_LDS), c->stream>>>(S, P, c->fuse, c->d_g0, c->d_g1, c->d_rec[in_phys], c->d_rad[in_phys], c->d_rec[out_phys], c->d_rad[out_phys]);
    }
#ifdef RT_EXPERIMENTS
    else if (lds_variant && c->tune_spatial_variant == 3)
    {
        /* software-pipelined cooperative kernel (r04): staged shaded bits + the next neighbour's record in flight */
        const int rc = refresh_shaded_bits(c);
        if (rc != RT_OK) return rc;
#define RT_SPP(WV) k_spatial_pipe<WV><<<launch_grid(c), BLOCK, (size_t)c->tune_spatial_lds, c->stream>>>(P, c->d_shaded_bits, c->d_g0, c->d_g1, c->d_rec[in_phys], c->d_rad[in_phys], c->d_rec[out_phys], c->d_rad[out_phys])
        switch (c->tune_spatial_waves < 0 ? RT_SPATIAL_PIPE_AUTO_WAVES : c->tune_spatial_waves) { case 4: RT_SPP(4); break; case 5: RT_SPP(5); break; case 6: RT_SPP(6); break; default: RT_SPP(0); break; }
#undef RT_SPP
    }
    else if (lds_variant)
    {
        const int rc = refresh_shaded_bits(c); /* no-op inside rt_frame / rt_frame_stage: done behind the raycast */
        if (rc != RT_OK) return rc;
#define RT_SPL(WV) k_spatial_lds<WV><<<launch_grid(c), BLOCK, (size_t)c->tune_spatial_lds, c->stream>>>(P, c->d_shaded_bits, c->d_g0, c->d_g1, c->d_rec[in_phys], c->d_rad[in_phys], c->d_rec[out_phys], c->d_rad[out_phys])
        switch (c->tune_spatial_waves) { case 4: RT_SPL(4); break; case 5: RT_SPL(5); break; case 6: RT_SPL(6); break; default: RT_SPL(0); break; }
#undef RT_SPL
    }
#endif
#ifdef RT_EXPERIMENTS
    else if (fuse_final)
    {
        /* the last pass + resolve (+ tone mapping) of the staged frame in one launch of one-wavefront workgroups */
        const bool fused = c->fuse.recv[0] || c->fuse.recv[1] || c->fuse.send[0] || c->fuse.send[1];
        const bool store = c->tune_fuse_final != 2; /* 2: A/B only, the pass's output buffer is NOT written */
        uint32_t* px = c->tune_fuse_tonemap ? (uint32_t*)c->d_pixels : nullptr;
#define RT_SPR(FU, ST) k_spatial_resolve<FU, ST><<<trace_grid(c), TRACE_BLOCK, 0, c->stream>>>(S, P, c->fuse, c->d_g0, c->d_g1, c->d_rec[in_phys], c->d_rad[in_phys], c->d_rec[out_phys], c->d_rad[out_phys], c->d_accum, px)
        if (fused) { if (store) RT_SPR(true, true); else RT_SPR(true, false); }
        else { if (store) RT_SPR(false, true); else RT_SPR(false, false); }
#undef RT_SPR
        c->final_fused = true;
    }
#endif
#ifdef RT_EXPERIMENTS
    else if (c->tune_spatial_variant == 4)
    {
        /* r05, A/B only: k_spatial_coop as one-wavefront workgroups on 8 x 8 tiles (a finished wavefront's slot is refilled at once; a
         * 256-thread workgroup's four slots wait for four free slots on one CU: 5 400 of 6 144 slots filled on average). More
         * wavefronts in flight (5 750), each slower (25.9 -> 27.9 us): the pass is bound by its misses in flight, +1.3 % */
        const bool fused = c->fuse.recv[0] || c->fuse.recv[1] || c->fuse.send[0] || c->fuse.send[1];
#define RT_SPC1(FU) k_spatial_coop<RT_SPATIAL_GATHER_AUTO_WAVES, FU, TRACE_BLOCK><<<trace_grid(c), TRACE_BLOCK, (size_t)c->tune_spatial_lds, c->stream>>>(S, P, c->fuse, c->d_g0, c->d_g1, c->d_rec[in_phys], c->d_rad[in_phys], c->d_rec[out_phys], c->d_rad[out_phys])
        if (fused) RT_SPC1(true); else RT_SPC1(false);
#undef RT_SPC1
    }
#endif
    else if (c->tune_spatial_variant == 2 || c->tune_spatial_variant == 3) /* 3 where the pipelined kernel does not apply (strips, radius > 30, > 5 neighbours) */
    {
#define RT_SPC2(WV, FU) k_spatial_coop<WV, FU><<<launch_grid(c), BLOCK, (size_t)c->tune_spatial_lds, c->stream>>>(S, P, c->fuse, c->d_g0, c->d_g1, c->d_rec[in_phys], c->d_rad[in_phys], c->d_rec[out_phys], c->d_rad[out_phys])
#define RT_SPC(WV) do { if (fused) RT_SPC2(WV, true); else RT_SPC2(WV, false); } while (0)
        const bool fused = c->fuse.recv[0] || c->fuse.recv[1] || c->fuse.send[0] || c->fuse.send[1];
#ifdef RT_EXPERIMENTS
        switch (c->tune_spatial_waves < 0 ? RT_SPATIAL_GATHER_AUTO_WAVES : c->tune_spatial_waves) { case 4: RT_SPC(4); break; case 5: RT_SPC(5); break; case 6: RT_SPC(6); break; default: RT_SPC(0); break; }
#else
        RT_SPC(RT_SPATIAL_GATHER_AUTO_WAVES); /* the product library carries the one register budget it uses (rt_tuning key 9: experiments build) */
#endif
#undef RT_SPC2
#undef RT_SPC
    }
#ifdef RT_EXPERIMENTS
    else
    {
#define RT_SPG(WV) k_spatial_gather<WV><<<launch_grid(c), BLOCK, (size_t)c->tune_spatial_lds, c->stream>>>(S, P, c->fuse, c->d_g0, c->d_g1, c->d_rec[in_phys], c->d_rad[in_phys], c->d_rec[out_phys], c->d_rad[out_phys])
        switch (c->tune_spatial_waves < 0 ? RT_SPATIAL_GATHER_AUTO_WAVES : c->tune_spatial_waves) { case 4: RT_SPG(4); break; case 5: RT_SPG(5); break; case 6: RT_SPG(6); break; default: RT_SPG(0); break; }
#undef RT_SPG
    }
#else
    else RT_FAIL(c, RT_ERR_UNSUPPORTED, "spatial variant %d is an A/B form of librestir_rt_exp.so", c->tune_spatial_variant);
#endif
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int rt_spatial_resampling(rt_ctx* c, int frame, int pass, int in, int out)
{
    RT_CHECK_CTX(c);
    JOIN_TAIL(c);
    JOIN_SPEC(c);
    NEED_SCENE(c);
    NEED_RES(c, in);
    NEED_RES(c, out);
    ++c->res_epoch; /* a reservoir buffer changes outside the staged frame: a pipelined stage 0 is stale */
    if (in == out) RT_FAIL(c, RT_ERR_ARG, "in and out must differ");
    c->last_frame = frame;
    c->fuse = HaloFuse{};
    if (c->rec_gserial[c->res_map[in]] != c->gbuf_serial)
    {
        /* the G-buffer changed since `in` was written: its neighbour test must see the current shaded flags (frame_kernels.h) */
        /* own rows always; a strip's halo rows only where the neighbour's flags in g1 are of the current epoch (rt_halo_flags_unpack
         * after the last camera / scene / option change): stale flags must not rewrite the bits of the neighbour's records */
        const int first_row = (c->row_begin > c->lrow0 && c->halo_flags_epoch[0] != c->epoch) ? c->row_begin : c->lrow0;
        const int last_row = (c->lrow0 + c->lrows > c->row_end && c->halo_flags_epoch[1] != c->epoch) ? c->row_end : c->lrow0 + c->lrows;
        const size_t off = (size_t)(first_row - c->lrow0) * c->W;
        const int n = (last_row - first_row) * c->W;
        k_refresh_shaded<<<(n + 255) / 256, 256, 0, c->stream>>>(n, c->d_g1 + off, c->d_rec[c->res_map[in]] + 4 * off);
        RT_HIP(c, hipGetLastError());
        c->rec_gserial[c->res_map[in]] = c->gbuf_serial;
    }
    return launch_spatial(c, frame, pass, c->res_map[in], c->res_map[out]);
}

/* pixels: the staged frame maps each pixel in the resolve kernel itself (rt_tuning key 20, default on); NULL = resolve alone */
static int launch_resolve(rt_ctx* c, int phys, uint32_t* pixels = nullptr)
{
    const int g = trace_grid(c);
#ifdef RT_EXPERIMENTS
    if (c->tune_stream)
    {
        /* persistent wavefronts, tiles dealt out round-robin (no job counter) */
        const int resident = c->n_cus * 4 * RT_RESOLVE_STREAM_WAVES; /* wavefronts the GPU holds at once: a multiple of 8 */
        const int wgs = g < resident ? g : resident;
        k_resolve_stream<<<wgs, TRACE_BLOCK, 0, c->stream>>>(make_scene(c), make_params(c, 0, 0, K_RESOLVE), c->d_g0, c->d_g1, c->d_rec[phys], c->d_rad[phys], c->d_accum, g);
    }
    else
#endif
    if (use_ws(c, g)) k_resolve<true><<<g, TRACE_BLOCK, 0, c->stream>>>(make_scene(c), make_params(c, 0, 0, K_RESOLVE), c->d_g0, c->d_g1, c->d_rec[phys], c->d_rad[phys], c->d_accum, pixels);
    else k_resolve<false><<<g, TRACE_BLOCK, 0, c->stream>>>(make_scene(c), make_params(c, 0, 0, K_RESOLVE), c->d_g0, c->d_g1, c->d_rec[phys], c->d_rad[phys], c->d_accum, pixels);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}
int rt_resolve(rt_ctx* c, int res)
{
    RT_CHECK_CTX(c);
    JOIN_TAIL(c);
    NEED_SCENE(c);
    NEED_RES(c, res);
    return launch_resolve(c, c->res_map[res]);
}

static int launch_tone_mapping(rt_ctx* c)
{
    k_tone_mapping<<<launch_grid(c), BLOCK, 0, c->stream>>>(make_params(c, 0, 0), c->d_accum, c->d_pixels);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}
int rt_tone_mapping(rt_ctx* c)
{
    RT_CHECK_CTX(c);
    JOIN_TAIL(c);
    return launch_tone_mapping(c);
}

/* `path_trace` of examples/07_pt/07_pt.cu (example 7), examples/08_nee/08_nee.cu (8) or examples/09_ris/09_ris.cu (9) */
int rt_path_trace(rt_ctx* c, int example, int frame)
{
    RT_CHECK_CTX(c);
    JOIN_TAIL(c);
    NEED_SCENE(c);
    if (example != 7 && example != 8 && example != 9) RT_FAIL(c, RT_ERR_ARG, "example must be 7 (07_pt), 8 (08_nee) or 9 (09_ris)");
    if (example != 7 && c->n_lights == 0) RT_FAIL(c, RT_ERR_STATE, "08_nee / 09_ris need at least one emissive triangle");
    const SceneView S = make_scene(c);
    FrameParams P = make_params(c, frame, 0, K_RAYCAST);
    const f3 sky = F3(c->opt.sky_color[0], c->opt.sky_color[1], c->opt.sky_color[2]);
    const int g = trace_grid(c);
    const int md = c->opt.max_depth;
    const bool sh = c->opt.use_shadowed_target_function != 0;
    const bool wavefront = c->pt_wavefront == 1 || (c->pt_wavefront == 2 && example == 9);
    /* one launch per bounce: the first launch's tile order is the order of the path list every later bounce walks — an XCD's
     * band of the image keeps neighbouring paths on one XCD (interleaved: +2 %); one launch per frame gains 5-10 % interleaved
     * like the frame's tracing kernels (profiles/r05_tile_interleave_ab.txt) */
    if (wavefront && c->tune_tile_mode[K_RAYCAST] < 0) P.tile_mode = 1;
    if (wavefront && md > 0 && md <= 60)
    {
        /* one launch per bounce over the compacted list of live paths */
        const size_t n = (size_t)c->W * (size_t)(c->row_end - c->row_begin);
        if (!c->d_paths[0])
        {
            RT_HIP(c, hipMalloc(&c->d_paths[0], n * 64));
            RT_HIP(c, hipMalloc(&c->d_paths[1], n * 64));
            RT_HIP(c, hipMalloc(&c->d_pt_counters, 64 * 8));
        }
        RT_HIP(c, hipMemsetAsync(c->d_pt_counters, 0, 64 * 8, c->stream));
        k_pt_init<<<g, TRACE_BLOCK, 0, c->stream>>>(P, c->d_paths[0], c->d_pt_counters);
        RT_HIP(c, hipGetLastError());
        const int gb = (int)((n + TRACE_BLOCK - 1) / TRACE_BLOCK);
        for (int d = 0; d < md; ++d)
        {
            const float4* in = c->d_paths[d & 1];
            float4* out = c->d_paths[(d + 1) & 1];
            if (example == 7) k_pt_bounce<7, false><<<gb, TRACE_BLOCK, 0, c->stream>>>(S, P, d, md, sky, in, out, c->d_accum, c->d_pt_counters);
            else if (example == 8) k_pt_bounce<8, false><<<gb, TRACE_BLOCK, 0, c->stream>>>(S, P, d, md, sky, in, out, c->d_accum, c->d_pt_counters);
            else if (sh) k_pt_bounce<9, true><<<gb, TRACE_BLOCK, 0, c->stream>>>(S, P, d, md, sky, in, out, c->d_accum, c->d_pt_counters);
            else k_pt_bounce<9, false><<<gb, TRACE_BLOCK, 0, c->stream>>>(S, P, d, md, sky, in, out, c->d_accum, c->d_pt_counters);
            RT_HIP(c, hipGetLastError());
        }
        RT_HIP(c, hipMemcpyAsync(c->d_counter, c->d_pt_counters, 8, hipMemcpyDeviceToDevice, c->stream));
        return RT_OK;
    }
    RT_HIP(c, hipMemsetAsync(c->d_counter, 0, 8, c->stream));
    if (example == 7) k_path_trace<7, false><<<launch_grid(c), BLOCK, 0, c->stream>>>(S, P, md, sky, c->d_accum, c->d_counter);
    else if (example == 8) k_path_trace<8, false><<<g, TRACE_BLOCK, 0, c->stream>>>(S, P, md, sky, c->d_accum, c->d_counter);
    else if (sh) k_path_trace<9, true><<<g, TRACE_BLOCK, 0, c->stream>>>(S, P, md, sky, c->d_accum, c->d_counter);
    else k_path_trace<9, false><<<g, TRACE_BLOCK, 0, c->stream>>>(S, P, md, sky, c->d_accum, c->d_counter);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}
/* raytrace() calls made by the last rt_path_trace (synchronises the stream) */
int rt_path_trace_rays(rt_ctx* c, uint64_t* rays)
{
    RT_CHECK_CTX(c);
    if (!rays) return RT_ERR_ARG;
    unsigned long long h = 0;
    RT_HIP(c, hipMemcpyAsync(&h, c->d_counter, 8, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    *rays = h;
    return RT_OK;
}

int rt_timing_enable(rt_ctx* c, int on)
{
    RT_CHECK_CTX(c);
    if (on && !c->ev_created)
    {
        for (auto& e : c->ev) RT_HIP(c, hipEventCreate(&e));
        c->ev_created = true;
    }
    c->timing = on != 0;
    c->last_valid = false;
    return RT_OK;
}

/* The frame as a sequence of stages, so that a strip context can exchange halos in between:
 *   stage 0            [clear] raycast, generate_candidate(+temporal) into the rotating buffers
 *   stage 1..passes    spatial pass stage-1      (its input buffer must have valid halos)
 *   stage passes+1     resolve, tone_mapping, buffer renaming
 * rt_frame runs them back to back. */
/* begin: buffer roles of the stage; run: its kernels over storage rows [row0,row1) of the owned
 * rows (several runs per stage allowed: boundary rows first, interior later, so that halos can
 * travel while the interior is computed); end: advance. */
int rt_frame_stage_begin(rt_ctx* c, int frame, int stage, int clear_first)
{
    RT_CHECK_CTX(c);
    NEED_SCENE(c);
    c->last_frame = frame;
    /* a spatial stage reads the shaded-bit rows from both lanes: (re)built here, on the main stream, if the halo flags
     * arrived after the raycast (cold frame of a strip) */
    if (stage >= 1) { const int rc = refresh_shaded_bits(c); if (rc != RT_OK) return rc; }
    RT_HIP(c, hipEventRecord(c->ev_stage, c->stream)); /* everything the stage reads is complete here */
    c->aux_used = false;
    c->fuse = HaloFuse{};
    const int passes = c->opt.spatial_resampling_passes;
    if (stage == 0)
    {
        /* X = history, Y = candidates(+temporal) -> next history, Z = spatial ping-pong partner */
        c->fX = c->res_map[RT_RES_TEMPORAL]; c->fY = c->res_map[RT_RES_0]; c->fZ = c->res_map[RT_RES_1];
        c->f_in = c->fY; c->f_out = c->fZ;
        c->f_clear = clear_first != 0;
        c->f_stage = 0;
        c->gen_taken = false;
        c->final_fused = false;
        c->f_frame = frame;
        c->frame_tag = next_ownv_tag(c);
        return RT_OK;
    }
    if (stage != c->f_stage) RT_FAIL(c, RT_ERR_STATE, "rt_frame_stage: expected stage %d, got %d", c->f_stage, stage);
    if (stage > passes + 1) RT_FAIL(c, RT_ERR_ARG, "bad stage %d", stage);
    /* buffer roles are a pure function of the stage index, so a repeated _begin (retry after a failed
     * _run) cannot swap the ping-pong pair twice: pass k reads what pass k-1 wrote (Y for k = 0) and
     * writes Z for even k, X for odd k */
    if (stage >= 1 && stage <= passes)
    {
        const int k = stage - 1;
        c->f_in = (k == 0) ? c->fY : ((k & 1) ? c->fZ : c->fX);
        c->f_out = (k & 1) ? c->fX : c->fZ;
    }
    return RT_OK;
}

static int stage_run_ranges(rt_ctx* c, int frame, int stage, int part, int row0, int row1, int rowb0, int rowb1);
int rt_frame_stage_run_part(rt_ctx* c, int frame, int stage, int part, int row0, int row1)
{
    return stage_run_ranges(c, frame, stage, part, row0, row1, 0, 0);
}
/* the stage's kernels over up to two disjoint row ranges in ONE launch each (a strip's two boundary bands);
 * ranges: n x {row0, row1}, n = 1 or 2; lane 1 = the second stream (as rt_frame_stage_run_async) */
int rt_frame_stage_run_ranges(rt_ctx* c, int frame, int stage, int part, int n, const int* ranges, int lane)
{
    RT_CHECK_CTX(c);
    if (!ranges || n < 1 || n > 2) RT_FAIL(c, RT_ERR_ARG, "1 or 2 row ranges");
    if (n == 2 && !(ranges[1] <= ranges[2] || ranges[3] <= ranges[0])) RT_FAIL(c, RT_ERR_ARG, "row ranges overlap");
    if (!lane) return stage_run_ranges(c, frame, stage, part, ranges[0], ranges[1], n == 2 ? ranges[2] : 0, n == 2 ? ranges[3] : 0);
    if (!c->aux_used) RT_HIP(c, hipStreamWaitEvent(c->aux_stream, c->ev_stage, 0));
    c->aux_used = true;
    hipStream_t main_stream = c->stream;
    const bool timing = c->timing;
    c->stream = c->aux_stream;
    c->timing = false;
    const int rc = stage_run_ranges(c, frame, stage, part, ranges[0], ranges[1], n == 2 ? ranges[2] : 0, n == 2 ? ranges[3] : 0);
    c->stream = main_stream;
    c->timing = timing;
    return rc;
}
static int stage_run_ranges(rt_ctx* c, int frame, int stage, int part, int row0, int row1, int rowb0, int rowb1)
{
    RT_CHECK_CTX(c);
    NEED_SCENE(c);
    if (stage != c->f_stage) RT_FAIL(c, RT_ERR_STATE, "rt_frame_stage_run: expected stage %d, got %d", c->f_stage, stage);
    if (row0 < c->row_begin) row0 = c->row_begin;
    if (row1 > c->row_end) row1 = c->row_end;
    if (rowb0 < c->row_begin) rowb0 = c->row_begin;
    if (rowb1 > c->row_end) rowb1 = c->row_end;
    if (rowb0 >= rowb1) rowb0 = rowb1 = 0;
    if (row0 >= row1) { row0 = rowb0; row1 = rowb1; rowb0 = rowb1 = 0; }
    if (row0 >= row1) return RT_OK;
    c->subb0 = rowb0; c->subb1 = rowb1;
    const int passes = c->opt.spatial_resampling_passes;
    const bool whole = row0 == c->row_begin && row1 == c->row_end && rowb0 >= rowb1;
    const bool T = c->timing && whole;
    auto mark = [&](int i) { if (T && i <= 8) hipEventRecord(c->ev[i], c->stream); };
    c->sub0 = row0; c->sub1 = row1;
    c->cur_tag = c->frame_tag; /* launches of the staged frame write / trust own-visibility flags under the frame's tag */
    int rc = RT_OK;
    if (stage == 0)
    {
        mark(0);
        if (part != 2 && c->f_clear) rc = rt_clear(c);
        mark(1);
        bool deferred = false; /* the primary rays are traced by the candidates' launch */
        if (part != 2 && rc == RT_OK) rc = raycast_or_take(c, whole, frame, part == 0 && whole && use_fused_stage0(c), &deferred);
        if (part != 2) c->stage0_one_launch = deferred;
        if (part != 2 && rc == RT_OK && row0 == c->row_begin && row1 == c->row_end && !deferred) rc = refresh_shaded_bits(c);
        mark(2);
        if (part != 1 && rc == RT_OK && !c->gen_taken)
        {
            rc = join_tail_for(c, c->fY); /* this frame's candidates may go where the previous frame's resolve still reads */
            if (rc == RT_OK) rc = launch_generate(c, frame, c->fY, c->fX, c->opt.use_temporal_resampling != 0, deferred ? c->d_vis : nullptr);
        }
        if (deferred && rc == RT_OK) rc = refresh_shaded_bits(c);
        mark(3);
    }
    else if (stage <= passes)
    {
        const int k = stage - 1;
        rc = join_tail_for(c, c->f_out); /* the pass's output buffer can be the one the previous frame's resolve reads */
        if (rc == RT_OK && k == passes - 1 && use_fused_final(c))
        {
            rc = join_tail(c); /* the kernel writes the accumulation buffer and the pixels the previous frame's tail writes */
            if (rc == RT_OK) rc = launch_spatial(c, frame, k, c->f_in, c->f_out, true);
        }
        else if (rc == RT_OK) rc = launch_spatial(c, frame, k, c->f_in, c->f_out);
        if (k < 3) mark(4 + k);
    }
    else
    {
        for (int k = passes; k < 3; ++k) mark(4 + k);
        /* passes == 0: the reference resolves reservoir_buffer1, which no kernel of that frame wrote
         * (10_restir_di.cpp:324-368): zeros after start-up, else what an earlier frame's last odd pass
         * left there. Logical RT_RES_1 (physical Z) holds exactly that content here, frame by frame
         * (rt_frame_stage_end keeps the logical names on the reference's buffers), so the result is the
         * reference's, stale data included. */
        const int final_phys = passes > 0 ? c->f_out : c->fZ;
        if (c->final_fused)
        {
            /* the last pass has shaded (and, key 20, tone-mapped) these rows already */
            mark(7);
            if (!c->tune_fuse_tonemap) rc = launch_tone_mapping(c);
            mark(8);
            c->sub0 = c->sub1 = -1;
            c->subb0 = c->subb1 = 0;
            c->cur_tag = 0u;
            return rc;
        }
        const bool tail = whole && use_tail(c);
        hipStream_t ms = c->stream;
        if (tail)
        {
            /* resolve + tone_mapping on the tail stream, behind everything enqueued so far (and behind the previous
             * frame's tail: same stream); the main stream does not wait for them */
            RT_HIP(c, hipEventRecord(c->ev_tail_go, ms));
            RT_HIP(c, hipStreamWaitEvent(c->tail_stream, c->ev_tail_go, 0));
            c->stream = c->tail_stream;
        }
        else rc = join_tail(c);
        /* tone_mapping reads the pixel's own accumulation value only (common/kernels/common.cu:30-74): k_resolve maps what it
         * has just written (r05; not the persistent-wavefront A/B form, which has no such parameter) */
        const bool fuse_tm = c->tune_fuse_tonemap != 0 && !c->tune_stream;
        if (rc == RT_OK) rc = launch_resolve(c, final_phys, fuse_tm ? (uint32_t*)c->d_pixels : nullptr);
        mark(7);
        if (rc == RT_OK && !fuse_tm) rc = launch_tone_mapping(c);
        mark(8);
        if (tail)
        {
            c->stream = ms;
            if (rc == RT_OK)
            {
                RT_HIP(c, hipEventRecord(c->ev_tail, c->tail_stream));
                c->tail_pending_main = true;
                c->tail_phys = final_phys;
            }
        }
        c->resolve_on_tail = tail;
    }
    c->sub0 = c->sub1 = -1;
    c->subb0 = c->subb1 = 0;
    c->cur_tag = 0u;
    return rc;
}

/* The same rows on the context's second stream: they run beside whatever the main stream is doing
 * for this stage (a strip computes its boundary rows first, posts their halo exchange, and lets the
 * interior rows fill the rest of the GPU meanwhile).
 * The lane starts after everything that was enqueued on the main stream when rt_frame_stage_begin
 * or, later, rt_frame_stage_fork was called (fork after a part the lane depends on, e.g. the
 * raycast part of stage 0), and rt_frame_stage_end joins it back into the main stream. Rows of the
 * two lanes must not overlap, and the lane must not read halo rows still in flight. */
int rt_frame_stage_fork(rt_ctx* c)
{
    RT_CHECK_CTX(c);
    if (c->aux_used) RT_FAIL(c, RT_ERR_STATE, "rt_frame_stage_fork after the stage's second lane has started");
    RT_HIP(c, hipEventRecord(c->ev_stage, c->stream));
    return RT_OK;
}
int rt_frame_stage_run_async(rt_ctx* c, int frame, int stage, int part, int row0, int row1)
{
    RT_CHECK_CTX(c);
    if (!c->aux_used) RT_HIP(c, hipStreamWaitEvent(c->aux_stream, c->ev_stage, 0));
    c->aux_used = true;
    hipStream_t main_stream = c->stream;
    const bool timing = c->timing;
    c->stream = c->aux_stream;
    c->timing = false; /* the per-kernel events belong to the main lane */
    const int rc = rt_frame_stage_run_part(c, frame, stage, part, row0, row1);
    c->stream = main_stream;
    c->timing = timing;
    return rc;
}

/* all parts of the stage (part 0); stage 0 can be split: part 1 = [clear,] raycast, part 2 = generate */
int rt_frame_stage_run(rt_ctx* c, int frame, int stage, int row0, int row1)
{
    return rt_frame_stage_run_part(c, frame, stage, 0, row0, row1);
}

int rt_frame_stage_end(rt_ctx* c, int stage)
{
    RT_CHECK_CTX(c);
    if (stage != c->f_stage) RT_FAIL(c, RT_ERR_STATE, "rt_frame_stage_end: expected stage %d, got %d", c->f_stage, stage);
    if (c->aux_used)
    {
        RT_HIP(c, hipEventRecord(c->ev_aux, c->aux_stream));
        RT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_aux, 0));
        c->aux_used = false;
    }
    const int passes = c->opt.spatial_resampling_passes;
    c->fuse = HaloFuse{};
    if (stage == 0) { const int rc = launch_next_raycast(c, c->f_frame); if (rc != RT_OK) return rc; }
    if (stage <= passes) { c->f_stage = stage + 1; return RT_OK; }
    {
        /* behind this frame's resolve (all its parts: the second lane has been joined above), on the stream it ran on: the
         * look-ahead stage 0 waits for the one before the latest */
        std::swap(c->ev_resolved[0], c->ev_resolved[1]);
        RT_HIP(c, hipEventRecord(c->ev_resolved[0], c->resolve_on_tail ? c->tail_stream : c->stream));
        if (c->n_resolved < 2) ++c->n_resolved;
        c->resolve_on_tail = false;
    }
    const int X = c->fX, Y = c->fY, Z = c->fZ;
    if (passes < 2)
    {
        /* logical RT_RES_0 still equals the post-temporal reservoirs: materialise the copy the
         * reference's save_temporal_reservoir makes (10_restir_di.cpp:314-321) */
        const size_t n = local_pixels(c);
        { const int jr = join_tail_for(c, X); if (jr != RT_OK) return jr; }
        c->rec_gserial[X] = c->rec_gserial[Y];
        RT_HIP(c, hipMemcpyAsync(c->d_rec[X], c->d_rec[Y], n * 64, hipMemcpyDeviceToDevice, c->stream));
        RT_HIP(c, hipMemcpyAsync(c->d_rad[X], c->d_rad[Y], n * 16, hipMemcpyDeviceToDevice, c->stream));
    }
    const int final_phys = passes > 0 ? c->f_out : Z;
    /* new logical names: TEMPORAL = Y; RES_1 = Z; RES_0 = X (pass-1 output / copy) */
    c->res_map[RT_RES_TEMPORAL] = Y;
    c->res_map[RT_RES_0] = X;
    c->res_map[RT_RES_1] = Z;
    c->f_final = (final_phys == Z) ? RT_RES_1 : RT_RES_0;
    c->f_stage = 0;
    c->last_valid = c->timing;
    return RT_OK;
}

/* The frame as a sequence of stages, so that a strip context can exchange halos in between:
 *   stage 0            [clear] raycast, generate_candidate(+temporal) into the rotating buffers
 *   stage 1..passes    spatial pass stage-1      (its input buffer must have valid halos)
 *   stage passes+1     resolve, tone_mapping, buffer renaming
 * rt_frame runs them back to back. */
int rt_frame_stage(rt_ctx* c, int frame, int stage, int clear_first)
{
    RT_CHECK_CTX(c);
    int rc = rt_frame_stage_begin(c, frame, stage, clear_first);
    if (rc == RT_OK) rc = rt_frame_stage_run(c, frame, stage, c->row_begin, c->row_end);
    if (rc == RT_OK) rc = rt_frame_stage_end(c, stage);
    return rc;
}

/* physical buffer that spatial pass `stage-1` of the running frame will read (for rt_halo_*_phys) */
int rt_frame_stage_input(rt_ctx* c, int stage, int* phys)
{
    RT_CHECK_CTX(c);
    if (!phys || stage < 1 || stage != c->f_stage) RT_FAIL(c, RT_ERR_STATE, "no such pending stage %d", stage);
    const int k = stage - 1;
    *phys = (k == 0) ? c->f_in : c->f_out;
    return RT_OK;
}
/* physical buffer the CURRENT stage (after its _begin) writes: stage 0 -> the candidates(+temporal)
 * buffer, spatial stage -> its output; this is what the next spatial pass will gather from */
int rt_frame_stage_output(rt_ctx* c, int stage, int* phys)
{
    RT_CHECK_CTX(c);
    if (!phys || stage != c->f_stage) RT_FAIL(c, RT_ERR_STATE, "stage %d is not running", stage);
    *phys = (stage == 0) ? c->fY : c->f_out;
    return RT_OK;
}

int rt_frame(rt_ctx* c, int frame, int clear_first, int* final_res)
{
    RT_CHECK_CTX(c);
    NEED_SCENE(c);
    if (c->row_begin != 0 || c->row_end != c->H)
        RT_FAIL(c, RT_ERR_STATE, "rt_frame is for single-strip contexts; strips run rt_frame_stage and exchange halos");
    const int passes = c->opt.spatial_resampling_passes;
    for (int st = 0; st <= passes + 1; ++st)
    {
        const int rc = rt_frame_stage(c, frame, st, clear_first);
        if (rc != RT_OK) return rc;
    }
    if (final_res) *final_res = c->f_final;
    return RT_OK;
}

int rt_timing(rt_ctx* c, float ms[9])
{
    RT_CHECK_CTX(c);
    if (!ms) return RT_ERR_ARG;
    if (!c->last_valid) RT_FAIL(c, RT_ERR_STATE, "no timed frame (rt_timing_enable + rt_frame first)");
    RT_HIP(c, hipEventSynchronize(c->ev[8]));
    for (int k = 0; k < 8; ++k) RT_HIP(c, hipEventElapsedTime(&ms[k], c->ev[k], c->ev[k + 1]));
    RT_HIP(c, hipEventElapsedTime(&ms[8], c->ev[0], c->ev[8]));
    /* the frame's primary rays were traced beside the previous frame (rt_tuning key 14): the duration of that launch,
     * which ran concurrently with other kernels and is not part of ms[8] */
    if (c->timed_spec_set >= 0)
    {
        RT_HIP(c, hipEventSynchronize(c->ev_spec_t[c->timed_spec_set][1]));
        RT_HIP(c, hipEventElapsedTime(&ms[1], c->ev_spec_t[c->timed_spec_set][0], c->ev_spec_t[c->timed_spec_set][1]));
    }
    return RT_OK;
}

int rt_stage0_one_launch(rt_ctx* c, int* one_launch)
{
    RT_CHECK_CTX(c);
    if (!one_launch) return RT_ERR_ARG;
    *one_launch = c->stage0_one_launch ? 1 : 0;
    return RT_OK;
}

int rt_local_rows(rt_ctx* c, int* first_row, int* n_rows)
{
    RT_CHECK_CTX(c);
    if (first_row) *first_row = c->lrow0;
    if (n_rows) *n_rows = c->lrows;
    return RT_OK;
}

int rt_download(rt_ctx* c, int buf, void* dst, size_t bytes)
{
    RT_CHECK_CTX(c);
    JOIN_TAIL(c);
    if (!dst) RT_FAIL(c, RT_ERR_ARG, "null dst");
    const size_t n = local_pixels(c);
    switch (buf)
    {
        case RT_BUF_VISIBILITY:
            if (bytes != n * 16) RT_FAIL(c, RT_ERR_ARG, "size mismatch: want %zu", n * 16);
            RT_HIP(c, hipMemcpyAsync(dst, c->d_vis, bytes, hipMemcpyDeviceToHost, c->stream));
            break;
        case RT_BUF_ACCUMULATION:
            if (bytes != n * 16) RT_FAIL(c, RT_ERR_ARG, "size mismatch: want %zu", n * 16);
            RT_HIP(c, hipMemcpyAsync(dst, c->d_accum, bytes, hipMemcpyDeviceToHost, c->stream));
            break;
        case RT_BUF_PIXELS:
            if (bytes != n * 4) RT_FAIL(c, RT_ERR_ARG, "size mismatch: want %zu", n * 4);
            RT_HIP(c, hipMemcpyAsync(dst, c->d_pixels, bytes, hipMemcpyDeviceToHost, c->stream));
            break;
        case RT_BUF_RES_0:
        case RT_BUF_RES_1:
        case RT_BUF_RES_TEMPORAL:
        {
            if (bytes != n * 76) RT_FAIL(c, RT_ERR_ARG, "size mismatch: want %zu", n * 76);
            const int phys = c->res_map[buf - RT_BUF_RES_0];
            int rc = ensure_stage(c, n * 76);
            if (rc != RT_OK) return rc;
            k_res_to_ref<<<(int)((n + 255) / 256), 256, 0, c->stream>>>((int)n, c->d_rec[phys], c->d_rad[phys], (uint32_t*)c->d_stage);
            RT_HIP(c, hipGetLastError());
            RT_HIP(c, hipMemcpyAsync(dst, c->d_stage, bytes, hipMemcpyDeviceToHost, c->stream));
            break;
        }
        default: RT_FAIL(c, RT_ERR_ARG, "unknown buffer %d", buf);
    }
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_upload(rt_ctx* c, int buf, const void* src, size_t bytes)
{
    RT_CHECK_CTX(c);
    JOIN_TAIL(c);
    JOIN_SPEC(c);
    if (!src) RT_FAIL(c, RT_ERR_ARG, "null src");
    const size_t n = local_pixels(c);
    switch (buf)
    {
        case RT_BUF_VISIBILITY:
        {
            if (bytes != n * 16) RT_FAIL(c, RT_ERR_ARG, "size mismatch: want %zu", n * 16);
            NEED_SCENE(c);
            RT_HIP(c, hipMemcpyAsync(c->d_vis, src, bytes, hipMemcpyHostToDevice, c->stream));
            FrameParams P = make_params(c, 0, 0);
            P.row0 = c->lrow0; P.row1 = c->lrow0 + c->lrows; /* all rows held, halos included */
            k_gbuffer_from_vis<<<tile_grid(c->W, c->lrows), BLOCK, 0, c->stream>>>(make_scene(c), P, c->d_vis, c->d_g0, c->d_g1);
            RT_HIP(c, hipGetLastError());
            c->has_gbuffer = true;
            ++c->gbuf_serial;
            c->shaded_bits_stale = true;
            ++c->epoch;
            break;
        }
        case RT_BUF_ACCUMULATION:
            if (bytes != n * 16) RT_FAIL(c, RT_ERR_ARG, "size mismatch: want %zu", n * 16);
            RT_HIP(c, hipMemcpyAsync(c->d_accum, src, bytes, hipMemcpyHostToDevice, c->stream));
            break;
        case RT_BUF_RES_0:
        case RT_BUF_RES_1:
        case RT_BUF_RES_TEMPORAL:
        {
            if (bytes != n * 76) RT_FAIL(c, RT_ERR_ARG, "size mismatch: want %zu", n * 76);
            if (!c->has_gbuffer) RT_FAIL(c, RT_ERR_STATE, "upload RT_BUF_VISIBILITY (or rt_raycast) before reservoirs");
            ++c->res_epoch;
            const int phys = c->res_map[buf - RT_BUF_RES_0];
            c->rec_gserial[phys] = c->gbuf_serial;
            /* the record keeps M in 30 bits: refuse what it cannot hold rather than truncate */
            for (size_t i = 0; i < n; ++i)
            {
                const int32_t M = ((const rt_reservoir*)src)[i].M;
                if (M < 0 || M >= (1 << 30)) RT_FAIL(c, RT_ERR_UNSUPPORTED, "reservoir %zu has M = %d outside [0, 2^30)", i, M);
            }
            int rc = ensure_stage(c, n * 76);
            if (rc != RT_OK) return rc;
            RT_HIP(c, hipMemcpyAsync(c->d_stage, src, bytes, hipMemcpyHostToDevice, c->stream));
            k_res_from_ref<<<(int)((n + 255) / 256), 256, 0, c->stream>>>((int)n, (const uint32_t*)c->d_stage, c->d_g1, c->d_rec[phys], c->d_rad[phys]);
            RT_HIP(c, hipGetLastError());
            break;
        }
        default: RT_FAIL(c, RT_ERR_ARG, "buffer %d cannot be uploaded", buf);
    }
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

size_t rt_halo_bytes(rt_ctx* c, int n_rows) { return c ? (size_t)n_rows * (size_t)c->W * 80 : 0; }

static int halo_range(rt_ctx* c, int row0, int n_rows)
{
    if (n_rows <= 0 || row0 < c->lrow0 || row0 + n_rows > c->lrow0 + c->lrows)
        RT_FAIL(c, RT_ERR_ARG, "rows [%d,%d) outside the rows held [%d,%d)", row0, row0 + n_rows, c->lrow0, c->lrow0 + c->lrows);
    return RT_OK;
}
/* res: RT_RES_* (logical) or RT_RES_PHYS + p for the physical buffer p that
 * rt_frame_stage_input reported */
static int halo_phys(rt_ctx* c, int res)
{
    if (res >= RT_RES_PHYS && res < RT_RES_PHYS + 5 && c->d_rec[res - RT_RES_PHYS]) return res - RT_RES_PHYS;
    if (res >= 0 && res <= 2) return c->res_map[res];
    return -1;
}
int rt_halo_pack(rt_ctx* c, int res, int row0, int n_rows, void* device_dst)
{
    RT_CHECK_CTX(c);
    JOIN_TAIL(c);
    const int phys = halo_phys(c, res);
    if (phys < 0) RT_FAIL(c, RT_ERR_ARG, "bad reservoir buffer id %d", res);
    int rc = halo_range(c, row0, n_rows);
    if (rc != RT_OK) return rc;
    const size_t off = (size_t)(row0 - c->lrow0) * c->W, n = (size_t)n_rows * c->W;
    RT_HIP(c, hipMemcpyAsync(device_dst, c->d_rec[phys] + 4 * off, n * 64, hipMemcpyDeviceToDevice, c->stream));
    RT_HIP(c, hipMemcpyAsync((char*)device_dst + n * 64, c->d_rad[phys] + off, n * 16, hipMemcpyDeviceToDevice, c->stream));
    return RT_OK;
}
int rt_halo_unpack(rt_ctx* c, int res, int row0, int n_rows, const void* device_src)
{
    RT_CHECK_CTX(c);
    JOIN_TAIL(c);
    const int phys = halo_phys(c, res);
    if (phys < 0) RT_FAIL(c, RT_ERR_ARG, "bad reservoir buffer id %d", res);
    int rc = halo_range(c, row0, n_rows);
    if (rc != RT_OK) return rc;
    const size_t off = (size_t)(row0 - c->lrow0) * c->W, n = (size_t)n_rows * c->W;
    RT_HIP(c, hipMemcpyAsync(c->d_rec[phys] + 4 * off, device_src, n * 64, hipMemcpyDeviceToDevice, c->stream));
    RT_HIP(c, hipMemcpyAsync(c->d_rad[phys] + off, (const char*)device_src + n * 64, n * 16, hipMemcpyDeviceToDevice, c->stream));
    return RT_OK;
}

/* ---- sparse halos: see k_halo_mark. side 0 = the strip below (rows [row_begin-halo, row_begin)),
 * side 1 = the strip above (rows [row_end, row_end+halo)). ---- */
static int halo_side_region(rt_ctx* c, int side, int* r0, int* n)
{
    if (side == 0) { *r0 = c->lrow0; *n = c->row_begin - c->lrow0; }
    else if (side == 1) { *r0 = c->row_end; *n = c->lrow0 + c->lrows - c->row_end; }
    else RT_FAIL(c, RT_ERR_ARG, "side must be 0 or 1");
    if (*n <= 0) RT_FAIL(c, RT_ERR_STATE, "no halo rows on side %d", side);
    return RT_OK;
}
size_t rt_halo_bitmap_words(rt_ctx* c, int n_rows)
{
    if (!c) return 0;
    const size_t nw = ((size_t)n_rows * c->W + 31) / 32;
    return 1 + 2 * nw; /* count, bits, prefix; only the first 1 + nw words travel */
}
size_t rt_halo_flags_bytes(rt_ctx* c, int n_rows) { return c ? (size_t)n_rows * c->W : 0; }
int rt_halo_flags_pack(rt_ctx* c, int row0, int n_rows, void* device_dst)
{
    RT_CHECK_CTX(c);
    int rc = halo_range(c, row0, n_rows);
    if (rc != RT_OK) return rc;
    const int n = n_rows * c->W;
    k_halo_flags<true><<<(n + 255) / 256, 256, 0, c->stream>>>(c->d_g1, (size_t)(row0 - c->lrow0) * c->W, n, (uint8_t*)device_dst);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}
int rt_halo_flags_unpack(rt_ctx* c, int row0, int n_rows, const void* device_src)
{
    RT_CHECK_CTX(c);
    int rc = halo_range(c, row0, n_rows);
    if (rc != RT_OK) return rc;
    if (row0 < c->row_end && row0 + n_rows > c->row_begin) RT_FAIL(c, RT_ERR_ARG, "flags may only be unpacked into halo rows");
    const int n = n_rows * c->W;
    for (int set = 0; set < rt_ctx::NGSET; ++set)
        if (c->d_gset[set][2])
            k_halo_flags<false><<<(n + 255) / 256, 256, 0, c->stream>>>(c->d_gset[set][2], (size_t)(row0 - c->lrow0) * c->W, n, (uint8_t*)const_cast<void*>(device_src));
    RT_HIP(c, hipGetLastError());
    c->shaded_bits_stale = true; /* the LDS-staged spatial pass reads these rows' bits too */
    c->mark_bits_epoch = 0;      /* and the halo marks' cached rows */
    if (row0 == c->lrow0 && row0 + n_rows == c->row_begin) c->halo_flags_epoch[0] = c->epoch;
    if (row0 == c->row_end && row0 + n_rows == c->lrow0 + c->lrows) c->halo_flags_epoch[1] = c->epoch;
    return RT_OK;
}
/* records of the neighbour on `side` that spatial passes [pass, pass + n_pass) of `frame` will
 * gather: n_pass consecutive bitmaps of rt_halo_bitmap_words() words each, one launch */
int rt_halo_mark_sides(rt_ctx* c, int frame, int pass, int n_pass, void* bitmaps_side0, void* bitmaps_side1)
{
    RT_CHECK_CTX(c);
    if (!c->has_gbuffer) RT_FAIL(c, RT_ERR_STATE, "no G-buffer yet");
    if (n_pass <= 0 || (!bitmaps_side0 && !bitmaps_side1)) return RT_OK;
    HaloRegions R = {};
    R.quick = c->tune_mark_quick;
    void* bm[2] = {bitmaps_side0, bitmaps_side1};
    for (int side = 0; side < 2; ++side)
    {
        if (!bm[side]) continue;
        int rc = halo_side_region(c, side, &R.row0[side], &R.rows[side]);
        if (rc != RT_OK) return rc;
        if (c->halo_flags_epoch[side] != c->epoch)
            RT_FAIL(c, RT_ERR_STATE, "rt_halo_mark: the neighbour's shaded flags (rt_halo_flags_unpack of all %d halo rows on side %d) "
                                     "must be refreshed after a camera, scene or option change", R.rows[side], side);
        R.words[side] = rt_halo_bitmap_words(c, R.rows[side]);
        R.bitmaps[side] = (uint32_t*)bm[side];
    }
    /* both sides in ONE allocation, side 0 first (rt_mg's arena): cleared with one memset, whatever lies between included */
    const size_t bytes0 = R.words[0] * 4 * (size_t)n_pass, bytes1 = R.words[1] * 4 * (size_t)n_pass;
    const bool together = bm[0] && bm[1] && (char*)bm[1] >= (char*)bm[0] + bytes0 && (size_t)((char*)bm[1] - (char*)bm[0]) <= 2 * bytes0 + (1u << 20);
    if (together) RT_HIP(c, hipMemsetAsync(bm[0], 0, (size_t)((char*)bm[1] - (char*)bm[0]) + bytes1, c->stream));
    else
    {
        if (bm[0]) RT_HIP(c, hipMemsetAsync(bm[0], 0, bytes0, c->stream));
        if (bm[1]) RT_HIP(c, hipMemsetAsync(bm[1], 0, bytes1, c->stream));
    }
    /* only own rows within `halo` rows of a side can reach across it: one band per side, one launch over both (or
     * over all own rows when the bands meet) */
    const int lo_end = c->row_begin + c->halo < c->row_end ? c->row_begin + c->halo : c->row_end;
    const int hi_beg = c->row_end - c->halo > c->row_begin ? c->row_end - c->halo : c->row_begin;
    if (bm[0] && bm[1] && lo_end < hi_beg) { c->sub0 = c->row_begin; c->sub1 = lo_end; c->subb0 = hi_beg; c->subb1 = c->row_end; }
    else if (bm[0] && bm[1]) { c->sub0 = c->row_begin; c->sub1 = c->row_end; c->subb0 = c->subb1 = 0; }
    else if (bm[0]) { c->sub0 = c->row_begin; c->sub1 = lo_end; c->subb0 = c->subb1 = 0; }
    else { c->sub0 = hi_beg; c->sub1 = c->row_end; c->subb0 = c->subb1 = 0; }
    const FrameParams P = make_params(c, frame, pass, K_OTHER);
    const int grid = launch_grid(c);
    c->sub0 = c->sub1 = -1; c->subb0 = c->subb1 = 0;
    /* marks collected per workgroup in an LDS window, one global atomic per non-zero word (frame_kernels.h) where it applies */
    const bool window = c->tune_mark_window && (c->W % 32) == 0 && n_pass <= MARK_MAX_PASSES && halo_rows_needed(c->opt) <= SPL_HALO;
    if (window)
    {
        /* the shaded bit of every local pixel (own rows + the neighbours' flags in the halo rows), rebuilt in front of every
         * mark on the marking stream: ~3 us, and no staleness to track across the two G-buffer sets of the pipelined stage 0 */
        const int words = c->W / 32;
        if (!c->d_mark_bits) RT_HIP(c, hipMalloc(&c->d_mark_bits, (size_t)c->lrows * words * 4));
        /* r05: the bits are a function of the camera, the scene and the options (own rows: what raycast writes, the same in both
         * G-buffer sets; halo rows: the neighbours' flags, unpacked once per epoch), not of the frame: built once per epoch
         * from a G-buffer traced under it and kept (r04 rebuilt them in front of every mark: 13 us alone, 84 us beside a 4K
         * strip's raycast). A G-buffer of another epoch (marks before the first raycast after a change) is not cached.
         * Ordering: a mark that finds the cache valid may run on another stream than the mark that built it — ev_mark_bits. */
        const bool cacheable = c->gbuf_epoch == c->epoch;
        if (!(cacheable && c->mark_bits_epoch == c->epoch) || !c->tune_mark_cache)
        {
            if (c->mark_bits_event_valid) RT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_mark_bits, 0)); /* earlier readers / writer */
            k_shaded_bitmap<<<dim3((c->W + 255) / 256, c->lrows), 256, 0, c->stream>>>(c->W, c->lrows, c->d_g1, c->d_mark_bits);
            RT_HIP(c, hipGetLastError());
            c->mark_bits_epoch = cacheable ? c->epoch : 0;
            c->mark_bits_rebuilt = true;
        }
        else RT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_mark_bits, 0)); /* the build (and every mark since) is in front of this one */
        k_halo_mark<true><<<dim3(grid, c->tune_mark_split ? n_pass : 1), BLOCK, 0, c->stream>>>(P, c->d_g1, c->d_mark_bits, R, pass, n_pass);
    }
    else k_halo_mark<false><<<dim3(grid, c->tune_mark_split ? n_pass : 1), BLOCK, 0, c->stream>>>(P, c->d_g1, nullptr, R, pass, n_pass);
    RT_HIP(c, hipGetLastError());
    if (window)
    {
        /* whoever rebuilds the bits next (another stream, another epoch) waits for this reader; whoever reads them next waits for
         * the build in front of it */
        RT_HIP(c, hipEventRecord(c->ev_mark_bits, c->stream));
        c->mark_bits_event_valid = true;
    }
    {
        const size_t wmax = R.words[0] > R.words[1] ? R.words[0] : R.words[1];
        k_halo_scan_sides<<<dim3(halo_scan_chunks((int)((wmax - 1) / 2)), n_pass, 2), HALO_SCAN_THREADS, 0, c->stream>>>(R);
    }
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}
int rt_halo_mark(rt_ctx* c, int frame, int pass, int n_pass, int side, void* device_bitmaps)
{
    RT_CHECK_CTX(c);
    if (side != 0 && side != 1) RT_FAIL(c, RT_ERR_ARG, "side must be 0 or 1");
    if (!device_bitmaps) RT_FAIL(c, RT_ERR_ARG, "null pointer");
    return rt_halo_mark_sides(c, frame, pass, n_pass, side == 0 ? device_bitmaps : nullptr, side == 1 ? device_bitmaps : nullptr);
}
/* rebuild the prefix part of `count` consecutive bitmaps received from a neighbour */
int rt_halo_scan(rt_ctx* c, int n_rows, int count, void* device_bitmaps)
{
    RT_CHECK_CTX(c);
    if (count <= 0) return RT_OK;
    const size_t words = rt_halo_bitmap_words(c, n_rows);
    k_halo_scan<<<dim3(halo_scan_chunks((int)((words - 1) / 2)), count), HALO_SCAN_THREADS, 0, c->stream>>>((uint32_t*)device_bitmaps, (int)((words - 1) / 2), words);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}
static int halo_sparse(rt_ctx* c, bool pack, int res, int n, const int* row0, const int* n_rows, const void* const* device_bitmap, void* const* device_list)
{
    const int phys = halo_phys(c, res);
    if (phys < 0) RT_FAIL(c, RT_ERR_ARG, "bad reservoir buffer id %d", res);
    if (n <= 0) return RT_OK;
    if (n > 2) RT_FAIL(c, RT_ERR_ARG, "at most two row ranges per call");
    HaloLists Hh = {};
    int most = 0;
    for (int i = 0; i < n; ++i)
    {
        int rc = halo_range(c, row0[i], n_rows[i]);
        if (rc != RT_OK) return rc;
        Hh.bitmap[i] = (const uint32_t*)device_bitmap[i];
        Hh.n_pix[i] = n_rows[i] * c->W;
        Hh.nw[i] = (int)((rt_halo_bitmap_words(c, n_rows[i]) - 1) / 2);
        Hh.region_off[i] = (size_t)(row0[i] - c->lrow0) * c->W;
        Hh.list[i] = (float4*)device_list[i];
        if (Hh.n_pix[i] > most) most = Hh.n_pix[i];
    }
    const dim3 grid((most + 255) / 256, n);
    if (pack) k_halo_sparse<true><<<grid, 256, 0, c->stream>>>(Hh, c->d_rec[phys], c->d_rad[phys]);
    else k_halo_sparse<false><<<grid, 256, 0, c->stream>>>(Hh, c->d_rec[phys], c->d_rad[phys]);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}
/* The running spatial stage reads its halo records from the received lists and writes the records its neighbours marked
 * to the send lists itself (no rt_halo_unpack_sparse before / rt_halo_pack_sparse after it): call between
 * rt_frame_stage_begin(stage in 1..passes) and the _run calls it shall apply to; rt_frame_stage_end clears it. Regions
 * are the context's own: received records belong to the halo rows below (side 0) / above (side 1) the strip, sent ones to
 * the first / last `halo` owned rows. Lists and bitmaps as for the separate calls. */
int rt_halo_fuse_set(rt_ctx* c, const rt_halo_fuse* f)
{
    RT_CHECK_CTX(c);
    c->fuse = HaloFuse{};
    if (!f) return RT_OK;
    if (c->f_stage < 1 || c->f_stage > c->opt.spatial_resampling_passes) RT_FAIL(c, RT_ERR_STATE, "rt_halo_fuse_set outside a spatial stage");
    if (use_lds_spatial(c)) RT_FAIL(c, RT_ERR_UNSUPPORTED, "the LDS-staged spatial variant reads halo rows, not lists");
    HaloFuse F = {};
    F.rows = c->halo;
    F.nw = (int)(((size_t)c->halo * c->W + 31) / 32);
    for (int side = 0; side < 2; ++side)
    {
        const bool has = side == 0 ? c->row_begin > 0 : c->row_end < c->H;
        if ((f->need_bitmap[side] || f->recv_list[side] || f->give_bitmap[side] || f->send_list[side]) && !has)
            RT_FAIL(c, RT_ERR_ARG, "no neighbour on side %d", side);
        if (!has) continue;
        int r0 = 0, n = 0;
        const int rc = halo_side_region(c, side, &r0, &n);
        if (rc != RT_OK) return rc;
        if (n != c->halo || c->row_end - c->row_begin < c->halo) RT_FAIL(c, RT_ERR_STATE, "fused halos need full %d-row regions", c->halo);
        if ((f->need_bitmap[side] != nullptr) != (f->recv_list[side] != nullptr) || (f->give_bitmap[side] != nullptr) != (f->send_list[side] != nullptr))
            RT_FAIL(c, RT_ERR_ARG, "bitmap and list go together");
        F.need_bm[side] = (const uint32_t*)f->need_bitmap[side];
        F.recv[side] = (const float4*)f->recv_list[side];
        F.need_row0[side] = r0;
        F.give_bm[side] = (const uint32_t*)f->give_bitmap[side];
        F.send[side] = (float4*)f->send_list[side];
        F.give_row0[side] = side == 0 ? c->row_begin : c->row_end - c->halo;
    }
    c->fuse = F;
    return RT_OK;
}
int rt_halo_pack_sparse(rt_ctx* c, int res, int row0, int n_rows, const void* device_bitmap, void* device_dst)
{
    RT_CHECK_CTX(c);
    return halo_sparse(c, true, res, 1, &row0, &n_rows, &device_bitmap, &device_dst);
}
int rt_halo_unpack_sparse(rt_ctx* c, int res, int row0, int n_rows, const void* device_bitmap, const void* device_src)
{
    RT_CHECK_CTX(c);
    void* src = const_cast<void*>(device_src);
    return halo_sparse(c, false, res, 1, &row0, &n_rows, &device_bitmap, &src);
}
/* the same for up to two row ranges (both neighbours) in one launch */
int rt_halo_pack_sparse_ranges(rt_ctx* c, int res, int n, const int* row0, const int* n_rows, const void* const* device_bitmaps, void* const* device_dsts)
{
    RT_CHECK_CTX(c);
    return halo_sparse(c, true, res, n, row0, n_rows, device_bitmaps, device_dsts);
}
int rt_halo_unpack_sparse_ranges(rt_ctx* c, int res, int n, const int* row0, const int* n_rows, const void* const* device_bitmaps, const void* const* device_srcs)
{
    RT_CHECK_CTX(c);
    return halo_sparse(c, false, res, n, row0, n_rows, device_bitmaps, const_cast<void* const*>(device_srcs));
}

/* ---- hooks of the native strip driver (strip_mg.cpp), plain C-ABI like everything else ---- */
int rt_state_epoch(rt_ctx* c, uint64_t* epoch)
{
    RT_CHECK_CTX(c);
    if (!epoch) return RT_ERR_ARG;
    *epoch = c->epoch;
    return RT_OK;
}
/* the context's tail stream (resolve + tone_mapping of a staged frame, rt_tuning key 17): a strip driver enqueues its halo-plan
 * marks there instead of on a stream of its own — one hardware queue less to compete for, and a plan is needed a frame later */
int rt_side_stream(rt_ctx* c, int which, void** hip_stream)
{
    RT_CHECK_CTX(c);
    if (!hip_stream || which != 0) RT_FAIL(c, RT_ERR_ARG, "which: 0 = tail stream");
    *hip_stream = (void*)c->tail_stream;
    return RT_OK;
}
/* n <= 8 device-to-device copies in one launch on the current stream */
int rt_copy_parts(rt_ctx* c, int n, const void* const* src, void* const* dst, const size_t* bytes)
{
    RT_CHECK_CTX(c);
    if (n < 0 || n > 8 || (n > 0 && (!src || !dst || !bytes))) RT_FAIL(c, RT_ERR_ARG, "0..8 parts");
    if (n == 0) return RT_OK;
    CopyParts P;
    memset(&P, 0, sizeof(P));
    size_t largest = 0;
    for (int i = 0; i < n; ++i)
    {
        if (bytes[i] && (!src[i] || !dst[i])) RT_FAIL(c, RT_ERR_ARG, "null part %d", i);
        P.src[i] = (const char*)src[i]; P.dst[i] = (char*)dst[i]; P.bytes[i] = bytes[i];
        largest = bytes[i] > largest ? bytes[i] : largest;
    }
    if (largest == 0) return RT_OK;
    const size_t want = (largest / 16 + 255) / 256;
    /* up to 2048 workgroups per part: fewer, fatter ones (8 / 32) were slower on a busy GPU, profiles/r04_strip_interior_late.txt */
    const unsigned gx = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    k_copy_parts<<<dim3(gx, (unsigned)n), 256, 0, c->stream>>>(P);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}
/* WIRE_MODEL (strip_mg.cpp): phase 0 = stamp the GPU clock on the current stream (in front of an exchange), phase 1 = hold
 * the current stream until `ns` nanoseconds after that stamp (behind the exchange). slot: 0..7. A slot's stamp and wait must be
 * enqueued on ONE stream, and a slot must not be shared between streams: stream order is what keeps the next stamp of a slot behind
 * the wait that still reads it (the strip driver: slots 0-3 for exchanges on the main stream, 4-7 on the communication stream). */
int rt_wire_delay(rt_ctx* c, int phase, int slot, unsigned long long ns)
{
    RT_CHECK_CTX(c);
    if (slot < 0 || slot > 7 || (phase != 0 && phase != 1)) RT_FAIL(c, RT_ERR_ARG, "rt_wire_delay: phase 0 / 1, slot 0..7");
    /* one sleeping wavefront holds the stream for `ns`: a mistyped RT_MG_WIRE_GBS / RT_MG_WIRE_LAT_US must not park it until the
     * watchdog fires. 5 ms = a 4K frame's whole dense halo at a hundredth of an xGMI link. */
    if (ns > 5000000ull) RT_FAIL(c, RT_ERR_ARG, "rt_wire_delay: %llu ns is more than the 5 ms a modelled exchange may take (RT_MG_WIRE_GBS / RT_MG_WIRE_LAT_US?)", ns);
    if (!c->d_wire)
    {
        RT_HIP(c, hipMalloc(&c->d_wire, 8 * sizeof(unsigned long long)));
        RT_HIP(c, hipMemset(c->d_wire, 0, 8 * sizeof(unsigned long long)));
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000; /* 100 MHz */
        c->wall_khz = khz;
    }
    if (phase == 0) k_wire_stamp<<<1, 1, 0, c->stream>>>(c->d_wire + slot);
    else k_wire_wait<<<1, 64, 0, c->stream>>>(c->d_wire + slot, (unsigned long long)((double)ns * (double)c->wall_khz * 1e-6));
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}
int rt_get_stream(rt_ctx* c, void** hip_stream)
{
    RT_CHECK_CTX(c);
    if (!hip_stream) return RT_ERR_ARG;
    *hip_stream = (void*)c->stream;
    return RT_OK;
}
int rt_geometry(rt_ctx* c, int* width, int* height, int* row_begin, int* row_end, int* halo)
{
    RT_CHECK_CTX(c);
    if (width) *width = c->W;
    if (height) *height = c->H;
    if (row_begin) *row_begin = c->row_begin;
    if (row_end) *row_end = c->row_end;
    if (halo) *halo = c->halo;
    return RT_OK;
}
/* device addresses of `n_rows` storage rows of a reservoir buffer (64-B records, 16-B radiance side
 * records): dense halos are sent from and received into the buffers themselves, no staging copy */
int rt_res_region(rt_ctx* c, int res, int row0, int n_rows, void** rec, size_t* rec_bytes, void** rad, size_t* rad_bytes)
{
    RT_CHECK_CTX(c);
    const int phys = halo_phys(c, res);
    if (phys < 0) RT_FAIL(c, RT_ERR_ARG, "bad reservoir buffer id %d", res);
    int rc = halo_range(c, row0, n_rows);
    if (rc != RT_OK) return rc;
    const size_t off = (size_t)(row0 - c->lrow0) * c->W, n = (size_t)n_rows * c->W;
    if (rec) *rec = (void*)(c->d_rec[phys] + 4 * off);
    if (rec_bytes) *rec_bytes = n * 64;
    if (rad) *rad = (void*)(c->d_rad[phys] + off);
    if (rad_bytes) *rad_bytes = n * 16;
    return RT_OK;
}
/* Second lane for arbitrary calls: between rt_lane(ctx, 1) and rt_lane(ctx, 0) every enqueue of this
 * context goes to the second stream (ordered after rt_frame_stage_begin / _fork, joined by
 * rt_frame_stage_end) — e.g. the halo marks of the NEXT frame beside this frame's passes. */
int rt_lane(rt_ctx* c, int second)
{
    RT_CHECK_CTX(c);
    if (second)
    {
        if (c->lane_saved) RT_FAIL(c, RT_ERR_STATE, "rt_lane(1) twice");
        if (!c->aux_used) RT_HIP(c, hipStreamWaitEvent(c->aux_stream, c->ev_stage, 0));
        c->aux_used = true;
        c->lane_main = c->stream;
        c->lane_timing = c->timing;
        c->stream = c->aux_stream;
        c->timing = false;
        c->lane_saved = true;
    }
    else if (c->lane_saved)
    {
        c->stream = c->lane_main;
        c->timing = c->lane_timing;
        c->lane_saved = false;
    }
    return RT_OK;
}

int rt_ray_count(rt_ctx* c, uint64_t* rays, uint64_t* shaded_pixels)
{
    RT_CHECK_CTX(c);
    if (!c->has_gbuffer) RT_FAIL(c, RT_ERR_STATE, "no G-buffer yet");
    RT_HIP(c, hipMemsetAsync(c->d_counter, 0, 8, c->stream));
    k_count_shaded<<<launch_grid(c), BLOCK, 0, c->stream>>>(make_params(c, 0, 0), c->d_g1, c->d_counter);
    RT_HIP(c, hipGetLastError());
    unsigned long long shaded = 0;
    RT_HIP(c, hipMemcpyAsync(&shaded, c->d_counter, 8, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    const uint64_t n = (uint64_t)c->W * (uint64_t)(c->row_end - c->row_begin);
    uint64_t per_shaded = 1; /* resolve (10_restir_di.cu:443-444) */
    if (c->opt.use_visibility_reuse) per_shaded += 1; /* :129-130 */
    uint64_t merged = 0;
    if (c->opt.use_shadowed_target_function)
    {
        /* p-hat of the candidate (:115-118), two of the temporal step (:195-199, :224-227), and per
         * spatial pass one per shaded pixel (:375-378) plus one per neighbour that reaches the target
         * function (:346-350). The latter is counted by replaying the passes' RNG against the shaded
         * bits of the last frame (any reservoir buffer of that frame carries them). */
        per_shaded += 1 + (c->opt.use_temporal_resampling ? 2 : 0);
        if (c->opt.use_spatial_resampling)
        {
            per_shaded += (uint64_t)c->opt.spatial_resampling_passes;
            unsigned long long* d = nullptr;
            RT_HIP(c, hipMalloc(&d, 24));
            RT_HIP(c, hipMemsetAsync(d, 0, 24, c->stream));
            for (int k = 0; k < c->opt.spatial_resampling_passes; ++k)
            {
                /* halo rows: G-buffer flags if the neighbours' were exchanged for this frame, else the records */
                const bool g1_ok = (c->row_begin == c->lrow0 || c->halo_flags_epoch[0] == c->epoch) && (c->row_end == c->lrow0 + c->lrows || c->halo_flags_epoch[1] == c->epoch);
                k_spatial_bytes<<<launch_grid(c), BLOCK, 0, c->stream>>>(make_params(c, c->last_frame, k), c->d_g1,
                                                                        c->d_rec[c->res_map[RT_RES_TEMPORAL]], d, g1_ok);
            }
            RT_HIP(c, hipGetLastError());
            unsigned long long h[3] = {0, 0, 0};
            RT_HIP(c, hipMemcpyAsync(h, d, 24, hipMemcpyDeviceToHost, c->stream));
            RT_HIP(c, hipStreamSynchronize(c->stream));
            hipFree(d);
            merged = h[2];
        }
    }
    if (rays) *rays = n + per_shaded * shaded + merged;
    if (shaded_pixels) *shaded_pixels = shaded;
    return RT_OK;
}

/* shaded pixels (hit and not emissive: the ones that run RIS, reuse and shadow rays) of each owned
 * storage row of the current G-buffer, row_end - row_begin host counters */
/* BVH walks the build really performs (include/restir_rt.h) */
int rt_walk_stats_enable(rt_ctx* c, int on)
{
    RT_CHECK_CTX(c);
    int rc = rt_sync(c);
    if (rc != RT_OK) return rc;
    if (on)
    {
        if (!c->d_walk) RT_HIP(c, hipMalloc(&c->d_walk, 16 * 8));
        RT_HIP(c, hipMemset(c->d_walk, 0, 16 * 8));
    }
    c->walk_on = on != 0;
    /* a pipelined stage 0 enqueued before the switch carries the other setting: the next frame runs its own */
    c->spec_valid = false; c->spec_gen_valid = false;
    return RT_OK;
}
#ifdef RT_EXPERIMENTS
/* Experiments library only (not in include/restir_rt.h; tools/wave_timeline.py binds it by name). Arms the per-wavefront clock of
 * ONE kernel of the frame — 0 raycast, 1 generate_candidate, 2 spatial_resampling pass `pass`, 3 resolve, -1 off — or, with
 * `out` non-null, reads the 2 x n words the launches since then left (synchronises). Results never depend on it. */
int rt_exp_wave_clock(rt_ctx* c, int kernel, int pass, uint64_t* out, size_t n_words)
{
    RT_CHECK_CTX(c);
    int rc = rt_sync(c);
    if (rc != RT_OK) return rc;
    if (out)
    {
        if (!c->d_wave_clock) RT_FAIL(c, RT_ERR_STATE, "rt_exp_wave_clock: arm first");
        RT_HIP(c, hipMemcpy(out, c->d_wave_clock, std::min(n_words, c->wave_clock_words) * 8, hipMemcpyDeviceToHost)); /* the rest stays as the caller left it */
        return RT_OK;
    }
    /* one-wavefront workgroups over 8 x 8 tiles at most: every order's grid fits in twice the tile count */
    const size_t words = 16 * ((size_t)trace_grid(c) + 1024) + 8 * (size_t)launch_grid(c); /* k_raycast_quad: four workgroups per tile */
    if (kernel >= 0 && words > c->wave_clock_words)
    {
        if (c->d_wave_clock) hipFree(c->d_wave_clock);
        c->d_wave_clock = nullptr;
        RT_HIP(c, hipMalloc(&c->d_wave_clock, words * 8));
        c->wave_clock_words = words;
    }
    if (c->d_wave_clock) RT_HIP(c, hipMemset(c->d_wave_clock, 0, c->wave_clock_words * 8));
    c->wave_clock_kernel = kernel; c->wave_clock_pass = pass;
    c->spec_valid = false; c->spec_gen_valid = false;
    return RT_OK;
}
#endif
#ifdef RT_EXPERIMENTS
/* Experiments library only (tools/tile_lpt.py). Workgroup b of kernel 0 raycast / 1 generate_candidate (the one-launch stage 0 too) /
 * 3 resolve takes the tile workgroup perm[b] would have taken; perm must be a permutation of 0 .. n-1 that keeps b % 8 (checked), n =
 * the kernel's grid for the context's own rows; n = 0 removes it. Results never depend on it (every tile still runs exactly once). */
int rt_exp_tile_perm(rt_ctx* c, int kernel, const uint32_t* perm, size_t n)
{
    RT_CHECK_CTX(c);
    if (kernel != K_RAYCAST && kernel != K_GENERATE && kernel != K_RESOLVE) RT_FAIL(c, RT_ERR_ARG, "kernel 0, 1 or 3");
    int rc = rt_sync(c);
    if (rc != RT_OK) return rc;
    if (c->d_tile_perm[kernel]) { hipFree(c->d_tile_perm[kernel]); c->d_tile_perm[kernel] = nullptr; c->tile_perm_n[kernel] = 0; }
    c->spec_valid = false; c->spec_gen_valid = false;
    if (n == 0) return RT_OK;
    if (!perm || n != (size_t)trace_grid(c)) RT_FAIL(c, RT_ERR_ARG, "the permutation must cover the kernel's %d workgroups", trace_grid(c));
    std::vector<uint8_t> seen(n, 0);
    for (size_t b = 0; b < n; ++b)
    {
        if (perm[b] >= n || seen[perm[b]] || (perm[b] & 7u) != (b & 7u)) RT_FAIL(c, RT_ERR_ARG, "not an XCD-keeping permutation at %zu", b);
        seen[perm[b]] = 1;
    }
    RT_HIP(c, hipMalloc(&c->d_tile_perm[kernel], n * 4));
    RT_HIP(c, hipMemcpy(c->d_tile_perm[kernel], perm, n * 4, hipMemcpyHostToDevice));
    c->tile_perm_n[kernel] = n;
    return RT_OK;
}
#endif
int rt_walk_stats(rt_ctx* c, uint64_t out[16])
{
    RT_CHECK_CTX(c);
    if (!out) return RT_ERR_ARG;
    if (!c->d_walk) RT_FAIL(c, RT_ERR_STATE, "rt_walk_stats_enable first");
    int rc = rt_sync(c);
    if (rc != RT_OK) return rc;
    RT_HIP(c, hipMemcpy(out, c->d_walk, 16 * 8, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_row_shaded(rt_ctx* c, uint32_t* counts)
{
    RT_CHECK_CTX(c);
    if (!counts) return RT_ERR_ARG;
    if (!c->has_gbuffer) RT_FAIL(c, RT_ERR_STATE, "no G-buffer yet");
    const int rows = c->row_end - c->row_begin;
    uint32_t* d = nullptr;
    RT_HIP(c, hipMalloc(&d, (size_t)rows * 4));
    RT_HIP(c, hipMemsetAsync(d, 0, (size_t)rows * 4, c->stream));
    k_row_shaded<<<rows, BLOCK, 0, c->stream>>>(c->W, c->row_begin - c->lrow0, c->d_g1, d);
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipMemcpyAsync(counts, d, (size_t)rows * 4, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    hipFree(d);
    return RT_OK;
}

/* visibility-reuse rays the last fused frame actually walked (rt_tuning key 11): the candidates that survived the
 * temporal merge. Synchronises the stream. With key 11 = 0 or outside rt_frame every shaded pixel walks one. */
int rt_visibility_rays_walked(rt_ctx* c, uint64_t* walked)
{
    RT_CHECK_CTX(c);
    if (!walked) return RT_ERR_ARG;
    if (!c->d_visq_count) RT_FAIL(c, RT_ERR_STATE, "no fused frame with deferred visibility rays has run");
    unsigned int h[2] = {0u, 0u};
    RT_HIP(c, hipMemcpyAsync(h, c->d_visq_count, 8, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    if (c->aux_stream) RT_HIP(c, hipStreamSynchronize(c->aux_stream));
    *walked = (uint64_t)h[0] + (uint64_t)h[1];
    return RT_OK;
}

int rt_spatial_bytes(rt_ctx* c, int frame, int pass, int in, uint64_t* bytes, uint64_t* accepted)
{
    RT_CHECK_CTX(c);
    NEED_RES(c, in);
    if (!c->has_gbuffer) RT_FAIL(c, RT_ERR_STATE, "no G-buffer yet");
    unsigned long long* d = nullptr;
    RT_HIP(c, hipMalloc(&d, 24));
    RT_HIP(c, hipMemsetAsync(d, 0, 24, c->stream));
    k_spatial_bytes<<<launch_grid(c), BLOCK, 0, c->stream>>>(make_params(c, frame, pass), c->d_g1, c->d_rec[c->res_map[in]], d);
    RT_HIP(c, hipGetLastError());
    unsigned long long h[2] = {0, 0};
    RT_HIP(c, hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    hipFree(d);
    if (bytes) *bytes = h[0];
    if (accepted) *accepted = h[1];
    return RT_OK;
}

int rt_trace_closest(rt_ctx* c, const float* rays, uint32_t n, float* hits)
{
    RT_CHECK_CTX(c);
    if (!c->has_scene) RT_FAIL(c, RT_ERR_STATE, "no scene");
    if (n == 0) return RT_OK;
    float *d_r = nullptr, *d_h = nullptr;
    RT_HIP(c, hipMalloc(&d_r, (size_t)n * 32));
    RT_HIP(c, hipMalloc(&d_h, (size_t)n * 16));
    RT_HIP(c, hipMemcpyAsync(d_r, rays, (size_t)n * 32, hipMemcpyHostToDevice, c->stream));
    hipEvent_t e0, e1;
    RT_HIP(c, hipEventCreate(&e0));
    RT_HIP(c, hipEventCreate(&e1));
    RT_HIP(c, hipEventRecord(e0, c->stream));
#ifdef RT_EXPERIMENTS
    if (c->trace_mode == 2 || c->trace_mode == 3)
    {
        unsigned int* d_head = (unsigned int*)c->d_counter;
        RT_HIP(c, hipMemsetAsync(d_head, 0, 8, c->stream));
        const int grid = 256 * 6; /* persistent: 6 workgroups per CU (24 KB LDS each) */
        if (c->trace_mode == 2) k_trace_queue<false><<<grid, BLOCK, 0, c->stream>>>(make_scene(c).wide, d_r, (int)n, d_h, d_head);
        else k_trace_queue<true><<<grid, BLOCK, 0, c->stream>>>(make_scene(c).wide, d_r, (int)n, d_h, d_head);
    }
    else
#endif
    if (c->trace_mode == 0) k_trace_closest<0><<<(n + 255) / 256, 256, 0, c->stream>>>(make_scene(c), d_r, (int)n, d_h);
    else if (c->trace_mode == 4) k_trace_closest<0, true><<<(n + 255) / 256, 256, 0, c->stream>>>(make_scene(c), d_r, (int)n, d_h);
    else if (c->trace_mode == 5) k_trace_anyhit<true><<<(n + TRACE_BLOCK - 1) / TRACE_BLOCK, TRACE_BLOCK, 0, c->stream>>>(make_scene(c), d_r, (int)n, d_h);
    else if (c->trace_mode == 6) k_trace_anyhit<false><<<(n + TRACE_BLOCK - 1) / TRACE_BLOCK, TRACE_BLOCK, 0, c->stream>>>(make_scene(c), d_r, (int)n, d_h);
#ifdef RT_EXPERIMENTS
    else if (c->trace_mode == 7) k_trace_closest_quad<<<(n + 15) / 16, TRACE_BLOCK, 0, c->stream>>>(make_scene(c), d_r, (int)n, d_h);
    else k_trace_closest<1><<<(n + 255) / 256, 256, 0, c->stream>>>(make_scene(c), d_r, (int)n, d_h);
#endif
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipEventRecord(e1, c->stream));
    RT_HIP(c, hipMemcpyAsync(hits, d_h, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    hipEventElapsedTime(&c->last_trace_ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(d_r); hipFree(d_h);
    return RT_OK;
}
/* device time of the traversal kernel of the last rt_trace_closest call */
int rt_trace_time(rt_ctx* c, float* ms)
{
    RT_CHECK_CTX(c);
    if (ms) *ms = c->last_trace_ms;
    return RT_OK;
}

/* per ray: {BVH nodes visited, triangle tests} of the closest-hit traversal */
int rt_trace_stats(rt_ctx* c, const float* rays, uint32_t n, uint32_t* stats)
{
    RT_CHECK_CTX(c);
    if (!c->has_scene) RT_FAIL(c, RT_ERR_STATE, "no scene");
    if (n == 0) return RT_OK;
    float* d_r = nullptr;
    uint32_t* d_s = nullptr;
    RT_HIP(c, hipMalloc(&d_r, (size_t)n * 32));
    RT_HIP(c, hipMalloc(&d_s, (size_t)n * 8));
    RT_HIP(c, hipMemcpyAsync(d_r, rays, (size_t)n * 32, hipMemcpyHostToDevice, c->stream));
    if (c->trace_mode == 5) k_trace_stats_ws<<<(n + TRACE_BLOCK - 1) / TRACE_BLOCK, TRACE_BLOCK, 0, c->stream>>>(make_scene(c), d_r, (int)n, d_s);
    else if (c->trace_mode == 0) k_trace_stats<0><<<(n + 255) / 256, 256, 0, c->stream>>>(make_scene(c), d_r, (int)n, d_s);
    else if (c->trace_mode == 4) k_trace_stats<0, true><<<(n + 255) / 256, 256, 0, c->stream>>>(make_scene(c), d_r, (int)n, d_s);
#ifdef RT_EXPERIMENTS
    else k_trace_stats<1><<<(n + 255) / 256, 256, 0, c->stream>>>(make_scene(c), d_r, (int)n, d_s);
#endif
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipMemcpyAsync(stats, d_s, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    hipFree(d_r); hipFree(d_s);
    return RT_OK;
}
/* BVH build knob (before rt_scene_set): fragment length of the triangle pre-split in median
 * triangle extents; 0 disables splitting. */
int rt_bvh_config(rt_ctx* c, float split_factor)
{
    RT_CHECK_CTX(c);
    if (!(split_factor >= 0.0f)) RT_FAIL(c, RT_ERR_ARG, "split_factor must be >= 0");
    c->bvh_split_factor = split_factor;
    return RT_OK;
}
/* performance knobs (results never depend on them): keys 0..3 = tile order of raycast /
 * generate_candidate(+temporal) / spatial_resampling / resolve (0 row-major, 1 column-major inside
 * each XCD band); key 4 = extra LDS bytes per spatial workgroup (limits resident workgroups per CU). */
/* keys / values that select code the product library does not carry (A/B forms measured and left off; csrc/Makefile builds
 * them into librestir_rt_exp.so with -DRT_EXPERIMENTS, which the variant tests and tools load) */
static bool experiment_only(int key, int value)
{
#ifdef RT_EXPERIMENTS
    (void)key; (void)value;
    return false;
#else
    switch (key)
    {
        case 5: return value != 3;                 /* BVH builders 0 (LBVH), 1 (host SAH), 2 (PLOC + host top): the default is 3 */
        case 8: return value != 2;                 /* spatial pass forms: gather, LDS-staged bits, software-pipelined */
        case 9: return value != -1 && value != RT_SPATIAL_GATHER_AUTO_WAVES; /* other register budgets of the pass */
        case 10: return true;                      /* PLOC search radius (builder 2) */
        case 11: case 12: case 15: return value != 0; /* deferred visibility queue, pipelined RIS loop form, resolve as a stream */
        case 23: return value > 0;                 /* last pass + resolve in one kernel */
        case 24: return value != 0;                /* half-density raycast with helper lanes */
        case 16: return value == 2;                /* four lanes per primary ray */
        default: return false;
    }
#endif
}
int rt_tuning(rt_ctx* c, int key, int value)
{
    RT_CHECK_CTX(c);
    if (experiment_only(key, value))
        RT_FAIL(c, RT_ERR_UNSUPPORTED, "rt_tuning %d = %d selects an A/B form that only librestir_rt_exp.so (built with -DRT_EXPERIMENTS) carries", key, value);
    if (key >= 0 && key <= 3 && value >= -1 && value <= 7) c->tune_tile_mode[key] = value;
    else if (key == 4 && value >= 0 && value <= 160 * 1024) c->tune_spatial_lds = value;
    else if (key == 5 && value >= 0 && value <= 3) c->bvh_builder = value; /* before rt_scene_set */
    else if (key == 6 && value >= 0 && value <= 2) c->pt_wavefront = value;
    else if (key == 7 && value >= 0) c->bvh_bfs_records = value; /* before rt_scene_set */
    else if (key == 8 && value >= 0 && value <= 4) { c->tune_spatial_variant = value; c->shaded_bits_stale = true; }
    else if (key == 9 && (value == 0 || value == -1 || (value >= 4 && value <= 6))) c->tune_spatial_waves = value;
    else if (key == 10 && value >= 1 && value <= 256) c->ploc_radius = value; /* before rt_scene_set */
    else if (key == 11 && (value == 0 || value == 1)) c->tune_defer_vis = value;
    else if (key == 12 && (value == 0 || value == 1)) c->tune_ris_pipe = value;
    else if (key == 13 && value >= -1 && value <= 1) c->tune_ws = value;
    else if (key == 15 && (value == 0 || value == 1)) c->tune_stream = value;
    else if (key == 16 && value >= -1 && value <= 2) c->tune_ws_primary = value;
    else if (key == 14 && value >= -1 && value <= 2) { c->tune_spec = value; if (!use_next_raycast(c)) c->spec_valid = false; if (!use_next_generate(c)) c->spec_gen_valid = false; }
    else if (key == 17 && value >= -1 && value <= 1) c->tune_tail = value;
    else if (key == 18 && (value == 0 || value == 1)) c->tune_mark_quick = value;
    else if (key == 19 && (value == 0 || value == 1)) c->tune_mark_window = value;
    else if (key == 20 && (value == 0 || value == 1)) c->tune_fuse_tonemap = value;
    else if (key == 21 && (value == 0 || value == 1)) { c->tune_mark_cache = value; c->mark_bits_epoch = 0; }
    else if (key == 23 && value >= -1 && value <= 2) c->tune_fuse_final = value;
    else if (key == 24 && (value == 0 || value == 1)) c->tune_half_raycast = value;
    else if (key == 25 && value >= -1 && value <= 1) c->tune_fuse_raycast = value;
    else if (key == 26 && (value == 0 || value == 1)) c->tune_mark_split = value;
    else if (key == 22 && value >= -1 && value <= 1) { c->tune_spec_free = value; c->spec_valid = false; c->spec_gen_valid = false; }
    else RT_FAIL(c, RT_ERR_ARG, "bad tuning key/value %d/%d", key, value);
    return RT_OK;
}
int rt_tuning_get(rt_ctx* c, int key, int* value)
{
    RT_CHECK_CTX(c);
    if (!value) return RT_ERR_ARG;
    switch (key)
    {
        case 0: case 1: case 2: case 3: *value = c->tune_tile_mode[key]; break;
        case 4: *value = c->tune_spatial_lds; break;
        case 5: *value = c->bvh_builder; break;
        case 6: *value = c->pt_wavefront; break;
        case 7: *value = c->bvh_bfs_records; break;
        case 8: *value = c->tune_spatial_variant; break;
        case 9: *value = c->tune_spatial_waves; break;
        case 10: *value = c->ploc_radius; break;
        case 11: *value = c->tune_defer_vis; break;
        case 12: *value = c->tune_ris_pipe; break;
        case 13: *value = c->tune_ws; break;
        case 14: *value = c->tune_spec; break;
        case 15: *value = c->tune_stream; break;
        case 16: *value = c->tune_ws_primary; break;
        case 17: *value = c->tune_tail; break;
        case 18: *value = c->tune_mark_quick; break;
        case 19: *value = c->tune_mark_window; break;
        case 20: *value = c->tune_fuse_tonemap; break;
        case 21: *value = c->tune_mark_cache; break;
        case 22: *value = c->tune_spec_free; break;
        case 23: *value = c->tune_fuse_final; break;
        case 24: *value = c->tune_half_raycast; break;
        case 25: *value = c->tune_fuse_raycast; break;
        case 26: *value = c->tune_mark_split; break;
        default: RT_FAIL(c, RT_ERR_ARG, "bad tuning key %d", key);
    }
    return RT_OK;
}
#ifndef RT_BUILD_ID
#define RT_BUILD_ID "unknown"
#endif
const char* rt_build_id(void) { return RT_BUILD_ID; }
/* which traversal rt_trace_closest / rt_trace_stats exercise: 0 = 4-wide quantised BVH with the
 * LDS stack (the one every frame kernel uses), 1 = binary LBVH with the stackless trail. */
int rt_trace_mode(rt_ctx* c, int mode)
{
    RT_CHECK_CTX(c);
    if (mode < 0 || mode > 7) RT_FAIL(c, RT_ERR_ARG, "mode must be 0..7");
#ifndef RT_EXPERIMENTS
    if ((mode >= 1 && mode <= 3) || mode == 7) RT_FAIL(c, RT_ERR_UNSUPPORTED, "trace mode %d (binary stackless walk / ray queue / four lanes per ray) is an A/B form of librestir_rt_exp.so", mode);
#endif
    c->trace_mode = mode;
    return RT_OK;
}

int rt_math_eval(rt_ctx* c, int fn, const float* in, uint32_t n, float* out)
{
    RT_CHECK_CTX(c);
    if (n == 0) return RT_OK;
    const size_t nin = (fn == 26 || fn == 33 || fn == 35) ? 2 : ((fn == 31 || fn == 32) ? 12 : ((fn == 36 || fn == 37) ? 3 : (fn == 38 ? 14 : 1))); /* floats per item */
    float *d_i = nullptr, *d_o = nullptr;
    RT_HIP(c, hipMalloc(&d_i, (size_t)n * 4 * nin));
    RT_HIP(c, hipMalloc(&d_o, (size_t)n * 4));
    RT_HIP(c, hipMemcpyAsync(d_i, in, (size_t)n * 4 * nin, hipMemcpyHostToDevice, c->stream));
    k_math_eval<<<(n + 255) / 256, 256, 0, c->stream>>>(fn, d_i, (int)n, d_o);
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipMemcpyAsync(out, d_o, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    hipFree(d_i); hipFree(d_o);
    return RT_OK;
}

} /* extern "C" */
