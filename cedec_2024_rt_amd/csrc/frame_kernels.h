/*
 * frame_kernels.h — every gfx950 kernel of the hot path (included once, by restir_rt.hip).
 *
 * Per pixel restatements of examples/10_restir_di/10_restir_di.cu and common/kernels/common.cu
 * (file:line cited at each kernel) over the layouts of rt_device.h, the path tracers of
 * examples/07_pt and examples/09_ris, the layout converters behind rt_upload / rt_download,
 * the sparse-halo kernels of the multi-GPU strips, and measurement / test utilities.
 */
#pragma once
#include "bvh.h"
#include "rt_device.h"

using namespace rt;

/* ------------------------------------------------------------------ params */

struct FrameParams
{
    int W, H;           /* full image */
    int row0, row1;     /* global storage rows processed by this launch */
    int rowb0, rowb1;   /* optional second row range of the same launch (rowb0 >= rowb1: none): a strip's two
                           boundary bands run as ONE launch — a launch that fits the GPU in one round lasts as
                           long as its slowest wavefront, so two of them back to back cost twice that */
    int lrow0, lrows;   /* global row of local buffer row 0, local rows held */
    int frame, pass;
    f3 eye;
    f3 rg_origin, rg_right, rg_up;
    int n_lights;
    /* options (common/options.hpp) */
    int accumulate, ris_sample_count, use_temporal, use_spatial, spatial_count, vis_reuse;
    float spatial_radius;
    int tile_mode; /* workgroup -> tile order inside an XCD's band: 0 row-major, 1 column-major; 2 / 3: tile rows interleaved over the XCDs, row- / column-major */
    uint32_t ownv_tag; /* own-visibility flags are written / trusted under this tag only (rt_device.h); 0 = never */
    /* rt_walk_stats (measurement, off = nullptr: one wave-uniform test per kernel): 4 counters per kernel slot
     * {rays the reference traces here, walked through the BVH, settled by the one-triangle self-occlusion test, not evaluated
     * (answer known from the own-visibility flags, or unobservable)}; slots WALK_RAYCAST .. WALK_RESOLVE */
    unsigned long long* stats;
#ifdef RT_EXPERIMENTS
    /* rt_exp_wave_clock (measurement, experiments library only): two words per wavefront of the chosen kernel — the constant
     * 100-MHz clock when it started, and when its last lane left | hardware id << 40 (XCD, SE, CU, SIMD) */
    unsigned long long* wave_clock;
    /* rt_exp_tile_perm (measurement): workgroup b of the chosen kernel takes the place of workgroup tile_perm[b] in the launch's tile
     * order (a permutation that keeps b % 8, i.e. the XCD): what would longest-first dispatch gain? */
    const uint32_t* tile_perm;
#endif
};
#ifdef RT_EXPERIMENTS
struct WaveClock
{
    unsigned long long* out;
    unsigned long long t0;
    RT_DEV explicit WaveClock(const FrameParams& P) : out(P.wave_clock), t0(0)
    {
        if (out) t0 = wall_clock64();
    }
    RT_DEV ~WaveClock()
    {
        if (!out) return;
        /* lanes leave a kernel at different returns: the first lane of every leaving group notes the time, the latest stands */
        const unsigned long long act = __ballot(true);
        if ((int)(threadIdx.x & 63) != __ffsll((long long)act) - 1) return;
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
        /* HW_ID: simd [5:4], cu [11:8], sh [12], se [15:13] */
        const unsigned long long id = ((unsigned long long)(xcc & 15u) << 12) | (((hw >> 13) & 7u) << 9) | (((hw >> 12) & 1u) << 8) | (((hw >> 8) & 15u) << 4) | (((hw >> 4) & 3u) << 2);
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = t0;
        atomicMax(&out[2 * w + 1], (wall_clock64() & 0xffffffffffull) | (id << 40));
    }
};
#define RT_WAVE_CLOCK(P) WaveClock wave_clock_(P)
#else
#define RT_WAVE_CLOCK(P)
#endif
enum { WALK_RAYCAST = 0, WALK_GENERATE = 1, WALK_SPATIAL = 2, WALK_RESOLVE = 3 };
/* one ray per lane at most; works under any exec mask (ballots count the active lanes) */
RT_DEV void count_walk_flags(unsigned long long* __restrict__ st, bool ref, bool walked, bool self, bool skipped)
{
    const unsigned long long b0 = __ballot(ref), b1 = __ballot(walked), b2 = __ballot(self), b3 = __ballot(skipped);
    const unsigned long long act = __ballot(true);
    if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1)
    {
        if (b0) atomicAdd(&st[0], (unsigned long long)__popcll(b0));
        if (b1) atomicAdd(&st[1], (unsigned long long)__popcll(b1));
        if (b2) atomicAdd(&st[2], (unsigned long long)__popcll(b2));
        if (b3) atomicAdd(&st[3], (unsigned long long)__popcll(b3));
    }
}
/* several rays per lane; every lane of the wavefront must be active */
RT_DEV void count_walk_counts(unsigned long long* __restrict__ st, uint32_t ref, uint32_t walked, uint32_t self, uint32_t skipped)
{
    for (int off = 32; off > 0; off >>= 1)
    {
        ref += __shfl_down(ref, off); walked += __shfl_down(walked, off);
        self += __shfl_down(self, off); skipped += __shfl_down(skipped, off);
    }
    if ((threadIdx.x & 63) == 0)
    {
        if (ref) atomicAdd(&st[0], (unsigned long long)ref);
        if (walked) atomicAdd(&st[1], (unsigned long long)walked);
        if (self) atomicAdd(&st[2], (unsigned long long)self);
        if (skipped) atomicAdd(&st[3], (unsigned long long)skipped);
    }
}

/* A/B switch of the deferred barycentric divisions (trace_wide's tv, r06): -DRT_NO_DEFER_BARY=0 computes the winner's
 * u, v once after the walk. MEASURED, NO GAIN (tools/lib_ab.py, three builds side by side in one process, profiles/r06_leaf_test_ab.txt):
 * un-pipelined frame 1.2528 (u, v kept in the walk) / 1.2567 (deferred + range test against the best hit) / 1.2562 ms (deferred only),
 * 3840x2160 4.660 / 4.644 / 4.683 - the divisions sit behind a branch few wavefront passes take (a hit must pass the range AND the
 * three sub-area tests in some lane). Default: as r01-r05. */
#ifndef RT_NO_DEFER_BARY
#define RT_NO_DEFER_BARY 1
#endif
#define RT_BARY_TV(S) (RT_NO_DEFER_BARY ? nullptr : (S).bvh.tv)
struct SceneView
{
    BvhView bvh;   /* binary LBVH, stackless trail traversal (kept for A/B measurements) */
    WideView wide; /* production traversal structure */
    const float4* __restrict__ trimat; /* 2 per triangle: {Kd.xyz, bits(emissive?)}, {Ke.xyz, 0} */
    const float4* __restrict__ lights;   /* 3 per light, see k_light_table */
    const float4* __restrict__ light_ke; /* 1 per light: {Ke.xyz, 0} */
};

#ifndef RT_TILE_W
#define RT_TILE_W 32 /* pixels per tile row: 32x8, 16x16 or 8x32 tiles of 256 pixels */
#endif
#ifndef RT_LIGHT_STRIDE
#define RT_LIGHT_STRIDE 4 /* float4 per light record: 3 = normal recomputed per candidate, 4 = normal stored (A/B: 4 is 3 % faster in the fused kernel) */
#endif
static_assert(RT_LIGHT_STRIDE == 4, "the 48-B light record (normal recomputed per candidate) is no longer maintained: kernels read the 4th float4");
constexpr int TILE_W = RT_TILE_W, TILE_H = 256 / RT_TILE_W, BLOCK = 256;
constexpr int TILE_W_LOG2 = RT_TILE_W == 32 ? 5 : (RT_TILE_W == 16 ? 4 : 3);
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 1 /* min waves per SIMD requested for the tracing kernels (register budget) */
#endif

/* XCD-aware workgroup -> tile -> pixel. Returns false for threads outside the row range.
 * Workgroup b runs on XCD b % 8 (round-robin dispatch); slot b / 8 walks that XCD's band of
 * tile rows either row by row (mode 0) or column by column (mode 1: the set of tiles in flight
 * on an XCD is then ~13 tiles wide x the band height instead of full-width x 4 rows, which is
 * what keeps the spatial pass's neighbour window inside the 4 MiB L2). */
/* The tracing kernels (raycast, generate_candidate, resolve) run 64-thread workgroups = one wavefront
 * on an 8x8 tile: a wavefront that finishes gives its LDS stack (6 KB) and its slot back at once,
 * whereas the four wavefronts of a 256-thread workgroup hold 24 KB until the slowest is done
 * (A/B: raycast -3 %, candidates -3 %, resolve -1 %). The spatial pass keeps 256 threads on 32x8
 * tiles (its L2 window and occupancy limiter were tuned for them; 64-thread groups: +4 %). */
#ifndef RT_TRACE_BLOCK
#define RT_TRACE_BLOCK 64
#endif
constexpr int TRACE_BLOCK = RT_TRACE_BLOCK;
template <int TB> struct TileShape { static constexpr int W = TILE_W, H = TILE_H; };
template <> struct TileShape<64> { static constexpr int W = 8, H = 8; };

/* workgroup index b, thread index t (the persistent kernels walk a job counter instead of blockIdx / threadIdx) */
template <int TB = BLOCK>
RT_DEV bool tile_pixel_at(const FrameParams& P, int b, const int t, int& x, int& row)
{
    constexpr int TILE_W = TileShape<TB>::W, TILE_H = TileShape<TB>::H;
#ifdef RT_EXPERIMENTS
    if (TB == TRACE_BLOCK && P.tile_perm) b = (int)P.tile_perm[b];
#endif
    const int tiles_x = (P.W + TILE_W - 1) / TILE_W;
    const int tiles_ya = (P.row1 - P.row0 + TILE_H - 1) / TILE_H;
    const int tiles_y = tiles_ya + (P.rowb1 > P.rowb0 ? (P.rowb1 - P.rowb0 + TILE_H - 1) / TILE_H : 0);
    int tx, ty;
    if (P.tile_mode == 1)
    {
        const int band_rows = (tiles_y + 7) / 8;
        const int slot = b >> 3;
        tx = slot / band_rows;
        ty = (b & 7) * band_rows + (slot - tx * band_rows);
        if (tx >= tiles_x || ty >= tiles_y) return false;
    }
    else if (P.tile_mode >= 4)
    {
        /* workgroup b = tile b: neighbouring tiles on different XCDs; 4: row by row, 5: in stripes 32 tiles wide */
        if (P.tile_mode == 4) { ty = b / tiles_x; tx = b - ty * tiles_x; }
        else if (P.tile_mode >= 6)
        {
            /* row-major in runs of R tiles: run r on XCD r % 8 (6: R = 4, 7: R = 16) */
            const int R = P.tile_mode == 6 ? 4 : 16, slot = b >> 3;
            const int tile = ((slot / R) * 8 + (b & 7)) * R + slot % R;
            ty = tile / tiles_x; tx = tile - ty * tiles_x;
        }
        else
        {
            const int per_stripe = 32 * tiles_y, s = b / per_stripe, r = b - s * per_stripe;
            const int w = tiles_x - 32 * s < 32 ? tiles_x - 32 * s : 32; /* the last stripe may be narrower */
            if (w <= 0) return false;
            ty = r / w; tx = 32 * s + (r - ty * w);
        }
        if (tx >= tiles_x || ty >= tiles_y) return false;
    }
    else if (P.tile_mode >= 2)
    {
        /* r05: INTERLEAVED tile rows — XCD k takes tile rows k, k + 8, k + 16, ... (every XCD sees a sample of the whole image: the
         * eight bands of modes 0 / 1 cost what their part of the scene costs, and the launch ends with the dearest band's XCD),
         * walked row by row (2) or column by column (3) */
        const int band_rows = (tiles_y + 7) / 8;
        const int slot = b >> 3;
        int j;
        if (P.tile_mode == 2) { j = slot / tiles_x; tx = slot - j * tiles_x; }
        else { tx = slot / band_rows; j = slot - tx * band_rows; }
        ty = 8 * j + (b & 7);
        if (tx >= tiles_x || ty >= tiles_y) return false;
    }
    else
    {
        const int n_tiles = tiles_x * tiles_y;
        const int per_xcd = (n_tiles + 7) / 8;
        const int slot = b >> 3;
        if (slot >= per_xcd) return false; /* the grid is sized for either order */
        const int tile = (b & 7) * per_xcd + slot;
        if (tile >= n_tiles) return false;
        ty = tile / tiles_x;
        tx = tile - ty * tiles_x;
    }
#ifndef RT_WAVE_8X8
#define RT_WAVE_8X8 1
#endif
    /* tile rows 0 .. tiles_ya-1 belong to the first row range, the rest to the second */
    const bool second = ty >= tiles_ya;
    const int base = second ? P.rowb0 + (ty - tiles_ya) * TILE_H : P.row0 + ty * TILE_H;
    const int end = second ? P.rowb1 : P.row1;
    if (TB == 64)
    {
        x = tx * 8 + (t & 7);
        row = base + (t >> 3);
    }
    else
    {
#if RT_WAVE_8X8 && RT_TILE_W == 32
        /* wavefront w of the workgroup covers the 8x8 sub-block w of the 32x8 tile */
        x = tx * TILE_W + 8 * (t >> 6) + (t & 7);
        row = base + ((t >> 3) & 7);
#else
        x = tx * TILE_W + (t & (TILE_W - 1));
        row = base + (t >> TILE_W_LOG2);
#endif
    }
    return x < P.W && row < end;
}
template <int TB = BLOCK>
RT_DEV bool tile_pixel(const FrameParams& P, int& x, int& row)
{
    return tile_pixel_at<TB>(P, (int)blockIdx.x, (int)threadIdx.x, x, row);
}
static inline int tile_grid(int W, int rows, int tile_w = TILE_W, int tile_h = TILE_H, int rows_b = 0)
{
    /* covers every order: modes 1-3 need 8 * ceil(tiles_y/8) * tiles_x workgroups */
    const int tx = (W + tile_w - 1) / tile_w, ty = (rows + tile_h - 1) / tile_h + (rows_b > 0 ? (rows_b + tile_h - 1) / tile_h : 0);
    const int a = ((tx * ty + 127) / 128) * 128 /* whole runs of 16 tiles on 8 XCDs (mode 7) */, b = 8 * ((ty + 7) / 8) * tx;
    return a > b ? a : b;
}

/* common/core.hpp:189-207: surface point + normal flipped toward the eye [parity] */
RT_DEV void surface_info(const BvhView& bvh, int tri, float u, float v, f3 eye, f3& p, f3& n)
{
    f3 v0, v1, v2;
    load_tri(bvh.tv, tri, v0, v1, v2);
    p = (1.0f - u - v) * v0 + u * v1 + v * v2;
    n = tri_normal(v0, v1, v2);
    const f3 view = normalize(eye - p);
    if (dot(view, n) < 0.0f) n = -n;
}

/* G-buffer entry from a Visibility record */
RT_DEV void gbuffer_make(const SceneView& S, const FrameParams& P, float u, float v, int index, float4& G0, float4& G1)
{
    if (index < 0)
    {
        G0 = make_float4(0.0f, 0.0f, 0.0f, as_float(-1));
        G1 = make_float4(0.0f, 0.0f, 0.0f, as_float(0u));
        return;
    }
    const bool emissive = as_uint(S.trimat[2 * (size_t)index].w) != 0u;
    f3 p, n;
    surface_info(S.bvh, index, u, v, P.eye, p, n);
    G0 = make_float4(p.x, p.y, p.z, as_float(index));
    G1 = make_float4(n.x, n.y, n.z, as_float(emissive ? GB_EMISSIVE : GB_SHADED));
}
RT_DEV void gbuffer_write(const SceneView& S, const FrameParams& P, float4* __restrict__ g0,
                          float4* __restrict__ g1, size_t li, float u, float v, int index)
{
    float4 G0, G1;
    gbuffer_make(S, P, u, v, index, G0, G1);
    g0[li] = G0;
    g1[li] = G1;
}
/* common/camera.hpp:27-35 shoot(xi / W, yi / H): no jitter, no + 0.5 (10_restir_di.cu:17-24) */
RT_DEV f3 primary_direction(const FrameParams& P, int x, int yi)
{
    const float u = (float)x / (float)P.W, v = (float)yi / (float)P.H;
    const f3 forward = normalize(cross(P.rg_up, P.rg_right));
    const f3 to = P.rg_origin + forward + mix(-P.rg_right, P.rg_right, u) + mix(P.rg_up, -P.rg_up, v);
    return normalize(to - P.rg_origin);
}

/* -------------------------------------------------------------------- raycast */
/* examples/10_restir_di/10_restir_di.cu:9-34 (+ common/camera.hpp:27-35) */
#ifndef RT_RAYCAST_WAVES
#define RT_RAYCAST_WAVES RT_TRACE_WAVES
#endif
#ifndef RT_RAYCAST_WS_WAVES
#define RT_RAYCAST_WS_WAVES RT_RAYCAST_WAVES
#endif
/* WS: the work-sharing closest-hit walk (bvh.h closest_ws; rt_tuning key 16) */
template <bool WS>
__global__ __launch_bounds__(TRACE_BLOCK, WS ? RT_RAYCAST_WS_WAVES : RT_RAYCAST_WAVES) void k_raycast(SceneView S, FrameParams P, float4* __restrict__ vis,
                                                    float4* __restrict__ g0, float4* __restrict__ g1)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[(WS ? WIDE_LDS_ROWS_CLOSEST : WIDE_LDS_STACK) * TRACE_BLOCK];
    RT_WAVE_CLOCK(P);
    int x, row;
    if (!tile_pixel<TRACE_BLOCK>(P, x, row)) return;
    const int yi = P.H - 1 - row;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
    if (P.stats) count_walk_flags(P.stats + 4 * WALK_RAYCAST, true, true, false, false); /* every primary ray is walked */

    const f3 rd = primary_direction(P, x, yi);

    Hit h;
    h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.prim = -1;
    if (WS) closest_ws<TRACE_BLOCK>(S.wide, S.bvh.tv, s_stack, P.rg_origin, rd, 0.0f, kFltMax, h);
    else trace_wide<false, false, TRACE_BLOCK>(S.wide, s_stack, P.rg_origin, rd, 0.0f, kFltMax, h, nullptr, RT_BARY_TV(S));
    vis[li] = make_float4(h.u, h.v, as_float(h.prim), as_float(0));
    gbuffer_write(S, P, g0, g1, li, h.u, h.v, h.prim);
}

#ifdef RT_EXPERIMENTS /* measured slower: profiles/r06_quad_walk_ab.txt */
/* rt_tuning key 16 = 2 (r06): FOUR LANES PER PRIMARY RAY (bvh.h closest_quad). Workgroup 4 T + q = the 4 x 4 quadrant q of the 8 x 8 tile
 * workgroup T of k_raycast takes; lane 4 r + k = child k of the quadrant's ray r. Four times the wavefronts of k_raycast, each with a
 * quarter of the rays and a step less than half as long: for launches that last as long as their slowest wavefront (a strip of the
 * multi-GPU frame: half a generation of wavefronts). Same Visibility record and G-buffer, byte for byte. */
__global__ __launch_bounds__(TRACE_BLOCK, RT_RAYCAST_WAVES) void k_raycast_quad(SceneView S, FrameParams P, float4* __restrict__ vis,
                                                                             float4* __restrict__ g0, float4* __restrict__ g1)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[QUAD_LDS_WORDS];
    RT_WAVE_CLOCK(P);
    const int t = (int)threadIdx.x, tile = (int)(blockIdx.x >> 2), q = (int)(blockIdx.x & 3), r = t >> 2;
    const int in_tile = ((q >> 1) * 4 + (r >> 2)) * 8 + (q & 1) * 4 + (r & 3);
    int x = 0, row = P.row0;
    const bool has = tile_pixel_at<TRACE_BLOCK>(P, tile, in_tile, x, row);
    const int yi = P.H - 1 - row;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
    if (P.stats) count_walk_flags(P.stats + 4 * WALK_RAYCAST, has && (t & 3) == 0, has && (t & 3) == 0, false, false);
    const f3 rd = has ? primary_direction(P, x, yi) : F3(0.0f, 0.0f, 1.0f);
    Hit h;
    h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.prim = -1;
    closest_quad(S.wide, S.bvh.tv, s_stack, P.rg_origin, rd, 0.0f, kFltMax, has, h);
    if (!has || (t & 3) != 0) return; /* the four lanes hold the same hit: the first one writes it */
    vis[li] = make_float4(h.u, h.v, as_float(h.prim), as_float(0));
    gbuffer_write(S, P, g0, g1, li, h.u, h.v, h.prim);
}
/* rt_trace_closest mode 7: rays from a list through closest_quad, 16 per wavefront */
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace_closest_quad(SceneView S, const float* __restrict__ rays, int n, float* __restrict__ hits)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[QUAD_LDS_WORDS];
    const int i = (int)blockIdx.x * 16 + (int)(threadIdx.x >> 2);
    const bool has = i < n;
    const float* r = rays + 8 * (size_t)(has ? i : 0);
    Hit h;
    h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.prim = -1;
    closest_quad(S.wide, S.bvh.tv, s_stack, F3(r[0], r[1], r[2]), F3(r[3], r[4], r[5]), r[6], r[7], has, h);
    if (!has || (threadIdx.x & 3) != 0) return;
    float* o = hits + 4 * (size_t)i;
    o[0] = h.t; o[1] = h.u; o[2] = h.v; o[3] = as_float(h.prim);
}

#endif /* RT_EXPERIMENTS */

#ifdef RT_EXPERIMENTS /* rt_tuning key 24 (r05): a prototype for launches of less than one generation of wavefronts */
/* HALF-DENSITY raycast: a wavefront carries 32 primary rays (the upper or lower 8 x 4 half of an 8 x 8 tile) in lanes 0-31 and 32
 * rayless lanes that only ever HELP — the work-sharing walk hands them parts of the busy lanes' stacks (a rayless lane enters with
 * tmin > tmax: its root culls everything, it is idle at the first sharing check). Twice the wavefronts, each with half the rays and
 * twice the hands per ray: a 135-row strip's raycast is 4 050 wavefronts on a GPU that holds 8 192, lasts as long as its slowest
 * wavefront, and keeps the vector ALUs 40 % busy — the idle half of the machine can work on the same rays. Same rays, same hits. */
__global__ __launch_bounds__(TRACE_BLOCK, RT_RAYCAST_WS_WAVES) void k_raycast_half(SceneView S, FrameParams P, float4* __restrict__ vis,
                                                                                float4* __restrict__ g0, float4* __restrict__ g1)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[WIDE_LDS_ROWS_CLOSEST * TRACE_BLOCK];
    const int t = (int)threadIdx.x, tile = (int)(blockIdx.x >> 1), half = (int)(blockIdx.x & 1);
    int x = 0, row = P.row0;
    const bool has = t < 32 && tile_pixel_at<TRACE_BLOCK>(P, tile, t + 32 * half, x, row);
    const int yi = P.H - 1 - row;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
    if (P.stats && has) count_walk_flags(P.stats + 4 * WALK_RAYCAST, true, true, false, false);
    const float u = (float)x / (float)P.W, v = (float)yi / (float)P.H;
    const f3 forward = normalize(cross(P.rg_up, P.rg_right));
    const f3 to = P.rg_origin + forward + mix(-P.rg_right, P.rg_right, u) + mix(P.rg_up, -P.rg_up, v);
    const f3 rd = normalize(to - P.rg_origin);
    Hit h;
    h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.prim = -1;
    closest_ws<TRACE_BLOCK>(S.wide, S.bvh.tv, s_stack, P.rg_origin, has ? rd : F3(0.0f, 0.0f, 1.0f), has ? 0.0f : 1.0f, has ? kFltMax : 0.0f, h);
    if (!has) return;
    vis[li] = make_float4(h.u, h.v, as_float(h.prim), as_float(0));
    gbuffer_write(S, P, g0, g1, li, h.u, h.v, h.prim);
}
#endif /* RT_EXPERIMENTS */

/* rebuild the G-buffer from an uploaded Visibility buffer */
__global__ __launch_bounds__(BLOCK) void k_gbuffer_from_vis(SceneView S, FrameParams P,
                                                             const float4* __restrict__ vis,
                                                             float4* __restrict__ g0, float4* __restrict__ g1)
{
    int x, row;
    if (!tile_pixel(P, x, row)) return;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
    const float4 vv = vis[li];
    gbuffer_write(S, P, g0, g1, li, vv.x, vv.y, as_int(vv.z));
}

/* ------------------------------------------------------- target function helper */
/* common/reservoir.hpp:42-59 */
template <bool SHADOWED, int STRIDE = BLOCK>
RT_DEV float target_function(const SceneView& S, uint32_t* s_stack, f3 op, f3 on, f3 hp, f3 hn, float lum, int own_tri = -1)
{
    if (SHADOWED)
    {
        const float brdf = 1.0f / kPI;
        const float G = geometry_term(op, on, hp, hn);
        const float V = check_visibility_wide<STRIDE>(S.wide, s_stack, op, on, hp, true, S.bvh.tv, own_tri) ? 1.0f : 0.0f;
        return brdf * G * V * lum;
    }
    return target_unshadowed(op, on, hp, hn, lum);
}

/* One RIS candidate's weight = evaluate_target_function(...) / light_sample.pdf (10_restir_di.cu:98-105: p_hat = (1/PI) G lum,
 * common/reservoir.hpp:42-59, unshadowed always, :104) as ONE guarded expression (r05): the shared-reciprocal forms of rt_device.h
 * (normalize, / sqr_dist, / pdf) each sit behind a range test on their operands; nested, that is three data-dependent branches per
 * candidate around 60 instructions. Here the three forms run unconditionally — plain arithmetic, no traps, garbage in = garbage out —
 * and ONE test of all their operand ranges afterwards decides whether the result stands; if not (zero / subnormal / huge operands:
 * never on the bench scene) the candidate is evaluated again by the nested-guard functions. Same operations on the same operands in
 * the same order as target_unshadowed + div_pdf when every guard passes, and those functions themselves otherwise. */
#ifndef RT_RIS_ONE_GUARD
#define RT_RIS_ONE_GUARD 1
#endif
RT_DEV float div_pdf(float p_hat, float pdf, float r1);
RT_DEV float ris_weight(f3 sp, f3 sn, f3 lp, f3 ln, float lum, float pdf, float r1_pdf)
{
#if RT_FAST_DIV && RT_RIS_ONE_GUARD
    const f3 v = lp - sp;
    const float sqr_dist = dot(v, v);
    const float len = sqrt_in_range(sqr_dist);
    const float rl = rcp_refined(len);
    const f3 vh = F3(div_by(v.x, len, rl), div_by(v.y, len, rl), div_by(v.z, len, rl));
    const float num = fabsf(dot(vh, sn)) * fabsf(dot(-vh, ln));
    const float G = div_by(num, sqr_dist, rcp_refined(sqr_dist));
    const float brdf = 1.0f / kPI;
    const float p_hat = brdf * G * lum;
    const float w = div_by(p_hat, pdf, r1_pdf);
    /* `&`, not `&&`: ONE condition from five compare masks (with `&&` the compiler rebuilds the nested branches and sinks the
     * arithmetic back into them) */
    const unsigned ok = (unsigned)div_den_ok(sqr_dist) & (unsigned)(__builtin_fminf(__builtin_fminf(fabsf(v.x), fabsf(v.y)), fabsf(v.z)) >= kDivNumLo) &
                        (unsigned)div_num_ok(num) & (unsigned)(r1_pdf == r1_pdf) & (unsigned)div_num_ok(p_hat);
    if (ok) return w;
#endif
    return div_pdf(target_unshadowed(sp, sn, lp, ln, lum), pdf, r1_pdf);
}
/* weight = p_hat / pdf (10_restir_di.cu:98-105) with the pdf's refined reciprocal from the light table (k_light_table) */
RT_DEV float div_pdf(float p_hat, float pdf, float r1)
{
#if RT_FAST_DIV
    if (r1 == r1 && div_num_ok(p_hat)) return div_by(p_hat, pdf, r1);
#endif
    return p_hat / pdf;
}
RT_DEV void res_take_sample(Res& r, const Res& o)
{
    r.hit_p = o.hit_p; r.hit_n = o.hit_n; r.org_p = o.org_p; r.org_n = o.org_n;
    r.rad = o.rad; r.lum = o.lum; r.vis = o.vis;
}

/* temporal merge of 10_restir_di.cu:177-233; r = current, pr = previous frame, same pixel */
/* evaluate_target_function with the shadow term already known (common/reservoir.hpp:42-59) */
RT_DEV float target_shadowed(f3 op, f3 on, f3 hp, f3 hn, float lum, float V)
{
    return (1.0f / kPI) * geometry_term(op, on, hp, hn) * V * lum;
}

/* The (<= 2) distinct shadow rays of a pixel's candidate + temporal step under the shadowed target
 * function: surface -> current sample (p-hat of :115-123, the visibility-reuse ray of :127-131 and, if
 * the current sample survives the merge, the p-hat of :225-229 are the same ray) and surface ->
 * previous sample (:195-199; again :225-229 if it wins). The previous sample's ray is not walked
 * when its weight is 0 whatever the answer. V = 1 visible, 0 occluded. */
template <int STRIDE = BLOCK>
RT_DEV void temporal_rays(const SceneView& S, uint32_t* s_stack, const FrameParams& P, f3 sp, f3 sn, const Res& r,
                          const Res& pr, bool with_prev, float& V_cur, float& V_prev, int own_tri = -1)
{
    const f3 tgt[2] = {r.hit_p, pr.hit_p};
    const bool moot = !with_prev || pr.ucw == 0.0f || (P.vis_reuse && !pr.vis);
    const uint32_t occl = occluded_batch<2, STRIDE>(S.wide, s_stack, sp, sn, tgt, moot ? 1u : 3u, S.bvh.tv, own_tri);
    V_cur = (occl & 1u) ? 0.0f : 1.0f;
    V_prev = (occl & 2u) ? 0.0f : 1.0f;
}

template <bool SHADOWED>
RT_DEV bool temporal_merge(const FrameParams& P, int x, int yi, f3 sp, f3 sn, Res& r, Res pr, float V_cur, float V_prev)
{
    bool took_prev = false;
    PCG rng = pcg_init(hashPCG4((uint32_t)x, (uint32_t)yi, (uint32_t)P.frame, 1u), 0);
    const int cap = 20 * P.ris_sample_count;
    pr.M = pr.M < cap ? pr.M : cap;
    float p_hat_y = SHADOWED ? target_shadowed(sp, sn, pr.hit_p, pr.hit_n, pr.lum, V_prev)
                             : target_unshadowed(sp, sn, pr.hit_p, pr.hit_n, pr.lum);
    if (P.vis_reuse) p_hat_y *= pr.vis ? 1.0f : 0.0f;
    pr.M = scale_M(pr.M, rejection_heuristics(r.org_p, r.org_n, pr.org_p, pr.org_n, P.eye));
    const float weight = p_hat_y * pr.ucw * (float)pr.M;
    const float u = rng.uniformf();
    r.w_sum += weight;
    r.M += pr.M;
    float V = V_cur;
    if (reservoir_accept(u, weight, r.w_sum))
    {
        res_take_sample(r, pr);
        V = V_prev;
        took_prev = true;
    }
    const float p_hat = SHADOWED ? target_shadowed(sp, sn, r.hit_p, r.hit_n, r.lum, V)
                                 : target_unshadowed(sp, sn, r.hit_p, r.hit_n, r.lum);
    r.ucw = p_hat > 0.0f ? r.w_sum / ((float)r.M * p_hat) : 0.0f;
    return took_prev;
}

#ifndef RT_RESOLVE_WAVES_FWD
#define RT_RESOLVE_WAVES_FWD 6 /* k_candidate_visibility: the register budget of k_resolve (one shadow ray per lane) */
#endif
/* --------------------------------------------------------- generate_candidate */
/* examples/10_restir_di/10_restir_di.cu:36-135; with FUSE_TEMPORAL also :137-237 on the
 * value still in registers (the reference round-trips it through reservoir_buffer0). */
/* DEFER (fused, unshadowed only): the visibility-reuse ray of :127-131 is NOT walked here. Its answer is observable
 * only if the candidate's sample survives the temporal merge — if the previous frame's sample is taken, the stored
 * visibility is that sample's (reservoir.hpp:36) and the ray was dead work (25 % of the rays in the bench scene:
 * 1.465 M of 1.957 M candidates survive). Survivors are appended to a queue (wave ballot + one atomic) and
 * k_candidate_visibility walks them with full wavefronts and sets the bit. A/B only (rt_tuning key 11): the work-sharing
 * kernel gets the same saving without a queue (LATE below). */
/* the work-sharing variant allocates 102 VGPRs unconstrained (4 wavefronts per SIMD); held to the 96 of the plain
 * kernel (5 per SIMD) it is 3 % faster (A/B on the GPU, profiles/r02_ws_register_budgets.txt) */
#ifndef RT_GENERATE_WS_WAVES
#define RT_GENERATE_WS_WAVES 5
#endif
#ifndef RT_GENERATE_SH_WAVES
#define RT_GENERATE_SH_WAVES RT_TRACE_WAVES /* the shadowed-target kernel (110 VGPRs) */
#endif
template <bool STREAM = true>
RT_DEV void wave_scatter_records(float4* __restrict__ rec, const int idx, float4* s_wave, const int lane, const float4& q0, const float4& q1,
                                 const float4& q2, const float4& q3);
RT_DEV void wave_gather_records_at(const float4* __restrict__ rec, const uint32_t idx, float4* s_wave, const int lane, float4& q0, float4& q1,
                                   float4& q2, float4& q3);
RT_DEV void wave_gather_request_at(const float4* __restrict__ rec, const uint32_t idx, float4* s_wave, const int lane);
RT_DEV void wave_gather_finish(float4* s_wave, const int lane, float4& q0, float4& q1, float4& q2, float4& q3);
RT_DEV void wave_gather_records(const float4* q, float4* s_wave, const int lane, float4& q0, float4& q1, float4& q2, float4& q3);
#ifndef RT_RIS_PIPE
#define RT_RIS_PIPE 1 /* 0: request the light record of candidate i after its draws and wait for it (A/B) */
#endif
#ifndef RT_SHADOWED_RIS_COOP
#define RT_SHADOWED_RIS_COOP 1 /* 0: the shadowed-target generate kernel gathers its light records per lane (A/B) */
#endif
#ifndef RT_RIS_COOP
#define RT_RIS_COOP 1 /* 0: per-lane light record gathers in the work-sharing generate kernel too (A/B) */
#endif
/* RAYCAST (r05, rt_tuning key 25): the kernel traces the pixel's primary ray first (raycast, 10_restir_di.cu:9-34) and writes the
 * Visibility record and the G-buffer (`vis_w`, `g0_w`, `g1_w`; `g0` / `g1` are not read) — generate_candidate needs the raycast of its
 * OWN pixel only, and as two launches the second waits until the first has drained (its last wavefront starts at 216 of 257 us). */
template <bool FUSE_TEMPORAL, bool SHADOWED, bool DEFER = false, bool PIPE = false, bool WS = false, bool RAYCAST = false>
__global__ __launch_bounds__(TRACE_BLOCK, WS ? RT_GENERATE_WS_WAVES : (SHADOWED ? RT_GENERATE_SH_WAVES : RT_TRACE_WAVES)) void k_generate_candidate(
    SceneView S, FrameParams P, const float4* __restrict__ g0, const float4* __restrict__ g1,
    const float4* __restrict__ prev_rec, const float4* __restrict__ prev_rad, float4* __restrict__ out_rec,
    float4* __restrict__ out_rad, uint32_t* __restrict__ vis_queue = nullptr, unsigned int* __restrict__ vis_count = nullptr,
    float4* __restrict__ vis_w = nullptr, float4* __restrict__ g0_w = nullptr, float4* __restrict__ g1_w = nullptr)
{
    static_assert(!DEFER || (FUSE_TEMPORAL && !SHADOWED), "deferred visibility: fused unshadowed kernel only");
    static_assert(!RAYCAST || (WS && FUSE_TEMPORAL && !SHADOWED && !DEFER && !PIPE), "primary rays in the product's fused kernel only");
    RT_WAVE_CLOCK(P);
    constexpr bool LATE = WS && FUSE_TEMPORAL && !SHADOWED && !DEFER; /* the visibility-reuse ray after the temporal merge */
    /* shadowed target: every lane stays through the RIS loop, so that the wavefront can fetch its light records together */
    constexpr bool STAY = SHADOWED && !DEFER && !PIPE && RT_SHADOWED_RIS_COOP && RT_RIS_COOP && RT_LIGHT_STRIDE == 4;
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[DEFER ? 4 : ((WS || (SHADOWED && RT_BATCH_WS)) ? WIDE_LDS_ROWS : WIDE_LDS_STACK) * TRACE_BLOCK];
    int x = 0, row = P.row0;
    const bool in_image = tile_pixel<TRACE_BLOCK>(P, x, row);
    if (!DEFER && !LATE && !STAY && !in_image) return;
    const int yi = P.H - 1 - row;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
    bool need_ray = false; /* DEFER: this lane's candidate survived and needs its visibility walked */
    bool late_live = false;  /* LATE: this lane walks a visibility-reuse ray of its own */
    f3 late_sp = F3(0.0f, 0.0f, 0.0f), late_sn = F3(0.0f, 1.0f, 0.0f);
    Res r = res_zero();
    float4 G0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), G1 = G0;
    if constexpr (RAYCAST)
    {
        if (in_image)
        {
            if (P.stats) count_walk_flags(P.stats + 4 * WALK_RAYCAST, true, true, false, false);
            Hit h;
            h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.prim = -1;
            trace_wide<false, false, TRACE_BLOCK>(S.wide, s_stack, P.rg_origin, primary_direction(P, x, yi), 0.0f, kFltMax, h, nullptr, RT_BARY_TV(S));
            vis_w[li] = make_float4(h.u, h.v, as_float(h.prim), as_float(0));
            gbuffer_make(S, P, h.u, h.v, h.prim, G0, G1);
            g0_w[li] = G0;
            g1_w[li] = G1;
        }
    }
    else if (in_image) { G0 = g0[li]; G1 = g1[li]; }
    const uint32_t flags = as_uint(G1.w);
    if (in_image && !(flags & GB_SHADED))
    {
        if (!LATE) res_store(out_rec, out_rad, li, r, false); /* Reservoir{} (:56-70); LATE: with the wavefront's other records below */
        if (!DEFER && !LATE && !STAY) return;
    }
    /* DEFER keeps every lane to the end (the queue append is a wave-level operation); so does LATE (every lane of the
     * wavefront joins the walk, with or without a ray of its own) */
    const bool act = in_image && (flags & GB_SHADED);
    const f3 sp = F3(G0.x, G0.y, G0.z), sn = F3(G1.x, G1.y, G1.z);
    PCG rng = pcg_init(hashPCG4((uint32_t)x, (uint32_t)yi, (uint32_t)P.frame, 0u), 0);
    const float fL = (float)(size_t)P.n_lights;
    int sel = -1;
    float sel_bx = 0.0f, sel_by = 0.0f; /* warped barycentrics of the selected candidate */
    /* RIS loop of the work-sharing kernel (every lane stays to the end): the wavefront fetches the 64 light records of a round
     * together, four lanes per 64-B record, through the walk's idle LDS (wave_gather_records_at, as the spatial pass does:
     * a per-lane gather of a random 64-B record is 4 wave-instructions x 64 cache lines, and the 32 candidates of a pixel
     * are 80 % of this kernel's 10 200 vector-L1 accesses per wavefront, at one access per cycle and CU). Same draws in the
     * same order, same arithmetic; lanes without a shaded pixel name light 0 and ignore it. */
    constexpr bool RIS_COOP = (LATE && !PIPE && RT_RIS_COOP && RT_LIGHT_STRIDE == 4) || STAY;
    if constexpr (RIS_COOP)
    {
        static_assert(sizeof(s_stack) >= 4096, "the record image needs 64 x 64 B");
        float4* s_img = reinterpret_cast<float4*>(s_stack);
        const int lane = (int)(threadIdx.x & 63);
#if RT_RIS_PIPE
        /* the four draws of a candidate do not depend on the reservoir: candidate i + 1 is drawn and its light record
         * requested before candidate i's arithmetic, which then runs with the fetch in flight (the record of candidate i
         * is in registers by then: one 4-KB image per wavefront is enough). Same draws in the same order. */
        float bx_n = 0.0f, by_n = 0.0f, u_n = 0.0f;
        uint32_t nth_n = 0u;
        auto draw_and_request = [&]() {
            if (act)
            {
                const float rv0 = rng.uniformf_chained();
                bx_n = rng.uniformf_chained();
                by_n = rng.uniformf_chained();
                u_n = rng.uniformf_chained();
                nth_n = (uint32_t)(rv0 * fL);
                if (nth_n == (uint32_t)P.n_lights) nth_n = (uint32_t)P.n_lights - 1u;
            }
            wave_gather_request_at(S.lights, nth_n, s_img, lane);
        };
        if (P.ris_sample_count > 0) draw_and_request();
        for (int i = 0; i < P.ris_sample_count; ++i)
        {
            float4 L0, L1, L2, L3n;
            wave_gather_finish(s_img, lane, L0, L1, L2, L3n);
            float bx = bx_n, by = by_n;
            const float u = u_n;
            const uint32_t nth = nth_n;
            if (i + 1 < P.ris_sample_count) draw_and_request();
            if (act)
            {
                const f3 v0 = F3(L0.x, L0.y, L0.z), v1 = F3(L0.w, L1.x, L1.y), v2 = F3(L1.z, L1.w, L2.x);
                warp_unit_triangle(bx, by);
                const f3 lp = (1.0f - bx - by) * v0 + bx * v1 + by * v2;
                const f3 ln = F3(L3n.x, L3n.y, L3n.z);
                const float weight = ris_weight(sp, sn, lp, ln, L2.y, L2.z, L3n.w); /* p_hat (unshadowed always, :104) / pdf = 1/L * 1/area (:98-99) */
                r.w_sum += weight;
                r.M += 1;
                if (reservoir_accept(u, weight, r.w_sum))
                {
                    sel = (int)nth; sel_bx = bx; sel_by = by;
                }
            }
        }
#else
        for (int i = 0; i < P.ris_sample_count; ++i)
        {
            float bx = 0.0f, by = 0.0f;
            uint32_t nth = 0u;
            if (act)
            {
                const float rv0 = rng.uniformf();
                bx = rng.uniformf();
                by = rng.uniformf();
                nth = (uint32_t)(rv0 * fL);
                if (nth == (uint32_t)P.n_lights) nth = (uint32_t)P.n_lights - 1u;
            }
            float4 L0, L1, L2, L3n;
            wave_gather_records_at(S.lights, nth, s_img, lane, L0, L1, L2, L3n);
            if (act)
            {
                const f3 v0 = F3(L0.x, L0.y, L0.z), v1 = F3(L0.w, L1.x, L1.y), v2 = F3(L1.z, L1.w, L2.x);
                warp_unit_triangle(bx, by);
                const f3 lp = (1.0f - bx - by) * v0 + bx * v1 + by * v2;
                const f3 ln = F3(L3n.x, L3n.y, L3n.z);
                const float p_hat = target_unshadowed(sp, sn, lp, ln, L2.y); /* unshadowed always (:104) */
                const float weight = div_pdf(p_hat, L2.z, L3n.w);           /* 1/L * 1/area (:98-99) */
                const float u = rng.uniformf();
                r.w_sum += weight;
                r.M += 1;
                if (reservoir_accept(u, weight, r.w_sum))
                {
                    sel = (int)nth; sel_bx = bx; sel_by = by;
                }
            }
        }
#endif
    }
    if (act)
    {
    if (RIS_COOP)
    {
    }
    else if (PIPE)
    {
        /* Software-pipelined form of the loop below (same draws in the same order, same arithmetic): the light record
         * of candidate i+1 is requested before the arithmetic of candidate i, so the L2 gather (four 16-B loads of a
         * random 64-B record) travels behind ~190 vector instructions instead of in front of them. */
        const int n = P.ris_sample_count;
        uint32_t nth = 0u;
        float4 C0 = make_float4(0, 0, 0, 0), C1 = C0, C2 = C0, C3 = C0;
        if (n > 0)
        {
            const float rv0 = rng.uniformf();
            nth = (uint32_t)(rv0 * fL);
            if (nth == (uint32_t)P.n_lights) nth = (uint32_t)P.n_lights - 1u;
            const float4* L = S.lights + RT_LIGHT_STRIDE * (size_t)nth;
            C0 = L[0]; C1 = L[1]; C2 = L[2]; C3 = L[3];
        }
        for (int i = 0; i < n; ++i)
        {
            float bx = rng.uniformf();
            float by = rng.uniformf();
            const float u = rng.uniformf();
            uint32_t nth_n = 0u;
            float4 N0 = C0, N1 = C1, N2 = C2, N3 = C3;
            if (i + 1 < n)
            {
                const float rv0n = rng.uniformf();
                nth_n = (uint32_t)(rv0n * fL);
                if (nth_n == (uint32_t)P.n_lights) nth_n = (uint32_t)P.n_lights - 1u;
                const float4* Ln = S.lights + RT_LIGHT_STRIDE * (size_t)nth_n;
                N0 = Ln[0]; N1 = Ln[1]; N2 = Ln[2]; N3 = Ln[3];
            }
            const f3 v0 = F3(C0.x, C0.y, C0.z), v1 = F3(C0.w, C1.x, C1.y), v2 = F3(C1.z, C1.w, C2.x);
            warp_unit_triangle(bx, by);
            const f3 lp = (1.0f - bx - by) * v0 + bx * v1 + by * v2;
            const f3 ln = F3(C3.x, C3.y, C3.z);
            const float p_hat = target_unshadowed(sp, sn, lp, ln, C2.y);
            const float weight = div_pdf(p_hat, C2.z, C3.w);
            r.w_sum += weight;
            r.M += 1;
            if (reservoir_accept(u, weight, r.w_sum))
            {
                sel = (int)nth; sel_bx = bx; sel_by = by;
            }
            nth = nth_n; C0 = N0; C1 = N1; C2 = N2; C3 = N3;
        }
    }
    else
    for (int i = 0; i < P.ris_sample_count; ++i)
    {
        /* draw order rv0, rv1, rv2, u: left-to-right argument evaluation (hipcc) */
        const float rv0 = rng.uniformf();
        float bx = rng.uniformf();
        float by = rng.uniformf();
        /* common/core.hpp:261-285 */
        uint32_t nth = (uint32_t)(rv0 * fL);
        if (nth == (uint32_t)P.n_lights) nth = (uint32_t)P.n_lights - 1u;
        /* 64-B light record: vertices + luminance(Ke) + pdf + the geometric normal (the reference's
         * exact expression, common/core.hpp:50-55, evaluated once at scene set: the loop is bound by
         * vector-ALU issue and a normalize costs 3 IEEE divisions + a square root); Ke is fetched
         * once at the end. */
        const float4* L = S.lights + RT_LIGHT_STRIDE * (size_t)nth;
        const float4 L0 = L[0], L1 = L[1], L2 = L[2];
        const f3 v0 = F3(L0.x, L0.y, L0.z), v1 = F3(L0.w, L1.x, L1.y), v2 = F3(L1.z, L1.w, L2.x);
        warp_unit_triangle(bx, by);
        const f3 lp = (1.0f - bx - by) * v0 + bx * v1 + by * v2;
        const float4 L3n = L[3];
        const f3 ln = F3(L3n.x, L3n.y, L3n.z);
        const float lum = L2.y;
        const float light_pdf = L2.z; /* 1/L * 1/area (:98-99) */
        const float p_hat = target_unshadowed(sp, sn, lp, ln, lum); /* unshadowed always (:104) */
        const float weight = div_pdf(p_hat, light_pdf, L3n.w);
        const float u = rng.uniformf();
        /* common/reservoir.hpp:22-29 */
        r.w_sum += weight;
        r.M += 1;
        /* only what identifies the winner is carried through the loop (3 registers instead of the
         * 8 of position, normal, luminance): its sample is rebuilt once below, bit for bit */
        if (reservoir_accept(u, weight, r.w_sum))
        {
            sel = (int)nth; sel_bx = bx; sel_by = by;
        }
    }
    if (sel >= 0)
    {
        const float4* L = S.lights + RT_LIGHT_STRIDE * (size_t)sel;
        const float4 L0 = L[0], L1 = L[1], L2 = L[2], L3n = L[3];
        const f3 v0 = F3(L0.x, L0.y, L0.z), v1 = F3(L0.w, L1.x, L1.y), v2 = F3(L1.z, L1.w, L2.x);
        r.hit_p = (1.0f - sel_bx - sel_by) * v0 + sel_bx * v1 + sel_by * v2;
        r.hit_n = F3(L3n.x, L3n.y, L3n.z);
        r.lum = L2.y;
        r.org_p = sp; r.org_n = sn; r.vis = false;
        const float4 ke = S.light_ke[sel];
        r.rad = F3(ke.x, ke.y, ke.z);
    }
    Res pr = res_zero();
    auto load_prev = [&]() {
        bool dummy;
        pr = res_load(prev_rec, li, dummy);
        const float4 pq = prev_rad[li];
        pr.rad = F3(pq.x, pq.y, pq.z);
    };
    if (FUSE_TEMPORAL && SHADOWED) load_prev(); /* its sample is a ray target */
    float V_cur = 1.0f, V_prev = 1.0f;
    if (SHADOWED) temporal_rays<TRACE_BLOCK>(S, s_stack, P, sp, sn, r, pr, FUSE_TEMPORAL, V_cur, V_prev, as_int(G0.w));
    else if (P.vis_reuse && !DEFER && !LATE) V_cur = check_visibility_wide<TRACE_BLOCK, WS>(S.wide, s_stack, sp, sn, r.hit_p, true, S.bvh.tv, as_int(G0.w)) ? 1.0f : 0.0f;
    {
        const float p_hat = SHADOWED ? target_shadowed(sp, sn, r.hit_p, r.hit_n, r.lum, V_cur)
                                     : target_unshadowed(sp, sn, r.hit_p, r.hit_n, r.lum);
        r.ucw = p_hat > 0.0f ? r.w_sum / ((float)r.M * p_hat) : 0.0f;
    }
    if (P.vis_reuse && !DEFER && !LATE) r.vis = V_cur != 0.0f; /* :127-131 */
    /* the ray surface point -> candidate has been walked from this pixel (shadowed target: always; else the visibility-reuse ray) */
    if (SHADOWED || (P.vis_reuse && !DEFER && !LATE)) r.ownv = ownv_of(P.ownv_tag, V_cur != 0.0f);

    if (FUSE_TEMPORAL)
    {
        if (!SHADOWED) load_prev(); /* after the walk: not live across it */
        const bool took_prev = temporal_merge<SHADOWED>(P, x, yi, sp, sn, r, pr, V_cur, V_prev);
        /* the previous frame's sample won: its ray from THIS surface point is known under the shadowed target only */
        if (took_prev) r.ownv = SHADOWED ? ownv_of(P.ownv_tag, V_prev != 0.0f) : 0u;
        /* unshadowed: neither the merge decision nor ucw depends on the candidate's visibility; the bit is stored
         * clear here and set by k_candidate_visibility if the ray finds the light unoccluded */
        if (DEFER) need_ray = P.vis_reuse && !took_prev;
        if (LATE) { late_live = P.vis_reuse && !took_prev; late_sp = sp; late_sn = sn; }
    }
    if (!LATE) res_store(out_rec, out_rad, li, r, true);
    }
    if constexpr (LATE)
    {
        /* Work-sharing kernel: the merge first, the visibility-reuse ray of :127-131 after it, at ONE call site for the
         * whole wavefront. The ray's answer is observable only if the candidate survived (otherwise the stored bit is
         * the previous sample's, reservoir.hpp:36); lanes whose candidate did not survive, sky / emissive pixels and
         * lanes outside the image walk no ray of their own and take over parts of the others' walks instead. */
        if (P.stats)
        {
            /* the reference walks the visibility-reuse ray of every shaded pixel (:127-131); here: only where the answer is
             * observable (the candidate survived the merge), less what the own triangle settles */
            const bool ref_ray = act && P.vis_reuse;
            const bool self = self_occluded(S.bvh.tv, as_int(G0.w), late_sp + 0.001f * late_sn, r.hit_p - late_sp, late_sn, late_live);
            count_walk_flags(P.stats + 4 * WALK_GENERATE, ref_ray, late_live && !self, late_live && self, ref_ray && !late_live);
        }
        const bool visible = check_visibility_wide<TRACE_BLOCK, true>(S.wide, s_stack, late_sp, late_sn, r.hit_p, late_live, S.bvh.tv, as_int(G0.w));
        if (late_live) { r.vis = visible; r.ownv = ownv_of(P.ownv_tag, visible); }
        /* the wavefront's 64 records leave together, a quad per record (wave_scatter_records below; the walk's LDS is idle now) */
        const bool shaded = in_image && (flags & GB_SHADED);
        const uint32_t mbits = ((uint32_t)r.M & RES_M_MASK) | (r.vis ? RES_VIS_BIT : 0u) | (shaded ? RES_SHADED_BIT : 0u);
        static_assert(sizeof(s_stack) >= 4096, "the record image needs 64 x 64 B");
        wave_scatter_records(out_rec, in_image ? (int)li : -1, reinterpret_cast<float4*>(s_stack), (int)(threadIdx.x & 63),
                             make_float4(r.hit_p.x, r.hit_p.y, r.hit_p.z, r.ucw), make_float4(r.hit_n.x, r.hit_n.y, r.hit_n.z, as_float(mbits)),
                             make_float4(r.org_p.x, r.org_p.y, r.org_p.z, r.lum), make_float4(r.org_n.x, r.org_n.y, r.org_n.z, r.w_sum));
        if (in_image) store_stream<2>(out_rad + li, make_float4(r.rad.x, r.rad.y, r.rad.z, as_float(r.ownv)));
    }
    if (DEFER)
    {
        const unsigned long long m = __ballot(need_ray);
        if (m)
        {
            const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
            unsigned int base = 0;
            if (lane == leader) base = atomicAdd(vis_count, (unsigned int)__popcll(m));
            base = __shfl(base, leader);
            if (need_ray) vis_queue[base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)li;
        }
    }
}

#ifdef RT_EXPERIMENTS /* rt_tuning key 11: A/B form, librestir_rt_exp.so only */
/* the deferred visibility-reuse rays of k_generate_candidate<.., DEFER>: a compact list of pixels, walked by full
 * wavefronts. Persistent-style grid: a lane takes queue entries blockIdx*64+lane, +gridDim*64, ... (the host sizes
 * the grid from the previous frame's count, so that is normally one entry per lane). Sets the visibility bit of the
 * pixel's record when the light sample is unoccluded (check_visibility, common/raytrace.hpp:45-52). */
__global__ __launch_bounds__(TRACE_BLOCK, RT_RESOLVE_WAVES_FWD) void k_candidate_visibility(
    SceneView S, const float4* __restrict__ g0, const float4* __restrict__ g1, float4* __restrict__ rec,
    const uint32_t* __restrict__ vis_queue, const unsigned int* __restrict__ vis_count)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[WIDE_LDS_STACK * TRACE_BLOCK];
    const unsigned int n = *vis_count;
    for (unsigned int i = blockIdx.x * TRACE_BLOCK + threadIdx.x; ; i += gridDim.x * TRACE_BLOCK)
    {
        const bool have = i < n;
        if (__ballot(have) == 0ull) break;
        if (have)
        {
            const size_t li = vis_queue[i];
            const float4 G0 = g0[li], G1 = g1[li];
            const float4 q0 = rec[4 * li + 0];
            const bool visible = check_visibility_wide<TRACE_BLOCK>(S.wide, s_stack, F3(G0.x, G0.y, G0.z), F3(G1.x, G1.y, G1.z), F3(q0.x, q0.y, q0.z), true, S.bvh.tv, as_int(G0.w));
            if (visible)
            {
                uint32_t* w = reinterpret_cast<uint32_t*>(rec) + 16 * li + 7; /* q1.w = M | vis << 31 | shaded << 30 */
                *w = *w | RES_VIS_BIT;
            }
        }
    }
}

#endif /* RT_EXPERIMENTS */
/* -------------------------------------------------------- temporal_resampling */
/* examples/10_restir_di/10_restir_di.cu:137-237 (stand-alone entry point) */
template <bool SHADOWED>
__global__ __launch_bounds__(BLOCK) void k_temporal(SceneView S, FrameParams P, const float4* __restrict__ g0,
                                                     const float4* __restrict__ g1,
                                                     const float4* __restrict__ prev_rec,
                                                     const float4* __restrict__ prev_rad,
                                                     float4* __restrict__ rec, float4* __restrict__ radb)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[WIDE_LDS_WORDS];
    int x, row;
    if (!tile_pixel(P, x, row)) return;
    const int yi = P.H - 1 - row;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
    const float4 G0 = g0[li], G1 = g1[li];
    if (!(as_uint(G1.w) & GB_SHADED)) return;
    if (!P.use_temporal) return;
    const f3 sp = F3(G0.x, G0.y, G0.z), sn = F3(G1.x, G1.y, G1.z);
    bool dummy;
    Res r = res_load(rec, li, dummy);
    const float4 rq = radb[li];
    r.rad = F3(rq.x, rq.y, rq.z);
    r.ownv = as_uint(rq.w);
    Res pr = res_load(prev_rec, li, dummy);
    const float4 pq = prev_rad[li];
    pr.rad = F3(pq.x, pq.y, pq.z);
    float V_cur = 1.0f, V_prev = 1.0f;
    if (SHADOWED) temporal_rays(S, s_stack, P, sp, sn, r, pr, true, V_cur, V_prev, as_int(G0.w));
    const bool took_prev = temporal_merge<SHADOWED>(P, x, yi, sp, sn, r, pr, V_cur, V_prev);
    if (SHADOWED) r.ownv = ownv_of(P.ownv_tag, (took_prev ? V_prev : V_cur) != 0.0f);
    else if (took_prev) r.ownv = 0u;
    res_store(rec, radb, li, r, true);
}

/* --------------------------------------------------------- spatial_resampling */
/* examples/10_restir_di/10_restir_di.cu:256-388 — the roofline kernel.
 * Per neighbour ONE 64-B aligned record is gathered (the reference gathers a 16-B Visibility,
 * a Triangle and a 76-B Reservoir); the "sky / emissive neighbour" test of :326-338 reads the
 * shaded bit kept inside the record; radiance (side record) is fetched once, for the sample
 * that survived. */
#ifndef RT_SHSPATIAL_WAVES
#define RT_SHSPATIAL_WAVES 7 /* register budget in wavefronts per SIMD. r02: 4 (<= 128 VGPRs; 5 and 6 spilled badly). r04, without the SLP
                                * vectoriser the kernel needs 112: shadowed frame 5.34 (4) / 4.93 (5) / 4.72 (6) / 4.66 (7) / 4.90 ms (8) */
#endif
/* one pixel of the pass (x, row already resolved by the caller's tile mapping) */
template <bool SHADOWED, int TB>
RT_DEV void spatial_pixel(const SceneView& S, const FrameParams& P, const HaloFuse& F, uint32_t* s_stack, int x, int row, const float4* __restrict__ g0,
                          const float4* __restrict__ g1, const float4* __restrict__ in_rec, const float4* __restrict__ in_rad,
                          float4* __restrict__ out_rec, float4* __restrict__ out_rad)
{
    const int yi = P.H - 1 - row;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
    const float4 G0 = g0[li], G1 = g1[li];
    if (!(as_uint(G1.w) & GB_SHADED))
    {
        /* the reference stores nothing here (:275-287); we keep the shaded bit valid */
        res_store_give(F, P.W, out_rec, out_rad, li, x, row, res_zero(), false);
        return;
    }
    const f3 sp = F3(G0.x, G0.y, G0.z), sn = F3(G1.x, G1.y, G1.z);
    PCG rng = pcg_init(hashPCG4((uint32_t)x, (uint32_t)yi, (uint32_t)P.frame, (uint32_t)(2 + P.pass)), 0);

    bool own_shaded;
    Res r = res_load(in_rec, li, own_shaded);
    const float4* rad_from = in_rad + li; /* radiance side record of the selected sample: a buffer entry or a halo list entry */
    bool took_other = false;

    if (SHADOWED && P.use_spatial && P.spatial_count <= 5)
    {
        /* Shadowed target function (:346-350, :370-374): the reference traces one shadow ray per accepted
         * neighbour (surface point -> the neighbour's light sample) and one for the finally selected
         * sample. None of those rays depends on the merge chain, and the final one repeats the ray of
         * whichever sample won, so: (1) draw all random numbers and pick the neighbours, (2) walk the
         * <= 6 distinct rays of the pixel back to back (occluded_batch), (3) run the merge chain with
         * the visibilities. A neighbour's ray is not traced when its weight is 0 whatever the answer
         * (ucw == 0, or visibility reuse with an occluded sample). Same results, same reference ray
         * count (rt_ray_count counts raytrace() calls of the reference, not walks). */
        const float scale = P.spatial_radius / 1.96f;
        const float4* prec[5]; /* the neighbour's record (reservoir buffer or received halo list), nullptr = none */
        const float4* prad[5];
        float ud[5];
        f3 tgt[6];
        uint32_t need = 0u;
#pragma unroll
        for (int k = 0; k < 5; ++k)
        {
            prec[k] = nullptr; prad[k] = nullptr; ud[k] = 0.0f; tgt[k] = F3(0.0f, 0.0f, 0.0f);
            if (k < P.spatial_count)
            {
                const float rv0 = rng.uniformf();
                const float rv1 = rng.uniformf();
                const float radius = sqrt_guarded(fmax_dev(-2.0f * pm_logf(rv0), 0.0f));
                const float phi = 2.0f * kPI * rv1;
                float sn_phi, cs_phi;
                pm_sincosf(phi, &sn_phi, &cs_phi);
                const float gx = radius * cs_phi, gy = radius * sn_phi;
                const int nx = f2i_sat((float)x + scale * gx);
                const int ny = f2i_sat((float)yi + scale * gy);
                const int lr = P.H - 1 - ny - P.lrow0;
                const bool ok = !(nx < 0 || nx >= P.W || ny < 0 || ny >= P.H) && !(nx == x && ny == yi) && !(lr < 0 || lr >= P.lrows);
                if (ok)
                {
                    const size_t id = (size_t)nx + (size_t)lr * P.W;
                    const float4* nrad;
                    const float4* nq = halo_record(F, P.W, in_rec, in_rad, id, nx, P.H - 1 - ny, nrad);
                    const float4 q0 = nq[0], q1 = nq[1];
                    const uint32_t mb = as_uint(q1.w);
                    if (mb & RES_SHADED_BIT)
                    {
                        prec[k] = nq; prad[k] = nrad;
                        ud[k] = rng.uniformf();
                        tgt[k] = F3(q0.x, q0.y, q0.z);
                        const bool moot = (q0.w == 0.0f) || (P.vis_reuse && !(mb & RES_VIS_BIT));
                        if (!moot) need |= 1u << k;
                    }
                }
            }
        }
        tgt[5] = r.hit_p;
        /* the own sample's ray: walked by the kernel that wrote this record if the flag says so (see rt_device.h) */
        const uint32_t own_flags = ownv_trusted(P.ownv_tag, as_uint(in_rad[li].w));
        if (!(own_flags & OWNV_KNOWN)) need |= 1u << 5;
        uint32_t occl = occluded_batch<6, TB>(S.wide, s_stack, sp, sn, tgt, need, S.bvh.tv, as_int(G0.w));
        if ((own_flags & OWNV_KNOWN) && !(own_flags & OWNV_VISIBLE)) occl |= 1u << 5;
        int sel = 5;
#pragma unroll
        for (int k = 0; k < 5; ++k)
        {
            if (prec[k])
            {
                bool n_shaded;
                Res nr = res_load_at(prec[k], n_shaded);
                /* evaluate_target_function with the shadow term (common/reservoir.hpp:42-59) */
                const float V = (occl >> k) & 1u ? 0.0f : 1.0f;
                float p_hat_y = (1.0f / kPI) * geometry_term(sp, sn, nr.hit_p, nr.hit_n) * V * nr.lum;
                if (P.vis_reuse) p_hat_y *= nr.vis ? 1.0f : 0.0f;
                nr.M = scale_M(nr.M, rejection_heuristics(r.org_p, r.org_n, nr.org_p, nr.org_n, P.eye));
                const float weight = p_hat_y * nr.ucw * (float)nr.M;
                r.w_sum += weight;
                r.M += nr.M;
                if (reservoir_accept(ud[k], weight, r.w_sum))
                {
                    res_take_sample(r, nr);
                    rad_from = prad[k];
                    sel = k;
                }
            }
        }
        const float Vf = (occl >> sel) & 1u ? 0.0f : 1.0f;
        const float p_hat = (1.0f / kPI) * geometry_term(sp, sn, r.hit_p, r.hit_n) * Vf * r.lum;
        r.ucw = p_hat > 0.0f ? r.w_sum / ((float)r.M * p_hat) : 0.0f;
        r.ownv = ownv_of(P.ownv_tag, Vf != 0.0f);
        took_other = true; /* flags set above */
    }
    else if (P.use_spatial)
    {
        const float scale = P.spatial_radius / 1.96f;
        for (int k = 0; k < P.spatial_count; ++k)
        {
            const float rv0 = rng.uniformf();
            const float rv1 = rng.uniformf();
            /* common/reservoir.hpp:89-95 with portable log/cos/sin */
            const float radius = sqrt_guarded(fmax_dev(-2.0f * pm_logf(rv0), 0.0f));
            const float phi = 2.0f * kPI * rv1;
            float sn_phi, cs_phi;
            pm_sincosf(phi, &sn_phi, &cs_phi);
            const float gx = radius * cs_phi, gy = radius * sn_phi;
            const int nx = f2i_sat((float)x + scale * gx);
            const int ny = f2i_sat((float)yi + scale * gy);
            if (nx < 0 || nx >= P.W || ny < 0 || ny >= P.H) continue;
            if (nx == x && ny == yi) continue;
            const int nrow = P.H - 1 - ny;
            const int lr = nrow - P.lrow0;
            if (lr < 0 || lr >= P.lrows) continue; /* only when halo < 87: outside the contract */
            const size_t pid = (size_t)nx + (size_t)lr * P.W;
            bool n_shaded;
            const float4* nrad;
            Res nr = res_load_at(halo_record(F, P.W, in_rec, in_rad, pid, nx, nrow, nrad), n_shaded);
            if (!n_shaded) continue; /* sky or emissive neighbour (:326-338) */

            float p_hat_y = target_function<SHADOWED, TB>(S, s_stack, sp, sn, nr.hit_p, nr.hit_n, nr.lum, as_int(G0.w));
            if (P.vis_reuse) p_hat_y *= nr.vis ? 1.0f : 0.0f;
            nr.M = scale_M(nr.M, rejection_heuristics(r.org_p, r.org_n, nr.org_p, nr.org_n, P.eye));
            const float weight = p_hat_y * nr.ucw * (float)nr.M;
            const float u = rng.uniformf();
            r.w_sum += weight;
            r.M += nr.M;
            if (reservoir_accept(u, weight, r.w_sum))
            {
                res_take_sample(r, nr);
                rad_from = nrad;
                took_other = true;
            }
        }
        const float p_hat = target_function<SHADOWED, TB>(S, s_stack, sp, sn, r.hit_p, r.hit_n, r.lum, as_int(G0.w));
        r.ucw = p_hat > 0.0f ? r.w_sum / ((float)r.M * p_hat) : 0.0f;
    }
    const float4 rq = *rad_from;
    r.rad = F3(rq.x, rq.y, rq.z);
    /* unshadowed passes and the one-ray-at-a-time shadowed form: the own sample keeps what is known about it, a sample
     * taken from a neighbour has not been tested from here */
    if (!took_other) r.ownv = as_uint(rq.w);
    else if (!(SHADOWED && P.use_spatial && P.spatial_count <= 5)) r.ownv = 0u;
    res_store_give(F, P.W, out_rec, out_rad, li, x, row, r, true);
}

/* shadowed target function: a tracing kernel (one-wavefront workgroups, LDS traversal stack) */
/* The shadowed pass with the wavefront's record traffic in the cooperative form of k_spatial_coop (four lanes per 64-B record
 * through the walk's LDS, which is idle outside the batch walk): the five neighbour records are fetched once for the ray
 * targets and once more for the merge, the 64 results leave a quad per record. Every lane stays to the end (a lane without a
 * shaded pixel names its own record, holds no ray and helps the others' walks). Same draws, same rays, same arithmetic as
 * spatial_pixel<true> (the <= 5 neighbour form). rt_tuning key 8 = 0 keeps the per-lane form. */
template <int TB>
RT_DEV void spatial_wave_shadowed(const SceneView& S, const FrameParams& P, const HaloFuse& F, uint32_t* s_stack, const float4* __restrict__ g0,
                                  const float4* __restrict__ g1, const float4* __restrict__ in_rec, const float4* __restrict__ in_rad,
                                  float4* __restrict__ out_rec, float4* __restrict__ out_rad)
{
    float4* s_img = reinterpret_cast<float4*>(s_stack);
    const int lane = (int)(threadIdx.x & 63);
    int x = 0, row = P.lrow0;
    const bool in_image = tile_pixel<TB>(P, x, row);
    const int yi = P.H - 1 - row;
    const size_t li = in_image ? (size_t)x + (size_t)(row - P.lrow0) * P.W : 0;
    float4 G0 = make_float4(0.0f, 0.0f, 0.0f, as_float(-1)), G1 = make_float4(0.0f, 1.0f, 0.0f, 0.0f);
    if (in_image) { G0 = g0[li]; G1 = g1[li]; }
    const bool active = in_image && (as_uint(G1.w) & GB_SHADED);
    const f3 sp = F3(G0.x, G0.y, G0.z), sn = F3(G1.x, G1.y, G1.z);
    PCG rng = pcg_init(hashPCG4((uint32_t)x, (uint32_t)yi, (uint32_t)P.frame, (uint32_t)(2 + P.pass)), 0);
    float4 q0, q1, q2, q3;
    wave_gather_records_at(in_rec, (uint32_t)li, s_img, lane, q0, q1, q2, q3);
    bool own_shaded;
    Res r = res_from_parts(q0, q1, q2, q3, own_shaded);
    const float4* rad_from = in_rad + li;

    const float scale = P.spatial_radius / 1.96f;
    uint32_t pcode[5]; /* where the neighbour's record is (halo_code: reservoir buffer or received halo list), HALO_CODE_NONE = none */
    float ud[5];
    f3 tgt[6];
    uint32_t need = 0u;
#pragma unroll
    for (int k = 0; k < 5; ++k)
    {
        pcode[k] = HALO_CODE_NONE; ud[k] = 0.0f; tgt[k] = F3(0.0f, 0.0f, 0.0f);
        if (k < P.spatial_count) /* wave-uniform */
        {
            uint32_t ncode = 4u * (uint32_t)li;
            bool have = false;
            if (active)
            {
                const float rv0 = rng.uniformf();
                const float rv1 = rng.uniformf();
                const float radius = sqrt_guarded(fmax_dev(-2.0f * pm_logf(rv0), 0.0f));
                const float phi = 2.0f * kPI * rv1;
                float sn_phi, cs_phi;
                pm_sincosf(phi, &sn_phi, &cs_phi);
                const float gx = radius * cs_phi, gy = radius * sn_phi;
                const int nx = f2i_sat((float)x + scale * gx);
                const int ny = f2i_sat((float)yi + scale * gy);
                const int lr = P.H - 1 - ny - P.lrow0;
                have = !(nx < 0 || nx >= P.W || ny < 0 || ny >= P.H) && !(nx == x && ny == yi) && !(lr < 0 || lr >= P.lrows);
                if (have) ncode = halo_code(F, P.W, (size_t)nx + (size_t)lr * P.W, nx, P.H - 1 - ny);
            }
            wave_gather_records(halo_code_record(F, in_rec, ncode), s_img, lane, q0, q1, q2, q3);
            const uint32_t mb = as_uint(q1.w);
            if (have && (mb & RES_SHADED_BIT))
            {
                pcode[k] = ncode;
                ud[k] = rng.uniformf();
                tgt[k] = F3(q0.x, q0.y, q0.z);
                const bool moot = (q0.w == 0.0f) || (P.vis_reuse && !(mb & RES_VIS_BIT));
                if (!moot) need |= 1u << k;
            }
        }
    }
    tgt[5] = r.hit_p;
    /* the own sample's ray: walked by the kernel that wrote this record if the flag says so (see rt_device.h) */
    uint32_t own_flags = 0u;
    if (active)
    {
        own_flags = ownv_trusted(P.ownv_tag, as_uint(in_rad[li].w));
        if (!(own_flags & OWNV_KNOWN)) need |= 1u << 5;
    }
    if (P.stats)
    {
        /* reference: one ray per neighbour that reaches the merge + the final sample's (:346-350, :370-378) */
        uint32_t n_ref = active ? 1u : 0u;
#pragma unroll
        for (int k = 0; k < 5; ++k) n_ref += pcode[k] != HALO_CODE_NONE ? 1u : 0u;
        const uint32_t self = self_occluded_mask<6>(S.bvh.tv, active ? as_int(G0.w) : -1, sp, sn, tgt, need);
        const uint32_t n_need = (uint32_t)__popc(need), n_self = (uint32_t)__popc(self);
        count_walk_counts(P.stats + 4 * WALK_SPATIAL, n_ref, n_need - n_self, n_self, n_ref - n_need);
    }
    uint32_t occl = occluded_batch<6, TB>(S.wide, s_stack, sp, sn, tgt, need, S.bvh.tv, active ? as_int(G0.w) : -1);
    if ((own_flags & OWNV_KNOWN) && !(own_flags & OWNV_VISIBLE)) occl |= 1u << 5;
    int sel = 5;
#pragma unroll
    for (int k = 0; k < 5; ++k)
    {
        if (k < P.spatial_count)
        {
            wave_gather_records(halo_code_record(F, in_rec, pcode[k] != HALO_CODE_NONE ? pcode[k] : 4u * (uint32_t)li), s_img, lane, q0, q1, q2, q3);
            if (pcode[k] != HALO_CODE_NONE)
            {
                bool n_shaded;
                Res nr = res_from_parts(q0, q1, q2, q3, n_shaded);
                /* evaluate_target_function with the shadow term (common/reservoir.hpp:42-59) */
                const float V = (occl >> k) & 1u ? 0.0f : 1.0f;
                float p_hat_y = (1.0f / kPI) * geometry_term(sp, sn, nr.hit_p, nr.hit_n) * V * nr.lum;
                if (P.vis_reuse) p_hat_y *= nr.vis ? 1.0f : 0.0f;
                nr.M = scale_M(nr.M, rejection_heuristics(r.org_p, r.org_n, nr.org_p, nr.org_n, P.eye));
                const float weight = p_hat_y * nr.ucw * (float)nr.M;
                r.w_sum += weight;
                r.M += nr.M;
                if (reservoir_accept(ud[k], weight, r.w_sum))
                {
                    res_take_sample(r, nr);
                    rad_from = halo_code_radiance(F, in_rec, in_rad, pcode[k]);
                    sel = k;
                }
            }
        }
    }
    const float Vf = (occl >> sel) & 1u ? 0.0f : 1.0f;
    const float p_hat = (1.0f / kPI) * geometry_term(sp, sn, r.hit_p, r.hit_n) * Vf * r.lum;
    r.ucw = p_hat > 0.0f ? r.w_sum / ((float)r.M * p_hat) : 0.0f;
    if (active)
    {
        const float4 rq = *rad_from;
        r.rad = F3(rq.x, rq.y, rq.z);
        r.ownv = ownv_of(P.ownv_tag, Vf != 0.0f);
    }
    else
        r = res_zero(); /* the reference stores nothing here (:275-287); we keep the shaded bit valid */
    const uint32_t mbits = ((uint32_t)r.M & RES_M_MASK) | (r.vis ? RES_VIS_BIT : 0u) | (active ? RES_SHADED_BIT : 0u);
    wave_scatter_records<false>(out_rec, in_image ? (int)li : -1, s_img, lane, make_float4(r.hit_p.x, r.hit_p.y, r.hit_p.z, r.ucw),
                         make_float4(r.hit_n.x, r.hit_n.y, r.hit_n.z, as_float(mbits)), make_float4(r.org_p.x, r.org_p.y, r.org_p.z, r.lum),
                         make_float4(r.org_n.x, r.org_n.y, r.org_n.z, r.w_sum));
    if (!in_image) return;
    store_stream<0>(out_rad + li, make_float4(r.rad.x, r.rad.y, r.rad.z, as_float(r.ownv)));
    res_give(F, P.W, li, x, row, r, active);
}
template <bool SHADOWED, bool COOP = false>
__global__ __launch_bounds__(TRACE_BLOCK, RT_SHSPATIAL_WAVES) void k_spatial(SceneView S, FrameParams P, HaloFuse F, const float4* __restrict__ g0,
                                                    const float4* __restrict__ g1, const float4* __restrict__ in_rec,
                                                    const float4* __restrict__ in_rad, float4* __restrict__ out_rec,
                                                    float4* __restrict__ out_rad)
{
    static_assert(SHADOWED, "the unshadowed pass is k_spatial_gather / k_spatial_lds");
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[(RT_BATCH_WS ? WIDE_LDS_ROWS : WIDE_LDS_STACK) * TRACE_BLOCK];
    if constexpr (COOP)
    {
        static_assert(sizeof(s_stack) >= 4096, "the record image needs 64 x 64 B");
        spatial_wave_shadowed<TRACE_BLOCK>(S, P, F, s_stack, g0, g1, in_rec, in_rad, out_rec, out_rad);
        return;
    }
    int x, row;
    if (!tile_pixel<TRACE_BLOCK>(P, x, row)) return;
    spatial_pixel<true, TRACE_BLOCK>(S, P, F, s_stack, x, row, g0, g1, in_rec, in_rad, out_rec, out_rad);
}

/* Unshadowed target function = the roofline kernel: no rays, five dependent 64-B record gathers per pixel.
 * The register allocation can be told to admit at most WAVES wavefronts per SIMD (template parameter; rt_tuning key 9;
 * round 1 did this with a dummy 32 KB LDS allocation, and rt_tuning key 4 still adds LDS for A/B runs, default 0): the
 * xnack-any code of rounds 1-2 wanted 4-5 to keep the neighbour window in the XCD's 4 MiB L2 (0.184 against 0.202 ms per
 * pass unbounded); the xnack- code the library ships is flat across 4 / 5 / 6 / unbounded (0.170 / 0.169 / 0.168 / 0.170 ms),
 * and the default is 6 (RT_SPATIAL_GATHER_AUTO_WAVES in restir_rt.hip).
 * amdgpu_waves_per_eu alone leaves .vgpr_count at the registers the kernel uses (7 wavefronts per SIMD at dispatch).
 * Naming the last register of an allocation step as clobbered makes the descriptor ask for that step: 128 / 96 / 80 VGPRs
 * = 4 / 5 / 6 wavefronts per SIMD (MI355X_MICROARCH.md, register-file table). */
template <int WAVES> RT_DEV void occupancy_bound()
{
    /* the last register of the allocation step that admits WAVES wavefronts per SIMD: 512 / {128, 96, 80} */
    if (WAVES == 4) asm volatile("; occupancy: 128 VGPRs -> 4 wavefronts per SIMD" ::: "v127");
    if (WAVES == 5) asm volatile("; occupancy: 96 VGPRs -> 5 wavefronts per SIMD" ::: "v95");
    if (WAVES == 6) asm volatile("; occupancy: 80 VGPRs -> 6 wavefronts per SIMD" ::: "v79");
}
#ifdef RT_EXPERIMENTS /* rt_tuning key 8 = 0: the per-lane gather form of the pass (r01), A/B only */
template <int WAVES>
__global__ __launch_bounds__(BLOCK) void k_spatial_gather(
    SceneView S, FrameParams P, HaloFuse F, const float4* __restrict__ g0, const float4* __restrict__ g1, const float4* __restrict__ in_rec,
    const float4* __restrict__ in_rad, float4* __restrict__ out_rec, float4* __restrict__ out_rad)
{
    occupancy_bound<WAVES>();
    int x, row;
    if (!tile_pixel<BLOCK>(P, x, row)) return;
    spatial_pixel<false, BLOCK>(S, P, F, nullptr, x, row, g0, g1, in_rec, in_rad, out_rec, out_rad);
}

#endif /* RT_EXPERIMENTS */
/* LDS-staged variant of the same pass (north star: "LDS-staged neighbour reservoirs"; rt_tuning key 8 = 1).
 * What can be staged: a 32x8 tile's +-87-pixel neighbour window holds (32+174) x (8+174) = 37 492 records = 2.4 MB
 * against 160 KB of LDS, and each pixel consumes 5 of them, so the RECORDS cannot be staged. What serialises the
 * gathers is the data dependence between them: neighbour k+1's address needs the random draw that is consumed only
 * if neighbour k is shaded, i.e. it waits for gather k. The shaded flags of the window are 1 bit per pixel:
 * 182 rows x 7 words = 5 KB. This kernel stages that bitmap (k_shaded_bitmap writes it once per frame), derives all
 * neighbour addresses and random draws from LDS alone, and then runs the merge chain with the record of
 * neighbour k+1 already in flight while neighbour k is merged. Same decisions, same arithmetic, same results. */
constexpr int SPL_HALO = 87, SPL_ROWS = TILE_H + 2 * SPL_HALO, SPL_WORDS = 7;
__global__ void k_shaded_bitmap(int W, int rows, const float4* __restrict__ g1, uint32_t* __restrict__ bits)
{
    /* one wavefront per 64 pixels of a row: two 32-bit words by ballot */
    const int words = (W + 31) / 32;
    const int row = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const bool sh = x < W && row < rows && (as_uint(g1[(size_t)row * W + x].w) & GB_SHADED) != 0u;
    const unsigned long long m = __ballot(sh);
    const int lane = threadIdx.x & 63, w0 = (x - lane) >> 5;
    if (lane == 0 && w0 < words) bits[(size_t)row * words + w0] = (uint32_t)m;
    if (lane == 32 && w0 + 1 < words) bits[(size_t)row * words + w0 + 1] = (uint32_t)(m >> 32);
}
#ifdef RT_EXPERIMENTS /* rt_tuning key 8 = 1: LDS-staged shaded-bit window (r02), A/B only */
template <int WAVES>
__global__ __launch_bounds__(BLOCK) void k_spatial_lds(
    FrameParams P, const uint32_t* __restrict__ bits, const float4* __restrict__ g0, const float4* __restrict__ g1,
    const float4* __restrict__ in_rec, const float4* __restrict__ in_rad, float4* __restrict__ out_rec, float4* __restrict__ out_rad)
{
    occupancy_bound<WAVES>();
    __shared__ uint32_t s_bits[SPL_ROWS * SPL_WORDS];
    /* the tile of this workgroup (all its threads share it): window = rows [trow0 - 87, trow0 + 8 + 87), words [tw0, tw0 + 7) */
    int x = 0, row = P.row0;
    const bool ok = tile_pixel<BLOCK>(P, x, row);
    /* every thread of a 32x8 tile derives the same tile origin from its own (x, row), in or out of the image; the
     * tile rows of a launch start at P.row0 or, for its second row range, at P.rowb0 (strips): row - (row within the
     * tile). A workgroup beyond the last tile (x, row left at their defaults) stages a window nobody reads. */
#if RT_WAVE_8X8 && RT_TILE_W == 32
    const int in_tile_row = (threadIdx.x >> 3) & 7;
#else
    const int in_tile_row = threadIdx.x >> TILE_W_LOG2;
#endif
    const int tx0 = x & ~(TILE_W - 1), trow0 = row - in_tile_row;
    const int words = (P.W + 31) / 32, tw0 = (tx0 >> 5) - 3;
    for (int i = threadIdx.x; i < SPL_ROWS * SPL_WORDS; i += BLOCK)
    {
        const int r = i / SPL_WORDS, w = i - r * SPL_WORDS;
        const int grow = trow0 - SPL_HALO + r, gw = tw0 + w;
        const int lr = grow - P.lrow0;
        s_bits[i] = (lr >= 0 && lr < P.lrows && gw >= 0 && gw < words) ? bits[(size_t)lr * words + gw] : 0u;
    }
    __syncthreads();
    if (!ok) return;
    const int yi = P.H - 1 - row;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
    const float4 G0 = g0[li], G1 = g1[li];
    if (!(as_uint(G1.w) & GB_SHADED))
    {
        res_store(out_rec, out_rad, li, res_zero(), false);
        return;
    }
    const f3 sp = F3(G0.x, G0.y, G0.z), sn = F3(G1.x, G1.y, G1.z);
    PCG rng = pcg_init(hashPCG4((uint32_t)x, (uint32_t)yi, (uint32_t)P.frame, (uint32_t)(2 + P.pass)), 0);
    bool own_shaded;
    Res r = res_load(in_rec, li, own_shaded);
    size_t rad_from = li;
    if (P.use_spatial)
    {
        /* 1. all neighbour choices from the RNG and the staged shaded bits (10_restir_di.cu:305-340) */
        const float scale = P.spatial_radius / 1.96f;
        long long pid[5];
        float ud[5];
#pragma unroll
        for (int k = 0; k < 5; ++k)
        {
            pid[k] = -1; ud[k] = 0.0f;
            if (k < P.spatial_count)
            {
                const float rv0 = rng.uniformf();
                const float rv1 = rng.uniformf();
                const float radius = sqrt_guarded(fmax_dev(-2.0f * pm_logf(rv0), 0.0f));
                const float phi = 2.0f * kPI * rv1;
                float sn_phi, cs_phi;
                pm_sincosf(phi, &sn_phi, &cs_phi);
                const int nx = f2i_sat((float)x + scale * (radius * cs_phi));
                const int ny = f2i_sat((float)yi + scale * (radius * sn_phi));
                const int nrow = P.H - 1 - ny, lr = nrow - P.lrow0;
                const bool okn = !(nx < 0 || nx >= P.W || ny < 0 || ny >= P.H) && !(nx == x && ny == yi) && !(lr < 0 || lr >= P.lrows);
                if (okn)
                {
                    /* |offset| <= 86.43 px for the default radius: inside the staged window */
                    const int wr = nrow - (trow0 - SPL_HALO), ww = (nx >> 5) - tw0;
                    const uint32_t word = s_bits[wr * SPL_WORDS + ww];
                    if (word & (1u << (nx & 31)))
                    {
                        pid[k] = (long long)((size_t)nx + (size_t)lr * P.W);
                        ud[k] = rng.uniformf();
                    }
                }
            }
        }
        /* 2. the merge chain of :340-371; the loads of neighbour k+1 do not depend on the merge of neighbour k */
#pragma unroll
        for (int k = 0; k < 5; ++k)
        {
            if (pid[k] >= 0)
            {
                bool n_shaded;
                Res nr = res_load(in_rec, (size_t)pid[k], n_shaded);
                float p_hat_y = target_unshadowed(sp, sn, nr.hit_p, nr.hit_n, nr.lum);
                if (P.vis_reuse) p_hat_y *= nr.vis ? 1.0f : 0.0f;
                nr.M = scale_M(nr.M, rejection_heuristics(r.org_p, r.org_n, nr.org_p, nr.org_n, P.eye));
                const float weight = p_hat_y * nr.ucw * (float)nr.M;
                r.w_sum += weight;
                r.M += nr.M;
                if (reservoir_accept(ud[k], weight, r.w_sum))
                {
                    res_take_sample(r, nr);
                    rad_from = (size_t)pid[k];
                }
            }
        }
        const float p_hat = target_unshadowed(sp, sn, r.hit_p, r.hit_n, r.lum);
        r.ucw = p_hat > 0.0f ? r.w_sum / ((float)r.M * p_hat) : 0.0f;
    }
    const float4 rq = in_rad[rad_from];
    r.rad = F3(rq.x, rq.y, rq.z);
    r.ownv = rad_from == li ? as_uint(rq.w) : 0u; /* own sample: what is known stays known; a neighbour's: untested from here */
    res_store(out_rec, out_rad, li, r, true);
}


#endif /* RT_EXPERIMENTS */
/* Cooperative gather variant of the same pass (rt_tuning key 8 = 2). What bounds k_spatial_gather is not bytes but
 * address-processing slots of the CU's vector L1: a per-lane gather of a 64-B record is 4 dwordx4 wave-instructions that
 * each touch 64 different cache lines (TCP_TOTAL_CACHE_ACCESSES = 1 521 per wavefront, 0.47 per cycle and CU). Here the
 * wavefront fetches its 64 records together: in round j lane l loads one 16-B part of the record lane 16 j + l / 4 wants,
 * so the four lanes of a quad read ONE 64-B segment (16 segments per wave-instruction instead of 64), as an LDS-DMA load
 * whose lane-linear destination puts the four parts of a record next to each other; every lane then reads its own
 * record back with 4 ds_read_b128. Bank conflicts: which part a lane fetches is rotated by the record's lane (the
 * source address is free, the destination is not), and the reader applies the same rotation.
 * Same decisions, same arithmetic, same results as spatial_pixel<false>. */
#ifndef RT_COOP_DMA
#define RT_COOP_DMA 1 /* 0: stage through VGPRs (global_load_dwordx4 + ds_write_b128) */
#endif
/* the common tail: the wavefront's loads have been issued; wait, then every lane reads its own record */
RT_DEV void wave_gather_finish(float4* s_wave, const int lane, float4& q0, float4& q1, float4& q2, float4& q3)
{
#if RT_COOP_DMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    RT_WAVE_LDS_FENCE();
    const int rot = (lane >> 2) & 3;
    q0 = s_wave[4 * lane + (0 ^ rot)];
    q1 = s_wave[4 * lane + (1 ^ rot)];
    q2 = s_wave[4 * lane + (2 ^ rot)];
    q3 = s_wave[4 * lane + (3 ^ rot)];
    RT_WAVE_LDS_FENCE(); /* the next round overwrites the image */
}
RT_DEV void wave_gather_issue(const float4* p, float4* s_dst, const int lane)
{
#if RT_COOP_DMA
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)s_dst, 16, 0, 0);
#else
    s_dst[lane] = *p;
#endif
}
/* Every lane names a record (never null: a lane that wants none names any valid one, e.g. its own, and ignores what comes
 * back - no branches around the loads). By address (records in halo lists), or by index into one buffer (half the lane
 * exchanges). */
RT_DEV void wave_gather_records(const float4* q, float4* s_wave, const int lane, float4& q0, float4& q1, float4& q2, float4& q3)
{
    const unsigned long long a = (unsigned long long)q;
    const int lo = (int)(uint32_t)a, hi = (int)(uint32_t)(a >> 32);
    const float4* p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) /* all lane exchanges first: they wait on one counter */
    {
        const int src = 16 * j + (lane >> 2);
        const uint32_t slo = (uint32_t)__shfl(lo, src), shi = (uint32_t)__shfl(hi, src);
        p[j] = (const float4*)(((unsigned long long)shi << 32) | slo);
    }
    const int part = (lane & 3) ^ ((lane >> 4) & 3); /* rotation by the record's lane: (src >> 2) & 3 = (lane >> 4) & 3 for every j */
#pragma unroll
    for (int j = 0; j < 4; ++j) wave_gather_issue(p[j] + part, s_wave + 64 * j, lane);
    wave_gather_finish(s_wave, lane, q0, q1, q2, q3);
}
/* the request half of wave_gather_records_at: the wavefront's 64 records are on their way into the image when this returns
 * (wave_gather_finish collects them); the image must not be in use */
RT_DEV void wave_gather_request_at(const float4* __restrict__ rec, const uint32_t idx, float4* s_wave, const int lane)
{
    uint32_t from[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) from[j] = (uint32_t)__shfl((int)idx, 16 * j + (lane >> 2));
    const uint32_t part16 = (uint32_t)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    /* a wave-uniform base + a 32-bit byte offset per lane: one address instruction per load instead of a 64-bit shift and add
     * (the buffers hold at most 2^26 records: rt_create, rt_scene_set) */
    const char* base = reinterpret_cast<const char*>(rec);
#pragma unroll
    for (int j = 0; j < 4; ++j) wave_gather_issue(reinterpret_cast<const float4*>(base + (from[j] * 64u + part16)), s_wave + 64 * j, lane);
}
RT_DEV void wave_gather_records_at(const float4* __restrict__ rec, const uint32_t idx, float4* s_wave, const int lane, float4& q0, float4& q1,
                                   float4& q2, float4& q3)
{
    wave_gather_request_at(rec, idx, s_wave, lane);
    wave_gather_finish(s_wave, lane, q0, q1, q2, q3);
}
/* the reverse for the 64 records a wavefront writes: every lane puts its record into the image, then in round j lane l
 * stores one 16-B part of the record of lane 16 j + l / 4: a quad writes one whole 64-B segment (64 write requests per
 * wavefront reach L2 instead of 256 partial ones). idx = the record's index in rec, < 0: this lane stores nothing. */
template <bool STREAM>
RT_DEV void wave_scatter_records(float4* __restrict__ rec, const int idx, float4* s_wave, const int lane, const float4& q0, const float4& q1,
                                 const float4& q2, const float4& q3)
{
    const int rot = (lane >> 2) & 3;
    s_wave[4 * lane + (0 ^ rot)] = q0;
    s_wave[4 * lane + (1 ^ rot)] = q1;
    s_wave[4 * lane + (2 ^ rot)] = q2;
    s_wave[4 * lane + (3 ^ rot)] = q3;
    int to[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) to[j] = __shfl(idx, 16 * j + (lane >> 2));
    RT_WAVE_LDS_FENCE();
    const uint32_t part16 = (uint32_t)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    char* base = reinterpret_cast<char*>(rec);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (to[j] >= 0) store_stream<STREAM ? 1 : 0>(reinterpret_cast<float4*>(base + ((uint32_t)to[j] * 64u + part16)), s_wave[64 * j + lane]);
    RT_WAVE_LDS_FENCE();
}
/* The pass for one wavefront of pixels (8 x 8 tile): everything of k_spatial_coop up to the stores. TB = threads of the
 * workgroup's tile (BLOCK: four wavefronts, TRACE_BLOCK: one). s_wave: the wavefront's 4-KB record image.
 * FUSED: halo records live in the exchange lists (HaloFuse, multi-GPU strips); otherwise every record is in in_rec */
struct SpatialOut
{
    Res r;
    float4 G0, G1;
    size_t li;
    int x, row;
    bool in_image, active;
};
template <int TB, bool FUSED>
RT_DEV void spatial_coop_wave(const FrameParams& P, const HaloFuse& F, const float4* __restrict__ g0, const float4* __restrict__ g1,
                              const float4* __restrict__ in_rec, const float4* __restrict__ in_rad, float4* s_wave, const int lane, SpatialOut& o)
{
    int x = 0, row = P.lrow0;
    const bool in_image = tile_pixel<TB>(P, x, row);
    const int yi = P.H - 1 - row;
    const size_t li = in_image ? (size_t)x + (size_t)(row - P.lrow0) * P.W : 0; /* out of the image: names record 0, stores nothing */
    float4 G0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), G1 = G0;
    if (in_image) { G0 = g0[li]; G1 = g1[li]; }
    const bool active = in_image && (as_uint(G1.w) & GB_SHADED);
    const f3 sp = F3(G0.x, G0.y, G0.z), sn = F3(G1.x, G1.y, G1.z);
    PCG rng = pcg_init(hashPCG4((uint32_t)x, (uint32_t)yi, (uint32_t)P.frame, (uint32_t)(2 + P.pass)), 0);
    float4 q0, q1, q2, q3;
    wave_gather_records_at(in_rec, (uint32_t)li, s_wave, lane, q0, q1, q2, q3);
    bool own_shaded;
    Res r = res_from_parts(q0, q1, q2, q3, own_shaded);
    const float4* rad_from = in_rad + li;
    bool took_other = false;
    if (P.use_spatial)
    {
        const float scale = P.spatial_radius / 1.96f;
        /* rejection_heuristics' first distance is that of r's origin: the own one until a neighbour's sample is taken, that
         * neighbour's from then on (res_take_sample copies the origin) - the same expression on the same operands */
        float d0 = length(r.org_p - P.eye);
        for (int k = 0; k < P.spatial_count; ++k) /* wave-uniform trip count: every lane takes part in every round's fetch */
        {
            bool have = false;
            uint32_t nidx = (uint32_t)li;
            const float4* nq = in_rec + 4 * li;
            const float4* nrad = nullptr;
            if (active)
            {
                const float rv0 = rng.uniformf();
                const float rv1 = rng.uniformf();
                /* common/reservoir.hpp:89-95 with portable log/cos/sin */
                const float radius = sqrt_guarded(fmax_dev(-2.0f * pm_logf(rv0), 0.0f));
                const float phi = 2.0f * kPI * rv1;
                float sn_phi, cs_phi;
                pm_sincosf(phi, &sn_phi, &cs_phi);
                const float gx = radius * cs_phi, gy = radius * sn_phi;
                const int nx = f2i_sat((float)x + scale * gx);
                const int ny = f2i_sat((float)yi + scale * gy);
                const int nrow = P.H - 1 - ny;
                const int lr = nrow - P.lrow0;
                have = !(nx < 0 || nx >= P.W || ny < 0 || ny >= P.H) && !(nx == x && ny == yi) && !(lr < 0 || lr >= P.lrows);
                if (have)
                {
                    nidx = (uint32_t)nx + (uint32_t)lr * (uint32_t)P.W;
                    if (FUSED) nq = halo_record(F, P.W, in_rec, in_rad, (size_t)nidx, nx, nrow, nrad);
                    else nrad = in_rad + nidx;
                }
            }
            if (FUSED) wave_gather_records(nq, s_wave, lane, q0, q1, q2, q3);
            else wave_gather_records_at(in_rec, nidx, s_wave, lane, q0, q1, q2, q3);
            bool n_shaded;
            Res nr = res_from_parts(q0, q1, q2, q3, n_shaded);
            if (have && n_shaded) /* not shaded: sky or emissive neighbour (10_restir_di.cu:326-338) */
            {
                float p_hat_y = target_unshadowed(sp, sn, nr.hit_p, nr.hit_n, nr.lum);
                if (P.vis_reuse) p_hat_y *= nr.vis ? 1.0f : 0.0f;
                /* rejection_heuristics(r.org_p, r.org_n, nr.org_p, nr.org_n, P.eye) (rt_device.h) with d0 carried */
                const float d1 = length(nr.org_p - P.eye);
                const float diff = (d1 - d0) * (d1 - d0) / d0;
                float rw = 1.0f;
                rw *= pm_expf(-32.0f * diff);
                rw *= pm_pow8f(fmax_dev(dot(r.org_n, nr.org_n), 0.0f));
                nr.M = scale_M(nr.M, rw);
                const float weight = p_hat_y * nr.ucw * (float)nr.M;
                const float u = rng.uniformf();
                r.w_sum += weight;
                r.M += nr.M;
                if (reservoir_accept(u, weight, r.w_sum))
                {
                    res_take_sample(r, nr);
                    d0 = d1;
                    rad_from = nrad;
                    took_other = true;
                }
            }
        }
        const float p_hat = target_unshadowed(sp, sn, r.hit_p, r.hit_n, r.lum);
        r.ucw = p_hat > 0.0f ? r.w_sum / ((float)r.M * p_hat) : 0.0f;
    }
    if (active)
    {
        const float4 rq = *rad_from;
        r.rad = F3(rq.x, rq.y, rq.z);
        r.ownv = took_other ? 0u : as_uint(rq.w);
    }
    else
        r = res_zero(); /* the reference stores nothing here (:275-287); we keep the shaded bit valid */
    o.r = r; o.G0 = G0; o.G1 = G1; o.li = li; o.x = x; o.row = row; o.in_image = in_image; o.active = active;
}
/* the stores of the pass: the 64-B record through the wavefront's image, the radiance side record, the halo lists (FUSED) */
template <bool FUSED>
RT_DEV void spatial_coop_store(const FrameParams& P, const HaloFuse& F, float4* s_wave, const int lane, const SpatialOut& o,
                               float4* __restrict__ out_rec, float4* __restrict__ out_rad)
{
    const Res& r = o.r;
    const uint32_t mbits = ((uint32_t)r.M & RES_M_MASK) | (r.vis ? RES_VIS_BIT : 0u) | (o.active ? RES_SHADED_BIT : 0u);
    wave_scatter_records(out_rec, o.in_image ? (int)o.li : -1, s_wave, lane, make_float4(r.hit_p.x, r.hit_p.y, r.hit_p.z, r.ucw),
                         make_float4(r.hit_n.x, r.hit_n.y, r.hit_n.z, as_float(mbits)), make_float4(r.org_p.x, r.org_p.y, r.org_p.z, r.lum),
                         make_float4(r.org_n.x, r.org_n.y, r.org_n.z, r.w_sum));
    if (!o.in_image) return;
    store_stream<2>(out_rad + o.li, make_float4(r.rad.x, r.rad.y, r.rad.z, as_float(r.ownv)));
    if (FUSED) res_give(F, P.W, o.li, o.x, o.row, r, o.active);
}
/* TB: 256 threads on a 32 x 8 tile, or (r05, rt_tuning key 8 = 4) one wavefront on an 8 x 8 tile */
template <int WAVES, bool FUSED, int TB = BLOCK>
__global__ __launch_bounds__(TB) void k_spatial_coop(
    SceneView S, FrameParams P, HaloFuse F, const float4* __restrict__ g0, const float4* __restrict__ g1, const float4* __restrict__ in_rec,
    const float4* __restrict__ in_rad, float4* __restrict__ out_rec, float4* __restrict__ out_rad)
{
    occupancy_bound<WAVES>();
    __shared__ __attribute__((aligned(16))) float4 s_img[TB / 64][256];
    RT_WAVE_CLOCK(P);
    const int lane = threadIdx.x & 63;
    float4* s_wave = s_img[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))]; /* a scalar: the LDS-DMA base (M0) without per-load lane reads */
    SpatialOut o;
    spatial_coop_wave<TB, FUSED>(P, F, g0, g1, in_rec, in_rad, s_wave, lane, o);
    spatial_coop_store<FUSED>(P, F, s_wave, lane, o, out_rec, out_rad);
}

#ifdef RT_EXPERIMENTS /* rt_tuning key 8 = 3: software-pipelined pass (r04: slower), A/B only */
/* Software-pipelined form of k_spatial_coop (rt_tuning key 8 = 3, r04). k_spatial_coop's neighbour loop is
 * draw -> fetch -> s_waitcnt vmcnt(0) -> merge, five times in series: 55 % of its wave cycles wait for memory with nothing in
 * flight during a merge. What serialises it is the reference's RNG protocol (the merge draw of neighbour k is consumed only if
 * that neighbour is shaded, so neighbour k+1's address waits for record k). As in k_spatial_lds, the tile's +-87-pixel window
 * of shaded bits (5 KB) is staged in LDS, so ALL draws and neighbour addresses are known before the first record arrives; the
 * records are then fetched four lanes per 64-B record as in k_spatial_coop, but into registers (4 x dwordx4 per lane), so that
 * the fetch of neighbour k+1 is in flight while neighbour k is transposed through the wavefront's 4 KB LDS image and merged.
 * Same draws in the same order, same arithmetic, same results. Whole-frame contexts, radius <= 30, <= 5 neighbours. */
RT_DEV void wave_stage_issue(const float4* __restrict__ rec, const uint32_t idx, const int lane, float4& s0, float4& s1, float4& s2, float4& s3)
{
    const uint32_t f0 = (uint32_t)__shfl((int)idx, (lane >> 2)), f1 = (uint32_t)__shfl((int)idx, 16 + (lane >> 2)),
                   f2 = (uint32_t)__shfl((int)idx, 32 + (lane >> 2)), f3 = (uint32_t)__shfl((int)idx, 48 + (lane >> 2));
    const uint32_t part16 = (uint32_t)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    const char* base = reinterpret_cast<const char*>(rec);
    s0 = *reinterpret_cast<const float4*>(base + (f0 * 64u + part16));
    s1 = *reinterpret_cast<const float4*>(base + (f1 * 64u + part16));
    s2 = *reinterpret_cast<const float4*>(base + (f2 * 64u + part16));
    s3 = *reinterpret_cast<const float4*>(base + (f3 * 64u + part16));
}
/* the staged parts -> the wavefront's image -> every lane's own record */
RT_DEV void wave_stage_take(float4* s_wave, const int lane, const float4& s0, const float4& s1, const float4& s2, const float4& s3, float4& q0,
                            float4& q1, float4& q2, float4& q3)
{
    s_wave[lane] = s0;
    s_wave[64 + lane] = s1;
    s_wave[128 + lane] = s2;
    s_wave[192 + lane] = s3;
    RT_WAVE_LDS_FENCE();
    const int rot = (lane >> 2) & 3;
    q0 = s_wave[4 * lane + (0 ^ rot)];
    q1 = s_wave[4 * lane + (1 ^ rot)];
    q2 = s_wave[4 * lane + (2 ^ rot)];
    q3 = s_wave[4 * lane + (3 ^ rot)];
    RT_WAVE_LDS_FENCE(); /* the next round overwrites the image */
}
template <int WAVES>
__global__ __launch_bounds__(BLOCK) void k_spatial_pipe(
    FrameParams P, const uint32_t* __restrict__ bits, const float4* __restrict__ g0, const float4* __restrict__ g1, const float4* __restrict__ in_rec,
    const float4* __restrict__ in_rad, float4* __restrict__ out_rec, float4* __restrict__ out_rad)
{
    occupancy_bound<WAVES>();
    __shared__ uint32_t s_bits[SPL_ROWS * SPL_WORDS];
    __shared__ __attribute__((aligned(16))) float4 s_img[BLOCK / 64][256];
    const int lane = threadIdx.x & 63;
    float4* s_wave = s_img[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
    int x = 0, row = P.row0;
    const bool in_image = tile_pixel<BLOCK>(P, x, row);
    /* the tile origin, from any thread's own (x, row) — see k_spatial_lds */
#if RT_WAVE_8X8 && RT_TILE_W == 32
    const int in_tile_row = (threadIdx.x >> 3) & 7;
#else
    const int in_tile_row = threadIdx.x >> TILE_W_LOG2;
#endif
    const int tx0 = x & ~(TILE_W - 1), trow0 = row - in_tile_row;
    const int words = (P.W + 31) / 32, tw0 = (tx0 >> 5) - 3;
    const size_t li = in_image ? (size_t)x + (size_t)(row - P.lrow0) * P.W : 0;
    /* the own record's fetch first: it travels while the window is staged and the draws are made */
    float4 s0, s1, s2, s3;
    wave_stage_issue(in_rec, (uint32_t)li, lane, s0, s1, s2, s3);
    float4 G0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), G1 = G0;
    /* radiance side record of the sample the reservoir holds: the own one now, a neighbour's when its sample is taken (the
     * request then travels behind the remaining merges instead of in front of the stores) */
    float4 rq = G0;
    if (in_image) { G0 = g0[li]; G1 = g1[li]; rq = in_rad[li]; }
    for (int i = threadIdx.x; i < SPL_ROWS * SPL_WORDS; i += BLOCK)
    {
        const int r = i / SPL_WORDS, w = i - r * SPL_WORDS;
        const int grow = trow0 - SPL_HALO + r, gw = tw0 + w;
        const int lr = grow - P.lrow0;
        s_bits[i] = (lr >= 0 && lr < P.lrows && gw >= 0 && gw < words) ? bits[(size_t)lr * words + gw] : 0u;
    }
    __syncthreads();
    const int yi = P.H - 1 - row;
    const bool active = in_image && (as_uint(G1.w) & GB_SHADED);
    const f3 sp = F3(G0.x, G0.y, G0.z), sn = F3(G1.x, G1.y, G1.z);
    /* 1. every draw and every neighbour address of the pass from the RNG and the staged bits (10_restir_di.cu:305-340); the
     * own record is taken and neighbour 0's fetch issued as soon as neighbour 0 is known, so that it travels behind the
     * draws of neighbours 1..4 */
    uint32_t nidx[5];
    float ud[5];
    uint32_t take = 0u; /* bit k: neighbour k reaches the merge */
    const int count = P.use_spatial ? P.spatial_count : 0;
    PCG rng = pcg_init(hashPCG4((uint32_t)x, (uint32_t)yi, (uint32_t)P.frame, (uint32_t)(2 + P.pass)), 0);
    const float scale = P.spatial_radius / 1.96f;
    float4 q0, q1, q2, q3;
    Res r = res_zero();
#pragma unroll
    for (int k = 0; k < 5; ++k)
    {
        nidx[k] = (uint32_t)li; ud[k] = 0.0f;
        if (k < count && active)
        {
            const float rv0 = rng.uniformf();
            const float rv1 = rng.uniformf();
            const float radius = sqrt_guarded(fmax_dev(-2.0f * pm_logf(rv0), 0.0f));
            const float phi = 2.0f * kPI * rv1;
            float sn_phi, cs_phi;
            pm_sincosf(phi, &sn_phi, &cs_phi);
            const float gx = radius * cs_phi, gy = radius * sn_phi;
            const int nx = f2i_sat((float)x + scale * gx);
            const int ny = f2i_sat((float)yi + scale * gy);
            const int nrow = P.H - 1 - ny, lr = nrow - P.lrow0;
            const bool okn = !(nx < 0 || nx >= P.W || ny < 0 || ny >= P.H) && !(nx == x && ny == yi) && !(lr < 0 || lr >= P.lrows);
            if (okn)
            {
                /* |offset| <= 86.43 px for radius <= 30: inside the staged window */
                const int wr = nrow - (trow0 - SPL_HALO), ww = (nx >> 5) - tw0;
                const uint32_t word = s_bits[wr * SPL_WORDS + ww];
                if (word & (1u << (nx & 31)))
                {
                    nidx[k] = (uint32_t)nx + (uint32_t)lr * (uint32_t)P.W;
                    ud[k] = rng.uniformf();
                    take |= 1u << k;
                }
            }
        }
        if (k == 0)
        {
            /* 2a. the own record (requested before the window was staged) */
            wave_stage_take(s_wave, lane, s0, s1, s2, s3, q0, q1, q2, q3);
            bool own_shaded;
            r = res_from_parts(q0, q1, q2, q3, own_shaded);
            if (count > 0) wave_stage_issue(in_rec, nidx[0], lane, s0, s1, s2, s3);
        }
    }
    /* 2. the merge chain of :340-371 with the next neighbour's record in flight */
    bool took_other = false;
    float d0 = length(r.org_p - P.eye);
#pragma unroll
    for (int k = 0; k < 5; ++k)
    {
        if (k < count) /* wave-uniform */
        {
            wave_stage_take(s_wave, lane, s0, s1, s2, s3, q0, q1, q2, q3);
            if (k + 1 < count) wave_stage_issue(in_rec, nidx[k + 1 < 5 ? k + 1 : 4], lane, s0, s1, s2, s3);
            if (take & (1u << k))
            {
                bool n_shaded;
                Res nr = res_from_parts(q0, q1, q2, q3, n_shaded);
                float p_hat_y = target_unshadowed(sp, sn, nr.hit_p, nr.hit_n, nr.lum);
                if (P.vis_reuse) p_hat_y *= nr.vis ? 1.0f : 0.0f;
                /* rejection_heuristics(r.org_p, r.org_n, nr.org_p, nr.org_n, P.eye) (rt_device.h) with d0 carried */
                const float d1 = length(nr.org_p - P.eye);
                const float diff = (d1 - d0) * (d1 - d0) / d0;
                float rw = 1.0f;
                rw *= pm_expf(-32.0f * diff);
                rw *= pm_pow8f(fmax_dev(dot(r.org_n, nr.org_n), 0.0f));
                nr.M = scale_M(nr.M, rw);
                const float weight = p_hat_y * nr.ucw * (float)nr.M;
                r.w_sum += weight;
                r.M += nr.M;
                if (reservoir_accept(ud[k], weight, r.w_sum))
                {
                    res_take_sample(r, nr);
                    d0 = d1;
                    rq = in_rad[nidx[k]];
                    took_other = true;
                }
            }
        }
    }
    if (count > 0)
    {
        const float p_hat = target_unshadowed(sp, sn, r.hit_p, r.hit_n, r.lum);
        r.ucw = p_hat > 0.0f ? r.w_sum / ((float)r.M * p_hat) : 0.0f;
    }
    if (active)
    {
        r.rad = F3(rq.x, rq.y, rq.z);
        r.ownv = took_other ? 0u : as_uint(rq.w);
    }
    else
        r = res_zero(); /* the reference stores nothing here (:275-287); we keep the shaded bit valid */
    const uint32_t mbits = ((uint32_t)r.M & RES_M_MASK) | (r.vis ? RES_VIS_BIT : 0u) | (active ? RES_SHADED_BIT : 0u);
    wave_scatter_records(out_rec, in_image ? (int)li : -1, s_wave, lane, make_float4(r.hit_p.x, r.hit_p.y, r.hit_p.z, r.ucw),
                         make_float4(r.hit_n.x, r.hit_n.y, r.hit_n.z, as_float(mbits)), make_float4(r.org_p.x, r.org_p.y, r.org_p.z, r.lum),
                         make_float4(r.org_n.x, r.org_n.y, r.org_n.z, r.w_sum));
    if (!in_image) return;
    store_stream<2>(out_rad + li, make_float4(r.rad.x, r.rad.y, r.rad.z, as_float(r.ownv)));
}

#endif /* RT_EXPERIMENTS */
/* SURVEY.md §8(d) ALGORITHMIC bytes of one spatial_resampling launch, counted with the
 * reference's record sizes (Visibility 16 B, Reservoir 76 B): per pixel 16; per shaded pixel
 * +76 in +76 out; per neighbour that passed the on-screen / not-self tests +16, and +76 more if
 * it is shaded. Replays exactly the RNG draws of k_spatial (the accept decisions depend only on
 * the RNG and the shaded bits, not on reservoir contents). Measurement aid, not on the hot path. */
__global__ __launch_bounds__(BLOCK) void k_spatial_bytes(FrameParams P, const float4* __restrict__ g1,
                                                          const float4* __restrict__ in_rec,
                                                          unsigned long long* __restrict__ out, bool flags_from_g1 = false)
{
    int x, row;
    const bool ok = tile_pixel(P, x, row);
    unsigned long long bytes = 0, accepted = 0, merged = 0;
    if (ok)
    {
        const int yi = P.H - 1 - row;
        const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
        bytes = 16;
        if (as_uint(g1[li].w) & GB_SHADED)
        {
            bytes += 152;
            if (P.use_spatial)
            {
                PCG rng = pcg_init(hashPCG4((uint32_t)x, (uint32_t)yi, (uint32_t)P.frame, (uint32_t)(2 + P.pass)), 0);
                const float scale = P.spatial_radius / 1.96f;
                for (int k = 0; k < P.spatial_count; ++k)
                {
                    const float rv0 = rng.uniformf();
                    const float rv1 = rng.uniformf();
                    const float radius = sqrt_guarded(fmax_dev(-2.0f * pm_logf(rv0), 0.0f));
                    const float phi = 2.0f * kPI * rv1;
                    float sn_phi, cs_phi;
                    pm_sincosf(phi, &sn_phi, &cs_phi);
                    const int nx = f2i_sat((float)x + scale * (radius * cs_phi));
                    const int ny = f2i_sat((float)yi + scale * (radius * sn_phi));
                    if (nx < 0 || nx >= P.W || ny < 0 || ny >= P.H) continue;
                    if (nx == x && ny == yi) continue;
                    const int lr = P.H - 1 - ny - P.lrow0;
                    if (lr < 0 || lr >= P.lrows) continue;
                    bytes += 16;
                    accepted += 1;
                    /* the neighbour's shaded flag: from its record, or (strips with sparse halos, where
                     * only the records that will be gathered travel) from the exchanged G-buffer flags */
                    const size_t nid = (size_t)nx + (size_t)lr * P.W;
                    const bool n_shaded = flags_from_g1 ? (as_uint(g1[nid].w) & GB_SHADED) != 0u
                                                        : (as_uint(in_rec[4 * nid + 1].w) & RES_SHADED_BIT) != 0u;
                    if (!n_shaded) continue;
                    bytes += 76;
                    merged += 1; /* neighbours that reach the target function (:346-350) */
                    rng.uniformf();
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1)
    {
        bytes += __shfl_down(bytes, off);
        accepted += __shfl_down(accepted, off);
        merged += __shfl_down(merged, off);
    }
    if ((threadIdx.x & 63) == 0)
    {
        atomicAdd(&out[0], bytes);
        atomicAdd(&out[1], accepted);
        atomicAdd(&out[2], merged);
    }
}

/* ------------------------------------------------------- sparse reservoir halos (multi-GPU)
 * A strip only needs those neighbour-strip records its own pixels will actually gather. Which
 * ones is a pure function of the RNG and the shaded bits (exactly the replay of
 * k_spatial_bytes), so the RECEIVER marks them in a bitmap over the neighbour's boundary rows
 * (k_halo_mark), ships the bitmap once per frame, and the owner answers every pass with the
 * marked records only, in bitmap order (k_halo_sparse). Bitmap buffer (uint32 words):
 *   [0] = number of marked records, [1 .. nw] = bits (bit i of word w = pixel 32*w + i of the
 *   region, row-major from region row 0), [1+nw .. 1+2nw) = exclusive prefix counts per word. */
/* the neighbours' regions (side 0 below, side 1 above) marked by ONE launch over the own rows that can reach either;
 * rows[s] == 0: no neighbour on that side */
struct HaloRegions
{
    int row0[2], rows[2];
    size_t words[2]; /* rt_halo_bitmap_words of the region = distance between the bitmaps of consecutive passes */
    uint32_t* bitmaps[2];
    int quick; /* 1 = rows far from the regions test their draws against a radius bound first (rt_tuning key 18) */
};
/* WINDOW (r04): the marks of a workgroup's 32x8 tile go into an LDS bitmap of the tile's +-87-pixel window first (one per
 * pass: 182 rows x 7 words, the window of k_spatial_lds) and only its non-zero words reach the global bitmaps, one atomicOr
 * per word and workgroup. The direct form issues one global atomic per marked neighbour — ~1.2 M per frame of a 1080p strip,
 * many to the same word, at ~140 ns per same-address atomic (measured in the BVH builder): the kernel took 84 us
 * for a 135-row strip and 310 us for a 270-row 4K strip with ~12 us of arithmetic, a fifth of the strip's frame
 * (profiles/r04_strip_timelines.txt). Needs W % 32 == 0 (window words = bitmap words), a reach <= 87 px and <= 3 passes per
 * launch; the host falls back to the direct form otherwise. Same marks (the tests compare both with the replay). */
constexpr int MARK_MAX_PASSES = 3;
template <bool WINDOW>
RT_DEV void halo_mark_pixel(const FrameParams& P, const float4* __restrict__ g1, const HaloRegions& R, int pass0, int pi0, int pi1, bool in_image, int x,
                            int row, int trow0, int tw0, uint32_t* s_win, const uint32_t* s_bits);
/* `bits` (WINDOW): shaded bit per pixel of all local rows (k_shaded_bitmap), staged for the tile's window: the replay's
 * "is the neighbour shaded" test — which decides whether the merge draw is consumed — reads LDS instead of gathering 4 bytes
 * of a 16-byte G-buffer record per neighbour (15 dependent 64-cache-line gathers per pixel were what the kernel waited for) */
/* r06: gridDim.y = 1: a workgroup replays all n_pass passes of its tile one after the other (r02-r05); gridDim.y = n_pass: one workgroup
 * per (tile, pass) - a strip's mark launch is half a generation of wavefronts whose every thread replays 3 x 5 neighbour draws in
 * series (log, sqrt, sincos each): it lasts as long as one thread's chain, so a third of the chain on three times the wavefronts
 * is a third of the launch (rt_tuning key 26). Same marks: a pass's bitmap depends on nothing of the other passes. */
template <bool WINDOW>
__global__ __launch_bounds__(BLOCK) void k_halo_mark(FrameParams P, const float4* __restrict__ g1, const uint32_t* __restrict__ bits, HaloRegions R, int pass0,
                                                      int n_pass_all)
{
    const int pi0 = gridDim.y > 1 ? (int)blockIdx.y : 0, pi1 = gridDim.y > 1 ? pi0 + 1 : n_pass_all; /* bitmap slots [pi0, pi1) */
    const int n_pass = pi1 - pi0;
    __shared__ uint32_t s_win[WINDOW ? MARK_MAX_PASSES * SPL_ROWS * SPL_WORDS : 1];
    __shared__ uint32_t s_bits[WINDOW ? SPL_ROWS * SPL_WORDS : 1];
    int x = 0, row = P.row0;
    const bool in_image = tile_pixel(P, x, row);
#if RT_WAVE_8X8 && RT_TILE_W == 32
    const int in_tile_row = (threadIdx.x >> 3) & 7;
#else
    const int in_tile_row = threadIdx.x >> TILE_W_LOG2;
#endif
    const int trow0 = row - in_tile_row, tw0 = ((x & ~(TILE_W - 1)) >> 5) - 3; /* the tile's origin, from any thread (k_spatial_lds) */
    if (WINDOW)
    {
        for (int i = threadIdx.x; i < n_pass * SPL_ROWS * SPL_WORDS; i += BLOCK) s_win[i] = 0u;
        const int words = P.W >> 5;
        for (int i = threadIdx.x; i < SPL_ROWS * SPL_WORDS; i += BLOCK)
        {
            const int r = i / SPL_WORDS, w = i - r * SPL_WORDS;
            const int lr = trow0 - SPL_HALO + r - P.lrow0, gw = tw0 + w;
            s_bits[i] = (lr >= 0 && lr < P.lrows && gw >= 0 && gw < words) ? bits[(size_t)lr * words + gw] : 0u;
        }
        __syncthreads();
    }
    halo_mark_pixel<WINDOW>(P, g1, R, pass0, pi0, pi1, in_image, x, row, trow0, tw0, s_win, s_bits);
    if (WINDOW)
    {
        __syncthreads();
        const int words_per_row = P.W >> 5;
        for (int i = threadIdx.x; i < n_pass * SPL_ROWS * SPL_WORDS; i += BLOCK)
        {
            const uint32_t w = s_win[i];
            if (!w) continue;
            const int pl = i / (SPL_ROWS * SPL_WORDS), rem = i - pl * (SPL_ROWS * SPL_WORDS), pi = pi0 + pl; /* window slot -> bitmap slot */
            const int r = rem / SPL_WORDS, ww = rem - r * SPL_WORDS;
            const int nrow = trow0 - SPL_HALO + r, gw = tw0 + ww;
#pragma unroll
            for (int sd = 0; sd < 2; ++sd)
                if (nrow >= R.row0[sd] && nrow < R.row0[sd] + R.rows[sd])
                    atomicOr(&R.bitmaps[sd][(size_t)pi * R.words[sd] + 1 + (size_t)(nrow - R.row0[sd]) * words_per_row + gw], w);
        }
    }
}
template <bool WINDOW>
RT_DEV void halo_mark_pixel(const FrameParams& P, const float4* __restrict__ g1, const HaloRegions& R, int pass0, int pi0, int pi1, bool in_image, int x,
                            int row, int trow0, int tw0, uint32_t* s_win, const uint32_t* s_bits)
{
    if (!in_image) return;
    const int yi = P.H - 1 - row;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
    if (!(as_uint(g1[li].w) & GB_SHADED) || !P.use_spatial) return;
    const float scale = P.spatial_radius / 1.96f;
    /* Quick reject (r03): a neighbour lands at most scale * radius + 1 rows from this pixel (|sin| <= 1, one unit for the
     * float add and the truncation), radius = sqrt(-2 log rv0); rv0 of neighbour k is draw 2k .. 3k of the pass's stream
     * (two or three draws per neighbour). A pixel dmin rows from the nearest region can therefore mark something only if
     * one of the first 3 (count - 1) + 1 draws is below exp(-((dmin - 1.01) / scale)^2 / 2) — taken 0.1 % larger, far more
     * than the <= 1 ulp of the portable log / exp / sqrt; rv0 = 0 (radius = inf, the reference's off-screen case) is below
     * any threshold and takes the full path. Rows more than ~3 sigma from the edge (half of a band) skip the log / sqrt /
     * sincos replay for 9 passes in 10: -30 % of the kernel at 4K. The marks are those of the full replay. */
    int dmin = 0x7fffffff;
#pragma unroll
    for (int sd = 0; sd < 2; ++sd)
        if (R.rows[sd] > 0)
        {
            const int d = row < R.row0[sd] ? R.row0[sd] - row : (row >= R.row0[sd] + R.rows[sd] ? row - (R.row0[sd] + R.rows[sd] - 1) : 0);
            dmin = d < dmin ? d : dmin;
        }
    const bool far_row = R.quick && dmin >= 40 && P.spatial_count >= 1 && P.spatial_count <= 8;
    float thr = 2.0f;
    if (far_row)
    {
        const float q = ((float)dmin - 1.01f) / scale;
        thr = pm_expf(-0.5f * q * q) * 1.001f;
    }
    for (int pi = pi0; pi < pi1; ++pi) /* one bitmap per spatial pass, same launch */
    {
        PCG rng = pcg_init(hashPCG4((uint32_t)x, (uint32_t)yi, (uint32_t)P.frame, (uint32_t)(2 + pass0 + pi)), 0);
        if (far_row)
        {
            PCG probe = rng;
            float lowest = 2.0f;
            const int n_draws = 3 * (P.spatial_count - 1) + 1;
            for (int k = 0; k < n_draws; ++k) lowest = fminf(lowest, probe.uniformf());
            if (!(lowest < thr)) continue; /* no neighbour of this pass can reach a region */
        }
        for (int k = 0; k < P.spatial_count; ++k)
        {
            const float rv0 = rng.uniformf();
            const float rv1 = rng.uniformf();
            const float radius = sqrt_guarded(fmax_dev(-2.0f * pm_logf(rv0), 0.0f));
            const float phi = 2.0f * kPI * rv1;
            float sn_phi, cs_phi;
            pm_sincosf(phi, &sn_phi, &cs_phi);
            const int nx = f2i_sat((float)x + scale * (radius * cs_phi));
            const int ny = f2i_sat((float)yi + scale * (radius * sn_phi));
            if (nx < 0 || nx >= P.W || ny < 0 || ny >= P.H) continue;
            if (nx == x && ny == yi) continue;
            const int nrow = P.H - 1 - ny;
            const int lr = nrow - P.lrow0;
            if (lr < 0 || lr >= P.lrows) continue;
            if (WINDOW)
            {
                /* every row of the window, region or not: the flush keeps the region rows */
                atomicOr(&s_win[((pi - pi0) * SPL_ROWS + (nrow - (trow0 - SPL_HALO))) * SPL_WORDS + ((nx >> 5) - tw0)], 1u << (nx & 31));
            }
            else
            {
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
                    if (nrow >= R.row0[sd] && nrow < R.row0[sd] + R.rows[sd])
                    {
                        const uint32_t bit = (uint32_t)(nrow - R.row0[sd]) * (uint32_t)P.W + (uint32_t)nx;
                        atomicOr(&R.bitmaps[sd][(size_t)pi * R.words[sd] + 1 + (bit >> 5)], 1u << (bit & 31u));
                    }
            }
            if (WINDOW)
            {
                if (!(s_bits[(nrow - (trow0 - SPL_HALO)) * SPL_WORDS + ((nx >> 5) - tw0)] & (1u << (nx & 31)))) continue;
            }
            else if (!(as_uint(g1[(size_t)nx + (size_t)lr * P.W].w) & GB_SHADED)) continue;
            rng.uniformf();
        }
    }
}
/* Exclusive prefix of the per-word popcounts of one bitmap + the total into word 0, CHUNKED over many small workgroups (r04):
 * workgroup `chunk` owns words [chunk * HALO_SCAN_CHUNK, +HALO_SCAN_CHUNK); it first sums the popcounts of everything in front
 * of its chunk itself (<= nw words spread over 256 threads: a few KB of L2-resident reads), then scans its own words. No
 * inter-workgroup dependence, no 1024-thread workgroup: r01-r03 scanned a bitmap with ONE 1024-thread workgroup, which on a
 * busy GPU waits for 16 free wave slots on one CU (26 -> 280 us at 4K in 8 strips, profiles/r04_strip_timelines.txt). */
constexpr int HALO_SCAN_CHUNK = 1024, HALO_SCAN_THREADS = 256;
RT_DEV uint32_t block_sum_256(uint32_t v, uint32_t* s_part)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    const uint32_t t = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    __syncthreads();
    return t;
}
RT_DEV void halo_scan_chunk(uint32_t* __restrict__ bitmap, int nw, int chunk, uint32_t* s_part)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int w_begin = chunk * HALO_SCAN_CHUNK;
    if (nw <= 0) { if (chunk == 0 && t == 0) bitmap[0] = 0u; return; } /* an empty region: count 0 */
    if (w_begin >= nw) return; /* wave-uniform: whole workgroups beyond the bitmap */
    /* 1. what lies in front of this chunk */
    uint32_t before = 0;
    for (int w = t; w < w_begin; w += HALO_SCAN_THREADS) before += (uint32_t)__popc(bitmap[1 + w]);
    const uint32_t base = block_sum_256(before, s_part);
    /* 2. this chunk: four consecutive words per thread */
    const int w0 = w_begin + 4 * t;
    uint32_t c[4], mine = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        c[k] = (w0 + k < nw) ? (uint32_t)__popc(bitmap[1 + w0 + k]) : 0u;
        mine += c[k];
    }
    uint32_t incl = mine; /* inclusive scan over the wavefront */
    for (int off = 1; off < 64; off <<= 1)
    {
        const uint32_t o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0;
    for (int k = 0; k < wave; ++k) wave_base += s_part[k];
    const uint32_t chunk_total = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    uint32_t run = base + wave_base + incl - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (w0 + k < nw) { bitmap[1 + nw + w0 + k] = run; run += c[k]; }
    if (t == 0 && w_begin + HALO_SCAN_CHUNK >= nw) bitmap[0] = base + chunk_total; /* the last chunk knows the total */
}
static inline int halo_scan_chunks(int nw) { const int n = (nw + HALO_SCAN_CHUNK - 1) / HALO_SCAN_CHUNK; return n > 0 ? n : 1; } /* never a grid of 0 */
/* blockIdx.x = chunk, blockIdx.y = bitmap */
__global__ __launch_bounds__(HALO_SCAN_THREADS) void k_halo_scan(uint32_t* __restrict__ bitmaps, int nw, size_t words_per_bitmap)
{
    __shared__ uint32_t s_part[4];
    halo_scan_chunk(bitmaps + (size_t)blockIdx.y * words_per_bitmap, nw, (int)blockIdx.x, s_part);
}
/* the bitmaps of both sides in one launch: blockIdx.x = chunk, blockIdx.y = pass, blockIdx.z = side */
__global__ __launch_bounds__(HALO_SCAN_THREADS) void k_halo_scan_sides(HaloRegions R)
{
    __shared__ uint32_t s_part[4];
    const int sd = blockIdx.z;
    if (R.rows[sd] <= 0) return;
    halo_scan_chunk(R.bitmaps[sd] + (size_t)blockIdx.y * R.words[sd], (int)((R.words[sd] - 1) / 2), (int)blockIdx.x, s_part);
}

/* PACK: marked records of rows [row0, row0+rows) -> dense list (64 B record + 16 B radiance each);
 * UNPACK: the reverse. */
struct HaloLists /* up to two (bitmap, rows, list) sets served by one launch: blockIdx.y picks the set */
{
    const uint32_t* bitmap[2];
    int nw[2], n_pix[2];
    size_t region_off[2];
    float4* list[2];
};
template <bool PACK>
__global__ void k_halo_sparse(HaloLists H, float4* __restrict__ rec, float4* __restrict__ radb)
{
    const int sd = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H.n_pix[sd]) return;
    const uint32_t* __restrict__ bitmap = H.bitmap[sd];
    const uint32_t word = bitmap[1 + (i >> 5)];
    if (!(word & (1u << (i & 31)))) return;
    const uint32_t idx = bitmap[1 + H.nw[sd] + (i >> 5)] + (uint32_t)__popc(word & ((1u << (i & 31)) - 1u));
    float4* L = H.list[sd] + 5 * (size_t)idx;
    const size_t p = H.region_off[sd] + (size_t)i;
    if (PACK)
    {
        L[0] = rec[4 * p + 0]; L[1] = rec[4 * p + 1]; L[2] = rec[4 * p + 2]; L[3] = rec[4 * p + 3];
        L[4] = radb[p];
    }
    else
    {
        rec[4 * p + 0] = L[0]; rec[4 * p + 1] = L[1]; rec[4 * p + 2] = L[2]; rec[4 * p + 3] = L[3];
        radb[p] = L[4];
    }
}
/* up to 8 device-to-device copies in ONE launch (blockIdx.y = part): the stand-in transports of the strip driver move the
 * parts of an exchange with it, as one grouped RCCL send/recv is one launch */
struct CopyParts
{
    const char* src[8];
    char* dst[8];
    size_t bytes[8];
};
__global__ void k_copy_parts(CopyParts P)
{
    const int part = blockIdx.y;
    const char* __restrict__ s = P.src[part];
    char* __restrict__ d = P.dst[part];
    const size_t n = P.bytes[part];
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)s | (uintptr_t)d) & 15u) == 0u)
    {
        const size_t n16 = n / 16;
        const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(s);
        uint4* __restrict__ d4 = reinterpret_cast<uint4*>(d);
        for (size_t i = tid; i < n16; i += stride) d4[i] = s4[i];
        for (size_t i = n16 * 16 + tid; i < n; i += stride) d[i] = s[i];
    }
    else if ((((uintptr_t)s | (uintptr_t)d) & 3u) == 0u)
    {
        const size_t n4 = n / 4;
        const uint32_t* __restrict__ s1 = reinterpret_cast<const uint32_t*>(s);
        uint32_t* __restrict__ d1 = reinterpret_cast<uint32_t*>(d);
        for (size_t i = tid; i < n4; i += stride) d1[i] = s1[i];
        for (size_t i = n4 * 4 + tid; i < n; i += stride) d[i] = s[i];
    }
    else
        for (size_t i = tid; i < n; i += stride) d[i] = s[i];
}

/* WIRE_MODEL transport of the strip driver (r05): the time an xGMI link would take, as a DEPENDENT delay on the stream the
 * exchange runs on — no host sleep, no host involvement. k_wire_stamp notes the GPU's wall clock when the stream reaches it
 * (= the moment the send's data is ready); k_wire_wait, enqueued behind the exchange's RCCL kernel, lets the stream go on no
 * earlier than `ticks` after that stamp. One wavefront, sleeping between its clock reads: it takes no slots worth naming. */
__global__ void k_wire_stamp(unsigned long long* __restrict__ slot) { *slot = wall_clock64(); }
__global__ void k_wire_wait(const unsigned long long* __restrict__ slot, unsigned long long ticks)
{
    const unsigned long long t0 = *slot;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

/* shaded flags of rows as bytes (halo rows of the G-buffer only ever hold these flags) */
template <bool PACK>
__global__ void k_halo_flags(float4* __restrict__ g1, size_t off, int n_pix, uint8_t* __restrict__ bytes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix) return;
    if (PACK) bytes[i] = (uint8_t)(as_uint(g1[off + i].w) & 0xffu);
    else g1[off + i] = make_float4(0.0f, 0.0f, 0.0f, as_float((uint32_t)bytes[i]));
}

/* -------------------------------------------------------------------- resolve */
RT_DEV float aces(float x)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return (x * (a * x + b)) / (x * (c * x + d) + e);
}
RT_DEV uint32_t to_u8(float v)
{
    const float c = fminf(fmax_dev(v, 0.0f), 255.0f);
    return (uint32_t)(int)c;
}
/* common/kernels/common.cu:30-74 for one pixel (display only; powf = portable exp(y*log(x))) */
RT_DEV uint32_t tone_map_rgba8(float4 a)
{
    const float gamma = 1.0f / 2.2f;
    const float r = pm_powf_pos(aces(a.x / a.w * 1.0f), gamma);
    const float g = pm_powf_pos(aces(a.y / a.w * 1.0f), gamma);
    const float b = pm_powf_pos(aces(a.z / a.w * 1.0f), gamma);
    return to_u8(r * 255.0f) | (to_u8(g * 255.0f) << 8) | (to_u8(b * 255.0f) << 16) | 0xff000000u;
}
/* examples/10_restir_di/10_restir_di.cu:390-459 */
/* register budget in wavefronts per SIMD. r01 (one lane, one ray): 6 (86 -> 80 VGPRs) was 3 % faster than none.
 * r02, with the work-sharing walk: 5 (96 VGPRs) is 3 % faster than 6 and equal to 4 (profiles/r02_ws_register_budgets.txt) */
#ifndef RT_RESOLVE_WAVES
#define RT_RESOLVE_WAVES 8 /* r04: 64 VGPRs without a spill once the SLP vectoriser is off: 0.257 (5) / 0.248 (7) / 0.240 ms (8) */
#endif
template <bool WS>
__global__ __launch_bounds__(TRACE_BLOCK, RT_RESOLVE_WAVES) void k_resolve(SceneView S, FrameParams P, const float4* __restrict__ g0,
                                                    const float4* __restrict__ g1,
                                                    const float4* __restrict__ rec,
                                                    const float4* __restrict__ radb, float4* __restrict__ accum,
                                                    uint32_t* __restrict__ pixels)
{
    /* `pixels` (rt_frame's tail, r05): tone_mapping (common/kernels/common.cu:30-74) reads only the pixel's own accumulation
     * value, the one this thread has just written — the staged frame maps it here instead of launching k_tone_mapping over the
     * buffer again (one launch and a 16-B read per pixel less). NULL for the per-kernel entry point rt_resolve: the reference's
     * resolve (10_restir_di.cu:390-459) does not touch the pixel buffer. */
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[(WS ? WIDE_LDS_ROWS : WIDE_LDS_STACK) * TRACE_BLOCK];
    RT_WAVE_CLOCK(P);
    int x, row;
    if (!tile_pixel<TRACE_BLOCK>(P, x, row)) return;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
    const float4 G0 = g0[li], G1 = g1[li];
    const int tri = as_int(G0.w);
    const uint32_t flags = as_uint(G1.w);
    if (tri < 0)
    {
        /* :414-417 stores {0,0,0,1} whatever `accumulate` says */
        accum[li] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        if (pixels) pixels[li] = tone_map_rgba8(make_float4(0.0f, 0.0f, 0.0f, 1.0f));
        return;
    }
    if (flags & GB_EMISSIVE)
    {
        const float4 ke = S.trimat[2 * (size_t)tri + 1];
        accum[li] = make_float4(ke.x, ke.y, ke.z, 1.0f);
        if (pixels) pixels[li] = tone_map_rgba8(make_float4(ke.x, ke.y, ke.z, 1.0f));
        return;
    }
    const f3 sp = F3(G0.x, G0.y, G0.z), sn = F3(G1.x, G1.y, G1.z);
    const float4 q0 = rec[4 * li + 0], q1 = rec[4 * li + 1];
    const float4 rq = radb[li];
    const float4 kd = S.trimat[2 * (size_t)tri];
    const f3 hp = F3(q0.x, q0.y, q0.z), hn = F3(q1.x, q1.y, q1.z);
    const f3 brdf = (1.0f / kPI) * F3(kd.x, kd.y, kd.z);
    const float G = geometry_term(sp, sn, hp, hn);
    /* the "fresh shadow ray" of :443-444 repeats the ray an earlier kernel of this frame walked from this pixel to this
     * sample if the record says so (rt_device.h, own-visibility flags): those lanes only help the others' walks */
    const uint32_t ownv = ownv_trusted(P.ownv_tag, as_uint(rq.w));
    const bool known = (ownv & OWNV_KNOWN) != 0u;
    if (P.stats)
    {
        const bool self = self_occluded(S.bvh.tv, tri, sp + 0.001f * sn, hp - sp, sn, !known);
        count_walk_flags(P.stats + 4 * WALK_RESOLVE, true, !known && !self, !known && self, known);
    }
    const bool walked = check_visibility_wide<TRACE_BLOCK, WS>(S.wide, s_stack, sp, sn, hp, !known, S.bvh.tv, tri);
    const float V = (known ? (ownv & OWNV_VISIBLE) != 0u : walked) ? 1.0f : 0.0f;
    const f3 radiance = brdf * G * V * F3(rq.x, rq.y, rq.z) * q0.w;
    float4 o = make_float4(radiance.x, radiance.y, radiance.z, 1.0f);
    if (P.accumulate)
    {
        const float4 a = accum[li];
        o = make_float4(a.x + radiance.x, a.y + radiance.y, a.z + radiance.z, a.w + 1.0f);
    }
    accum[li] = o;
    if (pixels) pixels[li] = tone_map_rgba8(o);
}


#ifdef RT_EXPERIMENTS /* rt_tuning keys 23 (last pass + resolve fused, r05: slower) and 15 (resolve as a stream, r02: slower), A/B only */
/* The LAST spatial pass and resolve in one kernel (r05, rt_tuning key 23). The reference launches spatial_resampling for the
 * last time (10_restir_di.cu:256-388) and then resolve (:390-459), which reads of that pass's output only the pixel's own
 * reservoir — and nothing else reads it (10_restir_di.cpp:355-379; the temporal history was saved before the passes, :314-321).
 * Here the wavefront that has merged a pixel's neighbours shades the pixel at once: the kernel-wide barrier between the two
 * becomes a per-pixel dependency, wavefronts waiting for gathered records (the pass: vector-L1 miss path) run beside wavefronts
 * walking shadow rays (resolve: issue), and the record's trip through HBM — 80 B written, 80 + 32 B read back per pixel — is
 * not on the path. STORE: the pass's output is written as the reference's kernel does (rt_download(RT_BUF_RES_*), the halo
 * lists of strips, the per-kernel sequence: identical buffers); without it the records stay in registers.
 * One-wavefront workgroups on 8 x 8 tiles; the 4-KB record image of the gathers is the LDS stack of the walk afterwards. */
#ifndef RT_SPATIAL_RESOLVE_WAVES
#define RT_SPATIAL_RESOLVE_WAVES 6
#endif
template <bool FUSED, bool STORE>
__global__ __launch_bounds__(TRACE_BLOCK, RT_SPATIAL_RESOLVE_WAVES) void k_spatial_resolve(
    SceneView S, FrameParams P, HaloFuse F, const float4* __restrict__ g0, const float4* __restrict__ g1, const float4* __restrict__ in_rec,
    const float4* __restrict__ in_rad, float4* __restrict__ out_rec, float4* __restrict__ out_rad, float4* __restrict__ accum,
    uint32_t* __restrict__ pixels)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[WIDE_LDS_ROWS * TRACE_BLOCK];
    static_assert(sizeof(s_stack) >= 4096, "the record image needs 64 x 64 B");
    float4* s_wave = reinterpret_cast<float4*>(s_stack);
    const int lane = threadIdx.x & 63;
    SpatialOut o;
    spatial_coop_wave<TRACE_BLOCK, FUSED>(P, F, g0, g1, in_rec, in_rad, s_wave, lane, o);
    if (STORE) spatial_coop_store<FUSED>(P, F, s_wave, lane, o, out_rec, out_rad);
    /* resolve (10_restir_di.cu:390-459) of the same pixel, from the registers */
    const int tri = as_int(o.G0.w);
    const bool shade = o.active; /* shaded surface: neither sky nor emissive */
    if (o.in_image && !shade)
    {
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        if (tri >= 0) { const float4 ke = S.trimat[2 * (size_t)tri + 1]; a = make_float4(ke.x, ke.y, ke.z, 1.0f); }
        accum[o.li] = a;
        if (pixels) pixels[o.li] = tone_map_rgba8(a);
    }
    const f3 sp = F3(o.G0.x, o.G0.y, o.G0.z), sn = F3(o.G1.x, o.G1.y, o.G1.z);
    const Res& r = o.r;
    const uint32_t ownv = ownv_trusted(P.ownv_tag, r.ownv);
    const bool known = (ownv & OWNV_KNOWN) != 0u;
    const bool need = shade && !known;
    if (P.stats && shade)
    {
        const bool self = self_occluded(S.bvh.tv, tri, sp + 0.001f * sn, r.hit_p - sp, sn, !known);
        count_walk_flags(P.stats + 4 * WALK_RESOLVE, true, !known && !self, !known && self, known);
    }
    RT_WAVE_LDS_FENCE(); /* the image becomes the stack */
    const bool walked = check_visibility_wide<TRACE_BLOCK, true>(S.wide, s_stack, sp, sn, r.hit_p, need, S.bvh.tv, shade ? tri : -1);
    if (!shade) return;
    const float4 kd = S.trimat[2 * (size_t)tri];
    const f3 brdf = (1.0f / kPI) * F3(kd.x, kd.y, kd.z);
    const float G = geometry_term(sp, sn, r.hit_p, r.hit_n);
    const float V = (known ? (ownv & OWNV_VISIBLE) != 0u : walked) ? 1.0f : 0.0f;
    const f3 radiance = brdf * G * V * r.rad * r.ucw;
    float4 a = make_float4(radiance.x, radiance.y, radiance.z, 1.0f);
    if (P.accumulate)
    {
        const float4 p = accum[o.li];
        a = make_float4(p.x + radiance.x, p.y + radiance.y, p.z + radiance.z, p.w + 1.0f);
    }
    accum[o.li] = a;
    if (pixels) pixels[o.li] = tone_map_rgba8(a);
}

/* resolve as a stream (bvh.h occluded_stream; rt_tuning key 15, evaluated and off by default): persistent one-wavefront
 * workgroups walk the launch's tiles round-robin (workgroup w: tiles w, w + gridDim.x, ...: the XCD band of tile rows
 * the other tracing kernels give it) and keep their lanes supplied with new pixels. Sky / emissive pixels are written
 * when they are fetched; a shaded pixel's ray is walked and the pixel is shaded when the ray is settled (its inputs
 * are read again then: the walk keeps only the pixel's index). */
#ifndef RT_RESOLVE_STREAM_WAVES
#define RT_RESOLVE_STREAM_WAVES 5
#endif
#ifndef RT_STREAM_CHUNK
#define RT_STREAM_CHUNK 256 /* jobs per atomic: four 8x8 tiles */
#endif
__global__ __launch_bounds__(TRACE_BLOCK, RT_RESOLVE_STREAM_WAVES) void k_resolve_stream(
    SceneView S, FrameParams P, const float4* __restrict__ g0, const float4* __restrict__ g1, const float4* __restrict__ rec,
    const float4* __restrict__ radb, float4* __restrict__ accum, int n_tiles)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[WIDE_LDS_ROWS * TRACE_BLOCK];
    /* No shared job counter (one returning atomicAdd per refill on one of eight counters: 1.30 ms; one per 256 jobs:
     * 0.87 ms; none: 0.48 ms): workgroup w walks the tiles w, w + gridDim.x, ... of the launch's tile order (gridDim.x
     * is a multiple of 8: the same XCD band every time); job = tile * 64 + thread. */
    unsigned int tile_b = blockIdx.x;             /* wave-uniform */
    unsigned int chunk_next = 0u, chunk_end = 0u; /* wave-uniform: job ids left of the current tile */
    bool first_tile = true;
    auto next_job = [&](unsigned int count, unsigned int& first, unsigned int& limit) -> bool {
        if (chunk_next >= chunk_end)
        {
            if (!first_tile) tile_b += gridDim.x;
            first_tile = false;
            if (tile_b >= (unsigned int)n_tiles) return false;
            chunk_next = tile_b * 64u;
            chunk_end = chunk_next + 64u;
        }
        first = chunk_next;
        limit = chunk_next + count < chunk_end ? chunk_next + count : chunk_end;
        chunk_next = limit;
        return true;
    };
    auto pixel_of = [&](unsigned int job, int& x, int& row) -> bool {
        return tile_pixel_at<TRACE_BLOCK>(P, (int)(job >> 6), (int)(job & 63u), x, row);
    };
    auto fetch = [&](unsigned int job, f3& ro, f3& rd, float& tmin, float& tmax) -> bool {
        int x, row;
        if (!pixel_of(job, x, row)) return false;
        const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
        const float4 G0 = g0[li], G1 = g1[li];
        const int tri = as_int(G0.w);
        if (tri < 0) { accum[li] = make_float4(0.0f, 0.0f, 0.0f, 1.0f); return false; }
        if (as_uint(G1.w) & GB_EMISSIVE)
        {
            const float4 ke = S.trimat[2 * (size_t)tri + 1];
            accum[li] = make_float4(ke.x, ke.y, ke.z, 1.0f);
            return false;
        }
        const float4 q0 = rec[4 * li + 0];
        const f3 sp = F3(G0.x, G0.y, G0.z), sn = F3(G1.x, G1.y, G1.z), hp = F3(q0.x, q0.y, q0.z);
        ro = sp + 0.001f * sn; /* check_visibility (common/raytrace.hpp:42-52) */
        rd = hp - sp;
        tmin = 0.0f; tmax = 0.99f;
        return true;
    };
    auto finish = [&](unsigned int job, bool occluded) {
        int x, row;
        pixel_of(job, x, row);
        const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
        const float4 G0 = g0[li], G1 = g1[li];
        const int tri = as_int(G0.w);
        const f3 sp = F3(G0.x, G0.y, G0.z), sn = F3(G1.x, G1.y, G1.z);
        const float4 q0 = rec[4 * li + 0], q1 = rec[4 * li + 1];
        const float4 rq = radb[li];
        const float4 kd = S.trimat[2 * (size_t)tri];
        const f3 hp = F3(q0.x, q0.y, q0.z), hn = F3(q1.x, q1.y, q1.z);
        const f3 brdf = (1.0f / kPI) * F3(kd.x, kd.y, kd.z);
        const float G = geometry_term(sp, sn, hp, hn);
        const float V = occluded ? 0.0f : 1.0f;
        const f3 radiance = brdf * G * V * F3(rq.x, rq.y, rq.z) * q0.w;
        if (P.accumulate)
        {
            const float4 a = accum[li];
            accum[li] = make_float4(a.x + radiance.x, a.y + radiance.y, a.z + radiance.z, a.w + 1.0f);
        }
        else { accum[li] = make_float4(radiance.x, radiance.y, radiance.z, 1.0f); }
    };
    occluded_stream<TRACE_BLOCK>(S.wide, s_stack, next_job, fetch, finish);
}

#endif /* RT_EXPERIMENTS */
/* ------------------------------------------------------- configs #2 / #3: path tracers */
/* shadow rays of 08_nee / 09_ris through the work-sharing walk (bvh.h occluded_ws) */
#ifndef RT_PT_WS
#define RT_PT_WS 1
#endif
/* common/core.hpp:76-89 with portable cos/sin [parity] */
RT_DEV f3 sample_hemisphere(float r0, float r1, float r2)
{
    const float theta = r0 * 2.0f * kPI;
    float radius = r1 + r2;
    if (1.0f < radius) radius = 2.0f - radius;
    float sn_t, cs_t;
    pm_sincosf(theta, &sn_t, &cs_t);
    const float x = cs_t * radius;
    const float z = sn_t * radius;
    const float a = 1.0f - radius * radius;
    const float y = sqrtf((a < 0.0f) ? 0.0f : a);
    return F3(x, y, z);
}

/* One path of examples/07_pt/07_pt.cu:11-90 (EXAMPLE 7) / examples/09_ris/09_ris.cu:11-166 (EXAMPLE 9). */
struct PathState
{
    f3 ro, rd, throughput, radiance;
    PCG rng;
};
RT_DEV void path_begin(const FrameParams& P, int x, int yi, PathState& st)
{
    st.rng = pcg_init(hashPCG3((uint32_t)x, (uint32_t)yi, (uint32_t)P.frame), 0);
    const float u = (float)x / (float)P.W, v = (float)yi / (float)P.H;
    const f3 forward = normalize(cross(P.rg_up, P.rg_right));
    const f3 to = P.rg_origin + forward + mix(-P.rg_right, P.rg_right, u) + mix(P.rg_up, -P.rg_up, v);
    st.ro = P.rg_origin;
    st.rd = normalize(to - P.rg_origin);
    st.radiance = F3(0.0f, 0.0f, 0.0f);
    st.throughput = F3(1.0f, 1.0f, 1.0f);
}
/* register budgets of the path tracers (A/B at 1080p): 08_nee at 6 wavefronts per SIMD -6.5 %, 09_ris at 5 -2 % */
/* one iteration of the depth loop (07_pt.cu:39-79 / 08_nee.cu:39-118 / 09_ris.cu:39-155); false = the path ended */
template <int EXAMPLE, bool SHADOWED, int STRIDE = TRACE_BLOCK>
RT_DEV bool path_bounce(const SceneView& S, uint32_t* s_stack, const FrameParams& P, int depth, f3 sky, PathState& st,
                        unsigned long long& nrays)
{
    Hit h;
    ++nrays;
    if (!trace_wide<false, false, STRIDE>(S.wide, s_stack, st.ro, st.rd, 0.0f, kFltMax, h, nullptr, RT_BARY_TV(S)))
    {
        if (EXAMPLE == 7) st.radiance = st.radiance + st.throughput * sky;
        return false;
    }
    const float4 kd4 = S.trimat[2 * (size_t)h.prim];
    if (as_uint(kd4.w) != 0u)
    {
        const float4 ke4 = S.trimat[2 * (size_t)h.prim + 1];
        if (EXAMPLE == 7 || depth == 0) st.radiance = st.radiance + st.throughput * F3(ke4.x, ke4.y, ke4.z);
        return false;
    }
    /* common/core.hpp:152-165 */
    f3 v0, v1, v2;
    load_tri(S.bvh.tv, h.prim, v0, v1, v2);
    const f3 sp = st.ro + h.t * st.rd;
    f3 sn = tri_normal(v0, v1, v2);
    if (dot(-st.rd, sn) < 0.0f) sn = -sn;
    const f3 kd = F3(kd4.x, kd4.y, kd4.z);

    if (EXAMPLE == 8)
    {
        /* next-event estimation (08_nee.cu:66-91): one uniformly picked light sample, one shadow ray */
        const float rv0 = st.rng.uniformf();
        float bx = st.rng.uniformf();
        float by = st.rng.uniformf();
        uint32_t nth = (uint32_t)(rv0 * (float)(size_t)P.n_lights);
        if (nth == (uint32_t)P.n_lights) nth = (uint32_t)P.n_lights - 1u;
        const float4* L = S.lights + RT_LIGHT_STRIDE * (size_t)nth;
        const float4 L0 = L[0], L1 = L[1], L2 = L[2];
        const f3 a0 = F3(L0.x, L0.y, L0.z), a1 = F3(L0.w, L1.x, L1.y), a2 = F3(L1.z, L1.w, L2.x);
        warp_unit_triangle(bx, by);
        const f3 lp = (1.0f - bx - by) * a0 + bx * a1 + by * a2;
        const f3 ln = tri_normal(a0, a1, a2);
        const float V = check_visibility_wide<STRIDE, RT_PT_WS>(S.wide, s_stack, sp, sn, lp, true, S.bvh.tv, h.prim) ? 1.0f : 0.0f;
        ++nrays;
        const f3 brdf = (1.0f / kPI) * kd;
        const float G = geometry_term(sp, sn, lp, ln);
        const float4 ke = S.light_ke[nth];
        st.radiance = st.radiance + st.throughput * brdf * G * V * F3(ke.x, ke.y, ke.z) / L2.z; /* :89-90 */
    }
    if (EXAMPLE == 9 && SHADOWED)
    {
        /* RIS with the shadowed target function (09_ris.cu:61-99 with options.use_shadowed_target_function):
         * one shadow ray per candidate. The rays do not depend on the reservoir chain, so the candidates are taken
         * eight at a time: draw their random numbers (same order: rv0, rv1, rv2, u per candidate), walk the eight
         * rays back to back in one traversal loop (occluded_batch: a lane starts its next ray when its current one
         * is settled), then run the reservoir updates with the visibilities. A candidate whose unshadowed weight
         * is 0 is not walked (V cannot matter); the final contribution's ray (:110-113) and its p-hat ray
         * (:116-120) repeat the selected candidate's ray. Same results, same reference ray count. */
        const float fL = (float)(size_t)P.n_lights;
        Res r = res_zero();
        float V_sel = 1.0f;
        bool have_sel = false;
        for (int i0 = 0; i0 < P.ris_sample_count; i0 += 8)
        {
            f3 tgt[8];
            uint32_t nthv[8];
            float uuv[8];
            uint32_t need = 0u, live = 0u;
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                tgt[j] = F3(0.0f, 0.0f, 0.0f); nthv[j] = 0u; uuv[j] = 0.0f;
                if (i0 + j < P.ris_sample_count)
                {
                    const float rv0 = st.rng.uniformf();
                    float bx = st.rng.uniformf();
                    float by = st.rng.uniformf();
                    uint32_t nth = (uint32_t)(rv0 * fL);
                    if (nth == (uint32_t)P.n_lights) nth = (uint32_t)P.n_lights - 1u;
                    const float4* L = S.lights + RT_LIGHT_STRIDE * (size_t)nth;
                    const float4 L0 = L[0], L1 = L[1], L2 = L[2];
                    const f3 a0 = F3(L0.x, L0.y, L0.z), a1 = F3(L0.w, L1.x, L1.y), a2 = F3(L1.z, L1.w, L2.x);
                    warp_unit_triangle(bx, by);
                    const f3 lp = (1.0f - bx - by) * a0 + bx * a1 + by * a2;
                    uuv[j] = st.rng.uniformf();
                    tgt[j] = lp; nthv[j] = nth;
                    live |= 1u << j;
                    ++nrays; /* the reference traces this ray (common/reservoir.hpp:52-57) */
                    const f3 ln = tri_normal(a0, a1, a2);
                    if (target_unshadowed(sp, sn, lp, ln, L2.y) != 0.0f) need |= 1u << j; /* weight 0 whatever V says otherwise */
                }
            }
            const uint32_t occl = occluded_batch<8, STRIDE>(S.wide, s_stack, sp, sn, tgt, need, S.bvh.tv, h.prim);
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                if (live & (1u << j))
                {
                    const float4* L = S.lights + RT_LIGHT_STRIDE * (size_t)nthv[j];
                    const float4 L0 = L[0], L1 = L[1], L2 = L[2];
                    const f3 a0 = F3(L0.x, L0.y, L0.z), a1 = F3(L0.w, L1.x, L1.y), a2 = F3(L1.z, L1.w, L2.x);
                    const f3 ln = tri_normal(a0, a1, a2);
                    const float V = (occl >> j) & 1u ? 0.0f : 1.0f;
                    const float p_hat = target_shadowed(sp, sn, tgt[j], ln, L2.y, V);
                    const float weight = p_hat / L2.z;
                    r.w_sum += weight;
                    r.M += 1;
                    if (reservoir_accept(uuv[j], weight, r.w_sum))
                    {
                        r.hit_p = tgt[j]; r.hit_n = ln; r.lum = L2.y;
                        const float4 ke = S.light_ke[nthv[j]];
                        r.rad = F3(ke.x, ke.y, ke.z);
                        V_sel = V; have_sel = true;
                    }
                }
            }
        }
        const f3 brdf = (1.0f / kPI) * kd;
        const float G = geometry_term(sp, sn, r.hit_p, r.hit_n);
        /* no candidate selected (all weights 0): the reference still walks surface -> Reservoir{}'s zero position */
        const float V = have_sel ? V_sel : (check_visibility_wide<STRIDE, RT_PT_WS>(S.wide, s_stack, sp, sn, r.hit_p, true, S.bvh.tv, h.prim) ? 1.0f : 0.0f);
        nrays += 2; /* :110-113 and :116-120 */
        const float p_hat = target_shadowed(sp, sn, r.hit_p, r.hit_n, r.lum, V);
        const float ucw = p_hat > 0.0f ? r.w_sum / ((float)r.M * p_hat) : 0.0f;
        st.radiance = st.radiance + st.throughput * brdf * G * V * r.rad * ucw;
    }
    if (EXAMPLE == 9 && !SHADOWED)
    {
        /* RIS over the lights (09_ris.cu:61-99), then the shaded contribution (:101-126) */
        const float fL = (float)(size_t)P.n_lights;
        Res r = res_zero();
        for (int i = 0; i < P.ris_sample_count; ++i)
        {
            const float rv0 = st.rng.uniformf();
            float bx = st.rng.uniformf();
            float by = st.rng.uniformf();
            uint32_t nth = (uint32_t)(rv0 * fL);
            if (nth == (uint32_t)P.n_lights) nth = (uint32_t)P.n_lights - 1u;
            const float4* L = S.lights + RT_LIGHT_STRIDE * (size_t)nth;
            const float4 L0 = L[0], L1 = L[1], L2 = L[2];
            const f3 a0 = F3(L0.x, L0.y, L0.z), a1 = F3(L0.w, L1.x, L1.y), a2 = F3(L1.z, L1.w, L2.x);
            warp_unit_triangle(bx, by);
            const f3 lp = (1.0f - bx - by) * a0 + bx * a1 + by * a2;
            const f3 ln = tri_normal(a0, a1, a2);
            const float p_hat = target_unshadowed(sp, sn, lp, ln, L2.y);
            const float weight = p_hat / L2.z;
            const float uu = st.rng.uniformf();
            r.w_sum += weight;
            r.M += 1;
            if (reservoir_accept(uu, weight, r.w_sum))
            {
                r.hit_p = lp; r.hit_n = ln; r.lum = L2.y;
                const float4 ke = S.light_ke[nth];
                r.rad = F3(ke.x, ke.y, ke.z);
            }
        }
        const f3 brdf = (1.0f / kPI) * kd;
        const float G = geometry_term(sp, sn, r.hit_p, r.hit_n);
        const float V = check_visibility_wide<STRIDE, RT_PT_WS>(S.wide, s_stack, sp, sn, r.hit_p, true, S.bvh.tv, h.prim) ? 1.0f : 0.0f;
        ++nrays;
        const float p_hat = target_unshadowed(sp, sn, r.hit_p, r.hit_n, r.lum);
        const float ucw = p_hat > 0.0f ? r.w_sum / ((float)r.M * p_hat) : 0.0f;
        st.radiance = st.radiance + st.throughput * brdf * G * V * r.rad * ucw;
    }
    /* next direction (07_pt.cu:61-70 / 09_ris.cu:128-137): common/core.hpp:216-235 */
    const float r0 = st.rng.uniformf();
    const float r1 = st.rng.uniformf();
    const float r2 = st.rng.uniformf();
    const f3 wl = sample_hemisphere(r0, r1, r2);
    const f3 tg = normalize(v1 - v0);
    const f3 bt = normalize(cross(tg, sn));
    const f3 wo = wl.x * tg + wl.y * sn + wl.z * bt;
    st.throughput = st.throughput * kd;
    st.ro = sp + 0.001f * sn;
    st.rd = wo;
    return true;
}
RT_DEV void path_write(const FrameParams& P, float4* __restrict__ accum, size_t li, f3 radiance)
{
    if (P.accumulate)
    {
        const float4 a = accum[li];
        accum[li] = make_float4(a.x + radiance.x, a.y + radiance.y, a.z + radiance.z, a.w + 1.0f);
    }
    else { accum[li] = make_float4(radiance.x, radiance.y, radiance.z, 1.0f); }
}
RT_DEV void count_rays(unsigned long long nrays, unsigned long long* __restrict__ rays)
{
    for (int off = 32; off > 0; off >>= 1) nrays += __shfl_down(nrays, off);
    if ((threadIdx.x & 63) == 0 && nrays) atomicAdd(rays, nrays);
}

/* the reference's shape: one thread per pixel, whole path in one launch. rays[0] accumulates the
 * number of raytrace() calls (one atomic per wave). */
template <int EXAMPLE, bool SHADOWED>
__global__ __launch_bounds__(EXAMPLE == 7 ? BLOCK : TRACE_BLOCK, EXAMPLE == 7 ? 1 : (EXAMPLE == 8 ? 6 : 5)) void k_path_trace(SceneView S, FrameParams P, int max_depth, f3 sky,
                                                       float4* __restrict__ accum,
                                                       unsigned long long* __restrict__ rays)
{
    /* 07_pt (one ray per bounce, no light sampling) is 3-8 % faster on 256-thread groups, 08/09 on one-wavefront groups */
    constexpr int TB = EXAMPLE == 7 ? BLOCK : TRACE_BLOCK;
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[(RT_PT_WS ? WIDE_LDS_ROWS : WIDE_LDS_STACK) * TB];
    int x, row;
    const bool ok = tile_pixel<TB>(P, x, row);
    unsigned long long nrays = 0;
    if (ok)
    {
        PathState st;
        path_begin(P, x, P.H - 1 - row, st);
        for (int depth = 0; depth < max_depth; ++depth)
            if (!path_bounce<EXAMPLE, SHADOWED, TB>(S, s_stack, P, depth, sky, st, nrays)) break;
        path_write(P, accum, (size_t)x + (size_t)(row - P.lrow0) * P.W, st.radiance);
    }
    count_rays(nrays, rays);
}

/* Wavefront form of the same paths: one launch per bounce over the list of LIVE paths; survivors
 * are appended to the next list by wave ballot + one aggregated atomic (compaction), so a wave of
 * bounce d+1 is full again instead of carrying the lanes whose paths ended. 64-B path records:
 *   {ro.xyz, rd.x} {rd.yz, thr.xy} {thr.z, rad.xyz} {rng.state lo/hi, bits(pixel), 0}
 * counters[0] = rays, counters[2 + d] = number of paths alive when bounce d starts. */
RT_DEV void path_store(float4* __restrict__ rec, size_t i, const PathState& st, uint32_t pixel)
{
    rec[4 * i + 0] = make_float4(st.ro.x, st.ro.y, st.ro.z, st.rd.x);
    rec[4 * i + 1] = make_float4(st.rd.y, st.rd.z, st.throughput.x, st.throughput.y);
    rec[4 * i + 2] = make_float4(st.throughput.z, st.radiance.x, st.radiance.y, st.radiance.z);
    rec[4 * i + 3] = make_float4(as_float((uint32_t)(st.rng.state & 0xffffffffull)), as_float((uint32_t)(st.rng.state >> 32)),
                                 as_float(pixel), 0.0f);
}
RT_DEV uint32_t path_load(const float4* __restrict__ rec, size_t i, PathState& st)
{
    const float4 a = rec[4 * i + 0], b = rec[4 * i + 1], c = rec[4 * i + 2], d = rec[4 * i + 3];
    st.ro = F3(a.x, a.y, a.z); st.rd = F3(a.w, b.x, b.y);
    st.throughput = F3(b.z, b.w, c.x); st.radiance = F3(c.y, c.z, c.w);
    st.rng.state = (uint64_t)as_uint(d.x) | ((uint64_t)as_uint(d.y) << 32);
    st.rng.inc = 1u; /* sequence 0 */
    return as_uint(d.z);
}
__global__ __launch_bounds__(TRACE_BLOCK) void k_pt_init(FrameParams P, float4* __restrict__ list, unsigned long long* __restrict__ counters)
{
    int x, row;
    const bool ok = tile_pixel<TRACE_BLOCK>(P, x, row);
    /* every owned pixel starts one path; list position = compacted order of this launch */
    const unsigned long long m = __ballot(ok);
    if (!m) return;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(&counters[2], (unsigned long long)__popcll(m));
    base = __shfl(base, leader);
    if (!ok) return;
    PathState st;
    path_begin(P, x, P.H - 1 - row, st);
    const uint32_t li = (uint32_t)((size_t)x + (size_t)(row - P.lrow0) * P.W);
    path_store(list, (size_t)(base + __popcll(m & ((1ull << lane) - 1ull))), st, li);
}
template <int EXAMPLE, bool SHADOWED>
__global__ __launch_bounds__(TRACE_BLOCK, EXAMPLE == 7 ? 1 : (EXAMPLE == 8 ? 6 : 5)) void k_pt_bounce(SceneView S, FrameParams P, int depth, int max_depth, f3 sky,
                                                      const float4* __restrict__ in, float4* __restrict__ out,
                                                      float4* __restrict__ accum, unsigned long long* __restrict__ counters)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[(RT_PT_WS ? WIDE_LDS_ROWS : WIDE_LDS_STACK) * TRACE_BLOCK];
    const size_t i = (size_t)blockIdx.x * TRACE_BLOCK + threadIdx.x;
    const bool have = i < counters[2 + depth];
    unsigned long long nrays = 0;
    bool alive = false;
    PathState st;
    uint32_t pixel = 0;
    if (have)
    {
        pixel = path_load(in, i, st);
        alive = path_bounce<EXAMPLE, SHADOWED>(S, s_stack, P, depth, sky, st, nrays);
        if (alive && depth + 1 >= max_depth) alive = false; /* the depth loop ends here */
        if (!alive) path_write(P, accum, pixel, st.radiance);
    }
    const unsigned long long m = __ballot(alive);
    if (m)
    {
        const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
        unsigned long long base = 0;
        if (lane == leader) base = atomicAdd(&counters[3 + depth], (unsigned long long)__popcll(m));
        base = __shfl(base, leader);
        if (alive) path_store(out, (size_t)(base + __popcll(m & ((1ull << lane) - 1ull))), st, pixel);
    }
    count_rays(nrays, &counters[0]);
}

/* --------------------------------------------------------- clear / tone_mapping */
/* common/kernels/common.cu:4-17 */
__global__ __launch_bounds__(BLOCK) void k_clear(FrameParams P, float4* __restrict__ accum)
{
    int x, row;
    if (!tile_pixel(P, x, row)) return;
    accum[(size_t)x + (size_t)(row - P.lrow0) * P.W] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
/* common/kernels/common.cu:30-74 */
__global__ __launch_bounds__(BLOCK) void k_tone_mapping(FrameParams P, const float4* __restrict__ accum,
                                                         uint32_t* __restrict__ pixels)
{
    int x, row;
    if (!tile_pixel(P, x, row)) return;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
    pixels[li] = tone_map_rgba8(accum[li]);
}

/* ------------------------------------------------ layout conversion (upload/download) */
__global__ void k_res_to_ref(int n, const float4* __restrict__ rec, const float4* __restrict__ radb,
                             uint32_t* __restrict__ out /* 19 words per pixel */)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool shaded;
    Res r = res_load(rec, (size_t)i, shaded);
    const float4 rq = radb[i];
    uint32_t* o = out + 19 * (size_t)i;
    const float f[15] = {r.org_p.x, r.org_p.y, r.org_p.z, r.org_n.x, r.org_n.y, r.org_n.z, r.hit_p.x, r.hit_p.y,
                         r.hit_p.z, r.hit_n.x, r.hit_n.y, r.hit_n.z, rq.x, rq.y, rq.z};
    for (int k = 0; k < 15; ++k) o[k] = as_uint(f[k]);
    o[15] = r.vis ? 1u : 0u;
    o[16] = as_uint(r.w_sum);
    o[17] = as_uint(r.ucw);
    o[18] = (uint32_t)r.M;
}
__global__ void k_res_from_ref(int n, const uint32_t* __restrict__ in, const float4* __restrict__ g1,
                               float4* __restrict__ rec, float4* __restrict__ radb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* s = in + 19 * (size_t)i;
    Res r;
    r.org_p = F3(as_float(s[0]), as_float(s[1]), as_float(s[2]));
    r.org_n = F3(as_float(s[3]), as_float(s[4]), as_float(s[5]));
    r.hit_p = F3(as_float(s[6]), as_float(s[7]), as_float(s[8]));
    r.hit_n = F3(as_float(s[9]), as_float(s[10]), as_float(s[11]));
    r.rad = F3(as_float(s[12]), as_float(s[13]), as_float(s[14]));
    r.vis = (s[15] & 0xffu) != 0u;
    r.w_sum = as_float(s[16]);
    r.ucw = as_float(s[17]);
    r.M = (int)s[18];
    r.ownv = 0u; /* an uploaded reservoir: nothing is known about its sample's visibility from this pixel */
    r.lum = luminance(r.rad);
    const bool shaded = (as_uint(g1[i].w) & GB_SHADED) != 0u;
    res_store(rec, radb, (size_t)i, r, shaded);
}

/* The "sky / emissive neighbour" test of spatial_resampling (10_restir_di.cu:326-338) reads the CURRENT Visibility buffer; the
 * record keeps that bit (RES_SHADED_BIT) as of the G-buffer it was written under. The per-kernel entry point
 * rt_spatial_resampling re-derives the bits of its input buffer when the G-buffer has changed since (a camera move or a
 * visibility upload between generate_candidate and the pass); the staged frame never needs it. */
__global__ void k_refresh_shaded(int n, const float4* __restrict__ g1, float4* __restrict__ rec)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t* w = reinterpret_cast<uint32_t*>(rec) + 16 * (size_t)i + 7; /* q1.w = M | vis << 31 | shaded << 30 */
    const uint32_t v = *w, want = (v & ~RES_SHADED_BIT) | ((as_uint(g1[i].w) & GB_SHADED) ? RES_SHADED_BIT : 0u);
    if (v != want) *w = want;
}

/* ------------------------------------------------------------- scene tables */
/* per emissive triangle (index order, 10_restir_di.cpp:196-205), RT_LIGHT_STRIDE x float4:
 *   {v0.xyz, v1.x} {v1.yz, v2.xy} {v2.z, luminance(Ke), pdf, bits(tri)}   + a side table {Ke.xyz, 0}
 * pdf = 1/L * 1/area_of — the exact expression of common/core.hpp:57-62 and
 * 10_restir_di.cu:98-99, evaluated once instead of once per candidate. */
__global__ void k_light_table(int n_lights, const uint32_t* __restrict__ light_ids, const float* __restrict__ tris,
                              float4* __restrict__ lights, float4* __restrict__ light_ke)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lights) return;
    const int ti = (int)light_ids[i];
    const float* t = tris + 15 * (size_t)ti;
    const f3 v0 = F3(t[0], t[1], t[2]), v1 = F3(t[3], t[4], t[5]), v2 = F3(t[6], t[7], t[8]);
    const f3 ke = F3(t[12], t[13], t[14]);
    const float pdf = 1.0f / (float)(size_t)n_lights * 1.0f / tri_area(v0, v1, v2);
    float4* L = lights + RT_LIGHT_STRIDE * (size_t)i;
    L[0] = make_float4(v0.x, v0.y, v0.z, v1.x);
    L[1] = make_float4(v1.y, v1.z, v2.x, v2.y);
    L[2] = make_float4(v2.z, luminance(ke), pdf, as_float(ti));
#if RT_LIGHT_STRIDE == 4
    const f3 nn = tri_normal(v0, v1, v2);
    /* .w: the refined reciprocal of the pdf for div_by (rt_device.h), NaN if the pdf is outside the range it is exact for */
    L[3] = make_float4(nn.x, nn.y, nn.z, div_den_ok(pdf) ? rcp_refined(pdf) : as_float(0x7fc00000u));
#endif
    light_ke[i] = make_float4(ke.x, ke.y, ke.z, 0.0f);
}
__global__ void k_trimat(int n, const float* __restrict__ tris, float4* __restrict__ trimat)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* t = tris + 15 * (size_t)i;
    /* has_emission, common/core.hpp:64-68 */
    const bool e = t[12] > 0.0f || t[13] > 0.0f || t[14] > 0.0f;
    trimat[2 * (size_t)i] = make_float4(t[9], t[10], t[11], as_float(e ? 1u : 0u));
    trimat[2 * (size_t)i + 1] = make_float4(t[12], t[13], t[14], 0.0f);
}

/* ------------------------------------------------------------------ utilities */
__global__ void k_count_shaded(FrameParams P, const float4* __restrict__ g1, unsigned long long* __restrict__ out)
{
    int x, row;
    const bool ok = tile_pixel(P, x, row);
    bool shaded = false;
    if (ok) shaded = (as_uint(g1[(size_t)x + (size_t)(row - P.lrow0) * P.W].w) & GB_SHADED) != 0u;
    const unsigned long long m = __ballot(shaded);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, (unsigned long long)__popcll(m));
}
/* shaded pixels per storage row (one workgroup per row): the cost model of the strip partition */
__global__ __launch_bounds__(BLOCK) void k_row_shaded(int W, int row0_local, const float4* __restrict__ g1, uint32_t* __restrict__ out)
{
    const size_t base = (size_t)(row0_local + (int)blockIdx.x) * W;
    uint32_t n = 0;
    for (int x = threadIdx.x; x < W; x += BLOCK) n += (as_uint(g1[base + x].w) & GB_SHADED) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(&out[blockIdx.x], n);
}
template <int MODE, bool ANY = false> /* 0 = wide (production), 1 = binary stackless; ANY = shadow-ray semantics */
__global__ __launch_bounds__(BLOCK) void k_trace_closest(SceneView S, const float* __restrict__ rays, int n, float* __restrict__ hits)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[MODE == 0 ? WIDE_LDS_WORDS : 4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + 8 * (size_t)i;
    Hit h;
    h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.prim = -1;
    if (MODE == 0) trace_wide<ANY>(S.wide, s_stack, F3(r[0], r[1], r[2]), F3(r[3], r[4], r[5]), r[6], r[7], h);
    else trace<ANY>(S.bvh, F3(r[0], r[1], r[2]), F3(r[3], r[4], r[5]), r[6], r[7], h);
    float* o = hits + 4 * (size_t)i;
    o[0] = h.t; o[1] = h.u; o[2] = h.v; o[3] = as_float(h.prim);
}
/* statistics of the work-sharing shadow-ray walk: one-wavefront workgroups as in the frame kernels;
 * stats[2i] = passes the ray's wavefront ran | leaf passes << 15 | this lane's triangle tests << 23 | occluded << 31,
 * stats[2i+1] = steals by this lane | inner records it visited << 16 */
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace_stats_ws(SceneView S, const float* __restrict__ rays, int n, uint32_t* __restrict__ stats)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[WIDE_LDS_ROWS * TRACE_BLOCK];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + 8 * (size_t)i;
    uint32_t st[4] = {0u, 0u, 0u, 0u};
    bool occ = false;
    if (r[7] >= 0.0f) occ = occluded_ws<TRACE_BLOCK>(S.wide, s_stack, F3(r[0], r[1], r[2]), F3(r[3], r[4], r[5]), r[6], r[7], st);
    /* [0]: passes (15 bits) | leaf passes << 15 (8 bits) | triangle tests by this lane << 23 (8 bits) | occluded << 31 */
    stats[2 * (size_t)i] = (st[0] & 0x7fffu) | ((st[2] > 255u ? 255u : st[2]) << 15) | ((st[3] > 255u ? 255u : st[3]) << 23) | (occ ? 0x80000000u : 0u);
    stats[2 * (size_t)i + 1] = st[1];
}
/* shadow rays from a list, walked as the frame kernels walk theirs (one-wavefront workgroups): hits[i].w = 0 if
 * occluded, -1 if not; rays with tmax < 0 are lanes without a ray. WS: the work-sharing walk, else one lane one ray. */
template <bool WS>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace_anyhit(SceneView S, const float* __restrict__ rays, int n, float* __restrict__ hits)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[(WS ? WIDE_LDS_ROWS : WIDE_LDS_STACK) * TRACE_BLOCK];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + 8 * (size_t)i;
    bool occ = false;
    if (r[7] >= 0.0f)
    {
        if (WS) occ = occluded_ws<TRACE_BLOCK>(S.wide, s_stack, F3(r[0], r[1], r[2]), F3(r[3], r[4], r[5]), r[6], r[7]);
        else { Hit h; occ = trace_wide<true, false, TRACE_BLOCK>(S.wide, s_stack, F3(r[0], r[1], r[2]), F3(r[3], r[4], r[5]), r[6], r[7], h); }
    }
    float4* out = (float4*)hits;
    out[i] = make_float4(0.0f, 0.0f, 0.0f, as_float(occ ? 0 : -1));
}
template <int MODE, bool ANY = false>
__global__ __launch_bounds__(BLOCK) void k_trace_stats(SceneView S, const float* __restrict__ rays, int n, uint32_t* __restrict__ stats)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_stack[MODE == 0 ? WIDE_LDS_WORDS : 4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + 8 * (size_t)i;
    Hit h;
    uint32_t st[2] = {0u, 0u};
    if (MODE == 0) trace_wide<ANY, true>(S.wide, s_stack, F3(r[0], r[1], r[2]), F3(r[3], r[4], r[5]), r[6], r[7], h, st);
    else trace<false, true>(S.bvh, F3(r[0], r[1], r[2]), F3(r[3], r[4], r[5]), r[6], r[7], h, st);
    stats[2 * (size_t)i] = st[0];
    stats[2 * (size_t)i + 1] = st[1];
}

#ifdef RT_EXPERIMENTS /* rt_trace_* mode: persistent-wavefront queue (r01), A/B only */
/* Persistent wavefront tracing over a ray queue with LANE refill: a lane whose ray is finished
 * pulls the next ray index while the other lanes keep traversing (ballot of idle lanes -> one
 * aggregated atomic -> prefix popcount), instead of idling until the slowest lane of its wave is
 * done. Refill is attempted when at least REFILL_MIN lanes are idle. ANY = shadow rays
 * (hits[i].x = 1 if occluded). Same traversal steps / results as trace_wide. */
template <bool ANY>
__global__ __launch_bounds__(BLOCK) void k_trace_queue(WideView wide, const float* __restrict__ rays, int n,
                                                        float* __restrict__ hits, unsigned int* __restrict__ head)
{
    __shared__ uint32_t s_stack[WIDE_LDS_STACK * BLOCK];
    constexpr int REFILL_MIN = 20;
    constexpr uint32_t IDLE = 0x7ffffffeu;
    const int lane = threadIdx.x & 63;
    const int slot = threadIdx.x;
    uint32_t ovf[WIDE_OVF_STACK];
    /* per-lane ray state */
    uint32_t cur = IDLE;
    int sp = 0, prim = -1, ray_id = -1;
    f3 ro = F3(0, 0, 0), rd = F3(0, 0, 1), inv = F3(0, 0, 1);
    float tmin = 0.0f, tmax = 0.0f, best = 0.0f, bu = 0.0f, bv = 0.0f;
    bool exhausted = false;
    for (;;)
    {
        const unsigned long long idle = __ballot(cur == IDLE);
        if (idle)
        {
            const int n_idle = __popcll(idle);
            if (!exhausted && (n_idle >= REFILL_MIN || n_idle == 64 || true))
            {
                if (n_idle >= REFILL_MIN || n_idle == 64)
                {
                    const int leader = __ffsll((long long)idle) - 1;
                    unsigned int base = 0;
                    if (lane == leader) base = atomicAdd(head, (unsigned)n_idle);
                    base = __shfl(base, leader);
                    if (cur == IDLE)
                    {
                        const unsigned my = base + (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
                        if (my < (unsigned)n)
                        {
                            const float* r = rays + 8 * (size_t)my;
                            ray_id = (int)my;
                            ro = F3(r[0], r[1], r[2]); rd = F3(r[3], r[4], r[5]);
                            tmin = r[6]; tmax = r[7];
                            inv = F3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                            inv.x = fminf(fmaxf(inv.x, -1e30f), 1e30f);
                            inv.y = fminf(fmaxf(inv.y, -1e30f), 1e30f);
                            inv.z = fminf(fmaxf(inv.z, -1e30f), 1e30f);
                            best = tmax; prim = -1; bu = 0.0f; bv = 0.0f; sp = 0;
                            cur = 0u;
                        }
                    }
                    if (base + (unsigned)n_idle >= (unsigned)n) exhausted = true;
                }
            }
            if (exhausted && __ballot(cur != IDLE) == 0ull) return;
        }
        /* no `continue` for idle lanes: they must fall through to the loop header together with
         * the working lanes (a spinning divergent path would starve the others) */
        bool done = false;
        const float4* r = wide.rec + WIDE_STRIDE * (size_t)(cur & ~WIDE_LEAF_BIT);
        if (cur == IDLE) {}
        else if (cur & WIDE_LEAF_BIT)
        {
            const float4 t0 = r[0], t1 = r[1], t2 = r[2];
            const f3 v0 = F3(t0.x, t0.y, t0.z), v1 = F3(t0.w, t1.x, t1.y), v2 = F3(t1.z, t1.w, t2.x);
            const int pi = as_int(t2.y);
            float t, u, v;
            if (intersect_ray_triangle(t, u, v, ro, rd, tmin, tmax, v0, v1, v2))
            {
                if (prim < 0 || t < best || (t == best && pi > prim))
                {
                    best = t; bu = u; bv = v; prim = pi;
                    if (ANY) done = true;
                }
            }
            if (!done)
            {
                if (sp == 0) done = true;
                else { --sp; cur = sp < WIDE_LDS_STACK ? s_stack[sp * BLOCK + slot] : ovf[sp - WIDE_LDS_STACK]; }
            }
        }
        else
        {
            const float4 q0 = r[0], q1f = r[1], q2f = r[2];
            const uint32_t e = as_uint(q0.w);
            const uint32_t base = as_uint(q1f.x), meta = as_uint(q1f.y);
            const uint32_t lx = as_uint(q1f.z), ly = as_uint(q1f.w), lz = as_uint(q2f.x);
            const uint32_t hx = as_uint(q2f.y), hy = as_uint(q2f.z), hz = as_uint(q2f.w);
            const float sx = as_float((e & 0xffu) << 23), sy = as_float(((e >> 8) & 0xffu) << 23),
                        sz = as_float(((e >> 16) & 0xffu) << 23);
            const float Ax = (q0.x - ro.x) * inv.x, Ay = (q0.y - ro.y) * inv.y, Az = (q0.z - ro.z) * inv.z;
            const float Bx = sx * inv.x, By = sy * inv.y, Bz = sz * inv.z;
            float td[4];
            uint32_t ce[4];
            int nhit = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const uint32_t m = (meta >> (8 * k)) & 0xffu;
                const float x0 = __builtin_fmaf(wide_byte(lx, k), Bx, Ax), x1 = __builtin_fmaf(wide_byte(hx, k), Bx, Ax);
                const float y0 = __builtin_fmaf(wide_byte(ly, k), By, Ay), y1 = __builtin_fmaf(wide_byte(hy, k), By, Ay);
                const float z0 = __builtin_fmaf(wide_byte(lz, k), Bz, Az), z1 = __builtin_fmaf(wide_byte(hz, k), Bz, Az);
                float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
                float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
                /* conservative: rounding may leave tn a few 1e-7 too large and tf too small; one factor on the far side covers both */
                tn = fmaxf(tn, tmin);
                tf = fminf(tf, best) * WIDE_SLAB_PAD;
                const bool h = (m != 0u) && (tn <= tf);
                td[k] = h ? tn : 3.0e38f;
                ce[k] = (base + (uint32_t)k) | (m == 2u ? WIDE_LEAF_BIT : 0u);
                nhit += h ? 1 : 0;
            }
            if (nhit > 0)
            {
#define RT_CSWAP(i, j)                                                     \
    if (td[j] < td[i])                                                     \
    {                                                                      \
        const float _t = td[i]; td[i] = td[j]; td[j] = _t;                 \
        const uint32_t _e = ce[i]; ce[i] = ce[j]; ce[j] = _e;             \
    }
                RT_CSWAP(0, 1) RT_CSWAP(2, 3) RT_CSWAP(0, 2) RT_CSWAP(1, 3) RT_CSWAP(1, 2)
#undef RT_CSWAP
                for (int k = nhit - 1; k >= 1; --k)
                {
                    if (sp < WIDE_LDS_STACK) s_stack[sp * BLOCK + slot] = ce[k];
                    else ovf[sp - WIDE_LDS_STACK] = ce[k];
                    ++sp;
                }
                cur = ce[0];
            }
            else
            {
                if (sp == 0) done = true;
                else { --sp; cur = sp < WIDE_LDS_STACK ? s_stack[sp * BLOCK + slot] : ovf[sp - WIDE_LDS_STACK]; }
            }
        }
        if (done)
        {
            float* o = hits + 4 * (size_t)ray_id;
            if (ANY) { o[0] = prim >= 0 ? 1.0f : 0.0f; o[1] = 0.0f; o[2] = 0.0f; o[3] = as_float(prim); }
            else { o[0] = prim >= 0 ? best : 0.0f; o[1] = bu; o[2] = bv; o[3] = as_float(prim); }
            cur = IDLE;
        }
    }
}

#endif /* RT_EXPERIMENTS */
__global__ void k_math_eval(int fn, const float* __restrict__ in, int n, float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0.0f;
    switch (fn)
    {
        case 20: r = pm_logf(in[i]); break;
        case 21: r = pm_cosf(in[i]); break;
        case 22: r = pm_sinf(in[i]); break;
        case 28: { float sn, cs; pm_sincosf(in[i], &sn, &cs); r = sn; break; } /* the fused form the kernels use */
        case 29: { float sn, cs; pm_sincosf(in[i], &sn, &cs); r = cs; break; }
        case 23: r = pm_expf(in[i]); break;
        case 24: r = pm_pow8f(in[i]); break;
        case 25: r = pm_powf_pos(in[i], 1.0f / 2.2f); break;
        case 26: r = in[2 * (size_t)i] / in[2 * (size_t)i + 1]; break;
        case 27: r = sqrtf(in[i]); break;
        case 30: r = as_float(f2i_sat(in[i])); break; /* the bits of the int */
        /* guarded divisions / square root against the compiler's (rt_device.h): the XOR of the two results' bits */
        case 31:
        {
            const float* q = in + 12 * (size_t)i;
            const f3 p0 = F3(q[0], q[1], q[2]), n0 = F3(q[3], q[4], q[5]), p1 = F3(q[6], q[7], q[8]), n1 = F3(q[9], q[10], q[11]);
            r = as_float(as_uint(geometry_term(p0, n0, p1, n1)) ^ as_uint(geometry_term_plain(p0, n0, p1, n1)));
            break;
        }
        case 32: /* was the fast path of geometry_term taken? (coverage of the test's inputs) */
        {
            const float* q = in + 12 * (size_t)i;
            const f3 v = F3(q[6], q[7], q[8]) - F3(q[0], q[1], q[2]);
            r = (div_den_ok(dot(v, v)) && __builtin_fminf(__builtin_fminf(fabsf(v.x), fabsf(v.y)), fabsf(v.z)) >= kDivNumLo) ? 1.0f : 0.0f;
            break;
        }
        case 33: /* n / d through the refined reciprocal wherever the range tests admit it */
        {
            const float n = in[2 * (size_t)i], d = in[2 * (size_t)i + 1];
            const bool ok = div_den_ok(fabsf(d)) && div_num_ok(fabsf(n));
            r = ok ? as_float(as_uint(div_by(n, d, rcp_refined(d))) ^ as_uint(n / d)) : as_float(0xffffffffu);
            break;
        }
        case 34: r = div_den_ok(in[i]) ? as_float(as_uint(sqrt_in_range(in[i])) ^ as_uint(sqrtf(in[i]))) : as_float(0xffffffffu); break;
        case 35: r = div_pdf(in[2 * (size_t)i], in[2 * (size_t)i + 1], div_den_ok(in[2 * (size_t)i + 1]) ? rcp_refined(in[2 * (size_t)i + 1]) : as_float(0x7fc00000u)); break;
        /* r05 device self-checks. 36: reservoir_accept(u, W, S) as 1 / 0, and +2 when the decision came from the division-free path
         * (coverage); 37: the same comparison with the plain division (what the host recomputes in IEEE binary32 too) */
        case 36:
        {
            const float u = in[3 * (size_t)i], W = in[3 * (size_t)i + 1], S = in[3 * (size_t)i + 2];
            const float t = u * S, d = W - t;
            const float margin = __builtin_fmaf(fabsf(W), 9.5367431640625e-07f, 7.52316384526264e-37f);
            r = (reservoir_accept(u, W, S) ? 1.0f : 0.0f) + ((fabsf(d) > margin && t > 0.0f) ? 2.0f : 0.0f);
            break;
        }
        case 37: r = (in[3 * (size_t)i] < in[3 * (size_t)i + 1] / in[3 * (size_t)i + 2]) ? 1.0f : 0.0f; break;
        /* 38: ris_weight against the nested-guard functions it replaces: XOR of the two results' bits; in: sp3 sn3 lp3 ln3 lum pdf (14) */
        case 38:
        {
            const float* q = in + 14 * (size_t)i;
            const f3 sp = F3(q[0], q[1], q[2]), sn = F3(q[3], q[4], q[5]), lp = F3(q[6], q[7], q[8]), ln = F3(q[9], q[10], q[11]);
            const float lum = q[12], pdf = q[13];
            const float r1 = div_den_ok(pdf) ? rcp_refined(pdf) : as_float(0x7fc00000u); /* as k_light_table stores it */
            r = as_float(as_uint(ris_weight(sp, sn, lp, ln, lum, pdf, r1)) ^ as_uint(target_unshadowed(sp, sn, lp, ln, lum) / pdf));
            break;
        }
        default: break;
    }
    out[i] = r;
}

