/*
 * restir_rt.hip — gfx950 kernels and C-ABI (include/restir_rt.h) of the ReSTIR DI hot path.
 *
 * Kernels restate, per pixel, examples/10_restir_di/10_restir_di.cu and
 * common/kernels/common.cu of the reference (file:line cited at each kernel) over the
 * MI355X-native data layout of rt_device.h (32-B G-buffer, 64-B aligned reservoir records)
 * and the software BVH of bvh.h. Compile with -ffp-contract=off (parity, rt_device.h).
 *
 * Launch shape: one thread per pixel, 256-thread workgroups covering 32 x 8 pixel tiles
 * (a wave = two 32-pixel row segments => own-pixel record traffic is 2-KiB contiguous runs).
 * Workgroup -> tile mapping is XCD-aware: workgroups are dealt round-robin over the 8 XCDs
 * (MI355X_MICROARCH "Workgroup dispatch"), so workgroup b works on tile
 * (b % 8) * ceil(T/8) + b / 8: every XCD owns one horizontal band of the image and the
 * spatial pass's neighbour gathers (sigma = 15 px) stay in that XCD's L2.
 */
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "../../include/restir_rt.h"
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

/* ------------------------------------------------------------------ params */

struct FrameParams
{
    int W, H;           /* full image */
    int row0, row1;     /* global storage rows processed by this launch */
    int lrow0, lrows;   /* global row of local buffer row 0, local rows held */
    int frame, pass;
    f3 eye;
    f3 rg_origin, rg_right, rg_up;
    int n_lights;
    /* options (common/options.hpp) */
    int accumulate, ris_sample_count, use_temporal, use_spatial, spatial_count, vis_reuse;
    float spatial_radius;
    int tile_mode; /* workgroup -> tile order inside an XCD's band: 0 row-major, 1 column-major */
};

struct SceneView
{
    BvhView bvh;   /* binary LBVH, stackless trail traversal (kept for A/B measurements) */
    WideView wide; /* production traversal structure */
    const float4* __restrict__ trimat; /* 2 per triangle: {Kd.xyz, bits(emissive?)}, {Ke.xyz, 0} */
    const float4* __restrict__ lights;   /* 3 per light, see k_light_table */
    const float4* __restrict__ light_ke; /* 1 per light: {Ke.xyz, 0} */
};

constexpr int TILE_W = 32, TILE_H = 8, BLOCK = 256;
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 1 /* min waves per SIMD requested for the tracing kernels (register budget) */
#endif

/* XCD-aware workgroup -> tile -> pixel. Returns false for threads outside the row range.
 * Workgroup b runs on XCD b % 8 (round-robin dispatch); slot b / 8 walks that XCD's band of
 * tile rows either row by row (mode 0) or column by column (mode 1: the set of tiles in flight
 * on an XCD is then ~13 tiles wide x the band height instead of full-width x 4 rows, which is
 * what keeps the spatial pass's neighbour window inside the 4 MiB L2). */
RT_DEV bool tile_pixel(const FrameParams& P, int& x, int& row)
{
    const int tiles_x = (P.W + TILE_W - 1) / TILE_W;
    const int tiles_y = (P.row1 - P.row0 + TILE_H - 1) / TILE_H;
    const int b = blockIdx.x;
    int tx, ty;
    if (P.tile_mode == 1)
    {
        const int band_rows = (tiles_y + 7) / 8;
        const int slot = b >> 3;
        tx = slot / band_rows;
        ty = (b & 7) * band_rows + (slot - tx * band_rows);
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
    x = tx * TILE_W + (threadIdx.x & (TILE_W - 1));
    row = P.row0 + ty * TILE_H + (threadIdx.x >> 5);
    return x < P.W && row < P.row1;
}
static inline int tile_grid(int W, int rows)
{
    /* covers both orders: mode 1 needs 8 * ceil(tiles_y/8) * tiles_x workgroups */
    const int tx = (W + TILE_W - 1) / TILE_W, ty = (rows + TILE_H - 1) / TILE_H;
    const int a = ((tx * ty + 7) / 8) * 8, b = 8 * ((ty + 7) / 8) * tx;
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
RT_DEV void gbuffer_write(const SceneView& S, const FrameParams& P, float4* __restrict__ g0,
                          float4* __restrict__ g1, size_t li, float u, float v, int index)
{
    if (index < 0)
    {
        g0[li] = make_float4(0.0f, 0.0f, 0.0f, as_float(-1));
        g1[li] = make_float4(0.0f, 0.0f, 0.0f, as_float(0u));
        return;
    }
    const bool emissive = as_uint(S.trimat[2 * (size_t)index].w) != 0u;
    f3 p, n;
    surface_info(S.bvh, index, u, v, P.eye, p, n);
    g0[li] = make_float4(p.x, p.y, p.z, as_float(index));
    g1[li] = make_float4(n.x, n.y, n.z, as_float(emissive ? GB_EMISSIVE : GB_SHADED));
}

/* -------------------------------------------------------------------- raycast */
/* examples/10_restir_di/10_restir_di.cu:9-34 (+ common/camera.hpp:27-35) */
__global__ __launch_bounds__(BLOCK, RT_TRACE_WAVES) void k_raycast(SceneView S, FrameParams P, float4* __restrict__ vis,
                                                    float4* __restrict__ g0, float4* __restrict__ g1)
{
    __shared__ uint32_t s_stack[WIDE_LDS_STACK * BLOCK];
    int x, row;
    if (!tile_pixel(P, x, row)) return;
    const int yi = P.H - 1 - row;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;

    const float u = (float)x / (float)P.W, v = (float)yi / (float)P.H;
    const f3 forward = normalize(cross(P.rg_up, P.rg_right));
    const f3 to = P.rg_origin + forward + mix(-P.rg_right, P.rg_right, u) + mix(P.rg_up, -P.rg_up, v);
    const f3 rd = normalize(to - P.rg_origin);

    Hit h;
    h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.prim = -1;
    trace_wide<false>(S.wide, s_stack, P.rg_origin, rd, 0.0f, kFltMax, h);
    vis[li] = make_float4(h.u, h.v, as_float(h.prim), as_float(0));
    gbuffer_write(S, P, g0, g1, li, h.u, h.v, h.prim);
}

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
template <bool SHADOWED>
RT_DEV float target_function(const SceneView& S, uint32_t* s_stack, f3 op, f3 on, f3 hp, f3 hn, float lum)
{
    if (SHADOWED)
    {
        const float brdf = 1.0f / kPI;
        const float G = geometry_term(op, on, hp, hn);
        const float V = check_visibility_wide(S.wide, s_stack, op, on, hp) ? 1.0f : 0.0f;
        return brdf * G * V * lum;
    }
    return target_unshadowed(op, on, hp, hn, lum);
}

RT_DEV void res_take_sample(Res& r, const Res& o)
{
    r.hit_p = o.hit_p; r.hit_n = o.hit_n; r.org_p = o.org_p; r.org_n = o.org_n;
    r.rad = o.rad; r.lum = o.lum; r.vis = o.vis;
}

/* temporal merge of 10_restir_di.cu:177-233; r = current, pr = previous frame, same pixel */
template <bool SHADOWED>
RT_DEV void temporal_merge(const SceneView& S, uint32_t* s_stack, const FrameParams& P, int x, int yi, f3 sp, f3 sn, Res& r,
                           Res pr)
{
    PCG rng = pcg_init(hashPCG4((uint32_t)x, (uint32_t)yi, (uint32_t)P.frame, 1u), 0);
    const int cap = 20 * P.ris_sample_count;
    pr.M = pr.M < cap ? pr.M : cap;
    float p_hat_y = target_function<SHADOWED>(S, s_stack, sp, sn, pr.hit_p, pr.hit_n, pr.lum);
    if (P.vis_reuse) p_hat_y *= pr.vis ? 1.0f : 0.0f;
    pr.M = scale_M(pr.M, rejection_heuristics(r.org_p, r.org_n, pr.org_p, pr.org_n, P.eye));
    const float weight = p_hat_y * pr.ucw * (float)pr.M;
    const float u = rng.uniformf();
    r.w_sum += weight;
    r.M += pr.M;
    if (u < weight / r.w_sum) res_take_sample(r, pr);
    const float p_hat = target_function<SHADOWED>(S, s_stack, sp, sn, r.hit_p, r.hit_n, r.lum);
    r.ucw = p_hat > 0.0f ? r.w_sum / ((float)r.M * p_hat) : 0.0f;
}

/* --------------------------------------------------------- generate_candidate */
/* examples/10_restir_di/10_restir_di.cu:36-135; with FUSE_TEMPORAL also :137-237 on the
 * value still in registers (the reference round-trips it through reservoir_buffer0). */
template <bool FUSE_TEMPORAL, bool SHADOWED>
__global__ __launch_bounds__(BLOCK, RT_TRACE_WAVES) void k_generate_candidate(
    SceneView S, FrameParams P, const float4* __restrict__ g0, const float4* __restrict__ g1,
    const float4* __restrict__ prev_rec, const float4* __restrict__ prev_rad, float4* __restrict__ out_rec,
    float4* __restrict__ out_rad)
{
    __shared__ uint32_t s_stack[WIDE_LDS_STACK * BLOCK];
    int x, row;
    if (!tile_pixel(P, x, row)) return;
    const int yi = P.H - 1 - row;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;

    const float4 G0 = g0[li], G1 = g1[li];
    const uint32_t flags = as_uint(G1.w);
    Res r = res_zero();
    if (!(flags & GB_SHADED))
    {
        res_store(out_rec, out_rad, li, r, false); /* Reservoir{} (:56-70) */
        return;
    }
    const f3 sp = F3(G0.x, G0.y, G0.z), sn = F3(G1.x, G1.y, G1.z);

    PCG rng = pcg_init(hashPCG4((uint32_t)x, (uint32_t)yi, (uint32_t)P.frame, 0u), 0);
    const float fL = (float)(size_t)P.n_lights;
    int sel = -1;
    for (int i = 0; i < P.ris_sample_count; ++i)
    {
        /* draw order rv0, rv1, rv2, u: left-to-right argument evaluation (hipcc) */
        const float rv0 = rng.uniformf();
        float bx = rng.uniformf();
        float by = rng.uniformf();
        /* common/core.hpp:261-285 */
        uint32_t nth = (uint32_t)(rv0 * fL);
        if (nth == (uint32_t)P.n_lights) nth = (uint32_t)P.n_lights - 1u;
        /* 48-B light record = 3 lane-loads (the loop is bound by the number of divergent per-lane
         * loads, not by ALU): vertices + luminance(Ke) + pdf; the normal is recomputed with the
         * reference's exact expression (common/core.hpp:50-55), Ke is fetched once at the end. */
        const float4* L = S.lights + 3 * (size_t)nth;
        const float4 L0 = L[0], L1 = L[1], L2 = L[2];
        const f3 v0 = F3(L0.x, L0.y, L0.z), v1 = F3(L0.w, L1.x, L1.y), v2 = F3(L1.z, L1.w, L2.x);
        warp_unit_triangle(bx, by);
        const f3 lp = (1.0f - bx - by) * v0 + bx * v1 + by * v2;
        const f3 ln = tri_normal(v0, v1, v2);
        const float lum = L2.y;
        const float light_pdf = L2.z; /* 1/L * 1/area (:98-99) */
        const float p_hat = target_unshadowed(sp, sn, lp, ln, lum); /* unshadowed always (:104) */
        const float weight = p_hat / light_pdf;
        const float u = rng.uniformf();
        /* common/reservoir.hpp:22-29 */
        r.w_sum += weight;
        r.M += 1;
        if (u < weight / r.w_sum)
        {
            r.hit_p = lp; r.hit_n = ln; r.lum = lum;
            r.org_p = sp; r.org_n = sn; r.vis = false;
            sel = (int)nth;
        }
    }
    if (sel >= 0)
    {
        const float4 ke = S.light_ke[sel];
        r.rad = F3(ke.x, ke.y, ke.z);
    }
    {
        const float p_hat = target_function<SHADOWED>(S, s_stack, sp, sn, r.hit_p, r.hit_n, r.lum);
        r.ucw = p_hat > 0.0f ? r.w_sum / ((float)r.M * p_hat) : 0.0f;
    }
    if (P.vis_reuse) r.vis = check_visibility_wide(S.wide, s_stack, sp, sn, r.hit_p);

    if (FUSE_TEMPORAL)
    {
        bool dummy;
        Res pr = res_load(prev_rec, li, dummy);
        const float4 pq = prev_rad[li];
        pr.rad = F3(pq.x, pq.y, pq.z);
        temporal_merge<SHADOWED>(S, s_stack, P, x, yi, sp, sn, r, pr);
    }
    res_store(out_rec, out_rad, li, r, true);
}

/* -------------------------------------------------------- temporal_resampling */
/* examples/10_restir_di/10_restir_di.cu:137-237 (stand-alone entry point) */
template <bool SHADOWED>
__global__ __launch_bounds__(BLOCK) void k_temporal(SceneView S, FrameParams P, const float4* __restrict__ g0,
                                                     const float4* __restrict__ g1,
                                                     const float4* __restrict__ prev_rec,
                                                     const float4* __restrict__ prev_rad,
                                                     float4* __restrict__ rec, float4* __restrict__ radb)
{
    __shared__ uint32_t s_stack[WIDE_LDS_STACK * BLOCK];
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
    Res pr = res_load(prev_rec, li, dummy);
    const float4 pq = prev_rad[li];
    pr.rad = F3(pq.x, pq.y, pq.z);
    temporal_merge<SHADOWED>(S, s_stack, P, x, yi, sp, sn, r, pr);
    res_store(rec, radb, li, r, true);
}

/* --------------------------------------------------------- spatial_resampling */
/* examples/10_restir_di/10_restir_di.cu:256-388 — the roofline kernel.
 * Per neighbour ONE 64-B aligned record is gathered (the reference gathers a 16-B Visibility,
 * a Triangle and a 76-B Reservoir); the "sky / emissive neighbour" test of :326-338 reads the
 * shaded bit kept inside the record; radiance (side record) is fetched once, for the sample
 * that survived. */
template <bool SHADOWED>
__global__ __launch_bounds__(BLOCK) void k_spatial(SceneView S, FrameParams P, const float4* __restrict__ g0,
                                                    const float4* __restrict__ g1,
                                                    const float4* __restrict__ in_rec,
                                                    const float4* __restrict__ in_rad,
                                                    float4* __restrict__ out_rec, float4* __restrict__ out_rad)
{
    __shared__ uint32_t s_stack[WIDE_LDS_STACK * BLOCK];
    int x, row;
    if (!tile_pixel(P, x, row)) return;
    const int yi = P.H - 1 - row;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
    const float4 G0 = g0[li], G1 = g1[li];
    if (!(as_uint(G1.w) & GB_SHADED))
    {
        /* the reference stores nothing here (:275-287); we keep the shaded bit valid */
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
        const float scale = P.spatial_radius / 1.96f;
        for (int k = 0; k < P.spatial_count; ++k)
        {
            const float rv0 = rng.uniformf();
            const float rv1 = rng.uniformf();
            /* common/reservoir.hpp:89-95 with portable log/cos/sin */
            const float radius = sqrtf(fmax_dev(-2.0f * pm_logf(rv0), 0.0f));
            const float phi = 2.0f * kPI * rv1;
            const float gx = radius * pm_cosf(phi), gy = radius * pm_sinf(phi);
            const int nx = f2i_sat((float)x + scale * gx);
            const int ny = f2i_sat((float)yi + scale * gy);
            if (nx < 0 || nx >= P.W || ny < 0 || ny >= P.H) continue;
            if (nx == x && ny == yi) continue;
            const int nrow = P.H - 1 - ny;
            const int lr = nrow - P.lrow0;
            if (lr < 0 || lr >= P.lrows) continue; /* only when halo < 87: outside the contract */
            const size_t pid = (size_t)nx + (size_t)lr * P.W;
            bool n_shaded;
            Res nr = res_load(in_rec, pid, n_shaded);
            if (!n_shaded) continue; /* sky or emissive neighbour (:326-338) */

            float p_hat_y = target_function<SHADOWED>(S, s_stack, sp, sn, nr.hit_p, nr.hit_n, nr.lum);
            if (P.vis_reuse) p_hat_y *= nr.vis ? 1.0f : 0.0f;
            nr.M = scale_M(nr.M, rejection_heuristics(r.org_p, r.org_n, nr.org_p, nr.org_n, P.eye));
            const float weight = p_hat_y * nr.ucw * (float)nr.M;
            const float u = rng.uniformf();
            r.w_sum += weight;
            r.M += nr.M;
            if (u < weight / r.w_sum)
            {
                res_take_sample(r, nr);
                rad_from = pid;
            }
        }
        const float p_hat = target_function<SHADOWED>(S, s_stack, sp, sn, r.hit_p, r.hit_n, r.lum);
        r.ucw = p_hat > 0.0f ? r.w_sum / ((float)r.M * p_hat) : 0.0f;
    }
    const float4 rq = in_rad[rad_from];
    r.rad = F3(rq.x, rq.y, rq.z);
    res_store(out_rec, out_rad, li, r, true);
}


/* SURVEY.md §8(d) ALGORITHMIC bytes of one spatial_resampling launch, counted with the
 * reference's record sizes (Visibility 16 B, Reservoir 76 B): per pixel 16; per shaded pixel
 * +76 in +76 out; per neighbour that passed the on-screen / not-self tests +16, and +76 more if
 * it is shaded. Replays exactly the RNG draws of k_spatial (the accept decisions depend only on
 * the RNG and the shaded bits, not on reservoir contents). Measurement aid, not on the hot path. */
__global__ __launch_bounds__(BLOCK) void k_spatial_bytes(FrameParams P, const float4* __restrict__ g1,
                                                          const float4* __restrict__ in_rec,
                                                          unsigned long long* __restrict__ out)
{
    int x, row;
    const bool ok = tile_pixel(P, x, row);
    unsigned long long bytes = 0, accepted = 0;
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
                    const float radius = sqrtf(fmax_dev(-2.0f * pm_logf(rv0), 0.0f));
                    const float phi = 2.0f * kPI * rv1;
                    const int nx = f2i_sat((float)x + scale * (radius * pm_cosf(phi)));
                    const int ny = f2i_sat((float)yi + scale * (radius * pm_sinf(phi)));
                    if (nx < 0 || nx >= P.W || ny < 0 || ny >= P.H) continue;
                    if (nx == x && ny == yi) continue;
                    const int lr = P.H - 1 - ny - P.lrow0;
                    if (lr < 0 || lr >= P.lrows) continue;
                    bytes += 16;
                    accepted += 1;
                    const uint32_t mb = as_uint(in_rec[4 * ((size_t)nx + (size_t)lr * P.W) + 1].w);
                    if (!(mb & RES_SHADED_BIT)) continue;
                    bytes += 76;
                    rng.uniformf();
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1)
    {
        bytes += __shfl_down(bytes, off);
        accepted += __shfl_down(accepted, off);
    }
    if ((threadIdx.x & 63) == 0)
    {
        atomicAdd(&out[0], bytes);
        atomicAdd(&out[1], accepted);
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
__global__ __launch_bounds__(BLOCK) void k_halo_mark(FrameParams P, const float4* __restrict__ g1, int reg_row0,
                                                      int reg_rows, uint32_t* __restrict__ bitmap)
{
    int x, row;
    if (!tile_pixel(P, x, row)) return;
    const int yi = P.H - 1 - row;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
    if (!(as_uint(g1[li].w) & GB_SHADED) || !P.use_spatial) return;
    PCG rng = pcg_init(hashPCG4((uint32_t)x, (uint32_t)yi, (uint32_t)P.frame, (uint32_t)(2 + P.pass)), 0);
    const float scale = P.spatial_radius / 1.96f;
    for (int k = 0; k < P.spatial_count; ++k)
    {
        const float rv0 = rng.uniformf();
        const float rv1 = rng.uniformf();
        const float radius = sqrtf(fmax_dev(-2.0f * pm_logf(rv0), 0.0f));
        const float phi = 2.0f * kPI * rv1;
        const int nx = f2i_sat((float)x + scale * (radius * pm_cosf(phi)));
        const int ny = f2i_sat((float)yi + scale * (radius * pm_sinf(phi)));
        if (nx < 0 || nx >= P.W || ny < 0 || ny >= P.H) continue;
        if (nx == x && ny == yi) continue;
        const int nrow = P.H - 1 - ny;
        const int lr = nrow - P.lrow0;
        if (lr < 0 || lr >= P.lrows) continue;
        if (nrow >= reg_row0 && nrow < reg_row0 + reg_rows)
        {
            const uint32_t bit = (uint32_t)(nrow - reg_row0) * (uint32_t)P.W + (uint32_t)nx;
            atomicOr(&bitmap[1 + (bit >> 5)], 1u << (bit & 31u));
        }
        if (!(as_uint(g1[(size_t)nx + (size_t)lr * P.W].w) & GB_SHADED)) continue;
        rng.uniformf();
    }
}
/* one workgroup: exclusive prefix of the per-word popcounts, total into word 0 */
__global__ void k_halo_scan(uint32_t* __restrict__ bitmap, int nw)
{
    __shared__ uint32_t s_sum[1024];
    const int t = threadIdx.x, T = blockDim.x;
    const int per = (nw + T - 1) / T;
    const int w0 = t * per, w1 = min(nw, w0 + per);
    uint32_t local = 0;
    for (int w = w0; w < w1; ++w) local += (uint32_t)__popc(bitmap[1 + w]);
    s_sum[t] = local;
    __syncthreads();
    for (int off = 1; off < T; off <<= 1)
    {
        const uint32_t v = t >= off ? s_sum[t - off] : 0u;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    uint32_t run = s_sum[t] - local;
    for (int w = w0; w < w1; ++w)
    {
        bitmap[1 + nw + w] = run;
        run += (uint32_t)__popc(bitmap[1 + w]);
    }
    if (t == T - 1) bitmap[0] = s_sum[t];
}
/* PACK: marked records of rows [row0, row0+rows) -> dense list (64 B record + 16 B radiance each);
 * UNPACK: the reverse. */
template <bool PACK>
__global__ void k_halo_sparse(const uint32_t* __restrict__ bitmap, int nw, int W, size_t region_off, int n_pix,
                              float4* __restrict__ rec, float4* __restrict__ radb, float4* __restrict__ list)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix) return;
    const uint32_t word = bitmap[1 + (i >> 5)];
    if (!(word & (1u << (i & 31)))) return;
    const uint32_t idx = bitmap[1 + nw + (i >> 5)] + (uint32_t)__popc(word & ((1u << (i & 31)) - 1u));
    float4* L = list + 5 * (size_t)idx;
    const size_t p = region_off + (size_t)i;
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
    (void)W;
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
/* examples/10_restir_di/10_restir_di.cu:390-459 */
__global__ __launch_bounds__(BLOCK, RT_TRACE_WAVES) void k_resolve(SceneView S, FrameParams P, const float4* __restrict__ g0,
                                                    const float4* __restrict__ g1,
                                                    const float4* __restrict__ rec,
                                                    const float4* __restrict__ radb, float4* __restrict__ accum)
{
    __shared__ uint32_t s_stack[WIDE_LDS_STACK * BLOCK];
    int x, row;
    if (!tile_pixel(P, x, row)) return;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
    const float4 G0 = g0[li], G1 = g1[li];
    const int tri = as_int(G0.w);
    const uint32_t flags = as_uint(G1.w);
    if (tri < 0) { accum[li] = make_float4(0.0f, 0.0f, 0.0f, 1.0f); return; }
    if (flags & GB_EMISSIVE)
    {
        const float4 ke = S.trimat[2 * (size_t)tri + 1];
        accum[li] = make_float4(ke.x, ke.y, ke.z, 1.0f);
        return;
    }
    const f3 sp = F3(G0.x, G0.y, G0.z), sn = F3(G1.x, G1.y, G1.z);
    const float4 q0 = rec[4 * li + 0], q1 = rec[4 * li + 1];
    const float4 rq = radb[li];
    const float4 kd = S.trimat[2 * (size_t)tri];
    const f3 hp = F3(q0.x, q0.y, q0.z), hn = F3(q1.x, q1.y, q1.z);
    const f3 brdf = (1.0f / kPI) * F3(kd.x, kd.y, kd.z);
    const float G = geometry_term(sp, sn, hp, hn);
    const float V = check_visibility_wide(S.wide, s_stack, sp, sn, hp) ? 1.0f : 0.0f;
    const f3 radiance = brdf * G * V * F3(rq.x, rq.y, rq.z) * q0.w;
    if (P.accumulate)
    {
        const float4 a = accum[li];
        accum[li] = make_float4(a.x + radiance.x, a.y + radiance.y, a.z + radiance.z, a.w + 1.0f);
    }
    else { accum[li] = make_float4(radiance.x, radiance.y, radiance.z, 1.0f); }
}


/* ------------------------------------------------------- configs #2 / #3: path tracers */
/* common/core.hpp:76-89 with portable cos/sin [parity] */
RT_DEV f3 sample_hemisphere(float r0, float r1, float r2)
{
    const float theta = r0 * 2.0f * kPI;
    float radius = r1 + r2;
    if (1.0f < radius) radius = 2.0f - radius;
    const float x = pm_cosf(theta) * radius;
    const float z = pm_sinf(theta) * radius;
    const float a = 1.0f - radius * radius;
    const float y = sqrtf((a < 0.0f) ? 0.0f : a);
    return F3(x, y, z);
}

/* examples/07_pt/07_pt.cu:11-90 (EXAMPLE 7) and examples/09_ris/09_ris.cu:11-166 (EXAMPLE 9): the
 * `path_trace` kernels, one thread per pixel, whole path in one launch. rays[0] accumulates the
 * number of raytrace() calls (one atomic per wave). */
template <int EXAMPLE, bool SHADOWED>
__global__ __launch_bounds__(BLOCK) void k_path_trace(SceneView S, FrameParams P, int max_depth, f3 sky,
                                                       float4* __restrict__ accum,
                                                       unsigned long long* __restrict__ rays)
{
    __shared__ uint32_t s_stack[WIDE_LDS_STACK * BLOCK];
    int x, row;
    const bool ok = tile_pixel(P, x, row);
    unsigned long long nrays = 0;
    if (ok)
    {
        const int yi = P.H - 1 - row;
        const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
        PCG rng = pcg_init(hashPCG3((uint32_t)x, (uint32_t)yi, (uint32_t)P.frame), 0);
        const float u = (float)x / (float)P.W, v = (float)yi / (float)P.H;
        const f3 forward = normalize(cross(P.rg_up, P.rg_right));
        const f3 to = P.rg_origin + forward + mix(-P.rg_right, P.rg_right, u) + mix(P.rg_up, -P.rg_up, v);
        f3 ro = P.rg_origin;
        f3 rd = normalize(to - P.rg_origin);
        f3 radiance = F3(0.0f, 0.0f, 0.0f), throughput = F3(1.0f, 1.0f, 1.0f);
        const float fL = (float)(size_t)P.n_lights;
        for (int depth = 0; depth < max_depth; ++depth)
        {
            Hit h;
            ++nrays;
            if (!trace_wide<false>(S.wide, s_stack, ro, rd, 0.0f, kFltMax, h))
            {
                if (EXAMPLE == 7) radiance = radiance + throughput * sky;
                break;
            }
            const float4 kd4 = S.trimat[2 * (size_t)h.prim];
            if (as_uint(kd4.w) != 0u)
            {
                const float4 ke4 = S.trimat[2 * (size_t)h.prim + 1];
                if (EXAMPLE == 7 || depth == 0) radiance = radiance + throughput * F3(ke4.x, ke4.y, ke4.z);
                break;
            }
            /* common/core.hpp:152-165 */
            f3 v0, v1, v2;
            load_tri(S.bvh.tv, h.prim, v0, v1, v2);
            const f3 sp = ro + h.t * rd;
            f3 sn = tri_normal(v0, v1, v2);
            if (dot(-rd, sn) < 0.0f) sn = -sn;
            const f3 kd = F3(kd4.x, kd4.y, kd4.z);

            if (EXAMPLE == 9)
            {
                /* RIS over the lights (09_ris.cu:61-99), then the shaded contribution (:101-126) */
                Res r = res_zero();
                for (int i = 0; i < P.ris_sample_count; ++i)
                {
                    const float rv0 = rng.uniformf();
                    float bx = rng.uniformf();
                    float by = rng.uniformf();
                    uint32_t nth = (uint32_t)(rv0 * fL);
                    if (nth == (uint32_t)P.n_lights) nth = (uint32_t)P.n_lights - 1u;
                    const float4* L = S.lights + 3 * (size_t)nth;
                    const float4 L0 = L[0], L1 = L[1], L2 = L[2];
                    const f3 a0 = F3(L0.x, L0.y, L0.z), a1 = F3(L0.w, L1.x, L1.y), a2 = F3(L1.z, L1.w, L2.x);
                    warp_unit_triangle(bx, by);
                    const f3 lp = (1.0f - bx - by) * a0 + bx * a1 + by * a2;
                    const f3 ln = tri_normal(a0, a1, a2);
                    float p_hat;
                    if (SHADOWED)
                    {
                        p_hat = target_function<true>(S, s_stack, sp, sn, lp, ln, L2.y);
                        ++nrays;
                    }
                    else { p_hat = target_unshadowed(sp, sn, lp, ln, L2.y); }
                    const float weight = p_hat / L2.z;
                    const float uu = rng.uniformf();
                    r.w_sum += weight;
                    r.M += 1;
                    if (uu < weight / r.w_sum)
                    {
                        r.hit_p = lp; r.hit_n = ln; r.lum = L2.y;
                        const float4 ke = S.light_ke[nth];
                        r.rad = F3(ke.x, ke.y, ke.z);
                    }
                }
                const f3 brdf = (1.0f / kPI) * kd;
                const float G = geometry_term(sp, sn, r.hit_p, r.hit_n);
                const float V = check_visibility_wide(S.wide, s_stack, sp, sn, r.hit_p) ? 1.0f : 0.0f;
                ++nrays;
                const float p_hat = target_function<SHADOWED>(S, s_stack, sp, sn, r.hit_p, r.hit_n, r.lum);
                if (SHADOWED) ++nrays;
                const float ucw = p_hat > 0.0f ? r.w_sum / ((float)r.M * p_hat) : 0.0f;
                radiance = radiance + throughput * brdf * G * V * r.rad * ucw;
            }
            /* next direction (07_pt.cu:61-70 / 09_ris.cu:128-137): common/core.hpp:216-235 */
            const float r0 = rng.uniformf();
            const float r1 = rng.uniformf();
            const float r2 = rng.uniformf();
            const f3 wl = sample_hemisphere(r0, r1, r2);
            const f3 tg = normalize(v1 - v0);
            const f3 bt = normalize(cross(tg, sn));
            const f3 wo = wl.x * tg + wl.y * sn + wl.z * bt;
            throughput = throughput * kd;
            ro = sp + 0.001f * sn;
            rd = wo;
        }
        if (P.accumulate)
        {
            const float4 a = accum[li];
            accum[li] = make_float4(a.x + radiance.x, a.y + radiance.y, a.z + radiance.z, a.w + 1.0f);
        }
        else { accum[li] = make_float4(radiance.x, radiance.y, radiance.z, 1.0f); }
    }
    for (int off = 32; off > 0; off >>= 1) nrays += __shfl_down(nrays, off);
    if ((threadIdx.x & 63) == 0 && nrays) atomicAdd(rays, nrays);
}

/* --------------------------------------------------------- clear / tone_mapping */
/* common/kernels/common.cu:4-17 */
__global__ __launch_bounds__(BLOCK) void k_clear(FrameParams P, float4* __restrict__ accum)
{
    int x, row;
    if (!tile_pixel(P, x, row)) return;
    accum[(size_t)x + (size_t)(row - P.lrow0) * P.W] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
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
/* common/kernels/common.cu:30-74 (display only; powf = portable exp(y*log(x))) */
__global__ __launch_bounds__(BLOCK) void k_tone_mapping(FrameParams P, const float4* __restrict__ accum,
                                                         uint32_t* __restrict__ pixels)
{
    int x, row;
    if (!tile_pixel(P, x, row)) return;
    const size_t li = (size_t)x + (size_t)(row - P.lrow0) * P.W;
    const float4 a = accum[li];
    const float gamma = 1.0f / 2.2f;
    const float r = pm_powf_pos(aces(a.x / a.w * 1.0f), gamma);
    const float g = pm_powf_pos(aces(a.y / a.w * 1.0f), gamma);
    const float b = pm_powf_pos(aces(a.z / a.w * 1.0f), gamma);
    pixels[li] = to_u8(r * 255.0f) | (to_u8(g * 255.0f) << 8) | (to_u8(b * 255.0f) << 16) | 0xff000000u;
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
    r.lum = luminance(r.rad);
    const bool shaded = (as_uint(g1[i].w) & GB_SHADED) != 0u;
    res_store(rec, radb, (size_t)i, r, shaded);
}

/* ------------------------------------------------------------- scene tables */
/* per emissive triangle (index order, 10_restir_di.cpp:196-205), 3 x float4:
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
    float4* L = lights + 3 * (size_t)i;
    L[0] = make_float4(v0.x, v0.y, v0.z, v1.x);
    L[1] = make_float4(v1.y, v1.z, v2.x, v2.y);
    L[2] = make_float4(v2.z, luminance(ke), pdf, as_float(ti));
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
template <int MODE> /* 0 = wide (production), 1 = binary stackless */
__global__ __launch_bounds__(BLOCK) void k_trace_closest(SceneView S, const float* __restrict__ rays, int n, float* __restrict__ hits)
{
    __shared__ uint32_t s_stack[MODE == 0 ? WIDE_LDS_STACK * BLOCK : 1];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + 8 * (size_t)i;
    Hit h;
    h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.prim = -1;
    if (MODE == 0) trace_wide<false>(S.wide, s_stack, F3(r[0], r[1], r[2]), F3(r[3], r[4], r[5]), r[6], r[7], h);
    else trace<false>(S.bvh, F3(r[0], r[1], r[2]), F3(r[3], r[4], r[5]), r[6], r[7], h);
    float* o = hits + 4 * (size_t)i;
    o[0] = h.t; o[1] = h.u; o[2] = h.v; o[3] = as_float(h.prim);
}
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_trace_stats(SceneView S, const float* __restrict__ rays, int n, uint32_t* __restrict__ stats)
{
    __shared__ uint32_t s_stack[MODE == 0 ? WIDE_LDS_STACK * BLOCK : 1];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + 8 * (size_t)i;
    Hit h;
    uint32_t st[2] = {0u, 0u};
    if (MODE == 0) trace_wide<false, true>(S.wide, s_stack, F3(r[0], r[1], r[2]), F3(r[3], r[4], r[5]), r[6], r[7], h, st);
    else trace<false, true>(S.bvh, F3(r[0], r[1], r[2]), F3(r[3], r[4], r[5]), r[6], r[7], h, st);
    stats[2 * (size_t)i] = st[0];
    stats[2 * (size_t)i + 1] = st[1];
}

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
        const float4* r = wide.rec + 3 * (size_t)(cur & ~WIDE_LEAF_BIT);
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
                tn = fmaxf(tn * (1.0f - 4e-7f), tmin);
                tf = fminf(tf * (1.0f + 4e-7f), best);
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
        case 23: r = pm_expf(in[i]); break;
        case 24: r = pm_pow8f(in[i]); break;
        case 25: r = pm_powf_pos(in[i], 1.0f / 2.2f); break;
        case 26: r = in[2 * (size_t)i] / in[2 * (size_t)i + 1]; break;
        case 27: r = sqrtf(in[i]); break;
        default: break;
    }
    out[i] = r;
}

/* ======================================================================= host */

struct rt_ctx
{
    int device = 0, W = 0, H = 0, row_begin = 0, row_end = 0, halo = 0;
    int lrow0 = 0, lrows = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::string err;

    int n_tris = 0, n_lights = 0, bvh_height = 0, n_refs = 0;
    /* rt_tuning: tile order per kernel {raycast, generate, spatial, resolve, other} and the spatial
     * pass's occupancy limiter (extra dynamic LDS). Defaults from the sweep in DESIGN.md §5. */
    int tune_tile_mode[5] = {1, 0, 1, 0, 0};
    int tune_spatial_lds = 32768;
    int trace_mode = 0; /* rt_trace_closest / rt_trace_stats: 0 = wide (what the frame kernels use), 1 = binary stackless */
    float bvh_split_factor = 8.0f; /* fragment length in median triangle extents; 0 = no pre-split */
    int bvh_builder = 1; /* 0 = device LBVH (Morton/Karras), 1 = host binned SAH (high quality) */
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
    float4* d_rec[3] = {nullptr, nullptr, nullptr};
    float4* d_rad[3] = {nullptr, nullptr, nullptr};
    int res_map[3] = {0, 1, 2};
    int sub0 = -1, sub1 = -1; /* row sub-range of the running rt_frame_stage_run (-1: all owned rows) */
    int fX = 0, fY = 1, fZ = 2, f_in = 0, f_out = 1, f_stage = 0, f_final = RT_RES_1;
    bool f_clear = false; /* rt_frame_stage state */
    unsigned long long* d_counter = nullptr;
    void* d_stage = nullptr;
    size_t stage_bytes = 0;

    rt_options opt;
    rt_raygen rg;
    float eye[3] = {0, 0, 0};
    bool has_camera = false, has_scene = false, has_gbuffer = false;
    float cam_eye[3] = {8.0f, 8.0f, 8.0f}, cam_at[3] = {0.0f, 0.0f, 0.0f}, cam_fovy = 0.78539816339f; /* misc.hpp:217-218 */
    bool cam_updated = false;

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
    P.tile_mode = c->tune_tile_mode[kernel];
    return P;
}
static SceneView make_scene(const rt_ctx* c)
{
    SceneView S;
    S.bvh.nodes = c->d_nodes; S.bvh.tv = c->d_tv; S.bvh.n_tris = c->n_tris;
    S.wide.rec = c->d_wide; S.wide.n_tris = c->n_tris;
    S.trimat = c->d_trimat; S.lights = c->d_lights; S.light_ke = c->d_light_ke;
    return S;
}
static size_t local_pixels(const rt_ctx* c) { return (size_t)c->W * (size_t)c->lrows; }

extern "C" {

int rt_create(int device, int width, int height, int row_begin, int row_end, int halo, rt_ctx** out)
{
    if (!out || width <= 0 || height <= 0 || row_begin < 0 || row_end > height || row_begin >= row_end || halo < 0)
        return RT_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev) return RT_ERR_NO_DEVICE;
    rt_ctx* c = new rt_ctx();
    *out = c;
    c->device = device; c->W = width; c->H = height;
    c->row_begin = row_begin; c->row_end = row_end; c->halo = halo;
    c->lrow0 = row_begin - halo < 0 ? 0 : row_begin - halo;
    const int lend = row_end + halo > height ? height : row_end + halo;
    c->lrows = lend - c->lrow0;
    c->opt = default_options();
    memset(&c->rg, 0, sizeof(c->rg));
    RT_HIP(c, hipSetDevice(device));
    RT_HIP(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
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
    if (c->own_stream) hipStreamSynchronize(c->own_stream);
    free_scene(c);
    hipFree(c->d_vis); hipFree(c->d_g0); hipFree(c->d_g1); hipFree(c->d_accum); hipFree(c->d_pixels);
    for (int k = 0; k < 3; ++k) { hipFree(c->d_rec[k]); hipFree(c->d_rad[k]); }
    hipFree(c->d_counter); hipFree(c->d_stage);
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

/* ---- early split clipping (host): cut triangles whose box is longer than L into fragments ---- */
struct BvhRef
{
    float lo[3], hi[3];
    int tri;
};
struct ClipPoly
{
    int n;
    float v[12][3];
};
static void poly_bounds(const ClipPoly& p, float* lo, float* hi)
{
    for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    for (int i = 0; i < p.n; ++i)
        for (int a = 0; a < 3; ++a)
        {
            lo[a] = fminf(lo[a], p.v[i][a]);
            hi[a] = fmaxf(hi[a], p.v[i][a]);
        }
}
/* keep the part of p with (sign > 0 ? x_a >= s : x_a <= s) */
static ClipPoly poly_clip(const ClipPoly& p, int a, float s, int sign)
{
    ClipPoly o;
    o.n = 0;
    for (int i = 0; i < p.n; ++i)
    {
        const float* c = p.v[i];
        const float* d = p.v[(i + 1) % p.n];
        const bool cin = sign > 0 ? c[a] >= s : c[a] <= s;
        const bool din = sign > 0 ? d[a] >= s : d[a] <= s;
        if (cin && o.n < 12) { memcpy(o.v[o.n++], c, 12); }
        if (cin != din && o.n < 12)
        {
            const float t = (s - c[a]) / (d[a] - c[a]);
            for (int k = 0; k < 3; ++k) o.v[o.n][k] = c[k] + (d[k] - c[k]) * t;
            o.v[o.n][a] = s;
            o.n++;
        }
    }
    return o;
}
static void split_refs(const rt_triangle* tris, int n, float L, float pad, std::vector<BvhRef>& out)
{
    out.clear();
    std::vector<ClipPoly> stack;
    for (int i = 0; i < n; ++i)
    {
        ClipPoly p;
        p.n = 3;
        for (int k = 0; k < 3; ++k) memcpy(p.v[k], tris[i].v[k], 12);
        stack.clear();
        stack.push_back(p);
        int emitted = 0;
        while (!stack.empty())
        {
            ClipPoly q = stack.back();
            stack.pop_back();
            float lo[3], hi[3];
            poly_bounds(q, lo, hi);
            int a = 0;
            for (int k = 1; k < 3; ++k)
                if (hi[k] - lo[k] > hi[a] - lo[a]) a = k;
            const float ext = hi[a] - lo[a];
            bool split = L > 0.0f && ext > L && emitted + (int)stack.size() < 4096 && q.n >= 3;
            float s = 0.0f;
            if (split)
            {
                /* split plane on the global L-grid so that fragments of neighbours line up */
                const float mid = 0.5f * (lo[a] + hi[a]);
                s = L * floorf(mid / L + 0.5f);
                if (!(s > lo[a] + 0.01f * ext && s < hi[a] - 0.01f * ext)) s = mid;
                if (!(s > lo[a] && s < hi[a])) split = false;
            }
            if (split)
            {
                ClipPoly l = poly_clip(q, a, s, -1), r = poly_clip(q, a, s, +1);
                if (l.n >= 3 && r.n >= 3)
                {
                    stack.push_back(l);
                    stack.push_back(r);
                    continue;
                }
            }
            BvhRef ref;
            for (int k = 0; k < 3; ++k) { ref.lo[k] = lo[k] - pad; ref.hi[k] = hi[k] + pad; }
            ref.tri = i;
            out.push_back(ref);
            ++emitted;
        }
    }
}

/* ---- collapse the binary tree (LBVH or SAH) into the 4-wide quantised structure of bvh.h (host) ---- */
struct WideRec
{
    uint32_t w[12];
}; /* 48 B */
static_assert(sizeof(WideRec) == 48, "wide record");

static inline float box_area6(const float* lo, const float* hi)
{
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

struct WideChild
{
    int bin; /* >= 0 binary inner node, < 0: ~triangle */
    float lo[3], hi[3];
};
static void bin_children(const BvhNode& n, WideChild out[2])
{
    out[0].bin = n.d.x; out[1].bin = n.d.y;
    out[0].lo[0] = n.a.x; out[0].lo[1] = n.a.y; out[0].lo[2] = n.a.z;
    out[0].hi[0] = n.b.x; out[0].hi[1] = n.b.y; out[0].hi[2] = n.b.z;
    out[1].lo[0] = n.a.w; out[1].lo[1] = n.b.w; out[1].lo[2] = n.c.w;
    out[1].hi[0] = n.c.x; out[1].hi[1] = n.c.y; out[1].hi[2] = n.c.z;
}
/* returns the wide height, fills recs */
static int collapse_wide(const std::vector<BvhNode>& bin, const rt_triangle* tris, std::vector<WideRec>& recs)
{
    struct Work { int bin; uint32_t out; int depth; };
    recs.clear();
    recs.reserve(bin.size() * 2 + 8);
    recs.push_back(WideRec());
    std::vector<Work> stack;
    stack.push_back({0, 0u, 1});
    int height = 1;
    while (!stack.empty())
    {
        const Work wk = stack.back();
        stack.pop_back();
        height = std::max(height, wk.depth);
        WideChild ch[4];
        int n = 2;
        bin_children(bin[(size_t)wk.bin], ch);
        while (n < 4)
        {
            int pick = -1;
            float best = -1.0f;
            for (int k = 0; k < n; ++k)
                if (ch[k].bin >= 0)
                {
                    const float a = box_area6(ch[k].lo, ch[k].hi);
                    if (a > best) { best = a; pick = k; }
                }
            if (pick < 0) break;
            WideChild two[2];
            bin_children(bin[(size_t)ch[pick].bin], two);
            ch[pick] = two[0];
            ch[n++] = two[1];
        }
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < n; ++k)
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], ch[k].lo[a]); hi[a] = fmaxf(hi[a], ch[k].hi[a]); }
        const uint32_t base = (uint32_t)recs.size();
        recs.resize(recs.size() + (size_t)n);
        /* per-axis power-of-two scale with 255 steps covering the node box */
        uint32_t ebits[3];
        float scale[3];
        for (int a = 0; a < 3; ++a)
        {
            const float ext = fmaxf(hi[a] - lo[a], 1e-30f);
            int e;
            frexpf(ext / 255.0f, &e); /* ext/255 = m * 2^e, m in [0.5,1) => 2^e >= ext/255 */
            int biased = e + 127;
            if (biased < 1) biased = 1;
            if (biased > 254) biased = 254;
            ebits[a] = (uint32_t)biased;
            scale[a] = ldexpf(1.0f, biased - 127);
        }
        uint32_t q[6] = {0, 0, 0, 0, 0, 0}, meta = 0;
        for (int k = 0; k < n; ++k)
        {
            for (int a = 0; a < 3; ++a)
            {
                int ql = (int)floorf((ch[k].lo[a] - lo[a]) / scale[a]);
                int qh = (int)ceilf((ch[k].hi[a] - lo[a]) / scale[a]);
                /* the device decodes lo + q*scale in binary32: make sure the decoded box contains the child box */
                while (ql > 0 && lo[a] + (float)ql * scale[a] > ch[k].lo[a]) --ql;
                while (qh < 255 && lo[a] + (float)qh * scale[a] < ch[k].hi[a]) ++qh;
                ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
                qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
                q[a] |= (uint32_t)ql << (8 * k);
                q[3 + a] |= (uint32_t)qh << (8 * k);
            }
            if (ch[k].bin >= 0)
            {
                meta |= 1u << (8 * k);
                stack.push_back({ch[k].bin, base + (uint32_t)k, wk.depth + 1});
            }
            else
            {
                meta |= 2u << (8 * k);
                const int ti = ~ch[k].bin;
                const rt_triangle& t = tris[ti];
                WideRec& L = recs[base + (size_t)k];
                const float f[9] = {t.v[0][0], t.v[0][1], t.v[0][2], t.v[1][0], t.v[1][1], t.v[1][2], t.v[2][0], t.v[2][1], t.v[2][2]};
                for (int i = 0; i < 9; ++i) L.w[i] = f2u(f[i]);
                L.w[9] = (uint32_t)ti;
                L.w[10] = L.w[11] = 0u;
            }
        }
        WideRec& R = recs[wk.out];
        R.w[0] = f2u(lo[0]); R.w[1] = f2u(lo[1]); R.w[2] = f2u(lo[2]);
        R.w[3] = ebits[0] | (ebits[1] << 8) | (ebits[2] << 16);
        R.w[4] = base; R.w[5] = meta; R.w[6] = q[0]; R.w[7] = q[1];
        R.w[8] = q[2]; R.w[9] = q[3]; R.w[10] = q[4]; R.w[11] = q[5];
    }
    return height;
}

/* ---- high-quality build (the reference asks HIPRT for hiprtBuildFlagBitPreferHighQualityBuild,
 * common/loader.hpp:98-99): top-down binned-SAH binary tree over the references on the host, in
 * the same BvhNode format the device LBVH emits (so both traversals and the wide collapse work
 * on either). One reference per leaf. ---- */
struct SahBuilder
{
    const std::vector<BvhRef>& refs;
    std::vector<int> order;
    std::vector<float> cent; /* 3 per ref */
    std::vector<BvhNode> nodes;
    int height = 0;
    explicit SahBuilder(const std::vector<BvhRef>& r) : refs(r)
    {
        order.resize(r.size());
        cent.resize(r.size() * 3);
        for (size_t i = 0; i < r.size(); ++i)
        {
            order[i] = (int)i;
            for (int a = 0; a < 3; ++a) cent[3 * i + a] = 0.5f * (r[i].lo[a] + r[i].hi[a]);
        }
        nodes.reserve(r.size());
    }
    void bounds(int first, int count, float* lo, float* hi, float* clo, float* chi) const
    {
        for (int a = 0; a < 3; ++a) { lo[a] = clo[a] = INFINITY; hi[a] = chi[a] = -INFINITY; }
        for (int i = first; i < first + count; ++i)
        {
            const BvhRef& r = refs[(size_t)order[i]];
            for (int a = 0; a < 3; ++a)
            {
                lo[a] = fminf(lo[a], r.lo[a]); hi[a] = fmaxf(hi[a], r.hi[a]);
                const float c = cent[3 * (size_t)order[i] + a];
                clo[a] = fminf(clo[a], c); chi[a] = fmaxf(chi[a], c);
            }
        }
    }
    /* returns child code: >= 0 node index, < 0 ~triangle; box of the subtree in lo/hi */
    int build(int first, int count, int parent, int depth, float* lo, float* hi)
    {
        float clo[3], chi[3];
        bounds(first, count, lo, hi, clo, chi);
        if (depth > height) height = depth;
        if (count == 1) return ~refs[(size_t)order[first]].tri;
        constexpr int NB = 16;
        int best_axis = -1, best_split = -1;
        float best_cost = INFINITY;
        for (int a = 0; a < 3; ++a)
        {
            const float ext = chi[a] - clo[a];
            if (!(ext > 0.0f)) continue;
            float blo[NB][3], bhi[NB][3];
            int bc[NB];
            for (int b = 0; b < NB; ++b) { bc[b] = 0; for (int k = 0; k < 3; ++k) { blo[b][k] = INFINITY; bhi[b][k] = -INFINITY; } }
            const float sc = (float)NB / ext;
            for (int i = first; i < first + count; ++i)
            {
                const int id = order[i];
                int b = (int)((cent[3 * (size_t)id + a] - clo[a]) * sc);
                b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                bc[b]++;
                for (int k = 0; k < 3; ++k) { blo[b][k] = fminf(blo[b][k], refs[(size_t)id].lo[k]); bhi[b][k] = fmaxf(bhi[b][k], refs[(size_t)id].hi[k]); }
            }
            float ra[NB]; int rc[NB];
            float l3[3] = {INFINITY, INFINITY, INFINITY}, h3[3] = {-INFINITY, -INFINITY, -INFINITY};
            int cnt = 0;
            for (int b = NB - 1; b > 0; --b)
            {
                for (int k = 0; k < 3; ++k) { l3[k] = fminf(l3[k], blo[b][k]); h3[k] = fmaxf(h3[k], bhi[b][k]); }
                cnt += bc[b];
                ra[b] = cnt ? box_area6(l3, h3) : 0.0f;
                rc[b] = cnt;
            }
            for (int k = 0; k < 3; ++k) { l3[k] = INFINITY; h3[k] = -INFINITY; }
            cnt = 0;
            for (int b = 0; b < NB - 1; ++b)
            {
                for (int k = 0; k < 3; ++k) { l3[k] = fminf(l3[k], blo[b][k]); h3[k] = fmaxf(h3[k], bhi[b][k]); }
                cnt += bc[b];
                if (cnt == 0 || rc[b + 1] == 0) continue;
                const float cost = box_area6(l3, h3) * (float)cnt + ra[b + 1] * (float)rc[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = b; }
            }
        }
        int mid;
        if (best_axis < 0) mid = first + count / 2;
        else
        {
            const float ext = chi[best_axis] - clo[best_axis];
            const float sc = (float)NB / ext;
            int i = first, j = first + count - 1;
            while (i <= j)
            {
                int b = (int)((cent[3 * (size_t)order[i] + best_axis] - clo[best_axis]) * sc);
                b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                if (b <= best_split) ++i;
                else { std::swap(order[i], order[j]); --j; }
            }
            mid = i;
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
        const int me = (int)nodes.size();
        nodes.push_back(BvhNode());
        float l0[3], h0[3], l1[3], h1[3];
        const int c0 = build(first, mid - first, me, depth + 1, l0, h0);
        const int c1 = build(mid, first + count - mid, me, depth + 1, l1, h1);
        BvhNode& n = nodes[(size_t)me];
        n.a = make_float4(l0[0], l0[1], l0[2], l1[0]);
        n.b = make_float4(h0[0], h0[1], h0[2], l1[1]);
        n.c = make_float4(h1[0], h1[1], h1[2], l1[2]);
        n.d = make_int4(c0, c1, parent, -1);
        return me;
    }
    void link_siblings()
    {
        for (size_t i = 0; i < nodes.size(); ++i)
        {
            const int c0 = nodes[i].d.x, c1 = nodes[i].d.y;
            if (c0 >= 0) nodes[(size_t)c0].d.w = c1;
            if (c1 >= 0) nodes[(size_t)c1].d.w = c0;
        }
    }
};

static int build_wide(rt_ctx* c, const rt_triangle* tris, int n_refs)
{
    const size_t n_bin = (size_t)(n_refs > 1 ? n_refs - 1 : 1);
    std::vector<BvhNode> bin(n_bin);
    RT_HIP(c, hipMemcpyAsync(bin.data(), c->d_nodes, n_bin * sizeof(BvhNode), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    std::vector<WideRec> recs;
    c->wide_height = collapse_wide(bin, tris, recs);
    if (3 * c->wide_height + 1 > WIDE_LDS_STACK + WIDE_OVF_STACK)
        RT_FAIL(c, RT_ERR_BVH_DEPTH, "wide BVH height %d exceeds the traversal stack (%d entries)", c->wide_height,
                WIDE_LDS_STACK + WIDE_OVF_STACK);
    c->n_wide = (int)recs.size();
    RT_HIP(c, hipMalloc(&c->d_wide, recs.size() * sizeof(WideRec)));
    RT_HIP(c, hipMemcpyAsync(c->d_wide, recs.data(), recs.size() * sizeof(WideRec), hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

/* LBVH build, see bvh.h */
static int build_bvh(rt_ctx* c, const rt_triangle* tris, int n_tris)
{
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
    RT_HIP(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    const int n = (int)count;
    c->n_tris = n;
    /* light list in index order (10_restir_di.cpp:196-205) */
    std::vector<uint32_t> lights;
    for (int i = 0; i < n; ++i)
        if (triangles[i].emissive[0] > 0.0f || triangles[i].emissive[1] > 0.0f || triangles[i].emissive[2] > 0.0f)
            lights.push_back((uint32_t)i);
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
        RT_HIP(c, hipMalloc(&c->d_lights, lights.size() * 48));
        RT_HIP(c, hipMalloc(&c->d_light_ke, lights.size() * 16));
        k_light_table<<<(c->n_lights + 255) / 256, 256, 0, c->stream>>>(c->n_lights, d_ids, c->d_tris, c->d_lights, c->d_light_ke);
        RT_HIP(c, hipGetLastError());
        RT_HIP(c, hipStreamSynchronize(c->stream));
        hipFree(d_ids);
    }
    const int rc = build_bvh(c, triangles, n);
    if (rc != RT_OK) return rc;
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
int rt_bvh_info(rt_ctx* c, uint32_t* n_refs, uint32_t* n_nodes)
{
    RT_CHECK_CTX(c);
    if (n_refs) *n_refs = (uint32_t)c->n_refs;
    if (n_nodes) *n_nodes = (uint32_t)c->n_wide;
    return RT_OK;
}

/* common/camera.hpp:11-25 (host; tan of a float argument = tanf) */
int rt_camera_lookat(rt_ctx* c, const float eye[3], const float center[3], const float up[3], float fovy)
{
    RT_CHECK_CTX(c);
    if (!eye || !center || !up) RT_FAIL(c, RT_ERR_ARG, "null camera vector");
    const f3 e = F3(eye[0], eye[1], eye[2]), ce = F3(center[0], center[1], center[2]), u0 = F3(up[0], up[1], up[2]);
    const f3 f = normalize(ce - e);
    const f3 s = normalize(cross(f, u0));
    const f3 u = cross(s, f);
    const float tanThetaY = tanf(fovy * 0.5f);
    const float tanThetaX = tanThetaY / (float)c->H * (float)c->W;
    const f3 r = s * tanThetaX, uu = u * tanThetaY;
    c->rg.origin[0] = e.x; c->rg.origin[1] = e.y; c->rg.origin[2] = e.z;
    c->rg.right[0] = r.x; c->rg.right[1] = r.y; c->rg.right[2] = r.z;
    c->rg.up[0] = uu.x; c->rg.up[1] = uu.y; c->rg.up[2] = uu.z;
    c->eye[0] = e.x; c->eye[1] = e.y; c->eye[2] = e.z;
    memcpy(c->cam_eye, eye, 12);
    memcpy(c->cam_at, center, 12);
    c->cam_fovy = fovy;
    c->has_camera = true;
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
    c->opt = *o;
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
    return c->sub0 >= 0 ? tile_grid(c->W, c->sub1 - c->sub0) : tile_grid(c->W, c->row_end - c->row_begin);
}

int rt_clear(rt_ctx* c)
{
    RT_CHECK_CTX(c);
    k_clear<<<launch_grid(c), BLOCK, 0, c->stream>>>(make_params(c, 0, 0), c->d_accum);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int rt_raycast(rt_ctx* c)
{
    RT_CHECK_CTX(c);
    NEED_SCENE(c);
    k_raycast<<<launch_grid(c), BLOCK, 0, c->stream>>>(make_scene(c), make_params(c, 0, 0, K_RAYCAST), c->d_vis, c->d_g0, c->d_g1);
    RT_HIP(c, hipGetLastError());
    c->has_gbuffer = true;
    return RT_OK;
}

static int launch_generate(rt_ctx* c, int frame, int dst_phys, int prev_phys, bool fuse)
{
    if (c->n_lights == 0 && c->opt.ris_sample_count > 0)
        RT_FAIL(c, RT_ERR_STATE, "scene has no emissive triangle (the reference divides by zero here)");
    const SceneView S = make_scene(c);
    const FrameParams P = make_params(c, frame, 0, K_GENERATE);
    const bool sh = c->opt.use_shadowed_target_function;
    float4 *orec = c->d_rec[dst_phys], *orad = c->d_rad[dst_phys];
    const float4 *prec = fuse ? c->d_rec[prev_phys] : nullptr, *prad = fuse ? c->d_rad[prev_phys] : nullptr;
    const int g = launch_grid(c);
    if (fuse && sh) k_generate_candidate<true, true><<<g, BLOCK, 0, c->stream>>>(S, P, c->d_g0, c->d_g1, prec, prad, orec, orad);
    else if (fuse) k_generate_candidate<true, false><<<g, BLOCK, 0, c->stream>>>(S, P, c->d_g0, c->d_g1, prec, prad, orec, orad);
    else if (sh) k_generate_candidate<false, true><<<g, BLOCK, 0, c->stream>>>(S, P, c->d_g0, c->d_g1, prec, prad, orec, orad);
    else k_generate_candidate<false, false><<<g, BLOCK, 0, c->stream>>>(S, P, c->d_g0, c->d_g1, prec, prad, orec, orad);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int rt_generate_candidate(rt_ctx* c, int frame, int dst)
{
    RT_CHECK_CTX(c);
    NEED_SCENE(c);
    NEED_RES(c, dst);
    if (!c->has_gbuffer) RT_FAIL(c, RT_ERR_STATE, "no G-buffer: call rt_raycast or upload RT_BUF_VISIBILITY first");
    return launch_generate(c, frame, c->res_map[dst], 0, false);
}

int rt_temporal_resampling(rt_ctx* c, int frame, int prev, int inout)
{
    RT_CHECK_CTX(c);
    NEED_SCENE(c);
    NEED_RES(c, prev);
    NEED_RES(c, inout);
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
    NEED_RES(c, src);
    NEED_RES(c, dst);
    if (src == dst) return RT_OK;
    const size_t off = (size_t)(c->row_begin - c->lrow0) * c->W;
    const size_t n = (size_t)(c->row_end - c->row_begin) * c->W;
    const int ps = c->res_map[src], pd = c->res_map[dst];
    RT_HIP(c, hipMemcpyAsync(c->d_rec[pd] + 4 * off, c->d_rec[ps] + 4 * off, n * 64, hipMemcpyDeviceToDevice, c->stream));
    RT_HIP(c, hipMemcpyAsync(c->d_rad[pd] + off, c->d_rad[ps] + off, n * 16, hipMemcpyDeviceToDevice, c->stream));
    return RT_OK;
}

static int launch_spatial(rt_ctx* c, int frame, int pass, int in_phys, int out_phys)
{
    const SceneView S = make_scene(c);
    const FrameParams P = make_params(c, frame, pass, K_SPATIAL);
    if (c->opt.use_shadowed_target_function)
        k_spatial<true><<<launch_grid(c), BLOCK, (size_t)c->tune_spatial_lds, c->stream>>>(S, P, c->d_g0, c->d_g1, c->d_rec[in_phys], c->d_rad[in_phys], c->d_rec[out_phys], c->d_rad[out_phys]);
    else
        k_spatial<false><<<launch_grid(c), BLOCK, (size_t)c->tune_spatial_lds, c->stream>>>(S, P, c->d_g0, c->d_g1, c->d_rec[in_phys], c->d_rad[in_phys], c->d_rec[out_phys], c->d_rad[out_phys]);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int rt_spatial_resampling(rt_ctx* c, int frame, int pass, int in, int out)
{
    RT_CHECK_CTX(c);
    NEED_SCENE(c);
    NEED_RES(c, in);
    NEED_RES(c, out);
    if (in == out) RT_FAIL(c, RT_ERR_ARG, "in and out must differ");
    return launch_spatial(c, frame, pass, c->res_map[in], c->res_map[out]);
}

static int launch_resolve(rt_ctx* c, int phys)
{
    k_resolve<<<launch_grid(c), BLOCK, 0, c->stream>>>(make_scene(c), make_params(c, 0, 0, K_RESOLVE), c->d_g0, c->d_g1,
                                                        c->d_rec[phys], c->d_rad[phys], c->d_accum);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}
int rt_resolve(rt_ctx* c, int res)
{
    RT_CHECK_CTX(c);
    NEED_SCENE(c);
    NEED_RES(c, res);
    return launch_resolve(c, c->res_map[res]);
}

int rt_tone_mapping(rt_ctx* c)
{
    RT_CHECK_CTX(c);
    k_tone_mapping<<<launch_grid(c), BLOCK, 0, c->stream>>>(make_params(c, 0, 0), c->d_accum, c->d_pixels);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

/* examples/07_pt/07_pt.cu (example 7) or examples/09_ris/09_ris.cu (example 9) `path_trace` */
int rt_path_trace(rt_ctx* c, int example, int frame)
{
    RT_CHECK_CTX(c);
    NEED_SCENE(c);
    if (example != 7 && example != 9) RT_FAIL(c, RT_ERR_ARG, "example must be 7 (07_pt) or 9 (09_ris)");
    if (example == 9 && c->n_lights == 0) RT_FAIL(c, RT_ERR_STATE, "09_ris needs at least one emissive triangle");
    const SceneView S = make_scene(c);
    const FrameParams P = make_params(c, frame, 0, K_RAYCAST);
    const f3 sky = F3(c->opt.sky_color[0], c->opt.sky_color[1], c->opt.sky_color[2]);
    const int g = launch_grid(c);
    const int md = c->opt.max_depth;
    RT_HIP(c, hipMemsetAsync(c->d_counter, 0, 8, c->stream));
    if (example == 7) k_path_trace<7, false><<<g, BLOCK, 0, c->stream>>>(S, P, md, sky, c->d_accum, c->d_counter);
    else if (c->opt.use_shadowed_target_function) k_path_trace<9, true><<<g, BLOCK, 0, c->stream>>>(S, P, md, sky, c->d_accum, c->d_counter);
    else k_path_trace<9, false><<<g, BLOCK, 0, c->stream>>>(S, P, md, sky, c->d_accum, c->d_counter);
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
    (void)frame;
    const int passes = c->opt.spatial_resampling_passes;
    if (stage == 0)
    {
        /* X = history, Y = candidates(+temporal) -> next history, Z = spatial ping-pong partner */
        c->fX = c->res_map[RT_RES_TEMPORAL]; c->fY = c->res_map[RT_RES_0]; c->fZ = c->res_map[RT_RES_1];
        c->f_in = c->fY; c->f_out = c->fZ;
        c->f_clear = clear_first != 0;
        c->f_stage = 0;
        return RT_OK;
    }
    if (stage != c->f_stage) RT_FAIL(c, RT_ERR_STATE, "rt_frame_stage: expected stage %d, got %d", c->f_stage, stage);
    if (stage >= 2 && stage <= passes) { c->f_in = c->f_out; c->f_out = (c->f_in == c->fZ) ? c->fX : c->fZ; }
    if (stage > passes + 1) RT_FAIL(c, RT_ERR_ARG, "bad stage %d", stage);
    return RT_OK;
}

int rt_frame_stage_run_part(rt_ctx* c, int frame, int stage, int part, int row0, int row1)
{
    RT_CHECK_CTX(c);
    NEED_SCENE(c);
    if (stage != c->f_stage) RT_FAIL(c, RT_ERR_STATE, "rt_frame_stage_run: expected stage %d, got %d", c->f_stage, stage);
    if (row0 < c->row_begin) row0 = c->row_begin;
    if (row1 > c->row_end) row1 = c->row_end;
    if (row0 >= row1) return RT_OK;
    const int passes = c->opt.spatial_resampling_passes;
    const bool whole = row0 == c->row_begin && row1 == c->row_end;
    const bool T = c->timing && whole;
    auto mark = [&](int i) { if (T && i <= 8) hipEventRecord(c->ev[i], c->stream); };
    c->sub0 = row0; c->sub1 = row1;
    int rc = RT_OK;
    if (stage == 0)
    {
        mark(0);
        if (part != 2 && c->f_clear) rc = rt_clear(c);
        mark(1);
        if (part != 2 && rc == RT_OK) rc = rt_raycast(c);
        mark(2);
        if (part != 1 && rc == RT_OK) rc = launch_generate(c, frame, c->fY, c->fX, c->opt.use_temporal_resampling != 0);
        mark(3);
    }
    else if (stage <= passes)
    {
        const int k = stage - 1;
        rc = launch_spatial(c, frame, k, c->f_in, c->f_out);
        if (k < 3) mark(4 + k);
    }
    else
    {
        for (int k = passes; k < 3; ++k) mark(4 + k);
        const int final_phys = passes > 0 ? c->f_out : c->fZ;
        rc = launch_resolve(c, final_phys);
        mark(7);
        if (rc == RT_OK) rc = rt_tone_mapping(c);
        mark(8);
    }
    c->sub0 = c->sub1 = -1;
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
    const int passes = c->opt.spatial_resampling_passes;
    if (stage <= passes) { c->f_stage = stage + 1; return RT_OK; }
    const int X = c->fX, Y = c->fY, Z = c->fZ;
    if (passes < 2)
    {
        /* logical RT_RES_0 still equals the post-temporal reservoirs: materialise the copy the
         * reference's save_temporal_reservoir makes (10_restir_di.cpp:314-321) */
        const size_t n = local_pixels(c);
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
            const int phys = c->res_map[buf - RT_BUF_RES_0];
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
    if (res >= RT_RES_PHYS && res < RT_RES_PHYS + 3) return res - RT_RES_PHYS;
    if (res >= 0 && res <= 2) return c->res_map[res];
    return -1;
}
int rt_halo_pack(rt_ctx* c, int res, int row0, int n_rows, void* device_dst)
{
    RT_CHECK_CTX(c);
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
    k_halo_flags<false><<<(n + 255) / 256, 256, 0, c->stream>>>(c->d_g1, (size_t)(row0 - c->lrow0) * c->W, n, (uint8_t*)const_cast<void*>(device_src));
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}
/* records of the neighbour on `side` that spatial pass `pass` of `frame` will gather */
int rt_halo_mark(rt_ctx* c, int frame, int pass, int side, void* device_bitmap)
{
    RT_CHECK_CTX(c);
    if (!c->has_gbuffer) RT_FAIL(c, RT_ERR_STATE, "no G-buffer yet");
    int r0, n;
    int rc = halo_side_region(c, side, &r0, &n);
    if (rc != RT_OK) return rc;
    const size_t words = rt_halo_bitmap_words(c, n);
    const int nw = (int)((words - 1) / 2);
    RT_HIP(c, hipMemsetAsync(device_bitmap, 0, words * 4, c->stream));
    /* only own rows within `halo` rows of that side can reach across */
    c->sub0 = side == 0 ? c->row_begin : (c->row_end - c->halo > c->row_begin ? c->row_end - c->halo : c->row_begin);
    c->sub1 = side == 0 ? (c->row_begin + c->halo < c->row_end ? c->row_begin + c->halo : c->row_end) : c->row_end;
    const FrameParams P = make_params(c, frame, pass, K_OTHER);
    const int grid = launch_grid(c);
    c->sub0 = c->sub1 = -1;
    k_halo_mark<<<grid, BLOCK, 0, c->stream>>>(P, c->d_g1, r0, n, (uint32_t*)device_bitmap);
    RT_HIP(c, hipGetLastError());
    k_halo_scan<<<1, 1024, 0, c->stream>>>((uint32_t*)device_bitmap, nw);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}
/* rebuild the prefix part of a bitmap received from a neighbour (rows [row0,row0+n_rows)) */
int rt_halo_scan(rt_ctx* c, int n_rows, void* device_bitmap)
{
    RT_CHECK_CTX(c);
    const int nw = (int)((rt_halo_bitmap_words(c, n_rows) - 1) / 2);
    k_halo_scan<<<1, 1024, 0, c->stream>>>((uint32_t*)device_bitmap, nw);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}
static int halo_sparse(rt_ctx* c, bool pack, int res, int row0, int n_rows, const void* device_bitmap, void* device_list)
{
    const int phys = halo_phys(c, res);
    if (phys < 0) RT_FAIL(c, RT_ERR_ARG, "bad reservoir buffer id %d", res);
    int rc = halo_range(c, row0, n_rows);
    if (rc != RT_OK) return rc;
    const int n = n_rows * c->W;
    const int nw = (int)((rt_halo_bitmap_words(c, n_rows) - 1) / 2);
    const size_t off = (size_t)(row0 - c->lrow0) * c->W;
    if (pack) k_halo_sparse<true><<<(n + 255) / 256, 256, 0, c->stream>>>((const uint32_t*)device_bitmap, nw, c->W, off, n, c->d_rec[phys], c->d_rad[phys], (float4*)device_list);
    else k_halo_sparse<false><<<(n + 255) / 256, 256, 0, c->stream>>>((const uint32_t*)device_bitmap, nw, c->W, off, n, c->d_rec[phys], c->d_rad[phys], (float4*)device_list);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}
int rt_halo_pack_sparse(rt_ctx* c, int res, int row0, int n_rows, const void* device_bitmap, void* device_dst)
{
    RT_CHECK_CTX(c);
    return halo_sparse(c, true, res, row0, n_rows, device_bitmap, device_dst);
}
int rt_halo_unpack_sparse(rt_ctx* c, int res, int row0, int n_rows, const void* device_bitmap, const void* device_src)
{
    RT_CHECK_CTX(c);
    return halo_sparse(c, false, res, row0, n_rows, device_bitmap, const_cast<void*>(device_src));
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
    if (c->opt.use_shadowed_target_function)
    {
        /* upper bound only: :115-118 (1), temporal :195-199,:224-227 (2), spatial per pass
         * count+1 (:346-350, :375-378); exact counts come from the oracle's counters */
        per_shaded += 1 + (c->opt.use_temporal_resampling ? 2 : 0) +
                      (uint64_t)c->opt.spatial_resampling_passes * (c->opt.use_spatial_resampling ? (uint64_t)c->opt.spatial_resampling_sample_count + 1 : 0);
    }
    if (rays) *rays = n + per_shaded * shaded;
    if (shaded_pixels) *shaded_pixels = shaded;
    return RT_OK;
}

int rt_spatial_bytes(rt_ctx* c, int frame, int pass, int in, uint64_t* bytes, uint64_t* accepted)
{
    RT_CHECK_CTX(c);
    NEED_RES(c, in);
    if (!c->has_gbuffer) RT_FAIL(c, RT_ERR_STATE, "no G-buffer yet");
    unsigned long long* d = nullptr;
    RT_HIP(c, hipMalloc(&d, 16));
    RT_HIP(c, hipMemsetAsync(d, 0, 16, c->stream));
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
    if (c->trace_mode == 2 || c->trace_mode == 3)
    {
        unsigned int* d_head = (unsigned int*)c->d_counter;
        RT_HIP(c, hipMemsetAsync(d_head, 0, 8, c->stream));
        const int grid = 256 * 6; /* persistent: 6 workgroups per CU (24 KB LDS each) */
        if (c->trace_mode == 2) k_trace_queue<false><<<grid, BLOCK, 0, c->stream>>>(make_scene(c).wide, d_r, (int)n, d_h, d_head);
        else k_trace_queue<true><<<grid, BLOCK, 0, c->stream>>>(make_scene(c).wide, d_r, (int)n, d_h, d_head);
    }
    else if (c->trace_mode == 0) k_trace_closest<0><<<(n + 255) / 256, 256, 0, c->stream>>>(make_scene(c), d_r, (int)n, d_h);
    else k_trace_closest<1><<<(n + 255) / 256, 256, 0, c->stream>>>(make_scene(c), d_r, (int)n, d_h);
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipMemcpyAsync(hits, d_h, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    hipFree(d_r); hipFree(d_h);
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
    if (c->trace_mode == 0) k_trace_stats<0><<<(n + 255) / 256, 256, 0, c->stream>>>(make_scene(c), d_r, (int)n, d_s);
    else k_trace_stats<1><<<(n + 255) / 256, 256, 0, c->stream>>>(make_scene(c), d_r, (int)n, d_s);
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
int rt_tuning(rt_ctx* c, int key, int value)
{
    RT_CHECK_CTX(c);
    if (key >= 0 && key <= 3 && (value == 0 || value == 1)) c->tune_tile_mode[key] = value;
    else if (key == 4 && value >= 0 && value <= 160 * 1024) c->tune_spatial_lds = value;
    else if (key == 5 && (value == 0 || value == 1)) c->bvh_builder = value; /* before rt_scene_set */
    else RT_FAIL(c, RT_ERR_ARG, "bad tuning key/value %d/%d", key, value);
    return RT_OK;
}
/* which traversal rt_trace_closest / rt_trace_stats exercise: 0 = 4-wide quantised BVH with the
 * LDS stack (the one every frame kernel uses), 1 = binary LBVH with the stackless trail. */
int rt_trace_mode(rt_ctx* c, int mode)
{
    RT_CHECK_CTX(c);
    if (mode < 0 || mode > 3) RT_FAIL(c, RT_ERR_ARG, "mode must be 0..3");
    c->trace_mode = mode;
    return RT_OK;
}

int rt_math_eval(rt_ctx* c, int fn, const float* in, uint32_t n, float* out)
{
    RT_CHECK_CTX(c);
    if (n == 0) return RT_OK;
    const size_t nin = (fn == 26) ? 2 : 1;
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
