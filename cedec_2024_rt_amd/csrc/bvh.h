/*
 * bvh.h — software BVH walks for gfx950 (CDNA4 has no ray-tracing hardware; replaces the traversal half of the
 * reference's HIPRT use, common/raytrace.hpp:18-52; the build half, common/loader.hpp:68-112, is bvh_build_device.h).
 *
 * WHAT THE PRODUCT WALKS: the 4-wide tree with 8-bit quantised child boxes ("wide" records, 48 B; layout below at
 * `WideView`), built on the device by bvh_build_device.h (pre-split of large triangles, top-down binned SAH, collapse
 * to four children per record). Every walk keeps a per-lane stack, 14 entries in LDS + 50 in scratch:
 *   trace_wide          one lane = one ray, leaf tests deferred and batched (rt_trace_closest, the per-kernel entry points)
 *   occluded_ws         work-sharing any-hit walk: idle lanes take halves of busy lanes' stacks (shadow rays of
 *                       generate_candidate / resolve)
 *   closest_ws          the same for closest hits (primary rays of strips)
 *   closest_quad (r06)  [experiments library] four lanes per ray, one child box each: measured slower (profiles/r06_quad_walk_ab.txt)
 * All of them run the reference's exact intersect_ray_triangle (common/core.hpp:91-136) at the leaves; boxes are rounded
 * outward and the slab tests carry a margin, so the tree only prunes.
 *
 * Result contract (the pinned definition of raytrace()): the hit is the one a brute-force loop over ALL
 * triangles reports: smallest t in [tmin,tmax], ties -> highest triangle index (examples/04_ao/04_ao.cu:8-29).
 * tests/test_gpu_parity.py::test_lbvh_equals_brute_force holds every builder x walk to it.
 *
 * ALSO HERE, A/B ONLY (librestir_rt_exp.so, rt_trace_mode 1; rounds 1-2): the binary LBVH of round 1 — Morton codes,
 * Karras 2012 hierarchy, bottom-up refit, 64-byte "pair" nodes (`BvhNode`), walked without a stack through a 64-bit
 * trail word (`trace`). The Morton-key and refit kernels at the end of the file also serve builder 0 and the
 * pre-split's ordering. The product library refuses that trace mode.
 */
#pragma once
#include "rt_device.h"

namespace rt
{

constexpr int BLOCK_THREADS = 256; /* every kernel that traces launches 256-thread workgroups */

/* 64 B. c0/c1 >= 0: internal node index; < 0: leaf, ~c = ORIGINAL triangle index. */
struct BvhNode
{
    float4 a; /* lo0.xyz, lo1.x */
    float4 b; /* hi0.xyz, lo1.y */
    float4 c; /* hi1.xyz, lo1.z */
    int4 d;   /* child0, child1, parent, sibling */
};

struct BvhView
{
    const BvhNode* __restrict__ nodes;
    const float4* __restrict__ tv; /* 3 x float4 per triangle: v0.xyz v1.x | v1.yz v2.xy | v2.z 0 0 0 */
    int n_tris;
};

struct Hit
{
    float t, u, v;
    int prim;
};

RT_DEV void load_tri(const float4* __restrict__ tv, int i, f3& v0, f3& v1, f3& v2)
{
    const float4 t0 = tv[3 * (size_t)i + 0];
    const float4 t1 = tv[3 * (size_t)i + 1];
    const float4 t2 = tv[3 * (size_t)i + 2];
    v0 = F3(t0.x, t0.y, t0.z);
    v1 = F3(t0.w, t1.x, t1.y);
    v2 = F3(t1.z, t1.w, t2.x);
}

/* conservative slab test; (b - o) * inv keeps the subtraction exact near the planes.
 * NaN (0 * inf) falls out of v_min/v_max => that slab does not constrain. */
RT_DEV bool slab(f3 lo, f3 hi, f3 ro, f3 inv, float t0, float t1, float& tnear)
{
    const float ax = (lo.x - ro.x) * inv.x, bx = (hi.x - ro.x) * inv.x;
    const float ay = (lo.y - ro.y) * inv.y, by = (hi.y - ro.y) * inv.y;
    const float az = (lo.z - ro.z) * inv.z, bz = (hi.z - ro.z) * inv.z;
    float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    /* relative slack as a product: +-inf (ray parallel to a slab) must stay inf, `x - |x|*eps`
     * would turn it into NaN and a NaN bound accepts the box. Negative values end up on the
     * non-conservative side by eps, but tmin >= 0 makes them irrelevant. */
    tn = tn * (1.0f - 4e-7f);
    tf = tf * (1.0f + 4e-7f);
    tn = fmaxf(tn, t0);
    tf = fminf(tf, t1);
    tnear = tn;
    return tn <= tf;
}

/* ANY = true: stop at the first accepted hit (shadow rays: only the boolean is consumed,
 * common/raytrace.hpp:45-52). */
template <bool ANY, bool STATS = false>
RT_DEV bool trace(const BvhView& bvh, f3 ro, f3 rd, float tmin, float tmax, Hit& hit, uint32_t* stats = nullptr)
{
    if (bvh.n_tris <= 0) return false;
    const f3 inv = F3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    float best = tmax;
    int prim = -1;
    float bu = 0.0f, bv = 0.0f;

    int node = 0;
    unsigned long long trail = 1ull; /* sentinel; LSB = "sibling of `node` is pending" */
    for (;;)
    {
        const BvhNode* nd = bvh.nodes + node;
        if (STATS) stats[0]++;
        const float4 a = nd->a, b = nd->b, c = nd->c;
        const int4 d = nd->d;
        float t0, t1;
        bool h0 = slab(F3(a.x, a.y, a.z), F3(b.x, b.y, b.z), ro, inv, tmin, best, t0);
        bool h1 = slab(F3(a.w, b.w, c.w), F3(c.x, c.y, c.z), ro, inv, tmin, best, t1);

#pragma unroll
        for (int k = 0; k < 2; ++k)
        {
            const bool h = k ? h1 : h0;
            const int ch = k ? d.y : d.x;
            if (h && ch < 0)
            {
                const int pi = ~ch;
                if (STATS) stats[1]++;
                f3 v0, v1, v2;
                load_tri(bvh.tv, pi, v0, v1, v2);
                float t, u, v;
                if (intersect_ray_triangle(t, u, v, ro, rd, tmin, tmax, v0, v1, v2))
                {
                    if (prim < 0 || t < best || (t == best && pi > prim))
                    {
                        best = t; bu = u; bv = v; prim = pi;
                        if (ANY) { hit.t = t; hit.u = u; hit.v = v; hit.prim = pi; return true; }
                    }
                }
            }
        }
        h0 = h0 && d.x >= 0;
        h1 = h1 && d.y >= 0;
        /* a leaf hit may have shortened the interval */
        if (h0 && t0 > best) h0 = false;
        if (h1 && t1 > best) h1 = false;

        if (h0 || h1)
        {
            if (h0 && h1)
            {
                node = (t0 <= t1) ? d.x : d.y;
                trail = (trail << 1) | 1ull;
            }
            else
            {
                node = h0 ? d.x : d.y;
                trail = trail << 1;
            }
            continue;
        }
        /* backtrack */
        int parent = d.z, sibling = d.w;
        for (;;)
        {
            if (trail == 1ull) goto done;
            if (trail & 1ull)
            {
                trail ^= 1ull;
                node = sibling;
                break;
            }
            trail >>= 1;
            node = parent;
            const int4 pd = bvh.nodes[node].d;
            parent = pd.z;
            sibling = pd.w;
        }
    }
done:
    if (prim < 0) return false;
    hit.t = best; hit.u = bu; hit.v = bv; hit.prim = prim;
    return true;
}

/* =====================================================================================
 * Wide traversal structure: 4-wide BVH with 8-bit quantised child boxes, 48-byte records.
 *
 * Why: the walk is bound by vector-ALU issue: every record visited costs a
 * wavefront ~150-180 vector instructions whatever its lanes do. One record here carries 4 child
 * boxes whose planes decode with one FMA each, and a ray visits half as many records as in the
 * binary tree (21.5 vs 43 on the benchmark shadow rays), 1.5-2x faster in every tracing kernel.
 *
 * One uniform array of 48-B records; children of a node are contiguous (`base + k`):
 *   inner: Q0 = { origin.xyz, bits(ex | ey<<8 | ez<<16) }   scale_a = 2^(e_a - 127)
 *          Q1 = { base, meta, qlo_x, qlo_y }                 meta: byte k = 0 empty / 1 inner / 2 leaf
 *          Q2 = { qlo_z, qhi_x, qhi_y, qhi_z }               byte k of each word = child k
 *          child box = origin + q * scale, rounded outward at build time (only prunes).
 *   leaf : one triangle: { v0.xyz, v1.x } { v1.yz, v2.xy } { v2.z, bits(original index), 0, 0 }
 * Built by collapsing the binary tree (host SAH or device LBVH; largest-area child expanded first).
 * Traversal keeps a per-lane stack: the first WIDE_LDS_STACK entries in LDS (bank-conflict
 * free: entry i of lane t at word i*256+t), the rest in scratch (reached only by very deep trees).
 * ===================================================================================== */
/* Lanes of one wavefront pass data through LDS in the work-sharing walks (hit flags, the rank -> lane table, a victim's
 * stack slots): a wavefront executes its LDS operations in order, this keeps the COMPILER from moving or caching them. */
#define RT_WAVE_LDS_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
struct WideView
{
    const float4* __restrict__ rec; /* 3 per record */
    int n_tris;
    int n_rec;
};
/* 14 entries (r04; 24 before): with the registers the kernels need without the SLP vectoriser (61 - 80), LDS is what limits
 * the wavefronts per CU, and the work-sharing kernels' 14 + 2 rows are exactly the 4-KB record image of the cooperative
 * fetches. 24 -> 22..14: generate -4 %, resolve -6 % (one more wavefront per SIMD); 14 vs 16..20: raycast -2 %. Deeper
 * walks continue in scratch (results identical: test_deep_stack). */
#ifndef RT_WIDE_LDS_STACK
#define RT_WIDE_LDS_STACK 14
#endif
static_assert(RT_WIDE_LDS_STACK >= 14, "the work-sharing kernels stage 64 x 64-B records in their stack rows");
constexpr int WIDE_LDS_STACK = RT_WIDE_LDS_STACK;
#ifndef RT_WIDE_TOTAL_STACK
#define RT_WIDE_TOTAL_STACK 64
#endif
constexpr int WIDE_OVF_STACK = RT_WIDE_TOTAL_STACK - RT_WIDE_LDS_STACK; /* total 64 >= 3 * wide height + 1 (checked at build) */
constexpr uint32_t WIDE_LEAF_BIT = 0x80000000u;
/* slab test margin: [tn, tf] is accepted iff max(tn, tmin) <= min(tf, best) * PAD. PAD >= (1 + 4e-7) / (1 - 4e-7), the margins
 * r01-r03 put on both ends (tn * (1 - 4e-7) <= tf * (1 + 4e-7)): whatever that test kept this one keeps (tmin, best >= 0) */
constexpr float WIDE_SLAB_PAD = 1.0f + 0x1p-20f;
#ifndef RT_WIDE_STRIDE
#define RT_WIDE_STRIDE 3 /* float4 per wide record in HBM: 3 = packed 48 B, 4 = 64-B slots (a record never straddles a line) */
#endif
constexpr int WIDE_STRIDE = RT_WIDE_STRIDE;
static_assert(WIDE_STRIDE == 3 || WIDE_STRIDE == 4, "wide record stride");
#ifndef WIDE_ANY_SORTED
#define WIDE_ANY_SORTED 0 /* any-hit rays: visit children nearest-first (1) or in slot order (0) */
#endif

/* rows of the LDS array a kernel declares per thread: the stack + two rows of per-lane words used by the
 * work-sharing shadow-ray walk (occluded_ws: hit flags, thief/victim matching) */
constexpr int WIDE_LDS_ROWS = WIDE_LDS_STACK + 2;
#define WIDE_LDS_WORDS (WIDE_LDS_ROWS * BLOCK_THREADS)

RT_DEV float wide_byte(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); }

/* Leaf tests are DEFERRED: a lane that reaches a leaf parks it (two slots) and keeps walking inner
 * records; the wave runs the triangle test only when the parked leaves number at least
 * RT_LEAF_NUM/RT_LEAF_DEN of its live lanes (or no lane has inner work left). With a leaf test per ~5
 * steps per lane, testing as soon as ANY lane holds a leaf executes the ~150-instruction leaf path at
 * ~10 % lane utilisation on nearly every iteration; traversal is VALU-issue bound, so
 * batching the leaf path is worth more than the few extra boxes visited because `best` shrinks later.
 * Results do not depend on the order of the tests (closest hit with the index tie-break). */
#ifndef RT_LEAF_NUM
#define RT_LEAF_NUM 3 /* shadow rays: r01 (one lane, one ray) 1/1; with the work-sharing walk 2/3 .. 3/4 is 1 % faster (r02 sweep) */
#define RT_LEAF_DEN 4
#endif
#ifndef RT_LEAF_NUM_CLOSEST
#define RT_LEAF_NUM_CLOSEST 1
#define RT_LEAF_DEN_CLOSEST 2
#endif
#ifndef RT_LEAF_RANGE_TMAX
#define RT_LEAF_RANGE_TMAX 1 /* 1 = closest-hit leaf tests range-check against the ray's tmax (r01-r05, the default); 0 = against the best hit so
                                far (r06 A/B: same winner, no gain - see RT_NO_DEFER_BARY in frame_kernels.h) */
#endif
/* STRIDE = threads of the calling workgroup = row pitch of the LDS stack (entry i of thread t at word i*STRIDE + t) */
/* tv (closest hit, r06): the ORIGINAL triangles (3 x float4 each, as closest_ws takes them). With it the walk keeps only
 * (t, index) of the best hit and the barycentrics of the WINNER are computed once, after the walk, by the same
 * intersect_ray_triangle on the same operands (same bits) — core.hpp:131-133's two IEEE divisions (a2 / a, a0 / a: 2 x 11
 * instructions whenever any lane of the wavefront accepts a hit in a leaf pass) leave the loop, and two registers with them. */
template <bool ANY, bool STATS = false, int STRIDE = BLOCK_THREADS>
RT_DEV bool trace_wide(const WideView& bvh, uint32_t* __restrict__ lds_stack, f3 ro, f3 rd, float tmin, float tmax,
                       Hit& hit, uint32_t* stats = nullptr, const float4* __restrict__ tv = nullptr)
{
    if (bvh.n_tris <= 0) return false;
    /* finite reciprocal: an exactly axis-parallel ray must still be culled by its slab */
    f3 inv = F3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    inv.x = fminf(fmaxf(inv.x, -1e30f), 1e30f);
    inv.y = fminf(fmaxf(inv.y, -1e30f), 1e30f);
    inv.z = fminf(fmaxf(inv.z, -1e30f), 1e30f);
    /* t(q) = A + q*B is monotone in q with the sign of B = scale*inv: the entry plane of an axis is
     * the low byte plane for inv >= 0 and the high one otherwise (same values as min/max of both) */
    const bool px = inv.x >= 0.0f, py = inv.y >= 0.0f, pz = inv.z >= 0.0f;
    float best = tmax;
    int prim = -1;
    float bu = 0.0f, bv = 0.0f;

    uint32_t ovf[WIDE_OVF_STACK];
    int sp = 0;
    const int lane_slot = threadIdx.x;
    auto push = [&](uint32_t e) {
        if (sp < WIDE_LDS_STACK) lds_stack[sp * STRIDE + lane_slot] = e;
        else ovf[sp - WIDE_LDS_STACK] = e;
        ++sp;
    };
    auto pop = [&]() -> uint32_t {
        --sp;
        uint32_t e;
        if (sp < WIDE_LDS_STACK) e = lds_stack[sp * STRIDE + lane_slot];
        else e = ovf[sp - WIDE_LDS_STACK];
        return e;
    };

    constexpr uint32_t NONE = 0x7fffffffu;
    /* share of live lanes with a parked leaf that triggers the leaf pass (A/B: 1/2 for shadow rays, 1/4 for closest hit) */
    constexpr int LN = ANY ? RT_LEAF_NUM : RT_LEAF_NUM_CLOSEST, LD = ANY ? RT_LEAF_DEN : RT_LEAF_DEN_CLOSEST;
    uint32_t cur = 0u;   /* root is always an inner record */
    uint32_t pend = NONE; /* parked leaf */
    uint32_t pend2 = NONE;
    for (;;)
    {
        if ((int)cur < 0 && pend2 == NONE)
        {
            if (pend == NONE) pend = cur; else pend2 = cur;
            cur = sp ? pop() : NONE;
        }
        const bool has_inner = cur < NONE;
        const bool has_pend = pend != NONE;
        if (!has_inner && !has_pend) break;
        const unsigned long long bi = __ballot(has_inner), bp = __ballot(has_pend), ba = __ballot(true);
        const int parked = __popcll(bp) + __popcll(__ballot(pend2 != NONE));
        if (bp != 0ull && (bi == 0ull || LD * parked >= LN * __popcll(ba)))
        {
            if (STATS) stats[1] += 0x10000u; /* leaf passes this lane's wave ran while the lane was live */
            if (has_pend)
            {
                if (STATS) stats[1]++;
                const float4* g = bvh.rec + WIDE_STRIDE * (size_t)(pend & ~WIDE_LEAF_BIT);
                const float4 t0 = g[0], t1 = g[1], t2 = g[2];
                pend = pend2; pend2 = NONE;
                const f3 v0 = F3(t0.x, t0.y, t0.z), v1 = F3(t0.w, t1.x, t1.y), v2 = F3(t1.z, t1.w, t2.x);
                const int pi = as_int(t2.y);
                float t, u, v;
                /* closest hit: a candidate beyond `best` cannot win (best == tmax until the first hit), so the range test of
                 * core.hpp:103 runs against best and the three sub-areas are computed for fewer lanes; same winner */
                if (intersect_ray_triangle(t, u, v, ro, rd, tmin, (ANY || RT_LEAF_RANGE_TMAX) ? tmax : best, v0, v1, v2))
                {
                    if (prim < 0 || t < best || (t == best && pi > prim))
                    {
                        best = t; prim = pi;
                        if (ANY || !tv) { bu = u; bv = v; } /* with tv the divisions behind u, v are dead code here */
                        if (ANY) { hit.t = t; hit.u = u; hit.v = v; hit.prim = pi; return true; }
                    }
                }
            }
            continue;
        }
        if (STATS) stats[0] += 0x10000u; /* inner passes of the wave */
        if (has_inner)
        {
            if (STATS) stats[0]++;
            const float4* g = bvh.rec + WIDE_STRIDE * (size_t)cur;
            const float4 q0 = g[0], q1f = g[1], q2f = g[2];
            const uint32_t e = as_uint(q0.w);
            const uint32_t base = as_uint(q1f.x), meta = as_uint(q1f.y);
            const uint32_t lx = as_uint(q1f.z), ly = as_uint(q1f.w), lz = as_uint(q2f.x);
            const uint32_t hx = as_uint(q2f.y), hy = as_uint(q2f.z), hz = as_uint(q2f.w);
            const uint32_t nx = px ? lx : hx, ny = py ? ly : hy, nz = pz ? lz : hz;
            const uint32_t fx = px ? hx : lx, fy = py ? hy : ly, fz = pz ? hz : lz;
            const float sx = as_float((e & 0xffu) << 23), sy = as_float(((e >> 8) & 0xffu) << 23),
                        sz = as_float(((e >> 16) & 0xffu) << 23);
            /* t(q) = (origin + q*scale - ro) * inv = A + q*B : one FMA per plane */
            const float Ax = (q0.x - ro.x) * inv.x, Ay = (q0.y - ro.y) * inv.y, Az = (q0.z - ro.z) * inv.z;
            const float Bx = sx * inv.x, By = sy * inv.y, Bz = sz * inv.z;
            float td[4];
            uint32_t ce[4];
            int nhit = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const uint32_t m = (meta >> (8 * k)) & 0xffu;
                float tn = fmaxf(fmaxf(__builtin_fmaf(wide_byte(nx, k), Bx, Ax), __builtin_fmaf(wide_byte(ny, k), By, Ay)),
                                 __builtin_fmaf(wide_byte(nz, k), Bz, Az));
                float tf = fminf(fminf(__builtin_fmaf(wide_byte(fx, k), Bx, Ax), __builtin_fmaf(wide_byte(fy, k), By, Ay)),
                                 __builtin_fmaf(wide_byte(fz, k), Bz, Az));
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
                /* the scratch half of the stack is never reached in practice: one wave-uniform test
                 * keeps its addressing out of the hot path */
                const bool deep = __ballot(sp + 3 > WIDE_LDS_STACK) != 0ull;
                if (!ANY || WIDE_ANY_SORTED)
                {
                    /* sort ascending by entry distance (5 compare-exchanges); misses (3e38) sink to the end */
#define RT_CSWAP(i, j)                                                     \
    if (td[j] < td[i])                                                     \
    {                                                                      \
        const float _t = td[i]; td[i] = td[j]; td[j] = _t;                 \
        const uint32_t _e = ce[i]; ce[i] = ce[j]; ce[j] = _e;             \
    }
                    RT_CSWAP(0, 1) RT_CSWAP(2, 3) RT_CSWAP(0, 2) RT_CSWAP(1, 3) RT_CSWAP(1, 2)
#undef RT_CSWAP
                    /* ce[0] is visited next, ce[nhit-1] .. ce[1] go on the stack (ce[1] on top) */
                    if (__builtin_expect(deep, 0))
                    {
                        if (nhit > 3) push(ce[3]);
                        if (nhit > 2) push(ce[2]);
                        if (nhit > 1) push(ce[1]);
                    }
                    else if (nhit > 1)
                    {
                        uint32_t* top = lds_stack + (sp + nhit - 2) * STRIDE + lane_slot;
                        top[0] = ce[1];
                        if (nhit > 2) top[-STRIDE] = ce[2];
                        if (nhit > 3) top[-2 * STRIDE] = ce[3];
                        sp += nhit - 1;
                    }
                    cur = ce[0];
                }
                else
                {
                    /* any-hit: the order is irrelevant for the result: visit the hit children in slot order
                     * (= ascending box area, see collapse_wide): the first one next, the others pushed last-first */
                    const bool h0 = td[0] < 3.0e38f, h1 = td[1] < 3.0e38f, h2 = td[2] < 3.0e38f, h3 = td[3] < 3.0e38f;
                    cur = h0 ? ce[0] : (h1 ? ce[1] : (h2 ? ce[2] : ce[3]));
                    if (__builtin_expect(deep, 0))
                    {
                        if (h1 && h0) push(ce[1]);
                        if (h2 && (h0 || h1)) push(ce[2]);
                        if (h3 && (h0 || h1 || h2)) push(ce[3]);
                    }
                    else
                    {
                        if (h3 && (h0 || h1 || h2)) { lds_stack[sp * STRIDE + lane_slot] = ce[3]; ++sp; }
                        if (h2 && (h0 || h1)) { lds_stack[sp * STRIDE + lane_slot] = ce[2]; ++sp; }
                        if (h1 && h0) { lds_stack[sp * STRIDE + lane_slot] = ce[1]; ++sp; }
                    }
                }
            }
            else cur = sp ? pop() : NONE;
        }
    }
    if (prim < 0) return false;
    if (!ANY && tv)
    {
        f3 v0, v1, v2;
        load_tri(tv, prim, v0, v1, v2); /* the leaf record's vertices are these values (k_wide_leaves copies them) */
        float t;
        intersect_ray_triangle(t, bu, bv, ro, rd, tmin, tmax, v0, v1, v2); /* accepts again: same operands */
    }
    hit.t = best; hit.u = bu; hit.v = bv; hit.prim = prim;
    return true;
}

/* ---- Work-sharing any-hit walk (r02). A shadow ray's answer is the OR over the subtrees on its stack, so a
 * lane that has run out of work can take the bottom half of a busy lane's stack — the big, far subtrees — and walk
 * them for it: the busy lane's remaining work is split, not waited for. Idle lanes stay in the loop; every
 * RT_WS_PERIOD passes, when at least RT_WS_MIN lanes are idle, the j-th idle lane takes from the j-th lane whose
 * stack holds two or more entries (ray by shuffles, entries by reading the victim's LDS slots); whoever finds a hit
 * raises the owner's flag in LDS and the owner's other helpers drop out at the next check. Per-wavefront passes are
 * then set by the wavefront's TOTAL work / 64, not by its longest ray (a wavefront of the benchmark's shadow rays
 * runs 49 passes for rays of 20 steps on average, 397 for the slowest: profiles/r02_wave_tail.txt). The boolean is
 * exactly the single-lane walk's: the same subtrees are visited unless a hit ends the ray, and every hit is a hit.
 * LDS rows WIDE_LDS_STACK (flags) and WIDE_LDS_STACK + 1 (matching) of the caller's array are used. ---- */
#ifndef RT_WS_PERIOD
#define RT_WS_PERIOD 2 /* check for idle lanes every 2^RT_WS_PERIOD... passes: mask = (1 << RT_WS_PERIOD) - 1 */
#endif
#ifndef RT_WS_MIN
#define RT_WS_MIN 4
#endif
#ifndef RT_WS_RICH
#define RT_WS_RICH 1 /* a lane can be robbed when its stack holds at least this many entries */
#endif
#ifndef RT_WS_MULTI
#define RT_WS_MULTI 1 /* several thieves per victim (r03); 0 = one thief takes the bottom half (r02) */
#endif
template <int STRIDE = BLOCK_THREADS>
RT_DEV bool occluded_ws(const WideView& bvh, uint32_t* __restrict__ lds_generic, f3 ro, f3 rd, float tmin, float tmax,
                        uint32_t* stats = nullptr /* [0] passes of the wavefront, [1] steals by this lane | own steps << 16, [2] leaf passes, [3] own triangle tests */,
                        const bool live = true /* false: this lane has no ray of its own and only helps (returns false) */)
{
    if (bvh.n_tris <= 0) return false;
    /* an LDS-typed pointer: generic-pointer accesses in this loop (entries of another lane's slots) make the gfx950
     * backend emit an aperture test it then rejects ("V_CMP_NE_U32 0, $src_shared_base: incorrect register class") */
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* lds_stack = (lds_u32*)lds_generic;
    constexpr uint32_t NONE = 0x7fffffffu;
    const int slot = threadIdx.x, lane = threadIdx.x & 63, wave0 = threadIdx.x & ~63;
    volatile lds_u32* s_hit = lds_stack + WIDE_LDS_STACK * STRIDE;         /* [slot]: ray owned by that lane is occluded */
    volatile lds_u32* s_match = lds_stack + (WIDE_LDS_STACK + 1) * STRIDE; /* [wave0 + rank]: lane of the rank-th rich lane */
    s_hit[slot] = 0u;
    f3 inv = F3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    inv.x = fminf(fmaxf(inv.x, -1e30f), 1e30f);
    inv.y = fminf(fmaxf(inv.y, -1e30f), 1e30f);
    inv.z = fminf(fmaxf(inv.z, -1e30f), 1e30f);
    bool px = inv.x >= 0.0f, py = inv.y >= 0.0f, pz = inv.z >= 0.0f;
    int owner = slot; /* LDS slot of the lane whose ray this lane is walking */
    uint32_t ovf[WIDE_OVF_STACK];
    int sp = 0, base = 0; /* live stack entries: [base, sp) */
    auto push = [&](uint32_t e) {
        if (sp < WIDE_LDS_STACK) lds_stack[sp * STRIDE + slot] = e;
        else ovf[sp - WIDE_LDS_STACK] = e;
        ++sp;
    };
    auto pop = [&]() -> uint32_t {
        --sp;
        uint32_t e;
        if (sp < WIDE_LDS_STACK) e = lds_stack[sp * STRIDE + slot];
        else e = ovf[sp - WIDE_LDS_STACK];
        if (sp == base) { sp = 0; base = 0; }
        return e;
    };
    const unsigned long long ba = __ballot(true); /* lanes that walk a ray of their own here */
    uint32_t cur = live ? 0u : NONE, pend = NONE, pend2 = NONE;
    uint32_t pass = 0u;
    for (;;)
    {
        if ((int)cur < 0 && pend2 == NONE)
        {
            if (pend == NONE) pend = cur; else pend2 = cur;
            cur = sp > base ? pop() : NONE;
        }
        bool has_inner = cur < NONE;
        bool has_pend = pend != NONE;
        const unsigned long long bl = __ballot(has_inner || has_pend);
        if (bl == 0ull) break; /* the whole wavefront is done */
        ++pass;
        if ((pass & ((1u << RT_WS_PERIOD) - 1u)) == 0u)
        {
            /* helpers of a ray that has been settled meanwhile drop their work */
            if ((has_inner || has_pend) && s_hit[owner] != 0u)
            {
                cur = NONE; pend = NONE; pend2 = NONE; sp = 0; base = 0;
                has_inner = false; has_pend = false;
            }
            const bool idle = !has_inner && !has_pend;
            const bool rich = !idle && (sp - base) >= RT_WS_RICH && sp <= WIDE_LDS_STACK;
            const unsigned long long bi = __ballot(idle), br = __ballot(rich);
            const int nidle = __popcll(bi), nrich = __popcll(br);
#if RT_WS_MULTI
            if (nidle >= RT_WS_MIN && nrich > 0)
            {
                /* Several thieves per victim (r03): idle lane j serves rich lane j mod nrich as its (j / nrich)-th helper.
                 * A victim with n stack entries and T helpers is cut into T + 1 runs of c ~ n / (T + 1) entries from the
                 * bottom (the largest pending subtrees first); the victim keeps what is left on top. With one thief per
                 * victim a long ray spreads over the idle lanes in log2 steps of RT_WS_PERIOD passes each; this way it
                 * spreads as far as its stack allows in one step. Small integer quotients by reciprocal: operands
                 * < 128, the +0.5 guard is far above the 1-ulp error of v_rcp_f32. */
                const unsigned long long lt = (1ull << lane) - 1ull;
                const int rank_i = __popcll(bi & lt), rank_r = __popcll(br & lt);
                if (rich) s_match[wave0 + rank_r] = (uint32_t)lane;
                RT_WAVE_LDS_FENCE(); /* the table and the victims' stack slots are read by OTHER lanes below */
                const float inv_rich = __builtin_amdgcn_rcpf((float)nrich);
                const int t_idx = (int)(((float)rank_i + 0.5f) * inv_rich);
                const int v_rank = idle ? rank_i - t_idx * nrich : rank_r;
                const int helpers = v_rank < nidle ? (int)(((float)(nidle - 1 - v_rank) + 0.5f) * inv_rich) + 1 : 0;
                const int victim = idle ? (int)s_match[wave0 + v_rank] : lane;
                const float vox = __shfl(ro.x, victim), voy = __shfl(ro.y, victim), voz = __shfl(ro.z, victim);
                const float vdx = __shfl(rd.x, victim), vdy = __shfl(rd.y, victim), vdz = __shfl(rd.z, victim);
                const float vix = __shfl(inv.x, victim), viy = __shfl(inv.y, victim), viz = __shfl(inv.z, victim);
                const float vtmin = __shfl(tmin, victim), vtmax = __shfl(tmax, victim);
                const int vbase = __shfl(base, victim), vsp = __shfl(sp, victim), vowner = __shfl(owner, victim);
                const int n = vsp - vbase; /* a rich lane looks at itself: victim == lane */
                const int t_eff = helpers < n ? helpers : n;
                const int c = (int)(((float)(n + t_eff) + 0.5f) * __builtin_amdgcn_rcpf((float)(t_eff + 1)));
                if (idle)
                {
                    const int lo = t_idx * c;
                    const int hi = lo + c < n ? lo + c : n;
                    if (t_idx < t_eff && lo < n)
                    {
                        tmin = vtmin; tmax = vtmax;
                        ro = F3(vox, voy, voz); rd = F3(vdx, vdy, vdz); inv = F3(vix, viy, viz);
                        px = inv.x >= 0.0f; py = inv.y >= 0.0f; pz = inv.z >= 0.0f;
                        owner = vowner;
                        const int vslot = wave0 + victim;
                        for (int e = lo; e < hi; ++e) lds_stack[(e - lo) * STRIDE + slot] = lds_stack[(vbase + e) * STRIDE + vslot];
                        base = 0; sp = hi - lo;
                        cur = pop();
                        if (stats) stats[1] += 1u;
                    }
                }
                else if (rich && t_eff > 0)
                {
                    const int taken = t_eff * c < n ? t_eff * c : n;
                    base += taken;
                    if (base == sp) { base = 0; sp = 0; }
                }
                has_inner = cur < NONE;
            }
#else
            if (nidle >= RT_WS_MIN && nrich > 0)
            {
                const unsigned long long lt = (1ull << lane) - 1ull;
                const int rank_i = __popcll(bi & lt), rank_r = __popcll(br & lt);
                if (rich) s_match[wave0 + rank_r] = (uint32_t)lane;
                RT_WAVE_LDS_FENCE(); /* the table and the victims' stack slots are read by OTHER lanes below */
                const bool thief = idle && rank_i < nrich;
                const bool robbed = rich && rank_r < nidle;
                const int victim = thief ? (int)s_match[wave0 + rank_i] : lane;
                /* the victim's ray and stack window, read by its thief */
                const float vox = __shfl(ro.x, victim), voy = __shfl(ro.y, victim), voz = __shfl(ro.z, victim);
                const float vdx = __shfl(rd.x, victim), vdy = __shfl(rd.y, victim), vdz = __shfl(rd.z, victim);
                const float vix = __shfl(inv.x, victim), viy = __shfl(inv.y, victim), viz = __shfl(inv.z, victim);
                const float vtmin = __shfl(tmin, victim), vtmax = __shfl(tmax, victim);
                const int vbase = __shfl(base, victim), vsp = __shfl(sp, victim), vowner = __shfl(owner, victim);
                if (thief)
                {
                    tmin = vtmin; tmax = vtmax;
                    const int k = (vsp - vbase + 1) >> 1; /* the bottom half: the largest pending subtrees */
                    ro = F3(vox, voy, voz); rd = F3(vdx, vdy, vdz); inv = F3(vix, viy, viz);
                    px = inv.x >= 0.0f; py = inv.y >= 0.0f; pz = inv.z >= 0.0f;
                    owner = vowner;
                    const int vslot = wave0 + victim;
                    for (int e = 0; e < k; ++e) lds_stack[e * STRIDE + slot] = lds_stack[(vbase + e) * STRIDE + vslot];
                    base = 0; sp = k;
                    cur = pop();
                    if (stats) stats[1] += 1u;
                }
                if (robbed) { base += (sp - base + 1) >> 1; if (base == sp) { base = 0; sp = 0; } }
                has_inner = cur < NONE;
            }
#endif
        }
        const unsigned long long bi2 = __ballot(has_inner), bp = __ballot(has_pend);
        const int parked = __popcll(bp) + __popcll(__ballot(pend2 != NONE));
        if (bp != 0ull && (bi2 == 0ull || RT_LEAF_DEN * parked >= RT_LEAF_NUM * __popcll(__ballot(has_inner || has_pend))))
        {
            if (stats) stats[2] += 1u; /* leaf passes of the wavefront */
            if (has_pend)
            {
                if (stats) stats[3] += 1u; /* triangle tests by this lane */
                const float4* g = bvh.rec + WIDE_STRIDE * (size_t)(pend & ~WIDE_LEAF_BIT);
                const float4 t0 = g[0], t1 = g[1], t2 = g[2];
                pend = pend2; pend2 = NONE;
                const f3 v0 = F3(t0.x, t0.y, t0.z), v1 = F3(t0.w, t1.x, t1.y), v2 = F3(t1.z, t1.w, t2.x);
                float t, u, v;
                if (intersect_ray_triangle(t, u, v, ro, rd, tmin, tmax, v0, v1, v2))
                {
                    s_hit[owner] = 1u; /* any hit settles a shadow ray */
                    cur = NONE; pend = NONE; pend2 = NONE; sp = 0; base = 0;
                }
            }
            continue;
        }
        if (has_inner)
        {
            if (stats) stats[1] += 0x10000u;
            const float4* g = bvh.rec + WIDE_STRIDE * (size_t)cur;
            const float4 q0 = g[0], q1f = g[1], q2f = g[2];
            const uint32_t e = as_uint(q0.w);
            const uint32_t cbase = as_uint(q1f.x), meta = as_uint(q1f.y);
            const uint32_t lx = as_uint(q1f.z), ly = as_uint(q1f.w), lz = as_uint(q2f.x);
            const uint32_t hx = as_uint(q2f.y), hy = as_uint(q2f.z), hz = as_uint(q2f.w);
            const uint32_t nx = px ? lx : hx, ny = py ? ly : hy, nz = pz ? lz : hz;
            const uint32_t fx = px ? hx : lx, fy = py ? hy : ly, fz = pz ? hz : lz;
            const float sx = as_float((e & 0xffu) << 23), sy = as_float(((e >> 8) & 0xffu) << 23),
                        sz = as_float(((e >> 16) & 0xffu) << 23);
            const float Ax = (q0.x - ro.x) * inv.x, Ay = (q0.y - ro.y) * inv.y, Az = (q0.z - ro.z) * inv.z;
            const float Bx = sx * inv.x, By = sy * inv.y, Bz = sz * inv.z;
            bool h[4];
            uint32_t ce[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const uint32_t m = (meta >> (8 * k)) & 0xffu;
                float tn = fmaxf(fmaxf(__builtin_fmaf(wide_byte(nx, k), Bx, Ax), __builtin_fmaf(wide_byte(ny, k), By, Ay)),
                                 __builtin_fmaf(wide_byte(nz, k), Bz, Az));
                float tf = fminf(fminf(__builtin_fmaf(wide_byte(fx, k), Bx, Ax), __builtin_fmaf(wide_byte(fy, k), By, Ay)),
                                 __builtin_fmaf(wide_byte(fz, k), Bz, Az));
                /* conservative: rounding may leave tn a few 1e-7 too large and tf too small; one factor on the far side covers both */
                tn = fmaxf(tn, tmin);
                tf = fminf(tf, tmax) * WIDE_SLAB_PAD;
                h[k] = (m != 0u) && (tn <= tf);
                ce[k] = (cbase + (uint32_t)k) | (m == 2u ? WIDE_LEAF_BIT : 0u);
            }
            if (h[0] || h[1] || h[2] || h[3])
            {
                const bool deep = __ballot(sp + 3 > WIDE_LDS_STACK) != 0ull;
                cur = h[0] ? ce[0] : (h[1] ? ce[1] : (h[2] ? ce[2] : ce[3]));
                if (__builtin_expect(deep, 0))
                {
                    if (h[1] && h[0]) push(ce[1]);
                    if (h[2] && (h[0] || h[1])) push(ce[2]);
                    if (h[3] && (h[0] || h[1] || h[2])) push(ce[3]);
                }
                else
                {
                    if (h[3] && (h[0] || h[1] || h[2])) { lds_stack[sp * STRIDE + slot] = ce[3]; ++sp; }
                    if (h[2] && (h[0] || h[1])) { lds_stack[sp * STRIDE + slot] = ce[2]; ++sp; }
                    if (h[1] && h[0]) { lds_stack[sp * STRIDE + slot] = ce[1]; ++sp; }
                }
            }
            else cur = sp > base ? pop() : NONE;
        }
    }
    (void)ba;
    if (stats) stats[0] = pass;
    return s_hit[slot] != 0u;
}

/* ---- Work-sharing CLOSEST-hit walk (r02). Same sharing rules as occluded_ws; the answer of a ray is the minimum over
 * its subtrees of (t, then the larger triangle index), which is order-independent, so the pieces of a ray walked by
 * different lanes are merged by one 64-bit LDS min per accepted triangle on the key (bits(t) << 32 | ~index). A lane
 * culls boxes against its cached copy of the owner's best t (refreshed at every sharing check and after its own hits):
 * a stale, larger value only prunes less. At the end the owner re-intersects the winning triangle to get (t, u, v):
 * the same arithmetic on the same ray and triangle as the lane that found it. Rows WIDE_LDS_STACK .. +1 hold the
 * keys, row +2 the thief/victim table: the caller's LDS array has WIDE_LDS_ROWS_CLOSEST rows. ---- */
constexpr int WIDE_LDS_ROWS_CLOSEST = WIDE_LDS_STACK + 3;
template <int STRIDE = BLOCK_THREADS>
RT_DEV bool closest_ws(const WideView& bvh, const float4* __restrict__ tv, uint32_t* __restrict__ lds_generic, const f3 own_ro, const f3 own_rd,
                       const float own_tmin, const float own_tmax, Hit& hit)
{
    if (bvh.n_tris <= 0) return false;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    lds_u32* lds_stack = (lds_u32*)lds_generic;
    constexpr uint32_t NONE = 0x7fffffffu;
    constexpr unsigned long long NOHIT = ~0ull;
    const int slot = threadIdx.x, lane = threadIdx.x & 63, wave0 = threadIdx.x & ~63;
    lds_u64* s_key = (lds_u64*)(lds_stack + WIDE_LDS_STACK * STRIDE);  /* [slot]: best (t, index) of the ray that lane owns */
    volatile lds_u32* s_match = lds_stack + (WIDE_LDS_STACK + 2) * STRIDE;
    s_key[slot] = NOHIT;
    f3 ro = own_ro, rd = own_rd;
    float tmin = own_tmin, tmax = own_tmax;
    f3 inv = F3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    inv.x = fminf(fmaxf(inv.x, -1e30f), 1e30f);
    inv.y = fminf(fmaxf(inv.y, -1e30f), 1e30f);
    inv.z = fminf(fmaxf(inv.z, -1e30f), 1e30f);
    bool px = inv.x >= 0.0f, py = inv.y >= 0.0f, pz = inv.z >= 0.0f;
    float best = tmax; /* cached: boxes beyond it cannot hold the answer */
    int owner = slot;
    uint32_t ovf[WIDE_OVF_STACK];
    int sp = 0, base = 0;
    auto push = [&](uint32_t e) {
        if (sp < WIDE_LDS_STACK) lds_stack[sp * STRIDE + slot] = e;
        else ovf[sp - WIDE_LDS_STACK] = e;
        ++sp;
    };
    auto pop = [&]() -> uint32_t {
        --sp;
        uint32_t e;
        if (sp < WIDE_LDS_STACK) e = lds_stack[sp * STRIDE + slot];
        else e = ovf[sp - WIDE_LDS_STACK];
        if (sp == base) { sp = 0; base = 0; }
        return e;
    };
    /* order-preserving map of a float's bits onto unsigned integers (negative t included), and back */
    auto sortable = [&](float f) -> uint32_t { const uint32_t b = as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); };
    auto key_t = [&](unsigned long long k) -> float { const uint32_t u = (uint32_t)(k >> 32); return as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); };
    uint32_t cur = 0u, pend = NONE, pend2 = NONE;
    uint32_t pass = 0u;
    for (;;)
    {
        if ((int)cur < 0 && pend2 == NONE)
        {
            if (pend == NONE) pend = cur; else pend2 = cur;
            cur = sp > base ? pop() : NONE;
        }
        bool has_inner = cur < NONE;
        bool has_pend = pend != NONE;
        if (__ballot(has_inner || has_pend) == 0ull) break;
        ++pass;
        if ((pass & ((1u << RT_WS_PERIOD) - 1u)) == 0u)
        {
            /* what the other lanes have found for this ray meanwhile */
            {
                const unsigned long long k = s_key[owner];
                if (k != NOHIT) best = fminf(best, key_t(k));
            }
            const bool idle = !has_inner && !has_pend;
            const bool rich = !idle && (sp - base) >= RT_WS_RICH && sp <= WIDE_LDS_STACK;
            const unsigned long long bi = __ballot(idle), br = __ballot(rich);
            const int nidle = __popcll(bi), nrich = __popcll(br);
            if (nidle >= RT_WS_MIN && nrich > 0)
            {
                const unsigned long long lt = (1ull << lane) - 1ull;
                const int rank_i = __popcll(bi & lt), rank_r = __popcll(br & lt);
                if (rich) s_match[wave0 + rank_r] = (uint32_t)lane;
                RT_WAVE_LDS_FENCE(); /* the table and the victims' stack slots are read by OTHER lanes below */
                const bool thief = idle && rank_i < nrich;
                const bool robbed = rich && rank_r < nidle;
                const int victim = thief ? (int)s_match[wave0 + rank_i] : lane;
                const float vox = __shfl(ro.x, victim), voy = __shfl(ro.y, victim), voz = __shfl(ro.z, victim);
                const float vdx = __shfl(rd.x, victim), vdy = __shfl(rd.y, victim), vdz = __shfl(rd.z, victim);
                const float vix = __shfl(inv.x, victim), viy = __shfl(inv.y, victim), viz = __shfl(inv.z, victim);
                const float vtmin = __shfl(tmin, victim), vtmax = __shfl(tmax, victim), vbest = __shfl(best, victim);
                const int vbase = __shfl(base, victim), vsp = __shfl(sp, victim), vowner = __shfl(owner, victim);
                if (thief)
                {
                    tmin = vtmin; tmax = vtmax; best = vbest;
                    const int k = (vsp - vbase + 1) >> 1; /* the bottom half: the far subtrees */
                    ro = F3(vox, voy, voz); rd = F3(vdx, vdy, vdz); inv = F3(vix, viy, viz);
                    px = inv.x >= 0.0f; py = inv.y >= 0.0f; pz = inv.z >= 0.0f;
                    owner = vowner;
                    const int vslot = wave0 + victim;
                    for (int e = 0; e < k; ++e) lds_stack[e * STRIDE + slot] = lds_stack[(vbase + e) * STRIDE + vslot];
                    base = 0; sp = k;
                    cur = pop();
                }
                if (robbed) { base += (sp - base + 1) >> 1; if (base == sp) { base = 0; sp = 0; } }
                has_inner = cur < NONE;
            }
        }
        const unsigned long long bi2 = __ballot(has_inner), bp = __ballot(has_pend);
        const int parked = __popcll(bp) + __popcll(__ballot(pend2 != NONE));
        if (bp != 0ull && (bi2 == 0ull || RT_LEAF_DEN_CLOSEST * parked >= RT_LEAF_NUM_CLOSEST * __popcll(__ballot(has_inner || has_pend))))
        {
            if (has_pend)
            {
                const float4* g = bvh.rec + WIDE_STRIDE * (size_t)(pend & ~WIDE_LEAF_BIT);
                const float4 t0 = g[0], t1 = g[1], t2 = g[2];
                pend = pend2; pend2 = NONE;
                const f3 v0 = F3(t0.x, t0.y, t0.z), v1 = F3(t0.w, t1.x, t1.y), v2 = F3(t1.z, t1.w, t2.x);
                const int pi = as_int(t2.y);
                float t, u, v;
                if (intersect_ray_triangle(t, u, v, ro, rd, tmin, RT_LEAF_RANGE_TMAX ? tmax : fminf(tmax, best), v0, v1, v2) && t <= best) /* beyond best no winner */
                {
                    /* -0.0f and +0.0f are the same distance: one key for both */
                    const unsigned long long key = ((unsigned long long)sortable(t + 0.0f) << 32) | (unsigned long long)(~(uint32_t)pi);
                    atomicMin((unsigned long long*)&s_key[owner], key);
                    best = t;
                }
            }
            continue;
        }
        if (has_inner)
        {
            const float4* g = bvh.rec + WIDE_STRIDE * (size_t)cur;
            const float4 q0 = g[0], q1f = g[1], q2f = g[2];
            const uint32_t e = as_uint(q0.w);
            const uint32_t cbase = as_uint(q1f.x), meta = as_uint(q1f.y);
            const uint32_t lx = as_uint(q1f.z), ly = as_uint(q1f.w), lz = as_uint(q2f.x);
            const uint32_t hx = as_uint(q2f.y), hy = as_uint(q2f.z), hz = as_uint(q2f.w);
            const uint32_t nx = px ? lx : hx, ny = py ? ly : hy, nz = pz ? lz : hz;
            const uint32_t fx = px ? hx : lx, fy = py ? hy : ly, fz = pz ? hz : lz;
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
                float tn = fmaxf(fmaxf(__builtin_fmaf(wide_byte(nx, k), Bx, Ax), __builtin_fmaf(wide_byte(ny, k), By, Ay)),
                                 __builtin_fmaf(wide_byte(nz, k), Bz, Az));
                float tf = fminf(fminf(__builtin_fmaf(wide_byte(fx, k), Bx, Ax), __builtin_fmaf(wide_byte(fy, k), By, Ay)),
                                 __builtin_fmaf(wide_byte(fz, k), Bz, Az));
                /* conservative: rounding may leave tn a few 1e-7 too large and tf too small; one factor on the far side covers both */
                tn = fmaxf(tn, tmin);
                tf = fminf(tf, best) * WIDE_SLAB_PAD;
                const bool h = (m != 0u) && (tn <= tf);
                td[k] = h ? tn : 3.0e38f;
                ce[k] = (cbase + (uint32_t)k) | (m == 2u ? WIDE_LEAF_BIT : 0u);
                nhit += h ? 1 : 0;
            }
            if (nhit > 0)
            {
                const bool deep = __ballot(sp + 3 > WIDE_LDS_STACK) != 0ull;
#define RT_CSWAP(i, j)                                                     \
    if (td[j] < td[i])                                                     \
    {                                                                      \
        const float _t = td[i]; td[i] = td[j]; td[j] = _t;                 \
        const uint32_t _e = ce[i]; ce[i] = ce[j]; ce[j] = _e;             \
    }
                RT_CSWAP(0, 1) RT_CSWAP(2, 3) RT_CSWAP(0, 2) RT_CSWAP(1, 3) RT_CSWAP(1, 2)
#undef RT_CSWAP
                if (__builtin_expect(deep, 0))
                {
                    if (nhit > 3) push(ce[3]);
                    if (nhit > 2) push(ce[2]);
                    if (nhit > 1) push(ce[1]);
                }
                else if (nhit > 1)
                {
                    lds_u32* top = lds_stack + (sp + nhit - 2) * STRIDE + slot;
                    top[0] = ce[1];
                    if (nhit > 2) top[-STRIDE] = ce[2];
                    if (nhit > 3) top[-2 * STRIDE] = ce[3];
                    sp += nhit - 1;
                }
                cur = ce[0];
            }
            else cur = sp > base ? pop() : NONE;
        }
    }
    const unsigned long long k = s_key[slot];
    if (k == NOHIT) return false;
    const int prim = (int)(~(uint32_t)k);
    f3 v0, v1, v2;
    load_tri(tv, prim, v0, v1, v2);
    float t, u, v;
    intersect_ray_triangle(t, u, v, own_ro, own_rd, own_tmin, own_tmax, v0, v1, v2);
    hit.t = t; hit.u = u; hit.v = v; hit.prim = prim;
    return true;
}

#ifdef RT_EXPERIMENTS /* rt_tuning key 16 = 2 / rt_trace_mode 7: measured and left off (profiles/r06_quad_walk_ab.txt), A/B only */
/* ---- FOUR LANES PER RAY (r06): closest hit for launches that cannot fill the GPU with one lane per ray.
 * A strip's raycast is half a generation of wavefronts and lasts as long as its slowest wavefront; what sets a wavefront's duration
 * is the serial chain of walk steps of its longest ray (fetch a record, four slab tests, sort, push), and more wavefronts do not
 * lengthen the launch. Here a wavefront carries 16 rays; lane 4 r + k tests child k of ray r's current record (one slab test instead
 * of four), the quad agrees on the order by DPP (no LDS, no shuffles), and each lane can park ONE leaf of its own, so a leaf pass
 * tests up to four triangles of a ray at once. Four times the wavefronts, each step less than half as long.
 *   stack: inner records only, one column per ray: entry i of ray r at lds[i * 16 + r] (64 entries = 4 KB per wavefront: no scratch part);
 *   leaves: parked in the lane that found them; a leaf pass runs when a lane would have to park a second one, when a quarter of the
 *           live lanes hold one, or when no ray has inner work left;
 *   result: (t, index) of the best hit, identical in the four lanes; the barycentrics come from one more intersect_ray_triangle on the
 *           winner (the same operands: the same bits), as in closest_ws.
 * MEASURED (profiles/r06_quad_walk_ab.txt): it LOSES. rt_raycast on a 135-row strip of 1080p 78 us (work-sharing walk) / 95 (plain) /
 * 95 (this); 270 rows 104 / 119 / 140; a whole frame 228 / 234 / 474; rank 4 of 8 with the wire 0.325 -> 0.356 ms, at 4K 0.910 -> 1.04.
 * The timelines say why: a 16-ray wavefront lives as long as a 64-ray one (mean 22.6 against 24.1 us) because 4 x the wavefronts issue
 * 2.5 x the vector instructions and fill every slot (8 192 in flight for 55 % of the span: the launch becomes issue-bound), while the
 * one-lane walk's launch is its slowest 1 % of wavefronts (p50 26, p99 71, max 100 us; work sharing: max 83) - rays with long walks, whose
 * chain a quad halves (max 65 us) at a price the rest of the launch cannot pay.
 * Same answer as trace_wide<false>: every leaf whose box chain the ray passes is tested by the reference's intersect_ray_triangle,
 * the winner is the smallest t, ties to the highest index; boxes only prune. All four lanes of a quad must call it with the same
 * ray; has_ray = false: a quad without a ray. */
constexpr int QUAD_STACK = RT_WIDE_TOTAL_STACK;
constexpr int QUAD_LDS_WORDS = QUAD_STACK * 16; /* per wavefront */
template <int CTRL> RT_DEV float quad_perm_f(float v) { return as_float(__builtin_amdgcn_update_dpp(0, as_int(v), CTRL, 0xf, 0xf, false)); }
template <int CTRL> RT_DEV int quad_perm_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
RT_DEV bool closest_quad(const WideView& bvh, const float4* __restrict__ tv, uint32_t* __restrict__ lds_generic, const f3 ro, const f3 rd,
                         const float tmin, const float tmax, const bool has_ray, Hit& hit)
{
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* __restrict__ stack = (lds_u32*)lds_generic;
    if (bvh.n_tris <= 0) return false;
    constexpr uint32_t NONE = 0x7fffffffu;
    constexpr float MISS = 3.0e38f;
    const int lane = threadIdx.x & 63, k = lane & 3, ray = lane >> 2;
    f3 inv = F3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    inv.x = fminf(fmaxf(inv.x, -1e30f), 1e30f);
    inv.y = fminf(fmaxf(inv.y, -1e30f), 1e30f);
    inv.z = fminf(fmaxf(inv.z, -1e30f), 1e30f);
    const bool px = inv.x >= 0.0f, py = inv.y >= 0.0f, pz = inv.z >= 0.0f;
    const uint32_t sh = 8u * (uint32_t)k; /* this lane's byte of every per-child word */
    float best = tmax;
    int prim = -1;
    int sp = 0;
    uint32_t cur = has_ray ? 0u : NONE; /* the root is an inner record */
    uint32_t pend = NONE;               /* this lane's parked leaf */

    /* every lane that holds a leaf tests it; the quad then agrees on its best candidate and on the ray's best hit */
    auto leaf_pass = [&]() {
        float ct = MISS;
        int cp = -1;
        if (pend != NONE)
        {
            const float4* g = bvh.rec + WIDE_STRIDE * (size_t)pend;
            const float4 t0 = g[0], t1 = g[1], t2 = g[2];
            pend = NONE;
            const f3 v0 = F3(t0.x, t0.y, t0.z), v1 = F3(t0.w, t1.x, t1.y), v2 = F3(t1.z, t1.w, t2.x);
            float t, u, v;
            if (intersect_ray_triangle(t, u, v, ro, rd, tmin, tmax, v0, v1, v2)) { ct = t; cp = as_int(t2.y); }
        }
        /* best of the quad: smaller t, ties to the higher index (-0 == +0 as in every other walk: IEEE compares) */
        {
            const float ot = quad_perm_f<0xB1>(ct); const int op = quad_perm_i<0xB1>(cp); /* lane ^ 1 */
            if (op >= 0 && (cp < 0 || ot < ct || (ot == ct && op > cp))) { ct = ot; cp = op; }
        }
        {
            const float ot = quad_perm_f<0x4E>(ct); const int op = quad_perm_i<0x4E>(cp); /* lane ^ 2 */
            if (op >= 0 && (cp < 0 || ot < ct || (ot == ct && op > cp))) { ct = ot; cp = op; }
        }
        if (cp >= 0 && (prim < 0 || ct < best || (ct == best && cp > prim))) { best = ct; prim = cp; }
    };

    for (;;)
    {
        const bool has_inner = cur != NONE; /* the same in the four lanes of a quad */
        const bool has_pend = pend != NONE;
        const unsigned long long bi = __ballot(has_inner), bp = __ballot(has_pend);
        if ((bi | bp) == 0ull) break;
        /* live lanes: those of rays that still have anything to do (a quad with a parked leaf but no inner work counts once per lane) */
        if (bp != 0ull && (bi == 0ull || 4 * __popcll(bp) >= __popcll(bi | bp)))
        {
            leaf_pass();
            continue;
        }
        /* the four slab tests of the record, one per lane */
        float tnv = MISS;
        uint32_t child = 0u;
        bool park = false;
        if (has_inner)
        {
            const float4* g = bvh.rec + WIDE_STRIDE * (size_t)cur;
            const float4 q0 = g[0], q1f = g[1], q2f = g[2];
            const uint32_t e = as_uint(q0.w);
            const uint32_t cbase = as_uint(q1f.x), meta = as_uint(q1f.y);
            const uint32_t lx = as_uint(q1f.z), ly = as_uint(q1f.w), lz = as_uint(q2f.x);
            const uint32_t hx = as_uint(q2f.y), hy = as_uint(q2f.z), hz = as_uint(q2f.w);
            const float nxq = (float)(((px ? lx : hx) >> sh) & 0xffu), nyq = (float)(((py ? ly : hy) >> sh) & 0xffu), nzq = (float)(((pz ? lz : hz) >> sh) & 0xffu);
            const float fxq = (float)(((px ? hx : lx) >> sh) & 0xffu), fyq = (float)(((py ? hy : ly) >> sh) & 0xffu), fzq = (float)(((pz ? hz : lz) >> sh) & 0xffu);
            const float sx = as_float((e & 0xffu) << 23), sy = as_float(((e >> 8) & 0xffu) << 23), sz = as_float(((e >> 16) & 0xffu) << 23);
            const float Ax = (q0.x - ro.x) * inv.x, Ay = (q0.y - ro.y) * inv.y, Az = (q0.z - ro.z) * inv.z;
            const float Bx = sx * inv.x, By = sy * inv.y, Bz = sz * inv.z;
            const uint32_t m = (meta >> sh) & 0xffu;
            float tn = fmaxf(fmaxf(__builtin_fmaf(nxq, Bx, Ax), __builtin_fmaf(nyq, By, Ay)), __builtin_fmaf(nzq, Bz, Az));
            float tf = fminf(fminf(__builtin_fmaf(fxq, Bx, Ax), __builtin_fmaf(fyq, By, Ay)), __builtin_fmaf(fzq, Bz, Az));
            tn = fmaxf(tn, tmin);
            tf = fminf(tf, best) * WIDE_SLAB_PAD; /* the same conservative test as trace_wide */
            const bool h = (m != 0u) && (tn <= tf);
            child = cbase + (uint32_t)k;
            park = h && m == 2u;
            if (h && m != 2u) tnv = tn;
        }
        /* a lane that must park a second leaf: everybody's parked leaves are tested first (best only shrinks: the slab results stay conservative) */
        if (__ballot(park && has_pend) != 0ull) leaf_pass();
        if (park) pend = child;
        if (has_inner)
        {
            /* the quad's inner children in the order of their entry distances (ties: the lower slot first) */
            const float t0 = quad_perm_f<0x00>(tnv), t1 = quad_perm_f<0x55>(tnv), t2 = quad_perm_f<0xAA>(tnv), t3 = quad_perm_f<0xFF>(tnv);
            const int n_inner = (t0 < MISS ? 1 : 0) + (t1 < MISS ? 1 : 0) + (t2 < MISS ? 1 : 0) + (t3 < MISS ? 1 : 0);
            if (n_inner == 0)
            {
                RT_WAVE_LDS_FENCE(); /* entries written by the other lanes of the quad */
                if (sp > 0) { --sp; cur = stack[sp * 16 + ray]; }
                else cur = NONE;
            }
            else
            {
                /* rank of this lane's child among the hit inner children; MISS never precedes a hit */
                const int rank = ((t0 < tnv || (t0 == tnv && 0 < k)) ? 1 : 0) + ((t1 < tnv || (t1 == tnv && 1 < k)) ? 1 : 0) +
                                 ((t2 < tnv || (t2 == tnv && 2 < k)) ? 1 : 0) + ((t3 < tnv || (t3 == tnv && 3 < k)) ? 1 : 0);
                const bool mine = tnv < MISS;
                /* the nearest goes next; rank 1 ends on top of the stack, the farthest at the bottom of the new entries */
                if (mine && rank >= 1) stack[(sp + n_inner - 1 - rank) * 16 + ray] = child;
                sp += n_inner - 1;
                /* the nearest: every lane holds the four distances (and the record, hence cbase): no exchange */
                int js = 0;
                float tm = t0;
                if (t1 < tm) { js = 1; tm = t1; }
                if (t2 < tm) { js = 2; tm = t2; }
                if (t3 < tm) js = 3;
                cur = (child - (uint32_t)k) + (uint32_t)js;
            }
        }
    }
    if (prim < 0) return false;
    f3 v0, v1, v2;
    load_tri(tv, prim, v0, v1, v2);
    float t, u, v;
    intersect_ray_triangle(t, u, v, ro, rd, tmin, tmax, v0, v1, v2);
    hit.t = t; hit.u = u; hit.v = v; hit.prim = prim;
    return true;
}

#endif /* RT_EXPERIMENTS */

/* ---- Shadow rays as a STREAM (r02). occluded_ws still starts 64 rays together and ends with the last one; the lanes
 * that are done early can only help. Here a persistent wavefront pulls jobs (pixels) from a counter: whenever at
 * least RT_STREAM_REFILL lanes are free, those lanes hand in their finished jobs and fetch new ones, so in steady
 * state the wavefront holds rays of all ages and its passes are nearly full. When the counter runs dry the remaining
 * rays are drained with the work-sharing rules of occluded_ws (idle lanes take half of a busy lane's stack).
 *   fetch(job, ro, rd, tmin, tmax) -> bool : prepare job; false = the job needs no ray (fetch has dealt with it)
 *   finish(job, occluded)                  : consume the answer of a job that had a ray
 *   next_job(count, first, limit) -> bool  : called by ALL lanes of the wavefront together (wave-uniform result):
 *                                            jobs [first, limit), limit - first <= count, or false when none are left
 * Same answers as the one-lane walk: any-hit over the same triangles. ---- */
#ifndef RT_STREAM_REFILL
#define RT_STREAM_REFILL 24
#endif
#ifndef RT_STREAM_LEAF_NUM
#define RT_STREAM_LEAF_NUM 1 /* leaf pass when lanes with a parked leaf * DEN >= lanes with a record to visit * NUM */
#define RT_STREAM_LEAF_DEN 1
#endif
template <int STRIDE, class NextJob, class Fetch, class Finish>
RT_DEV void occluded_stream(const WideView& bvh, uint32_t* __restrict__ lds_generic, NextJob next_job, Fetch fetch, Finish finish)
{
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* lds_stack = (lds_u32*)lds_generic;
    constexpr uint32_t NONE = 0x7fffffffu;
    const int slot = threadIdx.x, lane = threadIdx.x & 63, wave0 = threadIdx.x & ~63;
    volatile lds_u32* s_hit = lds_stack + WIDE_LDS_STACK * STRIDE;
    volatile lds_u32* s_match = lds_stack + (WIDE_LDS_STACK + 1) * STRIDE;
    s_hit[slot] = 0u;
    f3 ro = F3(0.0f, 0.0f, 0.0f), rd = F3(0.0f, 0.0f, 1.0f), inv = F3(1.0f, 1.0f, 1.0f);
    float tmin = 0.0f, tmax = 0.0f;
    bool px = true, py = true, pz = true;
    int owner = slot;
    uint32_t ovf[WIDE_OVF_STACK];
    int sp = 0, base = 0;
    auto push = [&](uint32_t e) {
        if (sp < WIDE_LDS_STACK) lds_stack[sp * STRIDE + slot] = e;
        else ovf[sp - WIDE_LDS_STACK] = e;
        ++sp;
    };
    auto pop = [&]() -> uint32_t {
        --sp;
        uint32_t e;
        if (sp < WIDE_LDS_STACK) e = lds_stack[sp * STRIDE + slot];
        else e = ovf[sp - WIDE_LDS_STACK];
        if (sp == base) { sp = 0; base = 0; }
        return e;
    };
    uint32_t cur = NONE, pend = NONE, pend2 = NONE;
    int job = -1;           /* the job whose ray this lane OWNS (answer in s_hit[slot]), -1 = none */
    bool exhausted = false; /* wave-uniform: the job counter has run dry, drain with work sharing */
    uint32_t pass = 0u;
    const bool empty_scene = bvh.n_tris <= 0;
    for (;;)
    {
        if ((int)cur < 0 && pend2 == NONE)
        {
            if (pend == NONE) pend = cur; else pend2 = cur;
            cur = sp > base ? pop() : NONE;
        }
        bool has_inner = cur < NONE;
        bool has_pend = pend != NONE;
        if (!exhausted)
        {
            /* no lane helps another yet: a ray is settled when its own lane has no work left or has found a hit */
            const bool settled = job >= 0 && ((!has_inner && !has_pend) || s_hit[slot] != 0u);
            const bool free_lane = job < 0 || settled;
            const unsigned long long bf = __ballot(free_lane);
            const int n_free = __popcll(bf);
            if (n_free >= RT_STREAM_REFILL || __ballot(has_inner || has_pend) == 0ull)
            {
                if (settled)
                {
                    finish(job, s_hit[slot] != 0u);
                    job = -1;
                    cur = NONE; pend = NONE; pend2 = NONE; sp = 0; base = 0;
                }
                unsigned int first = 0u, limit = 0u;
                const bool any = next_job((unsigned int)n_free, first, limit);
                if (!any) exhausted = true;
                if (any && free_lane)
                {
                    const unsigned int mine = first + (unsigned int)__popcll(bf & ((1ull << lane) - 1ull));
                    if (mine < limit && fetch(mine, ro, rd, tmin, tmax))
                    {
                        if (empty_scene) finish(mine, false);
                        else
                        {
                            job = (int)mine;
                            s_hit[slot] = 0u;
                            inv = F3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                            inv.x = fminf(fmaxf(inv.x, -1e30f), 1e30f);
                            inv.y = fminf(fmaxf(inv.y, -1e30f), 1e30f);
                            inv.z = fminf(fmaxf(inv.z, -1e30f), 1e30f);
                            px = inv.x >= 0.0f; py = inv.y >= 0.0f; pz = inv.z >= 0.0f;
                            owner = slot;
                            cur = 0u; /* the root record */
                        }
                    }
                }
                has_inner = cur < NONE;
                has_pend = pend != NONE;
            }
        }
        const unsigned long long bl = __ballot(has_inner || has_pend);
        if (bl == 0ull)
        {
            if (exhausted) break;
            continue; /* every lane was handed a job without a ray: fetch again */
        }
        ++pass;
        if (exhausted && (pass & ((1u << RT_WS_PERIOD) - 1u)) == 0u)
        {
            /* the drain: exactly the sharing step of occluded_ws */
            if ((has_inner || has_pend) && s_hit[owner] != 0u)
            {
                cur = NONE; pend = NONE; pend2 = NONE; sp = 0; base = 0;
                has_inner = false; has_pend = false;
            }
            const bool idle = !has_inner && !has_pend;
            const bool rich = !idle && (sp - base) >= RT_WS_RICH && sp <= WIDE_LDS_STACK;
            const unsigned long long bi = __ballot(idle), br = __ballot(rich);
            const int nidle = __popcll(bi), nrich = __popcll(br);
            if (nidle >= RT_WS_MIN && nrich > 0)
            {
                const unsigned long long lt = (1ull << lane) - 1ull;
                const int rank_i = __popcll(bi & lt), rank_r = __popcll(br & lt);
                if (rich) s_match[wave0 + rank_r] = (uint32_t)lane;
                RT_WAVE_LDS_FENCE(); /* the table and the victims' stack slots are read by OTHER lanes below */
                const bool thief = idle && rank_i < nrich;
                const bool robbed = rich && rank_r < nidle;
                const int victim = thief ? (int)s_match[wave0 + rank_i] : lane;
                const float vox = __shfl(ro.x, victim), voy = __shfl(ro.y, victim), voz = __shfl(ro.z, victim);
                const float vdx = __shfl(rd.x, victim), vdy = __shfl(rd.y, victim), vdz = __shfl(rd.z, victim);
                const float vix = __shfl(inv.x, victim), viy = __shfl(inv.y, victim), viz = __shfl(inv.z, victim);
                const float vtmin = __shfl(tmin, victim), vtmax = __shfl(tmax, victim);
                const int vbase = __shfl(base, victim), vsp = __shfl(sp, victim), vowner = __shfl(owner, victim);
                if (thief)
                {
                    tmin = vtmin; tmax = vtmax;
                    const int k = (vsp - vbase + 1) >> 1;
                    ro = F3(vox, voy, voz); rd = F3(vdx, vdy, vdz); inv = F3(vix, viy, viz);
                    px = inv.x >= 0.0f; py = inv.y >= 0.0f; pz = inv.z >= 0.0f;
                    owner = vowner;
                    const int vslot = wave0 + victim;
                    for (int e = 0; e < k; ++e) lds_stack[e * STRIDE + slot] = lds_stack[(vbase + e) * STRIDE + vslot];
                    base = 0; sp = k;
                    cur = pop();
                }
                if (robbed) { base += (sp - base + 1) >> 1; if (base == sp) { base = 0; sp = 0; } }
                has_inner = cur < NONE;
            }
        }
        /* which pass: the rays of a stream have all ages, so "wait until every live lane has parked a leaf" (the rule of
         * the one-shot walks) would leave lanes with two parked leaves stuck behind every newly fetched ray; run the
         * triangle test when it has at least as many takers as the record visit */
        const unsigned long long bi2 = __ballot(has_inner), bp = __ballot(has_pend);
        if (bp != 0ull && (bi2 == 0ull || RT_STREAM_LEAF_DEN * __popcll(bp) >= RT_STREAM_LEAF_NUM * __popcll(bi2)))
        {
            if (has_pend)
            {
                const float4* g = bvh.rec + WIDE_STRIDE * (size_t)(pend & ~WIDE_LEAF_BIT);
                const float4 t0 = g[0], t1 = g[1], t2 = g[2];
                pend = pend2; pend2 = NONE;
                const f3 v0 = F3(t0.x, t0.y, t0.z), v1 = F3(t0.w, t1.x, t1.y), v2 = F3(t1.z, t1.w, t2.x);
                float t, u, v;
                if (intersect_ray_triangle(t, u, v, ro, rd, tmin, tmax, v0, v1, v2))
                {
                    s_hit[owner] = 1u;
                    cur = NONE; pend = NONE; pend2 = NONE; sp = 0; base = 0;
                }
            }
            continue;
        }
        if (has_inner)
        {
            const float4* g = bvh.rec + WIDE_STRIDE * (size_t)cur;
            const float4 q0 = g[0], q1f = g[1], q2f = g[2];
            const uint32_t e = as_uint(q0.w);
            const uint32_t cbase = as_uint(q1f.x), meta = as_uint(q1f.y);
            const uint32_t lx = as_uint(q1f.z), ly = as_uint(q1f.w), lz = as_uint(q2f.x);
            const uint32_t hx = as_uint(q2f.y), hy = as_uint(q2f.z), hz = as_uint(q2f.w);
            const uint32_t nx = px ? lx : hx, ny = py ? ly : hy, nz = pz ? lz : hz;
            const uint32_t fx = px ? hx : lx, fy = py ? hy : ly, fz = pz ? hz : lz;
            const float sx = as_float((e & 0xffu) << 23), sy = as_float(((e >> 8) & 0xffu) << 23),
                        sz = as_float(((e >> 16) & 0xffu) << 23);
            const float Ax = (q0.x - ro.x) * inv.x, Ay = (q0.y - ro.y) * inv.y, Az = (q0.z - ro.z) * inv.z;
            const float Bx = sx * inv.x, By = sy * inv.y, Bz = sz * inv.z;
            bool h[4];
            uint32_t ce[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const uint32_t m = (meta >> (8 * k)) & 0xffu;
                float tn = fmaxf(fmaxf(__builtin_fmaf(wide_byte(nx, k), Bx, Ax), __builtin_fmaf(wide_byte(ny, k), By, Ay)),
                                 __builtin_fmaf(wide_byte(nz, k), Bz, Az));
                float tf = fminf(fminf(__builtin_fmaf(wide_byte(fx, k), Bx, Ax), __builtin_fmaf(wide_byte(fy, k), By, Ay)),
                                 __builtin_fmaf(wide_byte(fz, k), Bz, Az));
                /* conservative: rounding may leave tn a few 1e-7 too large and tf too small; one factor on the far side covers both */
                tn = fmaxf(tn, tmin);
                tf = fminf(tf, tmax) * WIDE_SLAB_PAD;
                h[k] = (m != 0u) && (tn <= tf);
                ce[k] = (cbase + (uint32_t)k) | (m == 2u ? WIDE_LEAF_BIT : 0u);
            }
            if (h[0] || h[1] || h[2] || h[3])
            {
                const bool deep = __ballot(sp + 3 > WIDE_LDS_STACK) != 0ull;
                cur = h[0] ? ce[0] : (h[1] ? ce[1] : (h[2] ? ce[2] : ce[3]));
                if (__builtin_expect(deep, 0))
                {
                    if (h[1] && h[0]) push(ce[1]);
                    if (h[2] && (h[0] || h[1])) push(ce[2]);
                    if (h[3] && (h[0] || h[1] || h[2])) push(ce[3]);
                }
                else
                {
                    if (h[3] && (h[0] || h[1] || h[2])) { lds_stack[sp * STRIDE + slot] = ce[3]; ++sp; }
                    if (h[2] && (h[0] || h[1])) { lds_stack[sp * STRIDE + slot] = ce[2]; ++sp; }
                    if (h[1] && h[0]) { lds_stack[sp * STRIDE + slot] = ce[1]; ++sp; }
                }
            }
            else cur = sp > base ? pop() : NONE;
        }
    }
    /* the rays still owned when the wavefront ran out of work: their answers are in the flags */
    if (job >= 0) finish(job, s_hit[slot] != 0u);
}

/* WS: walk with occluded_ws (idle lanes of the wavefront take over part of a busy lane's stack). The answer is the
 * same bit either way (any-hit of the same ray against the same triangles); it pays where a launch is a single round
 * of wavefronts, i.e. the strips of the multi-GPU frame, and costs ~1-3 % on the full frame (rt_tuning key 13). */
/* Self-occlusion pre-test (r03). The reference's shadow ray starts 0.001 above the surface along the shading normal n0
 * (common/raytrace.hpp:42-52) and the target function takes |cos| on both ends (core.hpp:287-295), so samples BEHIND the
 * surface are as likely as samples in front of it: on the benchmark scene 31 % of the visibility-reuse rays and 26 % of the
 * resolve rays head below their own surface, and 95 % of those hit the very triangle they start from at t ~ 2e-5
 * (profiles/r03_self_occlusion.txt) — after a root-to-leaf descent of the BVH. A ray with n0 . dir < 0 therefore tests
 * the triangle of its own surface point first, with the reference's intersect_ray_triangle on the reference's vertices
 * (tv = the triangles in file order): a hit settles the any-hit query exactly as the walk would (that triangle is in the
 * tree, boxes only prune); no hit -> the usual walk. Results are unchanged; the reference ray count is unchanged. */
#ifndef RT_SELF_TEST
#define RT_SELF_TEST 1
#endif
RT_DEV bool self_occluded(const float4* __restrict__ tv, int own_tri, f3 org, f3 dir, f3 n0, bool live)
{
    const bool cand = RT_SELF_TEST && live && tv != nullptr && own_tri >= 0 && dot(n0, dir) < 0.0f;
    bool hit = false;
    if (__ballot(cand) != 0ull) /* one triangle-test pass for the wavefront, only if some lane needs it */
    {
        if (cand)
        {
            f3 v0, v1, v2;
            load_tri(tv, own_tri, v0, v1, v2);
            float t, u, v;
            hit = intersect_ray_triangle(t, u, v, org, dir, 0.0f, 0.99f, v0, v1, v2);
        }
    }
    return hit;
}
/* the same for the rays of a batch (origin p0 + 0.001 n0, direction tgt[k] - p0): mask of the needed rays the own triangle
 * occludes; one pass per round in which some lane still has a candidate ray */
template <int NR>
RT_DEV uint32_t self_occluded_mask(const float4* __restrict__ tv, int own_tri, f3 p0, f3 n0, const f3 (&tgt)[NR], uint32_t need)
{
    if (!RT_SELF_TEST || tv == nullptr) return 0u;
    uint32_t below = 0u;
#pragma unroll
    for (int k = 0; k < NR; ++k)
        if (((need >> k) & 1u) && dot(n0, tgt[k] - p0) < 0.0f) below |= 1u << k;
    if (own_tri < 0) below = 0u;
    uint32_t occ = 0u;
    if (__ballot(below != 0u) == 0ull) return 0u;
    f3 v0 = F3(0.0f, 0.0f, 0.0f), v1 = v0, v2 = v0;
    if (below) load_tri(tv, own_tri, v0, v1, v2);
    const f3 org = p0 + 0.001f * n0;
    while (__ballot(below != 0u) != 0ull)
    {
        if (below)
        {
            const int k = __ffs((int)below) - 1;
            below &= below - 1u;
            f3 t3 = tgt[0];
#pragma unroll
            for (int j = 1; j < NR; ++j)
                if (k == j) t3 = tgt[j];
            float t, u, v;
            if (intersect_ray_triangle(t, u, v, org, t3 - p0, 0.0f, 0.99f, v0, v1, v2)) occ |= 1u << k;
        }
    }
    return occ;
}

/* live (work-sharing walk only): false = the caller does not need this lane's answer (returns true); the lane joins the
 * wavefront's walk as a helper. tv / own_tri: the triangle the ray starts from, for the self-occlusion pre-test. */
template <int STRIDE = BLOCK_THREADS, bool WS = false>
RT_DEV bool check_visibility_wide(const WideView& bvh, uint32_t* __restrict__ lds_stack, f3 p0, f3 n0, f3 p1, const bool live = true,
                                  const float4* __restrict__ tv = nullptr, const int own_tri = -1)
{
    const f3 org = p0 + 0.001f * n0;
    const f3 dir = p1 - p0;
    const bool self = self_occluded(tv, own_tri, org, dir, n0, live);
    if (WS) return !occluded_ws<STRIDE>(bvh, lds_stack, org, dir, 0.0f, 0.99f, nullptr, live && !self) && !self;
    if (!live) return true;
    if (self) return false;
    Hit h;
    return !trace_wide<true, false, STRIDE>(bvh, lds_stack, org, dir, 0.0f, 0.99f, h);
}

/* Up to NR shadow rays of ONE lane from a common surface point, walked back to back: ray k is
 * check_visibility(p0, n0, tgt[k]) (origin p0 + 0.001 n0, direction tgt[k] - p0, t in [0, 0.99]) and is
 * traced iff bit k of `need` is set; returns the mask of occluded rays. A lane starts its next ray as
 * soon as its current one is settled (refill pass, batched like the leaf pass: when at least
 * RT_BATCH_REFILL lanes wait or nobody walks), so the lanes of a wavefront stay busy for the SUM of
 * their ray lengths instead of 6 times the longest (used by the shadowed-target passes, where one
 * pixel needs up to 6 independent rays). Per-ray steps and results are those of trace_wide<true>. */
#ifndef RT_BATCH_REFILL
#define RT_BATCH_REFILL 1
#endif
template <int NR, int STRIDE = BLOCK_THREADS>
RT_DEV uint32_t occluded_batch_plain(const WideView& bvh, uint32_t* __restrict__ lds_stack, f3 p0, f3 n0, const f3 (&tgt)[NR],
                                     uint32_t need)
{
    if (bvh.n_tris <= 0) return 0u;
    constexpr uint32_t NONE = 0x7fffffffu;
    const f3 ro = p0 + 0.001f * n0;
    const float tmin = 0.0f, tmax = 0.99f;
    const int lane_slot = threadIdx.x;
    uint32_t occluded = 0u;
    uint32_t ovf[WIDE_OVF_STACK];
    int sp = 0;
    auto push = [&](uint32_t e) {
        if (sp < WIDE_LDS_STACK) lds_stack[sp * STRIDE + lane_slot] = e;
        else ovf[sp - WIDE_LDS_STACK] = e;
        ++sp;
    };
    auto pop = [&]() -> uint32_t {
        --sp;
        uint32_t e;
        if (sp < WIDE_LDS_STACK) e = lds_stack[sp * STRIDE + lane_slot];
        else e = ovf[sp - WIDE_LDS_STACK];
        return e;
    };
    f3 rd = F3(0.0f, 0.0f, 1.0f), inv = F3(0.0f, 0.0f, 1.0f);
    bool px = true, py = true, pz = true;
    uint32_t ray_bit = 0u;
    uint32_t cur = NONE, pend = NONE, pend2 = NONE;
    for (;;)
    {
        if ((int)cur < 0 && pend2 == NONE)
        {
            if (pend == NONE) pend = cur; else pend2 = cur;
            cur = sp ? pop() : NONE;
        }
        const bool has_inner = cur < NONE;
        const bool has_pend = pend != NONE;
        const bool live = has_inner || has_pend;
        const bool want = !live && need != 0u;
        const unsigned long long bl = __ballot(live), bw = __ballot(want);
        if (bl == 0ull && bw == 0ull) break; /* the whole wavefront is done */
        if (bw != 0ull && (bl == 0ull || __popcll(bw) >= RT_BATCH_REFILL))
        {
            if (want)
            {
                const int k = __ffs((int)need) - 1;
                ray_bit = 1u << k;
                need &= ~ray_bit;
                f3 t = tgt[0];
#pragma unroll
                for (int j = 1; j < NR; ++j)
                    if (k == j) t = tgt[j];
                rd = t - p0;
                inv = F3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                inv.x = fminf(fmaxf(inv.x, -1e30f), 1e30f);
                inv.y = fminf(fmaxf(inv.y, -1e30f), 1e30f);
                inv.z = fminf(fmaxf(inv.z, -1e30f), 1e30f);
                px = inv.x >= 0.0f; py = inv.y >= 0.0f; pz = inv.z >= 0.0f;
                cur = 0u; sp = 0;
            }
            continue;
        }
        if (!live) continue; /* waits for the refill pass */
        const unsigned long long bi = __ballot(has_inner), bp = __ballot(has_pend);
        const int parked = __popcll(bp) + __popcll(__ballot(pend2 != NONE));
        if (bp != 0ull && (bi == 0ull || RT_LEAF_DEN * parked >= RT_LEAF_NUM * __popcll(bl)))
        {
            if (has_pend)
            {
                const float4* g = bvh.rec + WIDE_STRIDE * (size_t)(pend & ~WIDE_LEAF_BIT);
                const float4 t0 = g[0], t1 = g[1], t2 = g[2];
                pend = pend2; pend2 = NONE;
                const f3 v0 = F3(t0.x, t0.y, t0.z), v1 = F3(t0.w, t1.x, t1.y), v2 = F3(t1.z, t1.w, t2.x);
                float t, u, v;
                if (intersect_ray_triangle(t, u, v, ro, rd, tmin, tmax, v0, v1, v2))
                {
                    occluded |= ray_bit; /* any hit settles a shadow ray */
                    cur = NONE; pend = NONE; sp = 0;
                }
            }
            continue;
        }
        if (has_inner)
        {
            const float4* g = bvh.rec + WIDE_STRIDE * (size_t)cur;
            const float4 q0 = g[0], q1f = g[1], q2f = g[2];
            const uint32_t e = as_uint(q0.w);
            const uint32_t base = as_uint(q1f.x), meta = as_uint(q1f.y);
            const uint32_t lx = as_uint(q1f.z), ly = as_uint(q1f.w), lz = as_uint(q2f.x);
            const uint32_t hx = as_uint(q2f.y), hy = as_uint(q2f.z), hz = as_uint(q2f.w);
            const uint32_t nx = px ? lx : hx, ny = py ? ly : hy, nz = pz ? lz : hz;
            const uint32_t fx = px ? hx : lx, fy = py ? hy : ly, fz = pz ? hz : lz;
            const float sx = as_float((e & 0xffu) << 23), sy = as_float(((e >> 8) & 0xffu) << 23),
                        sz = as_float(((e >> 16) & 0xffu) << 23);
            const float Ax = (q0.x - ro.x) * inv.x, Ay = (q0.y - ro.y) * inv.y, Az = (q0.z - ro.z) * inv.z;
            const float Bx = sx * inv.x, By = sy * inv.y, Bz = sz * inv.z;
            bool h[4];
            uint32_t ce[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const uint32_t m = (meta >> (8 * k)) & 0xffu;
                float tn = fmaxf(fmaxf(__builtin_fmaf(wide_byte(nx, k), Bx, Ax), __builtin_fmaf(wide_byte(ny, k), By, Ay)),
                                 __builtin_fmaf(wide_byte(nz, k), Bz, Az));
                float tf = fminf(fminf(__builtin_fmaf(wide_byte(fx, k), Bx, Ax), __builtin_fmaf(wide_byte(fy, k), By, Ay)),
                                 __builtin_fmaf(wide_byte(fz, k), Bz, Az));
                /* conservative: rounding may leave tn a few 1e-7 too large and tf too small; one factor on the far side covers both */
                tn = fmaxf(tn, tmin);
                tf = fminf(tf, tmax) * WIDE_SLAB_PAD;
                h[k] = (m != 0u) && (tn <= tf);
                ce[k] = (base + (uint32_t)k) | (m == 2u ? WIDE_LEAF_BIT : 0u);
            }
            if (h[0] || h[1] || h[2] || h[3])
            {
                const bool deep = __ballot(sp + 3 > WIDE_LDS_STACK) != 0ull;
                cur = h[0] ? ce[0] : (h[1] ? ce[1] : (h[2] ? ce[2] : ce[3]));
                if (__builtin_expect(deep, 0))
                {
                    if (h[1] && h[0]) push(ce[1]);
                    if (h[2] && (h[0] || h[1])) push(ce[2]);
                    if (h[3] && (h[0] || h[1] || h[2])) push(ce[3]);
                }
                else
                {
                    if (h[3] && (h[0] || h[1] || h[2])) { lds_stack[sp * STRIDE + lane_slot] = ce[3]; ++sp; }
                    if (h[2] && (h[0] || h[1])) { lds_stack[sp * STRIDE + lane_slot] = ce[2]; ++sp; }
                    if (h[1] && h[0]) { lds_stack[sp * STRIDE + lane_slot] = ce[1]; ++sp; }
                }
            }
            else cur = sp ? pop() : NONE;
        }
    }
    return occluded;
}

/* The same with work sharing (r02, as occluded_ws): a lane whose batch is exhausted takes the bottom half of the stack
 * of a lane that is still walking, with that lane's ray; hits go to a per-owner bit mask in LDS (bit k = ray k is
 * occluded), so an owner may move on to its next ray while pieces of the previous one are still being walked by
 * helpers. Rows WIDE_LDS_STACK and WIDE_LDS_STACK + 1 of the caller's array are used (WIDE_LDS_ROWS rows). */
#ifndef RT_BATCH_WS
#define RT_BATCH_WS 1
#endif
template <int NR, int STRIDE = BLOCK_THREADS>
RT_DEV uint32_t occluded_batch_ws(const WideView& bvh, uint32_t* __restrict__ lds_generic, f3 p0, f3 n0, const f3 (&tgt)[NR], uint32_t need)
{
    if (bvh.n_tris <= 0) return 0u;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* lds_stack = (lds_u32*)lds_generic;
    constexpr uint32_t NONE = 0x7fffffffu;
    const int slot = threadIdx.x, lane = threadIdx.x & 63, wave0 = threadIdx.x & ~63;
    volatile lds_u32* s_hit = lds_stack + WIDE_LDS_STACK * STRIDE;         /* [slot]: mask of occluded rays of the lane that owns them */
    volatile lds_u32* s_match = lds_stack + (WIDE_LDS_STACK + 1) * STRIDE;
    s_hit[slot] = 0u;
    f3 ro = p0 + 0.001f * n0;
    const float tmin = 0.0f, tmax = 0.99f;
    uint32_t ovf[WIDE_OVF_STACK];
    int sp = 0, base = 0;
    auto push = [&](uint32_t e) {
        if (sp < WIDE_LDS_STACK) lds_stack[sp * STRIDE + slot] = e;
        else ovf[sp - WIDE_LDS_STACK] = e;
        ++sp;
    };
    auto pop = [&]() -> uint32_t {
        --sp;
        uint32_t e;
        if (sp < WIDE_LDS_STACK) e = lds_stack[sp * STRIDE + slot];
        else e = ovf[sp - WIDE_LDS_STACK];
        if (sp == base) { sp = 0; base = 0; }
        return e;
    };
    f3 rd = F3(0.0f, 0.0f, 1.0f), inv = F3(0.0f, 0.0f, 1.0f);
    bool px = true, py = true, pz = true;
    uint32_t ray_bit = 0u;
    int owner = slot;
    uint32_t cur = NONE, pend = NONE, pend2 = NONE;
    uint32_t pass = 0u;
    for (;;)
    {
        if ((int)cur < 0 && pend2 == NONE)
        {
            if (pend == NONE) pend = cur; else pend2 = cur;
            cur = sp > base ? pop() : NONE;
        }
        bool has_inner = cur < NONE;
        bool has_pend = pend != NONE;
        bool live = has_inner || has_pend;
        const bool want = !live && need != 0u;
        unsigned long long bl = __ballot(live);
        const unsigned long long bw = __ballot(want);
        if (bl == 0ull && bw == 0ull) break; /* the whole wavefront is done */
        if (bw != 0ull && (bl == 0ull || __popcll(bw) >= RT_BATCH_REFILL))
        {
            if (want)
            {
                const int k = __ffs((int)need) - 1;
                ray_bit = 1u << k;
                need &= ~ray_bit;
                f3 t = tgt[0];
#pragma unroll
                for (int j = 1; j < NR; ++j)
                    if (k == j) t = tgt[j];
                rd = t - p0;
                inv = F3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                inv.x = fminf(fmaxf(inv.x, -1e30f), 1e30f);
                inv.y = fminf(fmaxf(inv.y, -1e30f), 1e30f);
                inv.z = fminf(fmaxf(inv.z, -1e30f), 1e30f);
                px = inv.x >= 0.0f; py = inv.y >= 0.0f; pz = inv.z >= 0.0f;
                cur = 0u; sp = 0; base = 0;
            }
            continue;
        }
        ++pass;
        if ((pass & ((1u << RT_WS_PERIOD) - 1u)) == 0u)
        {
            /* a ray that has been settled meanwhile: whoever still walks a piece of it drops it */
            if (live && (s_hit[owner] & ray_bit) != 0u)
            {
                cur = NONE; pend = NONE; pend2 = NONE; sp = 0; base = 0;
                has_inner = false; has_pend = false; live = false;
            }
            const bool idle = !live && need == 0u;
            const bool rich = live && (sp - base) >= RT_WS_RICH && sp <= WIDE_LDS_STACK;
            const unsigned long long bi = __ballot(idle), br = __ballot(rich);
            const int nidle = __popcll(bi), nrich = __popcll(br);
            if (nidle >= RT_WS_MIN && nrich > 0)
            {
                const unsigned long long lt = (1ull << lane) - 1ull;
                const int rank_i = __popcll(bi & lt), rank_r = __popcll(br & lt);
                if (rich) s_match[wave0 + rank_r] = (uint32_t)lane;
                RT_WAVE_LDS_FENCE(); /* the table and the victims' stack slots are read by OTHER lanes below */
                const bool thief = idle && rank_i < nrich;
                const bool robbed = rich && rank_r < nidle;
                const int victim = thief ? (int)s_match[wave0 + rank_i] : lane;
                const float vox = __shfl(ro.x, victim), voy = __shfl(ro.y, victim), voz = __shfl(ro.z, victim);
                const float vdx = __shfl(rd.x, victim), vdy = __shfl(rd.y, victim), vdz = __shfl(rd.z, victim);
                const float vix = __shfl(inv.x, victim), viy = __shfl(inv.y, victim), viz = __shfl(inv.z, victim);
                const int vbase = __shfl(base, victim), vsp = __shfl(sp, victim), vowner = __shfl(owner, victim);
                const uint32_t vbit = (uint32_t)__shfl((int)ray_bit, victim);
                if (thief)
                {
                    const int k = (vsp - vbase + 1) >> 1;
                    ro = F3(vox, voy, voz); rd = F3(vdx, vdy, vdz); inv = F3(vix, viy, viz);
                    px = inv.x >= 0.0f; py = inv.y >= 0.0f; pz = inv.z >= 0.0f;
                    owner = vowner; ray_bit = vbit;
                    const int vslot = wave0 + victim;
                    for (int e = 0; e < k; ++e) lds_stack[e * STRIDE + slot] = lds_stack[(vbase + e) * STRIDE + vslot];
                    base = 0; sp = k;
                    cur = pop();
                }
                if (robbed) { base += (sp - base + 1) >> 1; if (base == sp) { base = 0; sp = 0; } }
                has_inner = cur < NONE;
                live = has_inner || has_pend;
                bl = __ballot(live);
            }
        }
        if (!live) continue; /* waits for the refill pass, or for work to take over */
        const unsigned long long bi = __ballot(has_inner), bp = __ballot(has_pend);
        const int parked = __popcll(bp) + __popcll(__ballot(pend2 != NONE));
        if (bp != 0ull && (bi == 0ull || RT_LEAF_DEN * parked >= RT_LEAF_NUM * __popcll(bl)))
        {
            if (has_pend)
            {
                const float4* g = bvh.rec + WIDE_STRIDE * (size_t)(pend & ~WIDE_LEAF_BIT);
                const float4 t0 = g[0], t1 = g[1], t2 = g[2];
                pend = pend2; pend2 = NONE;
                const f3 v0 = F3(t0.x, t0.y, t0.z), v1 = F3(t0.w, t1.x, t1.y), v2 = F3(t1.z, t1.w, t2.x);
                float t, u, v;
                if (intersect_ray_triangle(t, u, v, ro, rd, tmin, tmax, v0, v1, v2))
                {
                    atomicOr((uint32_t*)&s_hit[owner], ray_bit); /* any hit settles a shadow ray */
                    cur = NONE; pend = NONE; pend2 = NONE; sp = 0; base = 0;
                }
            }
            continue;
        }
        if (has_inner)
        {
            const float4* g = bvh.rec + WIDE_STRIDE * (size_t)cur;
            const float4 q0 = g[0], q1f = g[1], q2f = g[2];
            const uint32_t e = as_uint(q0.w);
            const uint32_t cbase = as_uint(q1f.x), meta = as_uint(q1f.y);
            const uint32_t lx = as_uint(q1f.z), ly = as_uint(q1f.w), lz = as_uint(q2f.x);
            const uint32_t hx = as_uint(q2f.y), hy = as_uint(q2f.z), hz = as_uint(q2f.w);
            const uint32_t nx = px ? lx : hx, ny = py ? ly : hy, nz = pz ? lz : hz;
            const uint32_t fx = px ? hx : lx, fy = py ? hy : ly, fz = pz ? hz : lz;
            const float sx = as_float((e & 0xffu) << 23), sy = as_float(((e >> 8) & 0xffu) << 23),
                        sz = as_float(((e >> 16) & 0xffu) << 23);
            const float Ax = (q0.x - ro.x) * inv.x, Ay = (q0.y - ro.y) * inv.y, Az = (q0.z - ro.z) * inv.z;
            const float Bx = sx * inv.x, By = sy * inv.y, Bz = sz * inv.z;
            bool h[4];
            uint32_t ce[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const uint32_t m = (meta >> (8 * k)) & 0xffu;
                float tn = fmaxf(fmaxf(__builtin_fmaf(wide_byte(nx, k), Bx, Ax), __builtin_fmaf(wide_byte(ny, k), By, Ay)),
                                 __builtin_fmaf(wide_byte(nz, k), Bz, Az));
                float tf = fminf(fminf(__builtin_fmaf(wide_byte(fx, k), Bx, Ax), __builtin_fmaf(wide_byte(fy, k), By, Ay)),
                                 __builtin_fmaf(wide_byte(fz, k), Bz, Az));
                /* conservative: rounding may leave tn a few 1e-7 too large and tf too small; one factor on the far side covers both */
                tn = fmaxf(tn, tmin);
                tf = fminf(tf, tmax) * WIDE_SLAB_PAD;
                h[k] = (m != 0u) && (tn <= tf);
                ce[k] = (cbase + (uint32_t)k) | (m == 2u ? WIDE_LEAF_BIT : 0u);
            }
            if (h[0] || h[1] || h[2] || h[3])
            {
                const bool deep = __ballot(sp + 3 > WIDE_LDS_STACK) != 0ull;
                cur = h[0] ? ce[0] : (h[1] ? ce[1] : (h[2] ? ce[2] : ce[3]));
                if (__builtin_expect(deep, 0))
                {
                    if (h[1] && h[0]) push(ce[1]);
                    if (h[2] && (h[0] || h[1])) push(ce[2]);
                    if (h[3] && (h[0] || h[1] || h[2])) push(ce[3]);
                }
                else
                {
                    if (h[3] && (h[0] || h[1] || h[2])) { lds_stack[sp * STRIDE + slot] = ce[3]; ++sp; }
                    if (h[2] && (h[0] || h[1])) { lds_stack[sp * STRIDE + slot] = ce[2]; ++sp; }
                    if (h[1] && h[0]) { lds_stack[sp * STRIDE + slot] = ce[1]; ++sp; }
                }
            }
            else cur = sp > base ? pop() : NONE;
        }
    }
    return s_hit[slot];
}
template <int NR, int STRIDE = BLOCK_THREADS>
RT_DEV uint32_t occluded_batch(const WideView& bvh, uint32_t* __restrict__ lds_stack, f3 p0, f3 n0, const f3 (&tgt)[NR], uint32_t need,
                               const float4* __restrict__ tv = nullptr, const int own_tri = -1)
{
    /* rays that head below their own surface: the triangle they start from first (see self_occluded) */
    const uint32_t self = self_occluded_mask<NR>(tv, own_tri, p0, n0, tgt, need);
    need &= ~self;
#if RT_BATCH_WS
    return self | occluded_batch_ws<NR, STRIDE>(bvh, lds_stack, p0, n0, tgt, need);
#else
    return self | occluded_batch_plain<NR, STRIDE>(bvh, lds_stack, p0, n0, tgt, need);
#endif
}

/* common/core.hpp:32-36 + common/raytrace.hpp:45-52: 1 = visible, 0 = occluded */
RT_DEV bool check_visibility(const BvhView& bvh, f3 p0, f3 n0, f3 p1)
{
    const f3 org = p0 + 0.001f * n0;
    const f3 dir = p1 - p0;
    Hit h;
    return !trace<true>(bvh, org, dir, 0.0f, 0.99f, h);
}

/* ================================================================ build kernels */

RT_DEV uint64_t expand21(uint32_t v)
{
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

/* per triangle: traversal vertex records */
__global__ void k_bvh_tv(const float* __restrict__ tris /* 15 floats each */, int n, float4* __restrict__ tv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* t = tris + 15 * (size_t)i;
    tv[3 * (size_t)i + 0] = make_float4(t[0], t[1], t[2], t[3]);
    tv[3 * (size_t)i + 1] = make_float4(t[4], t[5], t[6], t[7]);
    tv[3 * (size_t)i + 2] = make_float4(t[8], 0.0f, 0.0f, 0.0f);
}

/* per reference (a triangle or a fragment of a pre-split triangle, see build_bvh): Morton key of
 * the centre of its (already padded) box */
__global__ void k_bvh_keys(const float* __restrict__ boxes /* 6 per ref */, int n, float3 slo, float3 sext,
                           uint64_t* __restrict__ keys, uint32_t* __restrict__ ids)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* bx = boxes + 6 * (size_t)i;
    const float cx = ((bx[0] + bx[3]) * 0.5f - slo.x) / sext.x;
    const float cy = ((bx[1] + bx[4]) * 0.5f - slo.y) / sext.y;
    const float cz = ((bx[2] + bx[5]) * 0.5f - slo.z) / sext.z;
    const uint32_t qx = (uint32_t)fminf(fmaxf(cx * 2097152.0f, 0.0f), 2097151.0f);
    const uint32_t qy = (uint32_t)fminf(fmaxf(cy * 2097152.0f, 0.0f), 2097151.0f);
    const uint32_t qz = (uint32_t)fminf(fmaxf(cz * 2097152.0f, 0.0f), 2097151.0f);
    keys[i] = (expand21(qx) << 2) | (expand21(qy) << 1) | expand21(qz);
    ids[i] = (uint32_t)i;
}

RT_DEV int lbvh_delta(const uint64_t* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

/* Karras 2012: internal node i in [0, n-2]. links: child0, child1 (>=0 internal, <0 ~leaf
 * position in sorted order), parents of internal nodes and of leaves. */
__global__ void k_bvh_hierarchy(const uint64_t* __restrict__ keys, int n, int2* __restrict__ children,
                                int* __restrict__ parent_inner, int* __restrict__ parent_leaf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvh_delta(keys, n, i, j);
    int s = 0;
    int t = l;
    do
    {
        t = (t + 1) >> 1;
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    int c0, c1;
    if (lo == gamma) { c0 = ~gamma; parent_leaf[gamma] = i; }
    else { c0 = gamma; parent_inner[gamma] = i; }
    if (hi == gamma + 1) { c1 = ~(gamma + 1); parent_leaf[gamma + 1] = i; }
    else { c1 = gamma + 1; parent_inner[gamma + 1] = i; }
    children[i] = make_int2(c0, c1);
    if (i == 0) parent_inner[0] = -1;
}

/* Bottom-up refit as a sequence of launches: pass k computes every internal node whose two
 * children were finished by passes < k. Kernel boundaries give the cross-XCD visibility the
 * classic single-launch "atomic arrival counter" refit would need agent-scope fences for;
 * the build is one-off and outside the timed region. level[i] = pass that finished node i
 * (0 = pending); the root's level is the tree height (trail-word capacity check). */
RT_DEV const float* bvh_child_box(int ch, const uint32_t* __restrict__ ids, const float* __restrict__ prim_boxes,
                                  const float* __restrict__ node_boxes)
{
    return ch < 0 ? prim_boxes + 6 * (size_t)ids[~ch] : node_boxes + 6 * (size_t)ch;
}
__global__ void k_bvh_refit_pass(int n, int pass, const uint32_t* __restrict__ ids,
                                 const float* __restrict__ prim_boxes, const int2* __restrict__ children,
                                 float* __restrict__ node_boxes, int* __restrict__ level,
                                 int* __restrict__ remaining)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    if (level[i] != 0) return;
    const int2 ch = children[i];
    const int l0 = ch.x < 0 ? 1 : level[ch.x];
    const int l1 = ch.y < 0 ? 1 : level[ch.y];
    if (l0 == 0 || l1 == 0 || l0 > pass || l1 > pass) { atomicAdd(remaining, 1); return; }
    const float* b0 = bvh_child_box(ch.x, ids, prim_boxes, node_boxes);
    const float* b1 = bvh_child_box(ch.y, ids, prim_boxes, node_boxes);
    float* nb = node_boxes + 6 * (size_t)i;
    for (int k = 0; k < 3; ++k)
    {
        nb[k] = fminf(b0[k], b1[k]);
        nb[3 + k] = fmaxf(b0[3 + k], b1[3 + k]);
    }
    level[i] = pass + 1;
}

/* emit traversal nodes */
__global__ void k_bvh_emit(int n, const uint32_t* __restrict__ ids, const int* __restrict__ ref_tri,
                           const float* __restrict__ prim_boxes, const int2* __restrict__ children,
                           const int* __restrict__ parent_inner, const float* __restrict__ node_boxes,
                           BvhNode* __restrict__ nodes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int2 ch = children[i];
    const float* b0 = bvh_child_box(ch.x, ids, prim_boxes, node_boxes);
    const float* b1 = bvh_child_box(ch.y, ids, prim_boxes, node_boxes);
    BvhNode nd;
    nd.a = make_float4(b0[0], b0[1], b0[2], b1[0]);
    nd.b = make_float4(b0[3], b0[4], b0[5], b1[1]);
    nd.c = make_float4(b1[3], b1[4], b1[5], b1[2]);
    const int c0 = ch.x < 0 ? ~ref_tri[ids[~ch.x]] : ch.x;
    const int c1 = ch.y < 0 ? ~ref_tri[ids[~ch.y]] : ch.y;
    const int parent = parent_inner[i];
    int sibling = -1;
    if (parent >= 0)
    {
        const int2 pc = children[parent];
        sibling = (pc.x == i) ? pc.y : pc.x; /* may be a leaf (<0): never followed, see trace() */
    }
    nd.d = make_int4(c0, c1, parent, sibling);
    nodes[i] = nd;
}

}  // namespace rt
