/*
 * bvh.h — software LBVH for gfx950 (CDNA4 has no ray-tracing hardware; replaces the
 * reference's HIPRT geometry: build common/loader.hpp:68-112, traversal
 * common/raytrace.hpp:18-52).
 *
 * Build: large triangles are first cut into box fragments ("references", host, early split
 * clipping) because a Morton-order hierarchy over mixed-size triangles overlaps badly; then on
 * the device: 63-bit Morton codes of the reference centres -> radix sort (rocPRIM) -> Karras
 * 2012 hierarchy (one thread per internal node) -> bottom-up AABB refit (one launch per tree
 * level) -> 64-byte "pair" nodes holding BOTH children's boxes. Leaves name the ORIGINAL
 * triangle, so a triangle may be reached through several leaves (harmless: same t, tie rule).
 *
 * Traversal (device, per lane): stackless. A 64-bit trail word records, per level, whether
 * the sibling subtree is still pending; backtracking follows parent/sibling links stored in
 * the node (no per-lane stack in LDS or scratch => registers only, full occupancy).
 *
 * Result contract (the pinned definition of raytrace(), DESIGN.md "Oracle"): the hit is the
 * one a brute-force loop over ALL triangles with the reference's intersect_ray_triangle
 * (common/core.hpp:91-136) reports: smallest t in [tmin,tmax], ties -> highest triangle
 * index. Boxes are padded and the slab test is conservative, so the BVH only prunes.
 */
#pragma once
#include "rt_device.h"

namespace rt
{

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
