/*
 * bvh_build_device.h — the BVH build entirely on the GPU (included once, by restir_rt.hip; replaces
 * the reference's hiprtBuildGeometry, common/loader.hpp:68-112; the host SAH builder of
 * bvh_build_host.h stays as the high-quality alternative).
 *
 *   1. early split clipping of large triangles into box fragments ("references"): count pass,
 *      exclusive scan, emit pass (the recursion of bvh_build_host.h::split_refs with a per-thread stack);
 *   2. 63-bit Morton keys of the reference centres, radix sort (rocPRIM)            [kernels in bvh.h]
 *   3. binary hierarchy by PLOC — parallel locally-ordered clustering (Meister & Bittner 2018): every
 *      cluster looks RADIUS places left and right in Morton order for the neighbour whose union box is
 *      smallest, mutual nearest neighbours merge, the array is compacted, repeat until one cluster is
 *      left. Near-SAH quality without a top-down pass; boxes come with the merges, so there is no refit.
 *      (The Karras 2012 hierarchy + level-by-level refit of round 1 remains as builder 0.)
 *   4. collapse into the 4-wide quantised records the kernels traverse, level by level on the device.
 * No host round trip carries tree data; the host reads back a handful of counters (reference count,
 * clusters left, record count, heights).
 */
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "bvh.h"

namespace rt
{

/* ------------------------------------------------------------------ 1. early split clipping */
struct DevPoly
{
    int n;
    float v[12][3];
};
RT_DEV void dpoly_bounds(const DevPoly& p, float* lo, float* hi)
{
    for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    for (int i = 0; i < p.n; ++i)
        for (int a = 0; a < 3; ++a)
        {
            lo[a] = fminf(lo[a], p.v[i][a]);
            hi[a] = fmaxf(hi[a], p.v[i][a]);
        }
}
RT_DEV void dpoly_clip(const DevPoly& p, int a, float s, int sign, DevPoly& o)
{
    o.n = 0;
    for (int i = 0; i < p.n; ++i)
    {
        const float* c = p.v[i];
        const float* d = p.v[(i + 1) % p.n];
        const bool cin = sign > 0 ? c[a] >= s : c[a] <= s;
        const bool din = sign > 0 ? d[a] >= s : d[a] <= s;
        if (cin && o.n < 12) { o.v[o.n][0] = c[0]; o.v[o.n][1] = c[1]; o.v[o.n][2] = c[2]; o.n++; }
        if (cin != din && o.n < 12)
        {
            const float t = (s - c[a]) / (d[a] - c[a]);
            for (int k = 0; k < 3; ++k) o.v[o.n][k] = c[k] + (d[k] - c[k]) * t;
            o.v[o.n][a] = s;
            o.n++;
        }
    }
}
constexpr int SPLIT_STACK = 20;
/* EMIT = false: counts[i] = fragments of triangle i; EMIT = true: writes them at offsets[i] */
template <bool EMIT>
__global__ void k_split_refs(const float* __restrict__ tris /* 15 floats each */, int n, float L, float pad,
                             const uint32_t* __restrict__ offsets, uint32_t* __restrict__ counts,
                             float* __restrict__ boxes /* 6 per reference */, int* __restrict__ ref_tri)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    DevPoly stack[SPLIT_STACK];
    int sp = 0;
    {
        const float* t = tris + 15 * (size_t)i;
        stack[0].n = 3;
        for (int k = 0; k < 3; ++k)
            for (int a = 0; a < 3; ++a) stack[0].v[k][a] = t[3 * k + a];
        sp = 1;
    }
    uint32_t emitted = 0;
    const uint32_t base = EMIT ? offsets[i] : 0u;
    while (sp > 0)
    {
        const DevPoly q = stack[--sp];
        float lo[3], hi[3];
        dpoly_bounds(q, lo, hi);
        int a = 0;
        for (int k = 1; k < 3; ++k)
            if (hi[k] - lo[k] > hi[a] - lo[a]) a = k;
        const float ext = hi[a] - lo[a];
        bool split = L > 0.0f && ext > L && emitted + (uint32_t)sp < 4096u && q.n >= 3 && sp + 2 <= SPLIT_STACK;
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
            DevPoly l, r;
            dpoly_clip(q, a, s, -1, l);
            dpoly_clip(q, a, s, +1, r);
            if (l.n >= 3 && r.n >= 3)
            {
                stack[sp++] = l;
                stack[sp++] = r;
                continue;
            }
        }
        if (EMIT)
        {
            float* b = boxes + 6 * (size_t)(base + emitted);
            for (int k = 0; k < 3; ++k) { b[k] = lo[k] - pad; b[3 + k] = hi[k] + pad; }
            ref_tri[base + emitted] = i;
        }
        ++emitted;
    }
    if (!EMIT) counts[i] = emitted;
}

/* largest box extent per triangle (for the median that sets the fragment length) + scene bounds */
__global__ void k_tri_extents(const float* __restrict__ tris, int n, float* __restrict__ extents, unsigned int* __restrict__ bounds /* 6 ordered-uint */)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* t = tris + 15 * (size_t)i;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < 3; ++k)
        for (int a = 0; a < 3; ++a)
        {
            lo[a] = fminf(lo[a], t[3 * k + a]);
            hi[a] = fmaxf(hi[a], t[3 * k + a]);
        }
    extents[i] = fmaxf(hi[0] - lo[0], fmaxf(hi[1] - lo[1], hi[2] - lo[2]));
    /* order-preserving float -> uint: min/max by integer atomics */
    auto enc = [](float f) -> unsigned int { const unsigned int u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    for (int a = 0; a < 3; ++a)
    {
        atomicMin(&bounds[a], enc(lo[a]));
        atomicMax(&bounds[3 + a], enc(hi[a]));
    }
}

/* ------------------------------------------------------------------ 3. PLOC */
#ifndef RT_PLOC_RADIUS
#define RT_PLOC_RADIUS 16
#endif
struct PlocState
{
    unsigned int m;        /* clusters left */
    unsigned int merges;   /* internal nodes created so far */
};
RT_DEV float union_area(const float* a, const float* b)
{
    const float dx = fmaxf(a[3], b[3]) - fminf(a[0], b[0]);
    const float dy = fmaxf(a[4], b[4]) - fminf(a[1], b[1]);
    const float dz = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
    return dx * dy + dy * dz + dz * dx;
}
/* cluster i -> index of the neighbour within RADIUS places whose union box has the smallest area */
__global__ void k_ploc_nn(const PlocState* __restrict__ st, const float* __restrict__ cbox, int* __restrict__ nn, int radius)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = (int)st->m;
    if (i >= m) return;
    float mine[6];
    for (int k = 0; k < 6; ++k) mine[k] = cbox[6 * (size_t)i + k];
    const int j0 = i - radius < 0 ? 0 : i - radius, j1 = i + radius >= m ? m - 1 : i + radius;
    float best = INFINITY;
    int bj = -1;
    for (int j = j0; j <= j1; ++j)
    {
        if (j == i) continue;
        const float a = union_area(mine, cbox + 6 * (size_t)j);
        if (a < best) { best = a; bj = j; } /* ties: the leftmost */
    }
    nn[i] = bj;
}
/* mutual nearest neighbours merge into a new internal node (ids are handed out from n-2 downwards, so the
 * last merge — the root — is node 0); keep[i] = 1 if slot i survives (merged pairs survive in the left slot) */
__global__ void k_ploc_merge(PlocState* __restrict__ st, int n_leaves, int* __restrict__ cid, float* __restrict__ cbox,
                             const int* __restrict__ nn, int2* __restrict__ children, int* __restrict__ parent_inner,
                             float* __restrict__ node_boxes, unsigned int* __restrict__ keep)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = (int)st->m;
    if (i >= n_leaves) return;
    if (i >= m) { keep[i] = 0u; return; }
    const int j = nn[i];
    if (j < 0 || nn[j] != i) { keep[i] = 1u; return; }
    if (i > j) { keep[i] = 0u; return; }
    const int id = n_leaves - 2 - (int)atomicAdd(&st->merges, 1u);
    const int c0 = cid[i], c1 = cid[j];
    children[id] = make_int2(c0, c1);
    if (c0 >= 0) parent_inner[c0] = id;
    if (c1 >= 0) parent_inner[c1] = id;
    float* nb = node_boxes + 6 * (size_t)id;
    float* a = cbox + 6 * (size_t)i;
    const float* b = cbox + 6 * (size_t)j;
    for (int k = 0; k < 3; ++k)
    {
        const float lo = fminf(a[k], b[k]), hi = fmaxf(a[3 + k], b[3 + k]);
        nb[k] = lo; nb[3 + k] = hi;
    }
    /* the merged cluster replaces slot i. Safe in place: slot i is only read by i's own thread and by j's
     * thread, which returned above without reading boxes (nn was computed by the previous kernel). */
    for (int k = 0; k < 6; ++k) a[k] = nb[k];
    cid[i] = id;
    keep[i] = 1u;
}
__global__ void k_ploc_compact(PlocState* __restrict__ st, int n_leaves, const unsigned int* __restrict__ keep,
                               const unsigned int* __restrict__ pos, const int* __restrict__ cid, const float* __restrict__ cbox,
                               int* __restrict__ cid_out, float* __restrict__ cbox_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_leaves) return;
    if (keep[i])
    {
        const unsigned int p = pos[i];
        cid_out[p] = cid[i];
        for (int k = 0; k < 6; ++k) cbox_out[6 * (size_t)p + k] = cbox[6 * (size_t)i + k];
    }
    if (i == n_leaves - 1) st->m = pos[i] + keep[i];
}
__global__ void k_ploc_init(int n, const uint32_t* __restrict__ ids /* sorted position -> reference */, const float* __restrict__ prim_boxes,
                            int* __restrict__ cid, float* __restrict__ cbox, PlocState* __restrict__ st, int* __restrict__ parent_inner)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { st->m = (unsigned int)n; st->merges = 0u; parent_inner[0] = -1; }
    if (i >= n) return;
    cid[i] = ~i; /* leaf = sorted position, as the Karras kernels name them */
    const float* b = prim_boxes + 6 * (size_t)ids[i];
    for (int k = 0; k < 6; ++k) cbox[6 * (size_t)i + k] = b[k];
}
/* parent links of the cluster roots under the host-built top of the tree */
__global__ void k_scatter_int(int n, const int* __restrict__ idx, const int* __restrict__ val, int* __restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = val[i];
}
/* binary height: every leaf climbs to the root */
__global__ void k_bvh_height(int n_leaves, const int2* __restrict__ children, const int* __restrict__ parent_inner, int* __restrict__ height)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_leaves - 1) return;
    /* start from internal nodes that have a leaf child: depth of that leaf = climbs + 1 */
    const int2 ch = children[i];
    if (ch.x >= 0 && ch.y >= 0) return;
    int d = 1, node = i;
    while (node >= 0 && d < 4096) { node = parent_inner[node]; ++d; }
    atomicMax(height, d);
}

/* ------------------------------------------------------------------ 4. wide collapse, level by level */
struct CollapseItem
{
    int bin;          /* binary inner node to expand */
    unsigned int out; /* record slot that becomes this wide node */
};
struct CollapseState
{
    unsigned int n_rec;     /* records allocated */
    unsigned int count[2];  /* queue sizes (ping-pong) */
    unsigned int height;    /* wide levels emitted */
};
struct DevChild
{
    int bin;
    float lo[3], hi[3];
};
RT_DEV void dev_bin_children(const BvhNode& n, DevChild out[2])
{
    out[0].bin = n.d.x; out[1].bin = n.d.y;
    out[0].lo[0] = n.a.x; out[0].lo[1] = n.a.y; out[0].lo[2] = n.a.z;
    out[0].hi[0] = n.b.x; out[0].hi[1] = n.b.y; out[0].hi[2] = n.b.z;
    out[1].lo[0] = n.a.w; out[1].lo[1] = n.b.w; out[1].lo[2] = n.c.w;
    out[1].hi[0] = n.c.x; out[1].hi[1] = n.c.y; out[1].hi[2] = n.c.z;
}
RT_DEV float dev_area6(const float* lo, const float* hi)
{
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}
__global__ void k_collapse_init(CollapseState* st, CollapseItem* q0)
{
    st->n_rec = 1u; st->count[0] = 1u; st->count[1] = 0u; st->height = 0u;
    q0[0].bin = 0; q0[0].out = 0u;
}
/* one wide level: queue `in` (parity = level & 1) -> records + queue `out`. Same node format, child order
 * (ascending box area) and conservative quantisation as bvh_build_host.h::collapse_wide. */
__global__ void k_collapse_level(const BvhNode* __restrict__ bin, const float* __restrict__ tris /* 15 floats */, int level,
                                 CollapseState* __restrict__ st, const CollapseItem* __restrict__ in, CollapseItem* __restrict__ out,
                                 uint32_t* __restrict__ recs /* 12 words per record */)
{
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int n_in = st->count[level & 1];
    if (t == 0 && n_in > 0) st->height = (unsigned int)level + 1u;
    if (t >= n_in) return;
    const CollapseItem wk = in[t];
    DevChild ch[4];
    int n = 2;
    dev_bin_children(bin[wk.bin], ch);
    while (n < 4)
    {
        int pick = -1;
        float best = -1.0f;
        for (int k = 0; k < n; ++k)
            if (ch[k].bin >= 0)
            {
                const float a = dev_area6(ch[k].lo, ch[k].hi);
                if (a > best) { best = a; pick = k; }
            }
        if (pick < 0) break;
        DevChild two[2];
        dev_bin_children(bin[ch[pick].bin], two);
        ch[pick] = two[0];
        ch[n++] = two[1];
    }
    /* ascending area (insertion sort of <= 4; stable like the host's order for distinct areas) */
    for (int i = 1; i < n; ++i)
    {
        const DevChild c = ch[i];
        const float ac = dev_area6(c.lo, c.hi);
        int j = i - 1;
        while (j >= 0 && dev_area6(ch[j].lo, ch[j].hi) > ac) { ch[j + 1] = ch[j]; --j; }
        ch[j + 1] = c;
    }
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < n; ++k)
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], ch[k].lo[a]); hi[a] = fmaxf(hi[a], ch[k].hi[a]); }
    const unsigned int base = atomicAdd(&st->n_rec, (unsigned int)n);
    uint32_t ebits[3];
    float scale[3];
    for (int a = 0; a < 3; ++a)
    {
        const float ext = fmaxf(hi[a] - lo[a], 1e-30f);
        int e;
        frexpf(ext / 255.0f, &e);
        int biased = e + 127;
        if (biased < 1) biased = 1;
        if (biased > 254) biased = 254;
        ebits[a] = (uint32_t)biased;
        scale[a] = ldexpf(1.0f, biased - 127);
    }
    uint32_t q[6] = {0, 0, 0, 0, 0, 0}, meta = 0;
    int n_inner = 0;
    for (int k = 0; k < n; ++k)
        if (ch[k].bin >= 0) ++n_inner;
    const unsigned int qbase = n_inner ? atomicAdd(&st->count[(level + 1) & 1], (unsigned int)n_inner) : 0u;
    int qi = 0;
    for (int k = 0; k < n; ++k)
    {
        for (int a = 0; a < 3; ++a)
        {
            int ql = (int)floorf((ch[k].lo[a] - lo[a]) / scale[a]);
            int qh = (int)ceilf((ch[k].hi[a] - lo[a]) / scale[a]);
            /* the traversal decodes lo + q*scale in binary32: the decoded box must contain the child box */
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
            out[qbase + (unsigned int)qi].bin = ch[k].bin;
            out[qbase + (unsigned int)qi].out = base + (unsigned int)k;
            ++qi;
        }
        else
        {
            meta |= 2u << (8 * k);
            const int ti = ~ch[k].bin;
            const float* tv = tris + 15 * (size_t)ti;
            uint32_t* Lr = recs + 12 * (size_t)(base + (unsigned int)k);
            for (int i = 0; i < 9; ++i) Lr[i] = __float_as_uint(tv[i]);
            Lr[9] = (uint32_t)ti;
            Lr[10] = Lr[11] = 0u;
        }
    }
    uint32_t* R = recs + 12 * (size_t)wk.out;
    R[0] = __float_as_uint(lo[0]); R[1] = __float_as_uint(lo[1]); R[2] = __float_as_uint(lo[2]);
    R[3] = ebits[0] | (ebits[1] << 8) | (ebits[2] << 16);
    R[4] = base; R[5] = meta; R[6] = q[0]; R[7] = q[1];
    R[8] = q[2]; R[9] = q[3]; R[10] = q[4]; R[11] = q[5];
}
/* the consumed queue is emptied for its next use two levels later */
__global__ void k_collapse_swap(int level, CollapseState* st) { st->count[level & 1] = 0u; }

}  // namespace rt
