/*
 * bvh_build_device.h — the BVH build entirely on the GPU (included once, by restir_rt.hip; replaces the reference's
 * hiprtBuildGeometry, common/loader.hpp:68-112). Builder 3 (rt_tuning key 5, the default and the product's only one):
 *
 *   1.  early split clipping of large triangles into box fragments ("references"): count pass, exclusive scan, emit
 *       pass (the recursion of bvh_build_host.h::split_refs with a per-thread stack);
 *   3b. top-down binned SAH over the references: nodes of more than SAH_MEDIUM references are binned by many
 *       workgroups and split by one wavefront each, medium nodes are binned and split by one workgroup, small ones by one
 *       wavefront down to the leaves — the tree the host builder (bvh_build_host.h, builder 1) makes, reproduced on the device;
 *   4.  collapse into the 4-wide quantised 48-B records the kernels walk (bvh.h), level by level on the device.
 * No host round trip carries tree data; the host reads back a handful of counters (reference count, record count,
 * heights). Wall time of rt_scene_set for the 212 k-triangle bench scene: rt_build_ms / `build_ms` of the bench line
 * (upload + light tables + this build, synchronised).
 *
 * A/B only (-DRT_EXPERIMENTS): section 3, PLOC — parallel locally-ordered clustering (Meister & Bittner 2018) over Morton-sorted
 * references (builder 2, round 2) — and, through bvh.h's kernels, the Karras hierarchy + refit of round 1 (builder 0).
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

#ifndef RT_PLOC_RADIUS
#define RT_PLOC_RADIUS 16
#endif
#ifdef RT_EXPERIMENTS /* builder 2 (PLOC + host SAH over the top, r02): A/B only since the device SAH builder (3) */
/* ------------------------------------------------------------------ 3. PLOC */
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
#endif /* RT_EXPERIMENTS */
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

/* ------------------------------------------------------------------ 3b. top-down binned SAH on the device (builder 3, r03)
 * The algorithm of bvh_build_host.h::SahBuilder (32 centroid bins per axis, cost = area x count of the two sides, one
 * reference per leaf, median split when no plane separates anything) run on the GPU, so that the tree quality of the
 * default builder no longer costs 230 ms of host time (the reference asks HIPRT for its high-quality build,
 * common/loader.hpp:98-99, and HIPRT builds on the GPU). On the benchmark scene it makes the host builder's tree
 * (380 779 wide records, 15 levels) in a few milliseconds.
 *   phase 1, level by level over the LARGE nodes (more than SAH_SMALL references; each a contiguous range of `order`):
 *     k_sah_bin        every reference adds itself to the bins of its node: per workgroup in LDS for the node most of the
 *                      workgroup belongs to, flushed with one atomic per non-empty word (same-address device atomics cost
 *                      ~140 ns each here: 252 k references binning straight into 96 bins took 43 of a 56 ms build)
 *     k_sah_split      one wavefront per node: prefix / suffix boxes over the bins, best plane, children created with
 *                      their boxes and centroid bounds (exact: both are unions over bins)
 *     k_sah_classify + exclusive scan + k_sah_scatter   a stable partition of every node's range: the rank of a
 *                      reference among the left ones of its node is a difference of two entries of ONE global prefix sum
 *   phase 2: every subtree of <= SAH_SMALL references is finished by ONE wavefront in LDS (k_sah_small).
 * Output = the arrays the PLOC path hands to k_bvh_height / k_bvh_emit / the wide collapse: children, parent, boxes of
 * the n - 1 inner nodes (root = 0), leaves named ~position with order[position] = reference. The tree is deterministic
 * (stable partitions; only the numbering of the inner nodes depends on the order of atomics). */
constexpr int SAH_BINS = 32;
constexpr int SAH_SMALL = 64;
constexpr int SAH_BIN_WORDS = 13; /* count, box lo[3], box hi[3], centroid lo[3], centroid hi[3] (ordered-uint keys) */
constexpr int SAH_NODE_BIN_WORDS = 3 * SAH_BINS * SAH_BIN_WORDS;
constexpr int SAH_BLOCK = 1024;
constexpr int SAH_MEDIUM = 4096; /* nodes up to this size are binned and split by one workgroup (k_sah_medium) */
RT_DEV unsigned int sah_enc(float f) { const unsigned int u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
RT_DEV float sah_dec(unsigned int k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
constexpr unsigned int SAH_KEY_PINF = 0xff800000u; /* sah_enc(+inf) */
constexpr unsigned int SAH_KEY_NINF = 0x007fffffu; /* sah_enc(-inf) */
/* word w of a bin starts as: count 0, minima +inf, maxima -inf */
RT_DEV unsigned int sah_bin_empty(int w) { return w == 0 ? 0u : ((w <= 3 || (w >= 7 && w <= 9)) ? SAH_KEY_PINF : SAH_KEY_NINF); }
RT_DEV bool sah_bin_is_min(int w) { return w <= 3 || (w >= 7 && w <= 9); }
struct SahNode /* a large node of the current level */
{
    int first, count, id, pad;
    float box[6];  /* lo[3], hi[3] */
    float cent[6]; /* centroid bounds */
};
struct SahSplit /* decided by k_sah_split, applied by k_sah_classify / k_sah_scatter */
{
    int first, count, lc, axis; /* axis < 0: no separating plane, the first lc positions go left */
    int split;
    float clo, sc;
    int child_slot[2]; /* slot among the next level's large nodes, or -1 (leaf / small subtree) */
};
struct SahSmallRoot { int first, count, id, pad; };
struct SahState
{
    unsigned int n_active[2]; /* large nodes of the level (parity) */
    unsigned int next_node;   /* inner node ids handed out */
    unsigned int n_small;
    unsigned int overflow;    /* != 0: a table was too small (never with the sizes the host allocates) */
    unsigned int max_level;
    unsigned int root_key[12]; /* root bounds: keys of box lo/hi, centroid lo/hi */
};
RT_DEV float sah_wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
RT_DEV float sah_wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__global__ void k_sah_begin(SahState* __restrict__ st)
{
    st->n_active[0] = 0u; st->n_active[1] = 0u; st->next_node = 1u; st->n_small = 0u; st->overflow = 0u; st->max_level = 0u;
    for (int k = 0; k < 3; ++k) { st->root_key[k] = SAH_KEY_PINF; st->root_key[3 + k] = SAH_KEY_NINF; st->root_key[6 + k] = SAH_KEY_PINF; st->root_key[9 + k] = SAH_KEY_NINF; }
}
/* root bounds: workgroup-level reduction, 12 atomics per workgroup; order = identity, everything in slot 0 */
__global__ __launch_bounds__(SAH_BLOCK) void k_sah_root_bounds(int n, const float* __restrict__ boxes, uint32_t* __restrict__ order, int* __restrict__ seg,
                                                               SahState* __restrict__ st)
{
    __shared__ float s_red[SAH_BLOCK / 64][12];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v[12];
    for (int k = 0; k < 3; ++k) { v[k] = INFINITY; v[3 + k] = -INFINITY; v[6 + k] = INFINITY; v[9 + k] = -INFINITY; }
    if (i < n)
    {
        const float* b = boxes + 6 * (size_t)i;
        for (int k = 0; k < 3; ++k) { v[k] = b[k]; v[3 + k] = b[3 + k]; v[6 + k] = v[9 + k] = 0.5f * (b[k] + b[3 + k]); }
        order[i] = (uint32_t)i;
        seg[i] = n > SAH_SMALL ? 0 : -1;
    }
    for (int k = 0; k < 12; ++k) v[k] = ((k % 6) < 3) ? sah_wave_min(v[k]) : sah_wave_max(v[k]);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) for (int k = 0; k < 12; ++k) s_red[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 12)
    {
        const int k = threadIdx.x;
        float r = s_red[0][k];
        for (int w = 1; w < SAH_BLOCK / 64; ++w) r = ((k % 6) < 3) ? fminf(r, s_red[w][k]) : fmaxf(r, s_red[w][k]);
        if ((k % 6) < 3) atomicMin(&st->root_key[k], sah_enc(r)); else atomicMax(&st->root_key[k], sah_enc(r));
    }
}
/* the root becomes the first large node, or (<= SAH_SMALL references) a small subtree from the start */
__global__ void k_sah_root(int n, SahState* __restrict__ st, SahNode* __restrict__ act0, SahSmallRoot* __restrict__ small)
{
    if (n <= SAH_SMALL) { st->n_small = 1u; small[0].first = 0; small[0].count = n; small[0].id = 0; small[0].pad = 0; return; }
    st->n_active[0] = 1u;
    act0[0].first = 0; act0[0].count = n; act0[0].id = 0; act0[0].pad = 0;
    for (int k = 0; k < 6; ++k) { act0[0].box[k] = sah_dec(st->root_key[k]); act0[0].cent[k] = sah_dec(st->root_key[6 + k]); }
}
/* bins of the level's nodes start empty */
__global__ void k_sah_clear_bins(int level, const SahState* __restrict__ st, unsigned int* __restrict__ bins, size_t max_words)
{
    const size_t words = (size_t)st->n_active[level & 1] * SAH_NODE_BIN_WORDS;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words && i < max_words; i += (size_t)gridDim.x * blockDim.x)
        bins[i] = sah_bin_empty((int)(i % SAH_BIN_WORDS));
}
__global__ __launch_bounds__(SAH_BLOCK) void k_sah_bin(int n, int level, const SahState* __restrict__ st, const SahNode* __restrict__ act,
                                                       const float* __restrict__ boxes, const uint32_t* __restrict__ order, const int* __restrict__ seg,
                                                       unsigned int* __restrict__ bins)
{
    if (st->n_active[level & 1] == 0u) return;
    __shared__ unsigned int s_bins[SAH_NODE_BIN_WORDS];
    __shared__ int s_home;
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    int s = pos < n ? seg[pos] : -1;
    if (s >= 0 && act[s].count <= SAH_MEDIUM) s = -1; /* k_sah_medium bins those */
    /* home node of the workgroup: the node of its middle position (most of a workgroup shares one node while nodes are
     * larger than workgroups, which is when contention matters) */
    if (threadIdx.x == SAH_BLOCK / 2) s_home = s;
    for (int i = threadIdx.x; i < SAH_NODE_BIN_WORDS; i += SAH_BLOCK) s_bins[i] = sah_bin_empty(i % SAH_BIN_WORDS);
    __syncthreads();
    const int home = s_home;
    if (s >= 0)
    {
        const SahNode& nd = act[s];
        const float* b = boxes + 6 * (size_t)order[pos];
        float bb[6], c[3];
        for (int k = 0; k < 6; ++k) bb[k] = b[k];
        for (int a = 0; a < 3; ++a) c[a] = 0.5f * (bb[a] + bb[3 + a]);
        unsigned int* nb = (s == home) ? s_bins : bins + (size_t)s * SAH_NODE_BIN_WORDS;
        for (int a = 0; a < 3; ++a)
        {
            const float clo = nd.cent[a], ext = nd.cent[3 + a] - clo;
            if (!(ext > 0.0f)) continue;
            const float sc = (float)SAH_BINS / ext;
            int bi = (int)((c[a] - clo) * sc);
            bi = bi < 0 ? 0 : (bi >= SAH_BINS ? SAH_BINS - 1 : bi);
            unsigned int* w = nb + (a * SAH_BINS + bi) * SAH_BIN_WORDS;
            atomicAdd(&w[0], 1u);
            for (int k = 0; k < 3; ++k)
            {
                atomicMin(&w[1 + k], sah_enc(bb[k])); atomicMax(&w[4 + k], sah_enc(bb[3 + k]));
                atomicMin(&w[7 + k], sah_enc(c[k])); atomicMax(&w[10 + k], sah_enc(c[k]));
            }
        }
    }
    __syncthreads();
    if (home < 0) return;
    unsigned int* hb = bins + (size_t)home * SAH_NODE_BIN_WORDS;
    for (int i = threadIdx.x; i < SAH_NODE_BIN_WORDS; i += SAH_BLOCK)
    {
        const int w = i % SAH_BIN_WORDS;
        const unsigned int v = s_bins[i];
        if (v == sah_bin_empty(w)) continue;
        if (w == 0) atomicAdd(&hb[i], v);
        else if (sah_bin_is_min(w)) atomicMin(&hb[i], v);
        else atomicMax(&hb[i], v);
    }
}
RT_DEV float sah_area(const float* lo, const float* hi)
{
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}
/* child of `parent` covering `count` references from position `first`: a leaf, a small subtree or a large node of the
 * next level (with its bounds). Returns the child code for children[parent]; *slot = its slot among the next level's nodes */
RT_DEV int sah_make_child(SahState* __restrict__ st, int level, int parent, int first, int count, const float* box, const float* cent,
                          int max_active, int max_small, SahNode* __restrict__ act_next, SahSmallRoot* __restrict__ small,
                          int* __restrict__ parent_inner, int* slot)
{
    *slot = -1;
    if (count == 1) return ~first;
    const int id = (int)atomicAdd(&st->next_node, 1u);
    parent_inner[id] = parent;
    if (count <= SAH_SMALL)
    {
        const unsigned int k = atomicAdd(&st->n_small, 1u);
        if ((int)k < max_small) { small[k].first = first; small[k].count = count; small[k].id = id; small[k].pad = 0; }
        else st->overflow = 1u;
        return id;
    }
    const unsigned int k = atomicAdd(&st->n_active[(level + 1) & 1], 1u);
    if ((int)k < max_active)
    {
        SahNode& nd = act_next[k];
        nd.first = first; nd.count = count; nd.id = id; nd.pad = 0;
        for (int i = 0; i < 6; ++i) { nd.box[i] = box[i]; nd.cent[i] = cent[i]; }
        *slot = (int)k;
    }
    else st->overflow = 1u;
    return id;
}
/* One wavefront decides the split of node `nd` from its bins `sb` (LDS) and creates the children. s_child: 24 floats of LDS. */
RT_DEV void sah_split_wave(int level, int s, const SahNode& nd, const unsigned int* sb, float (*s_child)[12], int lane, SahState* __restrict__ st,
                           SahNode* __restrict__ act_next, SahSplit* __restrict__ splits, SahSmallRoot* __restrict__ small, int2* __restrict__ children,
                           int* __restrict__ parent_inner, float* __restrict__ node_boxes, int max_active, int max_small)
{
    /* lane b < 31, each axis in turn: the plane behind bin b. Winner = smallest cost, then the smallest (axis, b): the
     * first minimum of the host's loops */
    float best = INFINITY;
    int best_key = 0x7fffffff;
    for (int a = 0; a < 3; ++a)
    {
        const float ext = nd.cent[3 + a] - nd.cent[a];
        if (!(ext > 0.0f) || lane >= SAH_BINS - 1) continue;
        float ll[3] = {INFINITY, INFINITY, INFINITY}, lh[3] = {-INFINITY, -INFINITY, -INFINITY};
        float rl[3] = {INFINITY, INFINITY, INFINITY}, rh[3] = {-INFINITY, -INFINITY, -INFINITY};
        int lc = 0, rc = 0;
        for (int b = 0; b < SAH_BINS; ++b)
        {
            const unsigned int* w = sb + (a * SAH_BINS + b) * SAH_BIN_WORDS;
            const int c = (int)w[0];
            if (c == 0) continue;
            if (b <= lane) { lc += c; for (int k = 0; k < 3; ++k) { ll[k] = fminf(ll[k], sah_dec(w[1 + k])); lh[k] = fmaxf(lh[k], sah_dec(w[4 + k])); } }
            else { rc += c; for (int k = 0; k < 3; ++k) { rl[k] = fminf(rl[k], sah_dec(w[1 + k])); rh[k] = fmaxf(rh[k], sah_dec(w[4 + k])); } }
        }
        if (lc == 0 || rc == 0) continue;
        const float cost = sah_area(ll, lh) * (float)lc + sah_area(rl, rh) * (float)rc;
        const int key = a * SAH_BINS + lane;
        if (cost < best) { best = cost; best_key = key; }
    }
    for (int o = 32; o > 0; o >>= 1)
    {
        const float oc = __shfl_xor(best, o);
        const int ok = __shfl_xor(best_key, o);
        if (oc < best || (oc == best && ok < best_key)) { best = oc; best_key = ok; }
    }
    const bool plane = best_key != 0x7fffffff && best < INFINITY;
    int lcount = nd.count / 2;
    if (plane)
    {
        /* lanes 0..11 (left) and 16..27 (right): one bound of one side each, over the bins of that side */
        const int a = best_key / SAH_BINS, sp = best_key % SAH_BINS;
        const int side = lane >> 4, k = lane & 15;
        if (side < 2 && k < 12)
        {
            const bool is_min = (k % 6) < 3;
            float r = is_min ? INFINITY : -INFINITY;
            for (int b = side ? sp + 1 : 0; b <= (side ? SAH_BINS - 1 : sp); ++b)
            {
                const unsigned int* w = sb + (a * SAH_BINS + b) * SAH_BIN_WORDS;
                if (w[0] == 0u) continue;
                const float v = sah_dec(w[1 + k]);
                r = is_min ? fminf(r, v) : fmaxf(r, v);
            }
            s_child[side][k] = r;
        }
        int lc = 0;
        if (lane == 0) for (int b = 0; b <= sp; ++b) lc += (int)sb[(a * SAH_BINS + b) * SAH_BIN_WORDS];
        lcount = __shfl(lc, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane != 0) return;
    SahSplit spl;
    spl.first = nd.first; spl.count = nd.count; spl.lc = lcount;
    spl.clo = 0.0f; spl.sc = 0.0f; spl.split = 0; spl.axis = -1;
    float cb[2][6], cc[2][6];
    if (plane)
    {
        spl.axis = best_key / SAH_BINS; spl.split = best_key % SAH_BINS;
        spl.clo = nd.cent[spl.axis];
        spl.sc = (float)SAH_BINS / (nd.cent[3 + spl.axis] - nd.cent[spl.axis]);
        for (int sd = 0; sd < 2; ++sd)
            for (int k = 0; k < 6; ++k) { cb[sd][k] = s_child[sd][k]; cc[sd][k] = s_child[sd][6 + k]; }
    }
    else
    {
        /* every centroid of the node is the same point: halves by position; their boxes are not known separately, the
         * node's box bounds both (only duplicates and concentric references come here) */
        for (int sd = 0; sd < 2; ++sd)
            for (int k = 0; k < 6; ++k) { cb[sd][k] = nd.box[k]; cc[sd][k] = nd.cent[k]; }
    }
    float* nb = node_boxes + 6 * (size_t)nd.id;
    for (int k = 0; k < 6; ++k) nb[k] = nd.box[k];
    int slot0, slot1;
    const int c0 = sah_make_child(st, level, nd.id, nd.first, spl.lc, cb[0], cc[0], max_active, max_small, act_next, small, parent_inner, &slot0);
    const int c1 = sah_make_child(st, level, nd.id, nd.first + spl.lc, nd.count - spl.lc, cb[1], cc[1], max_active, max_small, act_next, small, parent_inner, &slot1);
    children[nd.id] = make_int2(c0, c1);
    spl.child_slot[0] = slot0; spl.child_slot[1] = slot1;
    splits[s] = spl;
    if ((unsigned int)level + 1u > st->max_level) st->max_level = (unsigned int)level + 1u;
}
/* nodes of more than SAH_MEDIUM references (binned by k_sah_bin over many workgroups): one wavefront per node */
__global__ __launch_bounds__(64) void k_sah_split(int level, SahState* __restrict__ st, const SahNode* __restrict__ act, SahNode* __restrict__ act_next,
                                                  const unsigned int* __restrict__ bins, SahSplit* __restrict__ splits, SahSmallRoot* __restrict__ small,
                                                  int2* __restrict__ children, int* __restrict__ parent_inner, float* __restrict__ node_boxes,
                                                  int max_active, int max_small)
{
    const int s = blockIdx.x;
    if ((unsigned int)s >= st->n_active[level & 1]) return;
    const SahNode nd = act[s];
    if (nd.count <= SAH_MEDIUM) return; /* k_sah_medium's */
    __shared__ unsigned int sb[SAH_NODE_BIN_WORDS];
    __shared__ float s_child[2][12]; /* box lo/hi, centroid lo/hi of the two sides of the winning plane */
    const int lane = threadIdx.x;
    for (int i = lane; i < SAH_NODE_BIN_WORDS; i += 64) sb[i] = bins[(size_t)s * SAH_NODE_BIN_WORDS + i];
    __syncthreads();
    sah_split_wave(level, s, nd, sb, s_child, lane, st, act_next, splits, small, children, parent_inner, node_boxes, max_active, max_small);
}
/* nodes of SAH_SMALL < count <= SAH_MEDIUM references: ONE workgroup bins the node in LDS and its first wavefront splits it
 * (no global bins, no contended atomics: these are the levels where every workgroup of k_sah_bin would span several nodes) */
__global__ __launch_bounds__(SAH_BLOCK) void k_sah_medium(int level, SahState* __restrict__ st, const SahNode* __restrict__ act, SahNode* __restrict__ act_next,
                                                          const float* __restrict__ boxes, const uint32_t* __restrict__ order, SahSplit* __restrict__ splits,
                                                          SahSmallRoot* __restrict__ small, int2* __restrict__ children, int* __restrict__ parent_inner,
                                                          float* __restrict__ node_boxes, int max_active, int max_small)
{
    const int s = blockIdx.x;
    if ((unsigned int)s >= st->n_active[level & 1]) return;
    const SahNode nd = act[s];
    if (nd.count > SAH_MEDIUM) return;
    __shared__ unsigned int sb[SAH_NODE_BIN_WORDS];
    __shared__ float s_child[2][12];
    for (int i = threadIdx.x; i < SAH_NODE_BIN_WORDS; i += SAH_BLOCK) sb[i] = sah_bin_empty(i % SAH_BIN_WORDS);
    __syncthreads();
    for (int i = threadIdx.x; i < nd.count; i += SAH_BLOCK)
    {
        const float* b = boxes + 6 * (size_t)order[nd.first + i];
        float bb[6], c[3];
        for (int k = 0; k < 6; ++k) bb[k] = b[k];
        for (int a = 0; a < 3; ++a) c[a] = 0.5f * (bb[a] + bb[3 + a]);
        for (int a = 0; a < 3; ++a)
        {
            const float clo = nd.cent[a], ext = nd.cent[3 + a] - clo;
            if (!(ext > 0.0f)) continue;
            const float sc = (float)SAH_BINS / ext;
            int bi = (int)((c[a] - clo) * sc);
            bi = bi < 0 ? 0 : (bi >= SAH_BINS ? SAH_BINS - 1 : bi);
            unsigned int* w = sb + (a * SAH_BINS + bi) * SAH_BIN_WORDS;
            atomicAdd(&w[0], 1u);
            for (int k = 0; k < 3; ++k)
            {
                atomicMin(&w[1 + k], sah_enc(bb[k])); atomicMax(&w[4 + k], sah_enc(bb[3 + k]));
                atomicMin(&w[7 + k], sah_enc(c[k])); atomicMax(&w[10 + k], sah_enc(c[k]));
            }
        }
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    sah_split_wave(level, s, nd, sb, s_child, (int)threadIdx.x, st, act_next, splits, small, children, parent_inner, node_boxes, max_active, max_small);
}
/* side of every reference (1 = left); positions outside the level's nodes count as 0 */
__global__ void k_sah_classify(int n, int level, const SahState* __restrict__ st, const SahSplit* __restrict__ splits, const float* __restrict__ boxes,
                               const uint32_t* __restrict__ order, const int* __restrict__ seg, unsigned int* __restrict__ flag)
{
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    const int s = st->n_active[level & 1] != 0u ? seg[pos] : -1;
    unsigned int f = 0u;
    if (s >= 0)
    {
        const SahSplit& sp = splits[s];
        if (sp.axis < 0) f = (pos - sp.first < sp.lc) ? 1u : 0u;
        else
        {
            const float* b = boxes + 6 * (size_t)order[pos];
            int bi = (int)((0.5f * (b[sp.axis] + b[3 + sp.axis]) - sp.clo) * sp.sc);
            bi = bi < 0 ? 0 : (bi >= SAH_BINS ? SAH_BINS - 1 : bi);
            f = bi <= sp.split ? 1u : 0u;
        }
    }
    flag[pos] = f;
}
/* stable partition of every node's range: left rank = pre[pos] - pre[first] (pre = exclusive prefix sum of the flags) */
__global__ void k_sah_scatter(int n, int level, const SahState* __restrict__ st, const SahSplit* __restrict__ splits, const unsigned int* __restrict__ flag,
                              const unsigned int* __restrict__ pre, const uint32_t* __restrict__ order, const int* __restrict__ seg,
                              uint32_t* __restrict__ order_out, int* __restrict__ seg_out)
{
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    const int s = st->n_active[level & 1] != 0u ? seg[pos] : -1;
    if (s < 0) { order_out[pos] = order[pos]; seg_out[pos] = -1; return; }
    const SahSplit& sp = splits[s];
    const int lrank = (int)(pre[pos] - pre[sp.first]);
    const bool left = flag[pos] != 0u;
    const int dst = left ? sp.first + lrank : sp.first + sp.lc + (pos - sp.first - lrank);
    order_out[dst] = order[pos];
    seg_out[dst] = sp.child_slot[left ? 0 : 1];
}
__global__ void k_sah_next_level(int level, SahState* __restrict__ st) { st->n_active[level & 1] = 0u; }

/* phase 2: one wavefront finishes a subtree of <= SAH_SMALL references in LDS */
__global__ __launch_bounds__(64) void k_sah_small(SahState* __restrict__ st, const SahSmallRoot* __restrict__ small, const float* __restrict__ boxes,
                                                  uint32_t* __restrict__ order, int2* __restrict__ children, int* __restrict__ parent_inner,
                                                  float* __restrict__ node_boxes)
{
    const unsigned int w = blockIdx.x;
    if (w >= st->n_small) return;
    constexpr int SM_WORDS = 7; /* count, box lo[3], box hi[3]: the subtree's centroid bounds come from a wave reduction */
    __shared__ float s_box[SAH_SMALL][6];
    __shared__ uint32_t s_ref[SAH_SMALL];
    __shared__ int s_ord[2][SAH_SMALL];       /* slot -> local index, ping-pong */
    __shared__ unsigned int s_bins[3 * SAH_BINS * SM_WORDS];
    __shared__ int s_stack[SAH_SMALL][4];     /* first, count, id, which buffer holds its slots */
    __shared__ int s_sp;
    const int lane = threadIdx.x;
    const SahSmallRoot root = small[w];
    if (lane < root.count)
    {
        const uint32_t r = order[root.first + lane];
        s_ref[lane] = r;
        for (int k = 0; k < 6; ++k) s_box[lane][k] = boxes[6 * (size_t)r + k];
        s_ord[0][lane] = lane; s_ord[1][lane] = lane;
    }
    int id_base = 0;
    if (lane == 0)
    {
        s_stack[0][0] = 0; s_stack[0][1] = root.count; s_stack[0][2] = root.id; s_stack[0][3] = 0;
        s_sp = 1;
        if (root.count > 2) id_base = (int)atomicAdd(&st->next_node, (unsigned int)(root.count - 2));
    }
    id_base = __shfl(id_base, 0);
    int ids_used = 0;
    __syncthreads();
    while (s_sp > 0)
    {
        const int top = s_sp - 1;
        const int first = s_stack[top][0], count = s_stack[top][1], id = s_stack[top][2], buf = s_stack[top][3];
        __syncthreads();
        if (lane == 0) s_sp = top;
        if (count == 2)
        {
            /* two references: the host's first minimum is the first axis on which the centroids differ, plane behind bin 0
             * (every plane costs the same): the smaller centroid goes left; identical centroids keep their order */
            if (lane == 0)
            {
                const int i0 = s_ord[buf][first], i1 = s_ord[buf][first + 1];
                bool swap = false;
                for (int a = 0; a < 3; ++a)
                {
                    const float c0 = 0.5f * (s_box[i0][a] + s_box[i0][3 + a]), c1 = 0.5f * (s_box[i1][a] + s_box[i1][3 + a]);
                    if (c0 != c1) { swap = c1 < c0; break; }
                }
                const int l0 = swap ? i1 : i0, l1 = swap ? i0 : i1;
                s_ord[0][first] = l0; s_ord[1][first] = l0; s_ord[0][first + 1] = l1; s_ord[1][first + 1] = l1;
                float* nb = node_boxes + 6 * (size_t)id;
                for (int a = 0; a < 3; ++a) { nb[a] = fminf(s_box[i0][a], s_box[i1][a]); nb[3 + a] = fmaxf(s_box[i0][3 + a], s_box[i1][3 + a]); }
                children[id] = make_int2(~(root.first + first), ~(root.first + first + 1));
            }
            __syncthreads();
            continue;
        }
        const bool act = lane < count;
        const int li = act ? s_ord[buf][first + lane] : 0;
        float b[6], c[3];
        for (int k = 0; k < 6; ++k) b[k] = s_box[li][k];
        for (int a = 0; a < 3; ++a) c[a] = 0.5f * (b[a] + b[3 + a]);
        float lo[3], hi[3], cl[3], ch[3];
        for (int a = 0; a < 3; ++a)
        {
            lo[a] = sah_wave_min(act ? b[a] : INFINITY); hi[a] = sah_wave_max(act ? b[3 + a] : -INFINITY);
            cl[a] = sah_wave_min(act ? c[a] : INFINITY); ch[a] = sah_wave_max(act ? c[a] : -INFINITY);
        }
        if (lane == 0) { float* nb = node_boxes + 6 * (size_t)id; for (int a = 0; a < 3; ++a) { nb[a] = lo[a]; nb[3 + a] = hi[a]; } }
        for (int i = lane; i < (3 * SAH_BINS * SM_WORDS); i += 64)
        {
            const int wd = i % SM_WORDS;
            s_bins[i] = wd == 0 ? 0u : (wd <= 3 ? SAH_KEY_PINF : SAH_KEY_NINF);
        }
        __syncthreads();
        int mybin[3] = {0, 0, 0};
        for (int a = 0; a < 3; ++a)
        {
            const float ext = ch[a] - cl[a];
            if (!(ext > 0.0f) || !act) continue;
            const float sc = (float)SAH_BINS / ext;
            int bi = (int)((c[a] - cl[a]) * sc);
            bi = bi < 0 ? 0 : (bi >= SAH_BINS ? SAH_BINS - 1 : bi);
            mybin[a] = bi;
            unsigned int* wd = s_bins + (a * SAH_BINS + bi) * SM_WORDS;
            atomicAdd(&wd[0], 1u);
            for (int k = 0; k < 3; ++k) { atomicMin(&wd[1 + k], sah_enc(b[k])); atomicMax(&wd[4 + k], sah_enc(b[3 + k])); }
        }
        __syncthreads();
        float best = INFINITY;
        int best_key = 0x7fffffff;
        for (int a = 0; a < 3; ++a)
        {
            const float ext = ch[a] - cl[a];
            if (!(ext > 0.0f) || lane >= SAH_BINS - 1) continue;
            float ll[3] = {INFINITY, INFINITY, INFINITY}, lh[3] = {-INFINITY, -INFINITY, -INFINITY};
            float rl[3] = {INFINITY, INFINITY, INFINITY}, rh[3] = {-INFINITY, -INFINITY, -INFINITY};
            int lc = 0, rc = 0;
            for (int bb = 0; bb < SAH_BINS; ++bb)
            {
                const unsigned int* wd = s_bins + (a * SAH_BINS + bb) * SM_WORDS;
                const int cc = (int)wd[0];
                if (cc == 0) continue;
                if (bb <= lane) { lc += cc; for (int k = 0; k < 3; ++k) { ll[k] = fminf(ll[k], sah_dec(wd[1 + k])); lh[k] = fmaxf(lh[k], sah_dec(wd[4 + k])); } }
                else { rc += cc; for (int k = 0; k < 3; ++k) { rl[k] = fminf(rl[k], sah_dec(wd[1 + k])); rh[k] = fmaxf(rh[k], sah_dec(wd[4 + k])); } }
            }
            if (lc == 0 || rc == 0) continue;
            const float cost = sah_area(ll, lh) * (float)lc + sah_area(rl, rh) * (float)rc;
            const int key = a * SAH_BINS + lane;
            if (cost < best) { best = cost; best_key = key; }
        }
        for (int o = 32; o > 0; o >>= 1)
        {
            const float oc = __shfl_xor(best, o);
            const int ok = __shfl_xor(best_key, o);
            if (oc < best || (oc == best && ok < best_key)) { best = oc; best_key = ok; }
        }
        bool left;
        if (best_key != 0x7fffffff && best < INFINITY)
        {
            const int a = best_key / SAH_BINS, sp = best_key % SAH_BINS;
            left = act && (a == 0 ? mybin[0] : (a == 1 ? mybin[1] : mybin[2])) <= sp;
        }
        else left = act && lane < count / 2;
        const unsigned long long ml = __ballot(left), mr = __ballot(act && !left);
        const int lc = __popcll(ml);
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (act) s_ord[buf ^ 1][first + (left ? __popcll(ml & lt) : lc + __popcll(mr & lt))] = li;
        __syncthreads();
        /* the other buffer now holds this range partitioned; children inherit it */
        if (lane == 0)
        {
            int code[2];
            const int cf[2] = {first, first + lc}, cc[2] = {lc, count - lc};
            for (int k = 0; k < 2; ++k)
            {
                if (cc[k] == 1) { code[k] = ~(root.first + cf[k]); s_ord[buf][cf[k]] = s_ord[buf ^ 1][cf[k]]; /* keep both buffers' view of a finished slot */ }
                else
                {
                    const int cid = id_base + ids_used++;
                    parent_inner[cid] = id;
                    code[k] = cid;
                    const int t = s_sp++;
                    s_stack[t][0] = cf[k]; s_stack[t][1] = cc[k]; s_stack[t][2] = cid; s_stack[t][3] = buf ^ 1;
                }
            }
            children[id] = make_int2(code[0], code[1]);
            /* final resting place of single references: slot in buffer buf^1 (written above) */
        }
        ids_used = __shfl(ids_used, 0);
        __syncthreads();
    }
    /* every slot ended as a leaf of some range; its final local index is in the buffer that range was partitioned into.
     * Leaves were created from buffer buf^1 of their parent range: collect them by replaying which buffer is final per
     * slot is avoided by keeping s_final: written below during the loop would need more LDS; instead: a leaf slot's two
     * buffers were made equal above (s_ord[buf][slot] = s_ord[buf^1][slot]), and later ranges never touch that slot. */
    if (lane < root.count) order[root.first + lane] = s_ref[s_ord[0][lane]];
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
                                 uint32_t* __restrict__ recs /* 12 words per record, 4 * WIDE_STRIDE apart */)
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
            uint32_t* Lr = recs + 4 * WIDE_STRIDE * (size_t)(base + (unsigned int)k);
            for (int i = 0; i < 9; ++i) Lr[i] = __float_as_uint(tv[i]);
            Lr[9] = (uint32_t)ti;
            Lr[10] = Lr[11] = 0u;
        }
    }
    uint32_t* R = recs + 4 * WIDE_STRIDE * (size_t)wk.out;
    R[0] = __float_as_uint(lo[0]); R[1] = __float_as_uint(lo[1]); R[2] = __float_as_uint(lo[2]);
    R[3] = ebits[0] | (ebits[1] << 8) | (ebits[2] << 16);
    R[4] = base; R[5] = meta; R[6] = q[0]; R[7] = q[1];
    R[8] = q[2]; R[9] = q[3]; R[10] = q[4]; R[11] = q[5];
}
/* the consumed queue is emptied for its next use two levels later */
__global__ void k_collapse_swap(int level, CollapseState* st) { st->count[level & 1] = 0u; }

}  // namespace rt
