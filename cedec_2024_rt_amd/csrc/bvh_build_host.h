/*
 * bvh_build_host.h — host-side parts of the BVH build (included once, by restir_rt.hip):
 * early split clipping of large triangles into references, the binned-SAH binary builder
 * (high-quality build, cf. hiprtBuildFlagBitPreferHighQualityBuild, common/loader.hpp:98-99) and
 * the collapse of a binary tree (from either builder) into the 4-wide quantised records of bvh.h.
 * No HIP calls and no context in here: plain functions over std::vector.
 */
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <vector>

#include "../../include/restir_rt.h"
#include "bvh.h"

using namespace rt;

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
    uint32_t w[4 * WIDE_STRIDE]; /* 12 words used */
}; /* 48 B (64 with RT_WIDE_STRIDE = 4) */
static_assert(sizeof(WideRec) == 16 * WIDE_STRIDE, "wide record");

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
static int collapse_wide(const std::vector<BvhNode>& bin, const rt_triangle* tris, std::vector<WideRec>& recs, int bfs_records)
{
    struct Work { int bin; uint32_t out; int depth; };
    recs.clear();
    recs.reserve(bin.size() * 2 + 8);
    recs.push_back(WideRec());
    /* the first bfs_records records are emitted breadth-first (top levels contiguous at the
     * front: the traversal can stage them in LDS), the rest depth-first (subtrees contiguous) */
    std::deque<Work> stack;
    stack.push_back({0, 0u, 1});
    int height = 1;
    while (!stack.empty())
    {
        Work wk;
        if (recs.size() < (size_t)bfs_records) { wk = stack.front(); stack.pop_front(); }
        else { wk = stack.back(); stack.pop_back(); }
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
        /* any-hit walks visit hit children in slot order: smallest box first (A/B on the bench frame:
         * ascending area -1 % frame, descending +1 %, leaves first/last = ascending; closest-hit walks
         * sort by entry distance and do not care) */
        std::sort(ch, ch + n, [](const WideChild& x, const WideChild& y) { return box_area6(x.lo, x.hi) < box_area6(y.lo, y.hi); });
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

#ifdef RT_EXPERIMENTS /* builder 1: the host reference of the device SAH builder (3), A/B and tests only */
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
        constexpr int NB = 32; /* 16 -> 32 bins: -3 % on the resolve shadow rays, nothing elsewhere; an exact sweep below 64/512 references: +-0 */
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
#endif /* RT_EXPERIMENTS */
