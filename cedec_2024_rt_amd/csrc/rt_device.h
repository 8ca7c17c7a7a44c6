/*
 * rt_device.h — device-side vocabulary of the MI355X ReSTIR DI path: vector math, PCG,
 * the reference's small pure functions, and the HBM record layouts.
 *
 * Parity rule: every function marked [parity] must produce bit-identical results to the
 * reference's arithmetic (IEEE binary32, the reference's operation order, no FMA
 * contraction; the translation unit is compiled with -ffp-contract=off). Code that only
 * has to be conservative (BVH box tests) may use explicit __builtin_fmaf.
 */
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_DEV __device__ __forceinline__
#define RT_HD __host__ __device__ __forceinline__
#else
/* host-only translation units (app/restir_main.cpp through host_path.h: BASELINE config #1, the 04_ao host loop) compile
 * the [parity] functions below as plain inline C++ (g++ -ffp-contract=off); the HBM record layouts are device-side only */
#include <math.h>
#define RT_HD inline __attribute__((always_inline))
#endif

#include "portable_math.h"

namespace rt
{

struct f3
{
    float x, y, z;
};
RT_HD f3 F3(float x, float y, float z) { return f3{x, y, z}; }
RT_HD f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
RT_HD f3 operator*(float s, f3 a) { return F3(a.x * s, a.y * s, a.z * s); }
RT_HD f3 operator/(f3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
RT_HD f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }

/* common/math.hpp:109-130 [parity] */
RT_HD f3 cross(f3 a, f3 b)
{
    return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_HD float sqrt_guarded(float x); /* sqrtf, through sqrt_in_range on the device where the argument admits it (below) */
RT_HD float length(f3 a) { return sqrt_guarded(dot(a, a)); }
RT_HD f3 normalize(f3 a) { return a / length(a); }
RT_HD f3 mix(f3 a, f3 b, float t) { return a + (b - a) * t; }
RT_HD float luminance(f3 a) { return dot(a, F3(0.1762044f, 0.8129847f, 0.0108109f)); }

constexpr float kPI = 3.14159265358979323846f;
constexpr float kFltMax = 3.402823466e+38f;

RT_HD float as_float(int i) { return pm_u2f((uint32_t)i); }
RT_HD float as_float(uint32_t i) { return pm_u2f(i); }
RT_HD int as_int(float f) { return (int)pm_f2u(f); }
RT_HD uint32_t as_uint(float f) { return pm_f2u(f); }

/* device max(float,float): NaN operand yields the other one (v_max_f32 semantics) */
RT_HD float fmax_dev(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? b : a)); }

/* ------------------------------------------------------------------ RNG */
/* common/rng.hpp:8-58 [parity, integer] */
#ifndef RT_PCG_SERIAL
#define RT_PCG_SERIAL 2 /* 2: every draw takes one LCG step, 1: only the marked loops (uniformf_chained), 0: none */
#endif
struct PCG
{
    uint64_t state, inc;
    /* CHAINED (device): hipcc otherwise folds two steps into A^2 s + (A + 1) c beside the single step the output needs - a
     * shorter dependency chain for 10 multiply-class instructions per two draws instead of 8; the kernels are bound by vector
     * issue, not by that chain (RIS loop of generate_candidate -1.2 %, shadowed frame -0.7 %). */
    template <bool CHAINED = false>
    RT_HD uint32_t uniform()
    {
        const uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
#if defined(__HIP_DEVICE_COMPILE__)
        if (RT_PCG_SERIAL == 2 || (RT_PCG_SERIAL == 1 && CHAINED)) asm("" : "+v"(state));
#endif
        const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        const uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
    }
    template <bool CHAINED = false>
    RT_HD float uniformf()
    {
        const uint32_t bits = (uniform<CHAINED>() >> 9) | 0x3f800000u;
        return pm_u2f(bits) - 1.0f;
    }
    RT_HD float uniformf_chained() { return uniformf<true>(); }
};
RT_HD PCG pcg_init(uint64_t seed, uint64_t sequence)
{
    PCG r;
    r.state = 0u;
    r.inc = (sequence << 1u) | 1u;
    r.uniform();
    r.state += seed;
    r.uniform();
    return r;
}
RT_HD uint32_t hashPCG(uint32_t v)
{
    const uint32_t state = v * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28) + 4)) ^ state) * 277803737u;
    return (word >> 22) ^ word;
}
RT_HD uint32_t hashPCG3(uint32_t x, uint32_t y, uint32_t z) { return hashPCG(hashPCG(hashPCG(x) + y) + z); }
RT_HD uint32_t hashPCG4(uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
    return hashPCG(hashPCG(hashPCG(hashPCG(x) + y) + z) + w);
}

/* ------------------------------------------------ reference pure functions */
/* common/core.hpp:237-252 [parity] */
RT_HD void warp_unit_triangle(float& x, float& y)
{
    if (y > x) { x *= 0.5f; y -= x; }
    else { y *= 0.5f; x -= y; }
}
/* ---- IEEE binary32 division without its scaling and fix-up steps (device only, r04) ----------------------------------
 * hipcc expands `n / d` (no fast-math, fp32 denormals on) into the 11-instruction sequence
 *     ds = v_div_scale(d, d, n); ns = v_div_scale(n, d, n); r0 = v_rcp(ds); e0 = fma(-ds, r0, 1); r1 = fma(e0, r0, r0);
 *     q0 = ns * r1; e1 = fma(-ds, q0, ns); q1 = fma(e1, r1, q0); e2 = fma(-ds, q1, ns); q = v_div_fmas(e2, r1, q1);
 *     result = v_div_fixup(q, d, n)
 * v_div_scale returns its first operand UNCHANGED and clears VCC (so v_div_fmas is a plain fma) unless the denominator is
 * denormal or >= 2^126, the quotient would be denormal or overflow (exponent difference >= 96 or <= -126), the numerator
 * is below 2^-103, or an operand is 0 / inf / NaN; v_div_fixup returns q unchanged (with the sign of n / d, which q has)
 * unless an operand is 0 / inf / NaN or the quotient under- / overflows (CDNA3 ISA guide, V_DIV_SCALE_F32 /
 * V_DIV_FMAS_F32 / V_DIV_FIXUP_F32). So for
 *     2^-40 <= |d| <= 2^40   and   2^-79 <= |n| < 2^56        (exponent difference within [-119, 95])
 * the same result comes from r1 = rcp_refined(d) (3 instructions, once per DENOMINATOR) and div_by(n, d, r1)
 * (5 instructions per numerator) — literally the operations above with ds = d, ns = n. The kernels use it where several
 * numerators share a denominator (normalize: 18 instead of 33 instructions) or the denominator's r1 can be stored
 * (the light table's pdf), behind range tests on the operands' bits; anything outside (zeros, denormals, NaN, huge or
 * tiny values) takes the compiler's division. Checked on the device against `/` over random and edge operands
 * (tests/test_gpu_round4.py::test_guarded_division_is_ieee). RT_FAST_DIV=0 compiles the plain divisions everywhere (A/B). */
#ifndef RT_FAST_DIV
#define RT_FAST_DIV 1
#endif
#if defined(__HIPCC__)
/* (the host pass of hipcc parses device functions too: the two hardware instructions are named for the device pass only) */
#if defined(__HIP_DEVICE_COMPILE__)
RT_DEV float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
RT_DEV float hw_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
RT_DEV float hw_rcp(float x) { return 1.0f / x; }
RT_DEV float hw_sqrt(float x) { return sqrtf(x); }
#endif
RT_DEV float rcp_refined(float d)
{
    const float r0 = hw_rcp(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    return __builtin_fmaf(e0, r0, r0);
}
RT_DEV float div_by(float n, float d, float r1)
{
    const float q0 = n * r1;
    const float e1 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(e2, r1, q1);
}
/* d >= 0 (a sum of squares, a pdf, a weight sum): 2^-40 <= d <= 2^40; zero, negative, inf and NaN bits fall outside */
RT_DEV bool div_den_ok(float d) { return (pm_f2u(d) - 0x2b800000u) <= (0x53800000u - 0x2b800000u); }
/* n >= 0: 2^-79 <= n <= 2^55 (zero takes the compiler's division: rare, and its sign rules are v_div_fixup's) */
RT_DEV bool div_num_ok(float n) { return (pm_f2u(n) - 0x18000000u) <= (0x5b000000u - 0x18000000u); }
constexpr float kDivNumLo = 1.6543612251060553e-24f; /* 2^-79 */
/* sqrtf the same way: hipcc's expansion is scale-up of arguments below 2^-96, v_sqrt_f32, a +-1 ulp correction from two
 * fma residuals, scale-down, and a v_cmp_class fix-up for 0 / inf (15 instructions); for 2^-40 <= x <= 2^40 the scaling
 * and the fix-up are identities and the 9 instructions in between give the same bits */
RT_DEV float sqrt_in_range(float x)
{
    const float s = hw_sqrt(x);
    const float dn = pm_u2f(pm_f2u(s) - 1u), up = pm_u2f(pm_f2u(s) + 1u);
    const float e_dn = __builtin_fmaf(-dn, s, x), e_up = __builtin_fmaf(-up, s, x);
    float r = (0.0f >= e_dn) ? dn : s;
    r = (0.0f < e_up) ? up : r;
    return r;
}
#endif

RT_HD float sqrt_guarded(float x)
{
#if defined(__HIP_DEVICE_COMPILE__) && RT_FAST_DIV
    if (div_den_ok(x)) return sqrt_in_range(x);
#endif
    return sqrtf(x);
}

/* common/core.hpp:287-295 [parity] */
RT_HD float geometry_term(f3 p0, f3 n0, f3 p1, f3 n1)
{
    f3 v = p1 - p0;
    const float sqr_dist = dot(v, v);
#if defined(__HIP_DEVICE_COMPILE__) && RT_FAST_DIV
    /* normalize = three divisions by |v| = sqrt(sqr_dist): one refined reciprocal for the three; then / sqr_dist */
    if (div_den_ok(sqr_dist) && __builtin_fminf(__builtin_fminf(fabsf(v.x), fabsf(v.y)), fabsf(v.z)) >= kDivNumLo)
    {
        const float len = sqrt_in_range(sqr_dist); /* in [2^-20, 2^20]; |v.i| <= len (1 + 2^-22) */
        const float rl = rcp_refined(len);
        v = F3(div_by(v.x, len, rl), div_by(v.y, len, rl), div_by(v.z, len, rl));
        const float num = fabsf(dot(v, n0)) * fabsf(dot(-v, n1));
        if (div_num_ok(num)) return div_by(num, sqr_dist, rcp_refined(sqr_dist));
        return num / sqr_dist;
    }
#endif
    v = normalize(v);
    return fabsf(dot(v, n0)) * fabsf(dot(-v, n1)) / sqr_dist;
}
/* the same with the compiler's divisions and square root only (device self-check of the guarded forms: k_math_eval) */
RT_HD float geometry_term_plain(f3 p0, f3 n0, f3 p1, f3 n1)
{
    f3 v = p1 - p0;
    const float sqr_dist = dot(v, v);
    v = normalize(v);
    return fabsf(dot(v, n0)) * fabsf(dot(-v, n1)) / sqr_dist;
}
/* common/reservoir.hpp:42-59, unshadowed branch; `lum` = luminance(radiance) [parity] */
RT_HD float target_unshadowed(f3 op, f3 on, f3 hp, f3 hn, float lum)
{
    const float brdf = 1.0f / kPI;
    const float G = geometry_term(op, on, hp, hn);
    return brdf * G * lum;
}
/* `u < weight / w_sum` of Reservoir::update / merge (common/reservoir.hpp:22-37) WITHOUT the division where its outcome is certain
 * (r05). The quotient is used for nothing but this comparison, and the IEEE division is 11 vector instructions (a v_rcp_f32, two
 * v_div_scale, v_div_fmas, v_div_fixup among them) in a loop that runs 32 times per pixel at the issue limit. With S = w_sum > 0
 * and u > 0, `u < RN(W / S)` can differ from `u S < W` only if u S lies within a few units in the last place of W:
 *     t = RN(u S) = u S (1 + e1), |e1| <= 2^-24;  RN(W / S) = (W / S)(1 + e2), |e2| <= 2^-24 (normal quotients);
 *     W - t >  2^-20 |W|  =>  u S < W (1 - 2^-20 + 2^-24)  =>  u < (W / S)(1 - 2^-21) < RN(W / S):        accepted, exactly as the division says
 *     W - t < -2^-20 |W|  =>  u > (W / S)(1 + 2^-21) > RN(W / S)  (a subnormal RN(W / S) is < 2^-126 <= u too):   rejected, exactly as the division says
 * (RN(W - t) has the sign of W - t: the difference of two binary32 numbers is a multiple of 2^-149). Everything else — |W - t| within
 * the margin, u = 0 (where the decision is "is the quotient non-zero"), S <= 0, zero / subnormal / inf / NaN operands (the absolute
 * term 2^-120 of the margin and `t > 0` catch them) — takes the division itself. The decision is the reference's bit for bit;
 * 7 of 1 000 000 draws take the slow path on the bench scene. RT_FAST_ACCEPT=0 compiles the plain comparison (A/B). */
#ifndef RT_FAST_ACCEPT
#define RT_FAST_ACCEPT 1
#endif
RT_HD bool reservoir_accept(float u, float weight, float w_sum)
{
#if defined(__HIP_DEVICE_COMPILE__) && RT_FAST_ACCEPT
    const float t = u * w_sum;
    const float d = weight - t;
    const float margin = __builtin_fmaf(fabsf(weight), 9.5367431640625e-07f /* 2^-20 */, 7.52316384526264e-37f /* 2^-120 */);
    if (fabsf(d) > margin && t > 0.0f) return d > 0.0f;
#endif
    return u < weight / w_sum;
}
/* common/reservoir.hpp:61-87 with the portable expf / x^8 [parity] */
RT_HD float rejection_heuristics(f3 p0, f3 n0, f3 p1, f3 n1, f3 eye)
{
    const float d0 = length(p0 - eye);
    const float d1 = length(p1 - eye);
    const float diff = (d1 - d0) * (d1 - d0) / d0;
    float w = 1.0f;
    w *= pm_expf(-32.0f * diff);
    w *= pm_pow8f(fmax_dev(dot(n0, n1), 0.0f));
    return w;
}
/* float -> int as v_cvt_i32_f32 does it (NaN -> 0, saturating); C++ leaves it undefined */
RT_HD int f2i_sat(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    /* the instruction itself: the C++ below costs 8 vector + 14 scalar instructions around the same v_cvt_i32_f32 */
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
#endif
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}
/* `M *= w` for int M, float w (10_restir_di.cu:211-212, 362-363) [parity] */
RT_HD int scale_M(int M, float w) { return f2i_sat((float)M * w); }

/* common/core.hpp:45-68 on raw vertices [parity] */
RT_HD f3 tri_normal(f3 v0, f3 v1, f3 v2) { return normalize(cross(v1 - v0, v2 - v0)); }
RT_HD float tri_area(f3 v0, f3 v1, f3 v2) { return 0.5f * length(cross(v1 - v0, v2 - v0)); }

/* common/core.hpp:91-136 [parity]: the canonical leaf test */
RT_HD bool intersect_ray_triangle(float& tOut, float& uOut, float& vOut, f3 ro, f3 rd, float tmin,
                                  float tmax, f3 v0, f3 v1, f3 v2)
{
    const f3 e0 = v1 - v0;
    const f3 e1 = v2 - v1;
    const f3 e2 = v0 - v2;
    const f3 n = cross(e0, e1);
    const float t = dot(v0 - ro, n) / dot(n, rd);
    if (tmin <= t && t <= tmax)
    {
        const f3 p = ro + rd * t;
        const float a0 = dot(n, cross(e0, p - v0));
        const float a1 = dot(n, cross(e1, p - v1));
        const float a2 = dot(n, cross(e2, p - v2));
        if (a0 < 0.0f || a1 < 0.0f || a2 < 0.0f) return false;
        const float a = a0 + a1 + a2;
        tOut = t;
        uOut = a2 / a; /* bV */
        vOut = a0 / a; /* bW */
        return true;
    }
    return false;
}

#if defined(__HIPCC__)
/* --------------------------------------------------------- HBM record layouts
 *
 * G-buffer (written once per frame by raycast, read by every later pass), 32 B / pixel:
 *   g0 = { p.x, p.y, p.z, bits(triangle index, -1 = sky) }
 *   g1 = { n.x, n.y, n.z, bits(flags) }          flags: GB_SHADED | GB_EMISSIVE
 * p, n = make_surface_info(vis, triangles, eye) (common/core.hpp:189-207).
 *
 * Reservoir record, 64 B aligned (one gather = 4 x dwordx4 from one 64-B segment):
 *   q0 = { hit_position.xyz, ucw }
 *   q1 = { hit_normal.xyz,   bits(M | visibility << 31 | shaded << 30) }
 *   q2 = { origin_position.xyz, luminance(radiance) }
 *   q3 = { origin_normal.xyz,   w_sum }
 * plus a 16-B side record { radiance.xyz, bits(own-visibility flags) } that only moves with the finally
 * selected sample. Together they hold every field of the reference's 76-B Reservoir
 * (common/reservoir.hpp:5-38) losslessly for 0 <= M < 2^30.
 *
 * Own-visibility flags (r03; not part of the reference's Reservoir, never downloaded): OWNV_KNOWN = a kernel of THIS staged
 * frame has already evaluated the shadow ray from THIS pixel's surface point to the sample the record holds, OWNV_VISIBLE =
 * its answer. The reference traces that same ray again in the next kernel (the own sample's p-hat of the next spatial pass
 * under the shadowed target function, 10_restir_di.cu:370-378; resolve's "fresh" ray, :443-444): same origin, same
 * target, same frame, same scene => same answer, so the kernels that find the flag set skip the walk.
 * r04: the flags carry a TAG (bits 31..2) naming the staged frame they were computed in — one number per rt_frame /
 * rt_frame_stage sequence and per pipelined stage 0, handed to the kernels in FrameParams::ownv_tag. A reader trusts
 * OWNV_KNOWN only under its own tag, so a record an EARLIER frame wrote (a frame without spatial passes resolves
 * reservoir_buffer1, which it did not write; a camera or option change in between) is walked afresh, as the reference
 * does. Tag 0 = "trust nothing, record nothing": the per-kernel entry points (rt_generate_candidate, rt_spatial_resampling,
 * rt_resolve, ...), where the caller may change camera, G-buffer or buffers between any two calls. Uploads store 0.
 */
constexpr uint32_t GB_SHADED = 1u;
constexpr uint32_t GB_EMISSIVE = 2u;
constexpr uint32_t RES_VIS_BIT = 0x80000000u;
constexpr uint32_t RES_SHADED_BIT = 0x40000000u;
constexpr uint32_t RES_M_MASK = 0x3fffffffu;
constexpr uint32_t OWNV_KNOWN = 1u;
constexpr uint32_t OWNV_VISIBLE = 2u;
#ifndef RT_OWNV
#define RT_OWNV 1 /* 0: never set the flags (A/B) */
#endif
RT_HD uint32_t ownv_of(uint32_t tag, bool visible) { return (RT_OWNV && tag) ? ((tag << 2) | OWNV_KNOWN | (visible ? OWNV_VISIBLE : 0u)) : 0u; }
/* the flags as this launch may use them: 0 unless they were written under this launch's tag */
RT_HD uint32_t ownv_trusted(uint32_t tag, uint32_t flags) { return (tag != 0u && (flags >> 2) == tag) ? (flags & (OWNV_KNOWN | OWNV_VISIBLE)) : 0u; }

struct Res
{
    f3 hit_p, hit_n, org_p, org_n, rad;
    float ucw, lum, w_sum;
    int M;
    bool vis;
    uint32_t ownv; /* own-visibility flags of the side record (not loaded by res_load) */
};
RT_HD Res res_zero()
{
    Res r;
    r.hit_p = r.hit_n = r.org_p = r.org_n = r.rad = F3(0.0f, 0.0f, 0.0f);
    r.ucw = r.lum = r.w_sum = 0.0f;
    r.M = 0;
    r.vis = false;
    r.ownv = 0u;
    return r;
}

/* Stores of the unshadowed frame's reservoir records (RT_NT_STORE bit 0: the wavefront's cooperative scatter, bit 1: the radiance
 * side records beside it) are non-temporal: 166 MB per launch that the launch does not read again would otherwise push the
 * records the gathers want out of the L2 (spatial pass -5 %, frame -1.5 %; loads with the hint: slower everywhere). Not in the
 * shadowed-target kernels (a 1.5-ms pass finds the previous pass's records in the 256-MB MALL: +1.2 % with the hint). */
#ifndef RT_NT_STORE
#define RT_NT_STORE 3
#endif
typedef float rt_v4f __attribute__((ext_vector_type(4)));
template <int BIT>
RT_DEV void store_stream(float4* p, const float4& v)
{
    if (RT_NT_STORE & BIT) __builtin_nontemporal_store(rt_v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<rt_v4f*>(p));
    else *p = v;
}
RT_DEV void res_store(float4* __restrict__ rec, float4* __restrict__ radb, size_t i, const Res& r, bool shaded)
{
    const uint32_t mbits = ((uint32_t)r.M & RES_M_MASK) | (r.vis ? RES_VIS_BIT : 0u) | (shaded ? RES_SHADED_BIT : 0u);
    rec[4 * i + 0] = make_float4(r.hit_p.x, r.hit_p.y, r.hit_p.z, r.ucw);
    rec[4 * i + 1] = make_float4(r.hit_n.x, r.hit_n.y, r.hit_n.z, as_float(mbits));
    rec[4 * i + 2] = make_float4(r.org_p.x, r.org_p.y, r.org_p.z, r.lum);
    rec[4 * i + 3] = make_float4(r.org_n.x, r.org_n.y, r.org_n.z, r.w_sum);
    radb[i] = make_float4(r.rad.x, r.rad.y, r.rad.z, as_float(r.ownv));
}
RT_DEV Res res_from_parts(const float4& q0, const float4& q1, const float4& q2, const float4& q3, bool& shaded)
{
    Res r;
    r.hit_p = F3(q0.x, q0.y, q0.z);
    r.ucw = q0.w;
    r.hit_n = F3(q1.x, q1.y, q1.z);
    const uint32_t mb = as_uint(q1.w);
    r.M = (int)(mb & RES_M_MASK);
    r.vis = (mb & RES_VIS_BIT) != 0u;
    shaded = (mb & RES_SHADED_BIT) != 0u;
    r.org_p = F3(q2.x, q2.y, q2.z);
    r.lum = q2.w;
    r.org_n = F3(q3.x, q3.y, q3.z);
    r.w_sum = q3.w;
    r.rad = F3(0.0f, 0.0f, 0.0f);
    r.ownv = 0u;
    return r;
}
/* loads everything except radiance; q = the record's four float4 (in a reservoir buffer or in a halo list) */
RT_DEV Res res_load_at(const float4* __restrict__ q, bool& shaded)
{
    const float4 q0 = q[0];
    const float4 q1 = q[1];
    const float4 q2 = q[2];
    const float4 q3 = q[3];
    return res_from_parts(q0, q1, q2, q3, shaded);
}
RT_DEV Res res_load(const float4* __restrict__ rec, size_t i, bool& shaded) { return res_load_at(rec + 4 * i, shaded); }

/* Halo records without pack / unpack launches (multi-GPU strips, r03). A strip exchanges with each neighbour the records
 * marked in a bitmap (frame_kernels.h, k_halo_mark), as a dense list in bitmap order: list entry = the 4 float4 of the
 * record + its radiance side record. UNPACK side: a spatial pass that gathers a neighbour from a halo row reads it from
 * the received list (index = prefix count of the bitmap word + bits below) instead of from halo rows an unpack kernel
 * filled. PACK side: a pass that writes a record of a row its neighbour's halo covers also writes it to the send list if
 * the neighbour marked it. Bitmap layout: [0] count, [1 .. nw] bits, [1+nw .. 1+2nw) exclusive prefix per word. */
struct HaloFuse
{
    const uint32_t* need_bm[2]; /* side 0 = rows below the strip, 1 = above; nullptr = none */
    const float4* recv[2];
    const uint32_t* give_bm[2];
    float4* send[2];
    int need_row0[2], give_row0[2];
    int rows, nw; /* rows per region, bitmap words = ceil(rows * W / 32) */
};
RT_DEV uint32_t halo_list_index(const uint32_t* __restrict__ bm, int nw, uint32_t i, bool& marked)
{
    const uint32_t word = bm[1 + (i >> 5)];
    marked = (word >> (i & 31u)) & 1u;
    return bm[1 + nw + (i >> 5)] + (uint32_t)__popc(word & ((1u << (i & 31u)) - 1u));
}
/* where the record of pixel (nx, nrow) = buffer index pid lives: the reservoir buffer, or the received list */
RT_DEV const float4* halo_record(const HaloFuse& F, int W, const float4* __restrict__ in_rec, const float4* __restrict__ in_rad, size_t pid,
                                 int nx, int nrow, const float4*& rad)
{
    const float4* q = in_rec + 4 * pid;
    rad = in_rad + pid;
#pragma unroll
    for (int s = 0; s < 2; ++s)
        if (F.recv[s] && (unsigned)(nrow - F.need_row0[s]) < (unsigned)F.rows)
        {
            bool marked;
            const uint32_t idx = halo_list_index(F.need_bm[s], F.nw, (uint32_t)(nrow - F.need_row0[s]) * (uint32_t)W + (uint32_t)nx, marked);
            q = F.recv[s] + 5 * (size_t)idx; /* marked by construction: the plan replays exactly these draws */
            rad = q + 4;
        }
    return q;
}
/* the same place as ONE word (a kernel that keeps five of them across a long walk): bits 31..30 = 0 reservoir buffer,
 * 1 / 2 = received list of side 0 / 1; bits 29..0 = the record's first float4 (buffer: 4 x pixel, list: 5 x entry).
 * Buffers of up to 2^28 pixels (rt_create refuses more local pixels than that in strips... the whole-frame limit is the same). */
constexpr uint32_t HALO_CODE_NONE = 0xffffffffu;
RT_DEV uint32_t halo_code(const HaloFuse& F, int W, size_t pid, int nx, int nrow)
{
    uint32_t code = 4u * (uint32_t)pid;
#pragma unroll
    for (int s = 0; s < 2; ++s)
        if (F.recv[s] && (unsigned)(nrow - F.need_row0[s]) < (unsigned)F.rows)
        {
            bool marked;
            const uint32_t idx = halo_list_index(F.need_bm[s], F.nw, (uint32_t)(nrow - F.need_row0[s]) * (uint32_t)W + (uint32_t)nx, marked);
            code = ((uint32_t)(s + 1) << 30) | (5u * idx);
        }
    return code;
}
RT_DEV const float4* halo_code_record(const HaloFuse& F, const float4* __restrict__ in_rec, uint32_t code)
{
    const uint32_t sel = code >> 30;
    const float4* base = sel == 0u ? in_rec : (sel == 1u ? F.recv[0] : F.recv[1]);
    return base + (code & 0x3fffffffu);
}
RT_DEV const float4* halo_code_radiance(const HaloFuse& F, const float4* __restrict__ in_rec, const float4* __restrict__ in_rad, uint32_t code)
{
    return (code >> 30) == 0u ? in_rad + ((code & 0x3fffffffu) >> 2) : halo_code_record(F, in_rec, code) + 4;
}
/* the send list's copy of the record of pixel (x, row), if a neighbour strip marked it */
RT_DEV void res_give(const HaloFuse& F, int W, size_t i, int x, int row, const Res& r, bool shaded)
{
#pragma unroll
    for (int s = 0; s < 2; ++s)
        if (F.send[s] && (unsigned)(row - F.give_row0[s]) < (unsigned)F.rows)
        {
            bool marked;
            const uint32_t idx = halo_list_index(F.give_bm[s], F.nw, (uint32_t)(row - F.give_row0[s]) * (uint32_t)W + (uint32_t)x, marked);
            if (marked)
            {
                const uint32_t mbits = ((uint32_t)r.M & RES_M_MASK) | (r.vis ? RES_VIS_BIT : 0u) | (shaded ? RES_SHADED_BIT : 0u);
                float4* L = F.send[s] + 5 * (size_t)idx;
                L[0] = make_float4(r.hit_p.x, r.hit_p.y, r.hit_p.z, r.ucw);
                L[1] = make_float4(r.hit_n.x, r.hit_n.y, r.hit_n.z, as_float(mbits));
                L[2] = make_float4(r.org_p.x, r.org_p.y, r.org_p.z, r.lum);
                L[3] = make_float4(r.org_n.x, r.org_n.y, r.org_n.z, r.w_sum);
                L[4] = make_float4(r.rad.x, r.rad.y, r.rad.z, 0.0f);
            }
        }
}
/* res_store + the send list */
RT_DEV void res_store_give(const HaloFuse& F, int W, float4* __restrict__ rec, float4* __restrict__ radb, size_t i, int x, int row, const Res& r,
                           bool shaded)
{
    res_store(rec, radb, i, r, shaded);
    res_give(F, W, i, x, row, r, shaded);
}

#endif /* __HIPCC__ */

}  // namespace rt
