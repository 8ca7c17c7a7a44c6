/*
 * portable_math.h — deterministic single-precision transcendental functions.
 *
 * WHY: the reference's kernels call device libm (`log`, `cos`, `sin` in
 * common/reservoir.hpp:89-95, `expf`/`powf` in common/reservoir.hpp:61-75,
 * `cos`/`sin` in common/core.hpp:76-89, `powf` in common/kernels/common.cu:58-61).
 * Device libm (ocml) and glibc disagree in the last bits, and ONE flipped
 * reservoir decision changes a whole pixel (SURVEY.md §7 "discrete-decision
 * parity"), so the GPU path and its CPU checker must evaluate these functions
 * with identical results. Everything here is built from IEEE-754 +,-,*,/, fma and
 * integer bit manipulation only, in a fixed operation order; it must be compiled
 * with -ffp-contract=off on both sides (no FMA contraction, no fast-math).
 *
 * The log/exp kernels are the classic fdlibm/msun ones (Sun Microsystems,
 * "Permission to use, copy, modify, and distribute this software is freely
 * granted, provided that this notice is preserved"), restated for this file;
 * everything is binary32 (sin/cos since round 3: Cody-Waite reduction with fma and
 * a carried tail, own minimax coefficients; rounds 1-2 used binary64 there).
 *
 * Accuracy (checked in tests/test_portable_math.py against glibc): <= 1 ulp for
 * logf/expf on the ranges the renderer uses, <= 1 ulp for sinf/cosf on
 * [-2*pi, 4*pi] (exhaustively: tools/sincos_exhaustive.c).
 *
 * This header is dual-use: C99 (the oracle includes it for its "portable"
 * math mode) and HIP device code (the product).
 */
#ifndef CEDEC_RT_PORTABLE_MATH_H
#define CEDEC_RT_PORTABLE_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PM_FN __host__ __device__ static inline
#else
#define PM_FN static inline
#endif

/* MEASUREMENT BUILD ONLY (csrc/Makefile target `ocml`, librestir_rt_ocml.so; r06): with -DRT_MATH_OCML the
 * DEVICE side of the functions below is the device libm the reference's kernels get when hiprtc compiles them on an AMD GPU
 * (common/shader.hpp:107-175): ocml's logf / expf / sinf / cosf / powf. Such a library is NOT bit-identical to the oracle; it
 * exists so that tools/ocml_drift.py can measure, on the MI355X itself, how far the product's portable functions are from what
 * "the reference as run" computes (flipped pixels, differing histories, relative L2 per frame). The product never defines it. */
#if defined(RT_MATH_OCML) && defined(__HIP_DEVICE_COMPILE__)
#define PM_OCML 1
#else
#define PM_OCML 0
#endif

PM_FN uint32_t pm_f2u(float f)
{
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}
PM_FN float pm_u2f(uint32_t u)
{
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

/* natural logarithm, binary32. log(+0) = -inf, log(x<0) = NaN. */
PM_FN float pm_logf(float x)
{
#if PM_OCML
    return ::logf(x); /* reservoir.hpp:92 `log(rv0)` on the device = __ocml_log_f32 */
#endif
    const float ln2_hi = 6.9313812256e-01f; /* 0x3f317180 */
    const float ln2_lo = 9.0580006145e-06f; /* 0x3717f7d1 */
    const float Lg1 = 0.66666662693f;       /* 0xaaaaaa.0p-24 */
    const float Lg2 = 0.40000972152f;       /* 0xccce13.0p-25 */
    const float Lg3 = 0.28498786688f;       /* 0x91e9ee.0p-25 */
    const float Lg4 = 0.24279078841f;       /* 0xf89e26.0p-26 */

    uint32_t ix = pm_f2u(x);
    int k = 0;
    if (ix < 0x00800000u || (ix >> 31))
    {
        if ((ix << 1) == 0) return pm_u2f(0xff800000u); /* -inf */
        if (ix >> 31) return pm_u2f(0x7fc00000u);       /* NaN */
        /* subnormal: scale up by 2^25 */
        k -= 25;
        x = x * 33554432.0f;
        ix = pm_f2u(x);
    }
    else if (ix >= 0x7f800000u) { return x; }
    else if (ix == 0x3f800000u) { return 0.0f; }

    /* reduce x into [sqrt(2)/2, sqrt(2)) */
    ix += 0x3f800000u - 0x3f3504f3u;
    k += (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    x = pm_u2f(ix);

    const float f = x - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float w = z * z;
    const float t1 = w * (Lg2 + w * Lg4);
    const float t2 = z * (Lg1 + w * Lg3);
    const float R = t2 + t1;
    const float hfsq = (0.5f * f) * f;
    const float dk = (float)k;
    return (((s * (hfsq + R) + dk * ln2_lo) - hfsq) + f) + dk * ln2_hi;
}

/* multiply y by 2^k, k in [-200, 200], deterministic two-step for tiny results */
PM_FN float pm_scale2f(float y, int k)
{
    if (k > 127)
    {
        y = y * pm_u2f(0x7f000000u); /* 2^127 */
        k -= 127;
        if (k > 127) k = 127;
    }
    else if (k < -126)
    {
        y = y * pm_u2f(0x0c800000u); /* 2^-102 */
        k += 102;
        if (k < -126) k = -126;
    }
    return y * pm_u2f((uint32_t)(0x7f + k) << 23);
}

/* e^x, binary32. */
PM_FN float pm_expf(float x)
{
#if PM_OCML
    return ::expf(x); /* reservoir.hpp:71 */
#endif
    const float ln2hi = 6.9314575195e-1f; /* 0x3f317200 */
    const float ln2lo = 1.4286067653e-6f; /* 0x35bfbe8e */
    const float invln2 = 1.4426950216e+0f;
    const float P1 = 1.6666625440e-1f;
    const float P2 = -2.7667332906e-3f;

    uint32_t hx = pm_f2u(x);
    const int sign = (int)(hx >> 31);
    hx &= 0x7fffffffu;

    if (hx >= 0x42aeac50u) /* |x| >= 87.33655 or NaN */
    {
        if (hx > 0x7f800000u) return x; /* NaN */
        if (hx >= 0x42b17218u && !sign) return pm_u2f(0x7f800000u); /* overflow */
        if (sign && hx >= 0x42cff1b5u) return 0.0f;                 /* underflow */
    }

    float hi, lo;
    int k;
    if (hx > 0x3eb17218u) /* |x| > 0.5 ln2 */
    {
        if (hx > 0x3f851592u) /* |x| > 1.5 ln2 */
        {
            k = (int)(invln2 * x + (sign ? -0.5f : 0.5f));
        }
        else { k = 1 - sign - sign; }
        hi = x - (float)k * ln2hi;
        lo = (float)k * ln2lo;
        x = hi - lo;
    }
    else if (hx > 0x39000000u) /* |x| > 2^-14 */
    {
        k = 0;
        hi = x;
        lo = 0.0f;
    }
    else { return 1.0f + x; }

    const float xx = x * x;
    const float c = x - xx * (P1 + xx * P2);
    const float y = 1.0f + (((x * c) / (2.0f - c) - lo) + hi);
    if (k == 0) return y;
    return pm_scale2f(y, k);
}

/* ---- sin / cos, binary32 throughout (r03; rounds 1-2 reduced and evaluated in binary64: 33 half-rate instructions
 * per call on the device, five calls per pixel and spatial pass). IEEE +, -, * and fma (one rounding: the same bits
 * on every IEEE machine; __builtin_fmaf is v_fma_f32 on the device and glibc's correctly rounded fmaf on the host).
 *
 * Reduction (Cody-Waite, three constants): n = rint(x * 2/pi), r = x - n*pi/2 carried as hi + lo.
 *   P1 has 8 significant bits, so t1 = fma(-n, P1, x) is exact for |x| <= 2^15;
 *   hi = fma(-n, P2, t1) rounds once; its rounding error is recovered (lo = fma(-n, P2, t1 - hi), t1 - hi exact) and
 *   the third constant is folded in (lo = fma(-n, P3, lo)): r = hi + lo to ~2^-48 relative, pi/2 = P1+P2+P3 to 8e-20.
 * Kernels on |r| <= pi/4 (+ a rounding): fdlibm's k_sin / k_cos shape with the tail `lo`, minimax coefficients for
 * binary32 (sin: 4 terms, |error| < 3.3e-11 relative; cos: 3 terms after 1 - z/2, < 1e-9 absolute); the last operation
 * of each is one addition whose second operand is < 0.11 |result|, so the result is within ~0.8 ulp of the exact one.
 * Accuracy against glibc, checked exhaustively over every binary32 in [-2 pi, 4 pi] (tools/sincos_exhaustive.c) and in
 * tests/test_portable_math.py: <= 1 ulp. Contract range |x| <= 32768; beyond it the functions are defined as (0, 1). */
PM_FN void pm_sincosf(float x, float* sn, float* cs)
{
#if PM_OCML
    *sn = ::sinf(x); *cs = ::cosf(x); /* reservoir.hpp:93-94, core.hpp:84-86: two calls, as the reference makes them */
    return;
#endif
    const uint32_t ax = pm_f2u(x) & 0x7fffffffu;
    if (ax >= 0x7f800000u) { *sn = pm_u2f(0x7fc00000u); *cs = pm_u2f(0x7fc00000u); return; }
    if (ax > 0x47000000u) { *sn = 0.0f; *cs = 1.0f; return; } /* |x| > 32768: out of contract */
    const float INVPIO2 = 0.6366197466850281f; /* 0x3f22f983 */
    const float P1 = 1.5703125f;               /* 0x3fc90000: 8 significant bits of pi/2 */
    const float P2 = 4.838267923332751e-4f;    /* 0x39fdaa22: pi/2 - P1 */
    const float P3 = 2.5633440682570896e-12f;  /* 0x2c34611a: pi/2 - P1 - P2 */
    const float S1 = -0.1666666716337204f, S2 = 0.008333331905305386f, S3 = -1.983999100048095e-4f, S4 = 2.723765874179662e-6f;
    const float C1 = 0.0416666641831398f, C2 = -0.001388825592584908f, C3 = 2.4537857825635e-5f;

    const float q = x * INVPIO2;
    /* round to nearest integer, ties away from zero (exact: |q| < 2^15): trunc(q +- 0.5) */
    const int n = (int)(q + (q < 0.0f ? -0.5f : 0.5f));
    const float fn = (float)n;
    const float t1 = __builtin_fmaf(-fn, P1, x);
    const float hi = __builtin_fmaf(-fn, P2, t1);
    float lo = __builtin_fmaf(-fn, P2, t1 - hi);
    lo = __builtin_fmaf(-fn, P3, lo);

    const float z = hi * hi;
    /* sin(hi + lo) = hi - ((z*(lo/2 - v*r) - lo) - v*S1),  v = z*hi,  r = S2 + z*(S3 + z*S4) */
    const float v = z * hi;
    float r = __builtin_fmaf(z, S4, S3);
    r = __builtin_fmaf(z, r, S2);
    float ts = __builtin_fmaf(-v, r, 0.5f * lo);
    ts = __builtin_fmaf(z, ts, -lo);
    ts = __builtin_fmaf(-v, S1, ts);
    const float s = hi - ts;
    /* cos(hi + lo) = w + (((1 - w) - z/2) + (z*(z*rc) - hi*lo)),  w = 1 - z/2,  rc = C1 + z*(C2 + z*C3) */
    float rc = __builtin_fmaf(z, C3, C2);
    rc = __builtin_fmaf(z, rc, C1);
    const float hz = 0.5f * z;
    const float w = 1.0f - hz;
    const float tc = __builtin_fmaf(z, z * rc, -(hi * lo));
    const float c = w + (((1.0f - w) - hz) + tc);

    const float a = (n & 1) ? c : s;  /* |sin| source */
    const float b = (n & 1) ? s : c;  /* |cos| source */
    *sn = (n & 2) ? -a : a;
    *cs = ((n + 1) & 2) ? -b : b;
}
/* the single functions are the fused one (bit-identical by construction) */
PM_FN float pm_sinf(float x)
{
    float s, c;
    pm_sincosf(x, &s, &c);
    return s;
}
PM_FN float pm_cosf(float x)
{
    float s, c;
    pm_sincosf(x, &s, &c);
    return c;
}

/* x^8 by three squarings: the portable definition of powf(x, 8.0f)
 * (common/reservoir.hpp:61-65). */
PM_FN float pm_pow8f(float x)
{
#if PM_OCML
    return ::powf(x, 8.0f); /* reservoir.hpp:64 `powf(max(dot, 0), 8)` */
#endif
    const float x2 = x * x;
    const float x4 = x2 * x2;
    return x4 * x4;
}

/* general x^y for x >= 0 (tone mapping, display only): exp(y*log(x)). */
PM_FN float pm_powf_pos(float x, float y)
{
#if PM_OCML
    return ::powf(x, y); /* common/kernels/common.cu:58-61 */
#endif
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : (y == 0.0f ? 1.0f : pm_u2f(0x7f800000u));
    if (x == 1.0f || y == 0.0f) return 1.0f;
    return pm_expf(y * pm_logf(x));
}

#endif /* CEDEC_RT_PORTABLE_MATH_H */
