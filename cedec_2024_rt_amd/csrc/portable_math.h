/*
 * portable_math.h — deterministic single-precision transcendental functions.
 *
 * WHY: the reference's kernels call device libm (`log`, `cos`, `sin` in
 * common/reservoir.hpp:89-95, `expf`/`powf` in common/reservoir.hpp:61-75,
 * `cos`/`sin` in common/core.hpp:76-89, `powf` in common/kernels/common.cu:58-61).
 * Device libm (ocml) and glibc disagree in the last bits, and ONE flipped
 * reservoir decision changes a whole pixel (SURVEY.md §7 "discrete-decision
 * parity"), so the GPU path and its CPU checker must evaluate these functions
 * with identical results. Everything here is built from IEEE-754 +,-,*,/ and
 * integer bit manipulation only, in a fixed operation order; it must be compiled
 * with -ffp-contract=off on both sides (no FMA contraction, no fast-math).
 *
 * The polynomial kernels are the classic fdlibm/msun ones (Sun Microsystems,
 * "Permission to use, copy, modify, and distribute this software is freely
 * granted, provided that this notice is preserved"), restated for this file;
 * sin/cos reduce and evaluate in binary64, log/exp in binary32.
 *
 * Accuracy (checked in tests/test_portable_math.py against glibc): <= 1 ulp for
 * logf/expf on the ranges the renderer uses, <= 1 ulp for sinf/cosf on
 * [-2*pi, 4*pi].
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

/* binary64 kernels on [-pi/4, pi/4] */
PM_FN double pm_sin_kernel(double x)
{
    const double S1 = -0.166666666416265235595;    /* -0x15555554cbac77.0p-55 */
    const double S2 = 0.0083333293858894631756;    /*  0x111110896efbb2.0p-59 */
    const double S3 = -0.000198393348360966317347; /* -0x1a00f9e2cae774.0p-65 */
    const double S4 = 0.0000027183114939898219064; /*  0x16cd878c3b46a7.0p-71 */
    const double z = x * x;
    const double w = z * z;
    const double r = S3 + z * S4;
    const double s = z * x;
    return (x + s * (S1 + z * S2)) + (s * w) * r;
}
PM_FN double pm_cos_kernel(double x)
{
    const double C0 = -0.499999997251031003120;    /* -0x1ffffffd0c5e81.0p-54 */
    const double C1 = 0.0416666233237390631894;    /*  0x155553e1053a42.0p-57 */
    const double C2 = -0.00138867637746099294692;  /* -0x16c087e80f1e27.0p-62 */
    const double C3 = 0.0000243904487962774090654; /*  0x199342e0ee5069.0p-68 */
    const double z = x * x;
    const double w = z * z;
    const double r = C2 + z * C3;
    return ((1.0 + z * C0) + w * C1) + (w * z) * r;
}

/* quadrant reduction in binary64; valid (sub-ulp) for |x| < ~1e6 */
PM_FN int pm_rem_pio2(float x, double* y)
{
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079631090164184570e+00;  /* first 25 bits of pi/2 */
    const double pio2_1t = 1.58932547735281966916e-08; /* pi/2 - pio2_1 */
    const double xd = (double)x;
    const double q = xd * invpio2;
    const int n = (int)(q + (q < 0.0 ? -0.5 : 0.5));
    const double fn = (double)n;
    *y = (xd - fn * pio2_1) - fn * pio2_1t;
    return n;
}

PM_FN float pm_sinf(float x)
{
    const uint32_t ax = pm_f2u(x) & 0x7fffffffu;
    if (ax >= 0x7f800000u) return pm_u2f(0x7fc00000u);
    if (ax > 0x4e000000u) return 0.0f; /* |x| > 2^29: out of contract, defined as 0 */
    double y;
    const int n = pm_rem_pio2(x, &y);
    switch (n & 3)
    {
        case 0: return (float)pm_sin_kernel(y);
        case 1: return (float)pm_cos_kernel(y);
        case 2: return (float)(-pm_sin_kernel(y));
        default: return (float)(-pm_cos_kernel(y));
    }
}

PM_FN float pm_cosf(float x)
{
    const uint32_t ax = pm_f2u(x) & 0x7fffffffu;
    if (ax >= 0x7f800000u) return pm_u2f(0x7fc00000u);
    if (ax > 0x4e000000u) return 1.0f; /* out of contract, defined as 1 */
    double y;
    const int n = pm_rem_pio2(x, &y);
    switch (n & 3)
    {
        case 0: return (float)pm_cos_kernel(y);
        case 1: return (float)(-pm_sin_kernel(y));
        case 2: return (float)(-pm_cos_kernel(y));
        default: return (float)pm_sin_kernel(y);
    }
}

/* pm_sinf(x) and pm_cosf(x) at once: one reduction and one evaluation of each kernel instead of two
 * of each (callers always need both of the same angle, and on a wavefront the quadrants of the lanes
 * differ, so the separate functions evaluate both kernels twice). Bit-identical to the two calls:
 * same reduced argument, same kernels, and (float)(-k) == -(float)k. */
PM_FN void pm_sincosf(float x, float* sn, float* cs)
{
    const uint32_t ax = pm_f2u(x) & 0x7fffffffu;
    if (ax >= 0x7f800000u) { *sn = pm_u2f(0x7fc00000u); *cs = pm_u2f(0x7fc00000u); return; }
    if (ax > 0x4e000000u) { *sn = 0.0f; *cs = 1.0f; return; }
    double y;
    const int n = pm_rem_pio2(x, &y);
    const float s = (float)pm_sin_kernel(y), c = (float)pm_cos_kernel(y);
    const float a = (n & 1) ? c : s;  /* |sin| source */
    const float b = (n & 1) ? s : c;  /* |cos| source */
    *sn = (n & 2) ? -a : a;
    *cs = ((n + 1) & 2) ? -b : b;
}

/* x^8 by three squarings: the portable definition of powf(x, 8.0f)
 * (common/reservoir.hpp:61-65). */
PM_FN float pm_pow8f(float x)
{
    const float x2 = x * x;
    const float x4 = x2 * x2;
    return x4 * x4;
}

/* general x^y for x >= 0 (tone mapping, display only): exp(y*log(x)). */
PM_FN float pm_powf_pos(float x, float y)
{
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : (y == 0.0f ? 1.0f : pm_u2f(0x7f800000u));
    if (x == 1.0f || y == 0.0f) return 1.0f;
    return pm_expf(y * pm_logf(x));
}

#endif /* CEDEC_RT_PORTABLE_MATH_H */
