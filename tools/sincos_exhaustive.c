/* Exhaustive accuracy check of pm_sincosf (csrc/portable_math.h) against glibc sinf / cosf over EVERY binary32 value in
 * [-2 pi, 4 pi] (the spatial pass calls it on [0, 2 pi)): maximum error in ulps of the glibc result, and the maximum
 * absolute error where |result| < 1e-3.
 *   gcc -O2 -ffp-contract=off -fopenmp -o /tmp/sincos_exhaustive tools/sincos_exhaustive.c -lm && /tmp/sincos_exhaustive */
#include <math.h>
#include <stdio.h>
#include <string.h>
#include "../cedec_2024_rt_amd/csrc/portable_math.h"

static double ulp_of(float r)
{
    float a = fabsf(r);
    if (a < 1.17549435e-38f) a = 1.17549435e-38f;
    return (double)(nextafterf(a, INFINITY) - a);
}

int main(void)
{
    const float lo = -6.2831855f, hi = 12.566371f;
    double max_s = 0, max_c = 0, max_abs = 0;
    unsigned long long n = 0, off1 = 0;
    /* positive and negative halves as runs of consecutive bit patterns */
    for (int half = 0; half < 2; ++half)
    {
        const float end = half ? -lo : hi;
        const uint32_t last = pm_f2u(end);
#pragma omp parallel for reduction(max : max_s, max_c, max_abs) reduction(+ : n, off1) schedule(static, 1 << 20)
        for (uint32_t u = 0; u <= last; ++u)
        {
            const float x = pm_u2f(u | (half ? 0x80000000u : 0u));
            float s, c;
            pm_sincosf(x, &s, &c);
            const float rs = sinf(x), rc = cosf(x);
            const double es = fabs((double)s - (double)rs), ec = fabs((double)c - (double)rc);
            if (fabsf(rs) > 1e-3f) { const double e = es / ulp_of(rs); if (e > max_s) max_s = e; } else if (es > max_abs) max_abs = es;
            if (fabsf(rc) > 1e-3f) { const double e = ec / ulp_of(rc); if (e > max_c) max_c = e; } else if (ec > max_abs) max_abs = ec;
            n += 1;
            off1 += (s != rs) + (c != rc);
        }
    }
    printf("%llu arguments in [%g, %g]: max error sin %.3f ulp, cos %.3f ulp (vs glibc), %llu of %llu results differ from glibc (%.2f %%), "
           "max abs error where |result| < 1e-3: %.3e\n", n, lo, hi, max_s, max_c, off1, 2 * n, 100.0 * off1 / (2.0 * n), max_abs);
    return (max_s <= 1.0 && max_c <= 1.0 && max_abs < 1e-9) ? 0 : 1;
}
