/* Exhaustive accuracy check of pm_logf and pm_expf (csrc/portable_math.h) against a DOUBLE-PRECISION evaluation — not against
 * themselves and not against a binary32 libm (the oracle takes its transcendental functions from the same
 * header the GPU uses, so a defect there is invisible to GPU-vs-oracle; this pins the header independently):
 *   pm_logf over EVERY binary32 in (0, 1]   (the Box-Muller radius, common/reservoir.hpp:89-95: log(rv0), rv0 in [0, 1); denormals
 *                                            included although PCG::uniformf never makes one) and in (1, 4] (tone mapping:
 *                                            aces() < 1.04, pm_powf_pos takes the log of it),
 *   pm_expf over EVERY binary32 in [-104, 0] (depth_rejection_heuristics, common/reservoir.hpp:67-75: exp(-32 d), d >= 0;
 *                                            below -103.97 the result is 0) and in (0, 0.1] (tone mapping: y log x with x <= 1.04).
 * Reference: glibc log / exp in binary64 (< 1 ulp of binary64, i.e. exact to ~2^-52 relative) — error reported in ulps of the
 * correctly rounded binary32 result. Exit code 0 iff every error is <= 1 ulp (subnormal results of exp: <= 1 ulp of the
 * subnormal spacing 2^-149).
 *   gcc -O2 -ffp-contract=off -fopenmp -o /tmp/logexp_exhaustive tools/logexp_exhaustive.c -lm && /tmp/logexp_exhaustive [stride] */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../cedec_2024_rt_amd/csrc/portable_math.h"

/* error of the binary32 value r against the real number ref, in units of the spacing of binary32 at |ref| */
static double ulp_err(float r, double ref)
{
    if (isinf(ref) || ref == 0.0) return ((double)r == ref) ? 0.0 : 1e30;
    const float fr = (float)ref; /* correctly rounded reference */
    float a = fabsf(fr);
    double sp;
    if (a < 1.17549435e-38f) sp = ldexp(1.0, -149);
    else sp = (double)(nextafterf(a, INFINITY) - a);
    return fabs((double)r - ref) / sp;
}

static int run(const char* what, uint32_t first, uint32_t last, uint32_t signbit, int is_log, unsigned stride)
{
    double worst = 0.0;
    unsigned long long n = 0, not_cr = 0; /* not_cr: results that are not the correctly rounded value */
    uint32_t worst_u = first;
#pragma omp parallel
    {
        double w = 0.0;
        uint32_t wu = first;
        unsigned long long ln = 0, lncr = 0;
#pragma omp for schedule(static, 1 << 18) nowait
        for (long long i = (long long)first; i <= (long long)last; i += stride)
        {
            const uint32_t u = (uint32_t)i | signbit;
            const float x = pm_u2f(u);
            const float r = is_log ? pm_logf(x) : pm_expf(x);
            const double ref = is_log ? log((double)x) : exp((double)x);
            const double e = ulp_err(r, ref);
            if (e > w) { w = e; wu = u; }
            ln += 1;
            lncr += (r != (float)ref);
        }
#pragma omp critical
        {
            if (w > worst) { worst = w; worst_u = wu; }
            n += ln; not_cr += lncr;
        }
    }
    const float wx = pm_u2f(worst_u);
    printf("%-34s %11llu arguments%s: max error %.4f ulp at x = %.9g (0x%08x); %.3f %% of the results are not the correctly rounded value\n", what, n,
           stride > 1 ? " (strided)" : "", worst, (double)wx, worst_u, 100.0 * (double)not_cr / (double)n);
    return worst <= 1.0 ? 0 : 1;
}

int main(int argc, char** argv)
{
    const unsigned stride = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    int bad = 0;
    bad |= run("pm_logf on (0, 1]", 0x00000001u, 0x3f800000u, 0u, 1, stride ? stride : 1);
    bad |= run("pm_logf on (1, 4]", 0x3f800001u, 0x40800000u, 0u, 1, stride ? stride : 1);
    bad |= run("pm_expf on [-104, -0]", 0x00000000u, pm_f2u(104.0f), 0x80000000u, 0, stride ? stride : 1);
    bad |= run("pm_expf on [+0, 0.1]", 0x00000000u, pm_f2u(0.1f), 0u, 0, stride ? stride : 1);
    printf("%s\n", bad ? "FAIL: an error above 1 ulp" : "OK: every result within 1 ulp of the exact value");
    return bad;
}
