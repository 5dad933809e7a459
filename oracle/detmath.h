/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see slic_oracle.c header).
 *
 * Deterministic binary64 exp and log from +, -, *, / only (each IEEE-754
 * correctly rounded, no fused multiply-add): < 1e-14 relative error and the
 * same bits under gcc on any x86-64 host and under hipcc on gfx950.  Used where
 * the reference's value is only pinned to a tolerance anyway: the prior's
 * exp (pool_oracle.c), the Gaussian weights of felzenszwalb (fz_oracle.c) and
 * the float64 Lab of the uint8 SLIC (slic_oracle.c, x^2.4 and cbrt as
 * exp(p * log x)).  The float32 Lab of the hot path does NOT use these: it
 * follows the C library bit for bit (glibc_flt32.h).
 */
#ifndef ORC_DETMATH_H
#define ORC_DETMATH_H
#include <stdint.h>
#include <string.h>
#include <math.h>

static inline double det_log_pos(double x)
{
    uint64_t b;
    memcpy(&b, &x, 8);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    memcpy(&m, &b, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    /* atanh series: ln m = 2 s (1 + z/3 + z^2/5 + ... + z^12/25) */
    double p = 1.0 / 25.0;
    p = p * z + 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    double lnm = 2.0 * s * p;
    double ed = (double)e;
    return ed * 6.93147180369123816490e-01 + (ed * 1.90821492927058770002e-10 + lnm);
}

static inline double det_exp(double t)
{
    double kf = floor(t * 1.44269504088896338700e+00 + 0.5);
    double r = (t - kf * 6.93147180369123816490e-01) - kf * 1.90821492927058770002e-10;
    /* Taylor, degree 14, Horner */
    double p = 1.0 / 87178291200.0;
    p = p * r + 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    int k = (int)kf;
    uint64_t b = (uint64_t)(k + 1023) << 52;
    double sc;
    memcpy(&sc, &b, 8);
    return p * sc;
}

#endif
