/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see slic_oracle.c header).
 *
 * powf(x, y) and cbrtf(x) of GNU libc 2.35, restated.
 *
 * Why libm is part of the reference's semantics: scikit-image 0.18.3 keeps a
 * float32 image in float32 (skimage/color/colorconv.py:657-661 rgb2xyz,
 * :950-969 xyz2lab), so `np.power(arr, 2.4)` raises float32 values to the
 * float32 exponent float32(2.4) = 0x1.333334p+1 (numpy 1.26 value-based
 * casting) and `np.cbrt(arr)` is the float32 loop.  On a host whose numpy does
 * not dispatch to its AVX-512 SVML kernels (any CPU without AVX512_SKX, or
 * NPY_DISABLE_CPU_FEATURES="AVX512F ..." — the ONE reference configuration
 * tests/golden/PROVENANCE.txt names) those loops call the C library's
 * powf / cbrtf.  The library is a third-party dependency that is absent from
 * /root/reference: GNU libc 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.11 in this image,
 * the same on the GPU box).  Its algorithms are published:
 *
 *   powf   sysdeps/ieee754/flt-32/e_powf.c (Szabolcs Nagy, ARM optimized
 *          routines), tables e_powf_log2_data.c (POWF_LOG2_TABLE_BITS = 4,
 *          POWF_SCALE_BITS = 0 on x86-64: TOINT_INTRINSICS is 0) and
 *          e_exp2f_data.c (EXP2F_TABLE_BITS = 5).  log2(x) from a 16-entry
 *          (1/c, log2 c) table and a degree-5 polynomial, y*log2(x) in
 *          binary64, 2^t from a 32-entry table and a cubic, one final rounding
 *          to binary32.  x86-64 selects the build of that file compiled with
 *          -mfma (sysdeps/x86_64/fpu/multiarch/e_powf-fma.c) on every CPU
 *          with FMA3: each a*b+c below is one fused operation there.
 *   cbrtf  sysdeps/ieee754/flt-32/s_cbrtf.c: frexpf, a quadratic start value,
 *          one Halley step in binary64, a factor 2^(±1/3, ±2/3), ldexpf.
 *
 * Pinning: tests/test_oracle_golden.py::test_glibc_restatement_vs_host_libm
 * compares both functions with the C library the test host runs (skipped
 * unless it is glibc 2.35); an exhaustive run over every float32 in
 * [1e-3, 1e7] (278 234 130 values, tools/glibc_exhaustive.c) finds 0
 * differences for powf(x, 2.4f) and 0 for cbrtf(x).  With these two the
 * oracle's Lab image is BIT IDENTICAL to skimage.color.rgb2lab on every
 * fixture under the reference configuration.
 *
 * Domain: x finite, positive, normal (the callers guarantee x > 0.0905 for
 * powf, x > 0.008856 for cbrtf); +inf is passed through.
 */
#ifndef ORC_GLIBC_FLT32_H
#define ORC_GLIBC_FLT32_H
#include <stdint.h>
#include <string.h>
#include <math.h>

/* e_powf_log2_data.c: tab[i] = {invc, logc}: invc ~ 1/c, logc = log2(c) for
 * c near the centre of [2^-0.5 * 2^(i/16) ...) sub-intervals of OFF..2*OFF */
static const double glibc_powf_log2_tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};
static const double glibc_powf_log2_poly[5] = {
    0x1.27616c9496e0bp-2, -0x1.71969a075c67ap-2, 0x1.ec70a6ca7baddp-2,
    -0x1.7154748bef6c8p-1, 0x1.71547652ab82bp0,
};
/* e_exp2f_data.c: tab[i] = bits(RN(2^(i/32))) - (i << 47) */
static const uint64_t glibc_exp2f_tab[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL,
};
static const double glibc_exp2f_poly[3] = {
    0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3, 0x1.62e42ff0c52d6p-1,
};

/* e_powf.c:__powf for finite positive normal x; fma() = the fused operations of
 * the -mfma build (this file is compiled -ffp-contract=off, so nothing else fuses) */
static inline float glibc_powf(float x, float y)
{
    uint32_t ix;
    memcpy(&ix, &x, 4);
    if (ix == 0x7f800000u) return x;                       /* +inf ** y (y > 0) */
    /* log2_inline */
    uint32_t tmp = ix - 0x3f330000u;
    int i = (int)((tmp >> 19) % 16);
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int k = (int32_t)top >> 23;
    double invc = glibc_powf_log2_tab[i][0], logc = glibc_powf_log2_tab[i][1];
    float zf;
    memcpy(&zf, &iz, 4);
    double z = (double)zf;
    const double *A = glibc_powf_log2_poly;
    double r = fma(z, invc, -1.0);
    double y0 = logc + (double)k;
    double r2 = r * r;
    double q0 = fma(A[0], r, A[1]);
    double p = fma(A[2], r, A[3]);
    double r4 = r2 * r2;
    double q = fma(A[4], r, y0);
    q = fma(p, r2, q);
    double logx = fma(q0, r4, q);
    double ylogx = (double)y * logx;
    if (ylogx > 0x1.fffffffd1d571p+6) return HUGE_VALF;    /* __math_oflowf */
    if (ylogx <= -150.0) return 0.0f;                      /* __math_uflowf */
    /* exp2_inline */
    const double SHIFT = 0x1.8p+52 / 32;
    double kd = ylogx + SHIFT;
    uint64_t ki;
    memcpy(&ki, &kd, 8);
    kd -= SHIFT;
    double rr = ylogx - kd;
    uint64_t t = glibc_exp2f_tab[ki % 32];
    t += ki << 47;
    double s;
    memcpy(&s, &t, 8);
    const double *C = glibc_exp2f_poly;
    double zz = fma(C[0], rr, C[1]);
    double rr2 = rr * rr;
    double w = fma(C[2], rr, 1.0);
    w = fma(zz, rr2, w);
    w = w * s;
    return (float)w;
}

/* s_cbrtf.c:__cbrtf for finite positive normal x */
static inline float glibc_cbrtf(float x)
{
    static const double factor[5] = {
        1.0 / 1.5874010519681994748, 1.0 / 1.2599210498948731648, 1.0,
        1.2599210498948731648, 1.5874010519681994748,
    };
    uint32_t ix;
    memcpy(&ix, &x, 4);
    if (ix == 0x7f800000u) return x;
    /* frexpf: x = xm * 2^xe, xm in [0.5, 1) */
    int xe = (int)(ix >> 23) - 126;
    uint32_t im = (ix & 0x007fffffu) | 0x3f000000u;
    float xm;
    memcpy(&xm, &im, 4);
    float u = (float)(0.492659620528969547 + (0.697570460207922770 - 0.191502161678719066 * xm) * xm);
    float t2 = u * u * u;
    float ym = (float)(u * (t2 + 2.0 * xm) / (2.0 * t2 + xm) * factor[2 + xe % 3]);
    /* ldexpf(ym, xe / 3): exact for the normal results of this domain */
    return ym * (float)ldexp(1.0, xe / 3);
}
#endif
