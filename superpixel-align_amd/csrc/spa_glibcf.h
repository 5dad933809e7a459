// powf(x, 2.4f) and cbrtf(x) as GNU libc 2.35 evaluates them, on the device.
//
// scikit-image keeps a float32 image in float32, so rgb2lab's np.power(arr, 2.4) and np.cbrt(arr)
// (skimage/color/colorconv.py:659, :959) are numpy's float32 loops, which end in the C library's powf / cbrtf on
// every host whose numpy does not dispatch to its AVX-512 SVML kernels (the reference configuration of
// tests/golden/PROVENANCE.txt).  Both are short binary64 computations with one final rounding:
//   powf   sysdeps/ieee754/flt-32/e_powf.c (+ e_powf_log2_data.c, e_exp2f_data.c; the -mfma build x86-64 selects
//          on CPUs with FMA3, hence the explicit fma()s — the file is compiled with -ffp-contract=off, nothing
//          else fuses): 16-entry (1/c, log2 c) table + degree-5 polynomial, y*log2(x), 32-entry 2^(i/32) table +
//          cubic: 11 fused operations, 5 multiplications, 3 additions;
//   cbrtf  sysdeps/ieee754/flt-32/s_cbrtf.c: frexpf, quadratic start value, one Halley step in binary64 (one
//          division), a factor 2^(k/3), ldexpf.
// Domain: finite positive normal x (callers: x > 0.0905 resp. x > 0.008856); +inf passes through.
// The tables (768 bytes) are read by lane-dependent indices: kernels on the hot path stage them in LDS
// (spa_glibcf_stage) and pass the LDS pointer; one-off kernels read the global copy.
#pragma once
#include <stdint.h>

struct spa_glibcf_tables {
    double log2_tab[16][2];        // {invc, logc}
    unsigned long long exp2_tab[32];
    double cbrt_factor[8];         // [2 + xe % 3]: 2^(-2/3), 2^(-1/3), 1, 2^(1/3), 2^(2/3)
};

__device__ static const spa_glibcf_tables spa_glibcf_global = {
    {{0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
     {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
     {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
     {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
     {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
     {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
     {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
     {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}},
    {0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
     0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
     0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
     0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
     0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
     0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
     0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
     0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL},
    {1.0 / 1.5874010519681994748, 1.0 / 1.2599210498948731648, 1.0, 1.2599210498948731648,
     1.5874010519681994748, 0.0, 0.0, 0.0},
};

// copy the tables into the workgroup's LDS block (call from every thread, then __syncthreads())
__device__ __forceinline__ void spa_glibcf_stage(spa_glibcf_tables *lds)
{
    const unsigned long long *src = (const unsigned long long *)&spa_glibcf_global;
    unsigned long long *dst = (unsigned long long *)lds;
    for (unsigned i = threadIdx.x; i < sizeof(spa_glibcf_tables) / 8; i += blockDim.x) dst[i] = src[i];
}

// e_powf.c:__powf(x, 2.4f): log2_inline, y * logx, exp2_inline
__device__ __forceinline__ float spa_glibc_powf_2p4(float x, const spa_glibcf_tables *__restrict__ T)
{
    const unsigned ix = __float_as_uint(x);
    if (ix == 0x7f800000u) return x;
    const unsigned tmp = ix - 0x3f330000u;
    const unsigned i = (tmp >> 19) & 15u;
    const unsigned top = tmp & 0xff800000u;
    const int k = (int)top >> 23;
    const double z = (double)__uint_as_float(ix - top);
    const double invc = T->log2_tab[i][0], logc = T->log2_tab[i][1];
    const double r = fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    const double q0 = fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    const double p = fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
    const double r4 = r2 * r2;
    double q = fma(0x1.71547652ab82bp0, r, y0);
    q = fma(p, r2, q);
    const double logx = fma(q0, r4, q);
    const double ylogx = (double)2.4f * logx;
    if (ylogx > 0x1.fffffffd1d571p+6) return __uint_as_float(0x7f800000u);      // __math_oflowf
    const double SHIFT = 0x1.8p+52 / 32;
    double kd = ylogx + SHIFT;
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= SHIFT;
    const double rr = ylogx - kd;
    const unsigned long long t = T->exp2_tab[ki & 31u] + (ki << 47);
    const double s = __longlong_as_double((long long)t);
    const double zz = fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
    const double rr2 = rr * rr;
    double w = fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
    w = fma(zz, rr2, w);
    return (float)(w * s);
}

// s_cbrtf.c:__cbrtf
__device__ __forceinline__ float spa_glibc_cbrtf(float x, const spa_glibcf_tables *__restrict__ T)
{
    const unsigned ix = __float_as_uint(x);
    if (ix == 0x7f800000u) return x;
    const int xe = (int)(ix >> 23) - 126;                                   // frexpf
    const float xm = __uint_as_float((ix & 0x007fffffu) | 0x3f000000u);
    const float u = (float)(0.492659620528969547 + (0.697570460207922770 - 0.191502161678719066 * (double)xm) * (double)xm);
    const float t2 = u * u * u;
    const float ym = (float)((double)u * ((double)t2 + 2.0 * (double)xm) / (2.0 * (double)t2 + (double)xm)
                             * T->cbrt_factor[2 + xe % 3]);
    return ym * __uint_as_float((unsigned)(127 + xe / 3) << 23);             // ldexpf, exact here
}
