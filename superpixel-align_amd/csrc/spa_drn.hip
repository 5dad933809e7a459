// Memory-bound glue of the DRN forward, fused.  The convolutions themselves stay in
// PyTorch-ROCm/MIOpen (MFMA implicit GEMM); what PyTorch eager adds around them is a chain of
// separate elementwise kernels — 7 for the input normalisation, and per convolution a bias add
// (BatchNorm folded into the convolution leaves a bias), an optional residual add and a ReLU,
// each a full read+write pass over activations of up to 134 MB per image.  These two kernels do
// each chain in one pass, with the same arithmetic and rounding as the eager sequence.
//   spa_drn_normalise : DRN.batch_predict (models/drn.py:319-321): x/255 (float32), then
//                       (x - mean) and (x / std) in float64 rounded to float32; planar NCHW float32
//                       in, channels-last (NHWC) float32 or bfloat16 out
//   spa_bias_act      : y = relu?(y + bias [+ residual]) in place on a channels-last tensor
#include "spa_common.h"

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f)
{
    // round to nearest even, NaN preserved (same as a plain cast to __hip_bfloat16)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x0040u);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

__global__ __launch_bounds__(256) void k_drn_normalise(const float *__restrict__ x, void *__restrict__ out,
                                                       long long npix, int out_bf16, double m0, double m1,
                                                       double m2, double s0, double s1, double s2)
{
    const int b = blockIdx.y;
    const float *src = x + (long long)b * 3 * npix;
    const double mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t = src[c * npix + p] / 255.0f;
            t = (float)((double)t - mean[c]);
            t = (float)((double)t / sd[c]);
            v[c] = t;
        }
        const long long o = ((long long)b * npix + p) * 3;
        if (out_bf16) {
            unsigned short *d = (unsigned short *)out + o;
            d[0] = f32_to_bf16(v[0]); d[1] = f32_to_bf16(v[1]); d[2] = f32_to_bf16(v[2]);
        } else {
            float *d = (float *)out + o;
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2];
        }
    }
}

extern "C" int spa_drn_normalise(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, void *out,
                                 int32_t out_dtype, const double *mean3_host, const double *std3_host,
                                 void *stream)
{
    SPA_ARG(ctx && x && out && mean3_host && std3_host && B > 0 && H > 0 && W > 0);
    SPA_ARG(out_dtype == 0 || out_dtype == 1);
    const long long npix = (long long)H * W;
    int gx = (int)((npix + 255) / 256);
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(k_drn_normalise, dim3(gx, B), dim3(256), 0, spa_stream(stream), x, out, npix, out_dtype,
                       mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// float32: 4 channels (16 bytes) per thread and step, four steps in flight.  The bias index is a 32-bit mask (channel
// counts are powers of two in the DRN) or a 32-bit remainder: the 64-bit `i % c4` of the first version cost ~100
// vector instructions per 16 bytes and made this "streaming" kernel instruction bound (0.60 of the HBM roof).
template <bool POW2>
__global__ __launch_bounds__(256) void k_bias_act_f32(float *__restrict__ y, const float *__restrict__ bias,
                                                      const float *__restrict__ res, unsigned n4, unsigned c4,
                                                      int relu, unsigned *__restrict__ amax = nullptr)
{
    const unsigned stride = gridDim.x * 256u;
    unsigned mx = 0u;         // amax: largest magnitude stored (the scale input of the split-plane convolutions)
    const float4 *b4 = (const float4 *)bias;
    auto bidx = [&](unsigned i) -> unsigned { return POW2 ? (i & (c4 - 1u)) : (i % c4); };
    auto apply = [&](float4 v, const float4 bb, const float4 r) -> float4 {
        v.x = v.x + bb.x; v.y = v.y + bb.y; v.z = v.z + bb.z; v.w = v.w + bb.w;
        if (res) { v.x = v.x + r.x; v.y = v.y + r.y; v.z = v.z + r.z; v.w = v.w + r.w; }
        if (relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
        if (amax)
            mx = max(max(mx, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu,
                     max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)));
        return v;
    };
    unsigned i = blockIdx.x * 256u + threadIdx.x;
    const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (; (unsigned long long)i + 3ull * stride < n4; i += 4u * stride) {
        float4 v[4], r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ((const float4 *)y)[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = res ? ((const float4 *)res)[i + u * stride] : zero;
#pragma unroll
        for (int u = 0; u < 4; ++u) ((float4 *)y)[i + u * stride] = apply(v[u], b4[bidx(i + u * stride)], r[u]);
    }
    for (; i < n4; i += stride) {
        const float4 r = res ? ((const float4 *)res)[i] : zero;
        ((float4 *)y)[i] = apply(((const float4 *)y)[i], b4[bidx(i)], r);
    }
    if (amax) {
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
        if ((threadIdx.x & 63) == 0 && mx > *(volatile unsigned *)amax) atomicMax(amax, mx);
    }
}

// bfloat16: 8 channels (16 bytes) per thread; every eager op rounds to bf16, and so does this
__global__ __launch_bounds__(256) void k_bias_act_bf16(unsigned short *__restrict__ y,
                                                       const unsigned short *__restrict__ bias,
                                                       const unsigned short *__restrict__ res, long long n8,
                                                       int c8, int relu)
{
    const bool pow2 = (c8 & (c8 - 1)) == 0;          // wave-uniform: no 64-bit remainder per vector
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        uint4 raw = ((const uint4 *)y)[i];
        const uint4 braw = ((const uint4 *)bias)[pow2 ? ((unsigned)i & (unsigned)(c8 - 1)) : (unsigned)(i % c8)];
        uint4 rraw = make_uint4(0, 0, 0, 0);
        if (res) rraw = ((const uint4 *)res)[i];
        unsigned w[4] = {raw.x, raw.y, raw.z, raw.w}, bw[4] = {braw.x, braw.y, braw.z, braw.w};
        unsigned rw[4] = {rraw.x, rraw.y, rraw.z, rraw.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned short h[2] = {(unsigned short)(w[k] & 0xffffu), (unsigned short)(w[k] >> 16)};
            unsigned short hb[2] = {(unsigned short)(bw[k] & 0xffffu), (unsigned short)(bw[k] >> 16)};
            unsigned short hr[2] = {(unsigned short)(rw[k] & 0xffffu), (unsigned short)(rw[k] >> 16)};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v = bf16_to_f32(f32_to_bf16(bf16_to_f32(h[j]) + bf16_to_f32(hb[j])));
                if (res) v = bf16_to_f32(f32_to_bf16(v + bf16_to_f32(hr[j])));
                if (relu) v = fmaxf(v, 0.0f);
                h[j] = f32_to_bf16(v);
            }
            w[k] = (unsigned)h[0] | ((unsigned)h[1] << 16);
        }
        ((uint4 *)y)[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

static int bias_act_impl(spa_ctx *ctx, void *y, int32_t dtype, int64_t rows, int32_t C, const void *bias,
                         const void *residual, int32_t relu, void *amax, void *stream);

extern "C" int spa_bias_act(spa_ctx *ctx, void *y, int32_t dtype, int64_t rows, int32_t C, const void *bias,
                            const void *residual, int32_t relu, void *stream)
{
    return bias_act_impl(ctx, y, dtype, rows, C, bias, residual, relu, nullptr, stream);
}

// float32 only: the same pass, and amax[0] = bit pattern of the largest magnitude it stored (spa_amax_f32's result for the
// output, without the extra pass): the scale input of the split-plane convolution that reads y next
extern "C" int spa_bias_act_amax(spa_ctx *ctx, float *y, int64_t rows, int32_t C, const float *bias,
                                 const float *residual, int32_t relu, void *amax, void *stream)
{
    SPA_ARG(amax);
    spa_zero_word(amax, spa_stream(stream));
    return bias_act_impl(ctx, y, 0, rows, C, bias, residual, relu, amax, stream);
}

static int bias_act_impl(spa_ctx *ctx, void *y, int32_t dtype, int64_t rows, int32_t C, const void *bias,
                         const void *residual, int32_t relu, void *amax, void *stream)
{
    SPA_ARG(ctx && y && bias && rows > 0 && C > 0);
    SPA_ARG(dtype == 0 || dtype == 1);
    const int vec = dtype == 0 ? 4 : 8;
    SPA_ARG(C % vec == 0);
    SPA_ARG((((uintptr_t)y | (uintptr_t)bias | (uintptr_t)residual) & 15) == 0);
    SpaProfScope prof_(ctx, PROF_DRN_BIAS_ACT, spa_stream(stream));
    const long long n = rows * C / vec;
    long long gx = (n + 255) / 256;
    if (gx > 4096) gx = 4096;
    if (dtype == 0) {
        SPA_ARG(n < (1ll << 32) - 4ll * 4096 * 256);           // 32-bit vector indices (64 GB of float32)
        const unsigned c4 = (unsigned)(C / vec);
        long long g4 = (n + 1023) / 1024;                      // four vectors per thread and step
        if (g4 > 8192) g4 = 8192;
        if (g4 < 1) g4 = 1;
        if ((c4 & (c4 - 1u)) == 0u)
            hipLaunchKernelGGL(k_bias_act_f32<true>, dim3((unsigned)g4), dim3(256), 0, spa_stream(stream), (float *)y,
                               (const float *)bias, (const float *)residual, (unsigned)n, c4, relu, (unsigned *)amax);
        else
            hipLaunchKernelGGL(k_bias_act_f32<false>, dim3((unsigned)g4), dim3(256), 0, spa_stream(stream), (float *)y,
                               (const float *)bias, (const float *)residual, (unsigned)n, c4, relu, (unsigned *)amax);
    } else
        hipLaunchKernelGGL(k_bias_act_bf16, dim3((unsigned)gx), dim3(256), 0, spa_stream(stream),
                           (unsigned short *)y, (const unsigned short *)bias,
                           (const unsigned short *)residual, n, C / vec, relu);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
