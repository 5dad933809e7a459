// bfloat16 convolutions of the LIGHT layers of the bf16 DRN (models/drn.py:134-151, 195-203: layer 2 of arch D, the
// stride-2 openers of layers 3 / 4 and their 1x1 stride-2 projections, the 1x1 projections of layers 5 / 6, the 16- and
// 32-channel BasicBlocks of arch C) — everything spa_conv3x3_bf16 (stride 1, Cin % 64 == 0) does not take, so that the
// bf16 network runs on libspalign's kernels end to end (BASELINE configs[4]).
//
//   Y[b, y, x, n] = relu?( bias[n] + res[b, y, x, n] + sum_{tap, c} X[b, S y + dy(tap) d, S x + dx(tap) d, c] * Wt[n, tap, c] )
//
// These layers are a few per cent of the network's FLOPs and are bound by their activations' bytes, so the kernel is the
// plain form: no pixel staging at all.
//   * a workgroup (4 waves) owns one block of 16 MI output channels; its weights — all taps, all input channels — go to
//     LDS ONCE, already in the lane order of the matrix instruction's A operand (a fragment read is 1 KB lane-linear:
//     conflict free), and the workgroup then walks pixel strips persistently;
//   * a wave owns a strip of 64 consecutive output pixels of one output row (4 tiles of 16): the B operand of
//     v_mfma_f32_16x16x32_bf16 is 8 consecutive input channels of one pixel per lane, i.e. one 16-byte global load per lane
//     and tile straight from the channels-last image (the four k-quarters of a pixel are 64 contiguous bytes; the taps'
//     re-reads of a pixel hit in L1 / L2); 16-channel inputs use v_mfma_f32_16x16x16_bf16 (8 bytes per lane);
//   * float32 accumulation, bias / residual / ReLU and ONE rounding to bf16 in the epilogue (8-byte stores: a lane holds
//     four consecutive output channels of one pixel).
#include "spa_common.h"
#include <stdlib.h>

typedef __bf16 cl_bf16x8 __attribute__((ext_vector_type(8)));
typedef short cl_s16x4 __attribute__((ext_vector_type(4)));
typedef float cl_f32x4 __attribute__((ext_vector_type(4)));

#define CL_PJ 4               // 16-pixel tiles per wave strip
#define CL_THREADS 256

__device__ __forceinline__ unsigned cl_bf16_bits(float f)
{
    __bf16 h = (__bf16)f;                          // round to nearest even, NaN stays NaN
    return (unsigned)__builtin_bit_cast(unsigned short, h);
}

template <int CIN, int TAPS, int S, int MI, int HAS_RES>
__global__ __launch_bounds__(CL_THREADS) void k_conv_bf16_light(const __bf16 *__restrict__ X, const __bf16 *__restrict__ Wt,
                                                                const float *__restrict__ bias, const __bf16 *__restrict__ R,
                                                                __bf16 *__restrict__ Y, int B, int Hi, int Wi, int Ho, int Wo,
                                                                int Cout, int dil, int relu, int xstrips, long long nstrips)
{
    constexpr int KSTEP = CIN >= 32 ? 32 : 16;        // k per matrix instruction
    constexpr int KS = CIN / KSTEP;                    // instructions per tap
    constexpr int NSTEP = TAPS * KS;
    constexpr int FB = KSTEP == 32 ? 16 : 8;           // bytes per lane and fragment
    extern __shared__ __attribute__((aligned(16))) char cl_lds[];      // [NSTEP][MI][64 lanes] x FB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * (16 * MI);
    const int frow = lane & 15, fk = lane >> 4;

    // the block's weights, in fragment order
    for (int e = tid; e < NSTEP * MI * 64; e += CL_THREADS) {
        const int l = e & 63, i = (e >> 6) % MI, st = (e >> 6) / MI;
        const int tap = st / KS, kc = st - tap * KS;
        const char *src = (const char *)(Wt + ((long long)(n0 + i * 16 + (l & 15)) * TAPS + tap) * CIN + kc * KSTEP + (l >> 4) * (FB / 2));
        if (FB == 16) *(uint4 *)(cl_lds + (long long)e * 16) = *(const uint4 *)src;
        else *(uint2 *)(cl_lds + (long long)e * 8) = *(const uint2 *)src;
    }
    __syncthreads();

    for (long long strip = (long long)blockIdx.x * 4 + wave; strip < nstrips; strip += (long long)gridDim.x * 4) {
        const int xs = (int)(strip % xstrips);
        const long long row_id = strip / xstrips;                  // b * Ho + y
        const int y = (int)(row_id % Ho), b = (int)(row_id / Ho);
        const int x0 = xs * (16 * CL_PJ);
        cl_f32x4 acc[MI][CL_PJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < CL_PJ; ++j) acc[i][j] = (cl_f32x4){0.f, 0.f, 0.f, 0.f};
        const __bf16 *xb = X + (long long)b * Hi * Wi * CIN + fk * (FB / 2);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dy = TAPS == 9 ? tap / 3 - 1 : 0, dx = TAPS == 9 ? tap % 3 - 1 : 0;
            const int yi = y * S + dy * dil;
            const bool yok = yi >= 0 && yi < Hi;
#pragma unroll
            for (int kc = 0; kc < KS; ++kc) {
                const int st = tap * KS + kc;
                // (the loop is fully unrolled and the compiler hoists every load it can: a fence every fourth step keeps at most
                // 16 fragment loads in flight per lane instead of 72 — 256 input channels x 128 output channels spilled without it)
                if ((st & 3) == 0 && st) asm volatile("" ::: "memory");
                if (FB == 16) {
                    cl_bf16x8 pf[CL_PJ], wf[MI];
#pragma unroll
                    for (int j = 0; j < CL_PJ; ++j) {
                        const int xi = (x0 + j * 16 + frow) * S + dx * dil;
                        const bool ok = yok && xi >= 0 && xi < Wi;
                        const uint4 v = ok ? *(const uint4 *)(xb + ((long long)yi * Wi + xi) * CIN + kc * KSTEP) : make_uint4(0u, 0u, 0u, 0u);
                        pf[j] = __builtin_bit_cast(cl_bf16x8, v);
                    }
#pragma unroll
                    for (int i = 0; i < MI; ++i) wf[i] = *(const cl_bf16x8 *)(cl_lds + ((st * MI + i) * 64 + lane) * 16);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < CL_PJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], pf[j], acc[i][j], 0, 0, 0);
                } else {
                    cl_s16x4 pf[CL_PJ], wf[MI];
#pragma unroll
                    for (int j = 0; j < CL_PJ; ++j) {
                        const int xi = (x0 + j * 16 + frow) * S + dx * dil;
                        const bool ok = yok && xi >= 0 && xi < Wi;
                        const uint2 v = ok ? *(const uint2 *)(xb + ((long long)yi * Wi + xi) * CIN + kc * KSTEP) : make_uint2(0u, 0u);
                        pf[j] = __builtin_bit_cast(cl_s16x4, v);
                    }
#pragma unroll
                    for (int i = 0; i < MI; ++i) wf[i] = *(const cl_s16x4 *)(cl_lds + ((st * MI + i) * 64 + lane) * 8);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < CL_PJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wf[i], pf[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        // epilogue: the lane holds channels c .. c + 3 (c = n0 + 16 i + 4 (lane >> 4)) of pixel (lane & 15) of tile j
#pragma unroll
        for (int j = 0; j < CL_PJ; ++j) {
            const int xx = x0 + j * 16 + frow;
            if (xx >= Wo) continue;
            const long long pix = row_id * Wo + xx;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int c = n0 + i * 16 + fk * 4;
                const float4 bv = *(const float4 *)(bias + c);
                float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
                if (HAS_RES) {
                    const uint2 rr = *(const uint2 *)(R + pix * Cout + c);
                    v0 += __uint_as_float(rr.x << 16); v1 += __uint_as_float(rr.x & 0xffff0000u);
                    v2 += __uint_as_float(rr.y << 16); v3 += __uint_as_float(rr.y & 0xffff0000u);
                }
                if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                uint2 o;
                o.x = cl_bf16_bits(v0) | (cl_bf16_bits(v1) << 16);
                o.y = cl_bf16_bits(v2) | (cl_bf16_bits(v3) << 16);
                *(uint2 *)(Y + pix * Cout + c) = o;
            }
        }
    }
}

template <int CIN, int TAPS, int S, int MI>
static int conv_bf16_light_launch(spa_ctx *ctx, const void *x, int B, int Hi, int Wi, int Ho, int Wo, const void *wt, int Cout,
                                  const float *bias, const void *residual, int relu, int dil, void *y, hipStream_t s)
{
    constexpr int KSTEP = CIN >= 32 ? 32 : 16;
    const size_t lds = (size_t)TAPS * (CIN / KSTEP) * MI * 64 * (KSTEP == 32 ? 16 : 8);
    SPA_ARG(lds <= 152 * 1024);
    const int xstrips = (Wo + 16 * CL_PJ - 1) / (16 * CL_PJ);
    const long long nstrips = (long long)B * Ho * xstrips;
    // persistent workgroups: as many as stay resident (LDS: the weight block; 8 per CU at most), 4 strips per pass each
    const int per_cu = lds ? (int)((160 * 1024) / (lds > 20 * 1024 ? lds : 20 * 1024)) : 8;
    long long gx = (long long)ctx->n_cu * (per_cu < 1 ? 1 : per_cu);
    if (gx > (nstrips + 3) / 4) gx = (nstrips + 3) / 4;
    const int nblk = Cout / (16 * MI);
    if (nblk > 1) gx = (gx + nblk - 1) / nblk > 0 ? (gx + nblk - 1) / nblk : 1;
    const void *k0 = (const void *)k_conv_bf16_light<CIN, TAPS, S, MI, 0>, *k1 = (const void *)k_conv_bf16_light<CIN, TAPS, S, MI, 1>;
    if (lds > 48 * 1024) {        // (cheap and idempotent: no per-instantiation flag to keep)
        SPA_HIP(hipFuncSetAttribute(k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SPA_HIP(hipFuncSetAttribute(k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    SpaProfScope prof_(ctx, PROF_DRN_CONV_LIGHT, s);
    if (residual)
        hipLaunchKernelGGL((k_conv_bf16_light<CIN, TAPS, S, MI, 1>), dim3((unsigned)gx, nblk), dim3(CL_THREADS), lds, s, (const __bf16 *)x,
                           (const __bf16 *)wt, bias, (const __bf16 *)residual, (__bf16 *)y, B, Hi, Wi, Ho, Wo, Cout, dil, relu, xstrips, nstrips);
    else
        hipLaunchKernelGGL((k_conv_bf16_light<CIN, TAPS, S, MI, 0>), dim3((unsigned)gx, nblk), dim3(CL_THREADS), lds, s, (const __bf16 *)x,
                           (const __bf16 *)wt, bias, (const __bf16 *)residual, (__bf16 *)y, B, Hi, Wi, Ho, Wo, Cout, dil, relu, xstrips, nstrips);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// x (B,Hi,Wi,Cin) bf16 channels-last; wt (Cout,taps,Cin) bf16 (tap = ky*3 + kx; taps = 9: 3x3, padding = dilation; taps = 1:
// 1x1, no padding); stride 1 or 2 (output (Hi + stride - 1) / stride x (Wi + stride - 1) / stride); bias (Cout) float32;
// residual (B,Ho,Wo,Cout) bf16 or NULL; y (B,Ho,Wo,Cout) bf16.  Cin in {16, 32, 64} (3x3 or 1x1) or {128, 256} (1x1 only: the
// stride-1 3x3 layers from 64 channels up are spa_conv3x3_bf16's); the block of output channels a workgroup keeps in LDS is
// 64 where Cout % 64 == 0 and Cin >= 32, else 32 (Cout % 32 == 0, Cin 16 / 32) or 16 (Cin 16).
extern "C" int spa_conv_bf16_light(spa_ctx *ctx, const void *x, int32_t B, int32_t Hi, int32_t Wi, int32_t Cin, const void *wt,
                                   int32_t taps, int32_t stride, int32_t Cout, const float *bias, const void *residual,
                                   int32_t relu, int32_t dilation, void *y, void *stream)
{
    SPA_ARG(ctx && x && wt && bias && y && B > 0 && Hi > 0 && Wi > 0 && dilation >= 1);
    SPA_ARG((taps == 1 || taps == 9) && (stride == 1 || stride == 2) && Cout > 0 && Cout % 16 == 0);
    SPA_ARG((((uintptr_t)x | (uintptr_t)wt | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)residual) % 16) == 0);
    hipStream_t s = spa_stream(stream);
    const int Ho = (Hi + stride - 1) / stride, Wo = (Wi + stride - 1) / stride;
    SPA_ARG((long long)B * Ho * ((Wo + 63) / 64) < (1ll << 40));
    // (the 1x1 projections of layers 5 / 6 keep 128 output channels per workgroup: every block of output channels re-reads the
    // input, 256 -> 512 on 30 x 128 x 256 pixels 1.13 ms with blocks of 64)
    const int mi = (taps == 1 && Cin >= 128 && Cout % 128 == 0) ? 8
                   : ((Cout % 64 == 0 && Cin >= 32) ? 4 : ((Cout % 32 == 0 && Cin <= 32) ? 2 : ((Cin == 16 && Cout % 16 == 0) ? 1 : 0)));
#define CL_CASE(C, T, S_, M)                                                                                                       \
    if (Cin == C && taps == T && stride == S_ && mi == M)                                                                          \
        return conv_bf16_light_launch<C, T, S_, M>(ctx, x, B, Hi, Wi, Ho, Wo, wt, Cout, bias, residual, relu, dilation, y, s);
#define CL_CASES(C, M) CL_CASE(C, 9, 1, M) CL_CASE(C, 9, 2, M) CL_CASE(C, 1, 1, M) CL_CASE(C, 1, 2, M)
    CL_CASES(32, 4) CL_CASES(64, 4) CL_CASE(128, 1, 1, 4) CL_CASE(128, 1, 2, 4) CL_CASE(256, 1, 1, 4) CL_CASE(256, 1, 2, 4)
    CL_CASE(128, 1, 1, 8) CL_CASE(128, 1, 2, 8) CL_CASE(256, 1, 1, 8) CL_CASE(256, 1, 2, 8)
    CL_CASES(16, 2) CL_CASES(32, 2)
    CL_CASES(16, 1)
#undef CL_CASES
#undef CL_CASE
    SPA_ARG(!"spa_conv_bf16_light: Cin must be 16, 32, 64 (3x3 or 1x1) or 128, 256 (1x1 only) and Cout a multiple of 64 (Cin >= 32), 32 (Cin 16 / 32) or 16 (Cin 16)");
    return SPA_OK;
}
