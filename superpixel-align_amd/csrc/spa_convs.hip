// The thin 3x3 convolutions at the top of DRN-C (models/drn.py:134-170: layer1 = BasicBlock(16 -> 16) at full resolution,
// layer2 = BasicBlock(16 -> 32, stride 2) with its 1x1 stride-2 projection; the reference's default backbone is DRN-C-26,
// batch_spalign_kmeans.py:524-526) on the 16-bit matrix cores at float32 accuracy — the layers that still ran on MIOpen with
// a separate epilogue pass (25 + 8 ms of the 101 ms forward of 30 full-size images).  16 or 32 input channels do not fill the
// 32-channel K steps / 64-channel tiles of k_conv3x3_f32, so these layers get the construction of DRN-D's layer 2
// (spa_stem.hip: k_drn_layer2_f16x3), generalised:
//
//   a workgroup stages the input pixels its TH x 32 output tile touches — (S TH + 2) x (32 S + 2) of them — in LDS as two
//   half-precision planes of the exactly scaled value (scale 2^(14 - e), e from the tracked maximum of the input), padded pixel
//   pitch for conflict-free 16-byte fragment reads; the weights' planes live in registers as MFMA A fragments; K = 9 taps x CIN
//   channels walks as 5 steps of (2 taps x 16 channels) or 9 steps of (1 tap x 32 channels); three v_mfma_f32_16x16x32_f16 per
//   product; bias, residual, ReLU and the largest stored magnitude in the epilogue; the next tile's patch is loaded while this
//   tile computes.  NPT > 0: the block's 1x1 stride-2 projection as extra output tiles that multiply the centre tap only and go
//   to a second tensor without ReLU — one pass over the input, two outputs.
//
// Arithmetic: the same two-plane products and float32 accumulation as every other split-plane kernel; tests/test_gpu_conv.py
// compares each instantiation with a float64 convolution.
#include "spa_common.h"

typedef _Float16 cs_h8 __attribute__((ext_vector_type(8)));
typedef float cs_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void cs_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned short cs_f16_bits(float f)
{
    const _Float16 h = (_Float16)f;
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float cs_f16_val(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }

// wp: main tiles [NCT][2 planes][STEPS][64 lanes] x 8 halfs, then projection tiles [NPT][2 planes][64 lanes] x 8 halfs (the A
// fragment of the step that holds the centre tap).  x (B,H,W,CIN) float32 channels-last; y (B,Ho,Wo,16 NCT); y2 (B,Ho,Wo,16 NPT)
template <int CIN, int NCT, int S, int NPT>
__global__ __launch_bounds__(256) void k_conv_small_f16x3(const float *__restrict__ x, int B, int H, int W, int Ho, int Wo,
                                                          const unsigned short *__restrict__ wp, const float *__restrict__ bias,
                                                          float inv_t, const unsigned *__restrict__ amax_in,
                                                          unsigned *__restrict__ amax_out, const float *__restrict__ res, int relu,
                                                          float *__restrict__ y, float *__restrict__ y2)
{
    constexpr int TH = S == 2 ? 4 : 8, TW = 32;
    constexpr int PH = S * TH + (S == 2 ? 1 : 2), PW = S * TW + (S == 2 ? 1 : 2);
    constexpr int PITCH = 2 * CIN + 8;                     // halfs per staged pixel: CIN h | CIN l | 8 pad
    constexpr int STEPS = CIN == 16 ? 5 : 9;
    constexpr int CSTEP = CIN == 16 ? 2 : 4;               // the step that holds the centre tap (tap 4)
    constexpr int CO = 16 * NCT, CP = 16 * NPT;
    __shared__ __attribute__((aligned(16))) unsigned short patch[PH * PW * PITCH + 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    const int n_tiles = tiles_x * tiles_y * B;
    cs_h8 wh[NCT][STEPS], wl[NCT][STEPS];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            wh[ct][s] = *(const cs_h8 *)(wp + ((size_t)((ct * 2 + 0) * STEPS + s) * 64 + lane) * 8);
            wl[ct][s] = *(const cs_h8 *)(wp + ((size_t)((ct * 2 + 1) * STEPS + s) * 64 + lane) * 8);
        }
    cs_h8 ph[NPT > 0 ? NPT : 1], pl[NPT > 0 ? NPT : 1];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        ph[pt] = *(const cs_h8 *)(wp + ((size_t)NCT * 2 * STEPS * 64 + (size_t)(pt * 2 + 0) * 64 + lane) * 8);
        pl[pt] = *(const cs_h8 *)(wp + ((size_t)NCT * 2 * STEPS * 64 + (size_t)(pt * 2 + 1) * 64 + lane) * 8);
    }
    float sc, unscale;
    {
        const unsigned bits = *amax_in;
        int e = (int)(bits >> 23) - 127;
        e = bits == 0u ? 0 : (e < -100 ? -100 : (e > 100 ? 100 : e));
        sc = __uint_as_float((unsigned)(127 + 14 - e) << 23);
        unscale = __uint_as_float((unsigned)(127 - 14 + e) << 23) * inv_t;
    }
    int toff[STEPS];                                       // patch offset (halfs) of this lane's k group in step s
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        int tap, ch;
        if (CIN == 16) { tap = 2 * s + (g >> 1); ch = 8 * (g & 1); if (tap > 8) tap = 8; }      // tap 9: zero weights, any valid address
        else { tap = s; ch = 8 * g; }
        toff[s] = ((tap / 3) * PW + tap % 3) * PITCH + ch;
    }
    constexpr int Q = CIN / 4;                             // float4 pieces of a pixel
    constexpr int NE = PH * PW * Q, NU = (NE + 255) / 256;
    float4 raw[NU];
    int epix[NU], eq[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        int e = tid + u * 256;
        if (e >= NE) e = -1;
        epix[u] = e < 0 ? -1 : e / Q;
        eq[u] = e < 0 ? 0 : e % Q;
    }
    auto patch_load = [&](int tile) {
        const int b = tile / (tiles_x * tiles_y), tr = tile - b * (tiles_x * tiles_y);
        const int ty0 = (tr / tiles_x) * TH, tx0 = (tr % tiles_x) * TW;
        const float *src = x + (long long)b * H * W * CIN;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int pix = epix[u] < 0 ? 0 : epix[u];
            const int iy = pix / PW, ix = pix - iy * PW;
            const int gy = S * ty0 - 1 + iy, gx = S * tx0 - 1 + ix;
            const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
            raw[u] = *(const float4 *)(src + ((long long)cy * W + cx) * CIN + 4 * eq[u]);
        }
    };
    unsigned amx = 0;
    if ((int)blockIdx.x < n_tiles) patch_load(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / (tiles_x * tiles_y), tr = tile - b * (tiles_x * tiles_y);
        const int ty0 = (tr / tiles_x) * TH, tx0 = (tr % tiles_x) * TW;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (epix[u] < 0) continue;
            const int iy = epix[u] / PW, ix = epix[u] - iy * PW;
            const int gy = S * ty0 - 1 + iy, gx = S * tx0 - 1 + ix;
            const bool in = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            const float v[4] = {in ? raw[u].x * sc : 0.f, in ? raw[u].y * sc : 0.f, in ? raw[u].z * sc : 0.f, in ? raw[u].w * sc : 0.f};
            unsigned short vh[4], vl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { vh[j] = cs_f16_bits(v[j]); vl[j] = cs_f16_bits(v[j] - cs_f16_val(vh[j])); }
            unsigned short *o = patch + epix[u] * PITCH + 4 * eq[u];
            *(uint2 *)o = make_uint2((unsigned)vh[0] | ((unsigned)vh[1] << 16), (unsigned)vh[2] | ((unsigned)vh[3] << 16));
            *(uint2 *)(o + CIN) = make_uint2((unsigned)vl[0] | ((unsigned)vl[1] << 16), (unsigned)vl[2] | ((unsigned)vl[3] << 16));
        }
        cs_lds_barrier();
        if (tile + (int)gridDim.x < n_tiles) patch_load(tile + (int)gridDim.x);     // next tile's input travels under the matrix work
        // TH * 2 pixel tiles of 16 (row t >> 1 of the output tile, columns (t & 1) * 16 ..)
        for (int t = wv; t < TH * 2; t += 4) {
            const int row = t >> 1, col = (t & 1) * 16 + m;
            const unsigned short *pb = patch + ((S * row) * PW + S * col) * PITCH;
            cs_f4 acc[NCT], accp[NPT > 0 ? NPT : 1];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = (cs_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) accp[pt] = (cs_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const cs_h8 fh = *(const cs_h8 *)(pb + toff[s]);
                const cs_h8 fl = *(const cs_h8 *)(pb + toff[s] + CIN);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ct][s], fh, acc[ct], 0, 0, 0);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ct][s], fl, acc[ct], 0, 0, 0);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ct][s], fh, acc[ct], 0, 0, 0);
                if (NPT > 0 && s == CSTEP) {
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) {
                        accp[pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl[pt], fh, accp[pt], 0, 0, 0);
                        accp[pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[pt], fl, accp[pt], 0, 0, 0);
                        accp[pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[pt], fh, accp[pt], 0, 0, 0);
                    }
                }
            }
            const int gy = ty0 + row, gx = tx0 + col;
            if (gy < Ho && gx < Wo) {
                const long long pix = ((long long)b * Ho + gy) * Wo + gx;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const float4 bv = *(const float4 *)(bias + 16 * ct + 4 * g);
                    float4 o = make_float4(acc[ct][0] * unscale + bv.x, acc[ct][1] * unscale + bv.y, acc[ct][2] * unscale + bv.z,
                                           acc[ct][3] * unscale + bv.w);
                    if (res) {
                        const float4 r = *(const float4 *)(res + pix * CO + 16 * ct + 4 * g);
                        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                    }
                    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                    *(float4 *)(y + pix * CO + 16 * ct + 4 * g) = o;
                    amx = max(amx, max(max(__float_as_uint(o.x) & 0x7fffffffu, __float_as_uint(o.y) & 0x7fffffffu),
                                       max(__float_as_uint(o.z) & 0x7fffffffu, __float_as_uint(o.w) & 0x7fffffffu)));
                }
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) {
                    const float4 bv = *(const float4 *)(bias + CO + 16 * pt + 4 * g);
                    *(float4 *)(y2 + pix * CP + 16 * pt + 4 * g) =
                        make_float4(accp[pt][0] * unscale + bv.x, accp[pt][1] * unscale + bv.y, accp[pt][2] * unscale + bv.z,
                                    accp[pt][3] * unscale + bv.w);
                }
            }
        }
        cs_lds_barrier();
    }
    if (amax_out) {
        for (int o = 32; o > 0; o >>= 1) amx = max(amx, (unsigned)__shfl_xor((int)amx, o));
        if (lane == 0 && amx > *(volatile unsigned *)amax_out) atomicMax(amax_out, amx);
    }
}

// x (B,H,W,Cin) float32 channels-last, Cin 16 or 32 -> y (B,Ho,Wo,Cout) = relu?(conv3x3(x; stride, padding 1) + bias [+ residual]),
// Cout 16 or 32; stride 1 or 2 (Ho = (H + stride - 1) / stride).  n_proj 0 or 32 (stride 2, Cin 16, Cout 32 only): the block's 1x1
// stride-2 projection (models/drn.py:195-203) as y2 (B,Ho,Wo,32) = its convolution + bias[Cout ..], no ReLU.  wp: the packed planes
// of t * w (Engine.small_planes: [Cout/16][2][steps][64][8] halfs, then [n_proj/16][2][64][8]); inv_t = 1 / t; bias Cout + n_proj
// floats; amax_in / amax_out as in spa_conv3x3_wino4_f16s (amax_out tracks y).
extern "C" int spa_conv_small_f16s(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin, const void *wp,
                                   float inv_t, int32_t Cout, int32_t stride, int32_t n_proj, const float *bias,
                                   const float *residual, int32_t relu, const void *amax_in, void *amax_out, float *y, float *y2,
                                   void *stream)
{
    SPA_ARG(ctx && x && wp && bias && amax_in && y && B > 0 && H > 0 && W > 0 && inv_t > 0.f);
    SPA_ARG((Cin == 16 || Cin == 32) && (Cout == 16 || Cout == 32) && (stride == 1 || stride == 2));
    SPA_ARG(n_proj == 0 || (n_proj == 32 && stride == 2 && Cin == 16 && Cout == 32 && y2 && !residual));
    SPA_ARG((((uintptr_t)x | (uintptr_t)wp | (uintptr_t)bias | (uintptr_t)y | (uintptr_t)y2 | (uintptr_t)residual) & 15) == 0);
    hipStream_t s = spa_stream(stream);
    if (amax_out) spa_zero_word(amax_out, s);
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    const int TH = stride == 2 ? 4 : 8;
    const long long n_tiles = (long long)((Wo + 31) / 32) * ((Ho + TH - 1) / TH) * B;
    SPA_ARG(n_tiles < (1ll << 31) && (long long)B * H * W * Cin < (1ll << 40));
    SpaProfScope prof_(ctx, PROF_DRN_CONV16_FRONT, s);
    long long grid = 2ll * ctx->n_cu;
    if (grid > n_tiles) grid = n_tiles;
#define CS_LAUNCH(CI, NC, ST, NP)                                                                                              \
    hipLaunchKernelGGL((k_conv_small_f16x3<CI, NC, ST, NP>), dim3((unsigned)grid), dim3(256), 0, s, x, B, H, W, Ho, Wo,       \
                       (const unsigned short *)wp, bias, inv_t, (const unsigned *)amax_in, (unsigned *)amax_out, residual, relu, y, y2)
    if (Cin == 16 && Cout == 16 && stride == 1) CS_LAUNCH(16, 1, 1, 0);
    else if (Cin == 16 && Cout == 32 && stride == 2 && n_proj == 32) CS_LAUNCH(16, 2, 2, 2);
    else if (Cin == 16 && Cout == 32 && stride == 2) CS_LAUNCH(16, 2, 2, 0);
    else if (Cin == 32 && Cout == 32 && stride == 1) CS_LAUNCH(32, 2, 1, 0);
    else if (Cin == 16 && Cout == 32 && stride == 1) CS_LAUNCH(16, 2, 1, 0);
    else {
        spa_set_error("spa_conv_small_f16s: no instantiation for Cin %d Cout %d stride %d", Cin, Cout, stride);
        return SPA_ERR_ARG;
    }
#undef CS_LAUNCH
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
