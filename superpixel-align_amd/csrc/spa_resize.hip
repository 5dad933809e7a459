// Input stage on the GPU: bicubic resize of decoded 8-bit images to the network's input
// size, i.e. datasets/resize_image_dataset.py:31-34 (chainercv.transforms.resize(image, shape, 3)) as
// Pillow computes it on an 8-bit image: per channel Image.resize((w, h), BICUBIC).
//
// Pillow's 8-bit resampling is integer arithmetic, so the kernels are bit exact with it: support-scaled
// Keys bicubic (a = -0.5) weights per output position, normalised, converted to 22-bit fixed point
// ((int)(+-0.5 + w * 2^22)); a horizontal pass over every input row, then a vertical pass, each
// clip8((2^21 + sum pixel * k) >> 22).  The weight tables are built on the host in float64 with the same
// operation order as Pillow's precompute_coeffs (contraction off) and cached per (input, output) size.
// Decoded images arrive interleaved (B, H, W, C) uint8 — what a PNG decoder produces — and leave as the
// planar float32 batch (B, C, h, w) the rest of the path takes, so the full-size image crosses PCIe as
// 3 bytes per pixel and is never resized on the host.
#include <math.h>
#include <stdlib.h>

#include "spa_common.h"

static double rs_bicubic(double x)
{
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// bounds (out, 2) {first tap, tap count}, kk (out, ksize) fixed-point weights; returns ksize
static int rs_coeffs(int in_size, int out_size, int32_t *bounds, int32_t **kk_out)
{
    const double scale = (double)in_size / (double)out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    int32_t *kk = (int32_t *)malloc((size_t)out_size * ksize * sizeof(int32_t));
    double *k = (double *)malloc((size_t)ksize * sizeof(double));
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        const double ss = 1.0 / filterscale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; ++x) {
            const double w = rs_bicubic((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; ++x) k[x] = 0;
        for (x = 0; x < ksize; ++x)
            kk[xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << 22)) : (int)(0.5 + k[x] * (1 << 22));
        bounds[xx * 2 + 0] = xmin;
        bounds[xx * 2 + 1] = xmax;
    }
    free(k);
    *kk_out = kk;
    return ksize;
}

__device__ __forceinline__ int rs_clip8(int v)
{
    v >>= 22;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: src (B, H, W, C) -> tmp (B, H, w, C), one thread per output sample
__global__ __launch_bounds__(256) void k_resize_h(const uint8_t *__restrict__ src, int H, int W, int C, int w,
                                                  const int32_t *__restrict__ bounds, const int32_t *__restrict__ kk,
                                                  int ksize, uint8_t *__restrict__ tmp)
{
    const int b = blockIdx.z, yy = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;            // xx * C + c
    if (i >= w * C) return;
    const int xx = i / C, c = i - xx * C;
    const int xmin = bounds[xx * 2], xmax = bounds[xx * 2 + 1];
    const uint8_t *row = src + ((long long)b * H + yy) * W * C + c;
    const int32_t *k = kk + xx * ksize;
    int ss = 1 << 21;
    for (int x = 0; x < xmax; ++x) ss += (int)row[(long long)(x + xmin) * C] * k[x];
    tmp[((long long)b * H + yy) * w * C + i] = (uint8_t)rs_clip8(ss);
}

// vertical pass: tmp (B, H, w, C) -> out (B, C, h, w) float32 (the 8-bit value Pillow returns, as a float)
__global__ __launch_bounds__(256) void k_resize_v(const uint8_t *__restrict__ tmp, int H, int w, int C, int h,
                                                  const int32_t *__restrict__ bounds, const int32_t *__restrict__ kk,
                                                  int ksize, float *__restrict__ out)
{
    const int b = blockIdx.z, yy = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;            // xx * C + c
    if (i >= w * C) return;
    const int xx = i / C, c = i - xx * C;
    const int ymin = bounds[yy * 2], ymax = bounds[yy * 2 + 1];
    const uint8_t *col = tmp + (long long)b * H * w * C + i;
    const int32_t *k = kk + yy * ksize;
    int ss = 1 << 21;
    for (int y = 0; y < ymax; ++y) ss += (int)col[(long long)(y + ymin) * w * C] * k[y];
    out[(((long long)b * C + c) * h + yy) * w + xx] = (float)rs_clip8(ss);
}

// interleaved uint8 -> planar float32 when no axis changes size.  C == 3 (the drivers' every batch: this is the widening of
// the uploaded 8-bit images in the host-to-host loop): a thread owns 4 pixels = 12 consecutive bytes (three aligned dword
// loads) and stores one float4 per plane — 1.26 ms per 30 full-size images with byte loads, bound by their issue
__global__ __launch_bounds__(256) void k_u8_to_planar(const uint8_t *__restrict__ src, long long npix, int C,
                                                      float *__restrict__ out)
{
    const int b = blockIdx.y;
    if (C == 3 && (npix & 3) == 0 && ((uintptr_t)src & 3) == 0 && ((uintptr_t)out & 15) == 0) {
        const uint32_t *s32 = (const uint32_t *)(src + (long long)b * npix * 3);        // npix * 3 bytes per image: a multiple of 12
        float *o = out + (long long)b * 3 * npix;
        for (long long q = blockIdx.x * 256ll + threadIdx.x; q < (npix >> 2); q += gridDim.x * 256ll) {
            const uint32_t w0 = s32[q * 3], w1 = s32[q * 3 + 1], w2 = s32[q * 3 + 2];      // r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
            const float4 r = make_float4((float)(w0 & 255u), (float)(w0 >> 24), (float)((w1 >> 16) & 255u), (float)((w2 >> 8) & 255u));
            const float4 g = make_float4((float)((w0 >> 8) & 255u), (float)(w1 & 255u), (float)(w1 >> 24), (float)((w2 >> 16) & 255u));
            const float4 bl = make_float4((float)((w0 >> 16) & 255u), (float)((w1 >> 8) & 255u), (float)(w2 & 255u), (float)(w2 >> 24));
            *(float4 *)(o + q * 4) = r;
            *(float4 *)(o + npix + q * 4) = g;
            *(float4 *)(o + 2 * npix + q * 4) = bl;
        }
        return;
    }
    for (long long p = blockIdx.x * 256ll + threadIdx.x; p < npix; p += gridDim.x * 256ll)
        for (int c = 0; c < C; ++c) out[((long long)b * C + c) * npix + p] = (float)src[((long long)b * npix + p) * C + c];
}

extern "C" int spa_resize_bicubic_u8(spa_ctx *ctx, const uint8_t *src, int32_t B, int32_t H, int32_t W, int32_t C,
                                     int32_t dst_h, int32_t dst_w, float *out, void *stream)
{
    SPA_ARG(ctx && src && out && B > 0 && H > 0 && W > 0 && C > 0 && C <= 4 && dst_h > 0 && dst_w > 0);
    hipStream_t s = spa_stream(stream);
    if (dst_h == H && dst_w == W) {
        const long long npix = (long long)H * W;
        int g = (int)((npix + 255) / 256);
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(k_u8_to_planar, dim3(g, B), dim3(256), 0, s, src, npix, C, out);
        SPA_LAUNCH_CHECK();
        return SPA_OK;
    }
    // weight tables of this (input, output) size: built once, kept on the device
    if (ctx->rs_key[0] != H || ctx->rs_key[1] != W || ctx->rs_key[2] != dst_h || ctx->rs_key[3] != dst_w) {
        int32_t *bx = (int32_t *)malloc((size_t)dst_w * 2 * 4), *by = (int32_t *)malloc((size_t)dst_h * 2 * 4);
        int32_t *kx, *ky;
        const int ksx = rs_coeffs(W, dst_w, bx, &kx), ksy = rs_coeffs(H, dst_h, by, &ky);
        const size_t nbx = (size_t)dst_w * 2, nkx = (size_t)dst_w * ksx, nby = (size_t)dst_h * 2, nky = (size_t)dst_h * ksy;
        int32_t *dev;
        int rc = spa_ws_reserve(ctx, WS_RESIZE_TAB, (nbx + nkx + nby + nky) * 4, (void **)&dev);
        if (rc == SPA_OK) {
            // synchronous copies (pageable host memory; rare: once per shape)
            SPA_HIP(hipStreamSynchronize(s));
            SPA_HIP(hipMemcpy(dev, bx, nbx * 4, hipMemcpyHostToDevice));
            SPA_HIP(hipMemcpy(dev + nbx, kx, nkx * 4, hipMemcpyHostToDevice));
            SPA_HIP(hipMemcpy(dev + nbx + nkx, by, nby * 4, hipMemcpyHostToDevice));
            SPA_HIP(hipMemcpy(dev + nbx + nkx + nby, ky, nky * 4, hipMemcpyHostToDevice));
            ctx->rs_key[0] = H; ctx->rs_key[1] = W; ctx->rs_key[2] = dst_h; ctx->rs_key[3] = dst_w;
            ctx->rs_ks[0] = ksx; ctx->rs_ks[1] = ksy;
        }
        free(bx); free(by); free(kx); free(ky);
        if (rc != SPA_OK) return rc;
    }
    const int ksx = ctx->rs_ks[0], ksy = ctx->rs_ks[1];
    int32_t *dev = (int32_t *)ctx->ws[WS_RESIZE_TAB];
    const int32_t *bx = dev, *kx = dev + (size_t)dst_w * 2, *by = kx + (size_t)dst_w * ksx, *ky = by + (size_t)dst_h * 2;
    uint8_t *tmp;
    int rc = spa_ws_reserve(ctx, WS_RESIZE_TMP, (size_t)B * H * dst_w * C, (void **)&tmp);
    if (rc != SPA_OK) return rc;
    // Pillow skips a pass whose size does not change; an unchanged axis has identity taps only when scale
    // is exactly 1 (support 2 -> 5 taps (0, 0, 1, 0, 0)), so running the pass anyway gives the same bytes
    hipLaunchKernelGGL(k_resize_h, dim3((dst_w * C + 255) / 256, H, B), dim3(256), 0, s, src, H, W, C, dst_w, bx, kx, ksx, tmp);
    hipLaunchKernelGGL(k_resize_v, dim3((dst_w * C + 255) / 256, dst_h, B), dim3(256), 0, s, (const uint8_t *)tmp, H, dst_w, C,
                       dst_h, by, ky, ksy, out);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}


// ---------------------------------------------------------------------------------------------------
// The OpenCV branch of the reference's input resize: datasets/resize_image_dataset.py:20-36 and
// datasets/zipped_cityscapes_road_dataset.py:78-85 decode to uint8, resize THE UINT8 IMAGE (chainercv.transforms.resize(img,
// shape, 3) with cv2 importable = cv2.resize(uint8 HWC, (w, h), interpolation=cv2.INTER_CUBIC)) and only then .astype(float32):
// OpenCV's 8-bit path, integer-valued results in 0..255.  (Round 3 restated the float32 path here; the reference never holds a
// float32 image at that point — ADVICE r3.)  NOT PINNED: there is no cv2 in the build image and no fixture in the reference;
// the kernel follows OpenCV's published scalar algorithm (modules/imgproc/src/resize.cpp: resizeGeneric_<HResizeCubic<uchar,
// int, short>, VResizeCubic<uchar, int, short, FixedPtCast<int, uchar, 22>>>):
//   per destination index d:  f = (float)((d + 0.5) * scale - 0.5);  s = floor(f);  f -= s;  interpolateCubic(f), A = -0.75f,
//   float32; the four taps as saturate_cast<short>(c * 2048) (cvRound = round half to even);
//   horizontal: int32 sums of byte x tap over s-1 .. s+2, indices clamped to the row (border replication);
//   vertical: int32 sum of the four rows clip(s - 1 + k, 0, H - 1) x tap, then (v + 2^21) >> 22 saturated to 0..255.
// and is bit-identical to the restatement oracle/resize_oracle.c:orc_resize_cvcubic_u8 (tests/test_gpu_parity.py).  OpenCV's
// vector form of the vertical pass (VResizeCubicVec_32s8u: float products, round to nearest even) can differ from the scalar
// fixed-point form above in the last bit of rare pixels: one more reason this branch is stated, not pinned.
// One thread = one destination pixel (all channels).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cv_cubic_taps_s16(float x, int (&t)[4])
{
    const float A = -0.75f;
    float c[4];
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = (int)rintf(c[k] * 2048.f);                  // saturate_cast<short>(float): cvRound, then the range
        t[k] = r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
    }
}

__global__ __launch_bounds__(256) void k_resize_cvcubic(const uint8_t *__restrict__ src, int H, int W, int C, int h, int w,
                                                        double sx, double sy, float *__restrict__ out)
{
    const int b = blockIdx.z, dy = blockIdx.y;
    const int dx = blockIdx.x * 256 + threadIdx.x;
    if (dx >= w) return;
    float fx = (float)((dx + 0.5) * sx - 0.5), fy = (float)((dy + 0.5) * sy - 0.5);
    const int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
    fx -= (float)x0; fy -= (float)y0;
    int a[4], bb[4];
    cv_cubic_taps_s16(fx, a);
    cv_cubic_taps_s16(fy, bb);
    int xs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int v = x0 - 1 + k; xs[k] = v < 0 ? 0 : (v >= W ? W - 1 : v); }
    const uint8_t *img = src + (long long)b * H * W * C;
    for (int c = 0; c < C; ++c) {
        int v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int yy = y0 - 1 + k;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            const uint8_t *S = img + (long long)yy * W * C + c;
            v += ((int)S[xs[0] * C] * a[0] + (int)S[xs[1] * C] * a[1] + (int)S[xs[2] * C] * a[2] + (int)S[xs[3] * C] * a[3]) * bb[k];
        }
        v = (v + (1 << 21)) >> 22;
        out[(((long long)b * C + c) * h + dy) * w + dx] = (float)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

// src (B,H,W,C) uint8 interleaved (a decoded PNG) -> out (B,C,dst_h,dst_w) float32 planar holding the resized BYTES (what
// the reference's .astype(float32) of the resized uint8 image holds).  With dst == src size only the layout/dtype change is made (the
// reference resizes only when the shapes differ).
extern "C" int spa_resize_cvcubic_u8(spa_ctx *ctx, const uint8_t *src, int32_t B, int32_t H, int32_t W, int32_t C,
                                     int32_t dst_h, int32_t dst_w, float *out, void *stream)
{
    SPA_ARG(ctx && src && out && B > 0 && H > 0 && W > 0 && C > 0 && C <= 4 && dst_h > 0 && dst_w > 0);
    SPA_ARG(dst_h < 65536 && B < 65536);
    hipStream_t s = spa_stream(stream);
    if (dst_h == H && dst_w == W) {
        const long long npix = (long long)H * W;
        int g = (int)((npix + 255) / 256);
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(k_u8_to_planar, dim3(g, B), dim3(256), 0, s, src, npix, C, out);
        SPA_LAUNCH_CHECK();
        return SPA_OK;
    }
    hipLaunchKernelGGL(k_resize_cvcubic, dim3((dst_w + 255) / 256, dst_h, B), dim3(256), 0, s, src, H, W, C, dst_h, dst_w,
                       (double)W / (double)dst_w, (double)H / (double)dst_h, out);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
