// SLIC on gfx950: rgb -> Lab, the Lloyd sweeps of skimage's _slic_cython, bit exact.
//
// What "bit exact" forces on the design ("Exactness"):
//  * skimage instantiates the Cython core in float32 for a float32 image, and rounds after
//    every operation (no FMA in its x86-64 wheels).  This file is compiled with
//    -ffp-contract=off and spells every float32 operation in skimage's order.
//  * the assignment loop of skimage is centre-major ("for each centre, for each pixel of its
//    2S window, if dist < best") so among equal distances the LOWEST centre index that is
//    strictly better wins.  slic_assign is pixel-major and walks the candidate centres of a
//    tile in increasing index order with the same strict comparison.
//  * the centroid update of skimage is a float32 running sum over the pixels of a segment in
//    raster order.  float32 addition is not associative and the sums are large (10^4 terms
//    of magnitude 10^2..10^3), so any tree reduction changes the centroids by ~1e-5 relative
//    and the labels of hundreds of boundary pixels with them.  slic_update therefore keeps
//    the serial order: one wavefront per segment gathers the segment's pixels in raster order
//    (coalesced label reads + ballot/mbcnt compaction into an LDS ring) and five lanes carry
//    the five running sums (y, x, L, a, b) through the ring.  The chains of different segments
//    are independent, so the GPU runs thousands of them concurrently (B * n_centroids waves).
#include "spa_common.h"
#include <stdlib.h>
#include "spa_glibcf.h"

int spa_slic_core_general_f32(spa_ctx *ctx, const float *lab, int32_t B, int32_t H, int32_t W, int32_t n_segments,
                              int32_t max_iter, int32_t *labels, float *centres, void *stream);

// ------------------------------------------------------------------------------------
// rgb -> scaled Lab (skimage/color/colorconv.py:657-661, :950-969), float32 steps in the
// reference's order; np.power(float32, 2.4) = powf(x, 2.4f) and np.cbrt = cbrtf as the C library of the
// reference configuration evaluates them (spa_glibcf.h): the Lab image is scikit-image's bit for bit.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float lab_f(float s, const spa_glibcf_tables *__restrict__ T)
{
    if (s > (float)0.008856) return spa_glibc_cbrtf(s, T);
    return (float)7.787 * s + (float)(16.0 / 116.0);
}

// sRGB companding of one channel value, exactly as the reference evaluates it
__device__ __forceinline__ float srgb_linear(float a, const spa_glibcf_tables *__restrict__ T)
{
    if (a > (float)0.04045) {
        float u = (a + (float)0.055) / (float)1.055;
        return spa_glibc_powf_2p4(u, T);
    }
    return a / (float)12.92;
}

// the 256 values a decoded 8-bit image can take (the reference passes 0..255 floats, unscaled):
// one table per context, filled by this very function, so a table hit is bit-identical to the
// evaluation it replaces and halves the transcendental work for such images
__global__ void k_srgb_lut(float *__restrict__ lut)
{
    lut[threadIdx.x] = srgb_linear((float)threadIdx.x, &spa_glibcf_global);
}

// does the image look like a decoded 8-bit one?  (a sample of the first image; the table path
// still checks every value, so a wrong guess costs time, never correctness)
__global__ void k_srgb_probe(const float *__restrict__ rgb, long long n, float *__restrict__ lut)
{
    const long long step = n / 4096 > 0 ? n / 4096 : 1;
    bool ok = true;
    for (int u = 0; u < 16; ++u) {
        const long long i = ((long long)threadIdx.x * 16 + u) * step;
        if (i < n) { const float a = rgb[i]; ok = ok && a >= 0.0f && a <= 255.0f && (float)(int)a == a; }
    }
    const int all = __syncthreads_and(ok ? 1 : 0);
    if (threadIdx.x == 0) ((int *)lut)[256] = all;
}

template <bool TABLE>
__device__ __forceinline__ void rgb2lab_px(float r, float g, float b, float ratio,
                                           const float *__restrict__ lut, const spa_glibcf_tables *__restrict__ T,
                                           float &L, float &A, float &Bc)
{
    float v[3] = {r, g, b};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a = v[c];
        if (TABLE) {
            const int ai = (int)a;
            v[c] = (a >= 0.0f && a <= 255.0f && (float)ai == a) ? lut[ai] : srgb_linear(a, T);
        } else {
            v[c] = srgb_linear(a, T);
        }
    }
    float X = (float)0.412453 * v[0];
    X = X + (float)0.357580 * v[1];
    X = X + (float)0.180423 * v[2];
    float Y = (float)0.212671 * v[0];
    Y = Y + (float)0.715160 * v[1];
    Y = Y + (float)0.072169 * v[2];
    float Z = (float)0.019334 * v[0];
    Z = Z + (float)0.119193 * v[1];
    Z = Z + (float)0.950227 * v[2];
    float fx = lab_f(X / (float)0.95047, T);
    float fy = lab_f(Y / (float)1.0, T);
    float fz = lab_f(Z / (float)1.08883, T);
    L = ((float)116.0 * fy - (float)16.0) * ratio;
    A = ((float)500.0 * (fx - fy)) * ratio;
    Bc = ((float)200.0 * (fy - fz)) * ratio;
}

template <bool TABLE>
__device__ __forceinline__ void rgb2lab_sweep(const float *__restrict__ src, float *__restrict__ dst,
                                              long long npix, float ratio, int vec4,
                                              const float *__restrict__ lut,
                                              const spa_glibcf_tables *__restrict__ T)
{
    long long stride = (long long)gridDim.x * blockDim.x;
    if (vec4) {
        long long n4 = npix >> 2;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            float4 r = ((const float4 *)src)[i];
            float4 g = ((const float4 *)(src + npix))[i];
            float4 bl = ((const float4 *)(src + 2 * npix))[i];
            float4 L, A, Bc;
            rgb2lab_px<TABLE>(r.x, g.x, bl.x, ratio, lut, T, L.x, A.x, Bc.x);
            rgb2lab_px<TABLE>(r.y, g.y, bl.y, ratio, lut, T, L.y, A.y, Bc.y);
            rgb2lab_px<TABLE>(r.z, g.z, bl.z, ratio, lut, T, L.z, A.z, Bc.z);
            rgb2lab_px<TABLE>(r.w, g.w, bl.w, ratio, lut, T, L.w, A.w, Bc.w);
            ((float4 *)dst)[i] = L;
            ((float4 *)(dst + npix))[i] = A;
            ((float4 *)(dst + 2 * npix))[i] = Bc;
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
            float L, A, Bc;
            rgb2lab_px<TABLE>(src[i], src[npix + i], src[2 * npix + i], ratio, lut, T, L, A, Bc);
            dst[i] = L;
            dst[npix + i] = A;
            dst[2 * npix + i] = Bc;
        }
    }
}

__global__ __launch_bounds__(256) void k_rgb2lab(const float *__restrict__ rgb,
                                                 float *__restrict__ lab, long long npix,
                                                 float ratio, int vec4, const float *__restrict__ lut)
{
    const int b = blockIdx.y;
    const float *src = rgb + (long long)b * 3 * npix;
    float *dst = lab + (long long)b * 3 * npix;
    __shared__ spa_glibcf_tables tabs;
    spa_glibcf_stage(&tabs);
    __syncthreads();
    if (((const int *)lut)[256]) rgb2lab_sweep<true>(src, dst, npix, ratio, vec4, lut, &tabs);     // wave-uniform
    else rgb2lab_sweep<false>(src, dst, npix, ratio, vec4, lut, &tabs);
}

extern "C" int spa_rgb2lab(spa_ctx *ctx, const float *rgb, int32_t B, int32_t H, int32_t W,
                           float ratio, float *lab, void *stream)
{
    SPA_ARG(ctx && rgb && lab && B > 0 && H > 0 && W > 0);
    long long npix = (long long)H * W;
    int vec4 = (npix % 4 == 0) && (((uintptr_t)rgb | (uintptr_t)lab) % 16 == 0);
    long long work = vec4 ? npix / 4 : npix;
    int gx = (int)((work + 255) / 256);
    if (gx > 2048) gx = 2048;
    float *lut;
    const bool fresh = ctx->ws_bytes[WS_SRGB_LUT] == 0;
    int rc = spa_ws_reserve(ctx, WS_SRGB_LUT, 257 * sizeof(float), (void **)&lut);
    if (rc != SPA_OK) return rc;
    if (fresh) hipLaunchKernelGGL(k_srgb_lut, dim3(1), dim3(256), 0, spa_stream(stream), lut);
    hipLaunchKernelGGL(k_srgb_probe, dim3(1), dim3(256), 0, spa_stream(stream), rgb, 3 * npix, lut);
    { SpaProfScope prof_(ctx, PROF_RGB2LAB, spa_stream(stream));
    hipLaunchKernelGGL(k_rgb2lab, dim3(gx, B), dim3(256), 0, spa_stream(stream), rgb, lab, npix,
                       ratio, vec4, (const float *)lut); }
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// ------------------------------------------------------------------------------------
// centre table: 12 words per (image, centre)
//   [0] cy [1] cx [2] cL [3] ca [4] cb [5] -  [6] y0 [7] y1 [8] x0 [9] x1 [10] count [11] -
//   [12] by0 [13] by1 [14] bx0 [15] bx1 : bounding box (inclusive) of the pixels the last
//   assignment sweep gave to this centre, accumulated by k_slic_assign with atomics and
//   consumed + reset by k_slic_update
// [y0,y1) x [x0,x1) is skimage's search window of the centre:
//   y_min = <Py_ssize_t>max(cy - 2*step_y, 0); y_max = <Py_ssize_t>min(cy + 2*step_y + 1, H)
// ------------------------------------------------------------------------------------
#define CEN_WORDS 16

__device__ __forceinline__ void slic_window(float cy, float cx, int s2y, int s2x, int H, int W,
                                            int &y0, int &y1, int &x0, int &x1)
{
    float fy0 = cy - (float)s2y;
    if (!(fy0 > 0.0f)) fy0 = 0.0f;
    float fy1 = (cy + (float)s2y) + 1.0f;
    if (!(fy1 < (float)H)) fy1 = (float)H;
    float fx0 = cx - (float)s2x;
    if (!(fx0 > 0.0f)) fx0 = 0.0f;
    float fx1 = (cx + (float)s2x) + 1.0f;
    if (!(fx1 < (float)W)) fx1 = (float)W;
    y0 = (int)fy0; y1 = (int)fy1; x0 = (int)fx0; x1 = (int)fx1;
}

__global__ void k_slic_init(uint32_t *__restrict__ cen, int nC, int grid_nx, int start_y,
                            int start_x, int step_y, int step_x, int s2y, int s2x, int H, int W)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    int b = blockIdx.y;
    if (k >= nC) return;
    int iy = k / grid_nx, ix = k % grid_nx;
    float cy = (float)(start_y + iy * step_y), cx = (float)(start_x + ix * step_x);
    int y0, y1, x0, x1;
    slic_window(cy, cx, s2y, s2x, H, W, y0, y1, x0, x1);
    uint32_t *c = cen + ((long long)b * nC + k) * CEN_WORDS;
    c[0] = __float_as_uint(cy); c[1] = __float_as_uint(cx);
    c[2] = 0u; c[3] = 0u; c[4] = 0u; c[5] = 0u;
    c[6] = (uint32_t)y0; c[7] = (uint32_t)y1; c[8] = (uint32_t)x0; c[9] = (uint32_t)x1;
    c[10] = 0u; c[11] = 0u;
    c[12] = 0x7fffffffu; c[13] = 0u; c[14] = 0x7fffffffu; c[15] = 0u;
}

// ------------------------------------------------------------------------------------
// assignment sweep: one 32x32 pixel tile per 256-thread workgroup, 4 pixels per thread.
// Candidate centres (windows intersecting the tile) are compacted in increasing index
// order into LDS, then every thread walks the list; LDS reads are wave-uniform broadcasts.
// Algorithmic HBM bytes per pixel per sweep: 12 (Lab) + 4 (label).
// ------------------------------------------------------------------------------------
#define TILE 32
typedef float f32x2 __attribute__((ext_vector_type(2)));

// LDSX (diagnostic builds only, make EXTRA=-DSPA_DIAG; spa_debug_set(ctx, 2, 1)): round 4's shelved variant of the candidate loop — the x
// part of the spatial term shared through an LDS table — kept as the REPRODUCER of the co-residency miscompare (DESIGN.md section 7,
// tools/race_probe8.py): bit-identical alone, 50-700 wrong centre words per batch when a wave of a matrix-instruction kernel shares the CU.
template <bool LDSX>
__global__ __launch_bounds__(256) void k_slic_assign(const float *__restrict__ lab,
                                                     uint32_t *__restrict__ cen, int nC,
                                                     int H, int W, float sw,
                                                     int32_t *__restrict__ labels,
                                                     unsigned long long *__restrict__ rowmask, int HG, int PW,
                                                     uint32_t *__restrict__ fine, int RW, int PWF,
                                                     uint32_t *__restrict__ status)
{
    __shared__ uint4 cand[256 * 3];
    __shared__ __attribute__((aligned(16))) float cdx[LDSX ? 32 * TILE : 4];
    __shared__ int wave_cnt[4];
    const int b = blockIdx.z;
    const int ty0 = blockIdx.y * TILE, tx0 = blockIdx.x * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long npix = (long long)H * W;
    const float *pl = lab + (long long)b * 3 * npix;
    uint32_t *cb = cen + (long long)b * nC * CEN_WORDS;

    const int y = ty0 + (tid >> 3);
    const int xb = tx0 + (tid & 7) * 4;
    const bool row_ok = y < H;
    float pL[4], pA[4], pB[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ok[i] = row_ok && (xb + i) < W;
    const long long base = (long long)y * W + xb;
    if (ok[3] && ((base & 3) == 0)) {
        float4 l4 = *(const float4 *)(pl + base);
        float4 a4 = *(const float4 *)(pl + npix + base);
        float4 b4 = *(const float4 *)(pl + 2 * npix + base);
        pL[0] = l4.x; pL[1] = l4.y; pL[2] = l4.z; pL[3] = l4.w;
        pA[0] = a4.x; pA[1] = a4.y; pA[2] = a4.z; pA[3] = a4.w;
        pB[0] = b4.x; pB[1] = b4.y; pB[2] = b4.z; pB[3] = b4.w;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pL[i] = ok[i] ? pl[base + i] : 0.0f;
            pA[i] = ok[i] ? pl[npix + base + i] : 0.0f;
            pB[i] = ok[i] ? pl[2 * npix + base + i] : 0.0f;
        }
    }
    float best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int bl[4] = {-1, -1, -1, -1};
    const float fy = (float)y;
    const f32x2 pL2[2] = {{pL[0], pL[1]}, {pL[2], pL[3]}}, pA2[2] = {{pA[0], pA[1]}, {pA[2], pA[3]}},
                pB2[2] = {{pB[0], pB[1]}, {pB[2], pB[3]}};
    const f32x2 fx2[2] = {{(float)xb, (float)(xb + 1)}, {(float)(xb + 2), (float)(xb + 3)}};

    for (int kb = 0; kb < nC; kb += 256) {
        const int k = kb + tid;
        bool hit = false;
        uint4 w0, w1, w2;
        if (k < nC) {
            const uint4 *c = (const uint4 *)(cb + (long long)k * CEN_WORDS);
            w0 = c[0]; w1 = c[1]; w2 = c[2];
            int y0 = (int)w1.z, y1 = (int)w1.w, x0 = (int)w2.x, x1 = (int)w2.y;
            hit = (y0 < ty0 + TILE) && (y1 > ty0) && (x0 < tx0 + TILE) && (x1 > tx0);
        }
        unsigned long long m = __ballot(hit);
        if (lane == 0) wave_cnt[wv] = __popcll(m);
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = wave_cnt[i];
            if (i < wv) off += c;
            total += c;
        }
        if (hit) {
            int pos = off + (int)spa_rank_in_mask(m);
            // entry: (cy, cx, cL, ca) (cb, k, y0, y1) (x0, x1, -, -)
            cand[pos * 3 + 0] = w0;
            cand[pos * 3 + 1] = make_uint4(w1.x, (uint32_t)k, w1.z, w1.w);
            // .z: the window contains the whole tile and the tile lies inside the image — no per-pixel window test for this entry
            const bool whole = (int)w1.z <= ty0 && (int)w1.w >= ty0 + TILE && (int)w2.x <= tx0 && (int)w2.y >= tx0 + TILE &&
                               ty0 + TILE <= H && tx0 + TILE <= W;
            cand[pos * 3 + 2] = make_uint4(w2.x, w2.y, whole ? 1u : 0u, 0u);
        }
        __syncthreads();
        if constexpr (LDSX) {
            // (cx - x)^2 per (candidate, column), +inf outside the candidate's columns or the image, once per tile in LDS; sub-chunks of 32
            // candidates.  Every read of the table is fenced by workgroup barriers; the table's contents check out after the loop — and
            // the values a lane READS from it in the loop are wrong now and then beside matrix-instruction waves (see above).
            for (int j0 = 0; j0 < total; j0 += 32) {
                const int jn = min(32, total - j0);
                for (int e = tid; e < jn * TILE; e += 256) {
                    const int j = j0 + (e >> 5), xi = e & 31;
                    const uint4 e0 = cand[j * 3 + 0], e2 = cand[j * 3 + 2];
                    const int xx = tx0 + xi;
                    const float t = __uint_as_float(e0.y) - (float)xx;
                    const bool in = ((unsigned)(xx - (int)e2.x) < (unsigned)((int)e2.y - (int)e2.x)) && xx < W;
                    cdx[(e >> 5) * TILE + xi] = in ? t * t : INFINITY;
                }
                __syncthreads();
                if (row_ok) {
#pragma unroll 1
                    for (int jj = 0; jj < jn; ++jj) {
                        const int j = j0 + jj;
                        const uint4 e0 = cand[j * 3 + 0], e1 = cand[j * 3 + 1];
                        const float4 dx4 = *(const float4 *)(cdx + jj * TILE + (tid & 7) * 4);
                        const int y0 = (int)e1.z, y1 = (int)e1.w;
                        const float cy = __uint_as_float(e0.x);
                        const float cl = __uint_as_float(e0.z), ca = __uint_as_float(e0.w);
                        const float cbb = __uint_as_float(e1.x);
                        const int kk = (int)e1.y;
                        const bool rowin = (y >= y0) && (y < y1);
                        const float ty = cy - fy;
                        const float dy = rowin ? ty * ty : INFINITY;
                        const f32x2 dxp[2] = {{dx4.x, dx4.y}, {dx4.z, dx4.w}};
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            f32x2 dc = (f32x2{dy, dy} + dxp[h]) * f32x2{sw, sw};
                            const f32x2 t0 = pL2[h] - f32x2{cl, cl}, t1 = pA2[h] - f32x2{ca, ca}, t2 = pB2[h] - f32x2{cbb, cbb};
                            f32x2 col = t0 * t0;
                            col = col + t1 * t1;
                            col = col + t2 * t2;
                            dc = dc + col;
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const int i = 2 * h + q;
                                const float d = q ? dc.y : dc.x;
                                const bool take = best[i] > d;
                                best[i] = take ? d : best[i];
                                bl[i] = take ? kk : bl[i];
                            }
                        }
                    }
                }
                __syncthreads();
            }
        } else {
            if (row_ok) {
                // straight-line body: the three 16-byte LDS reads of an entry are issued together and
                // the window test is folded into the final comparison (no divergent branches), so the
                // compiler can overlap the next entry's reads with this entry's arithmetic.  The
                // float arithmetic runs on pixel pairs (v_pk_add_f32 / v_pk_mul_f32: two IEEE float32
                // operations per instruction, each rounded exactly like the scalar one — no FMA).
    #pragma unroll 1
                for (int j = 0; j < total; ++j) {
                    const uint4 e0 = cand[j * 3 + 0], e1 = cand[j * 3 + 1], e2 = cand[j * 3 + 2];
                    const int y0 = (int)e1.z, y1 = (int)e1.w, x0 = (int)e2.x, x1 = (int)e2.y;
                    const float cy = __uint_as_float(e0.x), cx = __uint_as_float(e0.y);
                    const float cl = __uint_as_float(e0.z), ca = __uint_as_float(e0.w);
                    const float cbb = __uint_as_float(e1.x);
                    const int kk = (int)e1.y;
                    const float ty = cy - fy;
                    const float dy = ty * ty;
                    // (round 6) most entries' windows (4S x 4S) contain the whole 32 x 32 tile: a wave-uniform branch on the entry's
                    // flag drops the row / column tests (11 of 48 vector instructions per entry); same arithmetic, same comparisons
                    const bool whole = __builtin_amdgcn_readfirstlane((int)e2.z) != 0;
                    if (whole) {
    #pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const f32x2 tx = f32x2{cx, cx} - fx2[h];
                            const f32x2 dx = tx * tx;
                            f32x2 dc = (f32x2{dy, dy} + dx) * f32x2{sw, sw};
                            const f32x2 t0 = pL2[h] - f32x2{cl, cl}, t1 = pA2[h] - f32x2{ca, ca}, t2 = pB2[h] - f32x2{cbb, cbb};
                            f32x2 col = t0 * t0;
                            col = col + t1 * t1;
                            col = col + t2 * t2;
                            dc = dc + col;
    #pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const int i = 2 * h + q;
                                const float d = q ? dc.y : dc.x;
                                const bool take = best[i] > d;
                                best[i] = take ? d : best[i];
                                bl[i] = take ? kk : bl[i];
                            }
                        }
                        continue;
                    }
                    const bool rowin = (y >= y0) && (y < y1);
                    const unsigned xw = (unsigned)(x1 - x0);
    #pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2 tx = f32x2{cx, cx} - fx2[h];
                        const f32x2 dx = tx * tx;
                        f32x2 dc = (f32x2{dy, dy} + dx) * f32x2{sw, sw};
                        const f32x2 t0 = pL2[h] - f32x2{cl, cl}, t1 = pA2[h] - f32x2{ca, ca}, t2 = pB2[h] - f32x2{cbb, cbb};
                        f32x2 col = t0 * t0;
                        col = col + t1 * t1;
                        col = col + t2 * t2;
                        dc = dc + col;
    #pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int i = 2 * h + q;
                            const float d = q ? dc.y : dc.x;
                            const bool take = rowin && ok[i] && ((unsigned)(xb + i - x0) < xw) && (best[i] > d);
                            best[i] = take ? d : best[i];
                            bl[i] = take ? kk : bl[i];
                        }
                    }
                }
            }
            __syncthreads();
        }
        }
    bool uncovered = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) uncovered = uncovered || (ok[i] && bl[i] < 0);
    if (uncovered) atomicOr(status, SPA_ST_SLIC_UNCOVERED);
    // pixel masks of the new segments for the centroid update.  The distinct labels of a wave
    // (8 rows x 32 columns, lane = row*8 + column group, 4 pixels per lane) are enumerated with
    // ballots.  For each label the wave leaves
    //   * fine: one 32-bit word per row, bit (i*8 + column group) = pixel 4*group+i of the row
    //     carries the label, stored in the label's window-relative table (plain stores: each
    //     (label, row, 32-pixel half piece) belongs to exactly one wave);
    //   * coarse: one bit per (row, 64-pixel piece) in the mask word of its (8-row group, piece
    //     octet) — one atomicOr per (wave, label); scattered atomics run at a fixed chip-wide rate.
    // The update kernel walks the coarse bits, reads the fine words and never touches the labels.
    if (fine) {
        bool val[4], done[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { val[i] = ok[i] && bl[i] >= 0; done[i] = !val[i]; }
        const int piece = tx0 >> 6;
        const int half = (tx0 >> 5) & 1;
        while (true) {
            const unsigned long long pend = __ballot(!(done[0] && done[1] && done[2] && done[3]));
            if (!pend) break;
            const int leader = __ffsll((long long)pend) - 1;
            const int mine = !done[0] ? bl[0] : (!done[1] ? bl[1] : (!done[2] ? bl[2] : bl[3]));
            const int ll = __shfl(mine, leader);
            unsigned long long bi[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool m = val[i] && bl[i] == ll;
                bi[i] = __ballot(m);
                done[i] = done[i] || m;
            }
            uint32_t rowbits = 0u;
            if (lane < 8) {
                const int sh = lane * 8;
                rowbits = (uint32_t)((bi[0] >> sh) & 0xFFull) | ((uint32_t)((bi[1] >> sh) & 0xFFull) << 8) |
                          ((uint32_t)((bi[2] >> sh) & 0xFFull) << 16) | ((uint32_t)((bi[3] >> sh) & 0xFFull) << 24);
                if (rowbits) {
                    const uint32_t *c = cb + (long long)ll * CEN_WORDS;
                    const int wy0 = (int)c[6], wx0 = (int)c[8];
                    const int yy = ty0 + wv * 8 + lane;
                    fine[((((long long)b * nC + ll) * RW + (yy - wy0)) * PWF + (piece - (wx0 >> 6))) * 2 + half] = rowbits;
                }
            }
            const unsigned long long rb = __ballot(rowbits != 0u) & 0xFFull;
            if (lane == leader)
                atomicOr(rowmask + (((long long)b * nC + ll) * HG + ((ty0 >> 3) + wv)) * PW + (piece >> 3),
                         rb << ((piece & 7) * 8));
        }
    }
    int32_t *out = labels + (long long)b * npix;
    if (ok[3] && ((base & 3) == 0)) {
        *(int4 *)(out + base) = make_int4(bl[0], bl[1], bl[2], bl[3]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ok[i]) out[base + i] = bl[i];
    }
}

// ------------------------------------------------------------------------------------
// centroid update: one WAVEFRONT per (image, centre), four independent wavefronts per
// workgroup, no workgroup barriers.  Raster-order float32 sums.
//
// skimage adds the pixels of a segment to float32 accumulators in raster order; float32
// addition does not commute with regrouping, so the order is kept: lanes 0..4 of the wave carry
// the five running sums (y, x, L, a, b) through the segment's pixels one by one.  Feeding that
// serial chain is what the rest of the wave does, and none of it is per-piece scalar work:
//   * the assignment sweep left, per segment, one coarse bit per (row, 64-pixel piece) and one
//     fine 64-bit pixel mask per set coarse bit (window-relative table).  The wave expands the
//     coarse bits into a raster-ordered piece list (LDS) — the label image is never read;
//   * 64 list entries at a time, lane j owns entry j: it reads (and clears) the entry's fine
//     mask, and a wave prefix sum of the mask populations gives every entry its place in the
//     segment's pixel stream.  Each lane then writes the packed (row, x) of its mask's set bits
//     to that place of an LDS index buffer — as many leading entries as fit the buffer;
//   * the stream is consumed in rounds of UPD_ROUND pixels: lane i reads index i, loads L, a, b
//     of that pixel (every lane of every load is a wanted pixel; five statically named register
//     sets keep four rounds of loads in flight, the schedule is straight-line so that the
//     compiler's vmcnt waits stay exact) and writes y, x, L, a, b to the wave's staging rows;
//   * lanes 0..4 then walk the staging rows: four 16-byte LDS reads per 16 dependent adds, the
//     reads of the next 16 pixels in flight meanwhile.
// HBM traffic per sweep ~ 12 B/pixel (Lab, plus partially used lines at segment borders) and
// 8 B per (row, piece) of mask.
// ------------------------------------------------------------------------------------
#define UPD_WAVES 4                        // independent segments per workgroup
#define UPD_ROUND 128                      // pixels staged per round (two loads of 64 per plane)
#define UPD_ROW (UPD_ROUND + 36)           // floats per staging row (+read-ahead slack; 164 % 64 = 36 spreads the 5 rows over LDS banks)
#define UPD_PCAP 512                       // piece-list capacity (entries: row in window << 8 | piece)
#define UPD_PWMAX 8                        // occupancy words per 8-row group: 8 x 8 pieces x 64 pixels = 4096 pixels
#define UPD_XBITS 12                       // stream codes: row in window << 12 | x
#define UPD_ICAP 2048                      // index-buffer capacity (pixels)

// one round in flight: packed (row << UPD_XBITS | x) and Lab of 2 x 64 stream pixels
template <int PWM> struct UpdEntry;
template <> struct UpdEntry<4> { typedef unsigned short type; static constexpr int piece_bits = 5; };
template <> struct UpdEntry<8> { typedef unsigned type; static constexpr int piece_bits = 8; };

struct UpdRound { unsigned code[2]; float vL[2], vA[2], vB[2]; int fill; };

#ifdef SPA_UPD_TIMING      // development aid: per-phase wave cycles summed into the words after the queue heads
#define UPD_T0() unsigned long long t_ = __builtin_readcyclecounter(), tacc_[6] = {0, 0, 0, 0, 0, 0}
#define UPD_T(i) { const unsigned long long n_ = __builtin_readcyclecounter(); tacc_[i] += n_ - t_; t_ = n_; }
#define UPD_TFLUSH(q) if (lane == 0) { for (int i_ = 0; i_ < 6; ++i_) atomicAdd((unsigned long long *)(q) + 8 + i_, tacc_[i_]); }
#else
#define UPD_T0()
#define UPD_T(i)
#define UPD_TFLUSH(q)
#endif

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// inclusive prefix sum over the 64 lanes with DPP moves (no LDS round trips)
__device__ __forceinline__ int upd_wave_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);      // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);      // row_bcast:31 -> rows 2, 3
    return v;
}

// the sweep stores bit (i*8 + g) for pixel 4*g + i of a 32-pixel half row; back to bit = pixel
__device__ __forceinline__ unsigned upd_pixel_order(unsigned w)
{
    unsigned out = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned v = (w >> (8 * i)) & 0xFFu;
        v = (v | (v << 12)) & 0x000F000Fu;
        v = (v | (v << 6)) & 0x03030303u;
        v = (v | (v << 3)) & 0x11111111u;
        out |= v << i;
    }
    return out;
}

// Longest-first work lists.  Segment sizes differ by a factor of 3-4 and a segment is one serial
// chain, so the launch is as long as its unluckiest wave unless the big segments start first:
// the (image, centre) pairs are split into 8 contiguous ranges (one per XCD, so that neighbouring
// segments, which share border lines of the Lab image, meet in one L2) and each range is bucket
// sorted by the pixel count of the previous sweep, largest first.  Persistent waves then pull
// the next segment of their XCD's list with one atomic each.
#define UPD_CLASSES 64

__global__ __launch_bounds__(1024) void k_slic_order(const uint32_t *__restrict__ cen, int total,
                                                     int per_xcd, unsigned mean_px,
                                                     int *__restrict__ order, int *__restrict__ qhead)
{
    __shared__ int hist[UPD_CLASSES];
    const int x = blockIdx.x, tid = threadIdx.x;
    const int lo = x * per_xcd, hi = min(total, lo + per_xcd);
    if (tid < UPD_CLASSES) hist[tid] = 0;
    __syncthreads();
    for (int i = lo + tid; i < hi; i += 1024) {
        const unsigned n = cen[(long long)i * CEN_WORDS + 10];
        atomicAdd(&hist[UPD_CLASSES - 1 - (int)min((unsigned)(UPD_CLASSES - 1), n * 16u / mean_px)], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int q = 0; q < UPD_CLASSES; ++q) { const int h = hist[q]; hist[q] = run; run += h; }
        qhead[x] = 0;
    }
    __syncthreads();
    for (int i = lo + tid; i < hi; i += 1024) {
        const unsigned n = cen[(long long)i * CEN_WORDS + 10];
        const int pos = atomicAdd(&hist[UPD_CLASSES - 1 - (int)min((unsigned)(UPD_CLASSES - 1), n * 16u / mean_px)], 1);
        order[lo + pos] = i;
    }
}

// PWM = occupancy words per 8-row group the image width needs: 4 (<= 2048 pixels, 16-bit list entries)
// or 8 (<= 4096 pixels, 32-bit list entries)
//
// Workgroup = UPD2_NSEG gather waves + ONE chain wave.  A gather wave owns one segment at a time and does
// everything above except the sums: it publishes its rounds of 128 staged pixels (five rows y, x, L, a, b) in a
// two-slot ring of LDS and goes on gathering.  The chain wave carries the running sums of ALL the workgroup's
// segments at once: lanes 5g..5g+4 walk the five rows of gather wave g's next published round, so one dependent
// float32 add instruction advances 60 chains instead of 5 (the chain adds were half of the kernel's vector
// instructions with 5 of 64 lanes useful).  Per segment the order of the additions is unchanged — raster order,
// one chain — hence the same bits.  Waves talk through LDS words only (ready / consumed message counters per
// ring, no workgroup barrier: the gather waves keep four rounds of global loads in flight and a barrier's waitcnt
// would drain them); a ring message is a data round (header = fill), the end of a segment (the chain wave
// divides, derives the next search window and writes the centre record) or the end of the queue.
#ifndef UPD2_NSEG
#define UPD2_NSEG 8            // measured per launch (30 images): 12/1024 0.372 ms, 6/1024 0.372, 10/1536 0.365, 8/2048 0.358,
#endif                       // 6/512 0.400, 12/512 0.394 (gather waves / index-buffer pixels); without the Lab loads 0.290
#define UPD2_D 2                          // ring slots per gather wave
#define UPD2_ROW (UPD_ROUND + 4)          // 132 floats: lane l's row starts 4 l banks on (16-byte reads, 64 banks)
#ifndef UPD2_ICAP
#define UPD2_ICAP 2048
#endif
#define UPD2_MSG_END 0x40000000
#define UPD2_MSG_QUIT 0x20000000
template <int PWM> struct Upd2Lds {
    typedef typename UpdEntry<PWM>::type entry_t;
    static constexpr size_t stage = 0;
    static constexpr size_t idx = stage + (size_t)UPD2_NSEG * UPD2_D * 5 * UPD2_ROW * 4;
    static constexpr size_t plist = idx + (size_t)UPD2_NSEG * UPD2_ICAP * 4;
    static constexpr size_t ctrl = plist + (size_t)UPD2_NSEG * UPD_PCAP * sizeof(entry_t);
    // ctrl words: ready[NSEG] | consumed[NSEG] | hdr[NSEG][D] | segid[NSEG][D]
    static constexpr size_t bytes = ctrl;      // (the control words are a static LDS array of the kernel)
};

__device__ __forceinline__ void upd2_lds_settle()       // this wave's LDS operations have completed; loads in flight stay
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int PWM>
__global__ __launch_bounds__((UPD2_NSEG + 1) * 64)
void k_slic_update2(const float *__restrict__ lab, uint32_t *__restrict__ cen, int nC, int B, int H,
                   int W, int s2y, int s2x, unsigned long long *__restrict__ rowmask, int HG, int PW,
                   unsigned long long *__restrict__ fine, int RW, int PWF, int per_xcd,
                   const int *__restrict__ order, int *__restrict__ qhead,
                   uint32_t *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) char upd2_lds[];
    typedef typename UpdEntry<PWM>::type entry_t;
    typedef Upd2Lds<PWM> LD;
    constexpr int PB = UpdEntry<PWM>::piece_bits;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int xcd = (int)(blockIdx.x & 7u);          // workgroups are dealt round-robin to the XCDs
    const int qlo = xcd * per_xcd, qn = min(B * nC, qlo + per_xcd) - qlo;
    float *stage = (float *)(upd2_lds + LD::stage);
    // control words in a STATIC LDS array, read and written with relaxed workgroup-scope atomics: plain ds_read /
    // ds_write.  (A volatile pointer into the dynamic block loses its address space: the compiler then emits flat
    // loads with s_waitcnt vmcnt(0), which drains the gather waves' Lab loads at every poll — measured: half of
    // the launch.)
    __shared__ int ctrl_s[UPD2_NSEG * (2 + 2 * UPD2_D)];
    int *ready = ctrl_s;
    int *consumed = ready + UPD2_NSEG;
    int *hdr = consumed + UPD2_NSEG;                            // [NSEG][D]
    int *segid = hdr + UPD2_NSEG * UPD2_D;                      // [NSEG][D]
#define UPD2_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define UPD2_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
    if (tid < UPD2_NSEG * 2) ready[tid] = 0;                    // ready | consumed
    __syncthreads();
    if (wv == UPD2_NSEG) {
        // ---------------------------------------------------------------- the chain wave
        const int g = lane / 5, f = lane - g * 5;
        const bool active = g < UPD2_NSEG;
        const int gl = active ? g : 0;
        int r = 0;                 // messages of ring g taken so far
        float acc = 0.0f;          // running sum of feature f (y, x, L, a, b) of ring g's current segment
        unsigned n = 0;            // its pixel count
        bool quit = !active;
        for (;;) {
            const bool avail = !quit && UPD2_LD(ready + gl) > r;
            if (!__ballot(avail)) {
                if (!__ballot(!quit)) break;
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            asm volatile("" ::: "memory");
            const int slot = r & (UPD2_D - 1);
            const int h = avail ? UPD2_LD(hdr + gl * UPD2_D + slot) : 0;
            const bool data = avail && h > 0 && h <= UPD_ROUND;
            if (data) {
                // lane walks feature row f of the round: four 16-byte LDS reads per 16 dependent adds, the reads
                // of the next 16 pixels in flight meanwhile; the gather wave padded the row with zeros up to
                // UPD_ROUND, and x + 0.0f = x exactly, so every round is summed over its whole length
                const float4 *r4 = (const float4 *)(stage + ((gl * UPD2_D + slot) * 5 + f) * UPD2_ROW);
                float4 q0 = r4[0], q1 = r4[1], q2 = r4[2], q3 = r4[3];
                float4 n0, n1, n2, n3;
#define UPD_ADD16(a0, a1, a2, a3)                                                          \
    acc = acc + a0.x; acc = acc + a0.y; acc = acc + a0.z; acc = acc + a0.w;                \
    acc = acc + a1.x; acc = acc + a1.y; acc = acc + a1.z; acc = acc + a1.w;                \
    acc = acc + a2.x; acc = acc + a2.y; acc = acc + a2.z; acc = acc + a2.w;                \
    acc = acc + a3.x; acc = acc + a3.y; acc = acc + a3.z; acc = acc + a3.w;
#define UPD_PIN(a0, a1, a2, a3)                                                            \
    asm volatile("" : "+v"(a0.x), "+v"(a0.y), "+v"(a0.z), "+v"(a0.w), "+v"(a1.x), "+v"(a1.y), "+v"(a1.z), "+v"(a1.w)); \
    asm volatile("" : "+v"(a2.x), "+v"(a2.y), "+v"(a2.z), "+v"(a2.w), "+v"(a3.x), "+v"(a3.y), "+v"(a3.z), "+v"(a3.w));
#pragma unroll
                for (int j = 0; j < UPD_ROUND; j += 32) {
                    n0 = r4[(j >> 2) + 4]; n1 = r4[(j >> 2) + 5]; n2 = r4[(j >> 2) + 6]; n3 = r4[(j >> 2) + 7];
                    __builtin_amdgcn_sched_barrier(0);
                    UPD_ADD16(q0, q1, q2, q3)
                    UPD_PIN(n0, n1, n2, n3)
                    if (j + 32 < UPD_ROUND) {
                        q0 = r4[(j >> 2) + 8]; q1 = r4[(j >> 2) + 9]; q2 = r4[(j >> 2) + 10]; q3 = r4[(j >> 2) + 11];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    UPD_ADD16(n0, n1, n2, n3)
                    if (j + 32 < UPD_ROUND) { UPD_PIN(q0, q1, q2, q3) }
                }
                n += (unsigned)h;
            }
            // ---- end of a segment: mean, next window, centre record (wave-uniform section: the shuffles)
            const bool end = avail && h == UPD2_MSG_END;
            if (__ballot(end)) {
                const float mean = acc / (float)n;      // segments[k, c] /= n_segment_elems[k]
                const int g5 = gl * 5;
                const float cy = __shfl(mean, g5), cx = __shfl(mean, g5 + 1);
                const float cl = __shfl(mean, g5 + 2), ca = __shfl(mean, g5 + 3), cbb = __shfl(mean, g5 + 4);
                if (end && f == 0) {
                    uint32_t *c = cen + (long long)UPD2_LD(segid + gl * UPD2_D + slot) * CEN_WORDS;
                    if (n == 0u) {
                        // the seed lost all its pixels: scikit-image's 0/0 gives it a NaN centre, a NaN distance never
                        // wins `distance > dist_center`, and the NaN -> index casts that size its window yield an empty
                        // range (x86-64), so it stays dead for the remaining sweeps (tests/golden/slic_starve_*.npz).
                        // Here: NaN centre + empty window (never a candidate of any tile, no pixel ever again); the
                        // status bit is informational.
                        atomicOr(status, SPA_ST_SLIC_EMPTY_SEGMENT);
                        const uint32_t qnan = 0x7fc00000u;
                        c[0] = qnan; c[1] = qnan; c[2] = qnan; c[3] = qnan; c[4] = qnan;
                        c[6] = 0u; c[7] = 0u; c[8] = 0u; c[9] = 0u;
                        c[10] = 0u;
                    } else {
                        int ny0, ny1, nx0, nx1;
                        slic_window(cy, cx, s2y, s2x, H, W, ny0, ny1, nx0, nx1);
                        c[0] = __float_as_uint(cy); c[1] = __float_as_uint(cx);
                        c[2] = __float_as_uint(cl); c[3] = __float_as_uint(ca); c[4] = __float_as_uint(cbb);
                        c[6] = (uint32_t)ny0; c[7] = (uint32_t)ny1; c[8] = (uint32_t)nx0; c[9] = (uint32_t)nx1;
                        c[10] = n;
                    }
                }
                if (end) { acc = 0.0f; n = 0u; }
            }
            if (avail && h == UPD2_MSG_QUIT) quit = true;
            if (avail) {
                r += 1;
                upd2_lds_settle();                       // the round has been read
                if (f == 0) UPD2_ST(consumed + gl, r);
            }
        }
        return;
    }
    // ------------------------------------------------------------------- gather waves
    unsigned *idx = (unsigned *)(upd2_lds + LD::idx) + wv * UPD2_ICAP;
    entry_t *plist = (entry_t *)(upd2_lds + LD::plist) + wv * UPD_PCAP;
    const unsigned W4 = (unsigned)W * 4u;
    int msgs = 0;                  // messages this wave has published
    // next free ring slot (waits for the chain wave when both are in use)
    auto ring_slot = [&]() -> int {
        while (UPD2_LD(consumed + wv) + UPD2_D <= msgs) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        return msgs & (UPD2_D - 1);
    };
    auto publish = [&](int slot, int header, int seg_) {
        upd2_lds_settle();
        if (lane == 0) { UPD2_ST(hdr + wv * UPD2_D + slot, header); UPD2_ST(segid + wv * UPD2_D + slot, seg_); }
        upd2_lds_settle();
        msgs += 1;
        if (lane == 0) UPD2_ST(ready + wv, msgs);
    };
  for (;;) {
    int qi = 0;
    if (lane == 0) qi = atomicAdd(qhead + xcd, 1);
    qi = __builtin_amdgcn_readfirstlane(qi);
    if (qi >= qn) {
        publish(ring_slot(), UPD2_MSG_QUIT, 0);
        return;
    }
    const int seg = __builtin_amdgcn_readfirstlane(order[qlo + qi]);
    const int b = seg / nC;
    const int k = seg - b * nC;
    const unsigned npix = (unsigned)H * (unsigned)W;
    const float *pL = lab + (long long)b * 3 * npix;        // wave-uniform plane bases
    const float *pA = pL + npix;
    const float *pB = pA + npix;
    uint32_t *c = cen + ((long long)b * nC + k) * CEN_WORDS;
    const int wy0 = (int)c[6], wy1 = (int)c[7], wx0 = (int)c[8];   // the window the sweep used
    unsigned long long *rm = rowmask + ((long long)b * nC + k) * HG * PW;
    unsigned long long *fm = fine + ((long long)b * nC + k) * RW * PWF;
    const int pc0 = wx0 >> 6;

    UPD_T0();

    const int g0 = wy0 >> 3, g1 = (wy1 - 1) >> 3;
    const int ybase = g0 << 3;                    // plist rows are relative to this
    int gb = g0;
    while (gb <= g1) {
        // ---- piece list in raster order: groups in order, rows in order, pieces in order.
        // lane g expands group gb+g: PW mask words, byte q of word i = rows of piece 8i+q.
        // As many whole groups as fit the list are taken (and their masks cleared).
        int npieces, gtake;
        {
            const int gg = gb + lane;
            const bool has = gg <= g1;
            unsigned long long mw[PWM];
#pragma unroll
            for (int i = 0; i < PWM; ++i) mw[i] = 0ull;
            int cntl = 0;
            unsigned long long *w = rm + (long long)gg * PW;
            if (has) {
#pragma unroll
                for (int i = 0; i < PWM; ++i)
                    if (i < PW) { mw[i] = w[i]; cntl += __popcll(mw[i]); }
            }
            const int inc = upd_wave_scan(cntl);
            const bool take = has && inc <= UPD_PCAP;
            const unsigned long long tm = __ballot(take);
            gtake = __popcll(tm);                         // prefix property: lanes 0..gtake-1
            npieces = __builtin_amdgcn_readfirstlane(gtake ? __shfl(inc, gtake - 1) : 0);
            if (take && cntl) {
#pragma unroll
                for (int i = 0; i < PWM; ++i)
                    if (i < PW && mw[i]) w[i] = 0ull;                             // consumed
                int pos = inc - cntl;
                const int yrel = (gg << 3) - ybase;
#pragma unroll 1
                for (int r = 0; r < 8; ++r) {
#pragma unroll
                    for (int i = 0; i < PWM; ++i) {
                        unsigned long long bits = (mw[i] >> r) & 0x0101010101010101ull;
                        while (bits) {
                            const int q = (__ffsll((long long)bits) - 1) >> 3;
                            plist[pos++] = (entry_t)(((unsigned)(yrel + r) << PB) | (unsigned)(i * 8 + q));
                            bits &= bits - 1ull;
                        }
                    }
                }
            }
            if (gtake == 0) gtake = 1;
        }
        gb += gtake;
        wave_lds_sync();
        UPD_T(0)

        if (npieces == 0) continue;
        // fine masks of list entries e..e+63, one per lane (lanes past the end re-read the last
        // entry: the load itself is unconditional so that the wait counters stay exact)
        unsigned long long raw_n;
        unsigned long long *fa_n;
        unsigned code_n;
        auto fetch_masks = [&](int e) {
            const unsigned pe = (unsigned)plist[min(e + lane, npieces - 1)];
            const int yrel = (int)(pe >> PB), pc = (int)(pe & ((1u << PB) - 1u));
            fa_n = fm + (long long)(ybase + yrel - wy0) * PWF + (pc - pc0);
            raw_n = *fa_n;
            code_n = ((unsigned)yrel << UPD_XBITS) | ((unsigned)pc << 6);
        };
        // ---- the stream of this list: the index buffer is refilled when its last round has been
        // ISSUED (issued rounds carry their codes in registers), so rounds of the next part are in
        // flight while the last rounds of this part are summed
        int e0 = 0;              // next list entry
        int T = 0, nr = 0;       // pixels / rounds in the index buffer
        int lr = 0;              // next round of the buffer to issue
        int inflight = 0;        // live rounds issued and not yet summed
        if (lane == 0) idx[0] = 0u;                  // a valid code for the loads of idle rounds
        fetch_masks(0);
        auto refill = [&]() {
            // lane j owns list entry e0+j: fine mask in pixel order, place in the stream
            unsigned mlo = 0u, mhi = 0u;
            unsigned code0 = code_n;
            unsigned long long *fa = fa_n;
            if (e0 + lane < npieces) {
                mlo = upd_pixel_order((unsigned)raw_n);
                mhi = upd_pixel_order((unsigned)(raw_n >> 32));
            }
            const int cnt = __popc(mlo) + __popc(mhi);
            const int inc = upd_wave_scan(cnt);
            const bool take = (e0 + lane < npieces) && inc <= UPD2_ICAP;
            const int m = __popcll(__ballot(take));               // >= 1: one entry is at most 64 pixels
            T = __builtin_amdgcn_readlane(inc, m - 1);
            if (take) {
                *fa = 0ull;                                       // cleared for the next sweep
                // set bits -> packed (row, x), in x order: two 32-bit halves (32-bit bit tricks are
                // half the vector instructions of 64-bit ones); the loop runs as long as the fullest
                // mask of the wave needs
                unsigned *d = idx + (inc - cnt);
                while (mlo) {
                    *d++ = code0 + (unsigned)(__ffs((int)mlo) - 1);
                    mlo &= mlo - 1u;
                }
                code0 += 32u;
                while (mhi) {
                    *d++ = code0 + (unsigned)(__ffs((int)mhi) - 1);
                    mhi &= mhi - 1u;
                }
            }
            e0 += m;
            nr = (T + UPD_ROUND - 1) / UPD_ROUND;
            lr = 0;
            wave_lds_sync();
            fetch_masks(e0);     // the next entries' masks travel while this part of the stream is summed
            UPD_T(1)
        };
        auto issue = [&](UpdRound &R) {
            if (lr == nr && e0 < npieces) refill();
            const bool live = lr < nr;
            R.fill = live ? min(UPD_ROUND, T - lr * UPD_ROUND) : 0;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int p = g * 64 + lane;
                // lanes past the end re-read pixel 0 of the buffer (cached) and stage zeros
                const bool ok = p < R.fill;
                const unsigned v = idx[ok ? lr * UPD_ROUND + p : 0];
                const unsigned off = (unsigned)(ybase + (int)(v >> UPD_XBITS)) * W4 + ((v & ((1u << UPD_XBITS) - 1u)) << 2);
                R.code[g] = ok ? v : 0xFFFFFFFFu;
#ifdef SPA_UPD_NOLOAD          // experiment: how much of the launch is the Lab gather itself
                R.vL[g] = __uint_as_float(off); R.vA[g] = 1.0f; R.vB[g] = 2.0f;
#else
                R.vL[g] = *(const float *)((const char *)pL + off);
                R.vA[g] = *(const float *)((const char *)pA + off);
                R.vB[g] = *(const float *)((const char *)pB + off);
#endif
            }
            lr += live ? 1 : 0;
            inflight += live ? 1 : 0;
        };
        auto consume = [&](const UpdRound &R) {
            const int fill = R.fill;
            if (fill == 0) return;                         // idle round (its loads hit one cached line)
            UPD_T(5)
            const int slot = ring_slot();
            UPD_T(3)
            float *st = stage + (wv * UPD2_D + slot) * 5 * UPD2_ROW;
            if (fill == UPD_ROUND) {                       // full round: every lane is a pixel
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float *d = st + g * 64 + lane;
                    d[0 * UPD2_ROW] = (float)(ybase + (int)(R.code[g] >> UPD_XBITS));
                    d[1 * UPD2_ROW] = (float)(int)(R.code[g] & ((1u << UPD_XBITS) - 1u));
                    d[2 * UPD2_ROW] = R.vL[g];
                    d[3 * UPD2_ROW] = R.vA[g];
                    d[4 * UPD2_ROW] = R.vB[g];
                }
            } else {                                       // last round of a part: zeros behind the end
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const bool ok = R.code[g] != 0xFFFFFFFFu;
                    float *d = st + g * 64 + lane;
                    d[0 * UPD2_ROW] = ok ? (float)(ybase + (int)(R.code[g] >> UPD_XBITS)) : 0.0f;
                    d[1 * UPD2_ROW] = ok ? (float)(int)(R.code[g] & ((1u << UPD_XBITS) - 1u)) : 0.0f;
                    d[2 * UPD2_ROW] = ok ? R.vL[g] : 0.0f;
                    d[3 * UPD2_ROW] = ok ? R.vA[g] : 0.0f;
                    d[4 * UPD2_ROW] = ok ? R.vB[g] : 0.0f;
                }
            }
            publish(slot, fill, seg);
            UPD_T(2)
            inflight -= 1;
        };

        // straight-line schedule, no branch around the loads of an issue(): the compiler's vmcnt
        // accounting stays exact (waits leave the younger loads in flight) only when the number
        // of loads between an issue and its use is the same on every path.
        // Five register sets, four rounds (24 loads per lane) in flight.  (With two in flight the launch was 3 %
        // slower: the phase is not load-latency bound.  PMC per launch: 2.0 VALU wave-instructions per pixel — 1.0
        // of them the chain adds — = 54 % of the SIMDs' issue time; waves: 39 % issuing, 26 % issue-stalled,
        // 35 % waiting.  What remains is the instruction count, i.e. several segments' chains per wave.)
        UpdRound r0, r1, r2, r3, r4;
        issue(r0);
        issue(r1);
        issue(r2);
        issue(r3);
        do {
            issue(r4);
            consume(r0);
            issue(r0);
            consume(r1);
            issue(r1);
            consume(r2);
            issue(r2);
            consume(r3);
            issue(r3);
            consume(r4);
        } while (inflight > 0 || lr < nr || e0 < npieces);
        UPD_T(4)
    }
    UPD_TFLUSH(qhead)
    publish(ring_slot(), UPD2_MSG_END, seg);           // the chain wave closes the segment
  }
}

__global__ void k_slic_export_centres(const uint32_t *__restrict__ cen, float *__restrict__ out,
                                      long long total)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t *c = cen + i * CEN_WORDS;
    float *o = out + i * 6;
    o[0] = (c[0] == 0x7fc00000u && c[7] == 0u) ? __uint_as_float(0x7fc00000u) : 0.0f;   // dead seed: z is 0/0 too
    o[1] = __uint_as_float(c[0]); o[2] = __uint_as_float(c[1]);
    o[3] = __uint_as_float(c[2]); o[4] = __uint_as_float(c[3]); o[5] = __uint_as_float(c[4]);
}

extern "C" int spa_slic_core(spa_ctx *ctx, const float *lab, int32_t B, int32_t H, int32_t W,
                             int32_t n_segments, int32_t max_iter, int32_t *labels,
                             float *centres, void *stream)
{
    SPA_ARG(ctx && lab && labels && B > 0 && max_iter > 0);
    spa_slic_plan pl;
    int rc = spa_slic_make_plan(H, W, n_segments, &pl);
    if (rc != SPA_OK) return rc;
    const int nC = pl.n_centroids;
    hipStream_t s = spa_stream(stream);
    uint32_t *cen;
    rc = spa_ws_reserve(ctx, WS_CENTRES, (size_t)B * nC * CEN_WORDS * 4, (void **)&cen);
    if (rc != SPA_OK) return rc;
    const int s2y = 2 * pl.win_step_y, s2x = 2 * pl.win_step_x;
    // occupancy words hold <= 64 pieces of 64 pixels per row, stream codes keep the row in the upper 20 bits, the
    // centre table is read by 32-bit offsets: anything else takes the general float32 kernels (spa_slic64.hip:
    // same arithmetic and order, untuned), as does SPA_SLIC_GENERAL=1 (tests compare the two paths)
    if (W > 64 * 8 * UPD_PWMAX || 4 * pl.win_step_y + 24 >= (1 << 19) || ctx->slic_force_general)
        return spa_slic_core_general_f32(ctx, lab, B, H, W, n_segments, max_iter, labels, centres, stream);
    const int HG = (H + 7) / 8;
    const int PW = (((W + 63) / 64) + 7) / 8;      // mask words per (centre, 8-row group)
    unsigned long long *rowmask;
    rc = spa_ws_reserve(ctx, WS_ROWMASK, (size_t)B * nC * HG * PW * 8, (void **)&rowmask);
    if (rc != SPA_OK) return rc;
    SPA_HIP(hipMemsetAsync(rowmask, 0, (size_t)B * nC * HG * PW * 8, s));
    // fine pixel masks: per (centre, row of its search window, 64-pixel piece of the window) 64 bits
    const int RW = 2 * s2y + 2;
    const int PWF = ((2 * s2x + 1) >> 6) + 2;
    unsigned long long *fine;
    rc = spa_ws_reserve(ctx, WS_FINEMASK, (size_t)B * nC * RW * PWF * 8, (void **)&fine);
    if (rc != SPA_OK) return rc;
    SPA_HIP(hipMemsetAsync(fine, 0, (size_t)B * nC * RW * PWF * 8, s));
    // update sweep: per-XCD longest-first segment lists pulled by persistent waves
    const int upd_total = B * nC;
    const int upd_per_xcd = (upd_total + 7) / 8;
    int *upd_order;
    rc = spa_ws_reserve(ctx, WS_SLIC_ORDER, ((size_t)upd_total + 8 + 32) * 4, (void **)&upd_order);
    if (rc != SPA_OK) return rc;
    int *upd_qhead = upd_order + upd_total;
    if (!ctx->upd_wg_per_cu) {         // per context = per device
        SPA_HIP(hipFuncSetAttribute((const void *)k_slic_update2<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Upd2Lds<4>::bytes));
        SPA_HIP(hipFuncSetAttribute((const void *)k_slic_update2<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Upd2Lds<8>::bytes));
        SPA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&ctx->upd_wg_per_cu, (const void *)k_slic_update2<4>,
                                                             (UPD2_NSEG + 1) * 64, Upd2Lds<4>::bytes));
        SPA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&ctx->upd_wg_per_cu8, (const void *)k_slic_update2<8>,
                                                             (UPD2_NSEG + 1) * 64, Upd2Lds<8>::bytes));
        if (ctx->upd_wg_per_cu < 1) ctx->upd_wg_per_cu = 1;
        if (ctx->upd_wg_per_cu8 < 1) ctx->upd_wg_per_cu8 = 1;
    }
    const bool upd_narrow = PW <= 4 && 4 * pl.win_step_y + 24 < 2048;
    int upd_grid = (upd_narrow ? ctx->upd_wg_per_cu : ctx->upd_wg_per_cu8) * ctx->n_cu;
    if (upd_grid > (upd_total + UPD2_NSEG - 1) / UPD2_NSEG) upd_grid = (upd_total + UPD2_NSEG - 1) / UPD2_NSEG;
    upd_grid = (upd_grid + 7) & ~7;
    const unsigned mean_px = (unsigned)(((long long)H * W) / nC) ? (unsigned)(((long long)H * W) / nC) : 1u;
    hipLaunchKernelGGL(k_slic_init, dim3((nC + 127) / 128, B), dim3(128), 0, s, cen, nC,
                       pl.grid_nx, pl.start_y, pl.start_x, pl.step_y, pl.step_x, s2y, s2x, H, W);
    SPA_LAUNCH_CHECK();
    // cdef floating spatial_weight = 1.0 / (step * step)
    const float sw = (float)(1.0 / (double)(pl.step * pl.step));
    dim3 ga((W + TILE - 1) / TILE, (H + TILE - 1) / TILE, B);
    // development aid (tools/race_probe7.py): SPA_SLIC_STOP = n stops after the n-th launch of the sweep loop (assign, update, ...)
    const int stop_after = getenv("SPA_SLIC_STOP") ? atoi(getenv("SPA_SLIC_STOP")) : 1 << 30;
    int launched = 0;
    for (int it = 0; it < max_iter; ++it) {
        if (launched >= stop_after) break;
        ++launched;
        { SpaProfScope prof_(ctx, PROF_SLIC_ASSIGN, s);
        // (the masks of the last sweep would never be read)
        const bool upd = it + 1 < max_iter || centres;
#ifdef SPA_DIAG
        if (ctx->dbg_slic_ldsx)
            hipLaunchKernelGGL(k_slic_assign<true>, ga, dim3(256), 0, s, lab, cen, nC, H, W, sw, labels,
                               rowmask, HG, PW, upd ? (uint32_t *)fine : (uint32_t *)nullptr, RW, PWF, ctx->d_status);
        else
#endif
        hipLaunchKernelGGL(k_slic_assign<false>, ga, dim3(256), 0, s, lab, cen, nC, H, W, sw, labels,
                           rowmask, HG, PW, upd ? (uint32_t *)fine : (uint32_t *)nullptr, RW, PWF,
                           ctx->d_status); }
        SPA_LAUNCH_CHECK();
        // the centroids computed after the last sweep never influence the labels
        if (launched >= stop_after) break;
        if (it + 1 < max_iter || centres) {
            ++launched;
            SpaProfScope prof_(ctx, PROF_SLIC_UPDATE, s);
            hipLaunchKernelGGL(k_slic_order, dim3(8), dim3(1024), 0, s, cen, upd_total, upd_per_xcd, mean_px,
                               upd_order, upd_qhead);
            if (upd_narrow)
                hipLaunchKernelGGL(k_slic_update2<4>, dim3(upd_grid), dim3((UPD2_NSEG + 1) * 64), Upd2Lds<4>::bytes, s, lab, cen, nC, B,
                                   H, W, s2y, s2x, rowmask, HG, PW, fine, RW, PWF, upd_per_xcd,
                                   (const int *)upd_order, upd_qhead, ctx->d_status);
            else
                hipLaunchKernelGGL(k_slic_update2<8>, dim3(upd_grid), dim3((UPD2_NSEG + 1) * 64), Upd2Lds<8>::bytes, s, lab, cen, nC, B,
                                   H, W, s2y, s2x, rowmask, HG, PW, fine, RW, PWF, upd_per_xcd,
                                   (const int *)upd_order, upd_qhead, ctx->d_status);
            SPA_LAUNCH_CHECK();
        }
    }
    if (centres) {
        long long total = (long long)B * nC;
        hipLaunchKernelGGL(k_slic_export_centres, dim3((unsigned)((total + 255) / 256)), dim3(256),
                           0, s, cen, centres, total);
        SPA_LAUNCH_CHECK();
    }
    return SPA_OK;
}

extern "C" int spa_slic(spa_ctx *ctx, const float *rgb, int32_t B, int32_t H, int32_t W,
                        int32_t n_segments, float compactness, int32_t max_iter,
                        int32_t *labels, int32_t *n_labels, void *stream)
{
    SPA_ARG(ctx && rgb && labels && n_labels && compactness > 0.0f);
    spa_slic_plan pl;
    int rc = spa_slic_make_plan(H, W, n_segments, &pl);
    if (rc != SPA_OK) return rc;
    float *lab;
    int32_t *pre;
    size_t npix = (size_t)H * W;
    rc = spa_ws_reserve(ctx, WS_LAB, (size_t)B * 3 * npix * 4, (void **)&lab);
    if (rc != SPA_OK) return rc;
    rc = spa_ws_reserve(ctx, WS_PRE, (size_t)B * npix * 4, (void **)&pre);
    if (rc != SPA_OK) return rc;
    // ratio = 1.0 / compactness; image * ratio in float32 (slic_superpixels.py:303-305)
    float ratio = (float)(1.0 / (double)compactness);
    rc = spa_rgb2lab(ctx, rgb, B, H, W, ratio, lab, stream);
    if (rc != SPA_OK) return rc;
    rc = spa_slic_core(ctx, lab, B, H, W, n_segments, max_iter, pre, nullptr, stream);
    if (rc != SPA_OK) return rc;
    return spa_enforce_connectivity(ctx, pre, B, H, W, pl.min_size, pl.max_size, labels, n_labels,
                                    stream);
}
