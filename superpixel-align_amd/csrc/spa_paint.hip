// Painting the cluster ids back onto the pixels (weighted_kmeans paint loop,
// batch_spalign_kmeans.py:193-199, and `clustering_result == 0`, :207) and the per-image
// confusion counts of save_info (:398-405).  Pure streaming kernels: 4 B read + 2 B written
// per pixel (paint), 5 B read per pixel (confusion).
#include "spa_common.h"

__global__ __launch_bounds__(256) void k_paint(const int32_t *__restrict__ labels,
                                               const int32_t *__restrict__ assign,
                                               const int32_t *__restrict__ offsets, int npix,
                                               uint8_t *__restrict__ cluster,
                                               uint8_t *__restrict__ road,
                                               uint32_t *__restrict__ status)
{
    const int b = blockIdx.y;
    const int off = offsets[b], S = offsets[b + 1] - off;
    const long long o = (long long)b * npix;
    const int n4 = npix >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        int4 l = ((const int4 *)(labels + o))[i];
        int v[4] = {l.x, l.y, l.z, l.w};
        uchar4 c, r;
        unsigned char cc[4], rr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int a = 0;
            if (v[j] < 0 || v[j] >= S) atomicOr(status, SPA_ST_LABEL_RANGE);
            else a = assign[off + v[j]];
            cc[j] = (unsigned char)a; rr[j] = (a == 0) ? 1 : 0;
        }
        c = make_uchar4(cc[0], cc[1], cc[2], cc[3]);
        r = make_uchar4(rr[0], rr[1], rr[2], rr[3]);
        ((uchar4 *)(cluster + o))[i] = c;
        ((uchar4 *)(road + o))[i] = r;
    }
    // tail (npix not a multiple of 4)
    if (blockIdx.x == 0) {
        for (int p = (n4 << 2) + threadIdx.x; p < npix; p += 256) {
            int l = labels[o + p];
            int a = (l < 0 || l >= S) ? 0 : assign[off + l];
            cluster[o + p] = (unsigned char)a; road[o + p] = (a == 0) ? 1 : 0;
        }
    }
}

extern "C" int spa_paint(spa_ctx *ctx, const int32_t *labels, const int32_t *assign,
                         const int32_t *offsets, int32_t B, int32_t H, int32_t W, uint8_t *cluster,
                         uint8_t *road, void *stream)
{
    SPA_ARG(ctx && labels && assign && offsets && cluster && road && B > 0);
    const int npix = H * W;
    SPA_ARG((((uintptr_t)labels | (uintptr_t)cluster | (uintptr_t)road) & 15) == 0 || npix < 4);
    SPA_ARG(npix % 4 == 0 || B == 1);   // per-image base must stay 4-pixel aligned
    int gx = (npix / 4 + 255) / 256;
    if (gx > 1024) gx = 1024;
    if (gx < 1) gx = 1;
    SpaProfScope prof_(ctx, PROF_PAINT, spa_stream(stream));
    hipLaunchKernelGGL(k_paint, dim3(gx, B), dim3(256), 0, spa_stream(stream), labels, assign,
                       offsets, npix, cluster, road, ctx->d_status);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// confusion[gt, pred] with gt < 0 ignored: out = {TN, FP, FN, TP}
__global__ __launch_bounds__(256) void k_confusion(const uint8_t *__restrict__ road,
                                                   const int32_t *__restrict__ gt, long long npix,
                                                   unsigned long long *__restrict__ out)
{
    __shared__ unsigned s[4][4];
    const int b = blockIdx.y;
    const uint8_t *r = road + (long long)b * npix;
    const int32_t *g = gt + (long long)b * npix;
    unsigned c[4] = {0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        int gv = g[i];
        if (gv < 0) continue;
        c[(gv ? 2 : 0) + (r[i] ? 1 : 0)] += 1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        for (int o = 32; o > 0; o >>= 1) c[j] += __shfl_down(c[j], o);
    if ((threadIdx.x & 63) == 0)
        for (int j = 0; j < 4; ++j) s[threadIdx.x >> 6][j] = c[j];
    __syncthreads();
    if (threadIdx.x < 4) {
        unsigned t = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
        if (t) atomicAdd(out + (long long)b * 4 + threadIdx.x, (unsigned long long)t);
    }
}

extern "C" int spa_confusion(spa_ctx *ctx, const uint8_t *road, const int32_t *gt, int32_t B,
                             int64_t npix, int64_t *out, void *stream)
{
    SPA_ARG(ctx && road && gt && out && B > 0 && npix > 0);
    hipStream_t s = spa_stream(stream);
    SPA_HIP(hipMemsetAsync(out, 0, (size_t)B * 4 * sizeof(int64_t), s));
    int gx = (int)((npix + 255) / 256);
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(k_confusion, dim3(gx, B), dim3(256), 0, s, road, gt, (long long)npix,
                       (unsigned long long *)out);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}


// ------------------------------------------------------------------------------------------
// superpixel_overlaps.py:353-361 — a superpixel becomes road when it holds more than `threshold`
// of all predicted road pixels of its image:  overlap / float(n_road) > threshold  (float64).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_overlap_count(const int32_t *__restrict__ labels,
                                                       const uint8_t *__restrict__ road, long long npix,
                                                       int max_labels, int *__restrict__ counts,
                                                       int *__restrict__ n_road, uint32_t *__restrict__ status)
{
    const int b = blockIdx.y;
    const int32_t *L = labels + (long long)b * npix;
    const uint8_t *R = road + (long long)b * npix;
    int *C = counts + (long long)b * max_labels;
    int mine = 0;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        if (R[p]) {
            const int l = L[p];
            if (l < 0 || l >= max_labels) { atomicOr(status, SPA_ST_LABEL_RANGE); continue; }
            atomicAdd(C + l, 1);
            ++mine;
        }
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_road + b, mine);
}

__global__ __launch_bounds__(256) void k_overlap_paint(const int32_t *__restrict__ labels, long long npix,
                                                       int max_labels, const int *__restrict__ counts,
                                                       const int *__restrict__ n_road, double threshold,
                                                       uint8_t *__restrict__ refined)
{
    const int b = blockIdx.y;
    const int32_t *L = labels + (long long)b * npix;
    const int *C = counts + (long long)b * max_labels;
    const int nr = n_road[b];
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        const int l = L[p];
        uint8_t v = 0;
        if (nr > 0 && l >= 0 && l < max_labels) v = ((double)C[l] / (double)nr > threshold) ? 1 : 0;
        refined[(long long)b * npix + p] = v;
    }
}

extern "C" int spa_overlap_refine(spa_ctx *ctx, const int32_t *labels, const uint8_t *road, int32_t B,
                                  int64_t npix, int32_t max_labels, double threshold, uint8_t *refined,
                                  void *stream)
{
    SPA_ARG(ctx && labels && road && refined && B > 0 && npix > 0 && max_labels > 0);
    hipStream_t s = spa_stream(stream);
    int *counts;
    const size_t bytes = ((size_t)B * max_labels + B) * sizeof(int);
    int rc = spa_ws_reserve(ctx, WS_OVERLAP, bytes, (void **)&counts);
    if (rc != SPA_OK) return rc;
    int *n_road = counts + (size_t)B * max_labels;
    SPA_HIP(hipMemsetAsync(counts, 0, bytes, s));
    int gx = (int)((npix + 255) / 256);
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(k_overlap_count, dim3(gx, B), dim3(256), 0, s, labels, road, (long long)npix, max_labels,
                       counts, n_road, ctx->d_status);
    hipLaunchKernelGGL(k_overlap_paint, dim3(gx, B), dim3(256), 0, s, labels, (long long)npix, max_labels,
                       (const int *)counts, (const int *)n_road, threshold, refined);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
