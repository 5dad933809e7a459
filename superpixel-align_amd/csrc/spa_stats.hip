// Per-superpixel statistics: offsets, pixel counts, bounding boxes, centre of mass
// (scipy.ndimage.center_of_mass, batch_spalign_kmeans.py:229), the location prior
// (create_prior, :111-129) and rank -> pixel selection for the anchors (:230-234).
#include "spa_common.h"

__global__ void k_offsets(const int32_t *__restrict__ n_labels, int B, int32_t *__restrict__ offsets)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int run = 0;
        for (int b = 0; b < B; ++b) { offsets[b] = run; run += n_labels[b]; }
        offsets[B] = run;
    }
}

extern "C" int spa_segment_offsets(spa_ctx *ctx, const int32_t *n_labels, int32_t B,
                                   int32_t *offsets, void *stream)
{
    SPA_ARG(ctx && n_labels && offsets && B > 0);
    hipLaunchKernelGGL(k_offsets, dim3(1), dim3(64), 0, spa_stream(stream), n_labels, B, offsets);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// bbox layout: (Ncap, 4) int32 {y0, y1 (inclusive), x0, x1 (inclusive)}
__global__ void k_bbox_init(int32_t *__restrict__ bbox, int32_t *__restrict__ count, int Ncap)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ncap) return;
    bbox[i * 4 + 0] = 0x7fffffff; bbox[i * 4 + 1] = -1;
    bbox[i * 4 + 2] = 0x7fffffff; bbox[i * 4 + 3] = -1;
    count[i] = 0;
}

// pixel-major pass: counts and bounding boxes, one atomic set per distinct label per wave
__global__ __launch_bounds__(256) void k_bbox_count(const int32_t *__restrict__ labels, int W,
                                                    int npix, const int32_t *__restrict__ offsets,
                                                    int Ncap, int32_t *__restrict__ bbox,
                                                    int32_t *__restrict__ count,
                                                    uint32_t *__restrict__ status)
{
    const int b = blockIdx.y;
    const int off = offsets[b], S = offsets[b + 1] - off;
    const int lane = threadIdx.x & 63;
    for (int p0 = blockIdx.x * 256; p0 < npix; p0 += gridDim.x * 256) {
        const int p = p0 + threadIdx.x;
        int l = -1, y = 0, x = 0;
        if (p < npix) {
            l = labels[(long long)b * npix + p];
            y = p / W; x = p - y * W;
            if (l < 0 || l >= S || off + l >= Ncap) { atomicOr(status, SPA_ST_LABEL_RANGE); l = -1; }
        }
        unsigned long long todo = __ballot(l >= 0);
        while (todo) {
            int leader = __ffsll((long long)todo) - 1;
            int ll = __shfl(l, leader);
            unsigned long long same = __ballot(l == ll);
            int first = leader, last = 63 - __clzll((long long)same);
            int yf = __shfl(y, first), yl = __shfl(y, last);
            // lanes are consecutive pixels: y is monotone; x is monotone inside one row
            int xmin = x, xmax = x;
            if (yf != yl) {
                bool mine = (l == ll);
                int a = mine ? x : 0x7fffffff, c = mine ? x : -1;
                for (int o = 32; o > 0; o >>= 1) { a = min(a, __shfl_xor(a, o)); c = max(c, __shfl_xor(c, o)); }
                xmin = a; xmax = c;
            } else {
                xmin = __shfl(x, first); xmax = __shfl(x, last);
            }
            if (lane == leader) {
                int g = off + ll;
                atomicAdd(count + g, __popcll(same));
                atomicMin(bbox + g * 4 + 0, yf); atomicMax(bbox + g * 4 + 1, yl);
                atomicMin(bbox + g * 4 + 2, xmin); atomicMax(bbox + g * 4 + 3, xmax);
            }
            todo &= ~same;
        }
    }
}

// same, aggregated per workgroup in LDS first (labels per image <= BBOX_LDS_MAX): a workgroup
// walks a strip of rows, its waves merge their distinct labels into LDS tables with LDS
// atomics, and only the touched entries go to global memory: ~100x fewer global atomics.
#define BBOX_LDS_MAX 1024
__global__ __launch_bounds__(256) void k_bbox_count_lds(const int32_t *__restrict__ labels, int W,
                                                        int H, int rows_per_block,
                                                        const int32_t *__restrict__ offsets,
                                                        int Ncap, int32_t *__restrict__ bbox,
                                                        int32_t *__restrict__ count,
                                                        uint32_t *__restrict__ status)
{
    __shared__ int l_cnt[BBOX_LDS_MAX], l_y0[BBOX_LDS_MAX], l_y1[BBOX_LDS_MAX], l_x0[BBOX_LDS_MAX], l_x1[BBOX_LDS_MAX];
    const int b = blockIdx.y;
    const int off = offsets[b], S = offsets[b + 1] - off;
    const int lane = threadIdx.x & 63;
    const int SL = min(S, BBOX_LDS_MAX);          // labels beyond the LDS tables go straight to global
    for (int l = threadIdx.x; l < SL; l += 256) {
        l_cnt[l] = 0; l_y0[l] = 0x7fffffff; l_y1[l] = -1; l_x0[l] = 0x7fffffff; l_x1[l] = -1;
    }
    __syncthreads();
    const int r0 = blockIdx.x * rows_per_block, r1 = min(H, r0 + rows_per_block);
    const int p_lo = r0 * W, p_hi = r1 * W;
    const int32_t *L = labels + (long long)b * H * W;
    for (int p0 = p_lo; p0 < p_hi; p0 += 256) {
        const int p = p0 + threadIdx.x;
        int l = -1, y = 0, x = 0;
        if (p < p_hi) {
            l = L[p];
            y = p / W; x = p - y * W;
            if (l < 0 || l >= S || off + l >= Ncap) { atomicOr(status, SPA_ST_LABEL_RANGE); l = -1; }
        }
        unsigned long long todo = __ballot(l >= 0);
        while (todo) {
            int leader = __ffsll((long long)todo) - 1;
            int ll = __shfl(l, leader);
            unsigned long long same = __ballot(l == ll);
            int last = 63 - __clzll((long long)same);
            int yf = __shfl(y, leader), yl = __shfl(y, last);
            int xmin, xmax;
            if (yf != yl) {
                bool mine = (l == ll);
                int a = mine ? x : 0x7fffffff, c = mine ? x : -1;
                for (int o = 32; o > 0; o >>= 1) { a = min(a, __shfl_xor(a, o)); c = max(c, __shfl_xor(c, o)); }
                xmin = a; xmax = c;
            } else {
                xmin = __shfl(x, leader); xmax = __shfl(x, last);
            }
            if (lane == leader) {
                if (ll < BBOX_LDS_MAX) {
                    atomicAdd(&l_cnt[ll], __popcll(same));
                    atomicMin(&l_y0[ll], yf); atomicMax(&l_y1[ll], yl);
                    atomicMin(&l_x0[ll], xmin); atomicMax(&l_x1[ll], xmax);
                } else {
                    const int g = off + ll;
                    atomicAdd(count + g, __popcll(same));
                    atomicMin(bbox + g * 4 + 0, yf); atomicMax(bbox + g * 4 + 1, yl);
                    atomicMin(bbox + g * 4 + 2, xmin); atomicMax(bbox + g * 4 + 3, xmax);
                }
            }
            todo &= ~same;
        }
    }
    __syncthreads();
    for (int l = threadIdx.x; l < SL; l += 256) {
        if (l_cnt[l] > 0) {
            const int g = off + l;
            atomicAdd(count + g, l_cnt[l]);
            atomicMin(bbox + g * 4 + 0, l_y0[l]); atomicMax(bbox + g * 4 + 1, l_y1[l]);
            atomicMin(bbox + g * 4 + 2, l_x0[l]); atomicMax(bbox + g * 4 + 3, l_x1[l]);
        }
    }
}

__device__ __forceinline__ int seg_image(const int32_t *offsets, int B, int g)
{
    int b = 0;
    while (b + 1 < B && offsets[b + 1] <= g) ++b;
    return b;
}

// segment-major pass over the bounding box: exact integer coordinate sums (-> centre of
// mass) and the mean of the Gaussian location prior, summed in a fixed order.
__global__ __launch_bounds__(256) void k_seg_moments(const int32_t *__restrict__ labels, int B,
                                                     int H, int W,
                                                     const int32_t *__restrict__ offsets,
                                                     const int32_t *__restrict__ bbox,
                                                     const int32_t *__restrict__ count,
                                                     double ymean, double xmean, double dy2,
                                                     double dx2, double *__restrict__ centroid,
                                                     double *__restrict__ prior)
{
    __shared__ unsigned long long s_sy[4], s_sx[4];
    __shared__ double s_pw[4];
    const int g = blockIdx.x;
    if (g >= offsets[B]) return;
    const int b = seg_image(offsets, B, g);
    const int s = g - offsets[b];
    const int y0 = bbox[g * 4 + 0], y1 = bbox[g * 4 + 1], x0 = bbox[g * 4 + 2], x1 = bbox[g * 4 + 3];
    const int n = count[g];
    unsigned long long sy = 0, sx = 0;
    double pw = 0.0;
    if (n > 0) {
        const int32_t *L = labels + (long long)b * H * W;
        // wave w takes rows y0+w, y0+w+4, ...; lanes sweep the row in 64-pixel steps
        for (int yy = y0 + (threadIdx.x >> 6); yy <= y1; yy += 4) {
            const double ty = ((double)yy - ymean) * ((double)yy - ymean) / dy2;
            for (int xx = x0 + (threadIdx.x & 63); xx <= x1; xx += 64) {
                if (L[(long long)yy * W + xx] == s) {
                    sy += (unsigned)yy; sx += (unsigned)xx;
                    if (prior) {
                        double tx = ((double)xx - xmean) * ((double)xx - xmean) / dx2;
                        pw += spa_det_exp(-(ty + tx));
                    }
                }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        sy += __shfl_down(sy, o); sx += __shfl_down(sx, o); pw += __shfl_down(pw, o);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { s_sy[wv] = sy; s_sx[wv] = sx; s_pw[wv] = pw; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long ty = s_sy[0] + s_sy[1] + s_sy[2] + s_sy[3];
        unsigned long long tx = s_sx[0] + s_sx[1] + s_sx[2] + s_sx[3];
        double tp = ((s_pw[0] + s_pw[1]) + s_pw[2]) + s_pw[3];
        if (centroid) {
            centroid[(long long)g * 2 + 0] = (double)ty / (double)n;
            centroid[(long long)g * 2 + 1] = (double)tx / (double)n;
        }
        if (prior) prior[g] = tp / (double)n;
    }
}

extern "C" int spa_segment_stats(spa_ctx *ctx, const int32_t *labels, int32_t B, int32_t H,
                                 int32_t W, const int32_t *offsets, int32_t Ncap,
                                 double y_rel_pos, double x_rel_pos, double y_rel_sigma,
                                 double x_rel_sigma, int32_t *count, double *centroid,
                                 double *prior, void *stream)
{
    SPA_ARG(ctx && labels && offsets && count && B > 0 && Ncap > 0);
    hipStream_t s = spa_stream(stream);
    int32_t *bbox;
    int rc = spa_ws_reserve(ctx, WS_BBOX, (size_t)Ncap * 16, (void **)&bbox);
    if (rc != SPA_OK) return rc;
    const int npix = H * W;
    SpaProfScope prof_(ctx, PROF_STATS, s);
    hipLaunchKernelGGL(k_bbox_init, dim3((Ncap + 255) / 256), dim3(256), 0, s, bbox, count, Ncap);
    if (W >= 64) {
        const int rows = 16;
        hipLaunchKernelGGL(k_bbox_count_lds, dim3((H + rows - 1) / rows, B), dim3(256), 0, s, labels, W,
                           H, rows, offsets, Ncap, bbox, count, ctx->d_status);
    } else {
        int gx = (npix + 255) / 256;
        if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(k_bbox_count, dim3(gx, B), dim3(256), 0, s, labels, W, npix, offsets, Ncap,
                           bbox, count, ctx->d_status);
    }
    if (centroid || prior) {
        // ymean, xmean = int(h * y_rel_pos), int(w * x_rel_pos); sigma = h * rel_sigma (:116-118)
        double ymean = (double)(long long)((double)H * y_rel_pos);
        double xmean = (double)(long long)((double)W * x_rel_pos);
        double ys = (double)H * y_rel_sigma, xs = (double)W * x_rel_sigma;
        double dy2 = (2.0 * ys) * (2.0 * ys), dx2 = (2.0 * xs) * (2.0 * xs);
        hipLaunchKernelGGL(k_seg_moments, dim3(Ncap), dim3(256), 0, s, labels, B, H, W, offsets,
                           bbox, count, ymean, xmean, dy2, dx2, centroid, prior);
    }
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// rank -> pixel: one wavefront per superpixel walks its bounding box in raster order
__global__ __launch_bounds__(64) void k_select_pixels(const int32_t *__restrict__ labels, int B,
                                                      int H, int W,
                                                      const int32_t *__restrict__ offsets,
                                                      const int32_t *__restrict__ bbox,
                                                      const int32_t *__restrict__ ranks,
                                                      const int32_t *__restrict__ n_valid, int A,
                                                      int32_t *__restrict__ anchors)
{
    const int g = blockIdx.x;
    if (g >= offsets[B]) return;
    const int lane = threadIdx.x;
    const int b = seg_image(offsets, B, g);
    const int s = g - offsets[b];
    const int nv = min(n_valid[g], A);
    if (nv <= 0) return;
    const int y0 = bbox[g * 4 + 0], y1 = bbox[g * 4 + 1], x0 = bbox[g * 4 + 2], x1 = bbox[g * 4 + 3];
    if (y1 < y0) return;
    // lane a holds the a-th requested rank
    const int myrank = lane < nv ? ranks[(long long)g * A + lane] : 0x7fffffff;
    int lo = 0x7fffffff, hi = -1;
    {
        int a = myrank, c = lane < nv ? myrank : -1;
        for (int o = 32; o > 0; o >>= 1) { a = min(a, __shfl_xor(a, o)); c = max(c, __shfl_xor(c, o)); }
        lo = a; hi = c;
    }
    const int32_t *L = labels + (long long)b * H * W;
    int run = 0;
    for (int y = y0; y <= y1 && run <= hi; ++y) {
        for (int xb = x0; xb <= x1; xb += 64) {
            int x = xb + lane;
            bool match = (x <= x1) && (L[(long long)y * W + x] == s);
            unsigned long long m = __ballot(match);
            int c = __popcll(m);
            if (c && run + c > lo) {
                int r = run + (int)spa_rank_in_mask(m);       // rank of this lane's pixel
                for (int a = 0; a < nv; ++a) {
                    int want = __shfl(myrank, a);
                    if (match && r == want) {
                        anchors[((long long)g * A + a) * 2 + 0] = y;
                        anchors[((long long)g * A + a) * 2 + 1] = x;
                    }
                }
            }
            run += c;
        }
    }
}

extern "C" int spa_select_anchor_pixels(spa_ctx *ctx, const int32_t *labels, int32_t B, int32_t H,
                                        int32_t W, const int32_t *offsets, int32_t Ncap,
                                        const int32_t *ranks, const int32_t *n_valid,
                                        int32_t n_anchors, int32_t *anchors, void *stream)
{
    SPA_ARG(ctx && labels && offsets && ranks && n_valid && anchors);
    SPA_ARG(n_anchors > 0 && n_anchors <= 64 && Ncap > 0);
    SPA_ARG(ctx->ws[WS_BBOX] != nullptr && ctx->ws_bytes[WS_BBOX] >= (size_t)Ncap * 16);
    hipLaunchKernelGGL(k_select_pixels, dim3(Ncap), dim3(64), 0, spa_stream(stream), labels, B, H,
                       W, offsets, (const int32_t *)ctx->ws[WS_BBOX], ranks, n_valid, n_anchors,
                       anchors);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
