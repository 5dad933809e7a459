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
                                                    uint32_t *__restrict__ status, const int *__restrict__ flags)
{
    const int b = blockIdx.y;
    if (flags && !flags[b]) return;
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
                                                        uint32_t *__restrict__ status, const int *__restrict__ flags)
{
    __shared__ int l_cnt[BBOX_LDS_MAX], l_y0[BBOX_LDS_MAX], l_y1[BBOX_LDS_MAX], l_x0[BBOX_LDS_MAX], l_x1[BBOX_LDS_MAX];
    const int b = blockIdx.y;
    if (flags && !flags[b]) return;
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

// ---------------------------------------------------------------------------------------
// One streaming pass over the label image for everything a superpixel needs: pixel count, bounding
// box, exact integer coordinate sums (-> centre of mass) and the sum of the Gaussian location prior.
// A workgroup owns a strip of STRIP_ROWS rows; wave w walks rows w, w+4, ... of the strip 64 pixels at
// a time.  For every distinct label of a 64-pixel piece the member lanes' contributions are reduced
// in a fixed order (DPP scan steps; non-members add 0.0: exact) and added to the WAVE's own LDS table, so
// every float64 addition happens in an order fixed by the image alone; the four wave tables are then
// combined in wave order into the strip's row of a global table, and k_stats_final adds the strips in
// strip order.  The label image is read once (4 bytes per pixel; the first version read it about 3.8
// times: a count/bbox pass and a pass over every superpixel's bounding box).
// Labels beyond STATS_LDS_MAX per image take the bounding-box kernel below (k_seg_moments).
// ---------------------------------------------------------------------------------------
#define STRIP_ROWS 16
#define STATS_LDS_MAX 512
struct StripRec { double pw; unsigned long long sy, sx; int cnt, y0, y1, x0, x1, pad; };

// sum of the indices of the set bits of m
__device__ __forceinline__ int spa_bit_index_sum(unsigned long long m)
{
    return __popcll(m & 0xAAAAAAAAAAAAAAAAull) + 2 * __popcll(m & 0xCCCCCCCCCCCCCCCCull) +
           4 * __popcll(m & 0xF0F0F0F0F0F0F0F0ull) + 8 * __popcll(m & 0xFF00FF00FF00FF00ull) +
           16 * __popcll(m & 0xFFFF0000FFFF0000ull) + 32 * __popcll(m & 0xFFFFFFFF00000000ull);
}

// sum of v over the 64 lanes, in a fixed order, with DPP moves (no LDS round trips: a float64 __shfl_xor is two
// ds_bpermute, and six of them per distinct label and piece were most of this kernel's time).  Inclusive scan
// steps row_shr 1, 2, 4, 8, then row_bcast 15 and 31: lane 63 ends with the total; lanes without a source add 0.0.
template <int CTRL, int ROWS>
__device__ __forceinline__ double stats_dpp_add(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROWS, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROWS, 0xF, false);
    return v + __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double stats_wave_sum(double v)
{
    v = stats_dpp_add<0x111, 0xF>(v);
    v = stats_dpp_add<0x112, 0xF>(v);
    v = stats_dpp_add<0x114, 0xF>(v);
    v = stats_dpp_add<0x118, 0xF>(v);
    v = stats_dpp_add<0x142, 0xA>(v);
    v = stats_dpp_add<0x143, 0xC>(v);
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// flags[b] = 1: image b has more than STATS_LDS_MAX labels and takes the two-pass kernels instead
__global__ __launch_bounds__(256) void k_stats_strip(const int32_t *__restrict__ labels, int H, int W,
                                                     const int32_t *__restrict__ offsets, int Ncap,
                                                     double ymean, double xmean, double dy2, double dx2,
                                                     int want_prior, StripRec *__restrict__ table, int nstrip,
                                                     int *__restrict__ flags, uint32_t *__restrict__ status)
{
    // float64 prior sums: one table per wave (ordered additions); integers: one table, LDS atomics
    __shared__ double t_pw[4][STATS_LDS_MAX];
    __shared__ unsigned long long t_sy[STATS_LDS_MAX], t_sx[STATS_LDS_MAX];
    __shared__ int t_cnt[STATS_LDS_MAX], t_x0[STATS_LDS_MAX], t_x1[STATS_LDS_MAX];
    __shared__ int t_y0[STATS_LDS_MAX], t_y1[STATS_LDS_MAX];
    const int b = blockIdx.y, strip = blockIdx.x;
    const int off = offsets[b], S = offsets[b + 1] - off;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (S > STATS_LDS_MAX) {
        if (strip == 0 && tid == 0) flags[b] = 1;
        return;
    }
    for (int l = tid; l < S; l += 256) {
        t_pw[0][l] = 0.0; t_pw[1][l] = 0.0; t_pw[2][l] = 0.0; t_pw[3][l] = 0.0;
        t_sy[l] = 0; t_sx[l] = 0; t_cnt[l] = 0;
        t_x0[l] = 0x7fffffff; t_x1[l] = -1; t_y0[l] = 0x7fffffff; t_y1[l] = -1;
    }
    __syncthreads();
    const int32_t *L = labels + (long long)b * H * W;
    const int r0 = strip * STRIP_ROWS, r1 = min(H, r0 + STRIP_ROWS);
    for (int y = r0 + wv; y < r1; y += 4) {
        const double ty = ((double)y - ymean) * ((double)y - ymean) / dy2;
        // eight 64-pixel pieces of the row are loaded before the first is used: one memory latency per
        // 512 pixels instead of one per piece
        for (int xg = 0; xg < W; xg += 512) {
            int lab8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int x = xg + u * 64 + lane;
                lab8[u] = (x < W) ? L[(long long)y * W + x] : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int xb = xg + u * 64;
                if (xb >= W) break;
                const int x = xb + lane;
                int l = lab8[u];
                if (x < W && (l < 0 || l >= S || off + l >= Ncap)) { atomicOr(status, SPA_ST_LABEL_RANGE); l = -1; }
                double e = 0.0;
                if (want_prior && l >= 0) {
                    const double tx = ((double)x - xmean) * ((double)x - xmean) / dx2;
                    e = spa_det_exp(-(ty + tx));
                }
                unsigned long long todo = __ballot(l >= 0);
                while (todo) {
                    const int leader = __ffsll((long long)todo) - 1;
                    const int ll = __shfl(l, leader);
                    const unsigned long long same = __ballot(l == ll);
                    double v = (l == ll) ? e : 0.0;
                    if (want_prior) v = stats_wave_sum(v);
                    if (lane == leader) {
                        const int n = __popcll(same);
                        const int last = 63 - __clzll((long long)same);
                        // lanes are consecutive pixels of one row: sum of x = n * xb + sum of the lane indices
                        t_pw[wv][ll] = t_pw[wv][ll] + v;
                        atomicAdd(&t_sy[ll], (unsigned long long)n * (unsigned)y);
                        atomicAdd(&t_sx[ll], (unsigned long long)n * (unsigned)xb + (unsigned)spa_bit_index_sum(same));
                        atomicAdd(&t_cnt[ll], n);
                        atomicMin(&t_x0[ll], xb + leader);
                        atomicMax(&t_x1[ll], xb + last);
                        atomicMin(&t_y0[ll], y);
                        atomicMax(&t_y1[ll], y);
                    }
                    todo &= ~same;
                }
            }
        }
    }
    __syncthreads();
    for (int l = tid; l < S; l += 256) {
        StripRec r;
        r.pw = ((t_pw[0][l] + t_pw[1][l]) + t_pw[2][l]) + t_pw[3][l];
        r.sy = t_sy[l]; r.sx = t_sx[l]; r.cnt = t_cnt[l];
        r.y0 = t_y0[l]; r.y1 = t_y1[l]; r.x0 = t_x0[l]; r.x1 = t_x1[l];
        r.pad = 0;
        table[((long long)b * nstrip + strip) * STATS_LDS_MAX + l] = r;
    }
}

// strips in order -> count, bounding box, centre of mass, prior of every superpixel with label < smax
__global__ __launch_bounds__(256) void k_stats_final(const StripRec *__restrict__ table, int nstrip, int smax,
                                                     const int *__restrict__ flags,
                                                     const int32_t *__restrict__ offsets, int B, int Ncap,
                                                     int32_t *__restrict__ bbox, int32_t *__restrict__ count,
                                                     double *__restrict__ centroid, double *__restrict__ prior)
{
    const int b = blockIdx.y;
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int off = offsets[b], S = offsets[b + 1] - off;
    if (flags[b] || l >= S || l >= smax || off + l >= Ncap) return;
    const int g = off + l;
    double pw = 0.0;
    unsigned long long sy = 0, sx = 0;
    int n = 0, y0 = 0x7fffffff, y1 = -1, x0 = 0x7fffffff, x1 = -1;
    for (int q = 0; q < nstrip; ++q) {
        const StripRec r = table[((long long)b * nstrip + q) * smax + l];
        if (r.cnt == 0) continue;
        pw = pw + r.pw; sy += r.sy; sx += r.sx; n += r.cnt;
        y0 = min(y0, r.y0); y1 = max(y1, r.y1); x0 = min(x0, r.x0); x1 = max(x1, r.x1);
    }
    count[g] = n;
    bbox[g * 4 + 0] = y0; bbox[g * 4 + 1] = y1; bbox[g * 4 + 2] = x0; bbox[g * 4 + 3] = x1;
    if (centroid) {
        centroid[(long long)g * 2 + 0] = (double)sy / (double)n;
        centroid[(long long)g * 2 + 1] = (double)sx / (double)n;
    }
    if (prior) prior[g] = pw / (double)n;
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
                                                     double *__restrict__ prior, const int *__restrict__ flags)
{
    __shared__ unsigned long long s_sy[4], s_sx[4];
    __shared__ double s_pw[4];
    const int g = blockIdx.x;
    if (g >= offsets[B]) return;
    const int b = seg_image(offsets, B, g);
    if (flags && !flags[b]) return;
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
    // ymean, xmean = int(h * y_rel_pos), int(w * x_rel_pos); sigma = h * rel_sigma (:116-118)
    const double ymean = (double)(long long)((double)H * y_rel_pos);
    const double xmean = (double)(long long)((double)W * x_rel_pos);
    const double ys = (double)H * y_rel_sigma, xs = (double)W * x_rel_sigma;
    const double dy2 = (2.0 * ys) * (2.0 * ys), dx2 = (2.0 * xs) * (2.0 * xs);
    // Images with at most STATS_LDS_MAX superpixels (every SLIC map) take the single streaming pass; an
    // image with more (a fine felzenszwalb map) raises its flag there and takes the two-pass kernels, which
    // exit at once for all other images.
    const int *flags = nullptr;
    hipLaunchKernelGGL(k_bbox_init, dim3((Ncap + 255) / 256), dim3(256), 0, s, bbox, count, Ncap);
    if (W >= 64) {
        const int nstrip = (H + STRIP_ROWS - 1) / STRIP_ROWS;
        char *ws;
        const size_t fl_bytes = ((size_t)B * 4 + 255) & ~(size_t)255;
        if ((rc = spa_ws_reserve(ctx, WS_OVERLAP, fl_bytes + (size_t)B * nstrip * STATS_LDS_MAX * sizeof(StripRec),
                                 (void **)&ws)) != SPA_OK) return rc;
        int *fl = (int *)ws;
        StripRec *table = (StripRec *)(ws + fl_bytes);
        SPA_HIP(hipMemsetAsync(fl, 0, fl_bytes, s));
        hipLaunchKernelGGL(k_stats_strip, dim3(nstrip, B), dim3(256), 0, s, labels, H, W, offsets, Ncap, ymean,
                           xmean, dy2, dx2, prior ? 1 : 0, table, nstrip, fl, ctx->d_status);
        hipLaunchKernelGGL(k_stats_final, dim3((STATS_LDS_MAX + 255) / 256, B), dim3(256), 0, s,
                           (const StripRec *)table, nstrip, STATS_LDS_MAX, (const int *)fl, offsets, B, Ncap, bbox,
                           count, centroid, prior);
        flags = fl;
    }
    if (W >= 64) {
        const int rows = 16;
        hipLaunchKernelGGL(k_bbox_count_lds, dim3((H + rows - 1) / rows, B), dim3(256), 0, s, labels, W,
                           H, rows, offsets, Ncap, bbox, count, ctx->d_status, flags);
    } else {
        int gx = (npix + 255) / 256;
        if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(k_bbox_count, dim3(gx, B), dim3(256), 0, s, labels, W, npix, offsets, Ncap,
                           bbox, count, ctx->d_status, flags);
    }
    if (centroid || prior) {
        hipLaunchKernelGGL(k_seg_moments, dim3(Ncap), dim3(256), 0, s, labels, B, H, W, offsets,
                           bbox, count, ymean, xmean, dy2, dx2, centroid, prior, flags);
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
