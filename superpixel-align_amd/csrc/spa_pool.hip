// Pooling of the deep feature map into one descriptor per superpixel.
//   anchor mode : superpixel_align(), batch_spalign_kmeans.py:210-276 (the shipped semantics)
//   mean mode   : dense per-segment mean, notebooks/Superpixel_Align.ipynb cell 4
// The feature map must be channels-last (NHWC): the C channels of one feature pixel are
// contiguous, so every gather is a coalesced row read (2 KB for C = 512 float32) and the
// thread <-> channel mapping needs no atomics and no cross-lane reduction.
// HBM-bound: algorithmic bytes = C*fh*fw*sizeof(dtype) + H*W*4 + N*C*4 per image.
#include "spa_common.h"

__device__ __forceinline__ int seg_image_p(const int32_t *offsets, int B, int g)
{
    int b = 0;
    while (b + 1 < B && offsets[b + 1] <= g) ++b;
    return b;
}

__device__ __forceinline__ float load_feat(const void *base, long long idx, int dtype)
{
    if (dtype == 0) return ((const float *)base)[idx];
    unsigned short h = ((const unsigned short *)base)[idx];
    return __uint_as_float(((unsigned)h) << 16);      // bfloat16 -> float32, exact
}

__device__ __forceinline__ void store_x(void *X, int x_dtype, long long idx, double v)
{
    if (x_dtype == 0) ((float *)X)[idx] = (float)v;
    else ((double *)X)[idx] = v;
}

// ---------------------------------------------------------------------------------------
// anchor mode
// ---------------------------------------------------------------------------------------
struct AnchorGeom { int y0, y1, x0, x1; float w11, w12, w21, w22, inv; };

// 4 nearest feature-pixel centres of (py, px): float64 distances (sqrt), ties broken by the
// lowest flat index n = x*fh + y (stable argsort of the reference's x-major list, :219-221).
__device__ void anchor_geometry(double py, double px, int fh, int fw, int nn, AnchorGeom &g)
{
    double bd[16]; int by[16], bx[16]; long long bn[16];
    int nb = 0;
    if (nn > 16) nn = 16;
    const int cyi = (int)floor(py), cxi = (int)floor(px);
    for (int x = cxi - 3; x <= cxi + 3; ++x) {
        if (x < 0 || x >= fw) continue;
        for (int y = cyi - 3; y <= cyi + 3; ++y) {
            if (y < 0 || y >= fh) continue;
            double ddy = ((double)y + 0.5) - py, ddx = ((double)x + 0.5) - px;
            double d = sqrt(ddy * ddy + ddx * ddx);
            long long n = (long long)x * fh + y;
            int pos = nb;
            while (pos > 0 && (bd[pos - 1] > d || (bd[pos - 1] == d && bn[pos - 1] > n))) --pos;
            if (pos >= nn) continue;
            int last = nb < nn ? nb : nn - 1;
            for (int j = last; j > pos; --j) { bd[j] = bd[j - 1]; by[j] = by[j - 1]; bx[j] = bx[j - 1]; bn[j] = bn[j - 1]; }
            bd[pos] = d; by[pos] = y; bx[pos] = x; bn[pos] = n;
            if (nb < nn) ++nb;
        }
    }
    int y0 = by[0], y1 = by[0], x0 = bx[0], x1 = bx[0];
    for (int j = 1; j < nb; ++j) {
        y0 = min(y0, by[j]); y1 = max(y1, by[j]); x0 = min(x0, bx[j]); x1 = max(x1, bx[j]);
    }
    double min_y = y0 + 0.5, max_y = y1 + 0.5, min_x = x0 + 0.5, max_x = x1 + 0.5;
    g.y0 = y0; g.y1 = y1; g.x0 = x0; g.x1 = x1;
    // float64 scalar products narrowed to float32 before they meet the float32 features
    g.w11 = (float)((max_x - px) * (max_y - py));
    g.w12 = (float)((max_x - px) * (py - min_y));
    g.w21 = (float)((px - min_x) * (max_y - py));
    g.w22 = (float)((px - min_x) * (py - min_y));
    g.inv = (float)(1.0 / ((max_x - min_x) * (max_y - min_y)));
}

#define MAX_ANCH 64
__global__ __launch_bounds__(256) void k_pool_anchor(const void *__restrict__ fmap, int dtype, int C,
                                                     int fh, int fw, long long sb, long long sy,
                                                     long long sx, int B, int img_h,
                                                     const int32_t *__restrict__ offsets,
                                                     const int32_t *__restrict__ anchors,
                                                     const int32_t *__restrict__ n_valid, int A,
                                                     int nn, const double *__restrict__ centroid,
                                                     int append_pos, void *__restrict__ X,
                                                     int x_dtype, long long ld)
{
    __shared__ AnchorGeom geo[MAX_ANCH];
    const int g = blockIdx.x;
    if (g >= offsets[B]) return;
    const int b = seg_image_p(offsets, B, g);
    const int nv = min(n_valid[g], A);
    if ((int)threadIdx.x < nv) {
        // selected_points = coords * (fh / img_h) + 0.5, clipped (:235-240); y ratio for both axes
        const double ratio = (double)fh / (double)img_h;
        double py = (double)anchors[((long long)g * A + threadIdx.x) * 2 + 0] * ratio + 0.5;
        double px = (double)anchors[((long long)g * A + threadIdx.x) * 2 + 1] * ratio + 0.5;
        const double hy = (double)(fh - 1) + 0.5, hx = (double)(fw - 1) + 0.5;
        py = py < 0.0 ? 0.0 : (py > hy ? hy : py);
        px = px < 0.0 ? 0.0 : (px > hx ? hx : px);
        anchor_geometry(py, px, fh, fw, nn, geo[threadIdx.x]);
    }
    __syncthreads();
    const long long fb = (long long)b * sb;
    for (int c = threadIdx.x; c < C; c += 256) {
        double acc = 0.0;
        float acc32 = 0.0f;
        for (int a = 0; a < nv; ++a) {
            const AnchorGeom q = geo[a];
            float f11 = load_feat(fmap, fb + q.y0 * sy + q.x0 * sx + c, dtype);
            float f12 = load_feat(fmap, fb + q.y1 * sy + q.x0 * sx + c, dtype);
            float f21 = load_feat(fmap, fb + q.y0 * sy + q.x1 * sx + c, dtype);
            float f22 = load_feat(fmap, fb + q.y1 * sy + q.x1 * sx + c, dtype);
            float v = q.w11 * f11;
            v = v + q.w12 * f12;
            v = v + q.w21 * f21;
            v = v + q.w22 * f22;
            v = q.inv * v;
            if (append_pos) acc = acc + (double)v;     // float64 container after hstack (:270)
            else acc32 = acc32 + v;                     // float32 mean otherwise
        }
        double r = append_pos ? acc / (double)nv : (double)(acc32 / (float)nv);
        store_x(X, x_dtype, (long long)g * ld + c, r);
    }
    if (append_pos && threadIdx.x < 2) {
        // np.mean over nv identical centroid rows: sequential sum, then divide
        double cv = centroid[(long long)g * 2 + threadIdx.x], sacc = 0.0;
        for (int a = 0; a < nv; ++a) sacc = sacc + cv;
        store_x(X, x_dtype, (long long)g * ld + C + threadIdx.x, sacc / (double)nv);
    }
}

extern "C" int spa_pool_anchor(spa_ctx *ctx, const void *fmap, const spa_fmap_desc *d, int32_t B,
                               int32_t img_h, const int32_t *offsets, int32_t Ncap,
                               const int32_t *anchors, const int32_t *n_valid, int32_t n_anchors,
                               int32_t n_neighbors, const double *centroid, int32_t append_pos,
                               void *X, int32_t x_dtype, int64_t ld, void *stream)
{
    SPA_ARG(ctx && fmap && d && offsets && anchors && n_valid && X);
    SPA_ARG(n_anchors > 0 && n_anchors <= MAX_ANCH && n_neighbors > 0 && n_neighbors <= 16);
    SPA_ARG(!append_pos || centroid);
    SPA_ARG(ld >= d->C + (append_pos ? 2 : 0));
    if (d->stride_c != 1) {
        spa_set_error("feature map must be channels-last (stride_c == 1), got stride_c=%lld",
                      (long long)d->stride_c);
        return SPA_ERR_LAYOUT;
    }
    SpaProfScope prof_(ctx, PROF_POOL_ANCHOR, spa_stream(stream));
    hipLaunchKernelGGL(k_pool_anchor, dim3(Ncap), dim3(256), 0, spa_stream(stream), fmap, d->dtype,
                       d->C, d->fh, d->fw, (long long)d->stride_b, (long long)d->stride_y,
                       (long long)d->stride_x, B, img_h, offsets, anchors, n_valid, n_anchors,
                       n_neighbors, centroid, append_pos, X, x_dtype, (long long)ld);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// ---------------------------------------------------------------------------------------
// mean mode.  Step 1: per feature pixel, the (superpixel, weight) pairs of the image pixels
// that sample it, accumulated in raster order of the image pixels.
// ---------------------------------------------------------------------------------------
struct CellSlots { int n; int lab[SPA_CELL_SLOTS]; float w[SPA_CELL_SLOTS]; };
#define CELL_OVERFLOW (-1)      // n of a cell more than SPA_CELL_SLOTS superpixels touch: weights recomputed on demand

// every (image pixel, tap weight) that samples feature pixel (u, v), in raster order of the image pixels:
// f(label, tap).  sampling 0: nearest (image pixel (y, x) samples (y*fh//H, x*fw//W), tap 1);
// 1: chainer F.resize_images (u = y*(fh-1)/(H-1), 4 taps, float32 weights, taps of one pixel in the order
// (u0,v0) (u0,v1) (u1,v0) (u1,v1); several may hit this cell)
template <typename F>
__device__ __forceinline__ void cell_taps(const int32_t *__restrict__ L, int H, int W, int fh, int fw,
                                          int sampling, int u, int v, F f)
{
    if (sampling == 0) {
        const int ylo = (int)(((long long)u * H + fh - 1) / fh), yhi = (int)(((long long)(u + 1) * H + fh - 1) / fh);
        const int xlo = (int)(((long long)v * W + fw - 1) / fw), xhi = (int)(((long long)(v + 1) * W + fw - 1) / fw);
        for (int y = ylo; y < yhi; ++y)
            for (int x = xlo; x < xhi; ++x) f(L[(long long)y * W + x], 1.0f);
    } else {
        const float ry = (H > 1) ? ((float)(fh - 1) / (float)(H - 1)) : 0.0f;
        const float rx = (W > 1) ? ((float)(fw - 1) / (float)(W - 1)) : 0.0f;
        int ylo = 0, yhi = H - 1, xlo = 0, xhi = W - 1;
        if (ry > 0.0f) { ylo = max(0, (int)((float)(u - 1) / ry) - 1); yhi = min(H - 1, (int)((float)(u + 1) / ry) + 2); }
        if (rx > 0.0f) { xlo = max(0, (int)((float)(v - 1) / rx) - 1); xhi = min(W - 1, (int)((float)(v + 1) / rx) + 2); }
        for (int y = ylo; y <= yhi; ++y) {
            float uu = (H > 1) ? (float)y * ry : 0.0f;
            int u0 = (int)uu; if (u0 > fh - 1) u0 = fh - 1;
            int u1 = u0 + 1 < fh ? u0 + 1 : fh - 1;
            if (u0 != u && u1 != u) continue;
            float fu = uu - (float)u0;
            for (int x = xlo; x <= xhi; ++x) {
                float vv = (W > 1) ? (float)x * rx : 0.0f;
                int v0 = (int)vv; if (v0 > fw - 1) v0 = fw - 1;
                int v1 = v0 + 1 < fw ? v0 + 1 : fw - 1;
                if (v0 != v && v1 != v) continue;
                float fv = vv - (float)v0;
                const int s = L[(long long)y * W + x];
                if (u0 == u && v0 == v) f(s, (1.0f - fu) * (1.0f - fv));
                if (u0 == u && v1 == v) f(s, (1.0f - fu) * fv);
                if (u1 == u && v0 == v) f(s, fu * (1.0f - fv));
                if (u1 == u && v1 == v) f(s, fu * fv);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_cell_weights(const int32_t *__restrict__ labels, int H,
                                                      int W, int fh, int fw, int sampling,
                                                      const int32_t *__restrict__ offsets,
                                                      CellSlots *__restrict__ cells,
                                                      uint32_t *__restrict__ status)
{
    __shared__ int s_lab[SPA_CELL_SLOTS * 256];
    __shared__ float s_w[SPA_CELL_SLOTS * 256];
    const int b = blockIdx.y;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= fh * fw) return;
    const int S = offsets[b + 1] - offsets[b];
    const int u = cell / fw, v = cell - u * fw;
    const int32_t *L = labels + (long long)b * H * W;
    const int t = threadIdx.x;
    int n = 0;
    bool overflow = false, range = false;
    cell_taps(L, H, W, fh, fw, sampling, u, v, [&](int s, float tap) {
        if (s < 0 || s >= S) { range = true; return; }
        int j = 0;
        while (j < n && s_lab[j * 256 + t] != s) ++j;
        if (j == n) {
            if (n == SPA_CELL_SLOTS) { overflow = true; return; }
            s_lab[n * 256 + t] = s; s_w[n * 256 + t] = 0.0f; ++n;
        }
        s_w[j * 256 + t] = s_w[j * 256 + t] + tap;
    });
    if (range) atomicOr(status, SPA_ST_LABEL_RANGE);
    CellSlots *o = cells + (long long)b * fh * fw + cell;
    // more superpixels than slots (felzenszwalb with a small min_size puts up to one segment per pixel under a
    // feature pixel): the cell is flagged and whoever asks for a segment's weight recomputes it from the labels,
    // pixels in the same raster order, hence the same float32 sum (cell_weight)
    o->n = overflow ? CELL_OVERFLOW : n;
    if (!overflow)
        for (int j = 0; j < n; ++j) { o->lab[j] = s_lab[j * 256 + t]; o->w[j] = s_w[j * 256 + t]; }
}

// weight of superpixel s at feature pixel (u, v): slot lookup, or the recomputation for a flagged cell
__device__ __forceinline__ bool cell_weight(const CellSlots *__restrict__ cs, int s, const int32_t *__restrict__ L,
                                            int H, int W, int fh, int fw, int sampling, int u, int v, float &w)
{
    const int nn = cs->n;
    for (int j = 0; j < nn; ++j)
        if (cs->lab[j] == s) { w = cs->w[j]; return true; }
    if (nn != CELL_OVERFLOW) return false;
    bool found = false;
    float acc = 0.0f;
    cell_taps(L, H, W, fh, fw, sampling, u, v, [&](int l, float tap) {
        if (l == s) { acc = acc + tap; found = true; }
    });
    w = acc;
    return found;
}

// Step 2: one workgroup per superpixel.  All 256 threads first search the feature pixels of the
// segment's bounding box in parallel for the segment's weight and compact the hits, in raster
// order, into an LDS list (cell offset, weight); then thread t, owner of channels t, t+256, ...,
// walks the list: acc = acc + w * F (mul and add rounded apart), 4 independent row reads in flight.
#define MAX_CPT 8
#define POOL_LIST 1024
__global__ __launch_bounds__(256) void k_pool_mean(const void *__restrict__ fmap, int dtype, int C,
                                                   int fh, int fw, long long sb, long long sy,
                                                   long long sx, int B, int H, int W, int sampling,
                                                   const int32_t *__restrict__ offsets,
                                                   const int32_t *__restrict__ bbox,
                                                   const int32_t *__restrict__ count,
                                                   const CellSlots *__restrict__ cells,
                                                   const int32_t *__restrict__ labels,
                                                   const double *__restrict__ centroid,
                                                   int append_pos, void *__restrict__ X,
                                                   int x_dtype, long long ld)
{
    __shared__ long long l_off[POOL_LIST];
    __shared__ float l_w[POOL_LIST];
    __shared__ int wave_cnt[4];
    const int g = blockIdx.x;
    if (g >= offsets[B]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = seg_image_p(offsets, B, g);
    const int s = g - offsets[b];
    const int n = count[g];
    const int y0 = bbox[g * 4 + 0], y1 = bbox[g * 4 + 1], x0 = bbox[g * 4 + 2], x1 = bbox[g * 4 + 3];
    float acc[MAX_CPT];
#pragma unroll
    for (int i = 0; i < MAX_CPT; ++i) acc[i] = 0.0f;
    if (n > 0) {
        int u0, u1, v0, v1;
        if (sampling == 0) {
            u0 = (int)((long long)y0 * fh / H); u1 = (int)((long long)y1 * fh / H);
            v0 = (int)((long long)x0 * fw / W); v1 = (int)((long long)x1 * fw / W);
        } else {
            const float ry = (H > 1) ? ((float)(fh - 1) / (float)(H - 1)) : 0.0f;
            const float rx = (W > 1) ? ((float)(fw - 1) / (float)(W - 1)) : 0.0f;
            u0 = max(0, (int)((float)y0 * ry) - 1); u1 = min(fh - 1, (int)((float)y1 * ry) + 2);
            v0 = max(0, (int)((float)x0 * rx) - 1); v1 = min(fw - 1, (int)((float)x1 * rx) + 2);
        }
        const CellSlots *cb = cells + (long long)b * fh * fw;
        const long long fb = (long long)b * sb;
        const int bwc = v1 - v0 + 1;
        const int ncell = (u1 - u0 + 1) * bwc;
        int len = 0;                               // workgroup-uniform list length
        for (int c0 = 0; c0 < ncell || len > 0; c0 += 256) {
            if (c0 < ncell) {
                const int ci = c0 + tid;
                float w = 0.0f;
                bool found = false;
                long long fo = 0;
                if (ci < ncell) {
                    const int u = u0 + ci / bwc, v = v0 + ci % bwc;
                    found = cell_weight(cb + (long long)u * fw + v, s, labels + (long long)b * H * W, H, W, fh, fw,
                                        sampling, u, v, w);
                    fo = fb + u * sy + v * sx;
                }
                const unsigned long long m = __ballot(found);
                if (lane == 0) wave_cnt[wv] = __popcll(m);
                __syncthreads();
                int off = len, tot = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) { int c = wave_cnt[i]; if (i < wv) off += c; tot += c; }
                if (found) { int pos = off + (int)spa_rank_in_mask(m); l_off[pos] = fo; l_w[pos] = w; }
                len += tot;
                __syncthreads();
            }
            // consume when the next chunk might not fit, or at the end
            if (len + 256 > POOL_LIST || c0 + 256 >= ncell) {
                int j = 0;
                for (; j + 4 <= len; j += 4) {
                    const long long o0 = l_off[j], o1 = l_off[j + 1], o2 = l_off[j + 2], o3 = l_off[j + 3];
                    const float w0 = l_w[j], w1 = l_w[j + 1], w2 = l_w[j + 2], w3 = l_w[j + 3];
#pragma unroll
                    for (int i = 0; i < MAX_CPT; ++i) {
                        const int c = tid + i * 256;
                        if (c < C) {
                            const float f0 = load_feat(fmap, o0 + c, dtype), f1 = load_feat(fmap, o1 + c, dtype);
                            const float f2 = load_feat(fmap, o2 + c, dtype), f3 = load_feat(fmap, o3 + c, dtype);
                            float pr = w0 * f0; acc[i] = acc[i] + pr;
                            pr = w1 * f1; acc[i] = acc[i] + pr;
                            pr = w2 * f2; acc[i] = acc[i] + pr;
                            pr = w3 * f3; acc[i] = acc[i] + pr;
                        }
                    }
                }
                for (; j < len; ++j) {
                    const long long o0 = l_off[j];
                    const float w0 = l_w[j];
#pragma unroll
                    for (int i = 0; i < MAX_CPT; ++i) {
                        const int c = tid + i * 256;
                        if (c < C) { float pr = w0 * load_feat(fmap, o0 + c, dtype); acc[i] = acc[i] + pr; }
                    }
                }
                __syncthreads();
                len = 0;
            }
        }
    }
    const float tot = (float)n;
#pragma unroll
    for (int i = 0; i < MAX_CPT; ++i) {
        int c = threadIdx.x + i * 256;
        if (c < C) store_x(X, x_dtype, (long long)g * ld + c, (double)(acc[i] / tot));
    }
    if (append_pos && threadIdx.x < 2)
        store_x(X, x_dtype, (long long)g * ld + C + threadIdx.x, centroid[(long long)g * 2 + threadIdx.x]);
}

// Step 2, vector form (channels-last rows that are multiples of 16 bytes): 128 threads per superpixel,
// thread t owns VEC consecutive channels per pass (4 float32 or 8 bfloat16 = one 16-byte load), so a
// feature pixel is fetched by full-width loads and EIGHT cells are in flight per thread; the products and
// the running sums keep the order of the scalar kernel (cells in raster order, multiply and add rounded
// apart), so the descriptors are bit-identical to it and to the oracle.
#define POOLV_THREADS 128
template <int DT, int NP>
__global__ __launch_bounds__(POOLV_THREADS) void k_pool_mean_vec(const void *__restrict__ fmap, int C,
                                                                 int fh, int fw, long long sb, long long sy,
                                                                 long long sx, int B, int H, int W, int sampling,
                                                                 const int32_t *__restrict__ offsets,
                                                                 const int32_t *__restrict__ bbox,
                                                                 const int32_t *__restrict__ count,
                                                                 const CellSlots *__restrict__ cells,
                                                                 const int32_t *__restrict__ labels,
                                                                 const double *__restrict__ centroid,
                                                                 int append_pos, void *__restrict__ X,
                                                                 int x_dtype, long long ld)
{
    constexpr int VEC = DT == 0 ? 4 : 8;
    __shared__ long long l_off[POOL_LIST];
    __shared__ float l_w[POOL_LIST];
    __shared__ int wave_cnt[2];
    // (contiguous segment ranges per XCD — whole images per L2 — were measured: same FETCH_SIZE, 0.61 -> 0.84 ms;
    // the round-robin order spreads every image over all XCDs and memory channels)
    const int g = blockIdx.x;
    if (g >= offsets[B]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nthr = blockDim.x;                 // 64 or 128: as many threads as one pass over the channels needs
    const int b = seg_image_p(offsets, B, g);
    const int s = g - offsets[b];
    const int n = count[g];
    const int y0 = bbox[g * 4 + 0], y1 = bbox[g * 4 + 1], x0 = bbox[g * 4 + 2], x1 = bbox[g * 4 + 3];
    float acc[NP][VEC];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[p][v] = 0.0f;
    // channel base of pass p: (p * 128 + tid) * VEC; inactive when beyond C
    auto accumulate = [&](const uint4 (&q)[NP], float w) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float f[VEC];
            if (DT == 0) {
                f[0] = __uint_as_float(q[p].x); f[1] = __uint_as_float(q[p].y);
                f[2] = __uint_as_float(q[p].z); f[3] = __uint_as_float(q[p].w);
            } else {
                const unsigned u[4] = {q[p].x, q[p].y, q[p].z, q[p].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f[2 * k] = __uint_as_float(u[k] << 16);            // bfloat16 -> float32, exact
                    f[2 * k + 1] = __uint_as_float(u[k] & 0xffff0000u);
                }
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) { const float pr = w * f[v]; acc[p][v] = acc[p][v] + pr; }
        }
    };
    if (n > 0) {
        int u0, u1, v0, v1;
        if (sampling == 0) {
            u0 = (int)((long long)y0 * fh / H); u1 = (int)((long long)y1 * fh / H);
            v0 = (int)((long long)x0 * fw / W); v1 = (int)((long long)x1 * fw / W);
        } else {
            const float ry = (H > 1) ? ((float)(fh - 1) / (float)(H - 1)) : 0.0f;
            const float rx = (W > 1) ? ((float)(fw - 1) / (float)(W - 1)) : 0.0f;
            u0 = max(0, (int)((float)y0 * ry) - 1); u1 = min(fh - 1, (int)((float)y1 * ry) + 2);
            v0 = max(0, (int)((float)x0 * rx) - 1); v1 = min(fw - 1, (int)((float)x1 * rx) + 2);
        }
        const CellSlots *cb = cells + (long long)b * fh * fw;
        const long long fb = (long long)b * sb;
        const int bwc = v1 - v0 + 1;
        const int ncell = (u1 - u0 + 1) * bwc;
        const char *base = (const char *)fmap;
        constexpr int ES = DT == 0 ? 4 : 2;
        bool act[NP];
        long long choff[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int c0 = (p * nthr + tid) * VEC;
            act[p] = c0 < C;
            choff[p] = (long long)(act[p] ? c0 : 0) * ES;
        }
        int len = 0;                               // workgroup-uniform list length
        for (int c0 = 0; c0 < ncell || len > 0; c0 += nthr) {
            if (c0 < ncell) {
                const int ci = c0 + tid;
                float w = 0.0f;
                bool found = false;
                long long fo = 0;
                if (ci < ncell) {
                    const int u = u0 + ci / bwc, v = v0 + ci % bwc;
                    found = cell_weight(cb + (long long)u * fw + v, s, labels + (long long)b * H * W, H, W, fh, fw,
                                        sampling, u, v, w);
                    fo = (fb + u * sy + v * sx) * ES;
                }
                const unsigned long long m = __ballot(found);
                if (lane == 0) wave_cnt[wv] = __popcll(m);
                __syncthreads();
                const int off = len + (wv ? wave_cnt[0] : 0), tot = wave_cnt[0] + (nthr > 64 ? wave_cnt[1] : 0);
                if (found) { const int pos = off + (int)spa_rank_in_mask(m); l_off[pos] = fo; l_w[pos] = w; }
                len += tot;
                __syncthreads();
            }
            // consume when the next chunk might not fit, or at the end
            if (len + nthr > POOL_LIST || c0 + nthr >= ncell) {
                int j = 0;
                for (; j + 8 <= len; j += 8) {
                    uint4 q[8][NP];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
#pragma unroll
                        for (int p = 0; p < NP; ++p)
                            q[e][p] = *(const uint4 *)(base + l_off[j + e] + choff[p]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) accumulate(q[e], l_w[j + e]);
                }
                for (; j < len; ++j) {
                    uint4 q[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) q[p] = *(const uint4 *)(base + l_off[j] + choff[p]);
                    accumulate(q, l_w[j]);
                }
                __syncthreads();
                len = 0;
            }
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) (void)act[p];
    }
    const float tot = (float)n;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int c0 = (p * nthr + tid) * VEC;
        if (c0 < C) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) store_x(X, x_dtype, (long long)g * ld + c0 + v, (double)(acc[p][v] / tot));
        }
    }
    if (append_pos && tid < 2)
        store_x(X, x_dtype, (long long)g * ld + C + tid, centroid[(long long)g * 2 + tid]);
}

extern "C" int spa_pool_mean(spa_ctx *ctx, const void *fmap, const spa_fmap_desc *d,
                             const int32_t *labels, int32_t B, int32_t H, int32_t W,
                             const int32_t *offsets, int32_t Ncap, const int32_t *count,
                             int32_t sampling, const double *centroid, int32_t append_pos,
                             void *X, int32_t x_dtype, int64_t ld, void *stream)
{
    SPA_ARG(ctx && fmap && d && labels && offsets && count && X);
    SPA_ARG(sampling == 0 || sampling == 1);
    SPA_ARG(d->C > 0 && d->C <= MAX_CPT * 256);
    SPA_ARG(!append_pos || centroid);
    SPA_ARG(ld >= d->C + (append_pos ? 2 : 0));
    SPA_ARG(ctx->ws[WS_BBOX] != nullptr && ctx->ws_bytes[WS_BBOX] >= (size_t)Ncap * 16);
    if (d->stride_c != 1) {
        spa_set_error("feature map must be channels-last (stride_c == 1), got stride_c=%lld",
                      (long long)d->stride_c);
        return SPA_ERR_LAYOUT;
    }
    hipStream_t s = spa_stream(stream);
    CellSlots *cells;
    const int ncell = d->fh * d->fw;
    int rc = spa_ws_reserve(ctx, WS_CELLSLOT, (size_t)B * ncell * sizeof(CellSlots), (void **)&cells);
    if (rc != SPA_OK) return rc;
    { SpaProfScope prof_(ctx, PROF_CELL_WEIGHTS, s);
    hipLaunchKernelGGL(k_cell_weights, dim3((ncell + 255) / 256, B), dim3(256), 0, s, labels, H, W,
                       d->fh, d->fw, sampling, offsets, cells, ctx->d_status); }
    SpaProfScope prof_(ctx, PROF_POOL_MEAN, s);
    {
        // vector kernel: every feature pixel's row starts on a 16-byte boundary and holds whole vectors
        const int es = d->dtype == 0 ? 4 : 2, vec = 16 / es;
        const bool aligned = ((uintptr_t)fmap % 16 == 0) && (d->stride_x * es) % 16 == 0 &&
                             (d->stride_y * es) % 16 == 0 && (d->stride_b * es) % 16 == 0 && d->C % vec == 0;
        const int nthr = d->C <= 64 * vec ? 64 : POOLV_THREADS;
        const int np = (d->C + nthr * vec - 1) / (nthr * vec);
        if (aligned && np <= 4) {
#define POOLV_LAUNCH(DT, NP)                                                                                    \
            hipLaunchKernelGGL((k_pool_mean_vec<DT, NP>), dim3(Ncap), dim3(nthr), 0, s, fmap, d->C, d->fh,            \
                               d->fw, (long long)d->stride_b, (long long)d->stride_y, (long long)d->stride_x, B, H,  \
                               W, sampling, offsets, (const int32_t *)ctx->ws[WS_BBOX], count, cells, labels, centroid, \
                               append_pos, X, x_dtype, (long long)ld)
            if (d->dtype == 0) {
                if (np == 1) POOLV_LAUNCH(0, 1); else if (np == 2) POOLV_LAUNCH(0, 2); else POOLV_LAUNCH(0, 4);
            } else {
                if (np == 1) POOLV_LAUNCH(1, 1); else if (np == 2) POOLV_LAUNCH(1, 2); else POOLV_LAUNCH(1, 4);
            }
            SPA_LAUNCH_CHECK();
            return SPA_OK;
        }
    }
    hipLaunchKernelGGL(k_pool_mean, dim3(Ncap), dim3(256), 0, s, fmap, d->dtype, d->C, d->fh, d->fw,
                       (long long)d->stride_b, (long long)d->stride_y, (long long)d->stride_x, B, H,
                       W, sampling, offsets, (const int32_t *)ctx->ws[WS_BBOX], count, cells, labels, centroid,
                       append_pos, X, x_dtype, (long long)ld);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
