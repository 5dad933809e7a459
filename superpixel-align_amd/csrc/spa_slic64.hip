// float64 SLIC for uint8 images: superpixel_overlaps.py:301-304 calls slic(img.transpose(1, 2, 0), n_segments) on
// the ORIGINAL uint8 image, so scikit-image's img_as_float makes it float64 and rgb2lab and the compiled
// _slic_cython[double] run in binary64 (slic_superpixels.py: dtype = image.dtype).  This is the baseline script's
// path, not the labelling hot path: plain kernels, same arithmetic order as the float32 ones, no tuning beyond
// coalesced access.
//
//   k_s64_lab      u8 value -> linear sRGB through a 256-entry table (the deterministic binary64 power, filled by
//                  the workgroup), XYZ, Lab, x ratio; planar float64
//   k_s64_init     grid centres, first search windows
//   k_s64_assign   16 x 16 pixel tile per workgroup: candidate centres compacted into LDS in index order, every
//                  pixel walks them (strict <: first smallest in centre order, as the centre-major loop of the
//                  reference leaves it)
//   k_s64_update   one wave per (image, centre): walks the rows of the window the sweep used, 64 labels per
//                  load; the members' y, x, L, a, b are added in raster order (all lanes carry the five sums:
//                  values broadcast with v_readlane), then centre = sum / count (0/0 = NaN: the seed is dead
//                  from then on, as in scikit-image) and the next window
#include "spa_common.h"

// The kernels are written once for the Cython core's fused type `floating`: T = double is the uint8-image call
// above, T = float is the GENERAL float32 path of the labelling pipeline for images the tuned kernels of
// spa_slic.hip do not take (rows wider than 4 096 pixels: their occupancy masks hold 64 pieces of 64 pixels per
// row) — same arithmetic, same order, plain kernels.
template <typename T>
struct CenG {
    T cy, cx, cl, ca, cb;
    int y0, y1, x0, x1;          // search window [y0, y1) x [x0, x1) of the coming sweep (empty for a dead seed)
    int cnt, pad;
};
typedef CenG<double> Cen64;

template <typename T>
__device__ __forceinline__ void s64_window(CenG<T> &c, int s2y, int s2x, int H, int W)
{
    if (c.cy != c.cy) { c.y0 = c.y1 = c.x0 = c.x1 = 0; return; }
    // y_min = <Py_ssize_t>max(cy - 2 * step_y, 0); y_max = <Py_ssize_t>min(cy + 2 * step_y + 1, height)
    T fy0 = c.cy - (T)s2y; if (!(fy0 > (T)0)) fy0 = (T)0;
    T fy1 = (c.cy + (T)s2y) + (T)1; if (!(fy1 < (T)H)) fy1 = (T)H;
    T fx0 = c.cx - (T)s2x; if (!(fx0 > (T)0)) fx0 = (T)0;
    T fx1 = (c.cx + (T)s2x) + (T)1; if (!(fx1 < (T)W)) fx1 = (T)W;
    c.y0 = (int)fy0; c.y1 = (int)fy1; c.x0 = (int)fx0; c.x1 = (int)fx1;
}

__global__ __launch_bounds__(256) void k_s64_lab(const float *__restrict__ rgb, long long npix, double ratio,
                                                 double *__restrict__ lab, uint32_t *__restrict__ status)
{
    __shared__ double lut[256];
    {
        // img_as_float(uint8): value * (1 / 255) in float64; rgb2xyz companding (colorconv.py)
        const double a = (double)threadIdx.x * (1.0 / 255.0);
        lut[threadIdx.x] = a > 0.04045 ? spa_det_exp(2.4 * spa_det_log_pos((a + 0.055) / 1.055)) : a / 12.92;
    }
    __syncthreads();
    const int b = blockIdx.y;
    const float *src = rgb + (long long)b * 3 * npix;
    double *dst = lab + (long long)b * 3 * npix;
    const double m[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};
    const double white[3] = {0.95047, 1.0, 1.08883};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        double v[3], f[3];
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float p = src[c * npix + i];
            const int q = (int)p;
            bad = bad || !(p >= 0.0f && p <= 255.0f) || (float)q != p;
            v[c] = lut[q & 255];
        }
        if (bad) atomicOr(status, SPA_ST_LABEL_RANGE);        // not an 8-bit image
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double s = m[r][0] * v[0];
            s = s + m[r][1] * v[1];
            s = s + m[r][2] * v[2];
            s = s / white[r];
            f[r] = s > 0.008856 ? spa_det_exp(spa_det_log_pos(s) / 3.0) : 7.787 * s + 16.0 / 116.0;
        }
        const double L = 116.0 * f[1] - 16.0, A = 500.0 * (f[0] - f[1]), Bq = 200.0 * (f[1] - f[2]);
        dst[i] = L * ratio;
        dst[npix + i] = A * ratio;
        dst[2 * npix + i] = Bq * ratio;
    }
}

template <typename T>
__global__ void k_s64_init(CenG<T> *__restrict__ cen, int nC, int grid_nx, int start_y, int start_x, int step_y,
                           int step_x, int s2y, int s2x, int H, int W)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nC) return;
    CenG<T> c;
    c.cy = (T)(start_y + (k / grid_nx) * step_y);
    c.cx = (T)(start_x + (k % grid_nx) * step_x);
    c.cl = c.ca = c.cb = (T)0;
    c.cnt = 0; c.pad = 0;
    s64_window(c, s2y, s2x, H, W);
    cen[(long long)blockIdx.y * nC + k] = c;
}

#define S64_TILE 16
template <typename T>
__global__ __launch_bounds__(256) void k_s64_assign(const T *__restrict__ lab, const CenG<T> *__restrict__ cen,
                                                    int nC, int H, int W, T sw, int32_t *__restrict__ labels,
                                                    uint32_t *__restrict__ status)
{
    __shared__ CenG<T> cand[256];
    __shared__ int cand_k[256];
    __shared__ int wave_cnt[4];
    const int b = blockIdx.z;
    const int ty0 = blockIdx.y * S64_TILE, tx0 = blockIdx.x * S64_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long npix = (long long)H * W;
    const T *pl = lab + (long long)b * 3 * npix;
    const CenG<T> *cb = cen + (long long)b * nC;
    const int y = ty0 + (tid >> 4), x = tx0 + (tid & 15);
    const bool ok = y < H && x < W;
    const long long p = (long long)y * W + x;
    const T pL = ok ? pl[p] : (T)0, pA = ok ? pl[npix + p] : (T)0, pB = ok ? pl[2 * npix + p] : (T)0;
    // distance[...] = DBL_MAX: the largest double, +inf once stored in a float32 array
    T best = sizeof(T) == 8 ? (T)1.7976931348623157e308 : (T)INFINITY;
    int bl = -1;
    const T fy = (T)y, fx = (T)x;
    for (int kb = 0; kb < nC; kb += 256) {
        const int k = kb + tid;
        bool hit = false;
        CenG<T> c;
        if (k < nC) {
            c = cb[k];
            hit = c.y0 < ty0 + S64_TILE && c.y1 > ty0 && c.x0 < tx0 + S64_TILE && c.x1 > tx0;
        }
        const unsigned long long m = __ballot(hit);
        if (lane == 0) wave_cnt[wv] = __popcll(m);
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = wave_cnt[i];
            if (i < wv) off += n;
            total += n;
        }
        if (hit) {
            const int pos = off + (int)spa_rank_in_mask(m);
            cand[pos] = c;
            cand_k[pos] = k;
        }
        __syncthreads();
        for (int j = 0; j < total; ++j) {
            const CenG<T> &e = cand[j];
            if (!(ok && y >= e.y0 && y < e.y1 && x >= e.x0 && x < e.x1)) continue;
            const T ty = e.cy - fy;
            const T dy = ty * ty;
            const T tx = e.cx - fx;
            const T dx = tx * tx;
            T dc = (dy + dx) * sw;
            const T t0 = pL - e.cl, t1 = pA - e.ca, t2 = pB - e.cb;
            T col = t0 * t0;
            col = col + t1 * t1;
            col = col + t2 * t2;
            dc = dc + col;
            if (best > dc) { best = dc; bl = cand_k[j]; }
        }
        __syncthreads();
    }
    if (ok) {
        if (bl < 0) atomicOr(status, SPA_ST_SLIC_UNCOVERED);
        labels[(long long)b * npix + p] = bl;
    }
}

__device__ __forceinline__ double s64_readlane(double v, int l)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ float s64_readlane(float v, int l)
{
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), l));
}

template <typename T>
__global__ __launch_bounds__(256) void k_s64_update(const T *__restrict__ lab, const int32_t *__restrict__ labels,
                                                    CenG<T> *__restrict__ cen, int nC, int total, int H, int W,
                                                    int s2y, int s2x, uint32_t *__restrict__ status)
{
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= total) return;
    const int lane = threadIdx.x & 63;
    const int b = g / nC, k = g - b * nC;
    const long long npix = (long long)H * W;
    const T *pl = lab + (long long)b * 3 * npix;
    const int32_t *L = labels + (long long)b * npix;
    CenG<T> c = cen[g];
    T sy = (T)0, sx = (T)0, sl = (T)0, sa = (T)0, sb = (T)0;
    int cnt = 0;
    for (int y = c.y0; y < c.y1; ++y) {
        const T fy = (T)y;
        for (int xb = c.x0; xb < c.x1; xb += 64) {
            const int x = xb + lane;
            const bool in = x < c.x1;
            const long long p = (long long)y * W + x;
            const bool mem = in && L[p] == k;
            unsigned long long m = __ballot(mem);
            if (!m) continue;
            const T vl = mem ? pl[p] : (T)0, va = mem ? pl[npix + p] : (T)0, vb = mem ? pl[2 * npix + p] : (T)0;
            cnt += __popcll(m);
            while (m) {                                      // raster order: ascending x
                const int j = __ffsll((long long)m) - 1;
                m &= m - 1;
                sy = sy + fy;
                sx = sx + (T)(xb + j);
                sl = sl + s64_readlane(vl, j);
                sa = sa + s64_readlane(va, j);
                sb = sb + s64_readlane(vb, j);
            }
        }
    }
    if (lane == 0) {
        // segments[k, c] /= n_segment_elems[k]: 0/0 = NaN for a seed without pixels
        const T n = (T)cnt;
        c.cy = sy / n; c.cx = sx / n; c.cl = sl / n; c.ca = sa / n; c.cb = sb / n;
        c.cnt = cnt;
        if (cnt == 0) atomicOr(status, SPA_ST_SLIC_EMPTY_SEGMENT);
        s64_window(c, s2y, s2x, H, W);
        cen[g] = c;
    }
}

__global__ void k_s64_export(const Cen64 *__restrict__ cen, double *__restrict__ out, long long total)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const Cen64 c = cen[i];
    double *o = out + i * 6;
    o[0] = (c.cy != c.cy) ? c.cy : 0.0;          // a dead seed's z is 0/0 too
    o[1] = c.cy; o[2] = c.cx; o[3] = c.cl; o[4] = c.ca; o[5] = c.cb;
}

__global__ void k_sg_export_f32(const CenG<float> *__restrict__ cen, float *__restrict__ out, long long total)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const CenG<float> c = cen[i];
    float *o = out + i * 6;
    o[0] = (c.cy != c.cy) ? c.cy : 0.0f;
    o[1] = c.cy; o[2] = c.cx; o[3] = c.cl; o[4] = c.ca; o[5] = c.cb;
}

template <typename T>
static int slic_core_general(spa_ctx *ctx, const T *lab, int32_t B, int32_t H, int32_t W, int32_t n_segments,
                             int32_t max_iter, int32_t *labels, T *centres, void *stream)
{
    spa_slic_plan pl;
    int rc = spa_slic_make_plan(H, W, n_segments, &pl);
    if (rc != SPA_OK) return rc;
    const int nC = pl.n_centroids;
    hipStream_t s = spa_stream(stream);
    CenG<T> *cen;
    rc = spa_ws_reserve(ctx, WS_CENTRES, (size_t)B * nC * sizeof(CenG<T>), (void **)&cen);
    if (rc != SPA_OK) return rc;
    const int s2y = 2 * pl.win_step_y, s2x = 2 * pl.win_step_x;
    hipLaunchKernelGGL(k_s64_init<T>, dim3((nC + 127) / 128, B), dim3(128), 0, s, cen, nC, pl.grid_nx, pl.start_y,
                       pl.start_x, pl.step_y, pl.step_x, s2y, s2x, H, W);
    // cdef floating spatial_weight = 1.0 / (step * step)
    const T sw = (T)(1.0 / ((double)pl.step * (double)pl.step));
    const dim3 ga((W + S64_TILE - 1) / S64_TILE, (H + S64_TILE - 1) / S64_TILE, B);
    const int total = B * nC;
    for (int it = 0; it < max_iter; ++it) {
        hipLaunchKernelGGL(k_s64_assign<T>, ga, dim3(256), 0, s, lab, (const CenG<T> *)cen, nC, H, W, sw, labels,
                           ctx->d_status);
        // (the centroids computed after the last sweep never influence the labels)
        if (it + 1 < max_iter || centres)
            hipLaunchKernelGGL(k_s64_update<T>, dim3((total + 3) / 4), dim3(256), 0, s, lab, (const int32_t *)labels, cen,
                               nC, total, H, W, s2y, s2x, ctx->d_status);
    }
    if (centres) {
        if (sizeof(T) == 8)
            hipLaunchKernelGGL(k_s64_export, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const Cen64 *)cen,
                               (double *)centres, (long long)total);
        else
            hipLaunchKernelGGL(k_sg_export_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                               (const CenG<float> *)cen, (float *)centres, (long long)total);
    }
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// the float32 Cython instantiation on plain kernels: what spa_slic_core falls back to for images its tuned kernels
// do not take (spa_slic.hip); labels and centres bit-identical to them where both apply
int spa_slic_core_general_f32(spa_ctx *ctx, const float *lab, int32_t B, int32_t H, int32_t W, int32_t n_segments,
                              int32_t max_iter, int32_t *labels, float *centres, void *stream)
{
    return slic_core_general<float>(ctx, lab, B, H, W, n_segments, max_iter, labels, centres, stream);
}

// lab (B,3,H,W) float64 planar, already x 1/compactness; centres (B,nC,6) float64 or NULL
extern "C" int spa_slic_core_f64(spa_ctx *ctx, const double *lab, int32_t B, int32_t H, int32_t W,
                                 int32_t n_segments, int32_t max_iter, int32_t *labels, double *centres,
                                 void *stream)
{
    SPA_ARG(ctx && lab && labels && B > 0 && max_iter > 0);
    return slic_core_general<double>(ctx, lab, B, H, W, n_segments, max_iter, labels, centres, stream);
}

// rgb (B,3,H,W) float32 holding the uint8 values 0..255 -> scaled Lab (B,3,H,W) float64
extern "C" int spa_rgb2lab_u8_f64(spa_ctx *ctx, const float *rgb, int32_t B, int32_t H, int32_t W, double ratio,
                                  double *lab, void *stream)
{
    SPA_ARG(ctx && rgb && lab && B > 0 && H > 0 && W > 0);
    const long long npix = (long long)H * W;
    int gx = (int)((npix + 255) / 256);
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(k_s64_lab, dim3(gx, B), dim3(256), 0, spa_stream(stream), rgb, npix, ratio, lab, ctx->d_status);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// whole slic(uint8 image, n_segments) call of superpixel_overlaps.py:303
extern "C" int spa_slic_u8(spa_ctx *ctx, const float *rgb, int32_t B, int32_t H, int32_t W, int32_t n_segments,
                           double compactness, int32_t max_iter, int32_t *labels, int32_t *n_labels, void *stream)
{
    SPA_ARG(ctx && rgb && labels && n_labels && compactness > 0.0);
    spa_slic_plan pl;
    int rc = spa_slic_make_plan(H, W, n_segments, &pl);
    if (rc != SPA_OK) return rc;
    double *lab;
    int32_t *pre;
    const size_t npix = (size_t)H * W;
    rc = spa_ws_reserve(ctx, WS_LAB, (size_t)B * 3 * npix * 8, (void **)&lab);
    if (rc != SPA_OK) return rc;
    rc = spa_ws_reserve(ctx, WS_PRE, (size_t)B * npix * 4, (void **)&pre);
    if (rc != SPA_OK) return rc;
    rc = spa_rgb2lab_u8_f64(ctx, rgb, B, H, W, 1.0 / compactness, lab, stream);
    if (rc != SPA_OK) return rc;
    rc = spa_slic_core_f64(ctx, lab, B, H, W, n_segments, max_iter, pre, nullptr, stream);
    if (rc != SPA_OK) return rc;
    return spa_enforce_connectivity(ctx, pre, B, H, W, pl.min_size, pl.max_size, labels, n_labels, stream);
}
