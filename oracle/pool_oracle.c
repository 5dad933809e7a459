/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see slic_oracle.c header).
 *
 * Plain-C restatements of the per-superpixel descriptor ops of the reference:
 *   orc_create_prior      batch_spalign_kmeans.py:111-129
 *   orc_anchor_pool       batch_spalign_kmeans.py:210-276 (superpixel_align) given the
 *                         anchors that random.shuffle selected (:231-234)
 *   orc_select_anchors    the selection itself: CPython random.shuffle over the
 *                         raster-ordered pixel list of each superpixel (:230-234),
 *                         see rng_oracle.c for the generator
 *   orc_mean_pool         dense per-segment mean, notebooks/Superpixel_Align.ipynb cell 4
 *   orc_segment_stats     counts + centre of mass (scipy.ndimage.center_of_mass, :229)
 * Pinned against the reference functions themselves (imported with stub modules
 * under /opt/conda/bin/python3.9) by tests/golden/pool_*.npz, prior_*.npz.
 *
 * Build: gcc -O2 -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "detmath.h"

/* labels: H*W int32, contiguous ids 0..S-1 (as the reference requires, :194-199) */
void orc_segment_stats(const int32_t *labels, int64_t H, int64_t W, int64_t S,
                       int64_t *count, double *cy, double *cx)
{
    int64_t *sy = (int64_t *)calloc((size_t)S, sizeof(int64_t));
    int64_t *sx = (int64_t *)calloc((size_t)S, sizeof(int64_t));
    memset(count, 0, (size_t)S * sizeof(int64_t));
    for (int64_t y = 0; y < H; ++y)
        for (int64_t x = 0; x < W; ++x) {
            int32_t s = labels[y * W + x];
            count[s] += 1; sy[s] += y; sx[s] += x;
        }
    for (int64_t s = 0; s < S; ++s) {
        /* center_of_mass: sum(mask * grid) / sum(mask); integer sums are exact */
        cy[s] = (double)sy[s] / (double)count[s];
        cx[s] = (double)sx[s] / (double)count[s];
    }
    free(sy); free(sx);
}

/* create_prior (:111-129).  exp() is the deterministic det_exp of detmath.h so that
   the HIP twin can be compared bit for bit; against numpy's exp the result agrees to
   ~1e-15 relative (checked by the golden test). The per-segment mean is a raster-order
   sequential float64 sum (numpy uses pairwise summation: same to ~1e-15). */
void orc_create_prior(const int32_t *labels, int64_t H, int64_t W, int64_t S,
                      double y_rel_pos, double x_rel_pos, double y_rel_sigma, double x_rel_sigma,
                      double *out)
{
    int64_t ymean = (int64_t)((double)H * y_rel_pos);
    int64_t xmean = (int64_t)((double)W * x_rel_pos);
    double ys = (double)H * y_rel_sigma, xs = (double)W * x_rel_sigma;
    double dy2 = (2.0 * ys) * (2.0 * ys), dx2 = (2.0 * xs) * (2.0 * xs);
    double *sum = (double *)calloc((size_t)S, sizeof(double));
    int64_t *cnt = (int64_t *)calloc((size_t)S, sizeof(int64_t));
    for (int64_t y = 0; y < H; ++y) {
        double ty = (double)((y - ymean) * (y - ymean)) / dy2;
        for (int64_t x = 0; x < W; ++x) {
            double tx = (double)((x - xmean) * (x - xmean)) / dx2;
            double w = det_exp(-(ty + tx));
            int32_t s = labels[y * W + x];
            sum[s] += w; cnt[s] += 1;
        }
    }
    for (int64_t s = 0; s < S; ++s) out[s] = sum[s] / (double)cnt[s];
    free(sum); free(cnt);
}

/* One anchor: 4 nearest feature-pixel centres (canonical tie-break: lowest flat index
   n = x*fh + y, i.e. a stable argsort of the reference's x-major flat list, :219-221,
   :244-245), their bounding box, and the box-corner "bilinear" blend (:247-266).
   py, px: anchor already mapped to feature coordinates and clipped (:235-240).
   Outputs the corner cell indices and the four float32 weights (the float64 scalar
   products are narrowed to float32 before they touch the float32 feature vectors:
   numpy<2 value-based casting), plus the float32 normaliser. */
typedef struct { int y0, y1, x0, x1; float w11, w12, w21, w22, inv; } orc_anchor_geom;

static void anchor_geometry(double py, double px, int fh, int fw, int n_neighbor, orc_anchor_geom *g)
{
    /* candidates: the 4 nearest centres of a unit grid lie within +-2 cells */
    int cyi = (int)floor(py), cxi = (int)floor(px);
    double best_d[16]; int best_y[16], best_x[16]; long best_n[16]; int nb = 0;
    if (n_neighbor > 16) n_neighbor = 16;
    for (int x = cxi - 3; x <= cxi + 3; ++x) {
        if (x < 0 || x >= fw) continue;
        for (int y = cyi - 3; y <= cyi + 3; ++y) {
            if (y < 0 || y >= fh) continue;
            double ddy = ((double)y + 0.5) - py, ddx = ((double)x + 0.5) - px;
            double d = sqrt(ddy * ddy + ddx * ddx);
            long n = (long)x * fh + y;
            /* insertion into the sorted best list (ascending d, then ascending n) */
            int pos = nb;
            while (pos > 0 && (best_d[pos - 1] > d || (best_d[pos - 1] == d && best_n[pos - 1] > n))) --pos;
            if (pos >= n_neighbor) continue;
            int last = nb < n_neighbor ? nb : n_neighbor - 1;
            for (int j = last; j > pos; --j) {
                best_d[j] = best_d[j - 1]; best_y[j] = best_y[j - 1];
                best_x[j] = best_x[j - 1]; best_n[j] = best_n[j - 1];
            }
            best_d[pos] = d; best_y[pos] = y; best_x[pos] = x; best_n[pos] = n;
            if (nb < n_neighbor) ++nb;
        }
    }
    int y0 = best_y[0], y1 = best_y[0], x0 = best_x[0], x1 = best_x[0];
    for (int j = 1; j < nb; ++j) {
        if (best_y[j] < y0) y0 = best_y[j];
        if (best_y[j] > y1) y1 = best_y[j];
        if (best_x[j] < x0) x0 = best_x[j];
        if (best_x[j] > x1) x1 = best_x[j];
    }
    double min_y = y0 + 0.5, max_y = y1 + 0.5, min_x = x0 + 0.5, max_x = x1 + 0.5;
    g->y0 = y0; g->y1 = y1; g->x0 = x0; g->x1 = x1;
    g->w11 = (float)((max_x - px) * (max_y - py));
    g->w12 = (float)((max_x - px) * (py - min_y));
    g->w21 = (float)((px - min_x) * (max_y - py));
    g->w22 = (float)((px - min_x) * (py - min_y));
    g->inv = (float)(1.0 / ((max_x - min_x) * (max_y - min_y)));
}

/*
 * superpixel_align for one image given the selected anchor pixels.
 *   fmap     : feature map, element (c, y, x) at fmap[c*sc + y*sy + x*sx] (strides in elements)
 *   anchors  : S * n_anchors * 2 int32 (y, x) image-pixel coordinates, in selection order
 *   n_valid  : S int32 — min(n_anchors, pixels in the superpixel)
 *   cy, cx   : S float64 centre of mass (only read when append_pos)
 *   out      : S * (C + 2*append_pos) float64; without_pos rows hold float32 values
 *              (the reference then returns a float32 array; widened here for one ABI)
 */
void orc_anchor_pool(const float *fmap, int64_t C, int64_t fh, int64_t fw,
                     int64_t sc, int64_t sy, int64_t sx,
                     int64_t img_h, int64_t S, int64_t n_anchors, int64_t n_neighbor,
                     const int32_t *anchors, const int32_t *n_valid,
                     const double *cy, const double *cx, int append_pos, double *out)
{
    /* feature_ratio = float(feature_map_h) / img_h  — the y ratio serves both axes (:215,:235) */
    double ratio = (double)fh / (double)img_h;
    int64_t D = C + (append_pos ? 2 : 0);
    float *fp = (float *)malloc((size_t)C * sizeof(float));
    double *acc = (double *)malloc((size_t)C * sizeof(double));
    float *acc32 = (float *)malloc((size_t)C * sizeof(float));
    for (int64_t s = 0; s < S; ++s) {
        int nv = n_valid[s];
        for (int64_t c = 0; c < C; ++c) { acc[c] = 0.0; acc32[c] = 0.0f; }
        for (int a = 0; a < nv; ++a) {
            double py = (double)anchors[(s * n_anchors + a) * 2 + 0] * ratio + 0.5;
            double px = (double)anchors[(s * n_anchors + a) * 2 + 1] * ratio + 0.5;
            double hi_y = (double)(fh - 1) + 0.5, hi_x = (double)(fw - 1) + 0.5;
            if (py < 0.0) py = 0.0; if (py > hi_y) py = hi_y;
            if (px < 0.0) px = 0.0; if (px > hi_x) px = hi_x;
            orc_anchor_geom g;
            anchor_geometry(py, px, (int)fh, (int)fw, (int)n_neighbor, &g);
            for (int64_t c = 0; c < C; ++c) {
                const float *f = fmap + c * sc;
                float f11 = f[g.y0 * sy + g.x0 * sx], f12 = f[g.y1 * sy + g.x0 * sx];
                float f21 = f[g.y0 * sy + g.x1 * sx], f22 = f[g.y1 * sy + g.x1 * sx];
                float v = g.w11 * f11;
                v = v + g.w12 * f12;
                v = v + g.w21 * f21;
                v = v + g.w22 * f22;
                v = g.inv * v;
                fp[c] = v;
            }
            if (append_pos) { for (int64_t c = 0; c < C; ++c) acc[c] += (double)fp[c]; }
            else { for (int64_t c = 0; c < C; ++c) acc32[c] += fp[c]; }
        }
        if (append_pos) {
            for (int64_t c = 0; c < C; ++c) out[s * D + c] = acc[c] / (double)nv;
            /* mean of nv identical centroids: (c + c + ...)/nv — sequential sum then divide */
            double ay = 0.0, ax = 0.0;
            for (int a = 0; a < nv; ++a) { ay += cy[s]; ax += cx[s]; }
            out[s * D + C] = ay / (double)nv;
            out[s * D + C + 1] = ax / (double)nv;
        } else {
            for (int64_t c = 0; c < C; ++c) out[s * D + c] = (double)(acc32[c] / (float)nv);
        }
    }
    free(fp); free(acc); free(acc32);
}

/*
 * Dense per-segment mean (mean mode; notebook cell 4: resize the feature map to the
 * image size, then average it over the pixels of each superpixel).
 *   mode 0 "nearest" : the feature pixel under image pixel (y, x) is (y*fh//H, x*fw//W);
 *   mode 1 "bilinear": chainer F.resize_images sampling — corners aligned,
 *                      u = y*(fh-1)/(H-1), 4-tap weights in float32.
 * Definition of the arithmetic (shared with the HIP path so the two can be compared
 * bit for bit): for every (segment, feature pixel) pair a weight Wt = sum of the taps of
 * the segment's pixels on that feature pixel (integer count for nearest; float32 sum in
 * raster order of the image pixels for bilinear); then
 *   out[s][c] = ( sum over feature pixels in raster order of Wt * F[c][cell] ) / total_s
 * evaluated in float32, multiply and add rounded separately, total_s = float32(count_s).
 * Returns 0 (a feature pixel may be touched by any number of segments).
 */
typedef struct { int32_t n, cap; int32_t *lab; float *w; } orc_cell;

int orc_mean_pool(const float *fmap, int64_t C, int64_t fh, int64_t fw,
                  int64_t sc, int64_t sy, int64_t sx,
                  const int32_t *labels, int64_t H, int64_t W, int64_t S, int mode,
                  float *out)
{
    int64_t ncell = fh * fw;
    /* per feature pixel: the (segment, weight) pairs in order of first appearance, as many as there are */
    orc_cell *cellv = (orc_cell *)calloc((size_t)ncell, sizeof(orc_cell));
    int64_t *count = (int64_t *)calloc((size_t)S, sizeof(int64_t));
    int rc = 0;
    for (int64_t i = 0; i < H * W; ++i) count[labels[i]] += 1;
    /* cell-major accumulation in raster order of the image pixels */
    for (int64_t y = 0; y < H; ++y)
        for (int64_t x = 0; x < W; ++x) {
            int32_t s = labels[y * W + x];
            int64_t cells[4]; float taps[4]; int nt;
            if (mode == 0) {
                cells[0] = (y * fh / H) * fw + (x * fw / W); taps[0] = 1.0f; nt = 1;
            } else {
                float u = (H > 1) ? (float)y * ((float)(fh - 1) / (float)(H - 1)) : 0.0f;
                float v = (W > 1) ? (float)x * ((float)(fw - 1) / (float)(W - 1)) : 0.0f;
                int64_t u0 = (int64_t)u, v0 = (int64_t)v;
                if (u0 > fh - 1) u0 = fh - 1;
                if (v0 > fw - 1) v0 = fw - 1;
                int64_t u1 = u0 + 1 < fh ? u0 + 1 : fh - 1;
                int64_t v1 = v0 + 1 < fw ? v0 + 1 : fw - 1;
                float fu = u - (float)u0, fv = v - (float)v0;
                cells[0] = u0 * fw + v0; taps[0] = (1.0f - fu) * (1.0f - fv);
                cells[1] = u0 * fw + v1; taps[1] = (1.0f - fu) * fv;
                cells[2] = u1 * fw + v0; taps[2] = fu * (1.0f - fv);
                cells[3] = u1 * fw + v1; taps[3] = fu * fv;
                nt = 4;
            }
            for (int t = 0; t < nt; ++t) {
                orc_cell *cl = cellv + cells[t];
                int j = 0, n = cl->n;
                while (j < n && cl->lab[j] != s) ++j;
                if (j == n) {
                    if (n == cl->cap) {
                        cl->cap = cl->cap ? 2 * cl->cap : 8;
                        cl->lab = (int32_t *)realloc(cl->lab, (size_t)cl->cap * sizeof(int32_t));
                        cl->w = (float *)realloc(cl->w, (size_t)cl->cap * sizeof(float));
                    }
                    cl->lab[n] = s; cl->w[n] = 0.0f; cl->n = n + 1;
                }
                cl->w[j] += taps[t];
            }
        }
    memset(out, 0, (size_t)S * C * sizeof(float));
    for (int64_t cell = 0; cell < ncell; ++cell) {
        int64_t y = cell / fw, x = cell % fw;
        for (int j = 0; j < cellv[cell].n; ++j) {
            int32_t s = cellv[cell].lab[j];
            float w = cellv[cell].w[j];
            float *o = out + (int64_t)s * C;
            for (int64_t c = 0; c < C; ++c) {
                float prod = w * fmap[c * sc + y * sy + x * sx];
                o[c] = o[c] + prod;
            }
        }
    }
    for (int64_t s = 0; s < S; ++s) {
        float tot = (float)count[s];
        for (int64_t c = 0; c < C; ++c) out[s * C + c] = out[s * C + c] / tot;
    }
    for (int64_t cell = 0; cell < ncell; ++cell) { free(cellv[cell].lab); free(cellv[cell].w); }
    free(cellv); free(count);
    return rc;
}
