/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see slic_oracle.c header).
 *
 * Plain-C restatement of the felzenszwalb branch of batch_superpixel:
 *   reference call site : batch_spalign_kmeans.py:301-307
 *       felzenszwalb(img.transpose(1, 2, 0) / 255., scale, sigma, min_size)
 *   the algorithm lives in scikit-image (not vendored; 0.18.3 here), Cython core
 *   skimage/segmentation/_felzenszwalb_cy (compiled only) + scipy.ndimage.gaussian_filter.
 * Published algorithm restated: Gaussian smoothing (separable, radius int(4*sigma+.5), 'reflect'
 * borders, symmetric correlate1d summation order), 8-connectivity edge costs (Euclidean colour
 * distance, float64), edges sorted by cost, one greedy pass of Felzenszwalb-Huttenlocher merging
 * on a union-find forest whose root is the smallest pixel index (thresholds rounded to float32:
 * the Cython core declares them `cdef float`), a second pass merging components below min_size,
 * labels = rank of the root among the sorted roots (np.unique).
 *
 * Pinned by tests/golden/fz_*.npz (oracle/gen_golden_fz.py): the Cython core is run with
 * `ndi.gaussian_filter` and `np.argsort` proxied, so that (a) its input after smoothing is the
 * oracle's own smoothed image and (b) equal costs are ordered by edge index (numpy's default
 * introsort / AVX-512 sort leave tie order unspecified) — everything after that is bit exact.
 * The smoothing itself is pinned against scipy to 1e-15 relative (weights through np.exp) and
 * bit-exactly given scipy's weights.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "detmath.h"

/* scipy.ndimage._gaussian_kernel1d(sigma, 0, radius), radius = int(4.0 * sigma + 0.5) */
int orc_fz_gauss_weights(double sigma, double *w, int cap)
{
    int r = (int)(4.0 * sigma + 0.5);
    if (2 * r + 1 > cap) return -1;
    double sigma2 = sigma * sigma, sum = 0.0;
    for (int i = -r; i <= r; ++i) {
        w[i + r] = det_exp(-0.5 / sigma2 * (double)(i * i));
    }
    /* numpy pairwise sum of 2r+1 < 8 values is sequential; larger kernels: block of 8 rule */
    if (2 * r + 1 < 8) { for (int i = 0; i < 2 * r + 1; ++i) sum += w[i]; }
    else {
        double acc[8]; int n = 2 * r + 1, i;
        for (i = 0; i < 8; ++i) acc[i] = w[i];
        for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; ++j) acc[j] += w[i + j];
        sum = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        for (; i < n; ++i) sum += w[i];
    }
    for (int i = 0; i < 2 * r + 1; ++i) w[i] = w[i] / sum;
    return r;
}

static inline int64_t reflect(int64_t i, int64_t n)
{
    /* scipy 'reflect': d c b a | a b c d | d c b a */
    if (n == 1) return 0;
    int64_t period = 2 * n;
    i = i % period; if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

/* one separable pass along `axis` (0: rows / y, 1: columns / x) of an (H, W, C) float64 image,
   NI_Correlate1D symmetric branch: tmp = x[c]*w[0]; for j = -r..-1: tmp += (x[c+j] + x[c-j]) * w[j] */
static void blur_axis(const double *in, int64_t H, int64_t W, int64_t C, int axis,
                      const double *w, int r, double *out)
{
    const double *wc = w + r;
    for (int64_t y = 0; y < H; ++y)
        for (int64_t x = 0; x < W; ++x)
            for (int64_t c = 0; c < C; ++c) {
                double tmp = in[(y * W + x) * C + c] * wc[0];
                for (int j = -r; j < 0; ++j) {
                    double a, b;
                    if (axis == 0) {
                        a = in[(reflect(y + j, H) * W + x) * C + c];
                        b = in[(reflect(y - j, H) * W + x) * C + c];
                    } else {
                        a = in[(y * W + reflect(x + j, W)) * C + c];
                        b = in[(y * W + reflect(x - j, W)) * C + c];
                    }
                    tmp += (a + b) * wc[j];
                }
                out[(y * W + x) * C + c] = tmp;
            }
}

void orc_fz_blur(const double *img, int64_t H, int64_t W, int64_t C, const double *w, int r, double *out)
{
    double *tmp = (double *)malloc((size_t)H * W * C * sizeof(double));
    blur_axis(img, H, W, C, 0, w, r, tmp);
    blur_axis(tmp, H, W, C, 1, w, r, out);
    free(tmp);
}

/* edges in the order of the Cython core: right, down, down-right, up-right; each edge is
   (first, second) as np.c_[...] lists them.  costs = sqrt(sum_c d_c*d_c), channel sum sequential */
int64_t orc_fz_edges(const double *img, int64_t H, int64_t W, int64_t C, double *costs, int64_t *edges)
{
    int64_t n = 0;
#define EDGE(ya, xa, yb, xb)                                                            \
    do {                                                                                \
        double s = 0.0;                                                                 \
        for (int64_t c = 0; c < C; ++c) {                                               \
            double d = img[((ya) * W + (xa)) * C + c] - img[((yb) * W + (xb)) * C + c]; \
            s += d * d;                                                                 \
        }                                                                               \
        costs[n] = sqrt(s);                                                             \
        edges[2 * n] = (ya) * W + (xa); edges[2 * n + 1] = (yb) * W + (xb);             \
        ++n;                                                                            \
    } while (0)
    for (int64_t y = 0; y < H; ++y) for (int64_t x = 1; x < W; ++x) EDGE(y, x, y, x - 1);          /* right  */
    for (int64_t y = 1; y < H; ++y) for (int64_t x = 0; x < W; ++x) EDGE(y, x, y - 1, x);          /* down   */
    for (int64_t y = 1; y < H; ++y) for (int64_t x = 1; x < W; ++x) EDGE(y, x, y - 1, x - 1);      /* dright */
    for (int64_t y = 0; y < H - 1; ++y) for (int64_t x = 1; x < W; ++x) EDGE(y, x, y + 1, x - 1);  /* uright */
#undef EDGE
    return n;
}

static inline int64_t find_root(const int64_t *f, int64_t n)
{
    int64_t root = n;
    while (f[root] < root) root = f[root];
    return root;
}
static inline void set_root(int64_t *f, int64_t n, int64_t root)
{
    while (f[n] < n) { int64_t j = f[n]; f[n] = root; n = j; }
    f[n] = root;
}
static inline void join_trees(int64_t *f, int64_t n, int64_t m)
{
    if (n != m) {
        int64_t root = find_root(f, n), root_m = find_root(f, m);
        if (root > root_m) root = root_m;
        set_root(f, n, root);
        set_root(f, m, root);
    }
}

/* the two greedy passes over the cost-sorted edges + np.unique relabelling.
   order: permutation of the edges (argsort of costs).  Returns the number of labels. */
int64_t orc_fz_segment(const double *costs, const int64_t *edges, const int64_t *order, int64_t n_edges,
                       int64_t npix, double scale, int64_t min_size, int64_t *labels)
{
    int64_t *forest = (int64_t *)malloc((size_t)npix * sizeof(int64_t));
    int64_t *size = (int64_t *)malloc((size_t)npix * sizeof(int64_t));
    double *cint = (double *)calloc((size_t)npix, sizeof(double));
    for (int64_t i = 0; i < npix; ++i) { forest[i] = i; size[i] = 1; }
    for (int64_t e = 0; e < n_edges; ++e) {
        int64_t k = order[e];
        int64_t seg0 = find_root(forest, edges[2 * k]), seg1 = find_root(forest, edges[2 * k + 1]);
        if (seg0 == seg1) continue;
        /* cdef float inner_cost0, inner_cost1 */
        float inner0 = (float)(cint[seg0] + scale / (double)size[seg0]);
        float inner1 = (float)(cint[seg1] + scale / (double)size[seg1]);
        float m = inner0 < inner1 ? inner0 : inner1;
        if (costs[k] < (double)m) {
            join_trees(forest, seg0, seg1);
            int64_t seg_new = find_root(forest, seg0);
            size[seg_new] = size[seg0] + size[seg1];
            cint[seg_new] = costs[k];
        }
    }
    for (int64_t e = 0; e < n_edges; ++e) {
        int64_t k = order[e];
        int64_t seg0 = find_root(forest, edges[2 * k]), seg1 = find_root(forest, edges[2 * k + 1]);
        if (seg0 == seg1) continue;
        if (size[seg0] < min_size || size[seg1] < min_size) {
            join_trees(forest, seg0, seg1);
            int64_t seg_new = find_root(forest, seg0);
            size[seg_new] = size[seg0] + size[seg1];
        }
    }
    /* flat = forest followed to the roots; labels = rank of the root among the sorted roots */
    int64_t *rank = (int64_t *)malloc((size_t)npix * sizeof(int64_t));
    int64_t nl = 0;
    for (int64_t i = 0; i < npix; ++i) rank[i] = (forest[i] == i) ? nl++ : -1;
    for (int64_t i = 0; i < npix; ++i) labels[i] = rank[find_root(forest, i)];
    free(forest); free(size); free(cint); free(rank);
    return nl;
}

typedef struct { double c; int64_t i; } fz_key;
static int fz_cmp(const void *a, const void *b)
{
    const fz_key *x = (const fz_key *)a, *y = (const fz_key *)b;
    if (x->c < y->c) return -1;
    if (x->c > y->c) return 1;
    return (x->i > y->i) - (x->i < y->i);       /* canonical tie order: edge index */
}

/* whole call as the reference makes it: rgb (3,H,W) float32 0..255 -> labels (H,W) int64 */
int64_t orc_felzenszwalb(const float *rgb_chw, int64_t H, int64_t W, double scale, double sigma,
                         int64_t min_size, int64_t *labels)
{
    const int64_t C = 3, npix = H * W;
    double *img = (double *)malloc((size_t)npix * C * sizeof(double));
    double *sm = (double *)malloc((size_t)npix * C * sizeof(double));
    /* img / 255. is evaluated in float32 (float32 array / python float), then img_as_float64 */
    for (int64_t p = 0; p < npix; ++p)
        for (int64_t c = 0; c < C; ++c) img[p * C + c] = (double)(rgb_chw[c * npix + p] / 255.0f);
    double w[64];
    int r = orc_fz_gauss_weights(sigma, w, 64);
    if (r < 0) { free(img); free(sm); return -1; }
    orc_fz_blur(img, H, W, C, w, r, sm);
    int64_t cap = 4 * npix;
    double *costs = (double *)malloc((size_t)cap * sizeof(double));
    int64_t *edges = (int64_t *)malloc((size_t)cap * 2 * sizeof(int64_t));
    int64_t n = orc_fz_edges(sm, H, W, C, costs, edges);
    fz_key *keys = (fz_key *)malloc((size_t)n * sizeof(fz_key));
    for (int64_t i = 0; i < n; ++i) { keys[i].c = costs[i]; keys[i].i = i; }
    qsort(keys, (size_t)n, sizeof(fz_key), fz_cmp);
    int64_t *order = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) order[i] = keys[i].i;
    /* scale = float(scale) / 255. */
    int64_t nl = orc_fz_segment(costs, edges, order, n, npix, scale / 255.0, min_size, labels);
    free(img); free(sm); free(costs); free(edges); free(keys); free(order);
    return nl;
}
