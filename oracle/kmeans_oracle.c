/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see slic_oracle.c header).
 *
 * Plain-C restatements of
 *   orc_kmeans        batch_spalign_kmeans.py:136-183 (kmeans) + :132-133 (weighted_average)
 *   orc_paint         batch_spalign_kmeans.py:186-207 (weighted_kmeans paint loop)
 *   orc_confusion     batch_spalign_kmeans.py:398-405 (chainercv confusion / IoU)
 *   MT19937 + CPython random.shuffle + numpy legacy RandomState.shuffle
 *                     (:33-34 seeds, :148 xp.random.shuffle, :232 random.shuffle)
 * Pinned by tests/golden/kmeans_*.npz, rng_*.npz generated from the reference
 * functions / CPython / numpy themselves.
 *
 * Build: gcc -O2 -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ----------------------------------------------------------------------- */
/* MT19937                                                                  */
/* ----------------------------------------------------------------------- */
typedef struct { uint32_t mt[624]; int idx; } orc_mt;

static void mt_init_genrand(orc_mt *s, uint32_t seed)
{
    s->mt[0] = seed;
    for (int i = 1; i < 624; ++i)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}

static void mt_init_by_array(orc_mt *s, const uint32_t *key, int klen)
{
    mt_init_genrand(s, 19650218u);
    int i = 1, j = 0;
    int k = 624 > klen ? 624 : klen;
    for (; k; --k) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        ++i; ++j;
        if (i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
        if (j >= klen) j = 0;
    }
    for (k = 623; k; --k) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        ++i;
        if (i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
    }
    s->mt[0] = 0x80000000u;
    s->idx = 624;
}

static uint32_t mt_next(orc_mt *s)
{
    if (s->idx >= 624) {
        uint32_t *mt = s->mt;
        for (int k = 0; k < 624; ++k) {
            uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* opaque state handles for ctypes */
void *orc_mt_new(void) { return calloc(1, sizeof(orc_mt)); }
void orc_mt_free(void *s) { free(s); }
/* CPython random.seed(int): init_by_array over the 32-bit words of abs(seed) */
void orc_mt_seed_python(void *s, uint64_t seed)
{
    uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
    mt_init_by_array((orc_mt *)s, key, key[1] ? 2 : 1);
}
/* numpy np.random.seed(int): init_genrand */
void orc_mt_seed_numpy(void *s, uint32_t seed) { mt_init_genrand((orc_mt *)s, seed); }

/* CPython Random._randbelow_with_getrandbits: k = n.bit_length(); r = getrandbits(k)
   (top k bits of one 32-bit output for k <= 32); redraw while r >= n. */
static uint32_t py_randbelow(orc_mt *s, uint32_t n)
{
    int k = 32 - __builtin_clz(n);
    uint32_t r = mt_next(s) >> (32 - k);
    while (r >= n) r = mt_next(s) >> (32 - k);
    return r;
}

/* random.shuffle(x) for len(x) == n, then x[:n_select]: returns, for the first
   n_select positions of the shuffled list, the ORIGINAL index of the element that
   ends up there.  (for i in reversed(range(1, n)): j = randbelow(i+1); swap(i, j)) */
void orc_py_shuffle_select(void *state, int64_t n, int64_t n_select, int64_t *picked)
{
    orc_mt *s = (orc_mt *)state;
    int64_t *perm = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) perm[i] = i;
    for (int64_t i = n - 1; i >= 1; --i) {
        int64_t j = (int64_t)py_randbelow(s, (uint32_t)(i + 1));
        int64_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
    for (int64_t a = 0; a < n_select && a < n; ++a) picked[a] = perm[a];
    free(perm);
}

/* numpy legacy RandomState.shuffle on a 1-D int64 array (in place):
   for i in reversed(range(1, n)): j = random_interval(i); swap.  random_interval uses
   the smallest all-ones mask >= max and rejects masked 32-bit draws above max. */
void orc_np_shuffle_i64(void *state, int64_t *a, int64_t n)
{
    orc_mt *s = (orc_mt *)state;
    for (int64_t i = n - 1; i >= 1; --i) {
        uint64_t max = (uint64_t)i, mask = max, value;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
        mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        if (max <= 0xffffffffULL) {
            while ((value = ((uint64_t)mt_next(s) & mask)) > max) ;
        } else {
            do {
                uint64_t hi = mt_next(s); uint64_t lo = mt_next(s);
                value = ((hi << 32) | lo) & mask;
            } while (value > max);
        }
        int64_t t = a[i]; a[i] = a[value]; a[value] = t;
    }
}

/* ----------------------------------------------------------------------- */
/* numpy pairwise summation (add.reduce along a contiguous axis):           */
/* result = identity 0 + pairwise(a[0:n]) — probed against numpy 1.26.4 and */
/* 2.2.6 add.reduce on 1-D and last-axis reductions, and pinned by the      */
/* near-tie fixtures tests/golden/kmeans_tie.npz                            */
/* ----------------------------------------------------------------------- */
static double pairwise_sum(const double *a, int64_t n)
{
    if (n < 8) {
        double r = 0.0;
        for (int64_t i = 0; i < n; ++i) r += a[i];
        return r;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
    }
}
static double np_sum(const double *a, int64_t n)
{
    if (n == 0) return 0.0;
    return pairwise_sum(a, n);
}

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* float32 twin of pairwise_sum (numpy FLOAT_pairwise_sum: same blocking, float accumulators) */
static float pairwise_sum_f(const float *a, int64_t n)
{
    if (n < 8) {
        float r = 0.0f;
        for (int64_t i = 0; i < n; ++i) r += a[i];
        return r;
    } else if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum_f(a, n2) + pairwise_sum_f(a + n2, n - n2);
    }
}

/*
 * kmeans(k, X, weights, n_iter=1000)   (:136-183)
 *   X (N, D) row-major in T = float64 (descriptors with the position appended) or float32
 *   (--without_pos), w (N) float64.  The arithmetic follows numpy's for that dtype:
 *     - initial centres X[assign == i].mean(axis=0): sequential axis-0 sum IN T, divided by the
 *       count in T;
 *     - distances linalg.norm(X[:,None,:] - centers[None], axis=2): difference, square, pairwise
 *       add.reduce and sqrt all IN T;
 *     - update (X[m] * wj[m,None]).sum(0) / wj[m].sum(): products and sequential axis-0 sum in
 *       float64 (float32 * float64 promotes), float64 pairwise weight sum, float64 division, and
 *       the result is stored into the T centre array (rounded to float32 when T is float32).
 *   init_other: the shuffled `idx` vector for the points with w <= threshold (:147-149);
 *               NULL means k == 2 semantics (all ones).  Callers that emulate numpy's
 *               global RNG build it with orc_np_shuffle_i64.
 *   assign (N) int32 out; returns number of loop iterations executed (the iteration that
 *   detects convergence counts), status: 0 converged, 1 hit n_iter, 2 stopped on empty cluster.
 */
#define ORC_DEFINE_KMEANS(NAME, T, PWSUM, SQRT)                                                   \
int64_t NAME(int64_t k, const T *X, int64_t N, int64_t D, const double *w,                        \
             const int64_t *init_other, int64_t n_iter, int32_t *assign, int32_t *status)         \
{                                                                                                 \
    double *sorted = (double *)malloc((size_t)N * sizeof(double));                                \
    memcpy(sorted, w, (size_t)N * sizeof(double));                                                \
    qsort(sorted, (size_t)N, sizeof(double), cmp_double);                                         \
    double thr = sorted[N / 2];                                                                   \
    free(sorted);                                                                                 \
    int64_t m = 0;                                                                                \
    for (int64_t i = 0; i < N; ++i) {                                                             \
        if (w[i] > thr) assign[i] = 0;                                                            \
        else { assign[i] = init_other ? (int32_t)init_other[m] : (int32_t)(m % (k - 1) + 1); ++m; } \
    }                                                                                             \
    T *centers = (T *)malloc((size_t)k * D * sizeof(T));                                          \
    double *acc = (double *)malloc((size_t)D * sizeof(double));                                   \
    T *tmp = (T *)malloc((size_t)D * sizeof(T));                                                  \
    double *tw = (double *)malloc((size_t)N * sizeof(double));                                    \
    int32_t *na = (int32_t *)malloc((size_t)N * sizeof(int32_t));                                 \
    for (int64_t c = 0; c < k; ++c) {                                                             \
        int64_t cnt = 0;                                                                          \
        for (int64_t d = 0; d < D; ++d) centers[c * D + d] = (T)0;                                \
        for (int64_t i = 0; i < N; ++i) if (assign[i] == c) {                                     \
            for (int64_t d = 0; d < D; ++d) centers[c * D + d] = centers[c * D + d] + X[i * D + d]; \
            ++cnt;                                                                                \
        }                                                                                         \
        for (int64_t d = 0; d < D; ++d)                                                           \
            centers[c * D + d] = cnt ? centers[c * D + d] / (T)cnt : (T)NAN;                      \
    }                                                                                             \
    int64_t it = 0; *status = 1;                                                                  \
    for (; it < n_iter; ) {                                                                       \
        ++it;                                                                                     \
        int same = 1;                                                                             \
        for (int64_t i = 0; i < N; ++i) {                                                         \
            int best = 0; T bd = (T)0;                                                            \
            for (int64_t c = 0; c < k; ++c) {                                                     \
                for (int64_t d = 0; d < D; ++d) { T t = X[i * D + d] - centers[c * D + d]; tmp[d] = t * t; } \
                T dist = SQRT(PWSUM(tmp, D));                                                     \
                /* np.argmin: first minimum; a NaN is "smaller" than everything (first NaN wins) */ \
                if (c == 0) { best = 0; bd = dist; }                                              \
                else if (!isnan(bd) && (isnan(dist) || dist < bd)) { best = (int)c; bd = dist; }  \
            }                                                                                     \
            na[i] = best;                                                                         \
            if (na[i] != assign[i]) same = 0;                                                     \
        }                                                                                         \
        if (same) { *status = 0; break; }                                                         \
        memcpy(assign, na, (size_t)N * sizeof(int32_t));                                          \
        /* centers[j] = (X[m] * wj[m,None]).sum(0) / wj[m].sum(),  w0 = w, wj = 1 - w */          \
        int empty = 0;                                                                            \
        for (int64_t c = 0; c < k; ++c) {                                                         \
            int64_t cnt = 0;                                                                      \
            for (int64_t d = 0; d < D; ++d) acc[d] = 0.0;                                         \
            for (int64_t i = 0; i < N; ++i) if (assign[i] == c) {                                 \
                double wi = (c == 0) ? w[i] : 1.0 - w[i];                                         \
                for (int64_t d = 0; d < D; ++d) acc[d] += (double)X[i * D + d] * wi;              \
                tw[cnt++] = wi;                                                                   \
            }                                                                                     \
            double ws = np_sum(tw, cnt);                                                          \
            for (int64_t d = 0; d < D; ++d) centers[c * D + d] = (T)(acc[d] / ws);                \
            if (cnt == 0) empty = 1;                                                              \
        }                                                                                         \
        if (empty) { *status = 2; break; }                                                        \
    }                                                                                             \
    free(centers); free(acc); free(tmp); free(tw); free(na);                                      \
    return it;                                                                                    \
}

ORC_DEFINE_KMEANS(orc_kmeans, double, np_sum, sqrt)
ORC_DEFINE_KMEANS(orc_kmeans_f32, float, pairwise_sum_f, sqrtf)

/* paint (:193-199): cluster[p] = assign[offset + labels[p]]; road = cluster == 0 */
void orc_paint(const int32_t *labels, int64_t npix, const int32_t *assign_img,
               uint8_t *cluster, uint8_t *road)
{
    for (int64_t i = 0; i < npix; ++i) {
        int32_t c = assign_img[labels[i]];
        cluster[i] = (uint8_t)c; road[i] = (c == 0);
    }
}

/* chainercv calc_semantic_segmentation_confusion([pred], [gt]) for 2 classes:
   pixels with gt < 0 are ignored; confusion[gt, pred].  out = {TN, FP, FN, TP}
   with TP = conf[1,1], FP = conf[0,1], FN = conf[1,0]  (:400-402). */
void orc_confusion(const uint8_t *pred, const int32_t *gt, int64_t npix, int64_t out[4])
{
    out[0] = out[1] = out[2] = out[3] = 0;
    for (int64_t i = 0; i < npix; ++i) {
        if (gt[i] < 0) continue;
        out[(gt[i] ? 2 : 0) + (pred[i] ? 1 : 0)] += 1;
    }
}
