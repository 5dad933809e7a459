// Felzenszwalb-Huttenlocher graph-based superpixels on gfx950, bit exact with scikit-image's
// felzenszwalb() as the reference calls it (batch_spalign_kmeans.py:301-307: img/255, scale 300,
// sigma 0.8, min_size 20) given a canonical order of equal edge costs (edge index).
//
// The reference algorithm is a SEQUENTIAL greedy pass over the cost-sorted edges of the
// 8-connected pixel grid: an edge merges its two components iff its cost is below
// min(int(A) + k/|A|, int(B) + k/|B|) (thresholds rounded to float32 in the Cython core), where the
// component state (root = smallest pixel index, size, internal cost) is that left by all earlier
// edges; a second pass merges components below min_size.  The order dependence is kept exactly
// with deterministic reservations: the sorted edges are taken in windows; in every round each
// still-pending edge (1) finds its two roots and evaluates the merge test against the current
// state, and if it wants to merge, reserves both roots with atomicMin(sorted position); (2) edges
// blocked by an earlier reservation hold their components as well, to a fixed point (they may
// want to merge once the earlier edge has changed their component); (3) an edge is decided —
// merge committed, or dropped — only if no earlier edge holds either of its components,
// otherwise it waits for the next round.  An edge is therefore always decided against exactly
// the state the sequential pass would show it.  Zero-cost edges (always merge) are unioned up
// front with a lock-free union-find.  The passes are bound by the number of rounds (chains of
// dependent merges into a growing component), so each image gets ONE 1024-thread workgroup: a
// round then costs workgroup barriers (~1 us) rather than agent-scope barriers; the images of
// the batch proceed in parallel on different CUs.
//   smoothing : scipy.ndimage.gaussian_filter(img, [sigma, sigma, 0]) — symmetric correlate1d
//               summation order, 'reflect' borders, float64
//   costs     : float64 Euclidean colour distance of the 4 edge families (right, down, down-right,
//               up-right), keys sorted with a stable LSD radix sort (hipCUB) on the cost bits
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <stdlib.h>

#include "spa_common.h"

#ifndef FZ_THREADS
#define FZ_THREADS 1024
#endif
#define FZ_EPT 1                 // edges per thread per window (1: a window of 1 024 sorted edges; larger windows only add serial work per round — measured 11.6 / 14.1 / 19.0 / 41 ms per 30 images of 224x224 for 1 / 2 / 4 / 8)

#define FZ_MAXB 256               // images per launch (one resident workgroup each)
struct FzImg {
    unsigned barrier;            // monotonic arrival counter of this image's workgroup group
    int cnt[4];                  // ring of group-wide counters (see fz_group_sum)
    int pad[3];
};

// ---------------------------------------------------------------------------------------
// host: scipy.ndimage._gaussian_kernel1d with the deterministic exp of the oracle
// ---------------------------------------------------------------------------------------
static double h_det_exp(double t)
{
    double kf = floor(t * 1.44269504088896338700e+00 + 0.5);
    double r = (t - kf * 6.93147180369123816490e-01) - kf * 1.90821492927058770002e-10;
    double p = 1.0 / 87178291200.0;
    p = p * r + 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    int k = (int)kf;
    unsigned long long b = (unsigned long long)(k + 1023) << 52;
    double sc;
    memcpy(&sc, &b, 8);
    return p * sc;
}

#define FZ_MAXW 65
static int fz_weights(double sigma, double *w)
{
    int r = (int)(4.0 * sigma + 0.5);
    if (2 * r + 1 > FZ_MAXW) return -1;
    double sigma2 = sigma * sigma, sum = 0.0;
    for (int i = -r; i <= r; ++i) w[i + r] = h_det_exp(-0.5 / sigma2 * (double)(i * i));
    int n = 2 * r + 1;
    if (n < 8) {
        for (int i = 0; i < n; ++i) sum += w[i];
    } else {      // numpy pairwise summation of a short vector
        double acc[8];
        int i;
        for (i = 0; i < 8; ++i) acc[i] = w[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) acc[j] += w[i + j];
        sum = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        for (; i < n; ++i) sum += w[i];
    }
    for (int i = 0; i < n; ++i) w[i] = w[i] / sum;
    return r;
}

struct FzWeights { double w[FZ_MAXW]; int r; };

__device__ __forceinline__ int fz_reflect(int i, int n)
{
    if (n == 1) return 0;
    int period = 2 * n;
    i = i % period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

// pass 1: x/255 (float32), widen, smooth along y -> planar float64
// `wide` = the caller's image was uint8: the reference divides that by 255. in float64
// (superpixel_overlaps.py:297-298), a float32 image in float32 (batch_spalign_kmeans.py:303)
__device__ __forceinline__ double fz_unit(float v, bool wide)
{
    return wide ? (double)v / 255.0 : (double)(v / 255.0f);
}

__global__ __launch_bounds__(256) void k_fz_blur_y(const float *__restrict__ rgb, double *__restrict__ out,
                                                   int H, int W, FzWeights fw, bool wide)
{
    const long long npix = (long long)H * W;
    const long long plane = blockIdx.y;                      // b*3 + c
    const float *src = rgb + plane * npix;
    double *dst = out + plane * npix;
    const double *wc = fw.w + fw.r;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        double tmp = fz_unit(src[p], wide) * wc[0];
        for (int j = -fw.r; j < 0; ++j) {
            double a = fz_unit(src[(long long)fz_reflect(y + j, H) * W + x], wide);
            double b = fz_unit(src[(long long)fz_reflect(y - j, H) * W + x], wide);
            tmp = tmp + (a + b) * wc[j];
        }
        dst[p] = tmp;
    }
}

__global__ __launch_bounds__(256) void k_fz_blur_x(const double *__restrict__ in, double *__restrict__ out,
                                                   int H, int W, FzWeights fw)
{
    const long long npix = (long long)H * W;
    const long long plane = blockIdx.y;
    const double *src = in + plane * npix;
    double *dst = out + plane * npix;
    const double *wc = fw.w + fw.r;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        const double *row = src + (long long)y * W;
        double tmp = row[x] * wc[0];
        for (int j = -fw.r; j < 0; ++j)
            tmp = tmp + (row[fz_reflect(x + j, W)] + row[fz_reflect(x - j, W)]) * wc[j];
        dst[p] = tmp;
    }
}

// edge index -> endpoints; families in the reference's concatenation order
struct FzGeom { int H, W; long long nR, nD, nDR, nE; };

__device__ __forceinline__ void fz_endpoints(const FzGeom &g, long long idx64, int &a, int &b)
{
    // nE < 2^32 (checked by the caller): 32-bit unsigned divisions, a fifth of the instructions of 64-bit ones —
    // the passes decode an edge index per thread and chunk, and that was half of a chunk's time
    const unsigned W = (unsigned)g.W, idx = (unsigned)idx64;
    const unsigned nR = (unsigned)g.nR, nD = (unsigned)g.nD, nDR = (unsigned)g.nDR;
    if (idx < nR) {
        const unsigned y = idx / (W - 1), x = idx - y * (W - 1) + 1;
        a = (int)(y * W + x); b = a - 1;
    } else if (idx < nR + nD) {
        const unsigned i = idx - nR;
        const unsigned y = i / W + 1, x = i - (y - 1) * W;
        a = (int)(y * W + x); b = a - (int)W;
    } else if (idx < nR + nD + nDR) {
        const unsigned i = idx - nR - nD;
        const unsigned q = i / (W - 1);
        const unsigned y = q + 1, x = i - q * (W - 1) + 1;
        a = (int)(y * W + x); b = a - (int)W - 1;
    } else {
        const unsigned i = idx - nR - nD - nDR;
        const unsigned y = i / (W - 1), x = i - y * (W - 1) + 1;
        a = (int)(y * W + x); b = a + (int)W - 1;
    }
}

// costs = sqrt(((d0*d0) + d1*d1) + d2*d2), float64; key = bit pattern (non-negative doubles order
// like unsigned integers), value = edge index
__global__ __launch_bounds__(256) void k_fz_costs(const double *__restrict__ sm, FzGeom g,
                                                  unsigned long long *__restrict__ keys,
                                                  unsigned *__restrict__ vals)
{
    const int b = blockIdx.y;
    const long long npix = (long long)g.H * g.W;
    const double *s0 = sm + (long long)b * 3 * npix, *s1 = s0 + npix, *s2 = s1 + npix;
    unsigned long long *K = keys + (long long)b * g.nE;
    unsigned *V = vals + (long long)b * g.nE;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < g.nE; e += (long long)gridDim.x * 256) {
        int pa, pb;
        fz_endpoints(g, e, pa, pb);
        double d0 = s0[pa] - s0[pb], d1 = s1[pa] - s1[pb], d2 = s2[pa] - s2[pb];
        double s = d0 * d0;
        s = s + d1 * d1;
        s = s + d2 * d2;
        K[e] = (unsigned long long)__double_as_longlong(sqrt(s));
        V[e] = (unsigned)e;
    }
}

__global__ void k_fz_init(int *__restrict__ parent, int *__restrict__ size, double *__restrict__ cint,
                          unsigned long long *__restrict__ mark, long long total, FzImg *__restrict__ st, int B)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        parent[i] = -1; size[i] = 1; cint[i] = 0.0; mark[i] = ~0ull;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < B) {
        st[threadIdx.x].barrier = 0;
        for (int i = 0; i < 4; ++i) st[threadIdx.x].cnt[i] = 0;
        for (int i = 0; i < 3; ++i) st[threadIdx.x].pad[i] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x < 16) ((int *)((char *)st + FZ_MAXB * sizeof(FzImg)))[FZ_MAXB + threadIdx.x] = 0;    // pass diagnostics
}

__device__ __forceinline__ void fz_group_sync(unsigned *ctr, unsigned G, unsigned &epoch, uint32_t *status)
{
    if (G == 1u) {           // the whole image lives in one workgroup: a workgroup barrier is enough
        __syncthreads();
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    epoch += 1;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * G;
        long long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1ll << 27)) { atomicOr(status, SPA_ST_KMEANS_BARRIER); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// sum of `local` over all threads of the image's workgroup group (one barrier).  Counter slots
// are used round-robin; the slot two steps ahead is cleared on the way (it was last read two
// barriers ago).
__device__ __forceinline__ int fz_group_sum(FzImg *me, unsigned &slot, int local, bool is_first_thread,
                                            unsigned G, unsigned &epoch, uint32_t *status)
{
    if (G == 1u) return __syncthreads_or(local != 0);     // callers only test the sum against zero
    int *c = &me->cnt[slot & 3u];
    if (local) atomicAdd(c, local);
    if (is_first_thread) __hip_atomic_store(&me->cnt[(slot + 2u) & 3u], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fz_group_sync(&me->barrier, G, epoch, status);
    const int total = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    slot += 1u;
    return total;
}

// parent[] convention: -1 = root (self), otherwise the parent pixel (always a smaller index)
__device__ __forceinline__ int fz_find(const int *P, int i)
{
    int p;
    while ((p = P[i]) >= 0) i = p;
    return i;
}

// ---------------------------------------------------------------------------------------
// Zero-cost edges.  They sort first, and with scale > 0 every one of them passes the merge test
// whatever the state (0 < int + k/size), so the sequential pass simply unions the connected
// components of the zero-cost graph, root = smallest pixel, size = pixel count, internal cost 0.
// That is order free: done with a lock-free union-find up front.  (Flat regions would otherwise
// produce reservation chains as long as their rows, because equal costs are ordered by index.)
// ---------------------------------------------------------------------------------------
__global__ void k_fz_zero_count(const unsigned long long *__restrict__ keys, long long nE, int *__restrict__ zcount)
{
    // first sorted position whose cost is not +0.0 (binary search, one thread per image)
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const unsigned long long *K = keys + (long long)b * nE;
    long long lo = 0, hi = nE;
    while (lo < hi) {
        long long mid = (lo + hi) >> 1;
        if (K[mid] == 0ull) lo = mid + 1; else hi = mid;
    }
    zcount[b] = (int)lo;
}

__global__ __launch_bounds__(256) void k_fz_zero_union(const unsigned *__restrict__ vals, FzGeom g,
                                                       const int *__restrict__ zcount, int *__restrict__ parent)
{
    const int b = blockIdx.y;
    const long long npix = (long long)g.H * g.W;
    const unsigned *V = vals + (long long)b * g.nE;
    int *P = parent + (long long)b * npix;
    const int z = zcount[b];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < z; e += gridDim.x * 256) {
        int a, c;
        fz_endpoints(g, (long long)V[e], a, c);
        for (;;) {
            int ra = a, rb = c, p;
            while ((p = __hip_atomic_load(P + ra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= 0) ra = p;
            while ((p = __hip_atomic_load(P + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= 0) rb = p;
            if (ra == rb) break;
            const int lo_r = min(ra, rb), hi_r = max(ra, rb);
            if (atomicCAS(P + hi_r, -1, lo_r) == -1) break;      // hi_r was still a root: linked
        }
    }
}

__global__ __launch_bounds__(256) void k_fz_zero_sizes(const int *__restrict__ zcount, int *__restrict__ parent,
                                                       int *__restrict__ size, int npix)
{
    const int b = blockIdx.y;
    if (zcount[b] == 0) return;
    int *P = parent + (long long)b * npix;
    int *S = size + (long long)b * npix;
    const int p = blockIdx.x * 256 + threadIdx.x;
    int r = -1;
    if (p < npix && P[p] >= 0) {
        r = fz_find(P, p);
        P[p] = r;                                  // flatten
    }
    unsigned long long todo = __ballot(r >= 0);    // one atomic per distinct root per wave
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int rr = __shfl(r, leader);
        const unsigned long long same = __ballot(r == rr);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(S + rr, __popcll(same));
        todo &= ~same;
    }
}

// One greedy pass (mode 0: merge test of the paper; mode 1: min_size clean-up) over the sorted
// edges of every image; grid = (G, B), workgroups (., b) form the group of image b.
template <bool LDSP>
__global__ __launch_bounds__(FZ_THREADS) void k_fz_pass(const unsigned long long *__restrict__ keys,
                                                        const unsigned *__restrict__ vals, FzGeom g,
                                                        int *__restrict__ parent, int *__restrict__ size,
                                                        double *__restrict__ cint,
                                                        unsigned long long *__restrict__ mark,
                                                        FzImg *__restrict__ st, double scale, int min_size,
                                                        int mode, unsigned round0,
                                                        const int *__restrict__ zcount, int flatten_every,
                                                        uint32_t *__restrict__ status)
{
    const int b = blockIdx.y;
    const unsigned G = gridDim.x;
    const long long npix = (long long)g.H * g.W;
    const unsigned long long *K = keys + (long long)b * g.nE;
    const unsigned *V = vals + (long long)b * g.nE;
    int *P = parent + (long long)b * npix;
    int *S = size + (long long)b * npix;
    double *CI = cint + (long long)b * npix;
    unsigned long long *M = mark + (long long)b * npix;
    FzImg *me = st + b;
    unsigned epoch = me->barrier / G;             // barrier counter persists across launches
    const long long T = (long long)G * FZ_THREADS;
    const long long tg = (long long)blockIdx.x * FZ_THREADS + threadIdx.x;
    unsigned round = round0;
    unsigned slot = 0;
    int win = 0;
    // LDSP (images of <= 65 535 pixels, one workgroup per image): the parent array — the only pointer-chased
    // state, two to three dependent round trips per endpoint and round — lives in LDS as 16-bit indices
    // (0xFFFF = root) for the duration of the pass.
    extern __shared__ unsigned short lpar[];
    auto getp = [&](int i) -> int {
        if (LDSP) { const unsigned v = lpar[i]; return v == 0xFFFFu ? -1 : (int)v; }
        return P[i];
    };
    auto setp = [&](int i, int v) { if (LDSP) lpar[i] = (unsigned short)v; else P[i] = v; };
    auto find = [&](int i) { int p; while ((p = getp(i)) >= 0) i = p; return i; };
    if (LDSP) {
        for (int p = (int)tg; p < (int)npix; p += FZ_THREADS) { const int q = P[p]; lpar[p] = q < 0 ? (unsigned short)0xFFFFu : (unsigned short)q; }
        __syncthreads();
    }

    for (long long lo = zcount[b]; lo < g.nE; lo += T * FZ_EPT) {   // zero-cost edges: done up front
        if (tg == 0) me->pad[2] += 1;                           // diagnostics: windows
        // flattening the forest keeps the trees shallow; it costs a sweep over the image, so it is
        // done every `flatten_every` windows (about once per npix/4 edges)
        if ((win++ % flatten_every) == 0) {
            for (long long p = tg; p < npix; p += T) {
                int q = getp((int)p);
                if (q >= 0) {
                    int r = find(q);
                    if (r != q) setp((int)p, r);
                }
            }
        }
        fz_group_sync(&me->barrier, G, epoch, status);
        bool pend[FZ_EPT];
        int ea[FZ_EPT], eb[FZ_EPT];
        double cost[FZ_EPT];
        // roots carried from round to round (round 4): a root of the last round is either still a root (one read) or has
        // been linked to its new root by this window's merges (two or three reads) — the walk from the pixel through the
        // forest of the last flattening is paid once per window, not once per round.  ea/eb hold the pixel, then the roots.
        int ra[FZ_EPT], rb[FZ_EPT];
#pragma unroll
        for (int u = 0; u < FZ_EPT; ++u) {
            const long long e = lo + tg + (long long)u * T;     // sorted position
            pend[u] = e < g.nE;
            if (pend[u]) {
                fz_endpoints(g, (long long)V[e], ea[u], eb[u]);
                cost[u] = __longlong_as_double((long long)K[e]);
            }
            ra[u] = ea[u]; rb[u] = eb[u];
        }
        for (;;) {
            ++round;
            if (tg == 0) me->pad[0] += 1;                       // diagnostics: rounds
            const unsigned long long tag = (unsigned long long)(~round) << 32;
            bool want[FZ_EPT], resv[FZ_EPT];
            unsigned hold_a[FZ_EPT], hold_b[FZ_EPT];            // earliest reservation of either component, as last read
            // ---- phase 1: roots and the merge test against the current state; edges that want
            // to merge reserve both components with their position in the window
#pragma unroll
            for (int u = 0; u < FZ_EPT; ++u) {
                want[u] = false; resv[u] = false;
                hold_a[u] = hold_b[u] = 0xFFFFFFFFu;
                if (!pend[u]) continue;
                ra[u] = find(ra[u]);
                rb[u] = find(rb[u]);
                if (ra[u] == rb[u]) { pend[u] = false; continue; }      // same component for ever
                if (mode == 0) {
                    const float t0 = (float)(CI[ra[u]] + scale / (double)S[ra[u]]);
                    const float t1 = (float)(CI[rb[u]] + scale / (double)S[rb[u]]);
                    want[u] = cost[u] < (double)(t0 < t1 ? t0 : t1);
                } else {
                    want[u] = S[ra[u]] < min_size || S[rb[u]] < min_size;
                }
                if (want[u]) {
                    const unsigned long long key = tag | (unsigned long long)(unsigned)(tg + (long long)u * T);
                    atomicMin(M + ra[u], key);
                    atomicMin(M + rb[u], key);
                    resv[u] = true;
                }
            }
            fz_group_sync(&me->barrier, G, epoch, status);
            // ---- propagation: an edge that does not want to merge NOW may want to once an earlier
            // reserved edge has changed one of its components, so while it waits it must hold its
            // components too (later edges must not be decided against a state it might still change)
            // (every pending edge reads the two reservation words in every step, also the ones that hold already: the step
            // that changes nothing leaves the values the decision below needs — one dependent round trip less per round)
            for (;;) {
                int changed = 0;
#pragma unroll
                for (int u = 0; u < FZ_EPT; ++u) {
                    if (!pend[u]) continue;
                    const unsigned pos = (unsigned)(tg + (long long)u * T);
                    const unsigned long long ma = __hip_atomic_load(M + ra[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long mb = __hip_atomic_load(M + rb[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned pa = (ma >> 32) == (tag >> 32) ? (unsigned)ma : 0xFFFFFFFFu;
                    const unsigned pb = (mb >> 32) == (tag >> 32) ? (unsigned)mb : 0xFFFFFFFFu;
                    hold_a[u] = pa; hold_b[u] = pb;
                    if (resv[u]) continue;
                    if (pa < pos || pb < pos) {
                        const unsigned long long key = tag | (unsigned long long)pos;
                        atomicMin(M + ra[u], key);
                        atomicMin(M + rb[u], key);
                        resv[u] = true;
                        ++changed;
                    }
                }
                if (tg == 0) me->pad[1] += 1;                   // diagnostics: propagation steps
                if (fz_group_sum(me, slot, changed, tg == 0, G, epoch, status) == 0) break;
            }
            // ---- phase 2: decide every edge no earlier reservation can influence
            int left = 0;
#pragma unroll
            for (int u = 0; u < FZ_EPT; ++u) {
                if (!pend[u]) continue;
                const unsigned pos = (unsigned)(tg + (long long)u * T);
                const unsigned pa = hold_a[u], pb = hold_b[u];                   // read in the last (unchanged) propagation step
                if (pa < pos || pb < pos) { ++left; continue; }                  // waits for an earlier edge
                if (want[u]) {
                    const int lo_r = min(ra[u], rb[u]), hi_r = max(ra[u], rb[u]);
                    const int ns = S[ra[u]] + S[rb[u]];
                    setp(hi_r, lo_r);
                    S[lo_r] = ns;
                    if (mode == 0) CI[lo_r] = cost[u];
                }
                pend[u] = false;
            }
            if (fz_group_sum(me, slot, left, tg == 0, G, epoch, status) == 0) break;
        }
    }
    if (LDSP) {
        __syncthreads();
        for (int p = (int)tg; p < (int)npix; p += FZ_THREADS) { const unsigned v = lpar[p]; P[p] = v == 0xFFFFu ? -1 : (int)v; }
    }
}

// ---------------------------------------------------------------------------------------
// The pass with the state of the roots a window touches in LDS, one 1 024-thread workgroup per image.
//  * LIVE edges only.  An edge whose endpoints already share a component can never merge anything (components only
//    grow), and in the clean-up pass neither can an edge between two components of at least min_size pixels.  Most
//    of the sorted list is dead by the time the pass reaches it (a 224x224 image: ~50 k merges out of 200 k edges;
//    a smooth full-size image: 8.4 M edges for at most 2 M merges), so the workgroup first walks chunks of 1 024
//    sorted positions, tests them against the current forest and compacts the live ones, in order, into the
//    window — a window is 1 024 LIVE edges however many positions that takes.
//  * A window touches at most 2 048 components; their roots are entered into an open-addressing table in LDS at
//    the start of the window — key and size packed in one word, the internal cost, the reservation word — and
//    every round of the window (find, merge test, reservations, propagation, decisions) runs on that table: a
//    round costs workgroup barriers and LDS latencies instead of four or five dependent L2 round trips
//    (~11 us -> ~2-3 us).  Merged components keep the cell of their surviving root (always one of the two that
//    were entered); the cells go back to the global size / cost arrays when the window is done.
//  * LPAR: images of <= 57 000 pixels keep the parent array in LDS as 16-bit indices as well (224x224, the
//    reference's operating point); larger ones chase it through L2 (kept shallow by the flattening sweeps).
// Decisions are those of k_fz_pass: same tests on the same values in the same order.
// ---------------------------------------------------------------------------------------
#define FZL_EMPTY 0xFFFFFFFFu
#ifndef FZ_CK
#define FZ_CK 2
#endif
// one wave's LDS and global accesses have completed and are visible to its own lanes
// Large images (round 5): a root's internal cost and size in ONE 16-byte record (the table pass fetches and writes back both per
// root and window: one scattered access instead of two).  The records live where the internal costs and the legacy pass's
// reservation words do (WS_FZ_STATE, 16 bytes per pixel: the table pass needs no reservation words in global memory).
struct __attribute__((aligned(16))) FzRec { double ci; int size; int pad; };
__global__ __launch_bounds__(256) void k_fz_records(FzRec *__restrict__ rec, const int *__restrict__ size, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        FzRec r; r.ci = 0.0; r.size = size[i]; r.pad = 0;       // (before pass 0 every internal cost is 0: only zero-cost edges are merged)
        rec[i] = r;
    }
}
// inclusive scans over the 64 lanes of a wave (row-shift DPP inside the rows of 16, the three row totals by v_readlane)
__device__ __forceinline__ int fz_scan_add(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
    const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47);
    const int row = (int)(threadIdx.x & 63) >> 4;
    return v + (row >= 1 ? r0 : 0) + (row >= 2 ? r1 : 0) + (row >= 3 ? r2 : 0);
}
__device__ __forceinline__ int fz_scan_min(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, 0x111, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, 0x112, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, 0x114, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, 0x118, 0xF, 0xF, false));
    const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47);
    const int row = (int)(threadIdx.x & 63) >> 4;
    int m = v;
    if (row >= 1) m = min(m, r0);
    if (row >= 2) m = min(m, r1);
    if (row >= 3) m = min(m, r2);
    return m;
}
__device__ __forceinline__ void fz_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

template <bool LPAR>
__global__ __launch_bounds__(FZ_THREADS) void k_fz_pass_tab(const unsigned long long *__restrict__ keys,
                                                            const unsigned *__restrict__ vals, FzGeom g,
                                                            int *__restrict__ parent, int *__restrict__ size,
                                                            double *__restrict__ cint, double scale, int min_size,
                                                            int mode, const int *__restrict__ zcount,
                                                            int flatten_every, int cells, int *__restrict__ diag, int hub_on = 0,
                                                            const unsigned *__restrict__ idx = nullptr, int *__restrict__ seg = nullptr,
                                                            int seg_div = 1, int skip_first_flatten = 0)
{
    const int b = blockIdx.x;
    const int npix = g.H * g.W;
    const unsigned long long *K = keys + (long long)b * g.nE;
    const unsigned *V = vals + (long long)b * g.nE;
    // large images (round 5): the pass walks a LIST of sorted positions — idx[b * nE + e] (idx == NULL: e itself) for e in
    // [seg[2b], seg[2b + 1]) — and of that only the first 1 / seg_div: the prefilter kernels in front of fz_run cut pass 0 into
    // segments and drop, between two segments and before the clean-up pass, every edge that can no longer merge anything
    const unsigned *IX = idx ? idx + (long long)b * g.nE : nullptr;
    long long nEb = g.nE, seg_start = -1;
    if (seg) {
        const long long st0 = seg[2 * b], n0 = seg[2 * b + 1];
        seg_start = st0;
        nEb = st0 + (n0 - st0 + seg_div - 1) / seg_div;
    }
    int *P = parent + (long long)b * npix;
    int *S = size + (long long)b * npix;
    double *CI = cint + (long long)b * npix;
    FzRec *SC = (FzRec *)cint + (long long)b * npix;          // !LPAR with hub_on bit 2: records instead of S / CI
    const bool recs = !LPAR && (hub_on & 4);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    extern __shared__ __attribute__((aligned(16))) unsigned char fzl[];
    unsigned short *lpar = (unsigned short *)fzl;
    double *lci = (double *)(fzl + (LPAR ? (((size_t)npix * 2 + 15) & ~(size_t)15) : 0));
    unsigned *lkey = (unsigned *)(lci + cells);            // LPAR: (root << 16) | size; else root, size in lsz
    unsigned *lres = lkey + cells;                         // reservation: (tag << 10) | position in the window
    unsigned *lsz = lres + cells;                          // !LPAR only
    unsigned *wbuf = LPAR ? lsz : lsz + cells;             // the window: positions of its live edges in the sorted arrays
    // !LPAR (round 5): the forest of the window's OWN merges over the table's cells — lcp[c] = the cell c's component was merged
    // into (itself: still a root).  The rounds of a window then find their roots through LDS alone; the global parent array is
    // still written at every merge (the next window's set-up walks it) but never read inside a window's rounds
    unsigned short *lcp = (unsigned short *)(wbuf + FZ_THREADS);
    // !LPAR, hub chains (round 5, below): pending edges per cell, hub slot per cell, the edges' cells / costs / decided flags,
    // per-hub bitmaps of the window positions that touch the hub
    constexpr int NHUB = 16, HUB_TH = 4;
    unsigned short *lcnt = lcp + cells;
    unsigned short *e_ca = lcnt + cells, *e_cb = e_ca + FZ_THREADS;
    unsigned char *lslot = (unsigned char *)(e_cb + FZ_THREADS);
    unsigned char *e_done = lslot + cells;
    double *e_cost = (double *)(((uintptr_t)(e_done + FZ_THREADS) + 7) & ~(uintptr_t)7);
    unsigned *hbits = (unsigned *)(e_cost + FZ_THREADS);
    int *hcell = (int *)(hbits + NHUB * (FZ_THREADS / 32));
    int *nhub = hcell + NHUB;
    float *e_thr = (float *)(nhub + 4);                    // merge threshold of an edge's non-hub side (computed by its own thread)
    int *wra = (int *)(e_thr + FZ_THREADS), *wrb = wra + FZ_THREADS;      // !LPAR: the live edges' roots as the collect step found them
    auto cfind = [&](int c) -> int {
        int p;
        while ((p = (int)__hip_atomic_load(lcp + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != c) c = p;
        return c;
    };
    __shared__ int wave_cnt[2 * FZ_CK * (FZ_THREADS / 64)];
    __shared__ unsigned cut_s[2];
    __shared__ int tail_ea[64], tail_eb[64];
    __shared__ double tail_cost[64];
    auto getp = [&](int i) -> int {
        if (LPAR) { const unsigned v = lpar[i]; return v == 0xFFFFu ? -1 : (int)v; }
        return P[i];
    };
    auto setp = [&](int i, int v) { if (LPAR) lpar[i] = (unsigned short)v; else P[i] = v; };
    auto find = [&](int i) { int p; while ((p = getp(i)) >= 0) i = p; return i; };
    auto hash = [&](int root) -> int {
        return (int)(((unsigned long long)(((unsigned)root * 2654435761u) >> 8) * (unsigned)cells) >> 24);
    };
    auto keyroot = [&](unsigned w) -> unsigned { return LPAR ? (w >> 16) : w; };
    auto lookup = [&](int root) -> int {                   // the root is in the table
        int h = hash(root);
        for (;;) {
            const unsigned w = __hip_atomic_load(lkey + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (w != FZL_EMPTY && keyroot(w) == (unsigned)root) return h;
            h = h + 1 == cells ? 0 : h + 1;
        }
    };
    auto enter = [&](int root) -> int {
        int h = hash(root);
        for (;;) {
            unsigned w = __hip_atomic_load(lkey + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (w == FZL_EMPTY) {
                w = atomicCAS(lkey + h, FZL_EMPTY, LPAR ? (unsigned)root << 16 : (unsigned)root);
                if (w == FZL_EMPTY) {                       // this thread's cell: fetch the root's state
                    if (LPAR) __hip_atomic_store(lkey + h, ((unsigned)root << 16) | (unsigned)S[root], __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP);
                    else if (recs) { const FzRec q = SC[root]; lsz[h] = (unsigned)q.size; lci[h] = q.ci; return h; }
                    else lsz[h] = (unsigned)S[root];
                    if (mode == 0) lci[h] = CI[root];
                    return h;
                }
            }
            if (keyroot(w) == (unsigned)root) return h;
            h = h + 1 == cells ? 0 : h + 1;
        }
    };
    auto claim = [&](int root, bool &own) -> int {         // !LPAR: enter() without the fetch of the root's state (own: the caller's to fill)
        int h = hash(root);
        for (;;) {
            unsigned w = __hip_atomic_load(lkey + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (w == FZL_EMPTY) {
                w = atomicCAS(lkey + h, FZL_EMPTY, (unsigned)root);
                if (w == FZL_EMPTY) { own = true; return h; }
            }
            if (w == (unsigned)root) return h;
            h = h + 1 == cells ? 0 : h + 1;
        }
    };
    auto csize = [&](int c) -> unsigned { return LPAR ? (lkey[c] & 0xFFFFu) : lsz[c]; };
    if (LPAR) {
        for (int p = tid; p < npix; p += FZ_THREADS) { const int q = P[p]; lpar[p] = q < 0 ? (unsigned short)0xFFFFu : (unsigned short)q; }
    }
    __syncthreads();

    int win = skip_first_flatten ? 1 : 0, chunks = 0, rounds = 0, trounds = 0;      // (the prefilter has just flattened the forest)
    long long tc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // diagnostics: cycles of flatten + write-back | collect | window set-up (table entry) | full rounds | tail | (of the rounds) propagation | hub chains | (set-up) table reset + cost loads
    int props = 0;
    long long t_ = (long long)__builtin_readcyclecounter();
#define FZ_T(i) { const long long n_ = (long long)__builtin_readcyclecounter(); tc[i] += n_ - t_; t_ = n_; }
    long long cursor = seg ? seg_start : zcount[b];         // zero-cost edges: done up front
    unsigned vpre[FZ_CK], opre[FZ_CK];                      // edge indices and sorted-array positions of the next step
    long long vpre_at = -1;
    int step_par = 0;
    if (tid == 0) { cut_s[0] = 0xFFFFFFFFu; cut_s[1] = 0xFFFFFFFFu; }
    __syncthreads();
    while (cursor < nEb) {
        if ((win++ % flatten_every) == 0) {
            for (int p = tid; p < npix; p += FZ_THREADS) {
                const int q = getp(p);
                if (q >= 0) { const int r = find(q); if (r != q) setp(p, r); }
            }
            __syncthreads();
        }
        FZ_T(0)
        // ---- collect the next (up to) 1 024 live edges, in sorted order
        int nlive = 0;
        while (nlive < FZ_THREADS && cursor < nEb) {
            // FZ_CK chunks of 1 024 sorted positions per step; the edge indices of the next step are already on
            // their way (vpre, loaded for `vpre_at`), the step's counters alternate between two LDS sets so that
            // a step needs two barriers
            chunks += FZ_CK;
            unsigned vcur[FZ_CK], ocur[FZ_CK];
#pragma unroll
            for (int k = 0; k < FZ_CK; ++k) {
                const long long e = cursor + (long long)k * FZ_THREADS + tid;
                if (vpre_at == cursor) { vcur[k] = vpre[k]; ocur[k] = opre[k]; }
                else {
                    ocur[k] = e < nEb ? (IX ? IX[e] : (unsigned)e) : 0u;
                    vcur[k] = e < nEb ? V[ocur[k]] : 0u;
                }
            }
            {
                const long long nx = cursor + (long long)FZ_CK * FZ_THREADS;
#pragma unroll
                for (int k = 0; k < FZ_CK; ++k) {
                    const long long e = nx + (long long)k * FZ_THREADS + tid;
                    opre[k] = e < nEb ? (IX ? IX[e] : (unsigned)e) : 0u;
                    vpre[k] = e < nEb ? V[opre[k]] : 0u;
                }
                vpre_at = nx;
            }
            bool live[FZ_CK];
            unsigned long long m[FZ_CK];
            int *wc = wave_cnt + (step_par ? FZ_CK * (FZ_THREADS / 64) : 0);
            unsigned *cut = cut_s + step_par;
            int ea_[FZ_CK], eb_[FZ_CK];
            if (LPAR) {
#pragma unroll
                for (int k = 0; k < FZ_CK; ++k) {
                    const long long e = cursor + (long long)k * FZ_THREADS + tid;
                    live[k] = false;
                    if (e < nEb) {
                        int a, c;
                        fz_endpoints(g, (long long)vcur[k], a, c);
                        const int ra_ = find(a), rb_ = find(c);
                        live[k] = ra_ != rb_;
                        if (live[k] && mode == 1) live[k] = S[ra_] < min_size || S[rb_] < min_size;
                    }
                }
            } else {
                // parent array in global memory (round 5): the 2 FZ_CK walks of a thread advance LEVEL BY LEVEL — every level is one
                // batch of independent loads instead of 2 FZ_CK dependent chains one after the other — and an edge whose endpoints
                // point at the same parent is dead without looking further (most sorted positions of a late pass are: after a
                // flattening sweep both pixels point straight at their root)
                bool act[FZ_CK];
#pragma unroll
                for (int k = 0; k < FZ_CK; ++k) {
                    const long long e = cursor + (long long)k * FZ_THREADS + tid;
                    act[k] = e < nEb;
                    live[k] = false;
                    ea_[k] = eb_[k] = 0;
                    if (act[k]) fz_endpoints(g, (long long)vcur[k], ea_[k], eb_[k]);
                }
                for (;;) {
                    int pa_[FZ_CK], pb_[FZ_CK];
                    bool any = false;
#pragma unroll
                    for (int k = 0; k < FZ_CK; ++k) { pa_[k] = act[k] ? P[ea_[k]] : -1; pb_[k] = act[k] ? P[eb_[k]] : -1; }
#pragma unroll
                    for (int k = 0; k < FZ_CK; ++k) {
                        if (!act[k]) continue;
                        if (pa_[k] >= 0 && pa_[k] == pb_[k]) { act[k] = false; continue; }        // same parent: one component
                        if (pa_[k] >= 0) ea_[k] = pa_[k];
                        if (pb_[k] >= 0) eb_[k] = pb_[k];
                        if (pa_[k] < 0 && pb_[k] < 0) { act[k] = false; live[k] = ea_[k] != eb_[k]; }   // both are roots
                        else if (ea_[k] == eb_[k]) act[k] = false;                                  // met on the way up
                        else any = true;
                    }
                    if (!any) break;
                }
                if (mode == 1) {
#pragma unroll
                    for (int k = 0; k < FZ_CK; ++k)
                        if (live[k]) live[k] = recs ? (SC[ea_[k]].size < min_size || SC[eb_[k]].size < min_size)
                                                    : (S[ea_[k]] < min_size || S[eb_[k]] < min_size);
                }
            }
#pragma unroll
            for (int k = 0; k < FZ_CK; ++k) {
                m[k] = __ballot(live[k]);
                if (lane == 0) wc[k * (FZ_THREADS / 64) + wv] = __popcll(m[k]);
            }
            if (tid == 0) cut_s[step_par ^ 1] = 0xFFFFFFFFu;     // the next step's
            __syncthreads();
            int run = nlive;
#pragma unroll
            for (int k = 0; k < FZ_CK; ++k) {
                int off = run;
#pragma unroll
                for (int i = 0; i < FZ_THREADS / 64; ++i) { const int c = wc[k * (FZ_THREADS / 64) + i]; if (i < wv) off += c; run += c; }
                if (live[k]) {
                    const long long e = cursor + (long long)k * FZ_THREADS + tid;
                    const int pos = off + (int)spa_rank_in_mask(m[k]);
                    if (pos < FZ_THREADS) {
                        wbuf[pos] = ocur[k];
                        if (!LPAR) { wra[pos] = ea_[k]; wrb[pos] = eb_[k]; }      // (both are roots, and stay roots until the window runs)
                    }
                    else atomicMin(cut, (unsigned)e);       // the first live edge that does not fit starts the next window
                }
            }
            __syncthreads();
            if (run > FZ_THREADS) { cursor = (long long)*cut; nlive = FZ_THREADS; }
            else { nlive = run; cursor += (long long)FZ_CK * FZ_THREADS; }
            step_par ^= 1;
        }
        FZ_T(1)
        if (nlive == 0) break;
        // ---- the window
        for (int i = tid; i < cells; i += FZ_THREADS) { lkey[i] = FZL_EMPTY; lres[i] = 0xFFFFFFFFu; if (!LPAR) lcp[i] = (unsigned short)i; }
        bool pend = tid < nlive;
        int ea = 0, eb = 0;
        double cost = 0.0;
        int ra = 0, rb = 0, ca = 0, cb = 0;
        if (pend) {
            const unsigned e = wbuf[tid];
            if (LPAR) fz_endpoints(g, (long long)V[e], ea, eb);
            else { ra = wra[tid]; rb = wrb[tid]; }          // (no merge since the collect step walked up to them)
            cost = __longlong_as_double((long long)K[e]);
        }
        __syncthreads();
        FZ_T(7)
        if (pend) {
            if (LPAR) { ra = find(ea); rb = find(eb); }
            if (ra == rb) pend = false;                     // same component for ever
            else if (!LPAR && (hub_on & 2)) {
                // both cells claimed first, then the (up to four) loads of the roots' state in one batch: one global round trip
                // per window set-up instead of two (62 -> 31 M cycles of set-up per full-size image; SPA_FZ_HUB=1 keeps enter()).
                // Measured and NOT kept: fetching the next step's first level of parents before the rounds (-15 M cycles of collect,
                // +15 M here: the single compute unit's gather rate is the bound, not the latency of a level)
                bool oa = false, ob = false;
                ca = claim(ra, oa); cb = claim(rb, ob);
                unsigned sa = 0, sb = 0;
                double cia = 0.0, cib = 0.0;
                if (recs) {
                    FzRec qa, qb;
                    qa.ci = qb.ci = 0.0; qa.size = qb.size = 0;
                    if (oa) qa = SC[ra];
                    if (ob) qb = SC[rb];
                    sa = (unsigned)qa.size; cia = qa.ci; sb = (unsigned)qb.size; cib = qb.ci;
                } else {
                    if (oa) { sa = (unsigned)S[ra]; if (mode == 0) cia = CI[ra]; }
                    if (ob) { sb = (unsigned)S[rb]; if (mode == 0) cib = CI[rb]; }
                }
                if (oa) { lsz[ca] = sa; if (mode == 0) lci[ca] = cia; }
                if (ob) { lsz[cb] = sb; if (mode == 0) lci[cb] = cib; }
            }
            else { ca = enter(ra); cb = enter(rb); }
        }
        if (!LPAR && (hub_on & 1)) e_cost[tid] = cost;
        __syncthreads();
        FZ_T(2)
        for (unsigned round = 0;; ++round) {
            ++rounds;
            const unsigned tag = 0x3FFFFFu - round;
            const unsigned mykey = (tag << 10) | (unsigned)tid;
            bool want = false, resv = false;
            // ---- phase 1: roots and the merge test against the current state; edges that want to merge reserve
            // both components with their position in the window
            if (pend) {
                if (round) {
                    // from the roots of the last round (one LDS read each while they still are roots)
                    if (LPAR) {
                        const int na = find(ra), nb = find(rb);
                        if (na == nb) pend = false;
                        else {
                            if (na != ra) { ra = na; ca = lookup(ra); }
                            if (nb != rb) { rb = nb; cb = lookup(rb); }
                        }
                    } else {
                        ca = cfind(ca); cb = cfind(cb);
                        if (ca == cb) pend = false;
                        else { ra = (int)lkey[ca]; rb = (int)lkey[cb]; }
                    }
                }
                if (pend) {
                    const int wa = (int)csize(ca), wb = (int)csize(cb);
                    if (mode == 0) {
                        const float t0 = (float)(lci[ca] + scale / (double)wa);
                        const float t1 = (float)(lci[cb] + scale / (double)wb);
                        want = cost < (double)(t0 < t1 ? t0 : t1);
                    } else {
                        want = wa < min_size || wb < min_size;
                    }
                    if (want) { atomicMin(lres + ca, mykey); atomicMin(lres + cb, mykey); resv = true; }
                }
            }
            __syncthreads();
            // ---- propagation: an edge that does not want to merge NOW may want to once an earlier reserved edge
            // has changed one of its components, so while it waits it holds its components too
            FZ_T(3)
            for (;;) {
                int changed = 0;
                ++props;
                if (pend && !resv) {
                    const unsigned ma = __hip_atomic_load(lres + ca, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const unsigned mb = __hip_atomic_load(lres + cb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const unsigned pa = (ma >> 10) == tag ? (ma & 1023u) : 0xFFFFFFFFu;
                    const unsigned pb = (mb >> 10) == tag ? (mb & 1023u) : 0xFFFFFFFFu;
                    if (pa < (unsigned)tid || pb < (unsigned)tid) {
                        atomicMin(lres + ca, mykey);
                        atomicMin(lres + cb, mykey);
                        resv = true;
                        changed = 1;
                    }
                }
                if (__syncthreads_or(changed) == 0) break;
            }
            FZ_T(5)
            // ---- hub chains (!LPAR, round 5).  On a smooth full-size image nearly every live edge of a late window joins a
            // small component to ONE giant one; the protocol above decides the earliest of them per round (the others wait for
            // its reservation), 50 rounds per window.  Their decisions are a recurrence over the hub's state alone — internal
            // cost = the last merged edge's, size = the running sum — whenever the other side of each edge is touched by NO other
            // pending edge of the window (then nothing but the hub's own chain can change what the edge sees, and merging it pulls
            // no further edge into the chain).  So: count the pending edges per cell; the earliest reserver of a cell with >=
            // HUB_TH of them opens a hub slot; every pending edge sets its bit in the bitmaps of its cells' slots; one thread per
            // hub walks the bitmap in window order — exactly the sequential pass over those edges — and stops at the first edge whose
            // other side has company.  Walked edges are decided (merged with the state written as phase 2 writes it, or dropped);
            // the rest of the round proceeds as before.
            if (!LPAR && (hub_on & 1) && round >= 1) {
                for (int i = tid; i < cells; i += FZ_THREADS) { lcnt[i] = 0; lslot[i] = 0; }
                if (tid < NHUB * (FZ_THREADS / 32)) hbits[tid] = 0u;
                if (tid == 0) *nhub = 0;
                e_done[tid] = 0;
                __syncthreads();
                if (pend) {
                    atomicAdd((unsigned *)(lcnt + (ca & ~1)), (ca & 1) ? 0x10000u : 1u);       // 16-bit counters, two per word (no carry: <= 2 048 per cell)
                    atomicAdd((unsigned *)(lcnt + (cb & ~1)), (cb & 1) ? 0x10000u : 1u);
                    e_ca[tid] = (unsigned short)ca; e_cb[tid] = (unsigned short)cb;
                }
                __syncthreads();
                if (pend) {
                    const unsigned ma = lres[ca], mb = lres[cb];
                    if (ma == mykey && lcnt[ca] >= HUB_TH) { const int sl = atomicAdd(nhub, 1); if (sl < NHUB) { hcell[sl] = ca; lslot[ca] = (unsigned char)(sl + 1); } }
                    if (mb == mykey && lcnt[cb] >= HUB_TH) { const int sl = atomicAdd(nhub, 1); if (sl < NHUB) { hcell[sl] = cb; lslot[cb] = (unsigned char)(sl + 1); } }
                }
                __syncthreads();
                if (pend) {
                    const int sa = lslot[ca], sb = lslot[cb];
                    if (sa) atomicOr(hbits + (sa - 1) * (FZ_THREADS / 32) + (tid >> 5), 1u << (tid & 31));
                    if (sb) atomicOr(hbits + (sb - 1) * (FZ_THREADS / 32) + (tid >> 5), 1u << (tid & 31));
                    if (mode == 0 && (sa != 0) != (sb != 0)) {        // the leaf side's threshold, off the walking thread's chain
                        const int c = sa ? cb : ca;
                        e_thr[tid] = (float)(lci[c] + scale / (double)(int)lsz[c]);
                    }
                }
                __syncthreads();
                const int nh = min(*nhub, NHUB);
                if (wv < nh) {
                    // one WAVE per hub: lane i of block q stands for window position 64 q + i and fetches that edge's record (its
                    // leaf cell, whether the leaf is private, the leaf's size / root / threshold, the cost) with its own loads;
                    // the walk itself runs over the set lanes in order on wave-uniform values (v_readlane), so an edge costs a
                    // handful of register reads instead of three dependent LDS round trips; lane 0 writes what a merge changes
                    const int h0 = hcell[wv];
                    int hcur = h0, hroot = (int)lkey[h0];
                    unsigned hsz = lsz[h0];
                    double hci = mode == 0 ? lci[h0] : 0.0;
                    bool stop = false, t0_stale = true;
                    float t0 = 0.f;
                    for (int q = 0; q < FZ_THREADS / 64 && !stop; ++q) {
                        const int pp = q * 64 + lane;
                        const bool bit = (hbits[wv * (FZ_THREADS / 32) + (pp >> 5)] >> (pp & 31)) & 1u;
                        int c = 0, okc = 0, croot = 0;
                        unsigned csz = 0;
                        float thr = 0.f;
                        double ecost = 0.0;
                        if (bit) {
                            const int xa = e_ca[pp], xb = e_cb[pp];
                            c = xa == h0 ? xb : xa;
                            okc = lcnt[c] == 1;
                            csz = lsz[c];
                            croot = (int)lkey[c];
                            thr = e_thr[pp];
                            ecost = e_cost[pp];
                        }
                        unsigned long long mask = __ballot(bit);
                        // (round 5, later) the walk in RUNS.  Nearly every walked edge of a late window merges, so the lanes
                        // decide all at once under the assumption that every earlier edge of the block merged — the hub's size
                        // before lane i is then the hub's plus a prefix sum of leaf sizes, its root a prefix minimum of leaf
                        // roots, its internal cost the previous edge's — and the assumption is TRUE for every lane up to the
                        // first one that does not merge: those lanes write their merges in parallel (distinct parents, distinct
                        // dying cells), the first failing lane is dropped (or ends the walk: company), and the rest decides
                        // again from the new state.  Runs shorter than 4 hand the rest of the block to the edge-by-edge loop
                        // below, whose decisions these are by construction (same expressions on the same values).
                        while (mask) {
                            const bool act = (mask >> lane) & 1ull;
                            const int s_in = fz_scan_add(act ? (int)csz : 0);
                            const int m_in = fz_scan_min(act ? croot : 0x7FFFFFFF);
                            const int s_ex = s_in - (act ? (int)csz : 0);
                            int m_ex = __builtin_amdgcn_update_dpp(0x7FFFFFFF, m_in, 0x138, 0xF, 0xF, false);      // wave_shr:1
                            const unsigned long long below = mask & ((1ull << lane) - 1ull);
                            const int prev = below ? 63 - __clzll((long long)below) : -1;
                            const double pcost = __shfl(ecost, prev < 0 ? 0 : prev);
                            const double hci_b = prev < 0 ? hci : pcost;
                            const unsigned hsz_b = hsz + (unsigned)s_ex;
                            const int hroot_b = min(hroot, m_ex);
                            bool w_;
                            if (mode == 0) {
                                const float t0b = (float)(hci_b + scale / (double)(int)hsz_b);
                                w_ = ecost < (double)(t0b < thr ? t0b : thr);
                            } else {
                                w_ = (int)hsz_b < min_size || (int)csz < min_size;
                            }
                            const unsigned long long fail = __ballot(act && (!okc || !w_));
                            const int f = fail ? __ffsll((long long)fail) - 1 : 64;
                            const unsigned long long run = f == 64 ? mask : mask & ((1ull << f) - 1ull);
                            // the hub's cell before lane i: the leaf cell of the latest earlier lane that lowered the root
                            const unsigned long long rec = __ballot(act && croot < hroot_b) & run;
                            const unsigned long long rbelow = rec & ((1ull << lane) - 1ull);
                            const int rprev = rbelow ? 63 - __clzll((long long)rbelow) : -1;
                            const int rcell = __shfl(c, rprev < 0 ? 0 : rprev);
                            const int hcur_b = rprev < 0 ? hcur : rcell;
                            if (act && lane < f) {
                                const bool h_lo = hroot_b < croot;
                                const int lo_r = h_lo ? hroot_b : croot, hi_r = h_lo ? croot : hroot_b;
                                const int surv = h_lo ? hcur_b : c, dead = h_lo ? c : hcur_b;
                                setp(hi_r, lo_r);
                                lcp[dead] = (unsigned short)surv;
                                e_done[pp] = 1;
                            }
                            if (run) {
                                const int L = 63 - __clzll((long long)run);
                                hsz += (unsigned)__builtin_amdgcn_readlane(s_in, L);
                                hroot = min(hroot, __builtin_amdgcn_readlane(m_in, L));
                                if (rec) hcur = __builtin_amdgcn_readlane(c, 63 - __clzll((long long)rec));
                                if (mode == 0) hci = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ecost), L),
                                                                      __builtin_amdgcn_readlane(__double2loint(ecost), L));
                                if (lane == 0) { lsz[hcur] = hsz; if (mode == 0) lci[hcur] = hci; }
                                t0_stale = true;
                            }
                            if (f == 64) { mask = 0; break; }
                            if (!__builtin_amdgcn_readlane(okc, f)) { stop = true; mask = 0; break; }
                            if (lane == 0) e_done[q * 64 + f] = 1;                  // dropped
                            mask &= ~((2ull << f) - 1ull);
                            if (__popcll(run) < 4) break;
                        }
                        while (mask) {
                            const int j = __builtin_amdgcn_readfirstlane(__ffsll((long long)mask) - 1);
                            mask &= mask - 1;
                            if (!__builtin_amdgcn_readlane(okc, j)) { stop = true; break; }
                            const int c_j = __builtin_amdgcn_readlane(c, j), croot_j = __builtin_amdgcn_readlane(croot, j);
                            const unsigned csz_j = (unsigned)__builtin_amdgcn_readlane((int)csz, j);
                            const double cost_j = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ecost), j),
                                                                   __builtin_amdgcn_readlane(__double2loint(ecost), j));
                            bool w_;
                            if (mode == 0) {
                                if (t0_stale) { t0 = (float)(hci + scale / (double)(int)hsz); t0_stale = false; }
                                const float t1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(thr), j));
                                w_ = cost_j < (double)(t0 < t1 ? t0 : t1);
                            } else {
                                w_ = (int)hsz < min_size || (int)csz_j < min_size;
                            }
                            if (w_) {
                                const bool h_lo = hroot < croot_j;
                                const int lo_r = h_lo ? hroot : croot_j, hi_r = h_lo ? croot_j : hroot;
                                const int surv = h_lo ? hcur : c_j, dead = h_lo ? c_j : hcur;
                                const unsigned ns = hsz + csz_j;
                                if (lane == 0) {
                                    setp(hi_r, lo_r);
                                    lsz[surv] = ns;
                                    lcp[dead] = (unsigned short)surv;
                                    if (mode == 0) lci[surv] = cost_j;
                                }
                                if (mode == 0) hci = cost_j;
                                hcur = surv; hroot = lo_r; hsz = ns;
                                t0_stale = true;
                            }
                            if (lane == 0) e_done[q * 64 + j] = 1;
                        }
                    }
                }
                __syncthreads();
                if (pend && e_done[tid]) pend = false;
                FZ_T(6)
            }
            // ---- phase 2: decide every edge no earlier reservation can influence
            int left = 0;
            if (pend) {
                const unsigned ma = lres[ca], mb = lres[cb];
                const unsigned pa = (ma >> 10) == tag ? (ma & 1023u) : 0xFFFFFFFFu;
                const unsigned pb = (mb >> 10) == tag ? (mb & 1023u) : 0xFFFFFFFFu;
                if (pa < (unsigned)tid || pb < (unsigned)tid) {
                    left = 1;                                // waits for an earlier edge
                } else {
                    if (want) {
                        const bool a_lo = ra < rb;
                        const int lo_r = a_lo ? ra : rb, hi_r = a_lo ? rb : ra, c_lo = a_lo ? ca : cb;
                        const unsigned ns = csize(ca) + csize(cb);
                        setp(hi_r, lo_r);
                        if (LPAR) lkey[c_lo] = ((unsigned)lo_r << 16) | ns;
                        else { lsz[c_lo] = ns; lcp[a_lo ? cb : ca] = (unsigned short)c_lo; }
                        if (mode == 0) lci[c_lo] = cost;
                    }
                    pend = false;
                }
            }
            const int nleft = __syncthreads_count(left);
            if (nleft == 0) break;
            if (nleft <= 64) {
                FZ_T(3)
                // ---- the tail of the window on ONE wave.  What is left after the first rounds is a chain of
                // dependent merges into a growing component: a few edges, one more decided per round.  The
                // pending edges move (in order) to the lanes of wave 0, which plays the remaining rounds with
                // wave-level synchronisation only — a round then costs LDS latencies, not workgroup barriers.
                const unsigned long long pm = __ballot(pend);
                if (lane == 0) wave_cnt[wv] = __popcll(pm);
                __syncthreads();
                if (pend) {
                    int off = 0;
#pragma unroll
                    for (int i = 0; i < FZ_THREADS / 64; ++i) if (i < wv) off += wave_cnt[i];
                    const int q = off + (int)spa_rank_in_mask(pm);
                    tail_ea[q] = ra; tail_eb[q] = rb; tail_cost[q] = cost;       // the roots stand for the endpoints
                }
                __syncthreads();
                if (wv == 0) {
                    bool tp = lane < nleft;
                    int tra = tp ? tail_ea[lane] : 0, trb = tp ? tail_eb[lane] : 0, tca = -1, tcb = -1;
                    if (!LPAR && tp) { tca = lookup(tra); tcb = lookup(trb); }       // (the roots handed over are table roots)
                    const double tcost = tp ? tail_cost[lane] : 0.0;
                    for (unsigned tr = round + 1; __ballot(tp); ++tr) {
                        ++trounds;
                        const unsigned tag = 0x3FFFFFu - tr;
                        const unsigned mykey = (tag << 10) | (unsigned)lane;
                        bool want = false, resv = false;
                        if (tp) {
                            if (LPAR) {
                                const int na = find(tra), nb = find(trb);
                                if (na == nb) tp = false;
                                else {
                                    if (na != tra || tca < 0) { tra = na; tca = lookup(tra); }
                                    if (nb != trb || tcb < 0) { trb = nb; tcb = lookup(trb); }
                                }
                            } else {
                                tca = cfind(tca); tcb = cfind(tcb);
                                if (tca == tcb) tp = false;
                                else { tra = (int)lkey[tca]; trb = (int)lkey[tcb]; }
                            }
                        }
                        if (tp) {
                            const int wa = (int)csize(tca), wb = (int)csize(tcb);
                            if (mode == 0) {
                                const float t0 = (float)(lci[tca] + scale / (double)wa);
                                const float t1 = (float)(lci[tcb] + scale / (double)wb);
                                want = tcost < (double)(t0 < t1 ? t0 : t1);
                            } else {
                                want = wa < min_size || wb < min_size;
                            }
                            if (want) { atomicMin(lres + tca, mykey); atomicMin(lres + tcb, mykey); resv = true; }
                        }
                        fz_wave_sync();
                        for (;;) {
                            bool changed = false;
                            if (tp && !resv) {
                                const unsigned ma = __hip_atomic_load(lres + tca, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                const unsigned mb = __hip_atomic_load(lres + tcb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                const unsigned pa = (ma >> 10) == tag ? (ma & 1023u) : 0xFFFFFFFFu;
                                const unsigned pb = (mb >> 10) == tag ? (mb & 1023u) : 0xFFFFFFFFu;
                                if (pa < (unsigned)lane || pb < (unsigned)lane) {
                                    atomicMin(lres + tca, mykey);
                                    atomicMin(lres + tcb, mykey);
                                    resv = true;
                                    changed = true;
                                }
                            }
                            fz_wave_sync();
                            if (!__ballot(changed)) break;
                        }
                        if (tp) {
                            const unsigned ma = __hip_atomic_load(lres + tca, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            const unsigned mb = __hip_atomic_load(lres + tcb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            const unsigned pa = (ma >> 10) == tag ? (ma & 1023u) : 0xFFFFFFFFu;
                            const unsigned pb = (mb >> 10) == tag ? (mb & 1023u) : 0xFFFFFFFFu;
                            if (!(pa < (unsigned)lane || pb < (unsigned)lane)) {
                                if (want) {
                                    const bool a_lo = tra < trb;
                                    const int lo_r = a_lo ? tra : trb, hi_r = a_lo ? trb : tra, c_lo = a_lo ? tca : tcb;
                                    const unsigned ns = csize(tca) + csize(tcb);
                                    setp(hi_r, lo_r);
                                    if (LPAR) lkey[c_lo] = ((unsigned)lo_r << 16) | ns;
                                    else { lsz[c_lo] = ns; lcp[a_lo ? tcb : tca] = (unsigned short)c_lo; }
                                    if (mode == 0) lci[c_lo] = tcost;
                                }
                                tp = false;
                            }
                        }
                        fz_wave_sync();
                    }
                }
                __syncthreads();
                FZ_T(4)
                break;
            }
        }
        FZ_T(3)
        // the window's cells back to the global state (a cell whose root was merged away holds stale values for a
        // pixel that is no root any more: never read again)
        for (int i = tid; i < cells; i += FZ_THREADS) {
            const unsigned w = lkey[i];
            if (w != FZL_EMPTY) {
                if (recs) {
                    if (mode == 0) { FzRec r; r.ci = lci[i]; r.size = (int)csize(i); r.pad = 0; SC[keyroot(w)] = r; }
                    else SC[keyroot(w)].size = (int)csize(i);
                } else {
                    S[keyroot(w)] = (int)csize(i);
                    if (mode == 0) CI[keyroot(w)] = lci[i];
                }
            }
        }
        __syncthreads();
    }
    if (LPAR) {
        __syncthreads();
        for (int p = tid; p < npix; p += FZ_THREADS) { const unsigned v = lpar[p]; P[p] = v == 0xFFFFu ? -1 : (int)v; }
    }
    if (tid == 0 && seg) seg[2 * b] = (int)nEb;             // the next segment (or the filter in front of it) starts here
    if (tid == 0 && diag) {
        atomicAdd(diag + 0, win); atomicAdd(diag + 1, chunks); atomicAdd(diag + 2, rounds); atomicAdd(diag + 3, trounds);
        for (int i = 0; i < 8; ++i) atomicAdd(diag + 4 + i, (int)(tc[i] >> 10));
        atomicAdd(diag + 12, props);
    }
}

// labels = rank of the root among the roots in raster order (np.unique(flat, return_inverse=True)[1])
__global__ __launch_bounds__(256) void k_fz_count_roots(const int *__restrict__ parent, int npix,
                                                        int *__restrict__ blk, int nblk)
{
    __shared__ int ws[4];
    const int b = blockIdx.y;
    const int *P = parent + (long long)b * npix;
    const int base = blockIdx.x * 1024 + threadIdx.x * 4;
    int c = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) c += (base + i < npix && P[base + i] < 0) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk[(long long)b * nblk + blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(256) void k_fz_scan(int *__restrict__ blk, int nblk, int32_t *__restrict__ n_labels)
{
    __shared__ int part[256];
    const int b = blockIdx.x;
    int *B_ = blk + (long long)b * nblk;
    const int per = (nblk + 255) / 256;
    const int lo = threadIdx.x * per, hi = min(nblk, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += B_[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) { int t = part[i]; part[i] = run; run += t; }
        n_labels[b] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { int t = B_[i]; B_[i] = run; run += t; }
}

__global__ __launch_bounds__(256) void k_fz_number(const int *__restrict__ parent, int npix,
                                                   const int *__restrict__ blk, int nblk,
                                                   int *__restrict__ rank)
{
    __shared__ int ws[4];
    const int b = blockIdx.y;
    const int *P = parent + (long long)b * npix;
    int *R = rank + (long long)b * npix;
    const int base = blockIdx.x * 1024 + threadIdx.x * 4;
    bool k[4];
    int c = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { k[i] = base + i < npix && P[base + i] < 0; c += k[i] ? 1 : 0; }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = c;
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) ws[wv] = inc;
    __syncthreads();
    int off = blk[(long long)b * nblk + blockIdx.x];
    for (int i = 0; i < wv; ++i) off += ws[i];
    int r = off + inc - c;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (k[i]) R[base + i] = r++;
}

__global__ __launch_bounds__(256) void k_fz_relabel(const int *__restrict__ parent,
                                                    const int *__restrict__ rank, int npix,
                                                    int32_t *__restrict__ out)
{
    const int b = blockIdx.y;
    const int *P = parent + (long long)b * npix;
    const int *R = rank + (long long)b * npix;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256)
        out[(long long)b * npix + p] = R[fz_find(P, p)];
}

static int fz_run(spa_ctx *ctx, const float *rgb, int32_t B, int32_t H, int32_t W, double scale,
                  double sigma, int32_t min_size, int32_t *labels, int32_t *n_labels, void *stream,
                  bool wide);

extern "C" int spa_felzenszwalb(spa_ctx *ctx, const float *rgb, int32_t B, int32_t H, int32_t W,
                                double scale, double sigma, int32_t min_size, int32_t *labels,
                                int32_t *n_labels, void *stream)
{
    return fz_run(ctx, rgb, B, H, W, scale, sigma, min_size, labels, n_labels, stream, false);
}

extern "C" int spa_felzenszwalb_u8(spa_ctx *ctx, const float *rgb, int32_t B, int32_t H, int32_t W,
                                   double scale, double sigma, int32_t min_size, int32_t *labels,
                                   int32_t *n_labels, void *stream)
{
    return fz_run(ctx, rgb, B, H, W, scale, sigma, min_size, labels, n_labels, stream, true);
}

// ---- filters of a large image's passes (round 5).  The passes' single workgroup spent half its time testing sorted positions
// whose endpoints already share a component (and, in the clean-up pass, all 8.4 M positions again for the handful of edges that
// touch a component below min_size).  Dead edges stay dead — components only grow —, so every compute unit drops them ahead of
// the pass: the forest is flattened, the survivors of the list's unprocessed part are counted per block of 1 024 list
// positions, scanned, and their positions in the sorted list written out in order.  Pass 0 runs as SPA_FZ_SEGMENTS launches with
// such a filter between two of them, the clean-up pass as one launch behind a filter over the whole sorted list.
__global__ __launch_bounds__(256) void k_fz_flatten_all(int *__restrict__ parent, int npix)
{
    int *P = parent + (long long)blockIdx.y * npix;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        const int q = P[p];
        if (q < 0) continue;
        int r = q, n;
        while ((n = P[r]) >= 0) r = n;
        if (r != q) P[p] = r;
    }
}
__global__ void k_fz_seg_init(int *__restrict__ seg, const int *__restrict__ zcount, int nE, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { seg[2 * b] = zcount[b]; seg[2 * b + 1] = nE; }
}
// mode 0: the endpoints lie in different components; mode 1: ... one of which is below min_size
__device__ __forceinline__ bool fz_live_edge(const FzGeom &g, const unsigned *V, const int *P, const int *S, unsigned oe, int mode, int min_size,
                                             const FzRec *SC)
{
    int a, c;
    fz_endpoints(g, (long long)V[oe], a, c);
    const int pa = P[a], pc = P[c];
    int ra = pa < 0 ? a : pa, rc = pc < 0 ? c : pc;
    int n;
    while ((n = P[ra]) >= 0) ra = n;             // (flattened: the root already)
    while ((n = P[rc]) >= 0) rc = n;
    if (ra == rc) return false;
    if (mode == 0) return true;
    return SC ? (SC[ra].size < min_size || SC[rc].size < min_size) : (S[ra] < min_size || S[rc] < min_size);
}
__global__ __launch_bounds__(256) void k_fz_live_count(const unsigned *__restrict__ vals, FzGeom g, const int *__restrict__ parent,
                                                       const int *__restrict__ size, const unsigned *__restrict__ idx,
                                                       const int *__restrict__ seg, int mode, int min_size, int npix,
                                                       int *__restrict__ blkcnt, int nblk, const FzRec *__restrict__ rec)
{
    const FzRec *SC = rec ? rec + (long long)blockIdx.y * npix : nullptr;
    const int b = blockIdx.y;
    const unsigned *V = vals + (long long)b * g.nE;
    const unsigned *IX = idx ? idx + (long long)b * g.nE : nullptr;
    const int *P = parent + (long long)b * npix, *S = size + (long long)b * npix;
    const long long lo = seg[2 * b], hi = seg[2 * b + 1];
    __shared__ int wc[4];
    for (int kb = blockIdx.x; kb < nblk; kb += gridDim.x) {
        int cnt = 0;
        if ((long long)(kb + 1) * 1024 > lo && (long long)kb * 1024 < hi) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long e = (long long)kb * 1024 + u * 256 + threadIdx.x;
                const bool live = e >= lo && e < hi && fz_live_edge(g, V, P, S, IX ? IX[e] : (unsigned)e, mode, min_size, SC);
                cnt += __popcll(__ballot(live));
            }
        }
        if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) blkcnt[(long long)b * nblk + kb] = wc[0] + wc[1] + wc[2] + wc[3];
        __syncthreads();
    }
}
// exclusive scan of an image's block counts (in place); the new list is [0, total): seg_out
__global__ __launch_bounds__(1024) void k_fz_live_scan(int *__restrict__ blkcnt, int nblk, int *__restrict__ seg_out)
{
    int *c = blkcnt + (long long)blockIdx.x * nblk;
    __shared__ int part[1024];
    const int per = (nblk + 1023) / 1024, lo = min(nblk, (int)threadIdx.x * per), hi = min(nblk, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += c[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 1024; ++i) { const int v = part[i]; part[i] = run; run += v; }
        seg_out[2 * blockIdx.x] = 0;
        seg_out[2 * blockIdx.x + 1] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { const int v = c[i]; c[i] = run; run += v; }
}
__global__ __launch_bounds__(256) void k_fz_live_copy(const unsigned *__restrict__ vals, FzGeom g, const int *__restrict__ parent,
                                                      const int *__restrict__ size, const unsigned *__restrict__ idx,
                                                      const int *__restrict__ seg, int mode, int min_size, int npix,
                                                      const int *__restrict__ blkoff, int nblk, unsigned *__restrict__ idx_out,
                                                      const FzRec *__restrict__ rec)
{
    const FzRec *SC = rec ? rec + (long long)blockIdx.y * npix : nullptr;
    const int b = blockIdx.y;
    const unsigned *V = vals + (long long)b * g.nE;
    const unsigned *IX = idx ? idx + (long long)b * g.nE : nullptr;
    unsigned *OX = idx_out + (long long)b * g.nE;
    const int *P = parent + (long long)b * npix, *S = size + (long long)b * npix;
    const long long lo = seg[2 * b], hi = seg[2 * b + 1];
    __shared__ int wc[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int kb = blockIdx.x; kb < nblk; kb += gridDim.x) {
        if (!((long long)(kb + 1) * 1024 > lo && (long long)kb * 1024 < hi)) continue;
        int base = blkoff[(long long)b * nblk + kb];
        for (int u = 0; u < 4; ++u) {
            const long long e = (long long)kb * 1024 + u * 256 + threadIdx.x;
            const unsigned oe = (e >= lo && e < hi) ? (IX ? IX[e] : (unsigned)e) : 0u;
            const bool live = e >= lo && e < hi && fz_live_edge(g, V, P, S, oe, mode, min_size, SC);
            const unsigned long long m = __ballot(live);
            if (lane == 0) wc[wv] = __popcll(m);
            __syncthreads();
            int off = base;
            for (int i = 0; i < wv; ++i) off += wc[i];
            if (live) OX[off + (int)spa_rank_in_mask(m)] = oe;
            base += wc[0] + wc[1] + wc[2] + wc[3];
            __syncthreads();
        }
    }
}

static int fz_run(spa_ctx *ctx, const float *rgb, int32_t B, int32_t H, int32_t W, double scale,
                  double sigma, int32_t min_size, int32_t *labels, int32_t *n_labels, void *stream,
                  bool wide)
{
    SPA_ARG(ctx && rgb && labels && n_labels && B > 0 && H > 1 && W > 1 && sigma > 0.0 && scale > 0.0);
    SPA_ARG((long long)H * W < (1ll << 28));
    {
        // the persistent passes need every workgroup resident: at most n_cu images per launch
        // (one workgroup per image: 256 full-size images in one launch keep every CU busy — 64 per launch measured
        // 12.6 ms per image amortised, see HISTORY.md section 7; SPA_FZ_MAXB for experiments)
        int maxB = ctx->n_cu < FZ_MAXB ? ctx->n_cu : FZ_MAXB;
        if (const char *e = getenv("SPA_FZ_MAXB")) { const int v = atoi(e); if (v > 0 && v < maxB) maxB = v; }
        if (B > maxB) {
            const long long px = (long long)H * W;
            for (int b0 = 0; b0 < B; b0 += maxB) {
                const int nb = B - b0 < maxB ? B - b0 : maxB;
                int rc0 = fz_run(ctx, rgb + (long long)b0 * 3 * px, nb, H, W, scale, sigma, min_size,
                                 labels + (long long)b0 * px, n_labels + b0, stream, wide);
                if (rc0 != SPA_OK) return rc0;
            }
            return SPA_OK;
        }
    }
    hipStream_t s = spa_stream(stream);
    FzWeights fw;
    fw.r = fz_weights(sigma, fw.w);
    SPA_ARG(fw.r >= 0);
    const long long npix = (long long)H * W;
    FzGeom g;
    g.H = H; g.W = W;
    g.nR = (long long)H * (W - 1); g.nD = (long long)(H - 1) * W; g.nDR = (long long)(H - 1) * (W - 1);
    g.nE = g.nR + g.nD + 2 * g.nDR;
    SPA_ARG(g.nE < (1ll << 32));

    // workspaces (the connectivity pass of SLIC is not running at the same time: share its slots)
    double *sm0, *sm1, *cint;
    unsigned long long *keys0, *keys1, *mark;
    unsigned *vals0, *vals1;
    int *parent, *size, *rank, *blk;
    FzImg *st;
    void *tmp;
    int rc;
    const size_t planes = (size_t)B * 3 * npix * 8;
    const int nblk = (int)((npix + 1023) / 1024);
    if ((rc = spa_ws_reserve(ctx, WS_LAB, 2 * planes, (void **)&sm0)) != SPA_OK) return rc;
    sm1 = sm0 + (size_t)B * 3 * npix;
    if ((rc = spa_ws_reserve(ctx, WS_FZ_KEYS, (size_t)B * g.nE * 8 * 2, (void **)&keys0)) != SPA_OK) return rc;
    keys1 = keys0 + (size_t)B * g.nE;
    if ((rc = spa_ws_reserve(ctx, WS_FZ_VALS, (size_t)B * g.nE * 4 * 2, (void **)&vals0)) != SPA_OK) return rc;
    vals1 = vals0 + (size_t)B * g.nE;
    if ((rc = spa_ws_reserve(ctx, WS_PARENT, (size_t)B * npix * 4, (void **)&parent)) != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_SIZE, (size_t)B * npix * 4, (void **)&size)) != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_FINAL, (size_t)B * npix * 4, (void **)&rank)) != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_FZ_STATE, (size_t)B * npix * 16, (void **)&cint)) != SPA_OK) return rc;
    mark = (unsigned long long *)(cint + (size_t)B * npix);
    if ((rc = spa_ws_reserve(ctx, WS_BLK, (size_t)B * 2 * nblk * 4, (void **)&blk)) != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_CONNMISC, FZ_MAXB * sizeof(FzImg) + (FZ_MAXB + 64) * sizeof(int), (void **)&st)) != SPA_OK) return rc;
    size_t tmp_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys0, keys1, vals0, vals1, (int)g.nE, 0, 64, s);
    tmp_bytes = (tmp_bytes + 255) & ~(size_t)255;
    // small images: a sort of a few hundred thousand keys is ~8 short launches, so a batch of sorts is launch
    // bound on one stream — they are dealt to three streams (own scratch each)
    const bool par_sort = B > 2 && g.nE <= (1ll << 21);
    if ((rc = spa_ws_reserve(ctx, WS_FZ_TMP, tmp_bytes * (par_sort ? 3 : 1), &tmp)) != SPA_OK) return rc;
    if (par_sort && (rc = spa_aux_streams(ctx)) != SPA_OK) return rc;

    int gx = (int)((npix + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_fz_blur_y, dim3(gx, B * 3), dim3(256), 0, s, rgb, sm0, H, W, fw, wide);
    hipLaunchKernelGGL(k_fz_blur_x, dim3(gx, B * 3), dim3(256), 0, s, (const double *)sm0, sm1, H, W, fw);
    int ge = (int)((g.nE + 255) / 256);
    if (ge > 2048) ge = 2048;
    hipLaunchKernelGGL(k_fz_costs, dim3(ge, B), dim3(256), 0, s, (const double *)sm1, g, keys0, vals0);
    if (par_sort) {
        SPA_HIP(hipEventRecord(ctx->ev_fork, s));
        SPA_HIP(hipStreamWaitEvent(ctx->aux[0], ctx->ev_fork, 0));
        SPA_HIP(hipStreamWaitEvent(ctx->aux[1], ctx->ev_fork, 0));
    }
    for (int b = 0; b < B; ++b) {
        hipStream_t sb = par_sort ? (b % 3 == 0 ? s : ctx->aux[b % 3 - 1]) : s;
        void *tb = (char *)tmp + (par_sort ? (size_t)(b % 3) * tmp_bytes : 0);
        SPA_HIP(hipcub::DeviceRadixSort::SortPairs(tb, tmp_bytes, keys0 + (size_t)b * g.nE, keys1 + (size_t)b * g.nE,
                                                   vals0 + (size_t)b * g.nE, vals1 + (size_t)b * g.nE, (int)g.nE,
                                                   0, 64, sb));
    }
    if (par_sort) {
        SPA_HIP(hipEventRecord(ctx->ev_join[0], ctx->aux[0]));
        SPA_HIP(hipEventRecord(ctx->ev_join[1], ctx->aux[1]));
        SPA_HIP(hipStreamWaitEvent(s, ctx->ev_join[0], 0));
        SPA_HIP(hipStreamWaitEvent(s, ctx->ev_join[1], 0));
    }
    hipLaunchKernelGGL(k_fz_init, dim3(1024), dim3(256), 0, s, parent, size, cint, mark, (long long)B * npix, st, B);
    int *zcount = (int *)((char *)st + FZ_MAXB * sizeof(FzImg));
    hipLaunchKernelGGL(k_fz_zero_count, dim3(B), dim3(64), 0, s, (const unsigned long long *)keys1, g.nE, zcount);
    hipLaunchKernelGGL(k_fz_zero_union, dim3(256, B), dim3(256), 0, s, (const unsigned *)vals1, g,
                       (const int *)zcount, parent);
    hipLaunchKernelGGL(k_fz_zero_sizes, dim3((unsigned)((npix + 255) / 256), B), dim3(256), 0, s,
                       (const int *)zcount, parent, size, (int)npix);
    // persistent passes: every workgroup of the grid must be resident (one per CU at most)
    // The passes are bound by the number of reservation rounds (chains of dependent merges), not by
    // work per round: ONE 1024-thread workgroup per image makes a round cost a workgroup barrier
    // (~1 us) instead of an agent-scope barrier across workgroups (~10 us each, several per round).
    // SPA_FZ_GROUP overrides (experiments).
    int G = 1;
    if (const char *e = getenv("SPA_FZ_GROUP")) G = atoi(e) > 0 ? atoi(e) : 1;
    if ((long long)G * B > ctx->n_cu) G = ctx->n_cu / B > 0 ? ctx->n_cu / B : 1;
    SPA_ARG((long long)G * B <= ctx->n_cu);
    long long fdiv = 4;          // about 4 sweeps per image worth of edges (measured: 2-4 best, 16+ slower)
    if (const char *e = getenv("SPA_FZ_FLATTEN_DIV")) fdiv = atoi(e) > 0 ? atoi(e) : 4;       // experiments
    long long fe = npix / (fdiv * G * FZ_THREADS * FZ_EPT);
    const int flatten_every = fe < 1 ? 1 : (int)fe;
    // scale = float(scale) / 255.
    const double k = scale / 255.0;
    const bool ldsp = G == 1 && npix <= 65535;
    const size_t lds_par = ldsp ? (size_t)npix * 2 : 0;
    if (ldsp && !(ctx->fz_attr_done & 1)) {
        SPA_HIP(hipFuncSetAttribute((const void *)k_fz_pass<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        ctx->fz_attr_done |= 1;
    }
    // the window's root table (+ the parent array of a small image) in LDS
    const size_t lds_limit = 156 * 1024;
    const size_t par_bytes = ((size_t)npix * 2 + 15) & ~(size_t)15;
    // (SPA_FZ_LPAR=0: experiments — small images through the large images' kernel: parent array in global memory, hub chains)
    const bool lpar = G == 1 && npix <= 65535 && par_bytes + 2560 * 16 + FZ_THREADS * 4 <= lds_limit &&
                      !(getenv("SPA_FZ_LPAR") && atoi(getenv("SPA_FZ_LPAR")) == 0);
    long long cells = lpar ? (long long)((lds_limit - par_bytes - FZ_THREADS * 4) / 16) : 4096;
    if (cells > 4096) cells = 4096;
    // (larger images.  Round 3: the table kernel with the parent array in L2 measured 0.90 s per full-size image against 0.60 s of
    // k_fz_pass — a round of it still chased the parent array through L2.  Round 5: the window's own merges form a forest over
    // the table's CELLS (lcp), so a round finds its roots through LDS alone: 0.40 s against 0.52 s alone, 17.3 against 22.4 ms per
    // image at batch 30, the same labels — the table kernel serves every image size; SPA_FZ_TAB_LARGE=0 keeps k_fz_pass)
    const char *tab_large = getenv("SPA_FZ_TAB_LARGE");
    const bool tab = G == 1 && !getenv("SPA_FZ_NO_LDS_STATE") && (lpar || !tab_large || atoi(tab_large) != 0);
    // (!lpar: + the hub-chain state: cells * 3 + per-edge cells / flags / costs + 16 bitmaps)
    const size_t tab_lds = lpar ? par_bytes + (size_t)cells * 16 + FZ_THREADS * 4
                                : (size_t)cells * 22 + FZ_THREADS * 4 + (size_t)cells * 3 + FZ_THREADS * 13 + 16 + 16 * (FZ_THREADS / 32) * 4 + 16 * 4 + 16 + FZ_THREADS * 4 + FZ_THREADS * 8;
    // hub chains (k_fz_pass_tab, !LPAR): on by default — one 1024 x 2048 image alone 0.40 -> 0.22 s (109 000 -> 8 000 full rounds),
    // 16.9 -> 10.2 ms per image at batch 30, the same labels; SPA_FZ_HUB=0 switches them off
    const char *hub_env = getenv("SPA_FZ_HUB");
    const int hub_on = hub_env ? atoi(hub_env) : 7;          // bit 0: hub chains, bit 1: batched table entry, bit 2: 16-byte root records (99 against 102 ms alone)
    if (tab && !(ctx->fz_attr_done & 2)) {
        SPA_HIP(hipFuncSetAttribute((const void *)k_fz_pass_tab<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
        SPA_HIP(hipFuncSetAttribute((const void *)k_fz_pass_tab<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
        ctx->fz_attr_done |= 2;
    }
    int *diag = zcount + FZ_MAXB;              // windows, chunks, rounds of the batch (diagnostics, spa_debug_peek)
    // large images: pass 0 in SPA_FZ_SEGMENTS launches (default 8) with a filter of the remaining sorted list between two of them, the
    // clean-up pass behind a filter over the whole list (the filter kernels above); SPA_FZ_PREFILTER=0: one launch per pass over
    // the sorted list itself, as round 4 ran them
    const char *pf_env = getenv("SPA_FZ_PREFILTER");
    const int nblk_e = (int)((g.nE + 1023) / 1024);
    // (block counts and segment words in the radix sort's scratch — the sorts are done —, the lists in the sort's INPUT key buffer)
    const bool prefilter = tab && !lpar && (!pf_env || atoi(pf_env) != 0) &&
                           (size_t)B * nblk_e * 4 + (size_t)B * 32 <= tmp_bytes * (par_sort ? 3 : 1);
    int nseg = B < 4 ? 16 : 8;             // (measured at B = 1 / 8 / 30: 116 / 158 / 255 ms with 16, 122 / 152 / 231 ms with 8)
    if (const char *e = getenv("SPA_FZ_SEGMENTS")) nseg = atoi(e) > 0 ? atoi(e) : 1;
    if (prefilter) {
        int *blkcnt = (int *)tmp;
        int *segA = blkcnt + (size_t)B * nblk_e, *segB = segA + 2 * B, *segF = segB + 2 * B;
        unsigned *idxA = (unsigned *)keys0, *idxB = idxA + (size_t)B * g.nE;
        int gp = (int)((npix + 255) / 256);
        if (gp > 1024) gp = 1024;
        const int gb = nblk_e < 2048 ? nblk_e : 2048;
        // (hub_on bit 2: size and internal cost of a root in one 16-byte record, the table pass's only view of them)
        const FzRec *recp = (hub_on & 4) ? (const FzRec *)cint : nullptr;
        if (recp) hipLaunchKernelGGL(k_fz_records, dim3(2048), dim3(256), 0, s, (FzRec *)cint, (const int *)size, (long long)B * npix);
        auto filter = [&](const unsigned *in_idx, const int *seg_in, int fmode, int *seg_out, unsigned *out_idx) {
            hipLaunchKernelGGL(k_fz_flatten_all, dim3(gp, B), dim3(256), 0, s, parent, (int)npix);
            hipLaunchKernelGGL(k_fz_live_count, dim3(gb, B), dim3(256), 0, s, (const unsigned *)vals1, g, (const int *)parent,
                               (const int *)size, in_idx, seg_in, fmode, min_size, (int)npix, blkcnt, nblk_e, recp);
            hipLaunchKernelGGL(k_fz_live_scan, dim3(B), dim3(1024), 0, s, blkcnt, nblk_e, seg_out);
            hipLaunchKernelGGL(k_fz_live_copy, dim3(gb, B), dim3(256), 0, s, (const unsigned *)vals1, g, (const int *)parent,
                               (const int *)size, in_idx, seg_in, fmode, min_size, (int)npix, (const int *)blkcnt, nblk_e, out_idx, recp);
        };
        auto pass = [&](int pmode, const unsigned *in_idx, int *seg_io, int div, int skip_flat) {
            hipLaunchKernelGGL(k_fz_pass_tab<false>, dim3(B), dim3(FZ_THREADS), tab_lds, s,
                               (const unsigned long long *)keys1, (const unsigned *)vals1, g, parent, size, cint, k,
                               min_size, pmode, (const int *)zcount, flatten_every, (int)cells, diag, hub_on, in_idx, seg_io, div, skip_flat);
        };
        hipLaunchKernelGGL(k_fz_seg_init, dim3((B + 63) / 64), dim3(64), 0, s, segA, (const int *)zcount, (int)g.nE, B);
        hipLaunchKernelGGL(k_fz_seg_init, dim3((B + 63) / 64), dim3(64), 0, s, segF, (const int *)zcount, (int)g.nE, B);
        const unsigned *cur = nullptr;
        int *sc = segA, *sn = segB;
        unsigned *other = idxA;
        for (int i = 0; i < nseg; ++i) {
            pass(0, cur, sc, nseg - i, i > 0);
            if (i + 1 < nseg) {
                filter(cur, sc, 0, sn, other);
                cur = other;
                other = other == idxA ? idxB : idxA;
                int *t = sc; sc = sn; sn = t;
            }
        }
        filter(nullptr, segF, 1, sn, other);
        pass(1, other, sn, 1, 1);
    }
    // small images (parent array in LDS): pass 0 as one launch, then the same filter in front of the clean-up pass — the pass kernel
    // wrote the parent array back, the other compute units flatten it and list the few edges that can still merge a component below
    // min_size, and the clean-up launch walks that list instead of testing every sorted position again (SPA_FZ_PREFILTER=0: off)
    const bool prefilter_small = tab && lpar && !prefilter && (!pf_env || atoi(pf_env) != 0) &&
                                 (size_t)B * nblk_e * 4 + (size_t)B * 32 <= tmp_bytes * (par_sort ? 3 : 1);
    if (prefilter_small) {
        int *blkcnt = (int *)tmp;
        int *segF = blkcnt + (size_t)B * nblk_e, *segL = segF + 2 * B;
        unsigned *idxA = (unsigned *)keys0;
        int gp = (int)((npix + 255) / 256);
        if (gp > 1024) gp = 1024;
        const int gb = nblk_e < 2048 ? nblk_e : 2048;
        hipLaunchKernelGGL(k_fz_pass_tab<true>, dim3(B), dim3(FZ_THREADS), tab_lds, s,
                           (const unsigned long long *)keys1, (const unsigned *)vals1, g, parent, size, cint, k,
                           min_size, 0, (const int *)zcount, flatten_every, (int)cells, diag);
        hipLaunchKernelGGL(k_fz_seg_init, dim3((B + 63) / 64), dim3(64), 0, s, segF, (const int *)zcount, (int)g.nE, B);
        hipLaunchKernelGGL(k_fz_flatten_all, dim3(gp, B), dim3(256), 0, s, parent, (int)npix);
        hipLaunchKernelGGL(k_fz_live_count, dim3(gb, B), dim3(256), 0, s, (const unsigned *)vals1, g, (const int *)parent,
                           (const int *)size, (const unsigned *)nullptr, (const int *)segF, 1, min_size, (int)npix, blkcnt, nblk_e,
                           (const FzRec *)nullptr);
        hipLaunchKernelGGL(k_fz_live_scan, dim3(B), dim3(1024), 0, s, blkcnt, nblk_e, segL);
        hipLaunchKernelGGL(k_fz_live_copy, dim3(gb, B), dim3(256), 0, s, (const unsigned *)vals1, g, (const int *)parent,
                           (const int *)size, (const unsigned *)nullptr, (const int *)segF, 1, min_size, (int)npix, (const int *)blkcnt,
                           nblk_e, idxA, (const FzRec *)nullptr);
        hipLaunchKernelGGL(k_fz_pass_tab<true>, dim3(B), dim3(FZ_THREADS), tab_lds, s,
                           (const unsigned long long *)keys1, (const unsigned *)vals1, g, parent, size, cint, k,
                           min_size, 1, (const int *)zcount, flatten_every, (int)cells, diag, 0, (const unsigned *)idxA, segL, 1, 1);
    }
    for (int mode = 0; mode < 2 && !prefilter && !prefilter_small; ++mode) {
        const unsigned r0 = mode ? 0x40000000u : 0u;
        if (tab && lpar)
            hipLaunchKernelGGL(k_fz_pass_tab<true>, dim3(B), dim3(FZ_THREADS), tab_lds, s,
                               (const unsigned long long *)keys1, (const unsigned *)vals1, g, parent, size, cint, k,
                               min_size, mode, (const int *)zcount, flatten_every, (int)cells, diag);
        else if (tab)
            hipLaunchKernelGGL(k_fz_pass_tab<false>, dim3(B), dim3(FZ_THREADS), tab_lds, s,
                               (const unsigned long long *)keys1, (const unsigned *)vals1, g, parent, size, cint, k,
                               min_size, mode, (const int *)zcount, flatten_every, (int)cells, diag, hub_on & 3);
        else if (ldsp)
            hipLaunchKernelGGL(k_fz_pass<true>, dim3(G, B), dim3(FZ_THREADS), lds_par, s, (const unsigned long long *)keys1,
                               (const unsigned *)vals1, g, parent, size, cint, mark, st, k, min_size, mode, r0,
                               (const int *)zcount, flatten_every, ctx->d_status);
        else
            hipLaunchKernelGGL(k_fz_pass<false>, dim3(G, B), dim3(FZ_THREADS), 0, s, (const unsigned long long *)keys1,
                               (const unsigned *)vals1, g, parent, size, cint, mark, st, k, min_size, mode, r0,
                               (const int *)zcount, flatten_every, ctx->d_status);
    }
    hipLaunchKernelGGL(k_fz_count_roots, dim3(nblk, B), dim3(256), 0, s, parent, (int)npix, blk, nblk);
    hipLaunchKernelGGL(k_fz_scan, dim3(B), dim3(256), 0, s, blk, nblk, n_labels);
    hipLaunchKernelGGL(k_fz_number, dim3(nblk, B), dim3(256), 0, s, parent, (int)npix, blk, nblk, rank);
    int gr = (int)((npix + 255) / 256);
    if (gr > 2048) gr = 2048;
    hipLaunchKernelGGL(k_fz_relabel, dim3(gr, B), dim3(256), 0, s, parent, rank, (int)npix, labels);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
