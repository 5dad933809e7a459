// Weighted k-means of the reference (batch_spalign_kmeans.py:136-183) as ONE persistent launch.
//
// The reference issues ~10 CuPy kernels and k host synchronisations per Lloyd iteration; the
// problem itself is tiny (N = a few thousand superpixels of a batch, D = 514, k <= 8), so it
// is latency bound.  Here one cooperative grid (<= one 256-thread workgroup per CU, all
// co-resident) keeps the centres in LDS and runs every phase — median threshold of the prior
// (:144), initial assignment (:141-149), unweighted initial centres (:150-151), assignment
// (:155-157), convergence test (:158-159), weighted centre update (:163-171), empty-cluster
// exit (:173-181) — separated by an agent-scope grid barrier, with the convergence flag and the
// iteration count kept on the device.
//
// Determinism: every sum has a fixed order (per-workgroup sequential partial sums over a
// contiguous slice of points, then a sequential sum over workgroups), so assignments do not
// depend on scheduling.  Distances are sqrt of float64 sums like the reference's linalg.norm;
// argmin takes the first minimum and lets a NaN distance win (numpy semantics, which is what
// makes an initially empty cluster swallow every point).
#include "spa_common.h"
#include <stdlib.h>

#define KM_MAXK 8
#define KM_THREADS 1024      // 16 waves per workgroup: more points of the slice in flight per barrier interval
#define KM_COMB 4096          // doubles of LDS for staging partial sums (>= workgroups * clusters)

struct KmShared {
    unsigned barrier;      // monotonic arrival counter
    int n_changed;         // per-iteration flags live in `changed[]`
    int thr_set;
    int pad;
    double thr;
};

__device__ __forceinline__ void grid_sync(unsigned *ctr, unsigned G, unsigned &epoch,
                                          uint32_t *status)
{
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
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1ll << 26)) { atomicOr(status, SPA_ST_KMEANS_BARRIER); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// sum of n LDS doubles at stride `stride`, in index order; 16 reads are issued together so that the
// ordered additions do not each wait for an LDS round trip
__device__ __forceinline__ double km_ordered_sum(const double *p, int n, int stride)
{
    double s = 0.0;
    for (int q0 = 0; q0 < n; q0 += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p[(q0 + u < n ? q0 + u : n - 1) * stride];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (q0 + u < n) s = s + v[u];
    }
    return s;
}

template <typename T>
__device__ __forceinline__ double ldx(const T *X, long long i) { return (double)X[i]; }

// info: {iterations, status, N, -}
template <typename T>
__global__ __launch_bounds__(KM_THREADS) void k_kmeans(const T *__restrict__ X, long long ld, int D,
                                                       const double *__restrict__ w,
                                                       const int32_t *__restrict__ n_ptr, int Ncap,
                                                       int k, int max_iter,
                                                       const long long *__restrict__ init_other,
                                                       int32_t *__restrict__ assign,
                                                       int32_t *__restrict__ new_assign,
                                                       double *__restrict__ part,      // [G][k][D]
                                                       double *__restrict__ part_w,    // [G][k]
                                                       int *__restrict__ part_n,       // [G][k]
                                                       double *__restrict__ centres,   // [k][D]
                                                       int *__restrict__ changed,      // [max_iter+2]
                                                       KmShared *__restrict__ sh,
                                                       int32_t *__restrict__ info,
                                                       uint32_t *__restrict__ status)
{
    extern __shared__ double lds_c[];          // k * D centres
    __shared__ double comb[KM_COMB];          // partials staged for the ordered combination
    __shared__ double den_s[KM_MAXK];
    const unsigned G = gridDim.x;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned epoch = 0;
    int N = *n_ptr;
    if (N > Ncap) N = Ncap;
    const int per = (N + (int)G - 1) / (int)G;
    const int lo = min(N, g * per), hi = min(N, lo + per);
    const bool f32 = sizeof(T) == 4;

    // ---- prior threshold: sort(weights)[N // 2] by stable rank counting (:144)
    const int target = N / 2;
    for (int i = lo + tid; i < hi; i += KM_THREADS) {
        const double wi = w[i];
        int rank = 0;
        for (int j = 0; j < N; ++j) {
            double wj = w[j];
            rank += (wj < wi || (wj == wi && j < i)) ? 1 : 0;
        }
        if (rank == target) sh->thr = wi;
    }
    grid_sync(&sh->barrier, G, epoch, status);
    const double thr = __hip_atomic_load(&sh->thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- initial assignment (:141-149)
    for (int i = lo + tid; i < hi; i += KM_THREADS) {
        int a = 0;
        if (!(w[i] > thr)) {
            if (k == 2) a = 1;
            else {
                int m = 0;                       // index among the points with w <= thr
                for (int j = 0; j < i; ++j) m += (w[j] <= thr) ? 1 : 0;
                a = init_other ? (int)init_other[m] : (m % (k - 1) + 1);
            }
        }
        assign[i] = a;
        new_assign[i] = a;
    }
    __syncthreads();

    int it = 0, st = 1;
    // phase == 0: unweighted means of the initial assignment (:150-151); afterwards the
    // weighted update (:163-171)
#ifdef SPA_KM_TIMING
    unsigned long long kt_[6] = {0, 0, 0, 0, 0, 0}, kt0_ = __builtin_readcyclecounter();
#define KM_T(i) { unsigned long long n_ = __builtin_readcyclecounter(); kt_[i] += n_ - kt0_; kt0_ = n_; }
#else
#define KM_T(i)
#endif
    for (int phase = 0;; ++phase) {
        // ---- partial sums of this workgroup's slice, thread t owns dimensions t, t+256, ...
        const int32_t *asg = new_assign;
        for (int d0 = tid; d0 < D; d0 += 3 * KM_THREADS) {
            // this thread's (up to) three dimensions d0, d0+KM_THREADS, d0+2*KM_THREADS; the points of the slice in
            // order, 8 rows of all three dimensions requested together so that the ordered
            // additions do not each wait for their own row
            double acc[3][KM_MAXK];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int c = 0; c < KM_MAXK; ++c) acc[j][c] = 0.0;
            for (int i0 = lo; i0 < hi; i0 += 8) {
                int aa[8];
                double ww[8], xx[3][8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u < hi ? i0 + u : hi - 1;
                    aa[u] = asg[i];
                    ww[u] = w[i];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int d = d0 + j * KM_THREADS;
                        xx[j][u] = ldx(X, (long long)i * ld + (d < D ? d : D - 1));
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (i0 + u < hi) {
                        const int a = aa[u];
                        const double wi = (a == 0) ? ww[u] : 1.0 - ww[u];
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const double v = (phase > 0) ? xx[j][u] * wi : xx[j][u];
#pragma unroll
                            for (int c = 0; c < KM_MAXK; ++c) acc[j][c] = acc[j][c] + ((a == c) ? v : 0.0);
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int d = d0 + j * KM_THREADS;
                if (d < D) {
#pragma unroll
                    for (int c = 0; c < KM_MAXK; ++c)
                        if (c < k) part[((long long)g * k + c) * D + d] = acc[j][c];
                }
            }
        }
        if (tid < k) {
            double sw = 0.0;
            int cn = 0;
            for (int i = lo; i < hi; ++i)
                if (asg[i] == tid) { sw = sw + ((phase > 0) ? ((tid == 0) ? w[i] : 1.0 - w[i]) : 1.0); ++cn; }
            part_w[g * k + tid] = sw;
            part_n[g * k + tid] = cn;
        }
        KM_T(0)
        grid_sync(&sh->barrier, G, epoch, status);
        KM_T(1)

        // ---- convergence test of the sweep that produced new_assign (:158-159)
        if (phase > 0) {
            const int ch = __hip_atomic_load(&changed[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ch == 0) { st = 0; break; }
            for (int i = lo + tid; i < hi; i += KM_THREADS) assign[i] = new_assign[i];
        }
        // ---- centres = sum over workgroups in order / denominator.  Workgroup g owns a slice of the
        // (cluster, dimension) elements.  The additions keep their order, workgroup after workgroup,
        // but the partials are fetched by ALL threads at once into LDS and then summed from there:
        // one trip to L2 per pass instead of one per addition.
        bool empty = false;
        {
            int *cnt_s = (int *)comb;                       // member counts, staged like the sums below
            for (int idx = tid; idx < k * (int)G; idx += KM_THREADS) cnt_s[idx] = part_n[idx];
            __syncthreads();
            int cn_mine = 0;             // lane c < k of every wave: members of cluster c
            if (lane < k) {
#pragma unroll 16
                for (unsigned q = 0; q < G; ++q) cn_mine += cnt_s[q * k + lane];
            }
            empty = __ballot(lane < k && cn_mine == 0) != 0ull;
            __syncthreads();
        }
        {
            const int nE = k * D;
            const int per_e = (nE + (int)G - 1) / (int)G;
            const int e0 = min(nE, g * per_e), e1 = min(nE, e0 + per_e);
            const int EC = KM_COMB / (int)G > 0 ? KM_COMB / (int)G : 1;    // elements per pass
            // denominators: sum of part_w over workgroups, in order, per cluster
            for (int idx = tid; idx < k * (int)G; idx += KM_THREADS) comb[idx] = part_w[idx];   // [q][c]
            __syncthreads();
            if (tid < k) {
                den_s[tid] = km_ordered_sum(comb + tid, (int)G, k);
            }
            __syncthreads();
            for (int eb = e0; eb < e1; eb += EC) {
                const int ne = min(EC, e1 - eb);
                for (int idx = tid; idx < ne * (int)G; idx += KM_THREADS) {
                    const int el = idx / (int)G, q = idx - el * (int)G;
                    const int e = eb + el, c = e / D, d = e - c * D;
                    comb[idx] = part[((long long)q * k + c) * D + d];
                }
                __syncthreads();
                for (int t = tid; t < ne; t += KM_THREADS) {
                    const int e = eb + t, c = e / D;
                    const double sm = km_ordered_sum(comb + t * (int)G, (int)G, 1);
                    double v = sm / den_s[c];            // 0/0 -> NaN for an empty cluster, like numpy
                    if (f32) v = (double)(float)v;      // the reference stores centres in X's dtype
                    centres[e] = v;
                }
                __syncthreads();
            }
        }
        KM_T(2)
        grid_sync(&sh->barrier, G, epoch, status);
        KM_T(3)
        if (phase > 0 && empty) { st = 2; break; }      // (:173-181) after the update
        if (it >= max_iter) { st = 1; break; }
        for (int e = tid; e < k * D; e += KM_THREADS) lds_c[e] = centres[e];
        __syncthreads();

        // ---- assignment sweep (:155-157): one wavefront per point, lanes over dimensions
        ++it;
        int local_changed = 0;
        for (int i = lo + wv; i < hi; i += KM_THREADS / 64) {
            double dist[KM_MAXK];
#pragma unroll
            for (int c = 0; c < KM_MAXK; ++c) dist[c] = 0.0;
            for (int d = lane; d < D; d += 64) {
                const double x = ldx(X, (long long)i * ld + d);
#pragma unroll
                for (int c = 0; c < KM_MAXK; ++c) {
                    if (c < k) {
                        double t = f32 ? (double)((float)x - (float)lds_c[c * D + d]) : x - lds_c[c * D + d];
                        dist[c] = dist[c] + t * t;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < KM_MAXK; ++c)
                if (c < k)
                    for (int o = 32; o > 0; o >>= 1) dist[c] = dist[c] + __shfl_xor(dist[c], o);
            int best = 0;
            double bd = sqrt(dist[0]);
#pragma unroll
            for (int c = 1; c < KM_MAXK; ++c) {
                if (c < k) {
                    double dc = sqrt(dist[c]);
                    // np.argmin: first minimum; a NaN beats everything and the first NaN stays
                    if (!(bd != bd) && ((dc != dc) || dc < bd)) { best = c; bd = dc; }
                }
            }
            if (lane == 0) {
                new_assign[i] = best;
                if (best != assign[i]) local_changed = 1;
            }
        }
        if (lane == 0 && local_changed)
            __hip_atomic_fetch_add(&changed[it], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        KM_T(4)
    }
#ifdef SPA_KM_TIMING
    if (g == 0 && tid == 0) printf("kmeans cycles (workgroup 0): partial sums %llu | barrier %llu | centres %llu | barrier %llu | sweep %llu ; iterations %d, G %u\n", kt_[0], kt_[1], kt_[2], kt_[3], kt_[4], it, G);
#endif
    if (g == 0 && tid == 0) { info[0] = it; info[1] = st; info[2] = N; info[3] = 0; }
}

extern "C" int spa_kmeans_weighted(spa_ctx *ctx, const void *X, int32_t x_dtype, int64_t ld,
                                   int32_t D, const double *w, const int32_t *n_ptr, int32_t Ncap,
                                   int32_t k, int32_t max_iter, const int64_t *init_other,
                                   int32_t *assign, int32_t *info, void *stream)
{
    SPA_ARG(ctx && X && w && n_ptr && assign && info);
    SPA_ARG(k >= 2 && k <= KM_MAXK && D > 0 && Ncap > 0 && max_iter >= 0 && ld >= D);
    SPA_ARG(x_dtype == 0 || x_dtype == 1);
    const size_t lds = (size_t)k * D * sizeof(double);
    if (lds > 120 * 1024) {
        spa_set_error("k*D = %d*%d centres do not fit LDS", k, D);
        return SPA_ERR_ARG;
    }
    hipStream_t s = spa_stream(stream);
    int kdiv = 64;               // points per workgroup (measured: 64 best at N ~ 5 000; two grid barriers per sweep dominate)
    if (const char *e = getenv("SPA_KM_DIV")) kdiv = atoi(e) > 0 ? atoi(e) : 64;                 // experiments
    int G = (Ncap + kdiv - 1) / kdiv;
    if (G > ctx->n_cu) G = ctx->n_cu;
    if (G < 1) G = 1;
    double *part;
    char *misc;
    int rc;
    const size_t part_bytes = (size_t)G * k * D * 8;
    if ((rc = spa_ws_reserve(ctx, WS_KM_PART, part_bytes, (void **)&part)) != SPA_OK) return rc;
    // misc: part_w [G*k] f64 | centres [k*D] f64 | KmShared | part_n [G*k] i32 |
    //       changed [max_iter+2] i32 | new_assign [Ncap] i32
    size_t o_pw = 0, o_cen = o_pw + (size_t)G * k * 8, o_sh = o_cen + (size_t)k * D * 8;
    size_t o_pn = o_sh + 64, o_ch = o_pn + (size_t)G * k * 4;
    size_t o_na = o_ch + (size_t)(max_iter + 2) * 4;
    o_na = (o_na + 15) & ~(size_t)15;
    size_t total = o_na + (size_t)Ncap * 4;
    if ((rc = spa_ws_reserve(ctx, WS_KM_MISC, total, (void **)&misc)) != SPA_OK) return rc;
    SPA_HIP(hipMemsetAsync(misc + o_sh, 0, o_na - o_sh, s));   // barrier, thr, part_n, changed
    SpaProfScope prof_(ctx, PROF_KMEANS, s);
    static bool attr_done[2] = {false, false};
    if (x_dtype == 1) {
        if (!attr_done[1]) {
            SPA_HIP(hipFuncSetAttribute((const void *)k_kmeans<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
            attr_done[1] = true;
        }
        hipLaunchKernelGGL(k_kmeans<double>, dim3(G), dim3(KM_THREADS), lds, s, (const double *)X,
                           (long long)ld, D, w, n_ptr, Ncap, k, max_iter, (const long long *)init_other,
                           assign, (int32_t *)(misc + o_na), part, (double *)(misc + o_pw),
                           (int *)(misc + o_pn), (double *)(misc + o_cen), (int *)(misc + o_ch),
                           (KmShared *)(misc + o_sh), info, ctx->d_status);
    } else {
        if (!attr_done[0]) {
            SPA_HIP(hipFuncSetAttribute((const void *)k_kmeans<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
            attr_done[0] = true;
        }
        hipLaunchKernelGGL(k_kmeans<float>, dim3(G), dim3(KM_THREADS), lds, s, (const float *)X,
                           (long long)ld, D, w, n_ptr, Ncap, k, max_iter, (const long long *)init_other,
                           assign, (int32_t *)(misc + o_na), part, (double *)(misc + o_pw),
                           (int *)(misc + o_pn), (double *)(misc + o_cen), (int *)(misc + o_ch),
                           (KmShared *)(misc + o_sh), info, ctx->d_status);
    }
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
