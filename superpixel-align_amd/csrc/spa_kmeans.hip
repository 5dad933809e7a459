// Weighted k-means of the reference (batch_spalign_kmeans.py:136-183) as ONE persistent launch whose
// arithmetic is numpy's, operation for operation.
//
// The reference issues ~10 CuPy/NumPy kernels and k host synchronisations per Lloyd iteration; the
// problem itself is tiny (N = a few thousand superpixels of a batch, D = 514, k <= 8), so it is latency
// bound.  Here one grid (one 1024-thread workgroup per CU at most, all co-resident) runs every phase —
// median threshold of the prior (:144), initial assignment (:141-149), unweighted initial centres
// (:150-151), assignment (:155-157), convergence test (:158-159), weighted centre update (:163-171),
// empty-cluster exit (:173-181) — separated by an agent-scope grid barrier, with the convergence flag
// and the iteration count kept on the device.
//
// Rounding is part of the contract (tests/golden/kmeans_tie.npz holds inputs that sit on the
// reference's decision boundary to within the rounding noise of its sums), so every sum keeps numpy's
// order:
//   * distances, linalg.norm(X[:,None,:] - centers[None], axis=2): difference, square and sqrt in X's
//     dtype; the D squares are added by numpy's PAIRWISE add.reduce: blocks of <= 128 elements, eight
//     strided accumulators r[j] += a[i+j] per block combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)),
//     the n % 8 tail added one by one, blocks combined by the recursive halving n2 = (n/2) - (n/2)%8.
//     One wave per point: lane (q, j) carries accumulator j of block q (16 dependent adds), the eight
//     accumulators meet in a xor-butterfly (1, 2, 4: exactly the bracket above, addition being
//     commutative), and the block sums are combined by a small stack program built on the host.
//   * centres, X[assign == c].mean(axis=0) and (X * w[:,None]).sum(0) / w.sum(): axis-0 reductions are
//     SEQUENTIAL row additions in numpy, so each (cluster, column) is one serial chain over the
//     cluster's members in index order.  A chain workgroup owns one cluster and 64 columns: all 16
//     waves gather the next member rows (products x*w already rounded) into LDS, two chunks of loads
//     in flight, while wave 0 adds the previous chunk row by row.  w.sum() is numpy's pairwise sum of
//     the member weights (a separate workgroup per cluster).  float32 descriptors (--without_pos):
//     initial means and distances in float32, weighted sums in float64, centres rounded to float32.
//   * argmin takes the first minimum and lets a NaN distance win (numpy semantics, which is what makes
//     an initially empty cluster swallow every point).
// The median threshold is an 8-pass radix select (no O(N^2) ranking); the k > 2 initial assignment
// uses an ordered prefix count of the points at or below the threshold.
#include "spa_common.h"
#include <stdlib.h>

#define KM_MAXK 8
#define KM_THREADS 512        // 8 waves per workgroup (256 registers per lane: nothing spills to scratch,
                              // whose reloads would drain the global loads in flight through vmcnt)
#define KM_WAVES (KM_THREADS / 64)
#define KM_GROWS 8            // rows per gather wave and chunk
#define KM_ROWS (KM_GROWS * (KM_WAVES - 1))   // member rows per chain chunk (waves 1..7 gather; wave 0 adds)
#define KM_MAXLEAF 48         // blocks of the pairwise tree over D (>= 57 elements each: D <= 2736 always fits)
#define KM_CHUNK_BYTES (128 * 64 * 8)

struct KmShared {
    unsigned barrier;      // monotonic arrival counter
    int pad0, pad1, pad2;
    double thr;
};

// numpy's pairwise summation tree over D elements, built on the host
struct KmTree {
    int nleaf, nops;
    short leaf_start[KM_MAXLEAF], leaf_n[KM_MAXLEAF];
    short ops[2 * KM_MAXLEAF];           // nops pairs (lane a, lane b): lane a += lane b, in numpy's order
};

static inline int km_leaf_lane(int q) { return (q & 7) * 8 + (q >> 3); }

// returns the index of the first block of the subtree (its lane receives the subtree's sum), -1 on overflow
static int km_build_tree(KmTree &t, int start, int n)
{
    if (n <= 128) {
        if (t.nleaf >= KM_MAXLEAF) return -1;
        t.leaf_start[t.nleaf] = (short)start; t.leaf_n[t.nleaf] = (short)n;
        return t.nleaf++;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    const int a = km_build_tree(t, start, n2);
    if (a < 0) return -1;
    const int b = km_build_tree(t, start + n2, n - n2);
    if (b < 0) return -1;
    t.ops[2 * t.nops] = (short)km_leaf_lane(a);
    t.ops[2 * t.nops + 1] = (short)km_leaf_lane(b);
    t.nops += 1;
    return a;
}

// Grid-wide barrier of the persistent k-means launch.  Residency: the launch is a plain one — ROCm's
// cooperative launch checks the grid against the occupancy of an EMPTY device and serialises behind its own queue; it does not
// gang-schedule against kernels of other streams or processes either — so the host caps the grid at what the occupancy query
// says fits the device at once (spa_kmeans_weighted) and a workgroup that arrives early simply waits for the others to be
// placed: beside the DRN's persistent kernels of the next batch (run(join = False)) that is until one of them retires a
// workgroup, i.e. at most one such kernel's duration (<= 7 ms); none of those kernels waits for this one, so the wait ends.
// Two spin-barrier launches of DIFFERENT processes on one device could each hold compute units the other needs: the wait is
// bounded in WALL-CLOCK time (s_memrealtime, 100 MHz: 20 s without the counter reaching its target), after which the launch
// gives up, latches SPA_ST_KMEANS_BARRIER — every caller checks the status word — and runs to its end without hanging the device.
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
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 1023) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > 2000000000ull) {      // 20 s at 100 MHz
                atomicOr(status, SPA_ST_KMEANS_BARRIER);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// workgroup barrier that orders LDS only: global loads issued before it stay in flight across it
// (__syncthreads() waits vmcnt(0) here, which would put every gather's memory latency on the chain's path)
__device__ __forceinline__ void km_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// LDS written and read back by the same wave
__device__ __forceinline__ void km_wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// value of a wave-uniform lane (v_readlane: no LDS round trip)
__device__ __forceinline__ int km_readlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float km_readlane(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ double km_readlane(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// order-preserving key of a double (ascending)
__device__ __forceinline__ unsigned long long km_key(double v)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__device__ __forceinline__ double km_unkey(unsigned long long u)
{
    unsigned long long b = (u >> 63) ? (u & 0x7fffffffffffffffULL) : ~u;
    return __longlong_as_double((long long)b);
}

// ordered exclusive rank of `flag` over the workgroup's threads; *total = number of flags.
// scr: KM_WAVES + 1 ints of LDS.  Two workgroup barriers.
__device__ __forceinline__ int km_wg_rank(bool flag, int *scr, int &total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    __syncthreads();                          // scr free to overwrite
    if (lane == 0) scr[wv] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < KM_WAVES; ++i) {
        const int c = scr[i];
        if (i < wv) off += c;
        tot += c;
    }
    total = tot;
    return off + (int)spa_rank_in_mask(m);
}

// value of lane (lane ^ X) inside every octet, X = 1, 2, 4, by DPP moves (a __shfl_xor is a ds_bpermute per
// 32 bits: an LDS round trip on the per-point path of the sweep)
template <int X>
__device__ __forceinline__ int km_xor_i32(int v)
{
    if (X == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);           // quad_perm [1,0,3,2]
    if (X == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);           // quad_perm [2,3,0,1]
    const int up = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0xF, false);              // row_shl:4: lane i <- i + 4
    const int dn = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);              // row_shr:4: lane i <- i - 4
    return (threadIdx.x & 4) ? dn : up;
}
template <int X>
__device__ __forceinline__ double km_xor(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)km_xor_i32<X>((int)(unsigned)u), hi = (unsigned)km_xor_i32<X>((int)(unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int X>
__device__ __forceinline__ float km_xor(float v) { return __int_as_float(km_xor_i32<X>(__float_as_int(v))); }

// numpy's pairwise sum of wl[0..n) (float64) by ONE wave: lanes 0..7 (every octet redundantly) are the eight
// accumulators of the current block; the recursion is emulated with an explicit wave-uniform stack that lives in
// four VGPRs — lane t holds pending call t (v_readlane / lane select: no LDS round trip, no synchronisation per
// step).  A block's up to 16 values per accumulator are all fetched before the first of the dependent additions.
__device__ __forceinline__ double km_rl(double v, int l)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// (v_writelane has no builtin here: a compare + select does the same for a wave-uniform slot)
#define KM_WRLANE(vec_, slot_, val_) vec_ = (lane == (slot_)) ? (val_) : vec_

__device__ double km_pairwise_wave(const double *wl, int n, int *, double *)
{
    const int lane = threadIdx.x & 63, j = lane & 7;
    int st_s = 0, st_n = 0, st_st = 0;      // per lane t: start, length, stage of pending call t
    double st_v = 0.0;                      // ... and its left sum
    KM_WRLANE(st_n, 0, n);
    int sp = 1;
    double ret = 0.0;
    while (sp > 0) {
        const int t = __builtin_amdgcn_readfirstlane(sp - 1);
        const int stage = __builtin_amdgcn_readlane(st_st, t);
        const int s0 = __builtin_amdgcn_readlane(st_s, t), nn = __builtin_amdgcn_readlane(st_n, t);
        if (stage == 0) {
            if (nn <= 128) {
                const int n8 = nn - (nn % 8);
                double v[16], tl[7];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = (8 * i < n8) ? wl[s0 + 8 * i + j] : 0.0;
#pragma unroll
                for (int i = 0; i < 7; ++i) tl[i] = (n8 + i < nn) ? wl[s0 + n8 + i] : 0.0;
                double res = 0.0;
                if (nn >= 8) {
                    double r = v[0];
#pragma unroll
                    for (int i = 1; i < 16; ++i)
                        if (8 * i < n8) r = r + v[i];
                    r = r + km_xor<1>(r);
                    r = r + km_xor<2>(r);
                    r = r + km_xor<4>(r);
                    res = r;
                }
#pragma unroll
                for (int i = 0; i < 7; ++i)
                    if (n8 + i < nn) res = res + tl[i];
                ret = res;
                sp -= 1;
            } else {
                int n2 = nn / 2;
                n2 -= n2 % 8;
                const int u = __builtin_amdgcn_readfirstlane(sp);
                KM_WRLANE(st_st, t, 1);
                KM_WRLANE(st_s, u, s0);
                KM_WRLANE(st_n, u, n2);
                KM_WRLANE(st_st, u, 0);
                sp += 1;
            }
        } else if (stage == 1) {
            int n2 = nn / 2;
            n2 -= n2 % 8;
            const int u = __builtin_amdgcn_readfirstlane(sp);
            KM_WRLANE(st_v, t, ret);
            KM_WRLANE(st_st, t, 2);
            KM_WRLANE(st_s, u, s0 + n2);
            KM_WRLANE(st_n, u, nn - n2);
            KM_WRLANE(st_st, u, 0);
            sp += 1;
        } else {
            ret = km_rl(st_v, t) + ret;
            sp -= 1;
        }
    }
    return ret;
}

// Ordered list of the members of cluster c: indices (and their update weights) in index order.
// All KM_THREADS threads; 8 consecutive points per thread and tile.  Returns the member count.
__device__ int km_member_list(const int32_t *__restrict__ asg, const double *__restrict__ w, int N, int c,
                              bool weighted, int32_t *__restrict__ ml, double *__restrict__ wl, int *scr)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int base = 0;
    for (int t0 = 0; t0 < N; t0 += KM_THREADS * 8) {
        const int i0 = t0 + tid * 8;
        int a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = (i0 + u < N) ? asg[i0 + u] : -1;
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) cnt += (a[u] == c) ? 1 : 0;
        // inclusive scan of cnt over the wave, then over the waves
        int inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o);
            if (lane >= o) inc += v;
        }
        __syncthreads();
        if (lane == 63) scr[wv] = inc;
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < KM_WAVES; ++i) {
            const int v = scr[i];
            if (i < wv) off += v;
            tot += v;
        }
        int pos = base + off + inc - cnt;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (a[u] == c) {
                const int i = i0 + u;
                ml[pos] = i;
                if (wl) wl[pos] = weighted ? ((c == 0) ? w[i] : 1.0 - w[i]) : 1.0;
                pos += 1;
            }
        }
        base += tot;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                         // the list is read by other waves of this workgroup
    return base;
}

// info: {iterations, status, N, -}
template <typename T, int KM>
__global__ __launch_bounds__(KM_THREADS) void k_kmeans(const T *__restrict__ X, long long ld, int D,
                                                       const double *__restrict__ w,
                                                       const int32_t *__restrict__ n_ptr, int Ncap,
                                                       int k, int max_iter,
                                                       const long long *__restrict__ init_other, const int32_t *__restrict__ gate,
                                                       int32_t *__restrict__ assign,
                                                       int32_t *__restrict__ new_assign,
                                                       double *__restrict__ sums,      // [k][D]
                                                       double *__restrict__ wsum,      // [k]
                                                       int *__restrict__ cnt_c,        // [k]
                                                       int *__restrict__ part_n,       // [G]
                                                       int32_t *__restrict__ mlist,    // [regions][Ncap]
                                                       double *__restrict__ wlist,     // [regions][Ncap]
                                                       int *__restrict__ changed,      // [max_iter+2]
                                                       KmShared *__restrict__ sh,
                                                       int32_t *__restrict__ info,
                                                       uint32_t *__restrict__ status,
                                                       const KmTree tree)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    // a gated launch (a speculative retry run, spa_kmeans_weighted_gated) that is not wanted: every workgroup leaves before the
    // first grid barrier, outputs untouched
    if (gate && *gate <= 0) return;
    // [0, k*D*8): centres (T) | then KM_CHUNK_BYTES shared by the chain buffer and the sweep scratch
    T *lds_c = (T *)lds_raw;
    unsigned char *scratch = lds_raw + (((size_t)k * D * sizeof(double) + 15) & ~(size_t)15);
    double *chunk = (double *)scratch;                           // [2][KM_ROWS][64]: two chain chunks
    __shared__ int scr[KM_WAVES + 4];
    __shared__ short tl_start[KM_MAXLEAF], tl_n[KM_MAXLEAF], tl_ops[2 * KM_MAXLEAF];
    __shared__ unsigned hist[256];
    __shared__ unsigned long long sel_prefix;
    __shared__ int sel_rank;
    __shared__ int pw_fs[120];
    __shared__ double pw_fv[40];

    const unsigned G = gridDim.x;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned epoch = 0;
    int N = *n_ptr;
    if (N > Ncap) N = Ncap;
    const int per = (N + (int)G - 1) / (int)G;
    const int lo = min(N, g * per), hi = min(N, lo + per);
    const bool f32 = sizeof(T) == 4;
    const int nblk = (D + 63) / 64;
    const int n_chain = k * nblk, n_task = n_chain + k;
    // member-list region of this workgroup (only workgroups g < n_task ever run an update task)
    int32_t *ml = mlist + (long long)min(g, n_task - 1) * Ncap;
    double *wl = wlist + (long long)min(g, n_task - 1) * Ncap;
    const int nleaf = tree.nleaf, nops = tree.nops;
    for (int i = tid; i < KM_MAXLEAF; i += KM_THREADS) { tl_start[i] = tree.leaf_start[i]; tl_n[i] = tree.leaf_n[i]; }
    for (int i = tid; i < 2 * KM_MAXLEAF; i += KM_THREADS) tl_ops[i] = tree.ops[i];

    // ---- prior threshold sort(weights)[N // 2] (:144): radix select, most significant byte first
    if (g == 0) {
        if (tid == 0) { sel_prefix = 0ull; sel_rank = N / 2; }
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            for (int i = tid; i < 256; i += KM_THREADS) hist[i] = 0u;
            __syncthreads();
            const unsigned long long pre = sel_prefix;
            for (int i = tid; i < N; i += KM_THREADS) {
                const unsigned long long key = km_key(w[i]);
                const bool match = pass == 0 || (key >> (shift + 8)) == (pre >> (shift + 8));
                if (match) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                int r = sel_rank;
                unsigned b = 0;
                for (; b < 256u; ++b) {
                    const int c = (int)hist[b];
                    if (r < c) break;
                    r -= c;
                }
                sel_rank = r;
                sel_prefix = pre | ((unsigned long long)b << shift);
            }
            __syncthreads();
        }
        if (tid == 0) sh->thr = km_unkey(sel_prefix);
    }
    grid_sync(&sh->barrier, G, epoch, status);
    const double thr = __hip_atomic_load(&sh->thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- initial assignment (:141-149).  k > 2: the points at or below the threshold take the entries
    // of the (shuffled) index vector in index order -> ordered prefix count over the whole array
    if (k == 2) {
        for (int i = lo + tid; i < hi; i += KM_THREADS) {
            const int a = (w[i] > thr) ? 0 : 1;
            assign[i] = a;
            new_assign[i] = a;
        }
    } else {
        int mine = 0;
        for (int i = lo + tid; i < hi; i += KM_THREADS) mine += (w[i] > thr) ? 0 : 1;
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
        __syncthreads();
        if (lane == 0) scr[wv] = mine;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int i = 0; i < KM_WAVES; ++i) t += scr[i];
            part_n[g] = t;
        }
        grid_sync(&sh->barrier, G, epoch, status);
        int base = 0;
        for (int q = lane; q < g; q += 64) base += part_n[q];
        for (int o = 32; o > 0; o >>= 1) base += __shfl_xor(base, o);
        for (int i0 = lo; i0 < hi; i0 += KM_THREADS) {
            const int i = i0 + tid;
            const bool other = i < hi && !(w[i] > thr);
            int tot;
            const int m = base + km_wg_rank(other, scr, tot);
            if (i < hi) {
                const int a = other ? (init_other ? (int)init_other[m] : (m % (k - 1) + 1)) : 0;
                assign[i] = a;
                new_assign[i] = a;
            }
            base += tot;
        }
    }
    // the first update reads new_assign of every slice
    grid_sync(&sh->barrier, G, epoch, status);

    int it = 0, st = 1;
#ifdef SPA_KM_TIMING
    unsigned long long kt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, kt0_ = __builtin_readcyclecounter();
#define KM_T(i) { unsigned long long n_ = __builtin_readcyclecounter(); kt_[i] += n_ - kt0_; kt0_ = n_; }
#else
#define KM_T(i)
#endif
    for (int phase = 0;; ++phase) {
        // ================= centre sums of `new_assign`: phase 0 unweighted means in T (:150-151),
        // afterwards weighted float64 sums (:163-171)
        for (int u = g; u < n_task; u += (int)G) {
            if (u >= n_chain) {
                // ---- member count and w.sum() of one cluster
                const int c = u - n_chain;
                const int cnt = km_member_list(new_assign, w, N, c, phase > 0, ml, wl, scr);
                // the pairwise pass is one wave walking the weights block by block: give it LDS latency
                const bool in_lds = cnt <= KM_CHUNK_BYTES / 8;
                if (phase > 0 && in_lds) {
                    for (int i = tid; i < cnt; i += KM_THREADS) chunk[i] = wl[i];
                    __syncthreads();
                }
                if (wv == 0) {
                    const double ws = (phase > 0) ? km_pairwise_wave(in_lds ? chunk : wl, cnt, pw_fs, pw_fv) : 0.0;
                    if (lane == 0) { wsum[c] = ws; cnt_c[c] = cnt; }
                }
                __syncthreads();
                continue;
            }
            // ---- one cluster x 64 columns: sequential sums over the members in index order
            const int c = u / nblk, blk = u - c * nblk;
            const int col = blk * 64 + lane;
            const bool colok = col < D;
            KM_T(5)
            const int cnt = km_member_list(new_assign, w, N, c, phase > 0, ml, wl, scr);
            KM_T(6)
            const int nchunk = (cnt + KM_ROWS - 1) / KM_ROWS;
            double accd = 0.0;
            float accf = 0.0f;
            // Two register sets (raw values + weights of 8 rows each) = two chunks of loads in flight.
            // The list entries (index, weight) of a chunk are fetched one gather ahead; the row loads are
            // unconditional (clamped index): a branch around a load makes the compiler wait for each load
            // separately instead of keeping the eight in flight; the products are formed only when the set
            // is written to LDS, so a gather never waits for its own loads.
            double ra[KM_GROWS], rb[KM_GROWS], wa[KM_GROWS], wb[KM_GROWS];
            int l_idx = 0;
            double l_w = 0.0;
            const int gw = wv - 1;          // gather wave index; wave 0 only runs the chain
            auto fetch_list = [&](int q) {
                if (wv == 0) return;
                const int r = q * KM_ROWS + gw * KM_GROWS + (lane & (KM_GROWS - 1));
                const int rc = r < cnt ? r : (cnt > 0 ? cnt - 1 : 0);
                l_idx = cnt > 0 ? ml[rc] : 0;
                l_w = cnt > 0 ? wl[rc] : 0.0;
            };
            const int colc = colok ? col : D - 1;
            auto gather = [&](int q, double (&r)[KM_GROWS], double (&rw)[KM_GROWS]) {
                if (wv == 0) return;
#pragma unroll
                for (int e = 0; e < KM_GROWS; ++e) {
                    const int ie = km_readlane(l_idx, e);
                    rw[e] = km_readlane(l_w, e);
                    r[e] = (double)X[(long long)ie * ld + colc];
                }
                fetch_list(q + 1);          // entries of the next gather (chunks are gathered in order)
            };
            auto store = [&](int q, const double (&r)[KM_GROWS], const double (&rw)[KM_GROWS]) {
                if (wv == 0) return;
                const int r0 = q * KM_ROWS + gw * KM_GROWS;
#pragma unroll
                for (int e = 0; e < KM_GROWS; ++e) {
                    const double v = (phase > 0) ? r[e] * rw[e] : r[e];
                    chunk[((q & 1) * KM_ROWS + gw * KM_GROWS + e) * 64 + lane] = (colok && r0 + e < cnt) ? v : 0.0;
                }
            };
            auto chain = [&](int q) {
                if (wv != 0) return;
#ifdef SPA_KM_TIMING
                const unsigned long long c0_ = __builtin_readcyclecounter();
#endif
                // all KM_ROWS rows, unconditionally: the rows past the member count hold +0.0 (store), and x + 0.0 = x
                // (the running sum is never -0.0: it starts at +0.0).  Straight-line code: the LDS reads of the
                // later rows are in flight while the dependent additions of the earlier rows issue.
                const double *src = chunk + (q & 1) * KM_ROWS * 64 + lane;
                double v[KM_ROWS];
#pragma unroll
                for (int e = 0; e < KM_ROWS; ++e) v[e] = src[e * 64];
                if (f32 && phase == 0) {
#pragma unroll
                    for (int e = 0; e < KM_ROWS; ++e) accf = accf + (float)v[e];
                } else {
#pragma unroll
                    for (int e = 0; e < KM_ROWS; ++e) accd = accd + v[e];
                }
#ifdef SPA_KM_TIMING
                kt_[7] += __builtin_readcyclecounter() - c0_;
#endif
            };
            // Two LDS chunk buffers: while wave 0 adds chunk q (56 dependent additions per column), the gather
            // waves form the products of chunk q + 1 and write them to the other buffer, then issue the loads of
            // chunk q + 3 into the register set they just emptied: one barrier per chunk, and the chain never
            // waits for a store (it did, with one buffer: chain, barrier, store, barrier).  A third register set
            // (three chunks of loads in flight) was measured: no gain, and it spills.
            if (nchunk > 0) {
                fetch_list(0);
                gather(0, ra, wa);
                if (nchunk > 1) gather(1, rb, wb);
                store(0, ra, wa);
                if (nchunk > 2) gather(2, ra, wa);
                km_lds_barrier();
                for (int q = 0; q < nchunk; q += 2) {
                    // buffer 0 = chunk q; set b = chunk q + 1, set a = chunk q + 2
                    if (q + 1 < nchunk) store(q + 1, rb, wb);
                    if (q + 3 < nchunk) gather(q + 3, rb, wb);
                    chain(q);
                    km_lds_barrier();
                    if (q + 1 >= nchunk) break;
                    // buffer 1 = chunk q + 1; set a = chunk q + 2, set b = chunk q + 3
                    if (q + 2 < nchunk) store(q + 2, ra, wa);
                    if (q + 4 < nchunk) gather(q + 4, ra, wa);
                    chain(q + 1);
                    km_lds_barrier();
                }
            }
            if (wv == 0 && colok) sums[(long long)c * D + col] = (f32 && phase == 0) ? (double)accf : accd;
            __syncthreads();
        }
        KM_T(0)
        grid_sync(&sh->barrier, G, epoch, status);
        KM_T(1)

        // ================= centres = sums / denominator, into LDS (every workgroup)
        bool empty = false;
        for (int c = 0; c < k; ++c) empty = empty || (cnt_c[c] == 0);
        for (int e = tid; e < k * D; e += KM_THREADS) {
            const int c = e / D;
            T v;
            if (phase == 0) {
                // X[assign == c].mean(axis=0): sum in T, divided by the count in T; 0/0 = NaN
                v = (T)sums[e] / (T)cnt_c[c];
            } else {
                v = (T)(sums[e] / wsum[c]);       // 0/0 -> NaN for an empty cluster, like numpy
            }
            lds_c[e] = v;
        }
        __syncthreads();
        if (phase > 0 && empty) { st = 2; break; }      // (:173-181) after the update
        if (it >= max_iter) { st = 1; break; }

        KM_T(2)
        // ================= assignment sweep (:155-157): one wave per point
        ++it;
        int local_changed = 0;
        const int q8 = lane >> 3, j = lane & 7;
        // (clusters c >= k, when k is below the compile-time bound KM, are computed on whatever LDS holds
        // behind the centres — always inside the allocation — and never looked at: no branches in the loop)
        // block q of the pairwise tree lives in octet q & 7, round q >> 3; its sum ends in lane
        // (q & 7) * 8 + (q >> 3); the host-built program then adds lanes pairwise (readlane: uniform lanes)
        const int rounds = (nleaf + 7) >> 3;
        for (int i = lo + wv; i < hi; i += KM_WAVES) {
            const T *xr = X + (long long)i * ld;
            T val[KM];                      // per lane: the block sum this lane keeps, per cluster
#pragma unroll
            for (int c = 0; c < KM; ++c) val[c] = (T)0;
            for (int r0 = 0; r0 < rounds; r0 += 2) {
                // two rounds = 16 blocks per pass: all their loads are issued before any arithmetic
                int s0[2], n[2], steps[2];
                T xv[2][16];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int q = (r0 + u) * 8 + q8;
                    const bool act = q < nleaf;
                    s0[u] = act ? (int)tl_start[q] : 0;
                    n[u] = act ? (int)tl_n[q] : 0;
                    steps[u] = n[u] >> 3;
#pragma unroll
                    for (int t = 0; t < 16; ++t) xv[u][t] = xr[(t < steps[u]) ? s0[u] + 8 * t + j : 0];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    T acc[KM];
#pragma unroll
                    for (int c = 0; c < KM; ++c) acc[c] = (T)0;
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int d = (t < steps[u]) ? s0[u] + 8 * t + j : 0;
#pragma unroll
                        for (int c = 0; c < KM; ++c) {
                            const T df = xv[u][t] - lds_c[c * D + d];
                            const T sq = df * df;
                            const T nv = (t == 0) ? sq : acc[c] + sq;
                            acc[c] = (t < steps[u]) ? nv : acc[c];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < KM; ++c) {
                        T r = acc[c];
                        r = r + km_xor<1>(r);
                        r = r + km_xor<2>(r);
                        r = r + km_xor<4>(r);
                        acc[c] = (steps[u] > 0) ? r : (T)0;
                    }
                    // the n % 8 tail (and a whole block shorter than 8), one by one
                    for (int e = steps[u] * 8; e < n[u]; ++e) {
                        const T x = xr[s0[u] + e];
#pragma unroll
                        for (int c = 0; c < KM; ++c) {
                            const T df = x - lds_c[c * D + s0[u] + e];
                            acc[c] = acc[c] + df * df;
                        }
                    }
                    const bool keep = j == r0 + u;          // lane (q & 7) * 8 + (q >> 3)
#pragma unroll
                    for (int c = 0; c < KM; ++c) val[c] = keep ? acc[c] : val[c];
                }
            }
            // block sums -> total: numpy's recursion as a program of lane additions (left operand's lane
            // receives the sum); the total ends in the lane of block 0 = lane 0
            for (int o = 0; o < nops; ++o) {
                const int la = (int)tl_ops[2 * o], lb = (int)tl_ops[2 * o + 1];
#pragma unroll
                for (int c = 0; c < KM; ++c) {
                    const T sum = km_readlane(val[c], la) + km_readlane(val[c], lb);
                    val[c] = (lane == la) ? sum : val[c];
                }
            }
            int best = 0;
            T bd = km_readlane(val[0], 0);
            bd = f32 ? (T)sqrtf((float)bd) : (T)sqrt((double)bd);
#pragma unroll
            for (int c = 1; c < KM; ++c) {
                if (c < k) {
                    T dc = km_readlane(val[c], 0);
                    dc = f32 ? (T)sqrtf((float)dc) : (T)sqrt((double)dc);
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
        KM_T(3)
        grid_sync(&sh->barrier, G, epoch, status);
        KM_T(4)

        // ================= convergence test of the sweep (:158-159)
        const int ch = __hip_atomic_load(&changed[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ch == 0) { st = 0; break; }
        for (int i = lo + tid; i < hi; i += KM_THREADS) assign[i] = new_assign[i];
    }
#ifdef SPA_KM_TIMING
    if ((g == 0 || g == n_chain) && tid == 0)
        printf("kmeans cycles (workgroup %d): update %llu (of which member list %llu) | barrier %llu | centres %llu | sweep %llu | "
               "barrier %llu ; chain adds alone %llu ; iterations %d, G %u\n", g, kt_[0] + kt_[6], kt_[6], kt_[1], kt_[2], kt_[3], kt_[4], kt_[7], it, G);
#endif
    if (g == 0 && tid == 0) { info[0] = it; info[1] = st; info[2] = N; info[3] = 0; }
}

extern "C" int spa_kmeans_weighted_gated(spa_ctx *ctx, const void *X, int32_t x_dtype, int64_t ld,
                                         int32_t D, const double *w, const int32_t *n_ptr, int32_t Ncap,
                                         int32_t k, int32_t max_iter, const int64_t *init_other, const int32_t *gate,
                                         int32_t *assign, int32_t *info, void *stream);

extern "C" int spa_kmeans_weighted(spa_ctx *ctx, const void *X, int32_t x_dtype, int64_t ld,
                                   int32_t D, const double *w, const int32_t *n_ptr, int32_t Ncap,
                                   int32_t k, int32_t max_iter, const int64_t *init_other,
                                   int32_t *assign, int32_t *info, void *stream)
{
    return spa_kmeans_weighted_gated(ctx, X, x_dtype, ld, D, w, n_ptr, Ncap, k, max_iter, init_other, nullptr, assign, info, stream);
}

extern "C" int spa_kmeans_weighted_gated(spa_ctx *ctx, const void *X, int32_t x_dtype, int64_t ld,
                                         int32_t D, const double *w, const int32_t *n_ptr, int32_t Ncap,
                                         int32_t k, int32_t max_iter, const int64_t *init_other, const int32_t *gate,
                                         int32_t *assign, int32_t *info, void *stream)
{
    SPA_ARG(ctx && X && w && n_ptr && assign && info);
    SPA_ARG(k >= 2 && k <= KM_MAXK && D > 0 && Ncap > 0 && max_iter >= 0 && ld >= D);
    SPA_ARG(x_dtype == 0 || x_dtype == 1);
    const size_t lds = (((size_t)k * D * sizeof(double) + 15) & ~(size_t)15) + KM_CHUNK_BYTES;
    KmTree tree;
    memset(&tree, 0, sizeof(tree));
    if (lds > 150 * 1024 || km_build_tree(tree, 0, D) < 0) {
        spa_set_error("k*D = %d*%d centres do not fit LDS next to the chain buffer", k, D);
        return SPA_ERR_ARG;
    }
    hipStream_t s = spa_stream(stream);
    const int nblk = (D + 63) / 64;
    const int n_task = k * nblk + k;
    int kdiv = 32;               // points per workgroup of the sweep (8 waves: four points per wave; measured best at N ~ 5 000)
    if (const char *e = getenv("SPA_KM_DIV")) kdiv = atoi(e) > 0 ? atoi(e) : kdiv;               // experiments
    int G = (Ncap + kdiv - 1) / kdiv;
    if (G < n_task) G = n_task;
    if (G > ctx->n_cu) G = ctx->n_cu;
    if (G < 1) G = 1;
    // (G <= one workgroup per compute unit: resident at once on an otherwise idle device whenever the occupancy query below
    // answers >= 1 — grid_sync's comment has the rest)
    // member lists: one region per workgroup that runs update tasks
    const int regions = G < n_task ? G : n_task;
    char *lists;
    char *misc;
    int rc;
    const size_t list_bytes = (size_t)regions * Ncap * (4 + 8);
    if ((rc = spa_ws_reserve(ctx, WS_KM_PART, list_bytes, (void **)&lists)) != SPA_OK) return rc;
    // misc: sums [k*D] f64 | wsum [k] f64 | KmShared | cnt_c [k] i32 | part_n [G] i32 |
    //       changed [max_iter+2] i32 | new_assign [Ncap] i32
    size_t o_sum = 0, o_ws = o_sum + (size_t)k * D * 8, o_sh = o_ws + (size_t)KM_MAXK * 8;
    size_t o_cn = o_sh + 64, o_pn = o_cn + (size_t)KM_MAXK * 4, o_ch = o_pn + (size_t)G * 4;
    size_t o_na = o_ch + (size_t)(max_iter + 2) * 4;
    o_na = (o_na + 15) & ~(size_t)15;
    size_t total = o_na + (size_t)Ncap * 4;
    if ((rc = spa_ws_reserve(ctx, WS_KM_MISC, total, (void **)&misc)) != SPA_OK) return rc;
    SPA_HIP(hipMemsetAsync(misc + o_sh, 0, o_na - o_sh, s));   // barrier, thr, counters, changed
    SpaProfScope prof_(ctx, PROF_KMEANS, s);
    double *wlist = (double *)lists;
    int32_t *mlist = (int32_t *)(lists + (size_t)regions * Ncap * 8);
#define KM_LAUNCH(TT, KK, SLOT)                                                                              \
    do {                                                                                                         \
        if (!(ctx->km_attr_done[SLOT / 3] & (1 << (SLOT % 3)))) {                                                \
            SPA_HIP(hipFuncSetAttribute((const void *)k_kmeans<TT, KK>,                                          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));                \
            int per_cu = 0;                                                                                      \
            SPA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_kmeans<TT, KK>, KM_THREADS, 150 * 1024)); \
            if (per_cu < 1) { spa_set_error("k_kmeans: no workgroup fits a compute unit"); return SPA_ERR_ARG; } \
            ctx->km_attr_done[SLOT / 3] |= (1 << (SLOT % 3));                                                    \
        }                                                                                                        \
        hipLaunchKernelGGL((k_kmeans<TT, KK>), dim3(G), dim3(KM_THREADS), lds, s, (const TT *)X, (long long)ld, \
                           D, w, n_ptr, Ncap, k, max_iter, (const long long *)init_other, gate, assign,         \
                           (int32_t *)(misc + o_na), (double *)(misc + o_sum), (double *)(misc + o_ws),          \
                           (int *)(misc + o_cn), (int *)(misc + o_pn), mlist, wlist, (int *)(misc + o_ch),       \
                           (KmShared *)(misc + o_sh), info, ctx->d_status, tree);                                \
    } while (0)
    // compile-time cluster bound: registers and unrolled code sized for the common k = 2 / k <= 4 cases
    if (x_dtype == 1) {
        if (k == 2) KM_LAUNCH(double, 2, 3); else if (k <= 4) KM_LAUNCH(double, 4, 4); else KM_LAUNCH(double, 8, 5);
    } else {
        if (k == 2) KM_LAUNCH(float, 2, 0); else if (k <= 4) KM_LAUNCH(float, 4, 1); else KM_LAUNCH(float, 8, 2);
    }
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
