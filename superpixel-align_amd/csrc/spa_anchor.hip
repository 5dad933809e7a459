// Anchor selection on the device: CPython's random.shuffle(inside_coords)[:n_anchors] for every superpixel
// (batch_spalign_kmeans.py:231-234), with the generator state carried from superpixel to superpixel, image to
// image and batch to batch exactly as the reference's module-level `random` carries it.
//
// Which pixels end up in front depends only on the LENGTH of each list, so the work is
//   1. k_mt_generate   MT19937 output blocks (CPython's generator) into a device ring, ahead of use: the
//                      stream does not depend on the data, so it is produced in the background (one
//                      workgroup: the 624-word regeneration is three dependent phases per block);
//   2. k_anchor_scan   Random._randbelow for every swap of every shuffle, in stream order: candidate
//                      v = output >> (32 - bit_length(i + 1)), accepted iff v <= i, i counting down from n-1
//                      to 1.  One workgroup examines 1 024 stream outputs per step; whether output t is
//                      accepted depends on the number of acceptances before it, which is found as the
//                      (unique) fixed point of  a = prefix_count(v(a) <= i0 - a)  by iteration from a
//                      proportional guess (two to four rounds); the accepted j of swap i is stored at
//                      jbuf[offset(superpixel) + i];
//   3. k_anchor_trace  which original list positions end in the first n_anchors places: position p is
//                      traced BACKWARDS through the swaps (i = 1 .. n-1: the swap (i, j_i) is met in reverse
//                      order of application), q <- i whenever j_i == q; a wave tests 64 swaps per step and
//                      carries all n_anchors traces at once.  No permutation array exists anywhere.
// The ranks (raster-order rank of the chosen pixel inside its superpixel) then go to k_select_pixels.
#include "spa_common.h"

#define RNG_RING_LOG2 27
#define RNG_RING (1ull << RNG_RING_LOG2)            // 2^27 outputs = 512 MB
#define RNG_MASK (RNG_RING - 1ull)
#define SCAN_T 1024

struct RngDev {
    uint32_t mt[624];
    unsigned long long wpos, rpos;                  // outputs generated / consumed so far
    unsigned long long pad;
};

__device__ __forceinline__ uint32_t mt_temper(uint32_t t)
{
    t ^= (t >> 11);
    t ^= (t << 7) & 0x9d2c5680u;
    t ^= (t << 15) & 0xefc60000u;
    t ^= (t >> 18);
    return t;
}

// generate whole blocks until at least `want` outputs are available (or the ring is full)
__global__ __launch_bounds__(256) void k_mt_generate(RngDev *__restrict__ st, uint32_t *__restrict__ ring,
                                                     unsigned long long want)
{
    __shared__ uint32_t s[2][624];
    const int tid = threadIdx.x;
    for (int k = tid; k < 624; k += 256) s[0][k] = st->mt[k];
    unsigned long long wpos = st->wpos;
    const unsigned long long rpos = st->rpos;
    __syncthreads();
    int cur = 0;
    while (wpos - rpos < want && wpos - rpos + 624 <= RNG_RING) {
        const uint32_t *o = s[cur];
        uint32_t *n = s[cur ^ 1];
        // new[k] = (k < 227 ? old[k + 397] : new[k - 227]) ^ twist(old[k], k < 623 ? old[k + 1] : new[0])
        if (tid < 227) {
            const int k = tid;
            const uint32_t y = (o[k] & 0x80000000u) | (o[k + 1] & 0x7fffffffu);
            n[k] = o[k + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        }
        __syncthreads();
        if (tid < 227) {
            const int k = 227 + tid;
            const uint32_t y = (o[k] & 0x80000000u) | (o[k + 1] & 0x7fffffffu);
            n[k] = n[k - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        }
        __syncthreads();
        if (tid < 170) {
            const int k = 454 + tid;
            const uint32_t nx = (k < 623) ? o[k + 1] : n[0];
            const uint32_t y = (o[k] & 0x80000000u) | (nx & 0x7fffffffu);
            n[k] = n[k - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        }
        __syncthreads();
        for (int k = tid; k < 624; k += 256) ring[(wpos + k) & RNG_MASK] = mt_temper(n[k]);
        wpos += 624;
        cur ^= 1;
    }
    __syncthreads();
    for (int k = tid; k < 624; k += 256) st->mt[k] = s[cur][k];
    if (tid == 0) st->wpos = wpos;
}

// exclusive prefix count of `flag` over the 1 024 threads; total in *total.  scr: 16 ints.
__device__ __forceinline__ int scan_prefix(bool flag, int *scr, int &total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    __syncthreads();
    if (lane == 0) scr[wv] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = scr[i];
        if (i < wv) off += c;
        tot += c;
    }
    total = tot;
    return off + (int)spa_rank_in_mask(m);
}

// workgroup barrier that orders LDS only: the refill load and the swap-list stores stay in flight across it
__device__ __forceinline__ void scan_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__global__ __launch_bounds__(SCAN_T) void k_anchor_scan(RngDev *__restrict__ st, const uint32_t *__restrict__ ring,
                                                        const int32_t *__restrict__ count,
                                                        const int32_t *__restrict__ n_ptr, int Ncap, int A,
                                                        long long *__restrict__ joff, int32_t *__restrict__ jbuf,
                                                        long long jcap, int32_t *__restrict__ n_valid,
                                                        uint32_t *__restrict__ status)
{
    __shared__ long long run_s;
    __shared__ int stop_s;
    const int tid = threadIdx.x;
    int N = *n_ptr;
    if (N > Ncap) N = Ncap;
    // offsets of the superpixels' swap lists (jbuf[joff[s] + i] = j of swap i, i = 1 .. n-1)
    if (tid == 0) run_s = 0;
    __syncthreads();
    for (int s0 = 0; s0 < N; s0 += SCAN_T) {
        const int s = s0 + tid;
        const int n = (s < N) ? max(count[s], 0) : 0;
        // block exclusive scan of n (values, not flags): wave scan + wave totals
        long long inc = n;
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long v = __shfl_up(inc, o);
            if (lane >= o) inc += v;
        }
        __shared__ long long wtot[16];
        __syncthreads();
        if (lane == 63) wtot[wv] = inc;
        __syncthreads();
        long long off = run_s, tot = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { if (i < wv) off += wtot[i]; tot += wtot[i]; }
        if (s < N) {
            joff[s] = off + inc - n;
            n_valid[s] = n < A ? n : A;
        }
        __syncthreads();
        if (tid == 0) run_s += tot;
        __syncthreads();
    }
    if (run_s > jcap) { if (tid == 0) atomicOr(status, SPA_ST_RNG_UNDERRUN); return; }

    // The stream is staged through an LDS ring of 4 096 outputs, refilled 1 024 at a time one step before
    // it is needed (the global load of the next piece travels under the current step's voting rounds):
    // invariant before a step: [rpos, rpos + 1024) is in LDS.
    __shared__ uint32_t lring[4096];
    __shared__ int vote_cnt[2][16], vote_chg[2][16];
    unsigned long long rpos = st->rpos;
    const unsigned long long wpos = st->wpos;
    unsigned long long loaded = rpos;
    for (int c = 0; c < 2; ++c) {
        const unsigned long long pos = loaded + (unsigned)tid;
        lring[pos & 4095ull] = pos < wpos ? ring[pos & RNG_MASK] : 0u;
        loaded += SCAN_T;
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    int par = 0;
    for (int s = 0; s < N; ++s) {
        const int n = max(count[s], 0);
        int32_t *J = jbuf + joff[s];
        int i_cur = n - 1;                          // next swap index; the shuffle runs i = n-1 .. 1
        while (i_cur >= 1) {
            // refill piece (if the ring holds fewer than two steps): issued now, written to LDS after the step
            const bool refill = loaded - rpos < 2048ull;
            uint32_t nextraw = 0u;
            if (refill) {
                const unsigned long long pos = loaded + (unsigned)tid;
                nextraw = pos < wpos ? ring[pos & RNG_MASK] : 0u;
            }
            // candidate t of this step: stream output rpos + t
            const unsigned long long pos = rpos + (unsigned)tid;
            const bool have = pos < wpos;
            const uint32_t raw = lring[pos & 4095ull];
            // acceptances before t: fixed point of a = prefix(accept(a)); start from the expected count
            const int k0 = 32 - __clz((unsigned)i_cur + 1u);
            int a = (int)(((unsigned long long)tid * ((unsigned)i_cur + 1u)) >> k0);
            bool acc = false;
            int total = 0;
            for (;;) {                             // terminates: position t is exact after at most t + 1 rounds
                const int i_t = i_cur - a;          // the swap this candidate would serve
                acc = false;
                if (have && i_t >= 1) {
                    const uint32_t v = raw >> __clz((unsigned)i_t + 1u);
                    acc = v <= (unsigned)i_t;
                }
                // one barrier per round: wave counts and "changed" votes go to alternating LDS rows
                const unsigned long long m = __ballot(acc);
                if (lane == 0) vote_cnt[par][wv] = __popcll(m);
                int off = 0, tot = 0;
                // the previous round's `a` is compared after the new prefix is known: two-step protocol
                scan_lds_barrier();
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int c = vote_cnt[par][i];
                    if (i < wv) off += c;
                    tot += c;
                }
                const int a_new = off + (int)spa_rank_in_mask(m);
                const bool changed = a_new != a;
                a = a_new;
                total = tot;
                const unsigned long long mc = __ballot(changed);
                if (lane == 0) vote_chg[par][wv] = mc != 0ull;
                scan_lds_barrier();
                int any = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) any |= vote_chg[par][i];
                par ^= 1;
                if (!any) break;
            }
            // swaps left in this superpixel: i_cur .. 1 (an accepted candidate always has a < R)
            const int R = i_cur;
            if (acc) {
                const int i_t = i_cur - a;
                J[i_t] = (int32_t)(raw >> __clz((unsigned)i_t + 1u));
            }
            // outputs consumed: all examined ones, or up to and including the R-th acceptance
            if (tid == 0) stop_s = -1;
            scan_lds_barrier();
            if (acc && a == R - 1) stop_s = tid;
            if (refill) lring[(loaded + (unsigned)tid) & 4095ull] = nextraw;
            scan_lds_barrier();
            if (refill) loaded += SCAN_T;
            const unsigned long long avail = wpos > rpos ? wpos - rpos : 0ull;
            const int examined = avail < (unsigned long long)SCAN_T ? (int)avail : SCAN_T;
            const int consumed = stop_s >= 0 ? stop_s + 1 : examined;
            rpos += (unsigned)consumed;
            i_cur -= total;
            if (stop_s < 0 && examined < SCAN_T) {   // ring exhausted before the shuffle finished
                if (tid == 0) { atomicOr(status, SPA_ST_RNG_UNDERRUN); st->rpos = rpos; }
                return;
            }
        }
    }
    if (tid == 0) st->rpos = rpos;
}

// one wave per superpixel: ranks[s][p] = original list position that ends at place p < n_valid
__global__ __launch_bounds__(64) void k_anchor_trace(const int32_t *__restrict__ count,
                                                     const int32_t *__restrict__ n_ptr, int Ncap, int A,
                                                     const long long *__restrict__ joff,
                                                     const int32_t *__restrict__ jbuf,
                                                     int32_t *__restrict__ ranks)
{
    const int s = blockIdx.x;
    int N = *n_ptr;
    if (N > Ncap) N = Ncap;
    if (s >= N) return;
    const int lane = threadIdx.x;
    const int n = max(count[s], 0);
    const int nv = n < A ? n : A;
    for (int p = lane; p < A; p += 64) ranks[(long long)s * A + p] = 0;
    if (n <= 0) return;
    const int32_t *J = jbuf + joff[s];
    // lane p (< nv) carries the trace of place p in q; all lanes test 64 swaps per step
    int q = lane;                                   // place p starts at position p
    for (int i0 = 1; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const int j = (i < n) ? J[i] : -1;
        // swap (i, j), met in increasing i: the element that ends at q sat, before this swap, at j if q == i,
        // at i if q == j.  q == i can only be the trace's own start (q = p = i) or a value set by an earlier
        // match, which is smaller than the current i: only the start case remains, handled below.
        for (int p = 0; p < nv; ++p) {
            int qp = __shfl(q, p);
            // the chunk may hold several successive matches for one trace (q changes inside the chunk)
            int lo = 0;
            for (;;) {
                unsigned long long m = __ballot((j == qp || i == qp) && lane >= lo);
                if (!m) break;
                const int l = __ffsll((long long)m) - 1;
                const int il = i0 + l, jl = __shfl(j, l);
                qp = (qp == il) ? jl : il;          // q == i: comes from j; q == j: comes from i
                lo = l + 1;
                if (lo >= 64) break;
            }
            if (lane == p) q = qp;
        }
    }
    if (lane < nv) ranks[(long long)s * A + lane] = q;
}

// ---------------------------------------------------------------------------------------------------
extern "C" int spa_pyrandom_dev_seed(spa_ctx *ctx, uint64_t seed, void *stream)
{
    SPA_ARG(ctx);
    // random.seed(int): init_by_array over the 32-bit digits of abs(seed) (same as spa_pyrandom_create)
    RngDev h;
    memset(&h, 0, sizeof(h));
    uint32_t *mt = h.mt;
    mt[0] = 19650218u;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
    const int klen = key[1] ? 2 : 1;
    int i = 1, j = 0;
    for (int k = 624 > klen ? 624 : klen; k; --k) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        if (++i >= 624) { mt[0] = mt[623]; i = 1; }
        if (++j >= klen) j = 0;
    }
    for (int k = 623; k; --k) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        if (++i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
    RngDev *dev;
    int rc = spa_ws_reserve(ctx, WS_RNG_STATE, sizeof(RngDev), (void **)&dev);
    if (rc != SPA_OK) return rc;
    SPA_HIP(hipStreamSynchronize(spa_stream(stream)));
    SPA_HIP(hipMemcpy(dev, &h, sizeof(h), hipMemcpyHostToDevice));
    ctx->rng_seeded = 1;
    return SPA_OK;
}

// make at least `want` stream outputs available in the ring (asynchronous; data independent, so callers
// issue it a batch ahead on a side stream)
extern "C" int spa_pyrandom_dev_generate(spa_ctx *ctx, int64_t want, void *stream)
{
    SPA_ARG(ctx && ctx->rng_seeded && want >= 0);
    uint32_t *ring;
    int rc = spa_ws_reserve(ctx, WS_RNG_RING, RNG_RING * 4, (void **)&ring);
    if (rc != SPA_OK) return rc;
    if ((unsigned long long)want > RNG_RING - 1024) want = (int64_t)(RNG_RING - 1024);
    hipLaunchKernelGGL(k_mt_generate, dim3(1), dim3(256), 0, spa_stream(stream), (RngDev *)ctx->ws[WS_RNG_STATE], ring,
                       (unsigned long long)want);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// count (Ncap) int32 superpixel sizes in order, n_ptr -> N; ranks (Ncap, n_anchors), n_valid (Ncap) out.
// total_pixels: upper bound of the sum of the sizes (B*H*W): sizes the swap-list workspace.
extern "C" int spa_anchor_ranks_dev(spa_ctx *ctx, const int32_t *count, const int32_t *n_ptr, int32_t Ncap,
                                    int32_t n_anchors, int64_t total_pixels, int32_t *ranks, int32_t *n_valid,
                                    void *stream)
{
    SPA_ARG(ctx && ctx->rng_seeded && count && n_ptr && ranks && n_valid && Ncap > 0 && n_anchors > 0 && n_anchors <= 64);
    SPA_ARG(ctx->ws[WS_RNG_RING] != nullptr);
    hipStream_t s = spa_stream(stream);
    int32_t *jbuf;
    long long *joff;
    int rc = spa_ws_reserve(ctx, WS_RNG_JBUF, (size_t)(total_pixels + 64) * 4, (void **)&jbuf);
    if (rc != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_RNG_JOFF, (size_t)Ncap * 8, (void **)&joff)) != SPA_OK) return rc;
    hipLaunchKernelGGL(k_anchor_scan, dim3(1), dim3(SCAN_T), 0, s, (RngDev *)ctx->ws[WS_RNG_STATE],
                       (const uint32_t *)ctx->ws[WS_RNG_RING], count, n_ptr, Ncap, n_anchors, joff, jbuf,
                       (long long)total_pixels, n_valid, ctx->d_status);
    hipLaunchKernelGGL(k_anchor_trace, dim3(Ncap), dim3(64), 0, s, count, n_ptr, Ncap, n_anchors,
                       (const long long *)joff, (const int32_t *)jbuf, ranks);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
