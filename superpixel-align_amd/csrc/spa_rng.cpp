// Host half of the reference's random number use (plain C++, no device code):
//   * CPython's `random` module — random.seed(1111) at import (batch_spalign_kmeans.py:33) and
//     random.shuffle(inside_coords) per superpixel (:232).  Only the first n_anchors entries
//     of each shuffled list are used (:234), and which pixels those are depends only on the
//     LENGTH of the list, so the host needs just the superpixel sizes; the device maps the
//     returned raster ranks to pixels (spa_select_anchor_pixels).
//   * numpy's legacy global RandomState — np.random.seed(1111) (:34) and
//     xp.random.shuffle(idx) in the k-means initialisation (:148).
// Both are MT19937; what differs is seeding and how bounded integers are drawn.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <stdio.h>
#include <thread>
#include <vector>

#include "../../include/spalign.h"

void spa_set_error(const char *fmt, ...);

namespace {
struct MT {
    uint32_t mt[624];
    int idx;
    void init_genrand(uint32_t s)
    {
        mt[0] = s;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    void init_by_array(const uint32_t *key, int klen)
    {
        init_genrand(19650218u);
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
        idx = 624;
    }
    uint32_t out[624];
    void export_state(uint32_t *state, int *pos) const { for (int i = 0; i < 624; ++i) state[i] = mt[i]; *pos = idx; }
    void import_state(const uint32_t *state, int pos)
    {
        // (out[] holds the tempered words of the current block: rebuilt, so that a state taken in the middle of a block continues)
        for (int i = 0; i < 624; ++i) {
            uint32_t t = mt[i] = state[i];
            t ^= (t >> 11);
            t ^= (t << 7) & 0x9d2c5680u;
            t ^= (t << 15) & 0xefc60000u;
            t ^= (t >> 18);
            out[i] = t;
        }
        idx = pos;
    }
    void refill() { refill_to(out); idx = 0; }
    void refill_to(uint32_t *dst)
    {
        // the classic three-segment regeneration (no modulo in the loops), then tempering of
        // the whole block: both loops vectorise
        int k = 0;
        for (; k < 624 - 397; ++k) {
            uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
            mt[k] = mt[k + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        }
        for (; k < 623; ++k) {
            uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
            mt[k] = mt[k + (397 - 624)] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        }
        uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        for (int i = 0; i < 624; ++i) {
            uint32_t t = mt[i];
            t ^= (t >> 11);
            t ^= (t << 7) & 0x9d2c5680u;
            t ^= (t << 15) & 0xefc60000u;
            t ^= (t >> 18);
            dst[i] = t;
        }
    }
    inline uint32_t next()
    {
        if (idx >= 624) refill();
        return out[idx++];
    }
};
}  // namespace

static bool pyrandom_have_avx512();

// The CPython stream for the anchors: the generator's tempered outputs in a small buffer of the object (four refills = 10 KB,
// L1 resident) that the rejection sampling reads with 16-lane loads.  (Round 4 also tried MT19937 on a background thread
// through a ring: with a 66 MB ring the consumer was memory bound, with a 2 MB one the blocks crossed between two cores'
// caches at 4-6 GB/s — the consumer waited 26-35 of 73 ms for the producer; generating in place costs ~0.15 ns per output.)
struct spa_pyrandom {
    MT g;
    static const int BS = 624 * 4;
    uint32_t buf[BS + 16];
    const uint32_t *cur = nullptr;
    int avail = 0;
    bool vec = false;
    double waited = 0;                                  // (timing aid of the ring experiment: stays 0)
    void fill();
    inline void need()
    {
        if (avail > 0) return;
        fill();
        cur = buf;
        avail = BS;
    }
};
struct spa_nprandom { MT g; };

extern "C" int spa_pyrandom_create(uint64_t seed, spa_pyrandom **out)
{
    if (!out) return SPA_ERR_ARG;
    spa_pyrandom *r = new spa_pyrandom();
    // random.seed(int): init_by_array over the 32-bit digits of abs(seed)
    uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
    r->g.init_by_array(key, key[1] ? 2 : 1);
    r->vec = pyrandom_have_avx512();
    *out = r;
    return SPA_OK;
}
extern "C" void spa_pyrandom_destroy(spa_pyrandom *r)
{
    delete r;
}

// Two phases so that only the generator itself is sequential:
//   phase 1 (this thread, in stream order): the accepted draws j of every swap of every superpixel, stored in DRAW order —
//            for i in reversed(range(1, n)): j = randbelow(i + 1), where randbelow takes the top bit_length(i+1) bits of a
//            32-bit output and redraws while >= i+1; jd[t] is the draw for i = n - 1 - t;
//   phase 2 (worker threads, one superpixel each): which original ranks end in the first n_anchors places.  Not by replaying
//            the n swaps on an array: the place p of an anchor is traced BACKWARDS through the swaps (i = 1 .. n-1: p == i ->
//            p = j_i; p == j_i -> p = i).  A place only moves up to the current i and is never met again as `i` afterwards, so
//            beyond the first n_anchors steps the only question per swap is "is j_i one of the <= n_anchors tracked places":
//            a vector compare of 16 draws against each place, and a scalar fix-up on the rare hit (probability ~A / i).
// Superpixels are processed in groups of ~2 M draws; phase 2 of a group overlaps phase 1 of the next one.
// Both phases have an AVX-512 form chosen at run time (round 4: 103 ms per batch of 30 full-size images was the anchor mode's
// step time); the scalar forms below compute the same values on any x86-64.
#if defined(__x86_64__)
#include <immintrin.h>
#define SPA_RNG_X86 1
#endif

namespace {
// ---- phase 2 ------------------------------------------------------------------------------------------------------
// jd: the n - 1 draws of one shuffle in draw order (jd[t] = j for i = n - 1 - t); out[a] = original rank that ends at place a
#ifdef SPA_RNG_X86
__attribute__((target("avx512f"))) static void trace_avx512(const int32_t *jd, int32_t n, int32_t nv, int32_t *out)
{
    int32_t p[16];
    for (int a = 0; a < 16; ++a) p[a] = a < nv ? a : -1;
    // the first steps, where a place can still be `i` itself: i < nv (places start at 0 .. nv-1), plus the steps needed to
    // bring the remaining count to a multiple of 16
    int32_t i = 1;
    for (; i < n && (i < nv || ((n - i) & 15)); ++i) {
        const int32_t j = jd[n - 1 - i];
        for (int a = 0; a < nv; ++a) {
            if (p[a] == i) p[a] = j;
            else if (p[a] == j) p[a] = i;
        }
    }
    // i >= nv from here on and every tracked place is < i: only `j_i == place` can happen (the place then becomes i).  The
    // places live as the 16 lanes of one register; a block of 16 draws is tested against lane a broadcast, a = 0 .. nv-1.
    // jd index of step i is n - 1 - i: the steps i .. i + 15 are jd[n - 16 - i .. n - 1 - i], step i + k in lane 15 - k.
    __m512i pv = _mm512_loadu_si512((const void *)p);
    while (i < n) {
        const __m512i v = _mm512_loadu_si512((const void *)(jd + (n - 16 - i)));
        __mmask16 hit = 0;
        for (int a = 0; a < nv; ++a)
            hit |= _mm512_cmpeq_epi32_mask(v, _mm512_permutexvar_epi32(_mm512_set1_epi32(a), pv));
        if (hit) {
            // (probability ~ 16 nv / i) from the first hit on, step by step: a place that has just moved to i + k may be drawn
            // again by a later step of the same block
            _mm512_storeu_si512((void *)p, pv);
            const int first = 15 - (31 - __builtin_clz((unsigned)hit));        // k of the first step that hits
            for (int k = first; k < 16; ++k) {
                const int32_t j = jd[n - 1 - (i + k)];
                for (int a = 0; a < nv; ++a)
                    if (p[a] == j) p[a] = i + k;
            }
            pv = _mm512_loadu_si512((const void *)p);
        }
        i += 16;
    }
    _mm512_storeu_si512((void *)p, pv);
    for (int a = 0; a < nv; ++a) out[a] = p[a];
}
#endif

static bool have_avx512()
{
#ifdef SPA_RNG_X86
    static const bool ok = __builtin_cpu_supports("avx512f") && !getenv("SPA_RNG_SCALAR");
    return ok;
#else
    return false;
#endif
}

static void replay_group(const int32_t *count, const int32_t *draws, const int64_t *doff, int32_t s0,
                         int32_t s1, int32_t A, int32_t *ranks, const int32_t *n_valid, int tid, int nthreads)
{
    const bool vec = have_avx512() && A <= 16;
    for (int32_t s = s0 + tid; s < s1; s += nthreads) {
        const int32_t n = count[s];
        if (n <= 0 || n_valid[s] <= 0) continue;
        const int32_t *jd = draws + doff[s - s0];         // n - 1 draws, draw order
        int32_t *o = ranks + (int64_t)s * A;
#ifdef SPA_RNG_X86
        if (vec) { trace_avx512(jd, n, n_valid[s], o); continue; }
#endif
        // without the vector compares: replay the swaps on an array (one swap per step instead of n_anchors compares)
        std::vector<int32_t> perm((size_t)n);
        for (int32_t i = 0; i < n; ++i) perm[i] = i;
        for (int32_t i = n - 1; i >= 1; --i) {
            const int32_t j = jd[n - 1 - i];
            const int32_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
        }
        for (int a = 0; a < n_valid[s]; ++a) o[a] = perm[a];
    }
}

// ---- phase 1 ------------------------------------------------------------------------------------------------------
// the draws of one shuffle of n elements, in draw order, into jd[0 .. n-2]
static void draws_scalar(spa_pyrandom &q, int32_t n, int32_t *jd)
{
    // randbelow(i + 1) for i = n-1 .. 1: top bit_length(i+1) bits of a 32-bit output, redrawn while >= i+1.  Branch-free over
    // the generator's output block: every candidate is stored at the current slot (a rejected one is overwritten by the next
    // try) and the slot advances only on acceptance; the shift is constant while i+1 stays in (2^(k-1), 2^k].
    int32_t i = n - 1;
    int32_t *w = jd;
    while (i >= 1) {
        const int sh = __builtin_clz((uint32_t)i + 1u);          // 32 - bit_length(i + 1)
        const int32_t band_lo = (int32_t)(0x80000000u >> sh) - 1;  // bit_length(i + 1) stays k while i >= 2^(k-1) - 1
        const int32_t stop = band_lo > 1 ? band_lo : 1;
        while (i >= stop) {
            q.need();
            const uint32_t *o = q.cur;
            const int avail = q.avail;
            int used = 0;
            while (used < avail && i >= stop) {
                const uint32_t v = o[used++] >> sh;
                *w = (int32_t)v;
                const int32_t acc = (int32_t)(v <= (uint32_t)i);
                w += acc;
                i -= acc;
            }
            q.cur += used; q.avail -= used;
        }
    }
}

#ifdef SPA_RNG_X86
__attribute__((target("avx512f"))) static inline void twist16(uint32_t *mt, int k, int m)
{
    const __m512i up = _mm512_set1_epi32((int)0x80000000u), lo = _mm512_set1_epi32(0x7fffffff), one = _mm512_set1_epi32(1),
                  mag = _mm512_set1_epi32((int)0x9908b0dfu);
    const __m512i a = _mm512_loadu_si512((const void *)(mt + k)), b = _mm512_loadu_si512((const void *)(mt + k + 1));
    const __m512i y = _mm512_or_si512(_mm512_and_si512(a, up), _mm512_and_si512(b, lo));
    const __m512i odd = _mm512_sub_epi32(_mm512_setzero_si512(), _mm512_and_si512(y, one));
    const __m512i r = _mm512_xor_si512(_mm512_xor_si512(_mm512_loadu_si512((const void *)(mt + m)), _mm512_srli_epi32(y, 1)),
                                       _mm512_and_si512(odd, mag));
    _mm512_storeu_si512((void *)(mt + k), r);
}

__attribute__((target("avx512f"))) static void refill_avx512(MT &g, uint32_t *dst)
{
    // the same three-segment regeneration and tempering, 16 lanes at a time (the segments' dependences are 227 and 397
    // elements apart; the read of mt[k + 1] precedes the write of mt[k .. k + 15] inside an iteration)
    uint32_t *mt = g.mt;
    auto twist1 = [&](int k, int k1, int m) {
        const uint32_t y = (mt[k] & 0x80000000u) | (mt[k1] & 0x7fffffffu);
        mt[k] = mt[m] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
    };
    int k = 0;
    for (; k + 16 <= 227; k += 16) twist16(mt, k, k + 397);
    for (; k < 227; ++k) twist1(k, k + 1, k + 397);
    for (; k + 16 <= 623; k += 16) twist16(mt, k, k - 227);
    for (; k < 623; ++k) twist1(k, k + 1, k - 227);
    twist1(623, 0, 396);
    const __m512i m7 = _mm512_set1_epi32((int)0x9d2c5680u), m15 = _mm512_set1_epi32((int)0xefc60000u);
    for (int i = 0; i < 624; i += 16) {
        __m512i t = _mm512_loadu_si512((const void *)(mt + i));
        t = _mm512_xor_si512(t, _mm512_srli_epi32(t, 11));
        t = _mm512_xor_si512(t, _mm512_and_si512(_mm512_slli_epi32(t, 7), m7));
        t = _mm512_xor_si512(t, _mm512_and_si512(_mm512_slli_epi32(t, 15), m15));
        t = _mm512_xor_si512(t, _mm512_srli_epi32(t, 18));
        _mm512_storeu_si512((void *)(dst + i), t);
    }
}

__attribute__((target("avx512f,popcnt"))) static void draws_avx512(spa_pyrandom &q, int32_t n, int32_t *jd)
{
    int32_t i = n - 1;
    int32_t *w = jd;
    const __m512i lane = _mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    while (i >= 1) {
        const int sh = __builtin_clz((uint32_t)i + 1u);
        const int32_t band_lo = (int32_t)(0x80000000u >> sh) - 1;
        const int32_t stop = band_lo > 1 ? band_lo : 1;
        const __m128i shc = _mm_cvtsi32_si128(sh);
        while (i >= stop) {
            q.need();
            // 16 outputs at a time while all 16 could be accepted inside the band.  Output k is accepted iff
            // v_k <= i - (accepted before k): surely when v_k <= i - k, surely not when v_k > i; a block with an output in
            // between (probability ~ 16 * 8 / 2^bits) is taken one output at a time
            // (cursor and count live in locals: `avail` is an int like the draws stored through w, so as members the compiler
            // had to reload them after every store)
            const uint32_t *cur = q.cur;
            int avail16 = q.avail;
            while (avail16 >= 16 && i - 16 >= stop) {
                const __m512i v = _mm512_srl_epi32(_mm512_loadu_si512((const void *)cur), shc);
                const __m512i iv = _mm512_set1_epi32(i);
                const __mmask16 yes = _mm512_cmple_epu32_mask(v, _mm512_sub_epi32(iv, lane));
                const __mmask16 no = _mm512_cmpgt_epu32_mask(v, iv);
                if ((__mmask16)(yes | no) != (__mmask16)0xffff) break;
                // (compress in a register + one full store: the memory form of vpcompressd is microcoded — tens of cycles — on
                // several x86 cores; the 16 lanes stored beyond the accepted ones are overwritten by the next block)
                _mm512_storeu_si512((void *)w, _mm512_maskz_compress_epi32(yes, v));
                const int c = __builtin_popcount((unsigned)yes);
                w += c; i -= c; cur += 16; avail16 -= 16;
            }
            q.cur = cur; q.avail = avail16;
            // one block's worth (or the band's / generator block's tail) output by output
            if (q.avail == 0) continue;
            const uint32_t *o = q.cur;
            const int avail = q.avail < 16 ? q.avail : 16;
            int used = 0;
            while (used < avail && i >= stop) {
                const uint32_t v = o[used++] >> sh;
                *w = (int32_t)v;
                const int32_t acc = (int32_t)(v <= (uint32_t)i);
                w += acc;
                i -= acc;
            }
            q.cur += used; q.avail -= used;
        }
    }
}
#endif
}  // namespace

static bool pyrandom_have_avx512() { return have_avx512(); }

void spa_pyrandom::fill()
{
    uint32_t *dst = buf;
    for (int k = 0; k < BS / 624; ++k, dst += 624) {
#ifdef SPA_RNG_X86
        if (vec) refill_avx512(g, dst); else
#endif
            g.refill_to(dst);
    }
}

extern "C" int spa_pyrandom_shuffle_select_host(spa_pyrandom *r, const int32_t *count, int32_t N,
                                                int32_t A, int32_t *ranks, int32_t *n_valid)
{
    if (!r || !count || !ranks || !n_valid || A <= 0) return SPA_ERR_ARG;
    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = hw >= 16 ? 8 : (hw >= 4 ? (int)hw / 2 : 1);
    if (getenv("SPA_RNG_THREADS")) nthreads = atoi(getenv("SPA_RNG_THREADS")) > 0 ? atoi(getenv("SPA_RNG_THREADS")) : 1;
    const int64_t group_draws = getenv("SPA_RNG_GROUP") ? atoll(getenv("SPA_RNG_GROUP")) : (2 << 20);       // 8 MB of draws per group (smaller groups were measured no faster and start more threads)
    const bool vec = have_avx512();
    std::vector<int32_t> buf[2];
    std::vector<int64_t> doff[2];
    std::vector<std::thread> workers;
    int cur = 0;
    int32_t s = 0;
    const bool timing = getenv("SPA_RNG_TIMING") != nullptr;
    double t_draw = 0, t_join = 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    while (s < N) {
        const double t0 = timing ? now() : 0;
        // ---- phase 1 for the group [s, e)
        std::vector<int32_t> &d = buf[cur];
        std::vector<int64_t> &off = doff[cur];
        off.clear();
        int64_t total = 0;
        int32_t e = s;
        while (e < N && (e == s || total + (count[e] > 0 ? count[e] : 0) <= group_draws)) {
            off.push_back(total);
            total += count[e] > 0 ? count[e] : 0;
            ++e;
        }
        d.resize((size_t)(total > 0 ? total : 1) + 32);      // (+ slack: the vector forms store / load whole 16-lane blocks)
        for (int32_t t = s; t < e; ++t) {
            const int32_t n = count[t];
            const int32_t nv = n < A ? (n < 0 ? 0 : n) : A;
            n_valid[t] = nv;
            for (int a = 0; a < A; ++a) ranks[(int64_t)t * A + a] = 0;
            if (n <= 1) continue;                               // (a list of one element: no draw; its rank 0 is already there)
            int32_t *jd = d.data() + off[t - s];
#ifdef SPA_RNG_X86
            if (vec) { draws_avx512(*r, n, jd); continue; }
#endif
            draws_scalar(*r, n, jd);
        }
        // ---- phase 2 of the previous group must be done before its buffers are reused next time
        const double t1 = timing ? now() : 0;
        for (auto &w : workers) w.join();
        workers.clear();
        if (timing) { t_draw += t1 - t0; t_join += now() - t1; }
        const int32_t *dp = d.data();
        const int64_t *op = off.data();
        if (nthreads <= 1) {
            replay_group(count, dp, op, s, e, A, ranks, n_valid, 0, 1);
        } else {
            for (int t = 0; t < nthreads; ++t)
                workers.emplace_back(replay_group, count, dp, op, s, e, A, ranks, n_valid, t, nthreads);
        }
        cur ^= 1;
        s = e;
    }
    const double t2 = timing ? now() : 0;
    for (auto &w : workers) w.join();
    if (timing) {
        fprintf(stderr, "shuffle_select: draws %.1f ms, waiting for the traces %.1f ms (+ %.1f at the end), %d trace threads\n",
                t_draw * 1e3, t_join * 1e3, (now() - t2) * 1e3, nthreads);
        r->waited = 0;
    }
    return SPA_OK;
}

extern "C" int spa_nprandom_create(uint32_t seed, spa_nprandom **out)
{
    if (!out) return SPA_ERR_ARG;
    spa_nprandom *r = new spa_nprandom();
    r->g.init_genrand(seed);      // np.random.seed(int) -> init_genrand
    *out = r;
    return SPA_OK;
}
extern "C" void spa_nprandom_destroy(spa_nprandom *r) { delete r; }

extern "C" int spa_nprandom_state(spa_nprandom *r, uint32_t *state628)
{
    if (!r || !state628) return SPA_ERR_ARG;
    r->g.export_state(state628, (int *)&state628[624]);
    state628[625] = state628[626] = state628[627] = 0u;
    return SPA_OK;
}

extern "C" int spa_nprandom_set_state(spa_nprandom *r, const uint32_t *state628)
{
    if (!r || !state628 || state628[624] > 624u) return SPA_ERR_ARG;
    r->g.import_state(state628, (int)state628[624]);
    return SPA_OK;
}

extern "C" int spa_nprandom_shuffle_host(spa_nprandom *r, int64_t *a, int64_t n)
{
    if (!r || (!a && n > 0)) return SPA_ERR_ARG;
    for (int64_t i = n - 1; i >= 1; --i) {
        uint64_t max = (uint64_t)i, mask = max, value;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
        mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        if (max <= 0xffffffffULL) {
            while ((value = ((uint64_t)r->g.next() & mask)) > max) {}
        } else {
            do {
                uint64_t hi = r->g.next(), lo = r->g.next();
                value = ((hi << 32) | lo) & mask;
            } while (value > max);
        }
        int64_t t = a[i]; a[i] = a[value]; a[value] = t;
    }
    return SPA_OK;
}
