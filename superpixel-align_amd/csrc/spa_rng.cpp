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
    void refill()
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
            out[i] = t;
        }
        idx = 0;
    }
    inline uint32_t next()
    {
        if (idx >= 624) refill();
        return out[idx++];
    }
};
}  // namespace

struct spa_pyrandom { MT g; };
struct spa_nprandom { MT g; };

extern "C" int spa_pyrandom_create(uint64_t seed, spa_pyrandom **out)
{
    if (!out) return SPA_ERR_ARG;
    spa_pyrandom *r = new spa_pyrandom();
    // random.seed(int): init_by_array over the 32-bit digits of abs(seed)
    uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
    r->g.init_by_array(key, key[1] ? 2 : 1);
    *out = r;
    return SPA_OK;
}
extern "C" void spa_pyrandom_destroy(spa_pyrandom *r) { delete r; }

// Two phases so that only the generator itself is sequential:
//   phase 1 (this thread, in stream order): the accepted draws j of every swap of every
//            superpixel — for i in reversed(range(1, n)): j = randbelow(i + 1), where randbelow
//            takes the top bit_length(i+1) bits of a 32-bit output and redraws while >= i+1;
//   phase 2 (worker threads, one superpixel each): replay the swaps on an identity array and
//            read off its first n_anchors entries (which original ranks ended in front).
// Superpixels are processed in groups of ~2 M draws; phase 2 of a group overlaps phase 1 of the
// next one.
static void replay_group(const int32_t *count, const int32_t *draws, const int64_t *doff, int32_t s0,
                         int32_t s1, int32_t A, int32_t *ranks, const int32_t *n_valid, int tid, int nthreads)
{
    std::vector<int32_t> perm;
    for (int32_t s = s0 + tid; s < s1; s += nthreads) {
        const int32_t n = count[s];
        if (n <= 0) continue;
        perm.resize((size_t)n);
        for (int32_t i = 0; i < n; ++i) perm[i] = i;
        const int32_t *j_of = draws + doff[s - s0];       // j_of[i] for i = 1 .. n-1
        for (int32_t i = n - 1; i >= 1; --i) {
            const int32_t j = j_of[i];
            const int32_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
        }
        for (int a = 0; a < n_valid[s]; ++a) ranks[(int64_t)s * A + a] = perm[a];
    }
}

extern "C" int spa_pyrandom_shuffle_select_host(spa_pyrandom *r, const int32_t *count, int32_t N,
                                                int32_t A, int32_t *ranks, int32_t *n_valid)
{
    if (!r || !count || !ranks || !n_valid || A <= 0) return SPA_ERR_ARG;
    unsigned hw = std::thread::hardware_concurrency();
    const int nthreads = hw >= 16 ? 8 : (hw >= 4 ? (int)hw / 2 : 1);
    const int64_t group_draws = 2 << 20;
    std::vector<int32_t> buf[2];
    std::vector<int64_t> doff[2];
    std::vector<std::thread> workers;
    int cur = 0;
    int32_t s = 0;
    while (s < N) {
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
        d.resize((size_t)(total > 0 ? total : 1));
        for (int32_t t = s; t < e; ++t) {
            const int32_t n = count[t];
            const int32_t nv = n < A ? (n < 0 ? 0 : n) : A;
            n_valid[t] = nv;
            for (int a = 0; a < A; ++a) ranks[(int64_t)t * A + a] = 0;
            if (n <= 0) continue;
            int32_t *j_of = d.data() + off[t - s];
            // randbelow(i + 1) for i = n-1 .. 1: top bit_length(i+1) bits of a 32-bit output, redrawn
            // while >= i+1.  Branch-free over the generator's output block: every candidate is
            // stored at j_of[i] (a rejected one is overwritten by the next try for the same i) and
            // i steps down only on acceptance; the shift is constant while i+1 stays in (2^(k-1), 2^k].
            int32_t i = n - 1;
            while (i >= 1) {
                const int sh = __builtin_clz((uint32_t)i + 1u);          // 32 - bit_length(i + 1)
                const int32_t band_lo = (int32_t)(0x80000000u >> sh) - 1;  // bit_length(i + 1) stays k while i >= 2^(k-1) - 1
                const int32_t stop = band_lo > 1 ? band_lo : 1;
                while (i >= stop) {
                    if (r->g.idx >= 624) r->g.refill();
                    const uint32_t *o = r->g.out + r->g.idx;
                    const int avail = 624 - r->g.idx;
                    int used = 0;
                    while (used < avail && i >= stop) {
                        const uint32_t v = o[used++] >> sh;
                        j_of[i] = (int32_t)v;
                        i -= (int32_t)(v <= (uint32_t)i);
                    }
                    r->g.idx += used;
                }
            }
        }
        // ---- phase 2 of the previous group must be done before its buffers are reused next time
        for (auto &w : workers) w.join();
        workers.clear();
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
    for (auto &w : workers) w.join();
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
