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
    inline uint32_t next()
    {
        if (idx >= 624) {
            for (int k = 0; k < 624; ++k) {
                uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
};
}  // namespace

struct spa_pyrandom { MT g; std::vector<int32_t> draws; };
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

extern "C" int spa_pyrandom_shuffle_select_host(spa_pyrandom *r, const int32_t *count, int32_t N,
                                                int32_t A, int32_t *ranks, int32_t *n_valid)
{
    if (!r || !count || !ranks || !n_valid || A <= 0) return SPA_ERR_ARG;
    for (int32_t s = 0; s < N; ++s) {
        const int32_t n = count[s];
        const int32_t nv = n < A ? (n < 0 ? 0 : n) : A;
        n_valid[s] = nv;
        for (int a = 0; a < A; ++a) ranks[(int64_t)s * A + a] = 0;
        if (n <= 0) continue;
        // for i in reversed(range(1, n)): j = randbelow(i + 1); x[i], x[j] = x[j], x[i]
        // randbelow: k = (i+1).bit_length(); r = getrandbits(k); while r >= i+1: redraw
        r->draws.resize((size_t)n);
        int32_t *j_of = r->draws.data();
        for (int32_t i = n - 1; i >= 1; --i) {
            const uint32_t m = (uint32_t)i + 1u;
            const int k = 32 - __builtin_clz(m);
            uint32_t v = r->g.next() >> (32 - k);
            while (v >= m) v = r->g.next() >> (32 - k);
            j_of[i] = (int32_t)v;
        }
        // which original element ends at position a < nv: undo the swaps, last swap first
        for (int a = 0; a < nv; ++a) {
            int32_t q = a;
            for (int32_t i = 1; i < n; ++i) {
                const int32_t j = j_of[i];
                if (q == i) q = j;
                else if (q == j) q = i;
            }
            ranks[(int64_t)s * A + a] = q;
        }
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
