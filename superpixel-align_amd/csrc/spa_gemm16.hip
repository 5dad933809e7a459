// The Winograd-domain GEMMs of the float32 DRN on the 16-bit matrix cores, at float32 accuracy.
//
// A float32 operand x (after an exact power-of-two scaling into half precision's range, spa_wino.hip) is stored as TWO
// half-precision planes  h = rn16(x), l = rn16(x - h):  x = h + l up to 2^-22 |x| (11 + 11 significand bits, round to
// nearest at both levels), and a product is three matrix instructions
//
//        a . b  =  ah . bh  +  ah . bl  +  al . bh          ( al . bl <= 2^-22 |a b| dropped )
//
// accumulated in float32 by v_mfma_f32_16x16x32_f16 — the products of half-precision numbers are exact in the float32
// accumulator, so what is lost is the representation error of the planes, not the arithmetic: measured through the whole
// DRN-D-22 the final map is as far from the float64 network as with float32 operands (tools/wino_network_error.py;
// two bfloat16 planes are 10x worse and were rejected).  The 16-bit pipe runs 16x the float32 pipe's rate, so three
// instructions per product are 5.3x less matrix time than v_mfma_f32_16x16x4_f32 — the GEMM batches of a Winograd layer
// (36 GEMMs, K = Cin) stop being bound by the float32 matrix peak, which was the wall of the whole float32 network.
//
// Operands.  Wt (the Winograd weights, static): two planes prepared once — a row of Cin elements is Cin * 4 bytes and every
// group of 32 consecutive channels is one 128-byte LDS row = [32 x h | 32 x l], eight 16-byte chunks (chunk q < 4: h of
// channels 8q..8q+7, chunk 4 + q: their l).  X (the transformed activations V): plain float32, exactly what the float32
// path's input transform writes (same bytes per element as two planes); a lane reads the 32 bytes of its 8 channels,
// multiplies by the tile's power-of-two scale (position z: 2^(14 - e - p_i - p_j), e from the tracked maximum of the layer
// input, spa_wino.hip) and splits in registers — 24 vector instructions per fragment next to 3 x MI matrix instructions
// that consume it, so the split costs the matrix pipe nothing, and the input transform stays the streaming float32 kernel
// (writing the planes from the transform was measured: 978 us instead of 608 per 15 images of a 512-channel layer, the
// conversions serialise with its loads and stores at two waves per SIMD).  Staging (global_load_lds, XOR swizzle on the source address, two buffers, the next
// tile's first K step staged during the last K step, counted vmcnt over the epilogue stores, persistent workgroups on
// XCD-contiguous tile ranges) is the float32 kernel's (spa_conv32.hip), tile 256 x 256 (or 128 x 128 for 128 output
// channels), 8 waves; a lane's fragment is one 16-byte chunk per plane, and the K step of 32 channels is 3 x MI x NJ
// matrix instructions per wave.
//
// Measured on the 512 -> 512 layer, 30 images (36 x 61 440 x 512 x 512): 3.54-3.64 ms = 0.96-0.98 PFLOP/s executed.  The same
// loop without the in-register split 3.23, with the global loads compiled out (LDS reads + split + MFMA) 2.89 = 1.2 PFLOP/s,
// which is what dense 16-bit MFMA sustains on this part (the bf16 kernel's loop, spa_conv.hip, tops out at the same rate).
// Touching the lines of K step t + 2 one step early (a 4-byte global_load_lds per lane into a dump area, counted vmcnt so
// that it stays in flight) made it slower (3.70): the wait at the end of a K step is not HBM latency.  Splitting the pixel
// fragments one fragment ahead, a quarter fragment (6 vector instructions) after every 6 matrix instructions with the order
// pinned by scheduling barriers, changed nothing (3.53): the two waves of a SIMD already overlap each other's phases.
#include "spa_common.h"
#include <stdlib.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

#define G16_THREADS 512
// workgroup barrier that orders LDS only: __syncthreads() also waits for every outstanding global load and store
__device__ __forceinline__ void g16_lds_barrier()
{
    __builtin_amdgcn_sched_barrier(0);          // nothing (the split's vector work included) moves across
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int BM, int BN>
__global__ __launch_bounds__(G16_THREADS, (BM == 256 && BN == 128) ? 3 : 2) void k_gemm_f16x3(const char *__restrict__ X, const char *__restrict__ Wt,
                                                            float *__restrict__ Y, int rows_per_z, int Cin, int Cout,
                                                            int ntiles, int total_tiles, int zcount, long long xz,
                                                            long long wz, long long yz, const unsigned *__restrict__ amax)
{
    extern __shared__ __attribute__((aligned(1024))) char lds16[];   // [2] weight tiles | [2] row tiles, 128 bytes per row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // uniform: the staging's block and LDS addresses stay scalar
    const long long all_tiles = (long long)zcount * total_tiles;
    const int nwg = total_tiles;
    int r0, n0;
    const char *wbase, *xbase;
    float *ybase;
    // scale of the X operand: 2^(14 - e) for the layer, 2^-(p_i + p_j) for position z = 6 i + j (p = 4 4 4 3 3 4)
    float sb;
    {
        const unsigned bits = *amax;
        int e = (int)(bits >> 23) - 127;
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
        sb = __uint_as_float((unsigned)(127 + 14 - (bits == 0u ? 0 : e)) << 23);
    }
    float zscale_next = 0.f;
    auto locate = [&](long long vid) {
        const int z = (int)(vid / total_tiles);
        {
            const int zi = z / 6, zj = z - zi * 6;
            const int psum = ((0x433444 >> (4 * zi)) & 15) + ((0x433444 >> (4 * zj)) & 15);
            zscale_next = sb * __uint_as_float((unsigned)(127 - psum) << 23);
        }
        int id = (int)(vid - (long long)z * total_tiles);
        {
            const int q = nwg / 8, rem = nwg % 8, xcd = id % 8, idx = id / 8;
            id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
        }
        const int nt = id % ntiles, pt = id / ntiles;          // the channel tiles of one row tile are adjacent
        r0 = pt * BN; n0 = nt * BM;
        wbase = Wt + ((long long)z * wz + (long long)n0 * Cin) * 4;
        xbase = X + ((long long)z * xz + (long long)r0 * Cin) * 4;
        ybase = Y + (long long)z * yz;
    };
    constexpr int WN = 4;                                // waves along the rows of X
    constexpr int MI = BM == 256 ? 8 : 4;                // 16-channel MFMA tiles per wave
    constexpr int NJ = BN / WN / 16;                     // 16-row MFMA tiles per wave
    constexpr int WROWS = MI * 16;
    static_assert(BM / WROWS * WN == 8, "8 waves");

    char *wbuf = lds16, *xbuf = lds16 + 2 * (BM * 128);
    const int sub = lane >> 3, cs = lane & 7;
    const int chunk_byte = (cs ^ sub) << 4;        // staged row = block * 8 + sub: (row & 7) = sub for every block
    // (round 4: the two 16-byte reads of a fragment row are served with 2-way bank conflicts by this image — SQ_LDS_BANK_CONFLICT
    // 1.33 x SQ_ACTIVE_INST_LDS.  The conflict-free image of spa_conv32.hip's pixel segment, c ^ g(r & 7) with g = 0 0 1 1 4 4 5 5,
    // was built here too, counted 0 conflicts, and ran 1.3-2 % SLOWER in same-box A/B runs (512 -> 512: 6.96 vs 6.86 ms, 256 ->
    // 256: 2.69 vs 2.63): the LDS is not what this loop waits for, so the X tile keeps the weights' image)
    const int nk = Cin / 32;
    int par = 0;                                   // buffer parity carried from tile to tile
    auto stage = [&](int t, int buf) {
        const char *wk = wbase + (long long)t * 128 + chunk_byte;
        char *dw = wbuf + buf * (BM * 128);
#pragma unroll
        for (int r = 0; r < BM / 64; ++r) {
            const int blk = r * 8 + wave;                       // 8 rows = 1 KB per instruction
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wk + (long long)(blk * 8 + sub) * Cin * 4),
                                             (__attribute__((address_space(3))) void *)(dw + blk * 1024), 16, 0, 0);
        }
        const char *xk = xbase + (long long)t * 128 + chunk_byte;
        char *dx = xbuf + buf * (BN * 128);
#pragma unroll
        for (int r = 0; r < BN / 64; ++r) {
            const int blk = r * 8 + wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xk + (long long)(blk * 8 + sub) * Cin * 4),
                                             (__attribute__((address_space(3))) void *)(dx + blk * 1024), 16, 0, 0);
        }
    };

    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fk = lane >> 4;
    long long vid = blockIdx.x;
    if (vid >= all_tiles) return;
    bool first_tile = true;
    locate(vid);
    stage(0, 0);
    float zscale = zscale_next;
    int e_r0 = 0, e_n0 = 0;
    float *e_y = nullptr;
    bool more = false;
    for (;;) {
        f32x4 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // this tile's first K step was staged BEFORE the previous tile's MI * NJ epilogue stores per wave (vmcnt counts
        // both, in order): waiting until that many operations remain lets the stores drain under this tile's matrix work
        if (!first_tile) {
            static_assert(MI * NJ <= 63, "vmcnt range");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MI * NJ) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        first_tile = false;
        __syncthreads();
        for (int t = 0; t < nk; ++t) {
            const int cur = (t + par) & 1;
            if (t + 1 < nk) stage(t + 1, cur ^ 1);
            else {
                // last K step: the other buffers are free — stage the next tile's first K step under this step's matrix work
                e_r0 = r0; e_n0 = n0; e_y = ybase;
                vid += gridDim.x;
                more = vid < all_tiles;
                if (more) { locate(vid); stage(0, cur ^ 1); }
            }
            const char *lw = wbuf + cur * (BM * 128), *lx = xbuf + cur * (BN * 128);
            const float sc = zscale;
            if (t + 1 == nk) zscale = zscale_next;          // (locate() above has moved on to the next tile)
            f16x8 wh[MI], wl[MI], ph[NJ], pl[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = wm * WROWS + i * 16 + frow;
                wh[i] = *(const f16x8 *)(lw + row * 128 + ((fk ^ (row & 7)) << 4));
                wl[i] = *(const f16x8 *)(lw + row * 128 + (((4 + fk) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = wn * (NJ * 16) + j * 16 + frow;
                // float32 channels 8 fk .. 8 fk + 7 of the row: chunks 2 fk and 2 fk + 1; scale (exact), split
                const f32x4 a = *(const f32x4 *)(lx + row * 128 + (((2 * fk) ^ (row & 7)) << 4));
                const f32x4 b = *(const f32x4 *)(lx + row * 128 + (((2 * fk + 1) ^ (row & 7)) << 4));
                const f32x8 v = (f32x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]} * sc;
                ph[j] = __builtin_convertvector(v, f16x8);
                pl[j] = __builtin_convertvector(v - __builtin_convertvector(ph[j], f32x8), f16x8);
            }
            // small terms first: they meet the accumulator while it is small
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], ph[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], pl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], ph[j], acc[i][j], 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        par = (par + nk) & 1;
        // ---- epilogue: lane holds channels c..c+3 (c = tile channel base + (lane>>4)*4) of row (lane & 15)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const long long row = (long long)e_r0 + wn * (NJ * 16) + j * 16 + (lane & 15);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int c = e_n0 + wm * WROWS + i * 16 + (lane >> 4) * 4;
                *(float4 *)(e_y + row * Cout + c) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
        if (!more) break;
    }
}

// Round 5: the STAGGERED form (the default for the 256 x 256 tile).  In k_gemm_f16x3 every wave runs  [stage] -> [LDS reads +
// split] -> [MI * NJ * 3 matrix instructions] -> wait -> workgroup barrier  per K step, so the two waves of a SIMD reach their
// matrix work together and their reads together: the matrix pipe idles while both stage, read and split (45.7 % busy,
// profiles/r3_sq_counters_drn_split.txt).  In-kernel stamps (profiles/r5_gemm16_stagger_stamps.txt) put numbers on a wave's
// K step: issuing its 8 LDS-DMA pieces 430-690 cycles, reads + split 1 000-2 400 (2 400 beside a partner that multiplies:
// a matrix instruction holds the SIMD's vector issue for half its cycles and packed float32 operations are slow there), the 96
// matrix instructions 1 650, waits and barriers ~400 — one wave's chain is longer than the 3 072 matrix cycles of the SIMD's
// two waves, so nothing short of overlapping the two waves' phases helps.  Three changes, each measured:
//   1. HALF-PERIOD LAG.  A K step is two half periods  R = [stage, LDS reads, split of the first row fragment]  |  M = [matrix
//      instructions with the other fragments' splits between them (+ a tile's stores)]  with a barrier after each; the K steps
//      of ALL the tiles of a workgroup form one sequence q = 0, 1, ...; waves 4-7 run the same program HALF A PERIOD LATE (one
//      extra barrier at their start).  The second wave of every SIMD (a workgroup's waves go to the SIMDs in cyclic order, so
//      SIMD s holds waves s' and s' + 4: MI355X_MICROARCH.md, two waves per SIMD, item 9) multiplies while its partner stages,
//      reads and splits, and the other way round.  Alone (the split still in R) this was SLOWER, 3.97 against 3.70 ms on the
//      512 -> 512 layer: R beside a multiplying partner takes 2 000-2 400 cycles.
//   2. THE SPLIT INSIDE THE MATRIX WORK, as single-issue instructions: h = rn16(x * scale) and l = rn16(x * scale - h) are ONE
//      mixed-precision fma each (v_fma_mixlo/hi_f16: the product by a power of two and the difference are exact in float32, so
//      the only rounding is the conversion — the same bits as multiply, convert, convert back, subtract, convert), 16 vector
//      instructions per fragment instead of 24, none of them a packed float32 operation; M walks the row fragments j (per
//      accumulator still l.h, h.l, h.h) and splits fragment j + 1 between the matrix instructions of fragment j.  R shrinks to
//      ~750 cycles.  With 1.: 3.65 ms.
//   3. BOTH HALVES STAGE AT THE TOP OF THEIR PERIOD, i.e. while the partner multiplies, not in front of their own matrix work
//      (the late half's share of step q + 1 then has half a period to land instead of a whole one: enough).  With 1. and 2.:
//      3.42 ms = 1.02 PFLOP/s executed (0.41 of the dense 16-bit peak), 256 -> 256 1.13 against 1.22, 256 -> 512 2.09 against 2.26.
// Buffer safety: step q + 1 goes to buffer (q + 1) & 1, whose last readers are the R(q - 1) of both halves — the late half's
// ends at the barrier that ends the early half's period q - 1; the early waves stage at the top of their period q, the late
// waves at the top of theirs (half a period later); each waits for its own loads (the early half counted, so that a tile's
// stores issued behind them stay in flight) before the barrier that precedes the early half's R(q + 1).  Every accumulator
// receives the same matrix instructions in the same order as in k_gemm_f16x3 (K steps ascending; l.h, h.l, h.h inside a
// step): the outputs are bit-identical (tools/gemm16_ab.py compares digests across processes; tests/test_gpu_conv.py).
// RS: row fragments split in R (1-4, the others in M; 512 -> 512 layer: 3.48 / 3.41 / 3.35 / 3.37 ms for RS = 1 / 2 / 3 / 4: default 3).  XP (diagnostic builds, tools/gemm16_stamps.py): 2 = no split (timing
// only: the row tile read as if it held planes), 4 = no global loads (timing only), 16 = in-kernel stamps.
template <int BM, int BN, int RS = 1, int XP = 0>
__global__ __launch_bounds__(G16_THREADS) void k_gemm_f16x3_stag(const char *__restrict__ X, const char *__restrict__ Wt,
                                                                 float *__restrict__ Y, int rows_per_z, int Cin, int Cout,
                                                                 int ntiles, int total_tiles, int zcount, long long xz,
                                                                 long long wz, long long yz, const unsigned *__restrict__ amax,
                                                                 unsigned *__restrict__ dbg = nullptr)
{
    extern __shared__ __attribute__((aligned(1024))) char lds16[];   // [2] weight tiles | [2] row tiles, 128 bytes per row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = wave >= 4;
    // XP & 16: stamps (s_memtime, low word) of waves 0 and 4 of one workgroup over periods [Q0, Q0 + NQ), kept in LDS behind the
    // tiles (no global store inside the loop: the counted vmcnt waits stay what they are), copied out at the end
    constexpr int G16_Q0 = 24, G16_NQ = 40;
    unsigned *stamps = (unsigned *)(lds16 + 2 * (BM + BN) * 128) + (wave >> 2) * (G16_NQ * 8);
    const bool stamp_wave = (XP & 16) && blockIdx.x == 77 && (wave & 3) == 0;
    int stamp_q = -1;
    auto STAMP = [&](int k) {
        if ((XP & 16) && stamp_wave && stamp_q >= 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) stamps[stamp_q * 8 + k] = (unsigned)t;
        }
    };
    const int all_i = zcount * total_tiles, gstep = (int)gridDim.x;
    const int nwg = total_tiles;
    float sb;
    {
        const unsigned bits = *amax;
        int e = (int)(bits >> 23) - 127;
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
        sb = __uint_as_float((unsigned)(127 + 14 - (bits == 0u ? 0 : e)) << 23);
    }
    // a tile's position is wave-uniform: scalar registers
    int r0 = 0, n0 = 0, zz = 0;            // of the tile being staged
    float zscale_s = 0.f;
    auto locate = [&](int vid) {
        const int z = vid / total_tiles;
        const int zi = z / 6, zj = z - zi * 6;
        const int psum = ((0x433444 >> (4 * zi)) & 15) + ((0x433444 >> (4 * zj)) & 15);
        int id = vid - z * total_tiles;
        {
            const int q = nwg / 8, rem = nwg % 8, xcd = id % 8, idx = id / 8;
            id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
        }
        const int nt = id % ntiles, pt = id / ntiles;
        zz = __builtin_amdgcn_readfirstlane(z);
        r0 = __builtin_amdgcn_readfirstlane(pt * BN);
        n0 = __builtin_amdgcn_readfirstlane(nt * BM);
        zscale_s = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(sb * __uint_as_float((unsigned)(127 - psum) << 23))));
    };
    constexpr int WN = 4;
    constexpr int MI = BM == 256 ? 8 : 4;
    constexpr int NJ = BN / WN / 16;
    constexpr int WROWS = MI * 16;
    static_assert(BM / WROWS * WN == 8, "8 waves");
    static_assert(MI * NJ <= 63, "vmcnt range");
    static_assert(RS >= 1 && RS <= NJ && MI == 8, "split schedule: 4 element pairs over 8 groups of matrix instructions");
    constexpr bool LATE_PRESTAGE = (XP & 32) != 0;   // (XP & 32, SPA_GEMM16_LATE_PRESTAGE=1: round 6's experiment, no gain measured; off)
    constexpr bool EARLY_STORE = (XP & 8) != 0;   // (XP & 8, SPA_GEMM16_EARLY_STORE=1: round 6's experiment below; measured 7 % slower, not the default)

    char *wbuf = lds16, *xbuf = lds16 + 2 * (BM * 128);
    const int sub = lane >> 3, cs = lane & 7;
    const int chunk_byte = (cs ^ sub) << 4;
    const int nk = Cin / 32;
    // staging addresses = a scalar base (tile, K step, 64-row block) + ONE 32-bit lane offset shared by every load of the
    // kernel (both operands have rows of Cin * 4 bytes)
    const char *s_w = nullptr, *s_x = nullptr;          // staged tile's operand bases (uniform)
    const unsigned lane_off = (unsigned)((wave * 8 + sub) * Cin * 4 + chunk_byte);
    auto stage_bases = [&]() {
        s_w = Wt + ((long long)zz * wz + (long long)n0 * Cin) * 4;
        s_x = X + ((long long)zz * xz + (long long)r0 * Cin) * 4;
    };
    auto stage = [&](int t, int buf) {
        if (XP & 4) return;
        const char *wk = s_w + (long long)t * 128;
        char *dw = wbuf + buf * (BM * 128) + wave * 1024;
#pragma unroll
        for (int r = 0; r < BM / 64; ++r)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wk + (long long)r * 64 * Cin * 4 + lane_off),
                                             (__attribute__((address_space(3))) void *)(dw + r * 8192), 16, 0, 0);
        const char *xk = s_x + (long long)t * 128;
        char *dx = xbuf + buf * (BN * 128) + wave * 1024;
#pragma unroll
        for (int r = 0; r < BN / 64; ++r)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xk + (long long)r * 64 * Cin * 4 + lane_off),
                                             (__attribute__((address_space(3))) void *)(dx + r * 8192), 16, 0, 0);
    };
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fk = lane >> 4;
    if ((int)blockIdx.x >= all_i) return;
    const int my_tiles = (all_i - (int)blockIdx.x + gstep - 1) / gstep;
    const int S = my_tiles * nk;                                    // K steps of this workgroup

    // stage cursor (tile s_vid, step s_t, sequence number s_q); compute cursor (tile c_vid, step c_t)
    int s_vid = (int)blockIdx.x, s_t = 0, s_q = 0;
    locate(s_vid);
    stage_bases();
    int c_r0 = r0, c_n0 = n0, c_z = zz, c_t = 0, c_vid = s_vid;
    float c_zscale = zscale_s;
    auto stage_next = [&]() {                      // stage step s_q (if any) and move the cursor on
        if (s_q < S) {
            stage(s_t, s_q & 1);
            ++s_q;
            if (++s_t == nk) { s_t = 0; s_vid += gstep; if (s_vid < all_i) { locate(s_vid); stage_bases(); } }
        }
    };
    stage_next();                                  // step 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    g16_lds_barrier();
    if (late) g16_lds_barrier();                   // the extra barrier = half a period of delay
    bool stored = false, pre_staged = false, late_stored = false;

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < S; ++q) {
        const int buf = q & 1;
        if (XP & 16) stamp_q = (q >= G16_Q0 && q < G16_Q0 + G16_NQ) ? q - G16_Q0 : -1;
        STAMP(0);
        if (!pre_staged) stage_next();             // step q + 1 (the late half stages it in front of a tile's stores: below)
        pre_staged = false;
        STAMP(1);
        // ---- R: the LDS reads (weight fragments, raw float32 rows) and the split of the first RS row fragments
        const char *lw = wbuf + buf * (BM * 128), *lx = xbuf + buf * (BN * 128);
        f16x8 wh[MI], wl[MI];
        f32x4 ra[NJ], rb[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = wm * WROWS + i * 16 + frow;
            wh[i] = *(const f16x8 *)(lw + row * 128 + ((fk ^ (row & 7)) << 4));
            wl[i] = *(const f16x8 *)(lw + row * 128 + (((4 + fk) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = wn * (NJ * 16) + j * 16 + frow;
            ra[j] = *(const f32x4 *)(lx + row * 128 + (((2 * fk) ^ (row & 7)) << 4));
            rb[j] = *(const f32x4 *)(lx + row * 128 + (((2 * fk + 1) ^ (row & 7)) << 4));
        }
        const float sc = c_zscale;
        u32x4 hu[NJ], lu[NJ];
        // elements 2e, 2e + 1 of row fragment j: the two high halves, then (a dependent pair, issued a group of matrix
        // instructions later) the two low halves
        auto split_h = [&](int j, int e) {
            if (XP & 2) { hu[j][e] = __float_as_uint(e < 2 ? ra[j][2 * e] : rb[j][2 * e - 4]); return; }
            const float x0 = e < 2 ? ra[j][2 * e] : rb[j][2 * e - 4], x1 = e < 2 ? ra[j][2 * e + 1] : rb[j][2 * e - 3];
            unsigned h;
            asm volatile("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel_hi:[0,0,0]\n\t"
                         "v_fma_mixhi_f16 %0, %2, %3, 0 op_sel_hi:[0,0,0]"
                         : "=&v"(h) : "v"(x0), "v"(x1), "s"(sc));
            hu[j][e] = h;
        };
        auto split_l = [&](int j, int e) {
            if (XP & 2) { lu[j][e] = __float_as_uint(e < 2 ? ra[j][2 * e + 1] : rb[j][2 * e - 3]); return; }
            const float x0 = e < 2 ? ra[j][2 * e] : rb[j][2 * e - 4], x1 = e < 2 ? ra[j][2 * e + 1] : rb[j][2 * e - 3];
            unsigned l;
            asm volatile("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel_hi:[0,0,1]\n\t"
                         "v_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                         : "=&v"(l) : "v"(x0), "v"(x1), "s"(sc), "v"(hu[j][e]));
            lu[j][e] = l;
        };
#pragma unroll
        for (int j = 0; j < RS; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { split_h(j, e); split_l(j, e); }
        STAMP(2);
        // the late half's share of step q + 1 must have landed before the early half reads it behind this barrier.  Round 6 experiment
        // (LATE_PRESTAGE): a tile's 256 KB of stores per workgroup take ~10 k cycles to drain and this `vmcnt(0)` half a period behind
        // them holds the workgroup (stamps: periods of 15-25 k cycles at a tile's end against 4.3 k); with the late half staging this
        // share IN FRONT of its stores the wait can leave them in flight — measured: no gain (1.04 against 1.05 PFLOP/s): the stall only
        // moves to the next wait, a CU's memory pipe (1 MB of loads + 256 KB of stores per tile, ~18 B / cycle) is what the tile waits for
        if (late) {
            if (late_stored) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MI * NJ) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        late_stored = false;
        STAMP(3);
        g16_lds_barrier();
        STAMP(4);
        // ---- M: row fragment by row fragment, small terms first (they meet the accumulator while it is small); the split of
        // fragment j + RS rides between the matrix instructions of fragment j.
        // Round 6 experiment (EARLY_STORE): in-kernel stamps put the burst of 32 stores per wave behind a tile's last K step (every wave
        // of the workgroup within half a period of each other) at ~16 % of the kernel — periods that end a tile take 15-25 k cycles
        // against a median of 4.3 k (3 072 of them matrix work).  Storing (and clearing) every accumulator one accumulator behind its
        // final matrix instruction instead, between the matrix instructions of the last step, was measured 7 % SLOWER (1.07 -> 1.00
        // PFLOP/s executed, same bits): a store between matrix instructions waits for its accumulator and breaks the matrix stream
        // the way an LDS-DMA instruction does.  The burst stays.
        const bool last_step = c_t + 1 == nk;
        float *const e_y = Y + (long long)c_z * yz + ((long long)c_r0 + wn * (NJ * 16) + (lane & 15)) * Cout + (c_n0 + wm * WROWS + (lane >> 4) * 4);
        auto store_acc = [&](int i, int j) {
            *(float4 *)(e_y + (long long)(j * 16) * Cout + i * 16) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        };
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const f16x8 phj = __builtin_bit_cast(f16x8, hu[j]), plj = __builtin_bit_cast(f16x8, lu[j]);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], phj, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], plj, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], phj, acc[i][j], 0, 0, 0);
                if (j + RS < NJ) { if ((i & 1) == 0) split_h(j + RS, i >> 1); else split_l(j + RS, i >> 1); }
                if (EARLY_STORE && last_step && (i > 0 || j > 0)) { if (i > 0) store_acc(i - 1, j); else store_acc(MI - 1, j - 1); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        STAMP(5);
        if (++c_t == nk) {
            // ---- a tile is complete: lane holds channels c..c+3 (c = tile channel base + (lane>>4)*4) of row (lane & 15)
            if (late && LATE_PRESTAGE) {
                // (the buffer of step q + 2 = the buffer of step q, which both halves finished reading before this period's mid barrier)
                stage_next();
                pre_staged = true;
                late_stored = true;
            }
            if (EARLY_STORE) store_acc(MI - 1, NJ - 1);
            else {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) store_acc(i, j);
            }
            stored = true;
            c_t = 0; c_vid += gstep;
            if (c_vid < all_i) {
                // (the stage cursor is at most one step ahead: its tile is this one's successor or this one)
                const int sr0 = r0, sn0 = n0, sz = zz; const float szs = zscale_s;
                locate(c_vid);
                c_r0 = r0; c_n0 = n0; c_z = zz; c_zscale = zscale_s;
                r0 = sr0; n0 = sn0; zz = sz; zscale_s = szs;
            }
        }
        if (!late) {
            // the early half's share of step q + 1 (staged at the top of this period): a tile's stores issued after it may stay in flight
            if (stored) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MI * NJ) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP(6);
            g16_lds_barrier();
        } else if (q + 1 < S) {
            STAMP(6);
            g16_lds_barrier();
        }
        stored = false;
        STAMP(7);
    }
    if ((XP & 16) && stamp_wave && dbg) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = lane; i < G16_NQ * 8; i += 64) dbg[(wave >> 2) * (G16_NQ * 8) + i] = stamps[i];
    }
}

// Round 6: the PIPELINED form.  One barrier per K step and nothing in front of the matrix instructions: while step q multiplies out
// of registers, the same wave reads step q's weight fragments one channel block ahead (i-outer order: a block's two fragments die
// after its 12 matrix instructions), reads the raw float32 rows of step q + 1 and splits them into the planes of step q + 1 (a row
// fragment at a time: 8 transient registers), and stages weights(q + 1) and rows(q + 2) by LDS-DMA — all between matrix instructions,
// order pinned by scheduling barriers.  Buffers: two weight tiles (weights(q) are read during step q, weights(q + 1) land meanwhile)
// and two row tiles (rows(q + 1) are read during step q, rows(q + 2) land meanwhile in the buffer rows(q) left during step q - 1):
// the same 128 KB.  Registers: accumulators 128, planes of steps q and q + 1 32 + 32, weight fragments 8 + 8, raw rows 8.
// Every accumulator still receives l.h, h.l, h.h of K step after K step in ascending order: the same bits as k_gemm_f16x3(_stag).
template <int BM, int BN>
__global__ __launch_bounds__(G16_THREADS) void k_gemm_f16x3_pipe(const char *__restrict__ X, const char *__restrict__ Wt,
                                                                 float *__restrict__ Y, int rows_per_z, int Cin, int Cout,
                                                                 int ntiles, int total_tiles, int zcount, long long xz,
                                                                 long long wz, long long yz, const unsigned *__restrict__ amax)
{
    static_assert(BM == 256 && BN == 256, "one shape");
    extern __shared__ __attribute__((aligned(1024))) char lds16[];   // [2] weight tiles | [2] row tiles, 128 bytes per row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int all_i = zcount * total_tiles, gstep = (int)gridDim.x;
    const int nwg = total_tiles;
    float sb;
    {
        const unsigned bits = *amax;
        int e = (int)(bits >> 23) - 127;
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
        sb = __uint_as_float((unsigned)(127 + 14 - (bits == 0u ? 0 : e)) << 23);
    }
    struct Pos { int r0, n0, zz; float zscale; };
    auto locate = [&](int vid) {
        Pos p;
        const int z = vid / total_tiles;
        const int zi = z / 6, zj = z - zi * 6;
        const int psum = ((0x433444 >> (4 * zi)) & 15) + ((0x433444 >> (4 * zj)) & 15);
        int id = vid - z * total_tiles;
        {
            const int q = nwg / 8, rem = nwg % 8, xcd = id % 8, idx = id / 8;
            id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
        }
        const int nt = id % ntiles, pt = id / ntiles;
        p.zz = __builtin_amdgcn_readfirstlane(z);
        p.r0 = __builtin_amdgcn_readfirstlane(pt * BN);
        p.n0 = __builtin_amdgcn_readfirstlane(nt * BM);
        p.zscale = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(sb * __uint_as_float((unsigned)(127 - psum) << 23))));
        return p;
    };
    constexpr int WN = 4, MI = 8, NJ = 4, WROWS = MI * 16;
    char *const wbuf = lds16, *const xbuf = lds16 + 2 * (BM * 128);
    const int sub = lane >> 3, cs = lane & 7;
    const int chunk_byte = (cs ^ sub) << 4;
    const int nk = Cin / 32;
    const unsigned lane_off = (unsigned)((wave * 8 + sub) * Cin * 4 + chunk_byte);
    if ((int)blockIdx.x >= all_i) return;
    const int my_tiles = (all_i - (int)blockIdx.x + gstep - 1) / gstep;
    const int S = my_tiles * nk;                                    // K steps of this workgroup

    // cursors: the tile of the step being multiplied (cur), its successor (nxt); the weight stage cursor is one step ahead of the
    // multiplying one, the row stage cursor two, the split cursor one: each only ever sits in cur or nxt (nk >= 2)
    Pos cur = locate((int)blockIdx.x), nxt = cur;
    int nxt_vid = (int)blockIdx.x + gstep;
    if (nxt_vid < all_i) nxt = locate(nxt_vid);
    int c_t = 0;
    auto at = [&](int ahead, int &t_out) -> const Pos & {
        const int t = c_t + ahead;
        if (t < nk) { t_out = t; return cur; }
        t_out = t - nk;
        return nxt;
    };
    auto stage_w = [&](int ahead, int buf) {
        int t;
        const Pos &p = at(ahead, t);
        const char *wk = Wt + ((long long)p.zz * wz + (long long)p.n0 * Cin) * 4 + (long long)t * 128;
        char *dw = wbuf + buf * (BM * 128) + wave * 1024;
#pragma unroll
        for (int r = 0; r < BM / 64; ++r)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wk + (long long)r * 64 * Cin * 4 + lane_off),
                                             (__attribute__((address_space(3))) void *)(dw + r * 8192), 16, 0, 0);
    };
    auto stage_x = [&](int ahead, int buf) {
        int t;
        const Pos &p = at(ahead, t);
        const char *xk = X + ((long long)p.zz * xz + (long long)p.r0 * Cin) * 4 + (long long)t * 128;
        char *dx = xbuf + buf * (BN * 128) + wave * 1024;
#pragma unroll
        for (int r = 0; r < BN / 64; ++r)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xk + (long long)r * 64 * Cin * 4 + lane_off),
                                             (__attribute__((address_space(3))) void *)(dx + r * 8192), 16, 0, 0);
    };
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fk = lane >> 4;
    const int w_h = (wm * WROWS + frow) * 128 + ((fk ^ (frow & 7)) << 4), w_l = (wm * WROWS + frow) * 128 + (((4 + fk) ^ (frow & 7)) << 4);
    const int x_a = (wn * (NJ * 16) + frow) * 128 + (((2 * fk) ^ (frow & 7)) << 4), x_b = (wn * (NJ * 16) + frow) * 128 + (((2 * fk + 1) ^ (frow & 7)) << 4);

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto split_pair = [&](float x0, float x1, float sc, unsigned &l) {
        unsigned h, lo;
        asm("v_fma_mixlo_f16 %0, %2, %4, 0 op_sel_hi:[0,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %3, %4, 0 op_sel_hi:[0,0,0]\n\t"
            "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(h), "=&v"(lo) : "v"(x0), "v"(x1), "s"(sc));
        l = lo;
        return h;
    };

    // ---- prologue: weights(0), rows(0), rows(1); the planes of step 0
    stage_w(0, 0);
    stage_x(0, 0);
    if (S > 1) stage_x(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    g16_lds_barrier();
    u32x4 hu[NJ], lu[NJ], hn[NJ], ln[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const f32x4 a = *(const f32x4 *)(xbuf + j * 2048 + x_a), b = *(const f32x4 *)(xbuf + j * 2048 + x_b);
        unsigned l;
        hu[j][0] = split_pair(a[0], a[1], cur.zscale, l); lu[j][0] = l;
        hu[j][1] = split_pair(a[2], a[3], cur.zscale, l); lu[j][1] = l;
        hu[j][2] = split_pair(b[0], b[1], cur.zscale, l); lu[j][2] = l;
        hu[j][3] = split_pair(b[2], b[3], cur.zscale, l); lu[j][3] = l;
    }
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // (two steps per loop iteration with the two plane sets swapping roles: no register copies between steps)
    auto kstep = [&](int q, u32x4 (&hu)[NJ], u32x4 (&lu)[NJ], u32x4 (&hn)[NJ], u32x4 (&ln)[NJ]) {
        const bool s1 = q + 1 < S, s2 = q + 2 < S;
        const char *lw = wbuf + (q & 1) * (BM * 128);
        const char *lxn = xbuf + ((q + 1) & 1) * (BN * 128);              // rows of step q + 1
        int tn;
        const float scn = at(1, tn).zscale;                              // their scale
        f16x8 wh0, wl0, wh1, wl1;                                          // weight fragments of channel blocks i (even: 0, odd: 1)
        wh0 = *(const f16x8 *)(lw + w_h);
        wl0 = *(const f16x8 *)(lw + w_l);
        f32x4 ra, rb;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            // next block's weight fragments
            if (i + 1 < MI) {
                if (i & 1) { wh0 = *(const f16x8 *)(lw + (i + 1) * 2048 + w_h); wl0 = *(const f16x8 *)(lw + (i + 1) * 2048 + w_l); }
                else { wh1 = *(const f16x8 *)(lw + (i + 1) * 2048 + w_h); wl1 = *(const f16x8 *)(lw + (i + 1) * 2048 + w_l); }
            }
            // rows of step q + 1: fragment j = i / 2 is read at even i and split behind the matrix instructions of blocks i, i + 1
            if (s1 && (i & 1) == 0) { ra = *(const f32x4 *)(lxn + (i >> 1) * 2048 + x_a); rb = *(const f32x4 *)(lxn + (i >> 1) * 2048 + x_b); }
            const f16x8 wh = (i & 1) ? wh1 : wh0, wl = (i & 1) ? wl1 : wl0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f16x8 phj = __builtin_bit_cast(f16x8, hu[j]), plj = __builtin_bit_cast(f16x8, lu[j]);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, phj, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, plj, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, phj, acc[i][j], 0, 0, 0);
                if (s1 && j >= 2) {
                    // element pair e of row fragment jn = i / 2: pairs 0, 1 behind block i even, 2, 3 behind block i odd
                    const int jn = i >> 1, e = 2 * (i & 1) + (j - 2);
                    const float x0 = e < 2 ? ra[2 * e] : rb[2 * e - 4], x1 = e < 2 ? ra[2 * e + 1] : rb[2 * e - 3];
                    unsigned l;
                    hn[jn][e] = split_pair(x0, x1, scn, l);
                    ln[jn][e] = l;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // staging: weights of step q + 1 into the other weight tile, rows of step q + 2 into the tile rows(q) left a step ago
            if (i == 0) { if (s1) stage_w(1, (q + 1) & 1); __builtin_amdgcn_sched_barrier(0); }
            if (i == 1) { if (s2) stage_x(2, q & 1); __builtin_amdgcn_sched_barrier(0); }
        }
        bool stored = false;
        if (c_t == nk - 1) {
            // ---- a tile is complete: lane holds channels c..c+3 (c = tile channel base + (lane>>4)*4) of row (lane & 15)
            float *e_y = Y + (long long)cur.zz * yz;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const long long row = (long long)cur.r0 + wn * (NJ * 16) + j * 16 + (lane & 15);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int c = cur.n0 + wm * WROWS + i * 16 + (lane >> 4) * 4;
                    *(float4 *)(e_y + row * Cout + c) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                    acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            stored = true;
            c_t = 0;
            cur = nxt;
            nxt_vid += gstep;
            if (nxt_vid < all_i) nxt = locate(nxt_vid);
        } else ++c_t;
        if (s1) {
            // what this step staged has landed (a tile's stores, issued behind it, stay in flight); every wave has read the buffers of step q
            if (stored) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MI * NJ) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            g16_lds_barrier();
        }
    };
    for (int q = 0; q < S; q += 2) {
        kstep(q, hu, lu, hn, ln);
        if (q + 1 < S) kstep(q + 1, hn, ln, hu, lu);
    }
}

// zcount = 36 problems  y[z] (rows, Cout) float32 = (scale_z x[z]) (rows, Cin) . wt[z]^T, wt[z] (Cout, Cin): x float32, wt in
// the two-plane layout of the header (4 bytes per element); rows a multiple of 256, Cin a multiple of 32, Cout of 128;
// amax: device word, bit pattern of a bound on the largest magnitude of the layer input (the scale's exponent)
int gemm_f16x3_raw(spa_ctx *ctx, const float *x, long long rows, int32_t Cin, const void *wt, int32_t Cout, float *y,
                   void *stream, int zcount, const void *amax)
{
    SPA_ARG(ctx && x && wt && y && amax && rows > 0 && rows % 256 == 0 && rows < (1ll << 31) && zcount == 36);
    SPA_ARG(Cin % 32 == 0 && Cout % 128 == 0);
    SPA_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)wt % 16) == 0 && ((uintptr_t)y % 16) == 0);
    hipStream_t s = spa_stream(stream);
    static const int force_tile = getenv("SPA_GEMM16_TILE") ? atoi(getenv("SPA_GEMM16_TILE")) : 0;      // experiments: 128
    const int bm = (Cout % 256 == 0 && force_tile != 128) ? 256 : 128;
    // (SPA_GEMM16_TILE=2128, round 6 experiment: 256 channels x 128 rows, two LDS stages of 48 KB, <= 168 registers per wave —
    // room for a third wave per SIMD from another kernel, tools/coresidency_probe.py)
    const int bn = (bm == 256 && force_tile != 2128) ? 256 : 128;
    const int ntiles = Cout / bm;
    const long long total = rows / bn * ntiles;
    SPA_ARG(total < (1ll << 31));
    const size_t lds = 2 * (size_t)(bm + bn) * 128;
    if (!(ctx->gemm16_attr_done & 1)) {
        SPA_HIP(hipFuncSetAttribute((const void *)k_gemm_f16x3<256, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128));
        SPA_HIP(hipFuncSetAttribute((const void *)k_gemm_f16x3<128, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 128));
        SPA_HIP(hipFuncSetAttribute((const void *)k_gemm_f16x3<256, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 384 * 128));
        ctx->gemm16_attr_done |= 1;
    }
    SpaProfScope prof_(ctx, bm == 256 ? PROF_DRN_GEMM16 : PROF_DRN_GEMM16_N, s);
    static const int force_per_cu = getenv("SPA_GEMM16_PER_CU") ? atoi(getenv("SPA_GEMM16_PER_CU")) : 0;
    const int per_cu = force_per_cu > 0 ? force_per_cu : (lds > 80 * 1024 ? 1 : 2);
    long long grid = (long long)ctx->n_cu * per_cu;
    if (grid > total * zcount) grid = total * zcount;
    // SPA_GEMM16_STAGGER (read once): unset = the staggered kernel with three of the four row fragments split in R (the default for the 256 x 256
    // tile); 0 = k_gemm_f16x3 (round 3's kernel: kept for A/B runs and for the 128 x 128 tile); 1-4 = row fragments split in R;
    // diagnostic builds of the RS = 1 form (timing only unless stamps alone): + 8 no split, + 16 no global loads, + 32 in-kernel
    // stamps (tools/gemm16_stamps.py; RS = 1 or 2)
    static const int stagger = getenv("SPA_GEMM16_STAGGER") ? atoi(getenv("SPA_GEMM16_STAGGER")) : 3;
    static const int pipe = getenv("SPA_GEMM16_PIPE") ? atoi(getenv("SPA_GEMM16_PIPE")) : 0;
    if (pipe && bm == 256 && bn == 256 && Cin >= 64) {
        if (!(ctx->gemm16_attr_done & 2)) {
            SPA_HIP(hipFuncSetAttribute((const void *)k_gemm_f16x3_pipe<256, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128));
            ctx->gemm16_attr_done |= 2;
        }
        hipLaunchKernelGGL((k_gemm_f16x3_pipe<256, 256>), dim3((unsigned)grid), dim3(G16_THREADS), lds, s, (const char *)x, (const char *)wt, y,
                           (int)rows, Cin, Cout, ntiles, (int)total, zcount, rows * Cin, (long long)Cout * Cin, rows * Cout, (const unsigned *)amax);
        SPA_LAUNCH_CHECK();
        return SPA_OK;
    }
    if (bm == 256 && bn == 128) {
        hipLaunchKernelGGL((k_gemm_f16x3<256, 128>), dim3((unsigned)grid), dim3(G16_THREADS), lds, s, (const char *)x, (const char *)wt, y,
                           (int)rows, Cin, Cout, ntiles, (int)total, zcount, rows * Cin, (long long)Cout * Cin, rows * Cout, (const unsigned *)amax);
        SPA_LAUNCH_CHECK();
        return SPA_OK;
    }
    if (stagger && bm == 256) {
        if (!ctx->gemm16s_attr_done) {
#define G16S_ATTR(RS, XP) SPA_HIP(hipFuncSetAttribute((const void *)k_gemm_f16x3_stag<256, 256, RS, XP>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128 + ((XP) & 16 ? 4096 : 0)))
            G16S_ATTR(1, 0); G16S_ATTR(2, 0); G16S_ATTR(3, 0); G16S_ATTR(4, 0); G16S_ATTR(3, 8); G16S_ATTR(3, 32);
#ifdef SPA_DIAG
            G16S_ATTR(1, 2); G16S_ATTR(1, 4); G16S_ATTR(1, 6); G16S_ATTR(1, 16); G16S_ATTR(2, 16); G16S_ATTR(1, 18); G16S_ATTR(1, 20);
#endif
#undef G16S_ATTR
            ctx->gemm16s_attr_done = 1;
        }
        unsigned *dbg = nullptr;
        if (stagger & 32) {
            int rc = spa_ws_reserve(ctx, WS_DEBUG, 4096, (void **)&dbg);
            if (rc != SPA_OK) return rc;
        }
#define G16S_LAUNCH(RS, XP) hipLaunchKernelGGL((k_gemm_f16x3_stag<256, 256, RS, XP>), dim3((unsigned)grid), dim3(G16_THREADS), lds + ((XP) & 16 ? 4096 : 0), s, (const char *)x, (const char *)wt, y, \
                               (int)rows, Cin, Cout, ntiles, (int)total, zcount, rows * Cin, (long long)Cout * Cin, rows * Cout, (const unsigned *)amax, dbg)
        const int rs = stagger & 7, diag = stagger >> 3;          // diag: 1 no split, 2 no loads, 4 stamps
        static const int early_store = getenv("SPA_GEMM16_EARLY_STORE") ? atoi(getenv("SPA_GEMM16_EARLY_STORE")) : 0;
        static const int late_prestage = getenv("SPA_GEMM16_LATE_PRESTAGE") ? atoi(getenv("SPA_GEMM16_LATE_PRESTAGE")) : 0;
        if (diag == 0 && early_store) G16S_LAUNCH(3, 8);
        else if (diag == 0 && late_prestage) G16S_LAUNCH(3, 32);
        else if (diag == 0) { if (rs == 1) G16S_LAUNCH(1, 0); else if (rs == 2) G16S_LAUNCH(2, 0); else if (rs == 4) G16S_LAUNCH(4, 0); else G16S_LAUNCH(3, 0); }
#ifdef SPA_DIAG
        // timing-only forms (no split / no global loads: WRONG numbers) and the in-kernel stamps exist in diagnostic builds only
        // (make EXTRA=-DSPA_DIAG): a stray SPA_GEMM16_STAGGER cannot select them in the production library
        else if (diag == 1) G16S_LAUNCH(1, 2);
        else if (diag == 2) G16S_LAUNCH(1, 4);
        else if (diag == 3) G16S_LAUNCH(1, 6);
        else if (diag == 4) { if (rs == 2) G16S_LAUNCH(2, 16); else G16S_LAUNCH(1, 16); }
        else if (diag == 5) G16S_LAUNCH(1, 18);
        else G16S_LAUNCH(1, 20);
#else
        else SPA_ARG(!"SPA_GEMM16_STAGGER >= 8 selects a diagnostic form of the GEMM: build libspalign with EXTRA=-DSPA_DIAG");
#endif
#undef G16S_LAUNCH
        SPA_LAUNCH_CHECK();
        return SPA_OK;
    }
    if (bm == 256)
        hipLaunchKernelGGL((k_gemm_f16x3<256, 256>), dim3((unsigned)grid), dim3(G16_THREADS), lds, s, (const char *)x, (const char *)wt, y,
                           (int)rows, Cin, Cout, ntiles, (int)total, zcount, rows * Cin, (long long)Cout * Cin, rows * Cout, (const unsigned *)amax);
    else
        hipLaunchKernelGGL((k_gemm_f16x3<128, 128>), dim3((unsigned)grid), dim3(G16_THREADS), lds, s, (const char *)x, (const char *)wt, y,
                           (int)rows, Cin, Cout, ntiles, (int)total, zcount, rows * Cin, (long long)Cout * Cin, rows * Cout, (const unsigned *)amax);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
