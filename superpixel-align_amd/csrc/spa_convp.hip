// Round 6: the split-plane direct 3x3 convolution for the NARROW layers of the float32 DRN (64 and 128 output channels:
// layers 3 and 4 of models/drn.py:147-151, BasicBlock models/drn.py:23-57) with the planes built ONCE per staged pixel
// segment, in place in LDS.
//
// What k_conv3x3_f32<.., SPLIT> (spa_conv32.hip) pays on these layers (profiles/r5_sq_counters_drn_split.txt): a K step is
// one tap x 32 channels = 24-48 matrix instructions per wave, and around them ~650 other instructions — every wave re-reads
// its float32 pixel fragments for each of the three dx taps and splits them into the two half-precision planes again
// (16 vector instructions per fragment and tap), every K step recomputes its staging addresses and bounds tests on the
// scalar unit (4.3-5.7 scalar instructions per matrix instruction) and ends in `vmcnt(0)` + a workgroup barrier with its own
// loads only one K step old: matrix pipe 21-39 % busy.
//
// This kernel keeps the same arithmetic — every accumulator receives the same matrix instructions with the same operand
// bits in the same order (K order (dy, 32-channel step, dx); l.h, h.l, h.h inside a step), so the outputs are bit-identical
// to k_conv3x3_f32<.., SPLIT> (tools/convp_ab.py and tests/test_gpu_conv.py compare them) — and changes how the operands get there:
//   * a K GROUP = (dy, 32-channel step) = the three dx taps: one pixel segment of BN + 8 pixels, three weight tiles, 72
//     matrix instructions per wave and ONE workgroup barrier;
//   * the float32 pixel segment lands in LDS by global_load_lds during one group, is rewritten IN PLACE during the next —
//     every pair of 16-byte chunks (8 channels of a pixel) becomes [8 x h] [8 x l], sixteen mixed-precision fmas per pair,
//     1-2 pairs per thread and group, placed between matrix instructions — and is multiplied during the third (three
//     segment buffers).  A fragment read is then two 16-byte LDS reads and NO vector arithmetic.  Vector work per pixel
//     element: once per (tile, dy) instead of 3 taps x 1-2 wave rows;
//   * weights land during one group and are multiplied during the next (two buffers of three tap tiles); every load has a whole
//     group to arrive, the one wait per group is vmcnt(0) in front of the barrier (raw s_barrier: nothing else drains the queue);
//   * every LDS fragment read and the staging ride between matrix instructions of the previous tap (order pinned with
//     scheduling barriers); staging addresses = scalar cursors advanced by additions + per-lane offsets computed once per
//     kernel; the bias of a lane's channels in registers (Cout == BM); persistent workgroups on XCD-contiguous tile ranges as
//     before, the pipeline runs on across tiles (the located next tile is kept beside the current one).
// LDS: 2 x 3 x BM x 128 (weights) + 3 x (BN + 8) x 128 (segments) = 147 KB for both shapes (64 channels x 256 pixels,
// 128 channels x 128 pixels): one workgroup of 8 waves per CU.
// Measured (tools/convp_ab.py, 30 images, same box, k_conv3x3_f32<.., SPLIT> -> this kernel): 64 -> 64 at 256 x 512 1.32 -> 1.11 ms
// (with residual 1.50 -> 1.28), 128 -> 128 at 128 x 256 1.19 -> 0.99 ms (1.21 -> 1.04); matrix pipe 29-40 % -> 44-52 % busy.  What is
// left (in-kernel stamps, tools/convp_stamps.py, EXTRA=-DSPA_CP_STAMPS): a group is ~5 000 cycles for 2 304 cycles of matrix work per
// SIMD; tap 1's phase runs at the matrix rate (760 cycles), tap 2's phase with the 7-8 LDS-DMA instructions per wave in it takes
// 1 450-1 900 (an LDS-DMA instruction costs its wave ~100 cycles and both waves of a SIMD stage at the same time), tap 0's
// 840-1 650.  Fewer staged bytes per matrix instruction needs a 128 x 256 or 64 x 512 tile, which the three segment buffers
// do not leave LDS for.
#include "spa_common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CP_THREADS 512
#define CP_HALO 4

__device__ __forceinline__ void cp_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int N> __device__ __forceinline__ void cp_vmcnt()
{
    static_assert(N >= 0 && N <= 63, "vmcnt range");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int HAS_RES, int BM, int BN>
__global__ __launch_bounds__(CP_THREADS) void k_conv3x3_p16(const float *__restrict__ X, const char *__restrict__ Wt,
                                                            const float *__restrict__ bias, const float *__restrict__ R,
                                                            float *__restrict__ Y, const char *__restrict__ zero_line,
                                                            int H, int W, int Cin, int Cout, int dil, int relu, int xtiles,
                                                            int ntiles, int total_tiles, const unsigned *__restrict__ amax_in,
                                                            unsigned *__restrict__ amax_out, float inv_t, unsigned *__restrict__ dbg)
{
    constexpr int WN = BM == 64 ? 8 : 4;                 // waves along the pixels
    constexpr int WM = 8 / WN;
    constexpr int MI = BM / WM / 16;                     // 4
    constexpr int NJ = BN / WN / 16;                     // 2
    constexpr int WROWS = MI * 16;
    constexpr int SEGROWS = BN + 2 * CP_HALO;
    constexpr int NB = SEGROWS / 8;                      // 1 KB blocks of a segment: 33 / 17
    constexpr int SEGB = SEGROWS * 128;
    constexpr int WTAP = BM * 128;                       // one tap's weight tile
    constexpr int WGRP = 3 * WTAP;
    constexpr int NSL_HI = (NB - 1) / 8;                 // segment loads of every wave: 4 / 2 (+ one block more for wave 0: NB = 8 k + 1)
    static_assert(NB % 8 == 1, "one block beyond whole rounds: wave 0 takes it");
    constexpr int NPR = SEGROWS * 4;                     // pairs of 16-byte chunks (8 channels of a pixel) of a segment
    constexpr int NCV = (NPR + CP_THREADS - 1) / CP_THREADS;
    constexpr int NCVF = NPR / CP_THREADS;                // rounds in which every thread has a pair
    extern __shared__ __attribute__((aligned(1024))) char ldsp[];   // [2][3] weight tiles | [3] segments
    char *const wbuf = ldsp, *const sbuf = ldsp + 2 * WGRP;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane >> 3, cs = lane & 7;
    const int ks = Cin / 32, G = 3 * ks;
    const int row_bytes = Cin * 4;

    // ---- tiles (wave-uniform).  id -> XCD-contiguous order -> (channel tile, x tile, image row)
    struct Tile { int row_id, y, x0, n0; };
    auto locate = [&](int vid) {
        Tile t;
        const int q = total_tiles / 8, rem = total_tiles % 8, xcd = vid % 8, idx = vid / 8;
        const int id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
        const int nt = id % ntiles, pt = id / ntiles;
        const int xt = pt % xtiles;
        t.row_id = __builtin_amdgcn_readfirstlane(pt / xtiles);
        t.y = t.row_id % H;
        t.x0 = __builtin_amdgcn_readfirstlane(xt * BN);
        t.n0 = __builtin_amdgcn_readfirstlane(nt * BM);
        return t;
    };
    if ((int)blockIdx.x >= total_tiles) return;
    const int gstep = (int)gridDim.x;
    const int my_tiles = (total_tiles - (int)blockIdx.x + gstep - 1) / gstep;
    const int NG = my_tiles * G;                          // groups of this workgroup
    Tile cur = locate((int)blockIdx.x), nxt = cur;
#ifdef SPA_CP_STAMPS
    // diagnostic build (tools/convp_stamps.py): s_memtime of waves 0 and 4 (the two waves of one SIMD) of one workgroup at ten points
    // of groups [CP_Q0, CP_Q0 + CP_NQ), kept in LDS behind the buffers, copied out at the end
    constexpr int CP_Q0 = 30, CP_NQ = 48;
    unsigned *const stamps = (unsigned *)(ldsp + 2 * WGRP + 3 * SEGB) + (wave >> 2) * (CP_NQ * 10);
    const bool stamp_wave = blockIdx.x == 77 && (wave & 3) == 0;
    int stamp_q = -1;
#define CP_STAMP(k) do { if (stamp_wave && stamp_q >= 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) stamps[stamp_q * 10 + (k)] = (unsigned)t_; } } while (0)
#else
#define CP_STAMP(k) do { } while (0)
#endif
    int nxt_vid = (int)blockIdx.x + gstep;
    if (nxt_vid < total_tiles) nxt = locate(nxt_vid);

    // ---- staging: per-lane parts once, scalar cursors advanced group by group (no division, no multiplication in the loop)
    // weights: LDS row (blk * 8 + sub) of a tap tile holds chunk c of its 128 bytes at position c ^ (row & 7)
    const unsigned w_lane = (unsigned)(wave * 8 + sub) * (unsigned)(9 * row_bytes) + (unsigned)((cs ^ sub) << 4);
    const unsigned w_rstep = 64u * (unsigned)(9 * row_bytes);
    // pixels: chunk c of segment row r at position c ^ g(r & 7), g = 0 0 1 1 4 4 5 5 (spa_conv32.hip: conflict-free fragment reads)
    auto xg = [](int r) { return ((r >> 1) & 1) | (((r >> 2) & 1) << 2); };
    const int x_chunk = (cs ^ xg(sub)) << 4;
    const unsigned x_lane = (unsigned)(wave * 8 + sub) * (unsigned)row_bytes + (unsigned)x_chunk;
    const int x_lane_px = wave * 8 + sub;                  // pixel of this lane in round 0, relative to x0 - HALO
    // weight cursor: Cout == BM, so the weight tiles of a group do not depend on the tile — byte offset of (dy, step) in a row of Wt
    int w_dy = 0, w_kc = 0, w_off = 0;
    auto stage_w = [&](int buf) {
        const char *wk = Wt + w_off;
        char *dst = wbuf + buf * WGRP + wave * 1024;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int r = 0; r < BM / 64; ++r)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wk + dx * row_bytes + (w_lane + r * w_rstep)),
                                                 (__attribute__((address_space(3))) void *)(dst + dx * WTAP + r * 8192), 16, 0, 0);
        if (++w_kc < ks) w_off += 128;
        else {
            w_kc = 0;
            if (++w_dy < 3) w_off += 3 * row_bytes - (ks - 1) * 128;
            else { w_dy = 0; w_off = 0; }
        }
    };
    // pixel cursor: the tile it is in, (dy, step), the address of pixel x0 - HALO of that input row at that channel step (may lie
    // before the row: only lanes inside the image dereference), whether the row is inside the image
    Tile xt = cur;
    int x_dy = 0, x_kc = 0;
    const long long dy_step = (long long)dil * W * row_bytes - (long long)(ks - 1) * 128;
    auto x_origin = [&](const Tile &t) { return (const char *)X + ((long long)(t.row_id - dil) * W + (t.x0 - CP_HALO)) * row_bytes; };
    const char *x_ptr = x_origin(xt);
    bool x_yok = xt.y - dil >= 0;
    auto stage_x = [&](int buf) {
        char *dst = sbuf + buf * SEGB + wave * 1024;
        const int pbase = xt.x0 - CP_HALO + x_lane_px;
        // (block NB - 1 belongs to wave 0)
        if (wave == 0) {
            const bool ok = x_yok && (unsigned)(pbase + NSL_HI * 64) < (unsigned)W;
            const char *src = ok ? x_ptr + (x_lane + (unsigned)NSL_HI * 64u * (unsigned)row_bytes) : zero_line + x_chunk;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + NSL_HI * 8192), 16, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < NSL_HI; ++r) {
            const bool ok = x_yok && (unsigned)(pbase + r * 64) < (unsigned)W;
            const char *src = ok ? x_ptr + (x_lane + (unsigned)r * 64u * (unsigned)row_bytes) : zero_line + x_chunk;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + r * 8192), 16, 0, 0);
        }
        if (++x_kc < ks) x_ptr += 128;
        else {
            x_kc = 0;
            if (++x_dy < 3) x_ptr += dy_step;
            else { x_dy = 0; xt = nxt; x_ptr = x_origin(xt); }       // (two groups ahead of the multiplying cursor: nxt is its next tile)
            const int yy = xt.y + (x_dy - 1) * dil;
            x_yok = yy >= 0 && yy < H;
        }
    };
    int c_g = 0;                                            // group of the current tile being multiplied

    // ---- scale of the pixels / of the result (spa_conv32.hip)
    float sc16, unscale;
    {
        const unsigned bits = *amax_in;
        int e = (int)(bits >> 23) - 127;
        e = bits == 0u ? 0 : (e < -100 ? -100 : (e > 100 ? 100 : e));
        sc16 = __uint_as_float((unsigned)(127 + 14 - e) << 23);
        unscale = __uint_as_float((unsigned)(127 - 14 + e) << 23) * inv_t;
    }
    // in-place planes of segment buffer `buf`: pair k of this thread's share = the two 16-byte chunks of 8 consecutive channels of
    // a pixel (float32) -> [8 x h] at the first chunk's place, [8 x l] at the second's
    const int cv_row = tid >> 2, cv_m = tid & 3;              // pair (row, m) of round 0; a round is 128 rows further
    const int cv_off = cv_row * 128 + (((2 * cv_m) ^ xg(cv_row & 7)) << 4);      // (128 rows further: the same row & 7)
    auto convert = [&](int buf, int k) {
        if (NPR % CP_THREADS != 0 && k == NCV - 1 && tid + k * CP_THREADS >= NPR) return;
        char *pa = sbuf + buf * SEGB + k * (128 * 128) + cv_off;
        const u32x4 va = *(const u32x4 *)pa, vb = *(const u32x4 *)(pa + 16 - 32 * ((cv_off >> 4) & 1));
        u32x4 h, l;
        { unsigned t; h[0] = spa_split16_pair(__uint_as_float(va[0]), __uint_as_float(va[1]), sc16, t); l[0] = t; }
        { unsigned t; h[1] = spa_split16_pair(__uint_as_float(va[2]), __uint_as_float(va[3]), sc16, t); l[1] = t; }
        { unsigned t; h[2] = spa_split16_pair(__uint_as_float(vb[0]), __uint_as_float(vb[1]), sc16, t); l[2] = t; }
        { unsigned t; h[3] = spa_split16_pair(__uint_as_float(vb[2]), __uint_as_float(vb[3]), sc16, t); l[3] = t; }
        *(u32x4 *)pa = h;
        *(u32x4 *)(pa + 16 - 32 * ((cv_off >> 4) & 1)) = l;
    };

    // ---- fragment addresses
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fk = lane >> 4;
    const int wfrag_h = (wm * WROWS + frow) * 128 + ((fk ^ (frow & 7)) << 4);
    const int wfrag_l = (wm * WROWS + frow) * 128 + (((4 + fk) ^ (frow & 7)) << 4);
    int pfrag[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int row = CP_HALO + (dx - 1) * dil + wn * (NJ * 16) + frow;
        pfrag[dx] = row * 128 + (((2 * fk) ^ xg(row & 7)) << 4);          // chunk 2 fk; chunk 2 fk + 1 is at ^ 16
    }

    // ---- prologue (Cout == BM: one channel tile, so the bias of a lane's channels is loaded once)
    float4 bv[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) bv[i] = *(const float4 *)(bias + wm * WROWS + i * 16 + (lane >> 4) * 4);
    // groups 0 and 1 whole, the pixels of group 2 (the cursors only move pointers past the end of the sequence: NG >= 6)
    stage_x(0);
    stage_w(0);
    stage_x(1);
    stage_w(1);
    stage_x(2);
    cp_vmcnt<0>();
    cp_barrier();
#pragma unroll
    for (int k = 0; k < NCV; ++k) convert(0, k);
    cp_barrier();

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned amx = 0;
    int sb_cur = 0;                       // segment buffer of group q (q % 3)

    // ---- the group loop.  Fragments of one tap = 12 x 16 bytes per lane (8 weight, 4 pixel), its 24 matrix instructions in the order
    // l.h (all accumulators), h.l, h.h.  In-kernel stamps (tools/convp_stamps.py) of the first forms of this kernel: with the fragment
    // reads of a tap and the staging in FRONT of the tap's matrix instructions and two barriers per group (pixels landed | converted),
    // a group took 5 900 cycles for 2 304 cycles of matrix work per SIMD — every wave is in the same phase, so the matrix pipe idles
    // while all of them read (the LDS needs ~1 200 cycles per group for the fragments alone), stage (an LDS-DMA instruction costs its
    // wave ~100 cycles) or wait.  Hence: all LDS traffic and the staging ride BETWEEN matrix instructions, and ONE barrier per group:
    //   B   matrix instructions of tap 0 | reads of tap 1; first half of the conversion of pixels(q + 1)
    //   C   matrix instructions of tap 1 | reads of tap 2; second half of the conversion
    //   X   everything staged a group ago has landed (vmcnt(0): weights(q + 1), pixels(q + 2), a tile's stores), barrier: conversion
    //       visible, every fragment of group q is in registers, so the buffers of group q are free
    //   A'  matrix instructions of tap 2 | reads of tap 0 of group q + 1, staging of weights(q + 2) and pixels(q + 3) into the
    //       buffers group q just left; the epilogue when group q ended a tile
    // A segment thus lands during one group, is converted during the next and multiplied during the third (three buffers), weights
    // land during one group and are multiplied during the next (two buffers), and every load has a whole group to arrive.
    struct Frags { f16x8 wh[MI], wl[MI], ph[NJ], pl[NJ]; };
    auto read_one = [&](Frags &f, const char *lw, const char *lx, int dx, int idx) {      // idx 0..11 (compile-time after unrolling)
        if (idx < 2 * MI) {
            const int i = idx >> 1;
            if (idx & 1) f.wl[i] = *(const f16x8 *)(lw + dx * WTAP + i * 2048 + wfrag_l);
            else f.wh[i] = *(const f16x8 *)(lw + dx * WTAP + i * 2048 + wfrag_h);
        } else {
            const int j = (idx - 2 * MI) >> 1;
            if (idx & 1) f.pl[j] = *(const f16x8 *)(lx + j * 2048 + (pfrag[dx] ^ 16));
            else f.ph[j] = *(const f16x8 *)(lx + j * 2048 + pfrag[dx]);
        }
    };
    auto mfma_one = [&](const Frags &f, int t) {            // matrix instruction t = 0..23 of a tap
        const int pass = t / (MI * NJ), r = t % (MI * NJ), i = r / NJ, j = r % NJ;
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pass == 0 ? f.wl[i] : f.wh[i], pass == 1 ? f.pl[j] : f.ph[j], acc[i][j], 0, 0, 0);
    };
    constexpr int NM = 3 * MI * NJ, NR = 2 * MI + 2 * NJ;      // 24 matrix instructions, 12 reads per tap
    static_assert(2 * NR <= NM, "one read behind each matrix instruction");
    constexpr int NPIECE = 4 * NCVF, NP1 = NPIECE / 2;           // splits of a float32 pair per thread and group; those done in B
    static_assert(NPIECE % 8 == 0 || NPIECE == 4, "whole pairs of chunks per half");
    Frags F0, F1, F2;
#pragma unroll
    for (int i = 0; i < NR; ++i) read_one(F0, wbuf, sbuf, 0, i);      // tap 0 of group 0

    for (int q = 0; q < NG; ++q) {
        const bool iw = q + 1 < NG, i2 = q + 2 < NG, i3 = q + 3 < NG;
#ifdef SPA_CP_STAMPS
        stamp_q = (q >= CP_Q0 && q < CP_Q0 + CP_NQ) ? q - CP_Q0 : -1;
#endif
        CP_STAMP(0);
        const int sb_n1 = sb_cur == 2 ? 0 : sb_cur + 1;
        const bool tile_end = c_g == G - 1;
        const char *lw = wbuf + (q & 1) * WGRP, *lx = sbuf + sb_cur * SEGB;
        // conversion of pixels(q + 1): the full rounds of pairs of chunks here, one split of a float32 pair = 4 single-issue vector
        // instructions behind a matrix instruction, the two LDS writes of a pair of chunks once its four splits are through (every
        // thread; in the last group of the workgroup they rewrite a dead buffer); wave 0's 32 leftover pairs at the end of C
        char *pa[NCVF], *pb[NCVF];
        u32x4 va[NCVF], vb[NCVF], hv[NCVF], lv[NCVF];
        auto conv_read = [&](int k) {
            pa[k] = sbuf + sb_n1 * SEGB + k * (128 * 128) + cv_off;
            pb[k] = pa[k] + 16 - 32 * ((cv_off >> 4) & 1);
            va[k] = *(const u32x4 *)pa[k];
            vb[k] = *(const u32x4 *)pb[k];
        };
        auto piece = [&](int pc) {
            const int k = pc / 4, e = pc % 4;
            const float x0 = __uint_as_float(e < 2 ? va[k][2 * e] : vb[k][2 * e - 4]), x1 = __uint_as_float(e < 2 ? va[k][2 * e + 1] : vb[k][2 * e - 3]);
            unsigned lo;
            hv[k][e] = spa_split16_pair(x0, x1, sc16, lo);
            lv[k][e] = lo;
            if (e == 3) { *(u32x4 *)pa[k] = hv[k]; *(u32x4 *)pb[k] = lv[k]; }
        };
        // ---- B
#pragma unroll
        for (int t = 0; t < NM; ++t) {
            mfma_one(F0, t);
            if (t < NR) read_one(F1, lw, lx, 1, t);
            if (t == NM - NP1 - 6) {
#pragma unroll
                for (int k = 0; k < NCVF; ++k) conv_read(k);
            }
            if (t >= NM - NP1) piece(t - (NM - NP1));
            __builtin_amdgcn_sched_barrier(0);
        }
        CP_STAMP(1);
        // ---- C
#pragma unroll
        for (int t = 0; t < NM; ++t) {
            mfma_one(F1, t);
            if (t >= NM - NR) read_one(F2, lw, lx, 2, t - (NM - NR));
            if (t >= 2 && t - 2 < NPIECE - NP1) piece(NP1 + t - 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        CP_STAMP(2);
        if (NPR % CP_THREADS != 0 && wave == 0) convert(sb_n1, NCVF);
        CP_STAMP(3);
        // ---- X
        cp_vmcnt<0>();
        CP_STAMP(4);
        cp_barrier();
        CP_STAMP(5);
        // ---- A'
        // residual of a finishing tile: requested a tap early (held any longer, its 32 registers spill), consumed in the epilogue below
        float4 res[MI][NJ];
        if (HAS_RES && tile_end) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                int xx = cur.x0 + wn * (NJ * 16) + j * 16 + (lane & 15);
                xx = xx < W ? xx : W - 1;
                const long long pix = (long long)cur.row_id * W + xx;
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    res[i][j] = *(const float4 *)(R + pix * Cout + (cur.n0 + wm * WROWS + i * 16 + (lane >> 4) * 4));
            }
        }
        {
            const char *lwn = wbuf + ((q + 1) & 1) * WGRP, *lxn = sbuf + sb_n1 * SEGB;
#pragma unroll
            for (int t = 0; t < NM; ++t) {
                mfma_one(F2, t);
                if (t < NR && iw) read_one(F0, lwn, lxn, 0, t);
                __builtin_amdgcn_sched_barrier(0);
                if (t == 13) { if (i2) stage_w(q & 1); __builtin_amdgcn_sched_barrier(0); }
                if (t == 17) { if (i3) stage_x(sb_cur); __builtin_amdgcn_sched_barrier(0); }
            }
        }
        CP_STAMP(6);
        if (tile_end) {
            // ---- epilogue: lane holds channels c..c+3 (c = tile channel base + (lane>>4)*4) of pixel (lane & 15)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int xx = cur.x0 + wn * (NJ * 16) + j * 16 + (lane & 15);
                const long long pix = (long long)cur.row_id * W + xx;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int c = cur.n0 + wm * WROWS + i * 16 + (lane >> 4) * 4;
                    float v0 = acc[i][j][0] * unscale + bv[i].x, v1 = acc[i][j][1] * unscale + bv[i].y;
                    float v2 = acc[i][j][2] * unscale + bv[i].z, v3 = acc[i][j][3] * unscale + bv[i].w;
                    acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (HAS_RES) { v0 += res[i][j].x; v1 += res[i][j].y; v2 += res[i][j].z; v3 += res[i][j].w; }
                    if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                    if (xx < W) {
                        *(float4 *)(Y + pix * Cout + c) = make_float4(v0, v1, v2, v3);
                        amx = max(max(amx, __float_as_uint(v0) & 0x7fffffffu), max(__float_as_uint(v1) & 0x7fffffffu,
                                  max(__float_as_uint(v2) & 0x7fffffffu, __float_as_uint(v3) & 0x7fffffffu)));
                    }
                }
            }
            c_g = 0;
            cur = nxt;
            nxt_vid += gstep;
            if (nxt_vid < total_tiles) nxt = locate(nxt_vid);
        } else ++c_g;
        CP_STAMP(7);
        CP_STAMP(8);
        CP_STAMP(9);
        sb_cur = sb_n1;
    }
#ifdef SPA_CP_STAMPS
    if (stamp_wave && dbg) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = lane; i < CP_NQ * 10; i += 64) dbg[(wave >> 2) * (CP_NQ * 10) + i] = stamps[i];
    }
#endif
    if (amax_out) {
        for (int o = 32; o > 0; o >>= 1) amx = max(amx, (unsigned)__shfl_xor((int)amx, o));
        if (lane == 0 && amx > *(volatile unsigned *)amax_out) atomicMax(amax_out, amx);
    }
}

// the launcher behind spa_conv3x3_f16s for Cout % 64 == 0, Cout % 256 != 0 (spa_conv32.hip decides); arguments as there
int conv3x3_p16_launch(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin, const void *wt2, float inv_t,
                       int32_t Cout, const float *bias, const float *residual, int32_t relu, int32_t dilation,
                       const void *amax_in, void *amax_out, float *y, const char *zero, hipStream_t s)
{
    SPA_ARG((Cout == 64 || Cout == 128) && dilation >= 1 && dilation <= CP_HALO && Cin % 32 == 0 && Cin >= 64);
    const int bm = Cout, bn = bm == 128 ? 128 : 256;
    const int xtiles = (W + bn - 1) / bn, ntiles = Cout / bm;
    const long long total = (long long)B * H * xtiles * ntiles;
    SPA_ARG(total < (1ll << 31));
    SPA_ARG((long long)(2 * W + 1024) * Cin * 4 < (1ll << 31));
    const size_t lds = 2 * 3 * (size_t)bm * 128 + 3 * (size_t)(bn + 2 * CP_HALO) * 128;
    if (!(ctx->conv32_attr_done & 8)) {
        SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_p16<0, 64, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds + 4096));
        SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_p16<1, 64, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds + 4096));
        SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_p16<0, 128, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds + 4096));
        SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_p16<1, 128, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds + 4096));
        ctx->conv32_attr_done |= 8;
    }
    long long grid = ctx->n_cu;
    if (grid > total) grid = total;
    unsigned *dbg = nullptr;
#ifdef SPA_CP_STAMPS
    { int rc = spa_ws_reserve(ctx, WS_DEBUG, 4096, (void **)&dbg); if (rc != SPA_OK) return rc; }
    const size_t lds_x = 4096;
#else
    const size_t lds_x = 0;
#endif
#define CP_LAUNCH(R, M, N)                                                                                                     \
    hipLaunchKernelGGL((k_conv3x3_p16<R, M, N>), dim3((unsigned)grid), dim3(CP_THREADS), lds + lds_x, s, x, (const char *)wt2, bias, residual, y, \
                       zero, H, W, Cin, Cout, dilation, relu, xtiles, ntiles, (int)total, (const unsigned *)amax_in,           \
                       (unsigned *)amax_out, inv_t, dbg)
    if (bm == 64) { if (residual) CP_LAUNCH(1, 64, 256); else CP_LAUNCH(0, 64, 256); }
    else { if (residual) CP_LAUNCH(1, 128, 128); else CP_LAUNCH(0, 128, 128); }
#undef CP_LAUNCH
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
