// bfloat16 implicit-GEMM 3x3 (dilated) convolution for the heavy layers of the DRN (models/drn.py:230-285:
// layers 5-8, 256/512 channels at 1/8 resolution: ~80 % of the network's FLOPs), with the bias left by the
// folded BatchNorm, the residual add of a BasicBlock and the ReLU fused into the epilogue.
//
//   Y[b, y, x, n] = relu?( bias[n] + res[b, y, x, n] + sum_{tap, c} X[b, y + dy(tap)*d, x + dx(tap)*d, c] * Wt[n, tap, c] )
//
// channels-last activations (the layout the pooling kernels want anyway), weights repacked once to
// (Cout, 9, Cin), float32 accumulation on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16).
//
// Mapping: GEMM with M = Cout (weights are the MFMA A operand), N = pixels, K = 9 * Cin.  A workgroup
// (8 waves) owns 256 output channels x 256 consecutive pixels of one image row.  The K loop runs in the order
// (dy, 64 input channels, dx): the three dx taps of one (dy, channel step) read the SAME input pixels shifted by
// the dilation, so ONE row segment of 256 + 2*4 pixels x 64 channels is staged per (dy, channel step) — a third
// of it with each of the previous group's three K steps — and the taps read it at a row offset; the weight tile
// [256 channels][64 k] is staged per K step.  Everything goes global -> LDS by 16-byte global_load_lds (pixels
// outside the image read a zero line); LDS rows are 128 bytes with the 16-byte chunks XOR-swizzled by (row & 7)
// on the SOURCE address (the LDS image of a wave's load is lane-linear), which keeps the ds_read_b128 fragment
// reads of the 16x16x32 MFMA conflict-free at any row offset.  Two buffers of each: the loads of K step t+1 are
// issued before the MFMAs of step t.  The accumulator tile of a lane is four consecutive output channels of one
// pixel, so the epilogue stores 8 contiguous bytes per lane and tile.
//
// Measured (tools/conv_bench.py: 30 x 128 x 256 pixels, random operands; MIOpen's kernel alone 1 044-1 090):
//   512 -> 512: 1 126-1 163 TFLOP/s (0.45-0.47 of the nominal dense bf16 peak), 256 -> 512: 1 050, 256 -> 256: 919.
// What was tried around it, each built, verified and measured on the 512 -> 512 layer:
//   * one pixel tile per tap (9 x 32 KB instead of 3 x 33 KB per channel step; round 1): 1 064-1 091;
//   * four LDS buffers in half steps of 32 channels (three half tiles in flight, counted vmcnt): 1 050 — twice
//     the barriers cost more than the extra loads in flight bought;
//   * loads interleaved between the MFMA groups: 926 (every global_load_lds re-programs M0);
//   * four waves of 128 x 128 (256 accumulator registers, one wave per SIMD, a third less LDS traffic per FLOP)
//     with two fragment register sets and the step's barrier between its two half steps, so that no MFMA waits
//     for an LDS round trip: 985-1 041, with the row-segment staging 1 016-1 087; the same pipeline with eight
//     waves spills (224 registers of accumulators + fragments): 910-925;
//   * that four-wave loop with the global loads compiled out (timing only): 1 220-1 300 without the pixel loads,
//     1 115 without the weight loads — an LDS-read + MFMA loop on random operands tops out near 0.5 of the nominal
//     peak (the chip lowers its clock under dense bf16 MFMA on random data: MI355X_MICROARCH.md, DVFS give-back),
//     so the kernel is at ~0.9 of what its loop can deliver.
#include "spa_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// output channels per workgroup: template parameter BM = 256 (2 x 4 waves of 128 x 64), 128 (2 x 4 waves of
// 64 x 64) or 64 (1 x 8 waves of 64 x 32) — the narrow tiles serve the 64/128-channel layers, which are bound by
// their activations' HBM traffic, not by the matrix pipe
#define CV_BN 256            // pixels per workgroup
#define CV_BK 64             // K step (input channels of one tap)
#define CV_THREADS 512
#define CV_TILE_BYTES (256 * CV_BK * 2)          // one operand tile: 32 KB
#define CV_HALO 4                                // pixels either side of the 256-pixel segment: dilation <= 4
#define CV_XSEG_BYTES ((256 + 2 * CV_HALO) * CV_BK * 2)   // 33 row blocks of 8 pixels

__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f)
{
    __bf16 h = (__bf16)f;                          // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    return __builtin_bit_cast(unsigned short, h);
}

template <int HAS_RES, int BM>
__global__ __launch_bounds__(CV_THREADS) void k_conv3x3_bf16(const __bf16 *__restrict__ X, const __bf16 *__restrict__ Wt,
                                                             const float *__restrict__ bias,
                                                             const __bf16 *__restrict__ R, __bf16 *__restrict__ Y,
                                                             const char *__restrict__ zero_line, int B, int H, int W,
                                                             int Cin, int Cout, int dil, int relu, int xtiles,
                                                             int ntiles, int total_tiles)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];     // [2] weight tiles 32 KB | [2] pixel segments 33 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order: consecutive workgroup ids go round-robin over the 8 XCDs; give every XCD a
    // contiguous range of tiles (neighbouring rows of one image share two of their three input rows in L2)
    const int nwg = total_tiles;
    int id = blockIdx.x;
    {
        const int q = nwg / 8, rem = nwg % 8, xcd = id % 8, idx = id / 8;
        id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    // tile id -> (pixel tile, channel tile): the channel tiles of one pixel tile are adjacent
    const int nt = id % ntiles, pt = id / ntiles;
    const int xt = pt % xtiles, row_id = pt / xtiles;               // row_id = b * H + y
    const int y = row_id % H;
    const int x0 = xt * CV_BN, n0 = nt * BM;
    constexpr int WN = BM == 64 ? 8 : 4;                 // waves along the pixels
    constexpr int MI = BM == 256 ? 8 : 4;                // 16-channel MFMA tiles per wave
    constexpr int NJ = 256 / WN / 16;                    // 16-pixel MFMA tiles per wave
    constexpr int WROWS = MI * 16;                       // channels per wave

    // K order: (dy, 64-channel step, dx).  The three dx taps of one (dy, k step) read the SAME input pixels
    // shifted by the dilation: one row segment of 256 + 2*CV_HALO pixels is staged per (dy, k step) — a third
    // of it with each of the previous group's three K steps — and the taps read it at a row offset.  A K step
    // thus moves 32 KB of weights + 11 KB of pixels instead of 32 + 32.
    char *wbuf = lds, *xbuf = lds + 2 * CV_TILE_BYTES;
    const int sub = lane >> 3, cs = lane & 7;
    const int chunk_byte = (cs ^ sub) << 4;        // staged row = block * 8 + sub: (row & 7) = sub for every block
    const char *wbase = (const char *)(Wt + (long long)n0 * 9 * Cin);
    const char *xbase = (const char *)(X + (long long)row_id * W * Cin);      // input row y, pixel 0
    const int ks = Cin / CV_BK;
    const int nk = 9 * ks, ngroups = 3 * ks;

    auto stage_w = [&](int t, int buf) {
        const int g = t / 3, dxi = t - g * 3;
        const int dyi = g / ks, kc = g - dyi * ks;
        const char *wk = wbase + ((long long)(dyi * 3 + dxi) * Cin + (long long)kc * CV_BK) * 2 + chunk_byte;
        char *dst = wbuf + buf * CV_TILE_BYTES;
#pragma unroll
        for (int r = 0; r < BM / 64; ++r) {
            const int blk = r * 8 + wave;                       // 8 rows = 1 KB per instruction
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wk + (long long)(blk * 8 + sub) * 9 * Cin * 2),
                                             (__attribute__((address_space(3))) void *)(dst + blk * 1024), 16, 0, 0);
        }
    };
    // one third (11 of 33 row blocks) of the pixel segment of group g
    auto stage_x = [&](int g, int third) {
        const int dyi = g / ks, kc = g - dyi * ks;
        const int yy = y + (dyi - 1) * dil;
        const bool yok = yy >= 0 && yy < H;
        const char *xk = xbase + ((long long)(dyi - 1) * dil * W) * Cin * 2 + (long long)kc * CV_BK * 2 + chunk_byte;
        char *dst = xbuf + (g & 1) * CV_XSEG_BYTES;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = r * 8 + wave;
            if (i >= 11) break;
            const int blk = third * 11 + i;
            const int px = x0 - CV_HALO + blk * 8 + sub;
            const bool ok = yok && px >= 0 && px < W;
            // a zero line for padding pixels (its 128 bytes are read at the chunk offset only)
            const char *src = ok ? xk + (long long)px * Cin * 2 : zero_line + chunk_byte;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + blk * 1024), 16, 0, 0);
        }
    };

    // ---- accumulators: wave (wm, wn) owns channels [wm*WROWS, +WROWS) x pixels [wn*NJ*16, +NJ*16)
    const int wm = wave / WN, wn = wave % WN;
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fk = lane >> 4;                     // fragment row, 16-byte k chunk inside a 32-k step

    stage_w(0, 0);
    stage_x(0, 0); stage_x(0, 1); stage_x(0, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        const int g = t / 3, dxi = t - g * 3;
        // (the loads of the next K step go out in one burst: spreading them between the MFMA groups was
        // measured 20 % slower — every global_load_lds re-programs M0 and breaks the MFMA stream)
        if (t + 1 < nk) stage_w(t + 1, cur ^ 1);
        if (g + 1 < ngroups) stage_x(g + 1, dxi);
        const char *lw = wbuf + cur * CV_TILE_BYTES, *lx = xbuf + (g & 1) * CV_XSEG_BYTES;
        const int xshift = CV_HALO + (dxi - 1) * dil + wn * (NJ * 16) + frow;      // segment row of fragment 0
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 wf[MI], pf[NJ];
            const int chunk = kk * 4 + fk;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = wm * WROWS + i * 16 + frow;
                wf[i] = *(const bf16x8 *)(lw + row * 128 + ((chunk ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = xshift + j * 16;
                pf[j] = *(const bf16x8 *)(lx + row * 128 + ((chunk ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], pf[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane holds channels c..c+3 (c = tile channel base + (lane>>4)*4) of pixel (lane & 15)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int xx = x0 + wn * (NJ * 16) + j * 16 + (lane & 15);
        if (xx >= W) continue;
        const long long pix = (long long)row_id * W + xx;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int c = n0 + wm * WROWS + i * 16 + (lane >> 4) * 4;
            const float4 bv = *(const float4 *)(bias + c);
            float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
            if (HAS_RES) {
                const uint2 rr = *(const uint2 *)(R + pix * Cout + c);
                v0 += __uint_as_float(rr.x << 16); v1 += __uint_as_float(rr.x & 0xffff0000u);
                v2 += __uint_as_float(rr.y << 16); v3 += __uint_as_float(rr.y & 0xffff0000u);
            }
            if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            uint2 o;
            o.x = (unsigned)f32_to_bf16_bits(v0) | ((unsigned)f32_to_bf16_bits(v1) << 16);
            o.y = (unsigned)f32_to_bf16_bits(v2) | ((unsigned)f32_to_bf16_bits(v3) << 16);
            *(uint2 *)(Y + pix * Cout + c) = o;
        }
    }
}

// workgroup barrier that orders LDS only: __syncthreads() also waits for every outstanding global load and store
__device__ __forceinline__ void cv_lds_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Round 5: the same convolution with the two waves of every SIMD in OPPOSITE phases (the half-period lag of
// k_gemm_f16x3_stag, spa_gemm16.hip, where the in-kernel stamps and the argument are).  A K step is two half periods
// R = [stage step t + 1, read the step's 24 fragments from LDS]  |  M = [its 64 matrix instructions]  with an LDS-only barrier after
// each; waves 4-7 run the same program half a period late (one extra barrier at their start), so every SIMD has one wave
// staging / reading and one multiplying instead of both doing the same thing.  Buffers as before (two weight tiles, two pixel
// segments staged in thirds): a buffer's last readers are the R of both halves one step (one group) earlier, the late half's
// ends at the barrier that ends the early half's period; both halves stage at the top of their period and wait for their own
// loads before the barrier that precedes the early half's next R.  Every accumulator receives the same matrix instructions in the
// same order: bit-identical outputs (tests/test_gpu_pipeline.py, tools/conv_bench.py).
template <int HAS_RES, int BM>
__global__ __launch_bounds__(CV_THREADS) void k_conv3x3_bf16_stag(const __bf16 *__restrict__ X, const __bf16 *__restrict__ Wt,
                                                                  const float *__restrict__ bias,
                                                                  const __bf16 *__restrict__ R, __bf16 *__restrict__ Y,
                                                                  const char *__restrict__ zero_line, int B, int H, int W,
                                                                  int Cin, int Cout, int dil, int relu, int xtiles,
                                                                  int ntiles, int total_tiles)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];     // [2] weight tiles 32 KB | [2] pixel segments 33 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = wave >= 4;
    const int nwg = total_tiles;
    int id = blockIdx.x;
    {
        const int q = nwg / 8, rem = nwg % 8, xcd = id % 8, idx = id / 8;
        id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    const int nt = id % ntiles, pt = id / ntiles;
    const int xt = pt % xtiles, row_id = pt / xtiles;               // row_id = b * H + y
    const int y = row_id % H;
    const int x0 = xt * CV_BN, n0 = nt * BM;
    constexpr int WN = BM == 64 ? 8 : 4;
    constexpr int MI = BM == 256 ? 8 : 4;
    constexpr int NJ = 256 / WN / 16;
    constexpr int WROWS = MI * 16;

    char *wbuf = lds, *xbuf = lds + 2 * CV_TILE_BYTES;
    const int sub = lane >> 3, cs = lane & 7;
    const int chunk_byte = (cs ^ sub) << 4;
    const char *wbase = (const char *)(Wt + (long long)n0 * 9 * Cin);
    const char *xbase = (const char *)(X + (long long)row_id * W * Cin);
    const int ks = Cin / CV_BK;
    const int nk = 9 * ks, ngroups = 3 * ks;

    auto stage_w = [&](int t, int buf) {
        const int g = t / 3, dxi = t - g * 3;
        const int dyi = g / ks, kc = g - dyi * ks;
        const char *wk = wbase + ((long long)(dyi * 3 + dxi) * Cin + (long long)kc * CV_BK) * 2 + chunk_byte;
        char *dst = wbuf + buf * CV_TILE_BYTES;
#pragma unroll
        for (int r = 0; r < BM / 64; ++r) {
            const int blk = r * 8 + wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wk + (long long)(blk * 8 + sub) * 9 * Cin * 2),
                                             (__attribute__((address_space(3))) void *)(dst + blk * 1024), 16, 0, 0);
        }
    };
    auto stage_x = [&](int g, int third) {
        const int dyi = g / ks, kc = g - dyi * ks;
        const int yy = y + (dyi - 1) * dil;
        const bool yok = yy >= 0 && yy < H;
        const char *xk = xbase + ((long long)(dyi - 1) * dil * W) * Cin * 2 + (long long)kc * CV_BK * 2 + chunk_byte;
        char *dst = xbuf + (g & 1) * CV_XSEG_BYTES;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = r * 8 + wave;
            if (i >= 11) break;
            const int blk = third * 11 + i;
            const int px = x0 - CV_HALO + blk * 8 + sub;
            const bool ok = yok && px >= 0 && px < W;
            const char *src = ok ? xk + (long long)px * Cin * 2 : zero_line + chunk_byte;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + blk * 1024), 16, 0, 0);
        }
    };

    const int wm = wave / WN, wn = wave % WN;
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fk = lane >> 4;

    stage_w(0, 0);
    stage_x(0, 0); stage_x(0, 1); stage_x(0, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    cv_lds_barrier();
    if (late) cv_lds_barrier();                    // the extra barrier = half a period of delay
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        const int g = t / 3, dxi = t - g * 3;
        // ---- R: stage step t + 1 (while the partner wave multiplies), read this step's fragments
        if (t + 1 < nk) stage_w(t + 1, cur ^ 1);
        if (g + 1 < ngroups) stage_x(g + 1, dxi);
        const char *lw = wbuf + cur * CV_TILE_BYTES, *lx = xbuf + (g & 1) * CV_XSEG_BYTES;
        const int xshift = CV_HALO + (dxi - 1) * dil + wn * (NJ * 16) + frow;
        bf16x8 wf[2][MI], pf[2][NJ];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = kk * 4 + fk;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = wm * WROWS + i * 16 + frow;
                wf[kk][i] = *(const bf16x8 *)(lw + row * 128 + ((chunk ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = xshift + j * 16;
                pf[kk][j] = *(const bf16x8 *)(lx + row * 128 + ((chunk ^ (row & 7)) << 4));
            }
        }
        // the late half's loads (staged at the top of this period) must have landed before the early half reads them behind this barrier
        if (late) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cv_lds_barrier();
        // ---- M
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][i], pf[kk][j], acc[i][j], 0, 0, 0);
        if (!late) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            cv_lds_barrier();
        } else if (t + 1 < nk) {
            cv_lds_barrier();
        }
    }

    // ---- epilogue: lane holds channels c..c+3 (c = tile channel base + (lane>>4)*4) of pixel (lane & 15)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int xx = x0 + wn * (NJ * 16) + j * 16 + (lane & 15);
        if (xx >= W) continue;
        const long long pix = (long long)row_id * W + xx;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int c = n0 + wm * WROWS + i * 16 + (lane >> 4) * 4;
            const float4 bv = *(const float4 *)(bias + c);
            float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
            if (HAS_RES) {
                const uint2 rr = *(const uint2 *)(R + pix * Cout + c);
                v0 += __uint_as_float(rr.x << 16); v1 += __uint_as_float(rr.x & 0xffff0000u);
                v2 += __uint_as_float(rr.y << 16); v3 += __uint_as_float(rr.y & 0xffff0000u);
            }
            if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            uint2 o;
            o.x = (unsigned)f32_to_bf16_bits(v0) | ((unsigned)f32_to_bf16_bits(v1) << 16);
            o.y = (unsigned)f32_to_bf16_bits(v2) | ((unsigned)f32_to_bf16_bits(v3) << 16);
            *(uint2 *)(Y + pix * Cout + c) = o;
        }
    }
}

// x (B,H,W,Cin) bf16 channels-last, wt (Cout,9,Cin) bf16 (tap = ky*3 + kx), bias (Cout) float32,
// residual (B,H,W,Cout) bf16 or NULL, y (B,H,W,Cout) bf16.  stride 1, padding = dilation.
extern "C" int spa_conv3x3_bf16(spa_ctx *ctx, const void *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                const void *wt, int32_t Cout, const float *bias, const void *residual,
                                int32_t relu, int32_t dilation, void *y, void *stream)
{
    SPA_ARG(ctx && x && wt && bias && y && B > 0 && H > 0 && W > 0 && dilation >= 1);
    SPA_ARG(Cin % CV_BK == 0 && Cout % 64 == 0 && dilation <= CV_HALO);
    SPA_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)wt % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)bias % 16) == 0);
    SPA_ARG(((uintptr_t)residual % 16) == 0);
    hipStream_t s = spa_stream(stream);
    char *zero;
    int rc = spa_ws_reserve(ctx, WS_ZERO_LINE, 4096, (void **)&zero);
    if (rc != SPA_OK) return rc;
    if (!ctx->zero_line_ready) {
        SPA_HIP(hipMemsetAsync(zero, 0, 4096, s));
        ctx->zero_line_ready = 1;
    }
    const int bm = Cout % 256 == 0 ? 256 : (Cout % 128 == 0 ? 128 : 64);
    const int xtiles = (W + CV_BN - 1) / CV_BN, ntiles = Cout / bm;
    const long long total = (long long)B * H * xtiles * ntiles;
    SPA_ARG(total < (1ll << 31));
    const size_t lds = 2 * (size_t)CV_TILE_BYTES + 2 * (size_t)CV_XSEG_BYTES;
    if (!ctx->conv_attr_done) {
#define CV_ATTR(R, M) SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_bf16<R, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
        CV_ATTR(0, 256); CV_ATTR(1, 256); CV_ATTR(0, 128); CV_ATTR(1, 128); CV_ATTR(0, 64); CV_ATTR(1, 64);
#undef CV_ATTR
        ctx->conv_attr_done = 1;
    }
    SpaProfScope prof_(ctx, PROF_DRN_CONV, s);
    // SPA_CONV16_STAGGER (read once): unset / 1 = the half-period-lag kernel for the 256-channel tile (512 -> 512, 30 images: 4.14
    // against 4.48 ms with the residual, 3.99 against 4.39 without; 256 -> 512 2.24 against 2.45; 256 -> 256 1.31 against 1.39:
    // tools/conv16_ab.py, digests equal), 0 = round 1's kernel (also the narrow tiles')
    static const int stag = getenv("SPA_CONV16_STAGGER") ? atoi(getenv("SPA_CONV16_STAGGER")) : 1;
    if (stag && bm == 256) {
        if (!ctx->conv_stag_attr_done) {
            SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_bf16_stag<0, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_bf16_stag<1, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ctx->conv_stag_attr_done = 1;
        }
#define CV_LAUNCH_S(R)                                                                                                   \
    hipLaunchKernelGGL((k_conv3x3_bf16_stag<R, 256>), dim3((unsigned)total), dim3(CV_THREADS), lds, s, (const __bf16 *)x, \
                       (const __bf16 *)wt, bias, (const __bf16 *)residual, (__bf16 *)y, (const char *)zero, B, H, W,    \
                       Cin, Cout, dilation, relu, xtiles, ntiles, (int)total)
        if (residual) CV_LAUNCH_S(1); else CV_LAUNCH_S(0);
#undef CV_LAUNCH_S
        SPA_LAUNCH_CHECK();
        return SPA_OK;
    }
#define CV_LAUNCH(R, M)                                                                                                  \
    hipLaunchKernelGGL((k_conv3x3_bf16<R, M>), dim3((unsigned)total), dim3(CV_THREADS), lds, s, (const __bf16 *)x,       \
                       (const __bf16 *)wt, bias, (const __bf16 *)residual, (__bf16 *)y, (const char *)zero, B, H, W,    \
                       Cin, Cout, dilation, relu, xtiles, ntiles, (int)total)
    if (residual) {
        if (bm == 256) CV_LAUNCH(1, 256); else if (bm == 128) CV_LAUNCH(1, 128); else CV_LAUNCH(1, 64);
    } else {
        if (bm == 256) CV_LAUNCH(0, 256); else if (bm == 128) CV_LAUNCH(0, 128); else CV_LAUNCH(0, 64);
    }
#undef CV_LAUNCH
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
