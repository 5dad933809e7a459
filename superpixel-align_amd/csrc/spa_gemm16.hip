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

template <int BM, int BN>
__global__ __launch_bounds__(G16_THREADS) void k_gemm_f16x3(const char *__restrict__ X, const char *__restrict__ Wt,
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
    const int bn = bm == 256 ? 256 : 128;
    const int ntiles = Cout / bm;
    const long long total = rows / bn * ntiles;
    SPA_ARG(total < (1ll << 31));
    const size_t lds = 2 * (size_t)(bm + bn) * 128;
    if (!ctx->gemm16_attr_done) {
        SPA_HIP(hipFuncSetAttribute((const void *)k_gemm_f16x3<256, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128));
        SPA_HIP(hipFuncSetAttribute((const void *)k_gemm_f16x3<128, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 128));
        ctx->gemm16_attr_done = 1;
    }
    SpaProfScope prof_(ctx, bm == 256 ? PROF_DRN_GEMM16 : PROF_DRN_GEMM16_N, s);
    static const int force_per_cu = getenv("SPA_GEMM16_PER_CU") ? atoi(getenv("SPA_GEMM16_PER_CU")) : 0;
    const int per_cu = force_per_cu > 0 ? force_per_cu : (lds > 80 * 1024 ? 1 : 2);
    long long grid = (long long)ctx->n_cu * per_cu;
    if (grid > total * zcount) grid = total * zcount;
    if (bm == 256)
        hipLaunchKernelGGL((k_gemm_f16x3<256, 256>), dim3((unsigned)grid), dim3(G16_THREADS), lds, s, (const char *)x, (const char *)wt, y,
                           (int)rows, Cin, Cout, ntiles, (int)total, zcount, rows * Cin, (long long)Cout * Cin, rows * Cout, (const unsigned *)amax);
    else
        hipLaunchKernelGGL((k_gemm_f16x3<128, 128>), dim3((unsigned)grid), dim3(G16_THREADS), lds, s, (const char *)x, (const char *)wt, y,
                           (int)rows, Cin, Cout, ntiles, (int)total, zcount, rows * Cin, (long long)Cout * Cin, rows * Cout, (const unsigned *)amax);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
