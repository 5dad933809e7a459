// float32 implicit-GEMM 3x3 (dilated) convolution for the stride-1 3x3 layers of the float32 DRN (models/drn.py:
// 230-285: every BasicBlock / plain layer from 64 channels up, ~90 % of the network's FLOPs) on the float32 matrix
// cores (v_mfma_f32_16x16x4_f32: an exact fmaf chain per output, float32 in, float32 accumulate), with the bias left
// by the folded BatchNorm, the residual add of a BasicBlock and the ReLU fused into the epilogue — the float32
// network otherwise pays a separate read-modify-write pass over every convolution output (k_bias_act, 13 ms per 30
// images) plus MIOpen's own zero-fill of the output its split-K kernels accumulate into (4.5 ms).
//
//   Y[b, y, x, n] = relu?( bias[n] + res[b, y, x, n] + sum_{tap, c} X[b, y + dy(tap)*d, x + dx(tap)*d, c] * Wt[n, tap, c] )
//
// Same mapping as the bf16 kernel (spa_conv.hip): GEMM with M = Cout (weights are the MFMA A operand), N = 256
// consecutive pixels of one image row, K = 9 * Cin walked as (dy, 32 input channels, dx) — LDS rows are 128 bytes
// = 32 float32 channels, 16-byte chunks XOR-swizzled by (row & 7) on the source address, ONE row segment of
// 256 + 2*4 pixels per (dy, channel step) serving the three dx taps at a row offset, everything global -> LDS by
// 16-byte global_load_lds, two buffers of each, 8 waves.  The float32 matrix pipe is 16x slower than the bf16 one
// for the same bytes of operands, so a K step here is ~16 000 MFMA cycles per SIMD against one workgroup barrier
// and 43 KB of loads: the loop is MFMA bound by a wide margin.
#include "spa_common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define C32_BN 256            // pixels per workgroup
#define C32_BK 32             // K step (input channels of one tap): 128-byte LDS rows
#define C32_THREADS 512
#define C32_TILE_BYTES (256 * C32_BK * 4)          // one operand tile: 32 KB
#define C32_HALO 4                                 // pixels either side of the 256-pixel segment: dilation <= 4
#define C32_XSEG_BYTES ((256 + 2 * C32_HALO) * C32_BK * 4)   // 33 row blocks of 8 pixels

// TAPS = 9: 3x3 (dilated), padding = dilation; TAPS = 1: the 1x1 projection of a BasicBlock (models/drn.py:23-57,
// stride 1): the same loop with the centre tap only (wt (Cout, 1, Cin), one K step per 32 channels)
// BN = pixels per workgroup: 256, or 128 for the narrow channel tiles (BM 64 / 128): their K steps are short (a
// quarter / half of the MFMA work per barrier), and with half the pixel segment two or three workgroups fit a CU
// (67 / 51 KB of LDS), so one workgroup's barrier and load wait hide under another's matrix work
// SPLIT (round 3): the same convolution on the 16-bit matrix cores at float32 accuracy (spa_gemm16.hip has the argument):
// Wt holds, per 32-channel group of a (channel, tap) row, the two half-precision planes [32 x h | 32 x l] of t * w (t a
// power of two), the pixels stay float32 in memory and in LDS and are scaled by 2^(14 - e) (e = exponent of *amax_in, a
// bound on max |x|) and split in registers; three v_mfma_f32_16x16x32_f16 replace eight v_mfma_f32_16x16x4_f32 per
// 32-channel step (5.3x less matrix time).  The epilogue multiplies by unscale = 2^(e - 14) / t before the bias and,
// when amax_out is given, records the largest magnitude it stores (the next layer's scale).
template <int HAS_RES, int BM, int TAPS, int BN, bool SPLIT = false, int S = 1>
__global__ __launch_bounds__(C32_THREADS) void k_conv3x3_f32(const float *__restrict__ X, const float *__restrict__ Wt,
                                                             const float *__restrict__ bias,
                                                             const float *__restrict__ R, float *__restrict__ Y,
                                                             const char *__restrict__ zero_line, int B, int H, int W,
                                                             int Cin, int Cout, int dil, int relu, int xtiles,
                                                             int ntiles, int total_tiles, int zcount, long long xz,
                                                             long long wz, long long yz, int late_prefetch,
                                                             const unsigned *__restrict__ amax_in = nullptr,
                                                             unsigned *__restrict__ amax_out = nullptr, float inv_t = 1.f,
                                                             int Hi = 0, int Wi = 0, float *__restrict__ Y2 = nullptr, int csplit = 0)
{
    // S = 2 (round 3): stride 2, padding 1 (H, W = OUTPUT size, Hi, Wi = input size): the pixel segment of a K step is the
    // 2 BN + 8 input pixels the tile's taps touch, fragments read every other row of it.  Y2 / csplit: two outputs from one
    // pass over the input — output channels [0, csplit) go to Y with ReLU as asked, [csplit, Cout) to Y2 without (the 1x1
    // stride-2 projection of a BasicBlock, models/drn.py:195-203, as extra channels whose only non-zero tap is the centre)
    if (S == 1) { Hi = H; Wi = W; }
    extern __shared__ __attribute__((aligned(1024))) char lds32[];   // [2] weight tiles 32 KB | [2] pixel segments 33 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // uniform: block and LDS addresses of the staging stay scalar
    // XCD-aware tile order: consecutive workgroup ids go round-robin over the 8 XCDs; give every XCD a
    // contiguous range of tiles (neighbouring rows of one image share two of their three input rows in L2)
    // PERSISTENT tile loop: a workgroup takes tiles vid = blockIdx.x, + gridDim.x, ... of zcount problems of
    // total_tiles tiles each (zcount > 1: the 16 GEMMs of a Winograd layer, operands xz / wz / yz elements apart).
    // The first K step of the NEXT tile is staged before the epilogue of the current one, so the stores of a
    // 256 x 256 output tile (the bulk of a short-K GEMM's overhead) drain under the next tile's loads and matrix work.
    const long long all_tiles = (long long)zcount * total_tiles;
    const int nwg = total_tiles;
    int row_id, y, x0, n0;
    const char *wbase, *xbase;
    float *ybase;
    const float *rbase;
    auto locate = [&](long long vid) {
        const int z = (int)(vid / total_tiles);
        int id = (int)(vid - (long long)z * total_tiles);
        {
            const int q = nwg / 8, rem = nwg % 8, xcd = id % 8, idx = id / 8;
            id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
        }
        // tile id -> (pixel tile, channel tile): the channel tiles of one pixel tile are adjacent
        const int nt = id % ntiles, pt = id / ntiles;
        const int xt = pt % xtiles;
        row_id = pt / xtiles;                                           // b * H + y
        y = row_id % H;
        x0 = xt * BN; n0 = nt * BM;
        wbase = (const char *)(Wt + (long long)z * wz + (long long)n0 * TAPS * Cin);
        xbase = (const char *)(X + (long long)z * xz + ((long long)(row_id / H) * Hi + (long long)y * S) * Wi * Cin);      // input row y * S, pixel 0
        ybase = Y + (long long)z * yz;
        rbase = R + (long long)z * yz;
    };
    constexpr int WN = BM == 64 ? 8 : 4;                 // waves along the pixels (2 x 4 waves for the 64-channel split tile: 1.76 vs 1.68 ms)
    constexpr int MI = BM / (8 / WN) / 16;               // 16-channel MFMA tiles per wave: 8, 4, 4 (or 2)
    constexpr int NJ = BN / WN / 16;                     // 16-pixel MFMA tiles per wave
    // SPLIT: the pixel segment has its own LDS image (below), in 16-row blocks for the stride-2 form: an even block count
    constexpr int XBLK = (S * BN + 2 * C32_HALO) / 8 + ((SPLIT && S == 2) ? ((S * BN + 2 * C32_HALO) / 8) % 2 : 0);    // 8-pixel row blocks of the pixel segment
    constexpr int XTHIRD = (XBLK + 2) / 3;               // staged per K step: 11 or 6
    constexpr int XSEG = XBLK * 8 * 128;                 // bytes
    constexpr int WROWS = MI * 16;                       // channels per wave

    // K order: (dy, 32-channel step, dx).  The three dx taps of one (dy, k step) read the SAME input pixels
    // shifted by the dilation: one row segment of 256 + 2*C32_HALO pixels is staged per (dy, k step) — a third
    // of it with each of the previous group's three K steps — and the taps read it at a row offset.  A K step
    // thus moves 32 KB of weights + 11 KB of pixels instead of 32 + 32.
    char *wbuf = lds32, *xbuf = lds32 + 2 * (BM * 128);
    const int sub = lane >> 3, cs = lane & 7;
    const int chunk_byte = (cs ^ sub) << 4;        // staged row = block * 8 + sub: (row & 7) = sub for every block
    // LDS image of the PIXEL segment in the split-plane form (round 4).  A ds_read_b128 is served in four groups of 16 lanes —
    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) — i.e. the eight fragment rows
    // {0-3, 12-15} of k group 2p together with the eight rows {4-11} of k group 2p + 1 (or the other way round), and a group
    // is conflict free when its 16 chunks fall into 16 different 16-byte slots of a 256-byte bank line (slot = 8 (row & 1)
    // + position in the row).  The pixel fragments read chunk c = 2 fk (then 2 fk + 1) of row base + frow for ANY base (the
    // taps shift it by the dilation): either set of eight rows covers every residue mod 8 once, and the two chunks of a
    // group differ by 2.  With chunk c of row r stored at position c ^ g(r & 7), g(r) = ((r >> 1) & 1) + 4 ((r >> 2) & 1)
    // = 0 0 1 1 4 4 5 5, the four rows of one parity put chunk c at c ^ {0, 1, 4, 5} and chunk c ^ 2 at c ^ {2, 3, 6, 7}: 16
    // distinct slots for every base.  (The weights' image c ^ (r & 7) above gave these reads 2-way conflicts, 4-way in the
    // stride-2 form whose fragments take every other row and so stay in one half of the bank lines: SQ_LDS_BANK_CONFLICT
    // 4.0 x SQ_ACTIVE_INST_LDS, verdict r3.)  g only swaps neighbours and 64-byte halves: a quad of staging lanes still
    // fetches 64 contiguous bytes of one row — an image that scattered a row's chunks over the 1 KB block was conflict free
    // as well and 14-24 % SLOWER in the GEMM form, its global_load_lds no longer coalesced.
    // Stride 2: fragment rows are base + 2 frow, so rows are stored de-interleaved in blocks of 16 — LDS row
    // rho(r) = 16 (r >> 4) + 8 (r & 1) + ((r >> 1) & 7) — and a fragment's rows are consecutive LDS rows again.
    auto xg = [](int r) { return ((r >> 1) & 1) | (((r >> 2) & 1) << 2); };
    const int xsub = sub;
    const int xchunk_byte = SPLIT ? ((cs ^ xg(sub)) << 4) : chunk_byte;
    const int ks = Cin / C32_BK;
    const int nk = TAPS * ks, ngroups = TAPS == 9 ? 3 * ks : ks;

    int pw_par = 0, px_par = 0;                  // buffer parities carried from tile to tile
    // Index arithmetic of the staging (round 3, PMC: 5.5 scalar + 4.4 vector instructions per matrix instruction in the
    // split-plane 64-channel layers, where a K step is only 12-24 matrix instructions per wave): the (dy, channel step, dx)
    // of a K step are carried incrementally by the loop instead of being divided out of t, and everything of an address
    // that does not change with the K step — the lane's row inside a tile, its swizzled chunk — is computed once.
    const unsigned wlane = (unsigned)(wave * 8 + sub) * (unsigned)(TAPS * Cin * 4) + (unsigned)chunk_byte;   // + r * 64 rows
    const unsigned wrstep = 64u * (unsigned)(TAPS * Cin * 4);
    auto stage_w = [&](int tap, int kc, int buf) {               // tap = dy * 3 + dx (0 for the GEMM form)
        const char *wk = wbase + ((long long)tap * Cin + (long long)kc * C32_BK) * 4;
        char *dst = wbuf + buf * (BM * 128) + wave * 1024;
#pragma unroll
        for (int r = 0; r < BM / 64; ++r)                       // 8 rows = 1 KB per instruction
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wk + (wlane + r * wrstep)),
                                             (__attribute__((address_space(3))) void *)(dst + r * 8192), 16, 0, 0);
    };
    // one third (11 of 33 row blocks) of the pixel segment of the group (dyi, kc) into segment buffer `xb`
    int px0 = 0, xoff0 = 0;          // first pixel of this lane's staged rows (tile start - halo + sub) and its byte offset
    const int pxstep = 8 * Cin * 4;  // bytes between two row blocks of 8 pixels (32-bit offsets inside one image row)
    constexpr bool X16 = SPLIT && S == 2;      // de-interleaved 16-row blocks: 1 KB block b = image rows 16 (b >> 1) + 2 sub + (b & 1)
    auto locate_x = [&]() { px0 = x0 * S - C32_HALO + (X16 ? 2 * xsub : xsub); xoff0 = px0 * (Cin * 4); };
    auto stage_x = [&](int dyi, int kc, int third, int xb) {
        const int yy = y * S + (dyi - 1) * dil;
        const bool yok = yy >= 0 && yy < Hi;
        const char *xk = xbase + ((long long)(dyi - 1) * dil * Wi) * Cin * 4 + (long long)kc * C32_BK * 4 + xchunk_byte;
        char *dst = xbuf + xb * XSEG;
#pragma unroll
        for (int r = 0; r < (XTHIRD + 7) / 8; ++r) {
            const int i = r * 8 + wave;
            if (i >= XTHIRD) break;
            const int blk = third * XTHIRD + i;
            if (blk >= XBLK) break;
            const int bpx = X16 ? (blk >> 1) * 16 + (blk & 1) : blk * 8;
            const int px = px0 + bpx;
            const bool ok = yok && (unsigned)px < (unsigned)Wi;
            // a zero line for padding pixels (its 128 bytes are read at the chunk offset only)
            const char *src = ok ? xk + (xoff0 + (X16 ? bpx * (Cin * 4) : blk * pxstep)) : zero_line + xchunk_byte;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + blk * 1024), 16, 0, 0);
        }
    };

    // GEMM form (TAPS == 1): no tap ever reads beside the tile, so the segment is exactly the BN rows of the tile —
    // BN/64 loads per wave and K step at precomputed row addresses (a row beyond the image is clamped to the last
    // one: its products land in columns the epilogue never stores)
    const char *xrow[BN / 64];
    auto locate_rows = [&]() {
#pragma unroll
        for (int r = 0; r < BN / 64; ++r) {
            int px = x0 + (r * 8 + wave) * 8 + xsub;
            px = px < W ? px : W - 1;
            xrow[r] = xbase + (long long)px * Cin * 4 + xchunk_byte;
        }
    };
    auto stage_x1 = [&](int g, int xb) {
        char *dst = xbuf + xb * XSEG;
#pragma unroll
        for (int r = 0; r < BN / 64; ++r)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xrow[r] + (long long)g * C32_BK * 4),
                                             (__attribute__((address_space(3))) void *)(dst + (r * 8 + wave) * 1024), 16, 0, 0);
    };

    // ---- accumulators: wave (wm, wn) owns channels [wm*WROWS, +WROWS) x pixels [wn*NJ*16, +NJ*16)
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fk = lane >> 4;                     // fragment row, 16-byte k chunk inside a 32-k step
    long long vid = blockIdx.x;
    if (vid >= all_tiles) return;
    bool first_tile = true;
    const bool counted_wait = (W % BN) == 0;          // every lane of every epilogue store is live
    locate(vid);
    stage_w(0, 0, 0);
    if (TAPS == 1) { locate_rows(); stage_x1(0, 0); }
    else { locate_x(); stage_x(0, 0, 0, 0); stage_x(0, 0, 1, 0); stage_x(0, 0, 2, 0); }
    float sc16 = 1.f, unscale = 1.f;
    unsigned amx = 0;
    if (SPLIT) {
        const unsigned bits = *amax_in;
        int e = (int)(bits >> 23) - 127;
        e = bits == 0u ? 0 : (e < -100 ? -100 : (e > 100 ? 100 : e));
        sc16 = __uint_as_float((unsigned)(127 + 14 - e) << 23);
        unscale = __uint_as_float((unsigned)(127 - 14 + e) << 23) * inv_t;
    }
    int e_row = 0, e_x0 = 0, e_n0 = 0;
    float *e_y = nullptr;
    const float *e_r = nullptr;
    bool more = false;
  for (;;) {
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the staged loads of this tile's first K step must have landed.  They were issued BEFORE the previous tile's
    // epilogue stores (vmcnt counts both, in order), so when every wave issued exactly MI*NJ stores after them —
    // full tiles, no residual loads in between — waiting until MI*NJ operations remain is enough and the stores
    // keep draining under this tile's matrix work
    if (!HAS_RES && counted_wait && !first_tile) {
        static_assert(MI * NJ <= 63, "vmcnt range");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MI * NJ) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    first_tile = false;
    __syncthreads();
    // (dxi, kc, dyi) of step t and the segment buffer of its group, carried incrementally
    int dxi = TAPS == 9 ? 0 : 1, kc = 0, dyi = TAPS == 9 ? 0 : 1, g = 0, xcur = px_par;
    for (int t = 0; t < nk; ++t) {
        const int cur = (t + pw_par) & 1;
        // step t + 1 and group g + 1
        int ndxi = dxi, nkc = kc, ndyi = dyi, gkc = kc + 1, gdyi = dyi;
        if (TAPS == 9) {
            ndxi = dxi + 1;
            if (ndxi == 3) { ndxi = 0; nkc = kc + 1; if (nkc == ks) { nkc = 0; ndyi = dyi + 1; } }
            if (gkc == ks) { gkc = 0; gdyi = dyi + 1; }
        } else nkc = kc + 1;
        // (the loads of the next K step go out in one burst: spreading them between the MFMA groups was
        // measured 20 % slower — every global_load_lds re-programs M0 and breaks the MFMA stream; issuing the next group's
        // pixel thirds AFTER the step's wait, so that no wait covers an HBM load of the same step, was 4-6 % slower too)
#ifdef SPA_C32_FIXED_ADDR          // timing experiment (wrong numbers): step-invariant staging addresses, so that their arithmetic leaves the loop
        if (t + 1 < nk) stage_w(0, 0, cur ^ 1);
        if (g + 1 < ngroups) {
            if (TAPS == 9) stage_x(1, 0, dxi, xcur ^ 1);
            else stage_x1(0, xcur ^ 1);
        }
#else
        if (t + 1 < nk) stage_w(TAPS == 9 ? ndyi * 3 + ndxi : 0, nkc, cur ^ 1);
        if (g + 1 < ngroups) {
            if (TAPS == 9) stage_x(gdyi, gkc, dxi, xcur ^ 1);
            else stage_x1(g + 1, xcur ^ 1);
        }
#endif
        if (t + 1 == nk && !late_prefetch) {
            // LAST K step of the tile: nothing of this tile is left to load and the other buffer of each pair is
            // free, so the NEXT tile's first K step is staged now and travels under this step's matrix work (staged
            // after the loop it cost its full latency with the matrix pipe idle: ~7 us per tile, 5 % of a 16-step
            // tile and 20 % of a 4-step one).  The epilogue below needs this tile's coordinates: saved first.
            e_row = row_id; e_x0 = x0; e_n0 = n0; e_y = ybase; e_r = rbase;
            vid += gridDim.x;
            more = vid < all_tiles;
            if (more) {
                const int npw = (pw_par + nk) & 1, npx = (px_par + ngroups) & 1;
                locate(vid);
                stage_w(0, 0, npw);
                if (TAPS == 1) { locate_rows(); stage_x1(0, npx); }
                else { locate_x(); stage_x(0, 0, 0, npx); stage_x(0, 0, 1, npx); stage_x(0, 0, 2, npx); }
            }
        }
        const char *lw = wbuf + cur * (BM * 128), *lx = xbuf + xcur * XSEG;
        const int xshift = (TAPS == 9 ? C32_HALO + (dxi - 1) * dil : 0) + S * (wn * (NJ * 16) + frow);      // segment row of fragment 0
        if constexpr (SPLIT) {
            f16x8 wh[MI], wl[MI], ph[NJ], pl[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = wm * WROWS + i * 16 + frow;
                wh[i] = *(const f16x8 *)(lw + row * 128 + ((fk ^ (row & 7)) << 4));
                wl[i] = *(const f16x8 *)(lw + row * 128 + (((4 + fk) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = xshift + S * j * 16;
                const int lrow = X16 ? (row >> 4) * 16 + (row & 1) * 8 + ((row >> 1) & 7) : row;          // LDS row
                const char *pr = lx + lrow * 128;
                const int pos = (2 * fk) ^ xg(lrow);
                const f32x4 a = *(const f32x4 *)(pr + (pos << 4));                 // chunk 2 fk
                const f32x4 b = *(const f32x4 *)(pr + ((pos ^ 1) << 4));           // chunk 2 fk + 1
                // (round 5: the split as four mixed-precision fmas per element pair, spa_split16_pair — 16 single-issue vector
                // instructions per fragment instead of 24 with packed float32 ones; the same bits)
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                u32x4 hu, lu;
                { unsigned l; hu[0] = spa_split16_pair(a[0], a[1], sc16, l); lu[0] = l; }
                { unsigned l; hu[1] = spa_split16_pair(a[2], a[3], sc16, l); lu[1] = l; }
                { unsigned l; hu[2] = spa_split16_pair(b[0], b[1], sc16, l); lu[2] = l; }
                { unsigned l; hu[3] = spa_split16_pair(b[2], b[3], sc16, l); lu[3] = l; }
                ph[j] = __builtin_bit_cast(f16x8, hu);
                pl[j] = __builtin_bit_cast(f16x8, lu);
            }
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
        } else
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // a lane's 16 bytes are channels (kk*16 + fk*4 .. +3) of its row: MFMA q of the four multiplies channel
            // kk*16 + 4*kgroup + q of every k group (A and B use the same assignment, so the products pair up)
            f32x4 wf[MI], pf[NJ];
            const int chunk = kk * 4 + fk;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = wm * WROWS + i * 16 + frow;
                wf[i] = *(const f32x4 *)(lw + row * 128 + ((chunk ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = xshift + S * j * 16;
                pf[j] = *(const f32x4 *)(lx + row * 128 + ((chunk ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][q], pf[j][q], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (TAPS == 9) { if (ndxi == 0) { ++g; xcur ^= 1; } }
        else { ++g; xcur ^= 1; }
        dxi = ndxi; kc = nkc; dyi = ndyi;
    }

    pw_par = (pw_par + nk) & 1;
    px_par = (px_par + ngroups) & 1;
    if (late_prefetch) {            // A/B switch (SPA_CONV32_LATE_PREFETCH=1): stage the next tile only now
        e_row = row_id; e_x0 = x0; e_n0 = n0; e_y = ybase; e_r = rbase;
        vid += gridDim.x;
        more = vid < all_tiles;
        if (more) {
            locate(vid);
            stage_w(0, 0, pw_par);
            if (TAPS == 1) { locate_rows(); stage_x1(0, px_par); }
            else { locate_x(); stage_x(0, 0, 0, px_par); stage_x(0, 0, 1, px_par); stage_x(0, 0, 2, px_par); }
        }
    }
    // ---- epilogue: lane holds channels c..c+3 (c = tile channel base + (lane>>4)*4) of pixel (lane & 15)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int xx = e_x0 + wn * (NJ * 16) + j * 16 + (lane & 15);
        if (xx >= W) continue;
        const long long pix = (long long)e_row * W + xx;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int c = e_n0 + wm * WROWS + i * 16 + (lane >> 4) * 4;
            const float4 bv = *(const float4 *)(bias + c);
            float v0, v1, v2, v3;
            if (SPLIT) { v0 = acc[i][j][0] * unscale + bv.x; v1 = acc[i][j][1] * unscale + bv.y; v2 = acc[i][j][2] * unscale + bv.z; v3 = acc[i][j][3] * unscale + bv.w; }
            else { v0 = acc[i][j][0] + bv.x; v1 = acc[i][j][1] + bv.y; v2 = acc[i][j][2] + bv.z; v3 = acc[i][j][3] + bv.w; }
            if (HAS_RES) {
                const float4 rr = *(const float4 *)(e_r + pix * Cout + c);
                v0 += rr.x; v1 += rr.y; v2 += rr.z; v3 += rr.w;
            }
            if (S == 2 && Y2) {
                if (c >= csplit) { *(float4 *)(Y2 + pix * (Cout - csplit) + (c - csplit)) = make_float4(v0, v1, v2, v3); continue; }
                if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                *(float4 *)(e_y + pix * csplit + c) = make_float4(v0, v1, v2, v3);
            } else {
                if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                *(float4 *)(e_y + pix * Cout + c) = make_float4(v0, v1, v2, v3);
            }
            if (SPLIT)
                amx = max(max(amx, __float_as_uint(v0) & 0x7fffffffu), max(__float_as_uint(v1) & 0x7fffffffu,
                          max(__float_as_uint(v2) & 0x7fffffffu, __float_as_uint(v3) & 0x7fffffffu)));
        }
    }
    if (!more) break;
  }
    if (SPLIT && amax_out) {
        for (int o = 32; o > 0; o >>= 1) amx = max(amx, (unsigned)__shfl_xor((int)amx, o));
        if (lane == 0 && amx > *(volatile unsigned *)amax_out) atomicMax(amax_out, amx);
    }
}

// x (B,H,W,Cin) float32 channels-last, wt (Cout,taps,Cin) float32 (tap = ky*3 + kx), bias (Cout) float32,
// residual (B,H,W,Cout) float32 or NULL, y (B,H,W,Cout) float32.  stride 1, padding = dilation.
int conv3x3_p16_launch(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin, const void *wt2, float inv_t,
                       int32_t Cout, const float *bias, const float *residual, int32_t relu, int32_t dilation,
                       const void *amax_in, void *amax_out, float *y, const char *zero, hipStream_t s);          // spa_convp.hip

template <int TAPS>
static int conv_f32_launch(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                           const float *wt, int32_t Cout, const float *bias, const float *residual,
                           int32_t relu, int32_t dilation, float *y, void *stream, bool prof = true, int zcount = 1,
                           long long xz = 0, long long wz = 0, long long yz = 0, const void *amax_in = nullptr,
                           void *amax_out = nullptr, float inv_t = 0.f)
{
    const bool split = amax_in != nullptr;          // wt is then the two-plane form of t * w, inv_t = 1 / t
    SPA_ARG(ctx && x && wt && y && B > 0 && H > 0 && W > 0 && dilation >= 1);
    SPA_ARG(Cin % C32_BK == 0 && Cout % 64 == 0 && dilation <= C32_HALO);
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
    if (!bias) { SPA_ARG(Cout <= 1024); bias = (const float *)zero; }      // raw GEMM: the zero line as bias
    // (128 x 128 tiles with two workgroups per CU for the GEMM form were measured: 386 vs 393 effective TFLOP/s on a
    // 512 -> 512 Winograd layer; staging on waves 4-7 between the two halves of their matrix work, so that the two waves
    // of a SIMD are never in the staging code together: 81 vs 121 TFLOP/s — the wave-dependent branch breaks the
    // straight-line schedule of the K step; without the output store the GEMM form runs 126: the store is 4 %)
    const int bm = Cout % 256 == 0 ? 256 : (Cout % 128 == 0 ? 128 : 64);
    // (split-plane form, 64 channels: every wave reads the whole weight tile from LDS, so 256 pixels per tile halve those reads
    // per matrix instruction: 1.53 vs 1.69 ms on the 64 -> 64 layer of 30 images — where a row fills 256-pixel tiles to 80 %)
    const bool wide64 = split && bm == 64 && (long long)((W + 255) / 256) * 256 * 4 <= 5ll * W;
    // (round 5: 512 pixels per tile where a row fills them — every wave then owns 64 channels x 64 pixels, 16 LDS fragment reads per
    // 48 matrix instructions instead of 12 per 24; one workgroup per CU, 150 KB of LDS.  SPA_CONV32_BN512=0: the 256-pixel tile)
    static const int bn512_on = getenv("SPA_CONV32_BN512") ? atoi(getenv("SPA_CONV32_BN512")) : 1;
    const bool wide512 = split && bm == 64 && TAPS == 9 && bn512_on && (long long)((W + 511) / 512) * 512 * 4 <= 5ll * W;
    const int bn = wide512 ? 512 : (bm == 256 || wide64 || getenv("SPA_CONV32_BN256") ? 256 : 128);
    const int xtiles = (W + bn - 1) / bn, ntiles = Cout / bm;
    const long long total = (long long)B * H * xtiles * ntiles;
    SPA_ARG(total < (1ll << 31));
    SPA_ARG((long long)(2 * W + 1024) * Cin * 4 < (1ll << 31));            // 32-bit byte offsets inside an image row
    const size_t lds = 2 * (size_t)bm * 128 + 2 * (size_t)(bn + 2 * C32_HALO) * 128;
    if (split) {
        SPA_ARG(bias && zcount == 1 && inv_t > 0.f);
        if (amax_out) spa_zero_word(amax_out, s);
    }
    const int bit = TAPS == 9 ? 1 : 2;
    if (!(ctx->conv32_attr_done & bit)) {
#define C32_ATTR(R, M, N) SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_f32<R, M, TAPS, N>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                      2 * M * 128 + 2 * (N + 2 * C32_HALO) * 128))
        C32_ATTR(0, 256, 256); C32_ATTR(1, 256, 256); C32_ATTR(0, 128, 256); C32_ATTR(1, 128, 256); C32_ATTR(0, 64, 256); C32_ATTR(1, 64, 256);
        C32_ATTR(0, 128, 128); C32_ATTR(1, 128, 128); C32_ATTR(0, 64, 128); C32_ATTR(1, 64, 128);
#undef C32_ATTR
#define C32_ATTR(R, M, N) SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_f32<R, M, TAPS, N, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                      2 * M * 128 + 2 * (N + 2 * C32_HALO) * 128))
        C32_ATTR(0, 256, 256); C32_ATTR(1, 256, 256); C32_ATTR(0, 128, 128); C32_ATTR(1, 128, 128); C32_ATTR(0, 64, 128); C32_ATTR(1, 64, 128);
#undef C32_ATTR
        ctx->conv32_attr_done |= bit;
    }
    // one slot per kernel form: the 3x3 convolutions, and the GEMM form (1x1 projections and the Winograd GEMM batches)
    // (the GEMM form's 256 x 256 instance has its own slot: it is the kernel with the most time per step, and its
    // average must be comparable with the rocprofv3 row of exactly that instance)
    SpaProfScope prof_(ctx, prof ? (split ? (TAPS == 1 ? PROF_DRN_CONV16_1X1 : (bm == 256 ? PROF_DRN_CONV16_256 : (bm == 128 ? PROF_DRN_CONV16_128 : PROF_DRN_CONV16))) : (TAPS == 9 ? PROF_DRN_CONV32 : (bm == 256 && !residual ? PROF_DRN_GEMM32 : PROF_DRN_GEMM32_N))) : -1, s);
    // persistent workgroups: as many as are resident at once (LDS: one per CU for the wide tiles, two or three for the
    // 128-pixel ones), each looping over its share of the tiles
    // (the split-plane form of the 64-channel tile is bound by its LDS reads — all eight waves read the same weight tile —
    // not by load latency: 1.77 / 1.68 / 2.25 ms with 3 / 2 / 1 workgroups per CU on the 64 -> 64 layer of 30 images)
    const int per_cu = lds > 80 * 1024 ? 1 : (lds > 53 * 1024 || (split && bm == 64) ? 2 : 3);
    const int late = getenv("SPA_CONV32_LATE_PREFETCH") ? 1 : 0;
    long long grid = (long long)ctx->n_cu * per_cu;
    if (grid > total * zcount) grid = total * zcount;
#define C32_LAUNCH(R, M, N)                                                                                                 \
    hipLaunchKernelGGL((k_conv3x3_f32<R, M, TAPS, N>), dim3((unsigned)grid), dim3(C32_THREADS), lds, s, x, wt, bias, residual, y,  \
                       (const char *)zero, B, H, W, Cin, Cout, dilation, relu, xtiles, ntiles, (int)total, zcount, xz, wz, yz, late)
#define C32_LAUNCH_S(R, M, N)                                                                                               \
    hipLaunchKernelGGL((k_conv3x3_f32<R, M, TAPS, N, true>), dim3((unsigned)grid), dim3(C32_THREADS), lds, s, x, wt, bias, residual, y,  \
                       (const char *)zero, B, H, W, Cin, Cout, dilation, relu, xtiles, ntiles, (int)total, zcount, xz, wz, yz, late, \
                       (const unsigned *)amax_in, (unsigned *)amax_out, inv_t)
    if (split) {
        // round 6: the 64- and 128-channel tiles of the 3x3 layers take the planes-in-LDS kernel of spa_convp.hip (bit-identical
        // outputs; SPA_CONVP=0 / spa_debug_set(ctx, 1, 0): this file's kernel, kept for A/B runs)
        if (ctx->convp_on && TAPS == 9 && (Cout == 64 || Cout == 128) && Cin >= 64 && zcount == 1)
            return conv3x3_p16_launch(ctx, x, B, H, W, Cin, wt, inv_t, Cout, bias, residual, relu, dilation, amax_in, amax_out, y, zero, s);
        // (128 channels x 256 pixels, one workgroup per CU: 1.06 vs 1.04 ms — no gain, not kept)
        if (TAPS == 9 && bm == 64 && bn == 512) {
            if (!(ctx->conv32_attr_done & 16)) {
                SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_f32<0, 64, 9, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 128 + 2 * (512 + 2 * C32_HALO) * 128));
                SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_f32<1, 64, 9, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 128 + 2 * (512 + 2 * C32_HALO) * 128));
                ctx->conv32_attr_done |= 16;
            }
            if (residual)
                hipLaunchKernelGGL((k_conv3x3_f32<1, 64, 9, 512, true>), dim3((unsigned)grid), dim3(C32_THREADS), lds, s, x, wt, bias, residual, y,
                                   (const char *)zero, B, H, W, Cin, Cout, dilation, relu, xtiles, ntiles, (int)total, zcount, xz, wz, yz, late,
                                   (const unsigned *)amax_in, (unsigned *)amax_out, inv_t);
            else
                hipLaunchKernelGGL((k_conv3x3_f32<0, 64, 9, 512, true>), dim3((unsigned)grid), dim3(C32_THREADS), lds, s, x, wt, bias, residual, y,
                                   (const char *)zero, B, H, W, Cin, Cout, dilation, relu, xtiles, ntiles, (int)total, zcount, xz, wz, yz, late,
                                   (const unsigned *)amax_in, (unsigned *)amax_out, inv_t);
            SPA_LAUNCH_CHECK();
            return SPA_OK;
        }
        if (bm == 64 && bn == 256) {
            const int bit64 = TAPS == 9 ? 32 : 64;
            if (!(ctx->conv32_attr_done & bit64)) {
                SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_f32<0, 64, TAPS, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 128 + 2 * (256 + 2 * C32_HALO) * 128));
                SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_f32<1, 64, TAPS, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 128 + 2 * (256 + 2 * C32_HALO) * 128));
                ctx->conv32_attr_done |= bit64;
            }
            if (residual) C32_LAUNCH_S(1, 64, 256); else C32_LAUNCH_S(0, 64, 256);
            SPA_LAUNCH_CHECK();
            return SPA_OK;
        }
        SPA_ARG(bn == (bm == 256 ? 256 : 128));
        if (residual) { if (bm == 256) C32_LAUNCH_S(1, 256, 256); else if (bm == 128) C32_LAUNCH_S(1, 128, 128); else C32_LAUNCH_S(1, 64, 128); }
        else { if (bm == 256) C32_LAUNCH_S(0, 256, 256); else if (bm == 128) C32_LAUNCH_S(0, 128, 128); else C32_LAUNCH_S(0, 64, 128); }
        SPA_LAUNCH_CHECK();
        return SPA_OK;
    }
#define C32_PICK(R)                                                                     \
    if (bm == 256) C32_LAUNCH(R, 256, 256);                                             \
    else if (bm == 128) { if (bn == 256) C32_LAUNCH(R, 128, 256); else C32_LAUNCH(R, 128, 128); } \
    else { if (bn == 256) C32_LAUNCH(R, 64, 256); else C32_LAUNCH(R, 64, 128); }
    if (residual) { C32_PICK(1) } else { C32_PICK(0) }
#undef C32_PICK
#undef C32_LAUNCH
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

extern "C" int spa_conv3x3_f32(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                               const float *wt, int32_t Cout, const float *bias, const float *residual,
                               int32_t relu, int32_t dilation, float *y, void *stream)
{
    return conv_f32_launch<9>(ctx, x, B, H, W, Cin, wt, Cout, bias, residual, relu, dilation, y, stream);
}

// the 1x1 stride-1 projection (wt (Cout, Cin)): same kernel, centre tap only
extern "C" int spa_conv1x1_f32(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                               const float *wt, int32_t Cout, const float *bias, const float *residual,
                               int32_t relu, float *y, void *stream)
{
    return conv_f32_launch<1>(ctx, x, B, H, W, Cin, wt, Cout, bias, residual, relu, 1, y, stream);
}

// the same two layers on the 16-bit matrix cores at float32 accuracy (template parameter SPLIT above): wt2 = the two-plane
// form of t * wt per 32-channel group, (Cout, taps, Cin/32, 2, 32) half precision; inv_t = 1 / t; amax_in / amax_out as in
// spa_conv3x3_wino4_f16s
extern "C" int spa_conv3x3_f16s(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                const void *wt2, float inv_t, int32_t Cout, const float *bias, const float *residual,
                                int32_t relu, int32_t dilation, const void *amax_in, void *amax_out, float *y, void *stream)
{
    SPA_ARG(amax_in);
    return conv_f32_launch<9>(ctx, x, B, H, W, Cin, (const float *)wt2, Cout, bias, residual, relu, dilation, y, stream, true, 1,
                              0, 0, 0, amax_in, amax_out, inv_t);
}

extern "C" int spa_conv1x1_f16s(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                const void *wt2, float inv_t, int32_t Cout, const float *bias, const float *residual,
                                int32_t relu, const void *amax_in, void *amax_out, float *y, void *stream)
{
    SPA_ARG(amax_in);
    return conv_f32_launch<1>(ctx, x, B, H, W, Cin, (const float *)wt2, Cout, bias, residual, relu, 1, y, stream, true, 1,
                              0, 0, 0, amax_in, amax_out, inv_t);
}

// 3x3 stride-2 padding-1 convolution (the first convolution of layers 3 and 4, models/drn.py:204-206) on the 16-bit matrix
// cores at float32 accuracy, optionally together with the block's 1x1 stride-2 projection as output channels
// [csplit, Cout) (wt2 rows csplit.. hold the projection's weights at the centre tap, zeros elsewhere): one pass over the
// input, bias / ReLU fused, the output's maximum tracked.  x (B,Hi,Wi,Cin), y (B,Ho,Wo,csplit), y2 (B,Ho,Wo,Cout - csplit)
// or NULL (then csplit = Cout), Ho = (Hi + 1) / 2, Wo = (Wi + 1) / 2; Cout % 128 == 0, csplit % 64 == 0.
extern "C" int spa_conv3x3_s2_f16s(spa_ctx *ctx, const float *x, int32_t B, int32_t Hi, int32_t Wi, int32_t Cin,
                                   const void *wt2, float inv_t, int32_t Cout, int32_t csplit, const float *bias,
                                   int32_t relu, const void *amax_in, void *amax_out, float *y, float *y2, void *stream)
{
    SPA_ARG(ctx && x && wt2 && bias && y && amax_in && B > 0 && Hi > 0 && Wi > 0 && inv_t > 0.f);
    SPA_ARG(Cin % C32_BK == 0 && Cout % 128 == 0 && csplit % 64 == 0 && csplit > 0 && csplit <= Cout && (y2 || csplit == Cout));
    SPA_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)wt2 % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)y2 % 16) == 0 && ((uintptr_t)bias % 16) == 0);
    hipStream_t s = spa_stream(stream);
    char *zero;
    int rc = spa_ws_reserve(ctx, WS_ZERO_LINE, 4096, (void **)&zero);
    if (rc != SPA_OK) return rc;
    if (!ctx->zero_line_ready) {
        SPA_HIP(hipMemsetAsync(zero, 0, 4096, s));
        ctx->zero_line_ready = 1;
    }
    if (amax_out) spa_zero_word(amax_out, s);
    const int H = (Hi + 1) / 2, W = (Wi + 1) / 2;
    // (64-pixel tiles, two workgroups per CU for the 128-row form, were measured: 2.06 vs 1.95 ms on the 32 -> 64+64 layer)
    const int bm = Cout % 256 == 0 ? 256 : 128, bn = 128;
    const int xtiles = (W + bn - 1) / bn, ntiles = Cout / bm;
    const long long total = (long long)B * H * xtiles * ntiles;
    SPA_ARG(total < (1ll << 31));
    SPA_ARG((long long)(2 * W + 1024) * Cin * 4 < (1ll << 31));            // 32-bit byte offsets inside an image row
    // the stride-2 pixel segment is staged in 16-row blocks: 34 blocks of 8 rows for the 2 * 128 + 8 rows a tile touches
    const size_t lds = 2 * (size_t)bm * 128 + 2 * (size_t)(2 * bn + 2 * C32_HALO + 8) * 128;
    if (!(ctx->conv32_attr_done & 4)) {
        SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_f32<0, 256, 9, 128, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 128 + 2 * (256 + 2 * C32_HALO + 8) * 128));
        SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_f32<0, 128, 9, 128, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * 128 + 2 * (256 + 2 * C32_HALO + 8) * 128));
        ctx->conv32_attr_done |= 4;
    }
    SpaProfScope prof_(ctx, PROF_DRN_CONV16_FRONT, s);
    const int per_cu = lds > 80 * 1024 ? 1 : (lds > 53 * 1024 ? 2 : 3);
    long long grid = (long long)ctx->n_cu * per_cu;
    if (grid > total) grid = total;
    if (bm == 256)
        hipLaunchKernelGGL((k_conv3x3_f32<0, 256, 9, 128, true, 2>), dim3((unsigned)grid), dim3(C32_THREADS), lds, s, x, (const float *)wt2, bias,
                           (const float *)nullptr, y, (const char *)zero, B, H, W, Cin, Cout, 1, relu, xtiles, ntiles, (int)total, 1, 0ll, 0ll, 0ll, 0,
                           (const unsigned *)amax_in, (unsigned *)amax_out, inv_t, Hi, Wi, y2, csplit);
    else
        hipLaunchKernelGGL((k_conv3x3_f32<0, 128, 9, 128, true, 2>), dim3((unsigned)grid), dim3(C32_THREADS), lds, s, x, (const float *)wt2, bias,
                           (const float *)nullptr, y, (const char *)zero, B, H, W, Cin, Cout, 1, relu, xtiles, ntiles, (int)total, 1, 0ll, 0ll, 0ll, 0,
                           (const unsigned *)amax_in, (unsigned *)amax_out, inv_t, Hi, Wi, y2, csplit);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// The same stride-2 opener (+ projection) with float32 matrix instructions (round 6: the strict float32 network, `--fp32_mfma_gemm`,
// no longer hands these layers to the library): the non-split form of the kernel, wt (Cout, 9, Cin) float32 with the projection's rows
// holding its weights at tap 4; no scales, no tracked maximum.
extern "C" int spa_conv3x3_s2_f32(spa_ctx *ctx, const float *x, int32_t B, int32_t Hi, int32_t Wi, int32_t Cin,
                                  const float *wt, int32_t Cout, int32_t csplit, const float *bias,
                                  int32_t relu, float *y, float *y2, void *stream)
{
    SPA_ARG(ctx && x && wt && bias && y && B > 0 && Hi > 0 && Wi > 0);
    SPA_ARG(Cin % C32_BK == 0 && Cout % 128 == 0 && csplit % 64 == 0 && csplit > 0 && csplit <= Cout && (y2 || csplit == Cout));
    SPA_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)wt % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)y2 % 16) == 0 && ((uintptr_t)bias % 16) == 0);
    hipStream_t s = spa_stream(stream);
    char *zero;
    int rc = spa_ws_reserve(ctx, WS_ZERO_LINE, 4096, (void **)&zero);
    if (rc != SPA_OK) return rc;
    if (!ctx->zero_line_ready) {
        SPA_HIP(hipMemsetAsync(zero, 0, 4096, s));
        ctx->zero_line_ready = 1;
    }
    const int H = (Hi + 1) / 2, W = (Wi + 1) / 2;
    const int bm = Cout % 256 == 0 ? 256 : 128, bn = 128;
    const int xtiles = (W + bn - 1) / bn, ntiles = Cout / bm;
    const long long total = (long long)B * H * xtiles * ntiles;
    SPA_ARG(total < (1ll << 31));
    SPA_ARG((long long)(2 * W + 1024) * Cin * 4 < (1ll << 31));
    const size_t lds = 2 * (size_t)bm * 128 + 2 * (size_t)(2 * bn + 2 * C32_HALO) * 128;
    if (!(ctx->conv32_attr_done & 128)) {
        SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_f32<0, 256, 9, 128, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 128 + 2 * (256 + 2 * C32_HALO) * 128));
        SPA_HIP(hipFuncSetAttribute((const void *)k_conv3x3_f32<0, 128, 9, 128, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * 128 + 2 * (256 + 2 * C32_HALO) * 128));
        ctx->conv32_attr_done |= 128;
    }
    SpaProfScope prof_(ctx, PROF_DRN_CONV32, s);
    const int per_cu = lds > 80 * 1024 ? 1 : (lds > 53 * 1024 ? 2 : 3);
    long long grid = (long long)ctx->n_cu * per_cu;
    if (grid > total) grid = total;
    if (bm == 256)
        hipLaunchKernelGGL((k_conv3x3_f32<0, 256, 9, 128, false, 2>), dim3((unsigned)grid), dim3(C32_THREADS), lds, s, x, wt, bias,
                           (const float *)nullptr, y, (const char *)zero, B, H, W, Cin, Cout, 1, relu, xtiles, ntiles, (int)total, 1, 0ll, 0ll, 0ll, 0,
                           (const unsigned *)nullptr, (unsigned *)nullptr, 1.f, Hi, Wi, y2, csplit);
    else
        hipLaunchKernelGGL((k_conv3x3_f32<0, 128, 9, 128, false, 2>), dim3((unsigned)grid), dim3(C32_THREADS), lds, s, x, wt, bias,
                           (const float *)nullptr, y, (const char *)zero, B, H, W, Cin, Cout, 1, relu, xtiles, ntiles, (int)total, 1, 0ll, 0ll, 0ll, 0,
                           (const unsigned *)nullptr, (unsigned *)nullptr, 1.f, Hi, Wi, y2, csplit);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// plain GEMMs for the Winograd path (spa_wino.hip): y (rows, Cout) = x (rows, Cin) . wt^T, wt (Cout, Cin); rows is a
// multiple of 256 — the 1x1 form of the kernel on x seen as an image of 256-pixel rows, zero bias, no activation
int conv1x1_f32_raw(spa_ctx *ctx, const float *x, long long rows, int32_t Cin, const float *wt, int32_t Cout,
                    float *y, void *stream, int zcount)
{
    SPA_ARG(rows > 0 && rows % 256 == 0 && rows / 256 < (1ll << 31) && zcount >= 1);
    // zcount problems in one launch: operands rows * Cin / Cout * Cin / rows * Cout elements apart
    return conv_f32_launch<1>(ctx, x, 1, (int32_t)(rows / 256), 256, Cin, wt, Cout, nullptr, nullptr, 0, 1, y, stream, true,
                              zcount, rows * Cin, (long long)Cout * Cin, rows * Cout);
}
