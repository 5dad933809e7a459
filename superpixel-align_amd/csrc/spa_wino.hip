// Winograd F(2x2, 3x3) for the heavy 3x3 (dilated) stride-1 layers of the float32 DRN (models/drn.py:230-285:
// layers 5-8, 256/512 channels at 1/8 resolution).  The float32 matrix pipe is the wall of the float32 network
// (spa_conv3x3_f32 and MIOpen both sit at 0.88 of its peak), so the only way past it is fewer multiplications:
// the minimal filtering algorithm computes a 2x2 output tile from a 4x4 input tile with 16 instead of 36
// multiplications per (input channel, output channel) — 2.25x less matrix work — at the price of two streaming
// transforms.  float32 throughout; measured against a float64 convolution the result is as close as the direct
// float32 one (5e-7 vs 3e-7 of the map's scale: the transforms of F(2x2,3x3) only add and halve).
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A        per 2x2 output tile, d = its 4x4 input tile
//
// A dilated convolution is d*d independent ordinary convolutions on the sub-grids (y mod d, x mod d); tiles are
// cut on the sub-grids, so dilation only changes addresses.
//
//   k_wino_in    X (B,H,W,C) -> V [16][T][C]   (T tiles; position-major, so every GEMM reads one dense matrix)
//   GEMM         M[p] (T x K) = V[p] (T x C) . U[p]^T,  U[p] = (G g G^T)[p] as (K, C): the float32 MFMA kernel of
//                spa_conv32.hip in its 1x1 form (V[p] seen as an image of 256-"pixel" rows), all 16 in one
//                launch of persistent workgroups
//   k_wino_out   M [16][T][K] -> Y (B,H,W,K) with bias, residual and ReLU
//
// HBM traffic per layer: X once, V written and read (4x X), M written and read (4x Y), Y once — 38 GB per 30
// images of a 512 -> 512 layer, ~5 ms of streaming next to a GEMM of 16 ms, against 33 ms of direct convolution.
#include "spa_common.h"
#include "spa_wino_dev.h"
#include <stdlib.h>

int conv1x1_f32_raw(spa_ctx *ctx, const float *x, long long rows, int32_t Cin, const float *wt, int32_t Cout,
                    float *y, void *stream, int zcount);          // spa_conv32.hip


__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// workgroup ids go round-robin over the 8 XCDs; give every XCD (= every L2) a contiguous range of tiles, so that the
// rows two vertically adjacent tiles share are fetched into one L2, not eight (PMC: k_wino_in read 2.2x its input)
__device__ __forceinline__ long long wino_block()
{
    const long long nb = gridDim.x, id = blockIdx.x;
    const long long q = nb / 8, rem = nb % 8, xcd = id % 8, idx = id / 8;
    return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
}


// one thread = one tile x 4 channels: 16 float4 loads, B^T d B, 16 float4 stores
__global__ __launch_bounds__(256) void k_wino_in(const float *__restrict__ X, float *__restrict__ V, WinoGeom g, int C,
                                                 long long Tpad)
{
    const int c4 = C >> 2;
    const long long id = wino_block() * 256 + threadIdx.x;
    if (id >= g.T * c4) return;
    const long long t = id / c4;
    const int c = (int)(id - t * c4) << 2;
    int b, sy, sx, ty, tx;
    wino_tile(g, t, b, sy, sx, ty, tx);
    float4 dv[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = sy + (2 * ty - 1 + i) * g.d;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = sx + (2 * tx - 1 + j) * g.d;
            const bool ok = y >= 0 && y < g.H && x >= 0 && x < g.W;
            dv[i][j] = ok ? *(const float4 *)(X + (((long long)b * g.H + y) * g.W + x) * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // rows: B^T d
    float4 r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[0][j] = f4sub(dv[0][j], dv[2][j]);
        r[1][j] = f4add(dv[1][j], dv[2][j]);
        r[2][j] = f4sub(dv[2][j], dv[1][j]);
        r[3][j] = f4sub(dv[1][j], dv[3][j]);
    }
    // columns: (B^T d) B
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 v0 = f4sub(r[i][0], r[i][2]), v1 = f4add(r[i][1], r[i][2]);
        const float4 v2 = f4sub(r[i][2], r[i][1]), v3 = f4sub(r[i][1], r[i][3]);
        float *o = V + ((long long)(i * 4) * Tpad + t) * C + c;
        *(float4 *)(o) = v0;
        *(float4 *)(o + Tpad * C) = v1;
        *(float4 *)(o + 2 * Tpad * C) = v2;
        *(float4 *)(o + 3 * Tpad * C) = v3;
    }
}

// one thread = one tile x 4 output channels: 16 float4 loads, A^T m A, epilogue, up to 4 float4 stores
template <int HAS_RES>
__global__ __launch_bounds__(256) void k_wino_out(const float *__restrict__ M, float *__restrict__ Y,
                                                  const float *__restrict__ bias, const float *__restrict__ R,
                                                  WinoGeom g, int K, long long Tpad, int relu)
{
    const int k4 = K >> 2;
    const long long id = wino_block() * 256 + threadIdx.x;
    if (id >= g.T * k4) return;
    const long long t = id / k4;
    const int k = (int)(id - t * k4) << 2;
    int b, sy, sx, ty, tx;
    wino_tile(g, t, b, sy, sx, ty, tx);
    float4 m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[i][j] = *(const float4 *)(M + ((long long)(i * 4 + j) * Tpad + t) * K + k);
    // rows: A^T m  (2 x 4)
    float4 s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[0][j] = f4add(f4add(m[0][j], m[1][j]), m[2][j]);
        s[1][j] = f4sub(f4sub(m[1][j], m[2][j]), m[3][j]);
    }
    const float4 bv = *(const float4 *)(bias + k);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int y = sy + (2 * ty + i) * g.d;
        if (y >= g.H) continue;
        const float4 o0 = f4add(f4add(s[i][0], s[i][1]), s[i][2]);
        const float4 o1 = f4sub(f4sub(s[i][1], s[i][2]), s[i][3]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int x = sx + (2 * tx + j) * g.d;
            if (x >= g.W) continue;
            float4 v = f4add(j == 0 ? o0 : o1, bv);
            const long long off = (((long long)b * g.H + y) * g.W + x) * K + k;
            if (HAS_RES) v = f4add(v, *(const float4 *)(R + off));
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *(float4 *)(Y + off) = v;
        }
    }
}

static void wino_geom(int B, int H, int W, int d, WinoGeom *g)
{
    g->B = B; g->H = H; g->W = W; g->d = d;
    const int hs = (H + d - 1) / d, ws = (W + d - 1) / d;         // the largest sub-grid
    g->th = (hs + 1) / 2; g->tw = (ws + 1) / 2;
    g->T = (long long)B * d * d * g->th * g->tw;
}

// rows of V / M per position, padded to whole 256-row GEMM tiles
extern "C" int64_t spa_wino_tiles(int32_t B, int32_t H, int32_t W, int32_t dilation)
{
    WinoGeom g;
    wino_geom(B, H, W, dilation, &g);
    return (g.T + 255) / 256 * 256;
}

// x (B,H,W,Cin) float32 channels-last; u (16,Cout,Cin) float32 = (G g G^T)[4i+j] per (output, input) channel;
// v_scratch 16 * spa_wino_tiles * Cin floats, m_scratch 16 * spa_wino_tiles * Cout floats (caller-owned: calls on
// different streams do not share them); otherwise as spa_conv3x3_f32
extern "C" int spa_conv3x3_wino_f32(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                    const float *u, int32_t Cout, const float *bias, const float *residual,
                                    int32_t relu, int32_t dilation, float *v_scratch, float *m_scratch, float *y,
                                    void *stream)
{
    SPA_ARG(ctx && x && u && bias && y && v_scratch && m_scratch && B > 0 && H > 0 && W > 0 && dilation >= 1);
    SPA_ARG(Cin % 32 == 0 && Cout % 64 == 0);
    SPA_ARG((((uintptr_t)x | (uintptr_t)u | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)v_scratch |
              (uintptr_t)m_scratch) % 16) == 0);
    hipStream_t s = spa_stream(stream);
    WinoGeom g;
    wino_geom(B, H, W, dilation, &g);
    const long long Tpad = (g.T + 255) / 256 * 256;
    SPA_ARG(g.T * (Cin > Cout ? Cin : Cout) / 4 < (1ll << 31) * 256);
    {
        SpaProfScope prof_(ctx, PROF_WINO_IN, s);
        const long long n = g.T * (Cin / 4);
        hipLaunchKernelGGL(k_wino_in, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, v_scratch, g, Cin, Tpad);
    }
    // (rows T .. Tpad of V are never written: their products land in rows of M nothing reads)
    {
        // the 16 GEMMs as ONE launch of persistent workgroups
        int rc = conv1x1_f32_raw(ctx, v_scratch, Tpad, Cin, u, Cout, m_scratch, stream, 16);
        if (rc != SPA_OK) return rc;
    }
    {
        SpaProfScope prof_(ctx, PROF_WINO_OUT, s);
        const long long n = g.T * (Cout / 4);
        if (residual)
            hipLaunchKernelGGL(k_wino_out<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)m_scratch, y,
                               bias, residual, g, Cout, Tpad, relu);
        else
            hipLaunchKernelGGL(k_wino_out<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)m_scratch, y,
                               bias, residual, g, Cout, Tpad, relu);
    }
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// =========================================================================================================
// F(4x4, 3x3): a 4x4 output tile from a 6x6 input tile with 36 instead of 144 multiplications per channel pair —
// 4x less matrix work than the direct form, V and M 2.25x the activations instead of 4x.  Interpolation points
// 0, 1, -1, 1/2, -2, infinity (NOT the textbook 0, +-1, +-2: measured in float32 on a 256 -> 256 layer the textbook set
// is 1.0e-5 of the map's scale from a float64 convolution, this set 4.7e-6; through the whole DRN-D-22 the final map is
// 1.1e-6 from the float64 network against 2.4e-6 for direct float32 convolutions and 0.8e-6 for F(2x2,3x3), tools/wino_points.py).
//
//   B^T = | 2 -3 -4  3  2  0 |     A^T = | 1  1  1   1    1  0 |     G = | 1/2    0     0   |
//         | 0 -2  1  5  2  0 |           | 0  1 -1  1/2  -2  0 |         | 1/6   1/6   1/6  |
//         | 0 -2  5 -1 -2  0 |           | 0  1  1  1/4   4  0 |         | 1/6  -1/6   1/6  |
//         | 0  2  1 -2 -1  0 |           | 0  1 -1  1/8  -8  1 |         | 16/15 8/15  4/15 |
//         | 0  1 -2 -1  2  0 |                                           | 1/30 -1/15  2/15 |
//         | 0  2 -3 -4  3  2 |                                           |  0     0    1/2  |
// (row 3 of B^T is the generated row / 16 and row 3 of G x 16: powers of two, no rounding.)
// A thread owns one tile x 2 channels (float2): 36 values in flight.
// =========================================================================================================

// (numbering the F(4x4) tiles in blocks of 4 x 4 tiles, so that consecutive workgroups transform tiles whose 6 x 6
// input patches overlap in both directions, was measured: no change — the overlap is served by L2 / MALL anyway —
// and it padded small sub-grids to multiples of 4 tiles: 4x the work on a 7 x 7 sub-grid.  Row-major it is.)

template <typename V>
__global__ __launch_bounds__(256) void k_wino4_in(const float *__restrict__ X, float *__restrict__ Vout, WinoGeom g, int C,
                                                  long long Tpad)
{
    constexpr int VN = sizeof(V) / 4;
    const int c2 = C / VN;
    const long long id = wino_block() * 256 + threadIdx.x;
    if (id >= g.T * c2) return;
    const long long t = id / c2;
    const int c = (int)(id - t * c2) * VN;
    int b, sy, sx, ty, tx;
    wino_tile(g, t, b, sy, sx, ty, tx);
    // columns first: r[a][:] = (row a of d) . B  (6 values), then rows: out[i][j] = sum_a Bt[i][a] r[a][j]
    V r[6][6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const int y = sy + (4 * ty - 1 + a) * g.d;
        V dv[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int x = sx + (4 * tx - 1 + j) * g.d;
            const bool ok = y >= 0 && y < g.H && x >= 0 && x < g.W;
            dv[j] = ok ? *(const V *)(X + (((long long)b * g.H + y) * g.W + x) * C + c) : wino_zero<V>();
        }
        wino4_bt(dv, r[a]);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const V col[6] = {r[0][j], r[1][j], r[2][j], r[3][j], r[4][j], r[5][j]};
        V o[6];
        wino4_bt(col, o);
#pragma unroll
        // (non-temporal stores of V were measured: 1.60 instead of 1.31 ms per 30 images of a 512-channel layer)
        for (int i = 0; i < 6; ++i) *(V *)(Vout + ((long long)(i * 6 + j) * Tpad + t) * C + c) = o[i];
    }
}

// The same transform with the input patch of a block of tiles staged in LDS — an experiment that is NOT the default
// (SPA_WINO_IN_LDS=2|4 selects it).  k_wino4_in reads every input value 36 / 16 = 2.25 times (the 6 x 6 patches of
// neighbouring tiles overlap by two); the repeats hit in L2 but cross the compute unit's vector memory path all the
// same, and the question was whether that path bounds the kernel.  Here a workgroup of TBY waves owns TBY x 8 tiles x
// 32 channels of one sub-grid: the (4 TBY + 2) x 34 pixel patch x 128 B goes to LDS once (1.2 - 1.33 reads per value),
// every thread (one tile x 4 channels) reads its 36 float4 from there.  A wave holds 2 x 4 tiles x 8 channel quads;
// the row pitch (34 x 128 + 32 B) moves a tile row by half a 256-byte bank row, which makes the 16-lane groups of
// ds_read_b128 conflict-free.  V is the one k_wino4_in<float4> writes, bit for bit.
// Measured (30 images, same box, profiles/r6_wino_in_lds.txt): 512 channels 1.34-1.59 ms against 1.39-1.55, 256 channels
// 0.65-0.67 against 0.69-0.71: within the box's run-to-run spread.  Writing the 36 values of a tile side by side
// instead of 36 planes 125 MB apart (timing only, the GEMM cannot read that) gave 1.24-1.48.  So neither the repeated
// loads nor the scatter of the stores is the limit: 2 GB read + 4.5 GB written in 1.39 ms is 4.7 TB/s, and that is
// what this part's HBM gives a write-heavy stream (6.8 TB/s for stores alone, 6.1 for one read per write).
constexpr int WIL_PW = 34, WIL_PITCH = WIL_PW * 128 + 32;

template <int TBY>
__global__ __launch_bounds__(64 * TBY) void k_wino4_in_lds(const float *__restrict__ X, float *__restrict__ Vout, WinoGeom g, int C,
                                                           long long Tpad, int nby, int nbx)
{
    constexpr int NT = 64 * TBY, PH = 4 * TBY + 2, ITEMS = PH * WIL_PW * 8, NIT = (ITEMS + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char wil_lds[];
    const int ncb = C >> 5;
    long long blk = wino_block();
    const int cb = (int)(blk % ncb); blk /= ncb;
    const int bx = (int)(blk % nbx); blk /= nbx;
    const int by = (int)(blk % nby); blk /= nby;
    const int sx = (int)(blk % g.d); blk /= g.d;
    const int sy = (int)(blk % g.d);
    const int b = (int)(blk / g.d);
    const int tid = threadIdx.x;
    {
        const float *xb = X + (long long)b * g.H * g.W * C + cb * 32;
        float4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = it * NT + tid, px = item >> 3, q = item & 7;
            const int pu = px / WIL_PW, pv = px - pu * WIL_PW;
            const int y = sy + (4 * TBY * by - 1 + pu) * g.d, x = sx + (32 * bx - 1 + pv) * g.d;
            const bool ok = item < ITEMS && y >= 0 && y < g.H && x >= 0 && x < g.W;
            v[it] = ok ? *(const float4 *)(xb + ((long long)y * g.W + x) * C + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = it * NT + tid, px = item >> 3, q = item & 7;
            const int pu = px / WIL_PW, pv = px - pu * WIL_PW;
            if (item < ITEMS) *(float4 *)(wil_lds + pu * WIL_PITCH + pv * 128 + q * 16) = v[it];
        }
    }
    __syncthreads();
    const int w = tid >> 6, tl = (tid >> 3) & 7, q = tid & 7;
    const int ltx = 4 * (w & 1) + (tl & 1) + 2 * (tl >> 2), lty = 2 * (w >> 1) + ((tl >> 1) & 1);
    const int ty = TBY * by + lty, tx = 8 * bx + ltx;
    if (ty >= g.th || tx >= g.tw) return;
    const long long t = ((((long long)b * g.d + sy) * g.d + sx) * g.th + ty) * g.tw + tx;
    const int c = cb * 32 + q * 4;
    const unsigned char *lp = wil_lds + (4 * lty) * WIL_PITCH + (4 * ltx) * 128 + q * 16;
    float4 r[6][6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        float4 dv[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) dv[j] = *(const float4 *)(lp + a * WIL_PITCH + j * 128);
        wino4_bt(dv, r[a]);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float4 col[6] = {r[0][j], r[1][j], r[2][j], r[3][j], r[4][j], r[5][j]};
        float4 o[6];
        wino4_bt(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) *(float4 *)(Vout + ((long long)(i * 6 + j) * Tpad + t) * C + c) = o[i];
    }
}

// SPA_WINO_IN_LDS: 2 / 4 = k_wino4_in_lds<2 / 4>; anything else (the default) = k_wino4_in
static int wino4_in_lds_choice(int C)
{
    static const int env = getenv("SPA_WINO_IN_LDS") ? atoi(getenv("SPA_WINO_IN_LDS")) : 0;
    return C % 32 == 0 && (env == 2 || env == 4) ? env : 0;
}

// launches k_wino4_in_lds<TBY> (tby = 2 or 4)
static void wino4_in_lds_launch(spa_ctx *ctx, int tby, const float *x, float *v, const WinoGeom &g, int C, long long Tpad, hipStream_t s)
{
    const int nby = (g.th + tby - 1) / tby, nbx = (g.tw + 7) / 8;
    const long long nb = (long long)g.B * g.d * g.d * nby * nbx * (C / 32);
    const size_t lds = (size_t)(4 * tby + 2) * WIL_PITCH;
    if (!(ctx->conv32_attr_done & 256)) {        // per context = per device
        (void)hipFuncSetAttribute((const void *)k_wino4_in_lds<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 18 * WIL_PITCH);
        (void)hipFuncSetAttribute((const void *)k_wino4_in_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * WIL_PITCH);
        ctx->conv32_attr_done |= 256;
    }
    if (tby == 4) hipLaunchKernelGGL(k_wino4_in_lds<4>, dim3((unsigned)nb), dim3(256), lds, s, x, v, g, C, Tpad, nby, nbx);
    else hipLaunchKernelGGL(k_wino4_in_lds<2>, dim3((unsigned)nb), dim3(128), lds, s, x, v, g, C, Tpad, nby, nbx);
}

template <int HAS_RES, typename V>
__global__ __launch_bounds__(256) void k_wino4_out(const float *__restrict__ M, float *__restrict__ Y,
                                                   const float *__restrict__ bias, const float *__restrict__ R,
                                                   WinoGeom g, int K, long long Tpad, int relu)
{
    constexpr int VN = sizeof(V) / 4;
    const int k2 = K / VN;
    const long long id = wino_block() * 256 + threadIdx.x;
    if (id >= g.T * k2) return;
    const long long t = id / k2;
    const int k = (int)(id - t * k2) * VN;
    int b, sy, sx, ty, tx;
    wino_tile(g, t, b, sy, sx, ty, tx);
    // rows: s[:][j] = A^T m[:][j] (4 x 6), then columns
    V s[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        V col[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = *(const V *)(M + ((long long)(i * 6 + j) * Tpad + t) * K + k);
        V o[4];
        wino4_at(col, o);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i][j] = o[i];
    }
    const V bv = *(const V *)(bias + k);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = sy + (4 * ty + i) * g.d;
        V o[4];
        wino4_at(s[i], o);
        if (y >= g.H) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = sx + (4 * tx + j) * g.d;
            if (x >= g.W) continue;
            V v = o[j] + bv;
            const long long off = (((long long)b * g.H + y) * g.W + x) * K + k;
            if (HAS_RES) v = v + *(const V *)(R + off);
            if (relu) v = wino_relu(v);
            *(V *)(Y + off) = v;
        }
    }
}

// =========================================================================================================
// F(4x4, 3x3) with the GEMMs on the 16-bit matrix cores at float32 accuracy (spa_gemm16.hip): the GEMM kernel splits V
// (float32, written by k_wino4_in as for the float32 GEMMs) into two half-precision planes of an exactly (power-of-two)
// scaled value while it feeds the matrix cores.
//   scale of position (i, j):  2^(14 - e - p_i - p_j),  e = exponent of amax >= max |x| over the layer input,
//   p = ceil(log2(row sums of |B^T|)) = 4 4 4 3 3 4  ->  |V_ij| <= 2^(p_i + p_j) amax, scaled magnitude < 2^15:
//   half precision cannot overflow, and what it loses at the small end is 2^-25 of that bound in absolute terms
//   (2^-40 of the layer's largest activation) — its 40 binades are enough for float32-class accuracy of the sums.
// amax comes from the producer of x (k_wino4_out tracks the maximum of what it stores; k_amax for other producers);
// the weights' planes carry a static per-position scale t_ij, and k_wino4_out undoes both (cs[i][j] = 2^(p_i+p_j) / t_ij
// from the host, 2^(e - 14) from amax) while it reads M.
// =========================================================================================================

// amax[0] = bits of max |x| (as unsigned: non-negative floats order like their bit patterns)
__global__ __launch_bounds__(256) void k_amax(const float4 *__restrict__ x, long long n4, unsigned *__restrict__ amax)
{
    __shared__ unsigned red[4];
    unsigned m = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = x[i];
        m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu,
                max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)));
    }
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(red[0], red[1]), max(red[2], red[3]));
        if (m > *(volatile unsigned *)amax) atomicMax(amax, m);
    }
}

// k_wino4_out for the scaled GEMMs: M[i][j] * cs[i][j] * 2^(e - 14); optionally the maximum of what it stores
template <int HAS_RES, typename V = float4>
__global__ __launch_bounds__(256, 2) void k_wino4_out_s(const float *__restrict__ M, float *__restrict__ Y,
                                                     const float *__restrict__ bias, const float *__restrict__ R,
                                                     WinoGeom g, int K, long long Tpad, int relu, WinoScale cs,
                                                     const unsigned *__restrict__ amax_in, unsigned *__restrict__ amax_out)
{
    constexpr int VN = sizeof(V) / 4;       // channels per thread: 4 (the default) or 2 (SPA_WINO_VEC2: half the registers, round 6's co-residency probe)
    const int k2 = K / VN;
    const long long id = wino_block() * 256 + threadIdx.x;
    unsigned mx = 0;
    if (id < g.T * k2) {
        const long long t = id / k2;
        const int k = (int)(id - t * k2) * VN;
        int b, sy, sx, ty, tx;
        wino_tile(g, t, b, sy, sx, ty, tx);
        const float inv = wino_pow2(wino_amax_exp(*amax_in) - 14);
        V s[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            V col[6];
#pragma unroll
            for (int i = 0; i < 6; ++i)
                // M is read exactly once: non-temporal (1.66 instead of 1.70 ms per 30 images of a 512-channel layer)
                col[i] = (cs.c[i * 6 + j] * inv) * wino_nt_load((const V *)(M + ((long long)(i * 6 + j) * Tpad + t) * K + k));
            V o[4];
            wino4_at(col, o);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i][j] = o[i];
        }
        const V bv = *(const V *)(bias + k);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = sy + (4 * ty + i) * g.d;
            V o[4];
            wino4_at(s[i], o);
            if (y >= g.H) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = sx + (4 * tx + j) * g.d;
                if (x >= g.W) continue;
                V v = o[j] + bv;
                const long long off = (((long long)b * g.H + y) * g.W + x) * K + k;
                if (HAS_RES) v = v + *(const V *)(R + off);
                if (relu) v = wino_relu(v);
                *(V *)(Y + off) = v;
                mx = max(mx, wino_absmax_bits(v));
            }
        }
    }
    if (amax_out) {
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
        if ((threadIdx.x & 63) == 0 && mx > *(volatile unsigned *)amax_out) atomicMax(amax_out, mx);
    }
}

extern "C" int64_t spa_wino4_tiles(int32_t B, int32_t H, int32_t W, int32_t dilation)
{
    WinoGeom g;
    wino4_geom(B, H, W, dilation, &g);
    return (g.T + 255) / 256 * 256;
}

// as spa_conv3x3_wino_f32 with u (36,Cout,Cin) = (G g G^T)[6i+j] of the matrices above and scratch of
// 36 * spa_wino4_tiles rows
extern "C" int spa_conv3x3_wino4_f32(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                     const float *u, int32_t Cout, const float *bias, const float *residual,
                                     int32_t relu, int32_t dilation, float *v_scratch, float *m_scratch, float *y,
                                     void *stream)
{
    SPA_ARG(ctx && x && u && bias && y && v_scratch && m_scratch && B > 0 && H > 0 && W > 0 && dilation >= 1);
    SPA_ARG(Cin % 32 == 0 && Cout % 64 == 0);
    SPA_ARG((((uintptr_t)x | (uintptr_t)u | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)v_scratch |
              (uintptr_t)m_scratch) % 16) == 0);
    hipStream_t s = spa_stream(stream);
    WinoGeom g;
    wino4_geom(B, H, W, dilation, &g);
    const long long Tpad = (g.T + 255) / 256 * 256;
    SPA_ARG(g.T * (Cin > Cout ? Cin : Cout) / 2 < (1ll << 31) * 256);
    {
        SpaProfScope prof_(ctx, PROF_WINO_IN, s);
        // (measured per 512-channel launch, 30 images: 2 channels per thread 1.44 ms, 4 per thread see the header)
        if (getenv("SPA_WINO_VEC2")) {
            const long long n = g.T * (Cin / 2);
            hipLaunchKernelGGL(k_wino4_in<float2>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, v_scratch, g, Cin, Tpad);
        } else {
            const long long n = g.T * (Cin / 4);
            hipLaunchKernelGGL(k_wino4_in<float4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, v_scratch, g, Cin, Tpad);
        }
    }
    {
        int rc = conv1x1_f32_raw(ctx, v_scratch, Tpad, Cin, u, Cout, m_scratch, stream, 36);
        if (rc != SPA_OK) return rc;
    }
    {
        SpaProfScope prof_(ctx, PROF_WINO_OUT, s);
        if (getenv("SPA_WINO_VEC2")) {
            const long long n = g.T * (Cout / 2);
            if (residual)
                hipLaunchKernelGGL((k_wino4_out<1, float2>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)m_scratch, y,
                                   bias, residual, g, Cout, Tpad, relu);
            else
                hipLaunchKernelGGL((k_wino4_out<0, float2>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)m_scratch, y,
                                   bias, residual, g, Cout, Tpad, relu);
        } else {
            const long long n = g.T * (Cout / 4);
            if (residual)
                hipLaunchKernelGGL((k_wino4_out<1, float4>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)m_scratch, y,
                                   bias, residual, g, Cout, Tpad, relu);
            else
                hipLaunchKernelGGL((k_wino4_out<0, float4>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)m_scratch, y,
                                   bias, residual, g, Cout, Tpad, relu);
        }
    }
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

int gemm_f16x3_raw(spa_ctx *ctx, const float *x, long long rows, int32_t Cin, const void *wt, int32_t Cout, float *y,
                   void *stream, int zcount, const void *amax);          // spa_gemm16.hip

// amax[0] = bit pattern of max |x| over n floats (n a multiple of 4): the scale input of spa_conv3x3_wino4_f16s for a
// tensor whose producer did not track it
extern "C" int spa_amax_f32(spa_ctx *ctx, const float *x, int64_t n, void *amax, void *stream)
{
    SPA_ARG(ctx && x && amax && n > 0 && n % 4 == 0 && ((uintptr_t)x % 16) == 0);
    hipStream_t s = spa_stream(stream);
    spa_zero_word(amax, s);
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_amax, dim3((unsigned)blocks), dim3(256), 0, s, (const float4 *)x, (long long)(n / 4), (unsigned *)amax);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// F(4x4,3x3) with the 36 GEMMs on the 16-bit matrix cores at float32 accuracy.  u2: the two-plane form of the scaled
// (G g G^T)[6i+j] (36, Cout, Cin/32, 2, 32) half precision; cs: 36 floats 2^(p_i + p_j) / t_ij (host); amax_in: device
// word holding the bit pattern of a bound on max |x| (spa_amax_f32, or the amax_out of the call that produced x);
// amax_out: device word that receives the bound for y, or NULL.  The rest as spa_conv3x3_wino4_f32.
extern "C" int spa_conv3x3_wino4_f16s(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                      const void *u2, const float *cs, int32_t Cout, const float *bias,
                                      const float *residual, int32_t relu, int32_t dilation, const void *amax_in,
                                      void *amax_out, void *v_scratch, float *m_scratch, float *y, void *stream)
{
    SPA_ARG(ctx && x && u2 && cs && bias && y && v_scratch && m_scratch && amax_in && B > 0 && H > 0 && W > 0 && dilation >= 1);
    SPA_ARG(Cin % 32 == 0 && Cout % 128 == 0);
    SPA_ARG((((uintptr_t)x | (uintptr_t)u2 | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)v_scratch |
              (uintptr_t)m_scratch) % 16) == 0);
    hipStream_t s = spa_stream(stream);
    WinoGeom g;
    wino4_geom(B, H, W, dilation, &g);
    const long long Tpad = (g.T + 255) / 256 * 256;
    SPA_ARG(g.T * (Cin > Cout ? Cin : Cout) / 4 < (1ll << 31) * 256);
    WinoScale sc;
    for (int i = 0; i < 36; ++i) sc.c[i] = cs[i];
    if (amax_out) spa_zero_word(amax_out, s);
    static const int vec2 = getenv("SPA_WINO_VEC2") ? atoi(getenv("SPA_WINO_VEC2")) : 0;      // 1: out, 2: in, 3: both transforms 2 channels per thread
    {
        SpaProfScope prof_(ctx, PROF_WINO_IN, s);
        if (vec2 & 2) {
            const long long n = g.T * (Cin / 2);
            hipLaunchKernelGGL(k_wino4_in<float2>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, (float *)v_scratch, g, Cin, Tpad);
        } else if (const int tby = wino4_in_lds_choice(Cin)) {
            wino4_in_lds_launch(ctx, tby, x, (float *)v_scratch, g, Cin, Tpad, s);
        } else {
            const long long n = g.T * (Cin / 4);
            hipLaunchKernelGGL(k_wino4_in<float4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, (float *)v_scratch, g, Cin, Tpad);
        }
    }
    {
        int rc = gemm_f16x3_raw(ctx, (const float *)v_scratch, Tpad, Cin, u2, Cout, m_scratch, stream, 36, amax_in);
        if (rc != SPA_OK) return rc;
    }
    if (vec2 & 1) {
        SpaProfScope prof_(ctx, PROF_WINO_OUT, s);
        const long long n = g.T * (Cout / 2);
        if (residual)
            hipLaunchKernelGGL((k_wino4_out_s<1, float2>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)m_scratch, y,
                               bias, residual, g, Cout, Tpad, relu, sc, (const unsigned *)amax_in, (unsigned *)amax_out);
        else
            hipLaunchKernelGGL((k_wino4_out_s<0, float2>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)m_scratch, y,
                               bias, residual, g, Cout, Tpad, relu, sc, (const unsigned *)amax_in, (unsigned *)amax_out);
    } else {
        SpaProfScope prof_(ctx, PROF_WINO_OUT, s);
        const long long n = g.T * (Cout / 4);
        if (residual)
            hipLaunchKernelGGL((k_wino4_out_s<1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)m_scratch, y,
                               bias, residual, g, Cout, Tpad, relu, sc, (const unsigned *)amax_in, (unsigned *)amax_out);
        else
            hipLaunchKernelGGL((k_wino4_out_s<0>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)m_scratch, y,
                               bias, residual, g, Cout, Tpad, relu, sc, (const unsigned *)amax_in, (unsigned *)amax_out);
    }
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
