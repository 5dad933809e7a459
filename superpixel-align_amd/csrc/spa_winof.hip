// One Winograd F(4x4,3x3) layer of the float32 DRN as ONE persistent launch: input transform, the 36 GEMMs on the 16-bit
// matrix cores (two half-precision planes per float32 operand, spa_gemm16.hip) and output transform with bias / residual /
// ReLU (models/drn.py:23-57, 186-208, 230-285: the 256/512-channel layers 5-8).
//
// Why.  As three launches (spa_conv3x3_wino4_f16s) the transforms are pure HBM streaming — X -> V = 2.25 X, M = 2.25 Y -> Y,
// 21.5 ms of the 77 ms step at 4.5-4.8 TB/s — and run while the matrix cores idle; the GEMMs (23 ms) run while HBM idles at
// a third of its rate.  Fusing them at register level is impossible (a tile block needs the accumulators of all 36
// positions: 36 x 256 x 256 floats), so they are fused in TIME: the layer is cut into work items
//
//     IN (rb, part)   input transform of a slice of row block rb (256 tiles): X -> V[0..35][rows of rb]
//     MM (rb, cb, z)  one 256 x 256 GEMM tile: M[z][rows of rb][channels of cb] = V[z] . U[z]^T   (the body of k_gemm_f16x3)
//     OUT(rb, cb, part) output transform of a slice of (rb, cb): M[0..35] -> Y (+ bias, residual, ReLU, tracked maximum)
//
// kept in per-XCD lists that interleave the three kinds at the ratio of their work, and every workgroup (one per CU) pops
// the next item of its XCD's list (agent-scope atomic head; an empty list steals from the next XCD).  At any time most CUs
// are inside a GEMM tile and a few are streaming a transform slice at the per-CU rate a single streaming workgroup reaches
// (60-120 GB/s, MI355X_MICROARCH.md handoff-payload), so the transforms' HBM traffic is spread under the matrix work
// instead of being paid separately.  V and M keep their full-size position-major layout in HBM.
//
// Dependencies travel through counters: IN adds to in_cnt[rb], MM waits for in_cnt[rb] == parts and adds to mm_cnt[rb, cb],
// OUT waits for mm_cnt == 36.  A list orders every producer before its consumers, and an item is popped only by a running
// workgroup, so a waiting workgroup always waits for running ones: no residency assumption, no deadlock (the spin is
// bounded anyway and latches SPA_ST_WINO_SYNC).  Hand-off form (MI355X_MICROARCH.md, inter-workgroup visibility): V and M
// are stored write-through (`global_store_dwordx4 ... sc0 sc1`), every storing wave drains (`s_waitcnt vmcnt(0)`), workgroup
// barrier, ONE lane adds to the counter (agent scope); the consumer polls with an sc1 load, runs an agent acquire
// (buffer_inv sc1), waits for it, workgroup barrier, then plain loads (global_load_lds for the GEMM operands).  Correctness
// never depends on which XCD a workgroup runs on: the XCD id (HW_REG_XCC_ID) only selects the list, for L2 locality of the
// weights' planes (an XCD works on one position z of a window of row blocks at a time).
#include "spa_common.h"
#include "spa_wino_dev.h"
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

#define WF_THREADS 512
#define WF_IN 0u
#define WF_MM 1u
#define WF_OUT 2u
#define WF_NONE 0xffffffffu
// item word: type (2) | cb (2) | z or part (6) | rb (22)
#define WF_ITEM(type, cb, zp, rb) ((unsigned)(type) | ((unsigned)(cb) << 2) | ((unsigned)(zp) << 4) | ((unsigned)(rb) << 10))

struct WfParams {
    const float *X; float *V; float *M; float *Y; const char *U2; const float *bias; const float *R;
    WinoGeom g;
    int Cin, Cout, relu, PI, PO, NCB;
    long long Tpad;
    const unsigned *amax_in; unsigned *amax_out;
    const unsigned *items;        // [0..8]: offsets of the 8 lists (in items, relative to items + 16), then the lists
    unsigned *sync;               // [0..7] heads, [16..51] the 36 scale constants (float), [64 ..] in_cnt[NRB], then mm_cnt[NRB * NCB]
    int NRB;
    uint32_t *status;
};

__device__ __forceinline__ int wf_xcc_id()
{
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7;
}
// write-through 16-byte store: the bytes leave the XCD's L2 at once (what a consumer on any XCD may read after the counter).
// (uniform base, 32-bit byte offset per lane) as a buffer store the compiler sees — an inline-asm `global_store ... sc0 sc1`
// was tried first and stored stale registers: the hazard recogniser does not look inside inline asm (a VALU result read as
// the data of a 16-byte store, a v_readfirstlane'd base read as its address, need wait states).  aux 17 = sc0 | sc1.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wf_store_wt(float *base, unsigned off, f32x4 v)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)off, 0, 17);
}
__device__ __forceinline__ void wf_store_wt(float *base, unsigned off, float4 v) { wf_store_wt(base, off, (f32x4){v.x, v.y, v.z, v.w}); }

// pointers that travelled through a parameter block in memory are generic to the compiler: say "global" at every access
#define WF_G(T, ptr) ((__attribute__((address_space(1))) T *)(ptr))
__device__ __forceinline__ float4 wf_ld(const float *q)
{
    const wino_v4 v = *WF_G(const wino_v4, q);
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float4 wf_ld_nt(const float *q)
{
    const wino_v4 v = __builtin_nontemporal_load(WF_G(const wino_v4, q));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void wf_st(float *q, float4 v) { *WF_G(wino_v4, q) = (wino_v4){v.x, v.y, v.z, v.w}; }
// (uniform base, 32-bit byte offset): the form that compiles to `global_* v, v_off, s[base]` — no 64-bit address per access
__device__ __forceinline__ float4 wf_ld(const float *base, unsigned off) { return wf_ld((const float *)((const char *)base + off)); }
__device__ __forceinline__ float4 wf_ld_nt(const float *base, unsigned off) { return wf_ld_nt((const float *)((const char *)base + off)); }
__device__ __forceinline__ void wf_st(float *base, unsigned off, float4 v) { wf_st((float *)((char *)base + off), v); }
__device__ __forceinline__ unsigned wf_ld_u32(const unsigned *q) { return *WF_G(const unsigned, q); }

// lane 0 of the workgroup: wait until *cnt >= want (sc1 poll), bounded
__device__ __forceinline__ void wf_wait(const unsigned *cnt, unsigned want, uint32_t *status)
{
    unsigned spins = 0;
    while (__hip_atomic_load(WF_G(const unsigned, cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 21)) { __hip_atomic_fetch_or(WF_G(uint32_t, status), SPA_ST_WINO_SYNC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
}

// a uniform view of the parameter block inside a non-inlined function
struct WfView {
    const float *X; float *V; float *M; float *Y; const char *U2; const float *bias; const float *R;
    WinoGeom g;
    int Cin, Cout, relu, PI, PO, NCB, NRB;
    long long Tpad;
    const float *cs;
    const unsigned *amax_in; unsigned *amax_out;
    unsigned *sync;
    uint32_t *status;
};
#define WF_UNIFORM_PARAMS                                                                                                      \
    WinoGeom g_; g_.B = wf_uni(pr.g.B); g_.H = wf_uni(pr.g.H); g_.W = wf_uni(pr.g.W); g_.d = wf_uni(pr.g.d);                   \
    g_.th = wf_uni(pr.g.th); g_.tw = wf_uni(pr.g.tw); g_.T = (long long)wf_uni((int)pr.g.T);                                   \
    const WfView p = {wf_uni(pr.X), wf_uni(pr.V), wf_uni(pr.M), wf_uni(pr.Y), wf_uni(pr.U2), wf_uni(pr.bias), wf_uni(pr.R), g_, \
                      wf_uni(pr.Cin), wf_uni(pr.Cout), wf_uni(pr.relu), wf_uni(pr.PI), wf_uni(pr.PO), wf_uni(pr.NCB),          \
                      wf_uni(pr.NRB), (long long)wf_uni((int)pr.Tpad), (const float *)(wf_uni(pr.sync) + 16), wf_uni(pr.amax_in), wf_uni(pr.amax_out), \
                      wf_uni(pr.sync), wf_uni(pr.status)};

// arguments of a non-inlined function arrive in vector registers: tell the compiler what is wave-uniform
__device__ __forceinline__ int wf_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T> __device__ __forceinline__ T *wf_uni(T *q)
{
    const unsigned long long a = (unsigned long long)q;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}

// the three item bodies are separate (not inlined) functions: each gets the register file to itself
__device__ __noinline__ void wf_item_in(const WfParams &pr, int rb, int zp)
{
    WF_UNIFORM_PARAMS
    rb = wf_uni(rb); zp = wf_uni(zp);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int BM = 256, BN = 256;
    const int Cin = p.Cin, Cout = p.Cout;
    unsigned *const in_cnt = p.sync + 64, *const mm_cnt = p.sync + 64 + p.NRB;
    (void)lane; (void)wave; (void)Cout; (void)in_cnt; (void)mm_cnt; (void)BM;
    // ---------------- input transform of tiles [t0, t1) of row block rb: one lane = one tile x 4 channels
    const int per = BN / p.PI;
    const long long t0 = (long long)rb * BN + (long long)zp * per;
    long long t1 = t0 + per;
    if (t1 > p.g.T) t1 = p.g.T;
    const int c4 = Cin >> 2;
    const int n = t1 > t0 ? (int)(t1 - t0) * c4 : 0;
    for (int u = tid; u < n; u += WF_THREADS) {
        const long long t = t0 + u / c4;
        const int c = (u % c4) << 2;
        int b, sy, sx, ty, tx;
        wino_tile(p.g, t, b, sy, sx, ty, tx);
        const long long plane = p.Tpad * Cin;                    // floats of one position of V
        const unsigned voff = (unsigned)((int)t * Cin + c) * 4u;
        float4 r[6][6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const int y = sy + (4 * ty - 1 + a) * p.g.d;
            float4 dv[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int x = sx + (4 * tx - 1 + j) * p.g.d;
                const bool ok = y >= 0 && y < p.g.H && x >= 0 && x < p.g.W;
                dv[j] = ok ? wf_ld(p.X, (unsigned)(((b * p.g.H + y) * p.g.W + x) * Cin + c) * 4u) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            wino4_bt(dv, r[a]);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float4 col[6] = {r[0][j], r[1][j], r[2][j], r[3][j], r[4][j], r[5][j]};
            float4 o[6];
            wino4_bt(col, o);
#pragma unroll
            for (int i = 0; i < 6; ++i) wf_store_wt(p.V + (long long)(i * 6) * plane, voff + (unsigned)j * (unsigned)(plane * 4), o[i]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(WF_G(unsigned, &in_cnt[rb]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __noinline__ void wf_item_mm(const WfParams &pr, int rb, int cb, int zp)
{
    WF_UNIFORM_PARAMS
    rb = wf_uni(rb); cb = wf_uni(cb); zp = wf_uni(zp);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int BM = 256, BN = 256;
    const int Cin = p.Cin, Cout = p.Cout;
    unsigned *const in_cnt = p.sync + 64, *const mm_cnt = p.sync + 64 + p.NRB;
    (void)lane; (void)wave; (void)Cout; (void)in_cnt; (void)mm_cnt; (void)BM;
    extern __shared__ __attribute__((aligned(1024))) char lds16[];   // [2] weight tiles | [2] row tiles, 128 bytes per row
    constexpr int WN = 4, MI = 8, NJ = 4, WROWS = MI * 16;
    const float sb = wino_pow2(14 - wf_uni(wino_amax_exp(wf_ld_u32(p.amax_in))));
    char *wbuf = lds16, *xbuf = lds16 + 2 * (BM * 128);
    const int sub = lane >> 3, cs8 = lane & 7;
    const int chunk_byte = (cs8 ^ sub) << 4;
    const int nk = Cin / 32;
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fk = lane >> 4;
    // ---------------- one GEMM tile: rows of rb x channels of cb at position z = zp
    if (wave == 0) {
        if (lane == 0) wf_wait(&in_cnt[rb], (unsigned)p.PI, p.status);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const int z = zp, r0 = rb * BN, n0 = cb * BM;
    const char *wbase = p.U2 + ((long long)z * Cout + n0) * Cin * 4;
    const char *xbase = (const char *)p.V + ((long long)z * p.Tpad + r0) * Cin * 4;
    float *ybase = p.M + (long long)z * p.Tpad * Cout;
    float sc;
    {
        const int zi = z / 6, zj = z - zi * 6;
        const int psum = ((0x433444 >> (4 * zi)) & 15) + ((0x433444 >> (4 * zj)) & 15);
        sc = sb * wino_pow2(-psum);
    }
    auto stage = [&](int t, int buf) {
        const char *wk = wbase + (long long)t * 128 + chunk_byte;
        char *dw = wbuf + buf * (BM * 128);
#pragma unroll
        for (int r = 0; r < BM / 64; ++r) {
            const int blk = r * 8 + wave;
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
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) stage(t + 1, cur ^ 1);
        const char *lw = wbuf + cur * (BM * 128), *lx = xbuf + cur * (BN * 128);
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
            const f32x4 a = *(const f32x4 *)(lx + row * 128 + (((2 * fk) ^ (row & 7)) << 4));
            const f32x4 b = *(const f32x4 *)(lx + row * 128 + (((2 * fk + 1) ^ (row & 7)) << 4));
            const f32x8 v = (f32x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]} * sc;
            ph[j] = __builtin_convertvector(v, f16x8);
            pl[j] = __builtin_convertvector(v - __builtin_convertvector(ph[j], f32x8), f16x8);
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int row = r0 + wn * (NJ * 16) + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int c = n0 + wm * WROWS + i * 16 + (lane >> 4) * 4;
            wf_store_wt(ybase, (unsigned)(row * Cout + c) * 4u, acc[i][j]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(WF_G(unsigned, &mm_cnt[rb * p.NCB + cb]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __noinline__ void wf_item_out(const WfParams &pr, int rb, int cb, int zp)
{
    WF_UNIFORM_PARAMS
    rb = wf_uni(rb); cb = wf_uni(cb); zp = wf_uni(zp);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int BM = 256, BN = 256;
    const int Cin = p.Cin, Cout = p.Cout;
    unsigned *const in_cnt = p.sync + 64, *const mm_cnt = p.sync + 64 + p.NRB;
    (void)lane; (void)wave; (void)Cout; (void)in_cnt; (void)mm_cnt; (void)BM;
    const float inv = wino_pow2(wf_uni(wino_amax_exp(wf_ld_u32(p.amax_in))) - 14);
    float csv[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) csv[i] = *(const __attribute__((address_space(4))) float *)(p.cs + i);
    // ---------------- output transform of tiles [t0, t1) of (rb, cb): one lane = one tile x 4 output channels
    if (wave == 0) {
        if (lane == 0) wf_wait(&mm_cnt[rb * p.NCB + cb], 36u, p.status);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const int per = BN / p.PO;
    const long long t0 = (long long)rb * BN + (long long)zp * per;
    long long t1 = t0 + per;
    if (t1 > p.g.T) t1 = p.g.T;
    constexpr int k4 = BM >> 2;
    const int n = t1 > t0 ? (int)(t1 - t0) * k4 : 0;
    unsigned mx = 0;
    const unsigned plane_b = (unsigned)wf_uni((int)(p.Tpad * Cout * 4));          // bytes of one position of M
    const float *mb[6];                                                          // row i of the 6 x 6 positions: scalar bases
#pragma unroll
    for (int i = 0; i < 6; ++i) mb[i] = wf_uni(p.M + (long long)(i * 6) * p.Tpad * Cout);
    for (int u = tid; u < n; u += WF_THREADS) {
        const long long t = t0 + u / k4;
        const int k = cb * BM + ((u % k4) << 2);
        int b, sy, sx, ty, tx;
        wino_tile(p.g, t, b, sy, sx, ty, tx);
        float4 s[4][6];
        const unsigned moff = (unsigned)((int)t * Cout + k) * 4u;
        // two halves of 18 loads: all 36 in flight next to s[][] would not fit the register file
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int j = 3 * half; j < 3 * half + 3; ++j) {
                float4 col[6];
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    col[i] = inv * (csv[i * 6 + j] * wf_ld_nt(mb[i], moff + (unsigned)j * plane_b));      // both powers of two: exact
                float4 o[4];
                wino4_at(col, o);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i][j] = o[i];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const float4 bv = wf_ld(p.bias + k);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = sy + (4 * ty + i) * p.g.d;
            float4 o[4];
            wino4_at(s[i], o);
            if (y >= p.g.H) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = sx + (4 * tx + j) * p.g.d;
                if (x >= p.g.W) continue;
                float4 v = o[j] + bv;
                const unsigned off = (unsigned)(((b * p.g.H + y) * p.g.W + x) * Cout + k) * 4u;
                if (p.R) v = v + wf_ld(p.R, off);
                if (p.relu) v = wino_relu(v);
                wf_st(p.Y, off, v);
                mx = max(max(mx, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu,
                         max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)));
            }
        }
    }
    if (p.amax_out) {
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
        if (lane == 0 && mx > wf_ld_u32(p.amax_out)) __hip_atomic_fetch_max(WF_G(unsigned, p.amax_out), mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(WF_THREADS) void k_wino4_fused(WfParams p)
{
    __shared__ unsigned s_item;
    const int tid = threadIdx.x;
    unsigned *const heads = p.sync;
    const unsigned *const lists = p.items + 16;
    int qsel = wf_xcc_id(), tried = 0;
    for (;;) {
        if (tid == 0) {
            unsigned it = WF_NONE;
            while (tried < 8) {
                const unsigned lo = p.items[qsel], hi = p.items[qsel + 1];
                const unsigned idx = __hip_atomic_fetch_add(&heads[qsel], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (idx < hi - lo) { it = lists[lo + idx]; break; }
                qsel = (qsel + 1) & 7;
                ++tried;
            }
            s_item = it;
        }
        __syncthreads();
        const unsigned it = s_item;
        __syncthreads();
        if (it == WF_NONE) break;
        const unsigned type = it & 3u;
        const int cb = (int)((it >> 2) & 3u), zp = (int)((it >> 4) & 63u), rb = (int)(it >> 10);
        if (type == WF_IN) wf_item_in(p, rb, zp);
        else if (type == WF_MM) wf_item_mm(p, rb, cb, zp);
        else wf_item_out(p, rb, cb, zp);
    }
}

// queue heads and counters to zero, the layer's 36 output scales into their slots, the tracked maximum to zero
__global__ void k_wf_reset(unsigned *sync, int n, unsigned *amax_out, WinoScale cs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sync[i] = (i >= 16 && i < 52) ? __float_as_uint(cs.c[i - 16]) : 0u;
    if (i == 0 && amax_out) *amax_out = 0u;
}

// ---------------------------------------------------------------------------------------------------------------
// item lists (host).  XCD x owns the contiguous row blocks [x * per, (x + 1) * per); its list walks them in windows:
//   window k: the MM tiles in (z, rb, cb) order (all workgroups of the XCD on one position: its weights' planes, 1 MB, stay
//   in L2), with the IN items of window k + 1 spread over its first 80 % and the OUT items of window k - 1 spread over it.
//   The last window runs pair-major ((rb, cb), z) with every pair's OUT items a few dozen items behind its last tile, so
//   that only the last pairs' output transforms are left without matrix work beside them.
// ---------------------------------------------------------------------------------------------------------------
static void wf_build_list(int rb0, int rb1, int NCB, int PI, int PO, const std::vector<int> &wins, std::vector<unsigned> &out)
{
    struct Key { double k; unsigned item; };
    const int nw = (int)wins.size();
    if (nw == 0) return;
    std::vector<int> start(nw + 1, rb0);
    for (int k = 0; k < nw; ++k) start[k + 1] = start[k] + wins[k];
    // prologue: the first window's input transform
    for (int rb = start[0]; rb < start[1]; ++rb)
        for (int q = 0; q < PI; ++q) out.push_back(WF_ITEM(WF_IN, 0, q, rb));
    for (int k = 0; k < nw; ++k) {
        std::vector<Key> keys;
        const int g = wins[k];
        const double N = (double)g * NCB * 36;
        const bool pair_major = (k == nw - 1);
        double pos = 0;
        if (!pair_major) {
            for (int z = 0; z < 36; ++z)
                for (int rb = start[k]; rb < start[k + 1]; ++rb)
                    for (int cb = 0; cb < NCB; ++cb) keys.push_back({pos++, WF_ITEM(WF_MM, cb, z, rb)});
        } else {
            for (int rb = start[k]; rb < start[k + 1]; ++rb)
                for (int cb = 0; cb < NCB; ++cb) {
                    for (int z = 0; z < 36; ++z) keys.push_back({pos++, WF_ITEM(WF_MM, cb, z, rb)});
                    for (int q = 0; q < PO; ++q) keys.push_back({pos + 40.0 + 0.01 * q, WF_ITEM(WF_OUT, cb, q, rb)});
                }
        }
        if (k + 1 < nw) {
            const int n_in = wins[k + 1] * PI;
            int i = 0;
            for (int rb = start[k + 1]; rb < start[k + 2]; ++rb)
                for (int q = 0; q < PI; ++q, ++i) keys.push_back({(i + 0.5) * 0.8 * N / n_in, WF_ITEM(WF_IN, 0, q, rb)});
        }
        if (k > 0) {
            const int n_out = wins[k - 1] * NCB * PO;
            int i = 0;
            for (int rb = start[k - 1]; rb < start[k]; ++rb)
                for (int cb = 0; cb < NCB; ++cb)
                    for (int q = 0; q < PO; ++q, ++i) keys.push_back({0.05 * N + (i + 0.5) * 0.9 * N / n_out, WF_ITEM(WF_OUT, cb, q, rb)});
        }
        std::stable_sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) { return a.k < b.k; });
        for (const Key &e : keys) out.push_back(e.item);
    }
}

static std::vector<int> wf_windows(int n, int NCB)
{
    // first and last window small (their transforms have no matrix work of their own XCD beside them), the middle ones
    // large enough that a position's tiles fill the XCD's 32 workgroups
    std::vector<int> w;
    if (n <= 0) return w;
    const int edge = getenv("SPA_WF_EDGE") ? atoi(getenv("SPA_WF_EDGE")) : 4;
    const int mid = getenv("SPA_WF_MID") ? atoi(getenv("SPA_WF_MID")) : (NCB == 1 ? 16 : 12);
    if (n <= 2 * edge) {
        w.push_back((n + 1) / 2);
        if (n / 2) w.push_back(n / 2);
        return w;
    }
    w.push_back(edge);
    int rest = n - 2 * edge;
    const int parts = (rest + mid - 1) / mid;
    for (int i = 0; i < parts; ++i) { const int s = rest / (parts - i); w.push_back(s); rest -= s; }
    w.push_back(edge);
    return w;
}

// the whole plan of a layer: 16 header words (offsets of the 8 lists, [8] = total) + the lists
static void wf_build_all(int NRB, int NCB, int PI, int PO, std::vector<unsigned> &words)
{
    words.assign(16, 0u);
    const int per = (NRB + 7) / 8;
    for (int xcd = 0; xcd < 8; ++xcd) {
        const int rb0 = std::min(NRB, xcd * per), rb1 = std::min(NRB, (xcd + 1) * per);
        words[xcd] = (unsigned)(words.size() - 16);
        std::vector<unsigned> l;
        wf_build_list(rb0, rb1, NCB, PI, PO, wf_windows(rb1 - rb0, NCB), l);
        words.insert(words.end(), l.begin(), l.end());
    }
    words[8] = (unsigned)(words.size() - 16);
}

// the work-item plan of a layer as the kernel reads it (host only: no GPU needed; the CPU suite checks that every item
// appears once and every producer precedes its consumers).  Returns the number of words, or SPA_ERR_CAPACITY.
// word = type (2 bits: 0 IN, 1 MM, 2 OUT) | channel block (2) | position z or slice (6) | row block (22)
extern "C" int64_t spa_wino4_fused_plan(int32_t n_row_blocks, int32_t n_channel_blocks, int32_t in_slices, int32_t out_slices,
                                        uint32_t *out, int64_t capacity)
{
    if (n_row_blocks < 0 || n_channel_blocks < 1 || n_channel_blocks > 4 || in_slices < 1 || in_slices > 8 || out_slices < 1 || out_slices > 8) {
        spa_set_error("spa_wino4_fused_plan: invalid argument");
        return SPA_ERR_ARG;
    }
    std::vector<unsigned> words;
    wf_build_all(n_row_blocks, n_channel_blocks, in_slices, out_slices, words);
    if ((int64_t)words.size() > capacity || !out) return SPA_ERR_CAPACITY;
    memcpy(out, words.data(), words.size() * 4);
    return (int64_t)words.size();
}

// 32-bit words of counter scratch the caller provides for a layer of tiles_padded = spa_wino4_tiles(...) rows
extern "C" int64_t spa_wino4_fused_scratch_words(int64_t tiles_padded, int32_t Cout)
{
    const long long NRB = tiles_padded / 256, NCB = Cout / 256;
    return 64 + NRB * (1 + NCB);          // queue heads, scale constants, in_cnt[NRB], mm_cnt[NRB * NCB]
}

// spa_conv3x3_wino4_f16s as one persistent launch (header).  scratch: spa_wino4_fused_scratch_words(...) 32-bit words of
// device memory owned by the caller for the duration of the call (queue heads and dependency counters; calls on different
// streams do not share it).  Cout % 256 == 0, Cin % 32 == 0; otherwise as spa_conv3x3_wino4_f16s.
extern "C" int spa_conv3x3_wino4_fused(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                       const void *u2, const float *cs, int32_t Cout, const float *bias,
                                       const float *residual, int32_t relu, int32_t dilation, const void *amax_in,
                                       void *amax_out, void *v_scratch, float *m_scratch, void *scratch, float *y, void *stream)
{
    SPA_ARG(ctx && x && u2 && cs && bias && y && v_scratch && m_scratch && scratch && amax_in && B > 0 && H > 0 && W > 0 && dilation >= 1);
    SPA_ARG(Cin % 32 == 0 && Cout % 256 == 0 && Cout / 256 <= 4);
    SPA_ARG((((uintptr_t)x | (uintptr_t)u2 | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)v_scratch |
              (uintptr_t)m_scratch | (uintptr_t)scratch) % 16) == 0);
    hipStream_t s = spa_stream(stream);
    WfParams p;
    wino4_geom(B, H, W, dilation, &p.g);
    p.Tpad = (p.g.T + 255) / 256 * 256;
    const int NRB = (int)(p.Tpad / 256), NCB = Cout / 256;
    SPA_ARG(NRB < (1 << 22) && p.Tpad < (1ll << 31));
    // 32-bit byte offsets inside x / y and inside six positions' planes of V / M
    SPA_ARG((long long)B * H * W * (Cin > Cout ? Cin : Cout) * 4 < (1ll << 32) && 6 * p.Tpad * (Cin > Cout ? Cin : Cout) * 4 < (1ll << 31));
    p.X = x; p.V = (float *)v_scratch; p.M = m_scratch; p.Y = y; p.U2 = (const char *)u2; p.bias = bias; p.R = residual;
    p.Cin = Cin; p.Cout = Cout; p.relu = relu; p.NCB = NCB; p.NRB = NRB;
    // slices of ~32 (input) / ~64 (output) tiles: an item streams 0.6-1.2 MB, comparable to a GEMM tile's duration
    p.PI = getenv("SPA_WF_PI") ? atoi(getenv("SPA_WF_PI")) : 8;
    p.PO = getenv("SPA_WF_PO") ? atoi(getenv("SPA_WF_PO")) : 4;
    SPA_ARG(p.PI >= 1 && p.PI <= 8 && 256 % p.PI == 0 && p.PO >= 1 && p.PO <= 8 && 256 % p.PO == 0);
    WinoScale sc;
    for (int i = 0; i < 36; ++i) sc.c[i] = cs[i];
    p.amax_in = (const unsigned *)amax_in; p.amax_out = (unsigned *)amax_out;
    p.status = ctx->d_status;

    // the lists of this (NRB, NCB): built once per context, kept in device memory (read-only: shared by calls on any stream)
    int hit = -1;
    for (int i = 0; i < ctx->wf_n; ++i)
        if (ctx->wf_lists[i].key[0] == NRB && ctx->wf_lists[i].key[1] == NCB && ctx->wf_lists[i].key[2] == p.PI && ctx->wf_lists[i].key[3] == p.PO) hit = i;
    if (hit < 0) {
        std::vector<unsigned> words;
        wf_build_all(NRB, NCB, p.PI, p.PO, words);
        SPA_ARG((long long)words.size() == 16 + (long long)NRB * (p.PI + 36 * NCB + p.PO * NCB));
        hit = ctx->wf_n < SPA_WF_LISTS ? ctx->wf_n++ : 0;                 // (a full cache recycles slot 0)
        if (ctx->wf_lists[hit].d) { SPA_HIP(hipDeviceSynchronize()); SPA_HIP(hipFree(ctx->wf_lists[hit].d)); ctx->wf_lists[hit].d = nullptr; }
        SPA_HIP(hipMalloc((void **)&ctx->wf_lists[hit].d, words.size() * 4));
        SPA_HIP(hipMemcpy(ctx->wf_lists[hit].d, words.data(), words.size() * 4, hipMemcpyHostToDevice));
        ctx->wf_lists[hit].key[0] = NRB; ctx->wf_lists[hit].key[1] = NCB; ctx->wf_lists[hit].key[2] = p.PI; ctx->wf_lists[hit].key[3] = p.PO;
    }
    const unsigned *d_items = ctx->wf_lists[hit].d;
    unsigned *d_sync = (unsigned *)scratch;
    p.items = d_items; p.sync = d_sync;
    const int n_sync = 64 + NRB * (1 + NCB);
    if (!ctx->winof_attr_done) {
        SPA_HIP(hipFuncSetAttribute((const void *)k_wino4_fused, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128));
        ctx->winof_attr_done = 1;
    }
    SpaProfScope prof_(ctx, PROF_WINO_FUSED, s);
    hipLaunchKernelGGL(k_wf_reset, dim3((unsigned)((n_sync + 255) / 256)), dim3(256), 0, s, d_sync, n_sync, (unsigned *)amax_out, sc);
    hipLaunchKernelGGL(k_wino4_fused, dim3((unsigned)ctx->n_cu), dim3(WF_THREADS), 2 * 512 * 128, s, p);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
