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
    unsigned *sync;               // [0..7] heads, [8] abort word, [16..51] the 36 scale constants (float), [64 ..] in_cnt[NRB], then mm_cnt[NRB * NCB]
    int NRB;
    uint32_t *status;
    int vec2;                     // bit 0: input transform on 2 channels per lane, bit 1: output transform (else 4)
    unsigned long long *dbg;      // development aid (SPA_WF_TIMING=1): per item kind {ticks of the 100 MHz clock, count}, [6] pop ticks
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
// (plain stores instead — aux 0 — were measured: the 512 -> 512 layer then differs from the three-launch result, i.e. a
// consumer did read stale bytes; the write-through form costs nothing measurable)
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

// lane 0 of the workgroup: wait until *cnt >= want (sc1 poll).  Bounded: ~0.3 s without progress latches SPA_ST_WINO_SYNC and
// raises the launch's abort word (sync[8]), which ends every other wait at once — a broken launch returns in well under a
// second with the status bit set instead of timing its waits out one after the other
__device__ __forceinline__ void wf_wait(const unsigned *cnt, unsigned want, uint32_t *status, unsigned long long *dbg, int slot,
                                        unsigned *abort_word)
{
    unsigned spins = 0;
    const unsigned long long c0 = dbg ? wall_clock64() : 0ull;
    while (__hip_atomic_load(WF_G(const unsigned, cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(8);
        if ((++spins & 63u) == 0u && __hip_atomic_load(WF_G(const unsigned, abort_word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        if (spins > (1u << 18)) {
            __hip_atomic_fetch_or(WF_G(uint32_t, status), SPA_ST_WINO_SYNC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(WF_G(unsigned, abort_word), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
    if (dbg) atomicAdd(&dbg[slot], wall_clock64() - c0);
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
// vector-typed global accesses of 2 or 4 floats at (uniform base, 32-bit byte offset)
template <typename V> struct WfVec;
template <> struct WfVec<float2> {
    typedef float raw __attribute__((ext_vector_type(2)));
    typedef unsigned uraw __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ float2 make(raw v) { return make_float2(v[0], v[1]); }
    static __device__ __forceinline__ raw unmake(float2 v) { return (raw){v.x, v.y}; }
    static __device__ __forceinline__ void store_wt(float *base, unsigned off, float2 v)
    {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uraw, unmake(v)), r, (int)off, 0, 17);
    }
    static __device__ __forceinline__ unsigned amax(float2 v) { return max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu); }
};
template <> struct WfVec<float4> {
    typedef float raw __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ float4 make(raw v) { return make_float4(v[0], v[1], v[2], v[3]); }
    static __device__ __forceinline__ raw unmake(float4 v) { return (raw){v.x, v.y, v.z, v.w}; }
    static __device__ __forceinline__ void store_wt(float *base, unsigned off, float4 v) { wf_store_wt(base, off, v); }
    static __device__ __forceinline__ unsigned amax(float4 v)
    {
        return max(max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu),
                   max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu));
    }
};
template <typename V> __device__ __forceinline__ V wf_ldv(const float *base, unsigned off)
{
    return WfVec<V>::make(*WF_G(const typename WfVec<V>::raw, (const char *)base + off));
}
template <typename V> __device__ __forceinline__ V wf_ldv_nt(const float *base, unsigned off)
{
    return WfVec<V>::make(__builtin_nontemporal_load(WF_G(const typename WfVec<V>::raw, (const char *)base + off)));
}
template <typename V> __device__ __forceinline__ void wf_stv(float *base, unsigned off, V v)
{
    *WF_G(typename WfVec<V>::raw, (char *)base + off) = WfVec<V>::unmake(v);
}

// Both transform bodies are written for EIGHT waves per compute unit (the GEMM tiles need the whole register file, so nothing
// else is resident): a lane issues all the loads of its unit before it touches any of them — straight-line code, clamped
// addresses and selects instead of branches around loads (a branch makes the compiler drain the loads in flight) — because at
// this occupancy the bytes a lane keeps in flight are the streaming rate.  V = float2 or float4 channels per lane.
template <typename V>
__device__ __noinline__ void wf_item_in(const WfParams &pr, int rb, int zp)
{
    WF_UNIFORM_PARAMS
    rb = wf_uni(rb); zp = wf_uni(zp);
    constexpr int VN = sizeof(V) / 4;
    const int tid = threadIdx.x;
    constexpr int BN = 256;
    const int Cin = p.Cin;
    unsigned *const in_cnt = p.sync + 64;
    // ---------------- input transform of tiles [t0, t1) of row block rb: one lane = one tile x VN channels
    const int per = BN / p.PI;
    const int t0 = rb * BN + zp * per;
    int t1 = t0 + per;
    if (t1 > (int)p.g.T) t1 = (int)p.g.T;
    const int cv = Cin / VN;
    const int n = t1 > t0 ? (t1 - t0) * cv : 0;
    const unsigned plane_b = (unsigned)wf_uni((int)(p.Tpad * Cin * 4));          // bytes of one position of V
    float *vb[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) vb[i] = wf_uni(p.V + (long long)(i * 6) * p.Tpad * Cin);
    const int H = p.g.H, W = p.g.W, d = p.g.d;
    for (int u = tid; u < n; u += WF_THREADS) {
        const int t = t0 + u / cv;
        const int c = (u % cv) * VN;
        int b, sy, sx, ty, tx;
        wino_tile(p.g, (long long)t, b, sy, sx, ty, tx);
        const unsigned voff = (unsigned)(t * Cin + c) * 4u;
        // all 36 loads first (rows / columns outside the image: a clamped address, the value replaced by zero)
        V dv[6][6];
        unsigned okx = 0, oky = 0;
        int xo[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int x = sx + (4 * tx - 1 + j) * d;
            okx |= (x >= 0 && x < W) ? (1u << j) : 0u;
            xo[j] = min(max(x, 0), W - 1);
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const int y = sy + (4 * ty - 1 + a) * d;
            oky |= (y >= 0 && y < H) ? (1u << a) : 0u;
            const int row = (b * H + min(max(y, 0), H - 1)) * W;
#pragma unroll
            for (int j = 0; j < 6; ++j) dv[a][j] = wf_ldv<V>(p.X, (unsigned)((row + xo[j]) * Cin + c) * 4u);
        }
        if (VN == 2) __builtin_amdgcn_sched_barrier(0);          // (two channels per lane: all 36 loads fit in flight)
        V r[6][6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (!((oky >> a) & (okx >> j) & 1u)) dv[a][j] = wino_zero<V>();
            wino4_bt(dv[a], r[a]);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const V col[6] = {r[0][j], r[1][j], r[2][j], r[3][j], r[4][j], r[5][j]};
            V o[6];
            wino4_bt(col, o);
#pragma unroll
            for (int i = 0; i < 6; ++i) WfVec<V>::store_wt(vb[i], voff + (unsigned)j * plane_b, o[i]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(WF_G(unsigned, &in_cnt[rb]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// workgroup-shared scratch words of the dispatcher: [0] broadcast item, [1] queue of lane 0's pops, [2] queues tried, [3] flag
__shared__ unsigned wf_s[8];

// lane 0 of the workgroup: pop the next item of this workgroup's list (an empty list: steal from the next XCD's)
__device__ __forceinline__ unsigned wf_pop(const unsigned *items, unsigned *heads)
{
    int qsel = (int)wf_s[1], tried = (int)wf_s[2];
    unsigned it = WF_NONE;
    while (tried < 8) {
        const unsigned lo = WF_G(const unsigned, items)[qsel], hi = WF_G(const unsigned, items)[qsel + 1];
        const unsigned idx = __hip_atomic_fetch_add(WF_G(unsigned, &heads[qsel]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (idx < hi - lo) { it = WF_G(const unsigned, items)[16 + lo + idx]; break; }
        qsel = (qsel + 1) & 7;
        ++tried;
    }
    wf_s[1] = (unsigned)qsel; wf_s[2] = (unsigned)tried;
    return it;
}

// A RUN of GEMM tiles: item `it` and every GEMM item this workgroup pops right behind it whose input is ready, as one
// software-pipelined loop (the structure of k_gemm_f16x3): the next tile's first K step is staged during this tile's last K
// step, the epilogue's write-through stores drain under the next tile's first K step (its mm_cnt is signalled after that
// step's drain), and the pop of the next item is spread over four earlier K steps — step nk-5: the queue head's atomic is
// issued (its latency sits under that step's matrix work, the end-of-step vmcnt(0) collects it), nk-4: the item word is
// loaded, nk-3: its row block's in_cnt, nk-2: verdict to LDS + agent acquire, nk-1: staging.  Returns the first item that is
// not part of the run (popped, not started), or WF_NONE.
__device__ __noinline__ unsigned wf_run_mm(const WfParams &pr, unsigned it_arg)
{
    WF_UNIFORM_PARAMS
    unsigned it = (unsigned)wf_uni((int)it_arg);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int BM = 256, BN = 256;
    const int Cin = p.Cin, Cout = p.Cout;
    unsigned *const heads = p.sync, *const in_cnt = p.sync + 64, *const mm_cnt = p.sync + 64 + p.NRB;
    const unsigned *const items = wf_uni(pr.items);
    extern __shared__ __attribute__((aligned(1024))) char lds16[];   // [2] weight tiles | [2] row tiles, 128 bytes per row
    constexpr int WN = 4, MI = 8, NJ = 4, WROWS = MI * 16;
    const float sb = wino_pow2(14 - wf_uni(wino_amax_exp(wf_ld_u32(p.amax_in))));
    char *wbuf = lds16, *xbuf = lds16 + 2 * (BM * 128);
    const int sub = lane >> 3, cs8 = lane & 7;
    const int chunk_byte = (cs8 ^ sub) << 4;
    const int nk = Cin / 32;
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fk = lane >> 4;

    int r0, n0, pair;
    const char *wbase, *xbase;
    float *ybase;
    float sc_next;
    auto locate = [&](unsigned item) {
        const int cb = (int)((item >> 2) & 3u), z = (int)((item >> 4) & 63u), rb = (int)(item >> 10);
        r0 = rb * BN; n0 = cb * BM; pair = rb * p.NCB + cb;
        wbase = p.U2 + ((long long)z * Cout + n0) * Cin * 4;
        xbase = (const char *)p.V + ((long long)z * p.Tpad + r0) * Cin * 4;
        ybase = p.M + (long long)z * p.Tpad * Cout;
        const int zi = z / 6, zj = z - zi * 6;
        const int psum = ((0x433444 >> (4 * zi)) & 15) + ((0x433444 >> (4 * zj)) & 15);
        sc_next = sb * wino_pow2(-psum);
    };
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

    // the first tile of the run: wait for its input (blocking), acquire, stage
    if (wave == 0) {
        if (lane == 0) wf_wait(&in_cnt[it >> 10], (unsigned)p.PI, p.status, pr.dbg, 8, p.sync + 8);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    locate(it);
    stage(0, 0);
    float sc = sc_next;
    bool first = true;
    int par = 0, pending = -1;                 // pending: pair index of the tile whose stores are still draining
    unsigned nxt = WF_NONE;                    // what the run hands back
    // lane 0's look-ahead state
    unsigned pf_idx = 0, pf_item = WF_NONE, pf_cnt = 0, pf_lo = 0, pf_n = 0;
    for (;;) {
        f32x4 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // this tile's first K step was staged BEFORE the previous tile's MI * NJ epilogue stores per wave (vmcnt counts both,
        // in order): waiting until that many operations remain lets the stores drain under this tile's matrix work
        if (!first) {
            static_assert(MI * NJ <= 63, "vmcnt range");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MI * NJ) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        first = false;
        __syncthreads();
        int e_r0 = 0, e_n0 = 0, e_pair = 0;
        float *e_y = nullptr;
        bool more = false;
        for (int t = 0; t < nk; ++t) {
            const int cur = (t + par) & 1;
            // ---- the look-ahead of lane 0 (wave 0), one dependent memory operation per K step
            if (wave == 0) {
                if (t == nk - 5) {
                    if (lane == 0) {
                        const int qsel = (int)wf_s[1];
                        pf_lo = WF_G(const unsigned, items)[qsel];
                        pf_n = WF_G(const unsigned, items)[qsel + 1] - pf_lo;
                        pf_idx = wf_s[2] < 8u ? __hip_atomic_fetch_add(WF_G(unsigned, &heads[qsel]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
                    }
                } else if (t == nk - 4) {
                    if (lane == 0) pf_item = pf_idx < pf_n ? WF_G(const unsigned, items)[16 + pf_lo + pf_idx] : WF_NONE;
                } else if (t == nk - 3) {
                    if (lane == 0 && pf_item != WF_NONE && (pf_item & 3u) == WF_MM)
                        pf_cnt = __hip_atomic_load(WF_G(const unsigned, &in_cnt[pf_item >> 10]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (t == nk - 2) {
                    if (lane == 0) {
                        // (a list that ran dry: the steal is left to the blocking pop after the run)
                        wf_s[0] = pf_item;
                        wf_s[3] = (pf_item != WF_NONE && (pf_item & 3u) == WF_MM && pf_cnt >= (unsigned)p.PI) ? 1u : 0u;
                        wf_s[4] = pf_idx < pf_n ? 1u : 0u;                       // 0: the list was empty, nothing was popped
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // before the next tile's operands are touched
                }
            }
            if (t + 1 < nk) stage(t + 1, cur ^ 1);
            else {
                // last K step: the other buffers are free — stage the next tile's first K step under this step's matrix work
                e_r0 = r0; e_n0 = n0; e_y = ybase; e_pair = pair;
                more = wf_s[3] != 0u;
                if (more) { locate(wf_s[0]); stage(0, cur ^ 1); }
            }
            const char *lw = wbuf + cur * (BM * 128), *lx = xbuf + cur * (BN * 128);
            const float scl = sc;
            if (t + 1 == nk) sc = sc_next;          // (locate() above has moved on to the next tile)
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
                const f32x8 v = (f32x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]} * scl;
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
            // the previous tile's stores were drained by every wave's wait above: its 36-counter may move
            if (t == 0 && pending >= 0) {
                if (tid == 0) __hip_atomic_fetch_add(WF_G(unsigned, &mm_cnt[pending]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pending = -1;
            }
        }
        par = (par + nk) & 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = e_r0 + wn * (NJ * 16) + j * 16 + (lane & 15);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int c = e_n0 + wm * WROWS + i * 16 + (lane >> 4) * 4;
                wf_store_wt(e_y, (unsigned)(row * Cout + c) * 4u, acc[i][j]);
            }
        }
        pending = e_pair;
        if (!more) break;
    }
    // the run ends: drain the last tile's stores, signal, and hand back what the look-ahead popped (or pop, blocking)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(WF_G(unsigned, &mm_cnt[pending]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned h = wf_s[0];
        if (wf_s[4] == 0u) {                   // the look-ahead found its list empty: move on to the next list
            wf_s[1] = (wf_s[1] + 1u) & 7u; wf_s[2] = wf_s[2] + 1u;
            h = wf_pop(items, heads);
        }
        wf_s[0] = h;
    }
    __syncthreads();
    nxt = wf_s[0];
    __syncthreads();
    return nxt;
}

template <typename V>
__device__ __noinline__ void wf_item_out(const WfParams &pr, int rb, int cb, int zp)
{
    WF_UNIFORM_PARAMS
    rb = wf_uni(rb); cb = wf_uni(cb); zp = wf_uni(zp);
    constexpr int VN = sizeof(V) / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int BM = 256, BN = 256;
    const int Cout = p.Cout;
    unsigned *const mm_cnt = p.sync + 64 + p.NRB;
    const float inv = wino_pow2(wf_uni(wino_amax_exp(wf_ld_u32(p.amax_in))) - 14);
    float csv[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) csv[i] = *(const __attribute__((address_space(4))) float *)(p.cs + i);
    // ---------------- output transform of tiles [t0, t1) of (rb, cb): one lane = one tile x VN output channels
    if (wave == 0) {
        if (lane == 0) wf_wait(&mm_cnt[rb * p.NCB + cb], 36u, p.status, pr.dbg, 9, p.sync + 8);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const int per = BN / p.PO;
    const int t0 = rb * BN + zp * per;
    int t1 = t0 + per;
    if (t1 > (int)p.g.T) t1 = (int)p.g.T;
    constexpr int kv = BM / VN;
    const int n = t1 > t0 ? (t1 - t0) * kv : 0;
    unsigned mx = 0;
    const unsigned plane_b = (unsigned)wf_uni((int)(p.Tpad * Cout * 4));          // bytes of one position of M
    const float *mb[6];                                                          // row i of the 6 x 6 positions: scalar bases
#pragma unroll
    for (int i = 0; i < 6; ++i) mb[i] = wf_uni(p.M + (long long)(i * 6) * p.Tpad * Cout);
    const int H = p.g.H, W = p.g.W, d = p.g.d;
    const bool has_res = p.R != nullptr;
    for (int u = tid; u < n; u += WF_THREADS) {
        const int t = t0 + u / kv;
        const int k = cb * BM + (u % kv) * VN;
        int b, sy, sx, ty, tx;
        wino_tile(p.g, (long long)t, b, sy, sx, ty, tx);
        const unsigned moff = (unsigned)(t * Cout + k) * 4u;
        // all 36 loads of M first (M is read exactly once: non-temporal), then the residual's 16 (clamped addresses)
        V m[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i][j] = wf_ldv_nt<V>(mb[i], moff + (unsigned)j * plane_b);
        unsigned yoff[4], okm = 0;
        int xo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = sx + (4 * tx + j) * d;
            okm |= (x < W) ? (1u << j) : 0u;
            xo[j] = min(x, W - 1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = sy + (4 * ty + i) * d;
            okm |= (y < H) ? (16u << i) : 0u;
            yoff[i] = (unsigned)((b * H + min(y, H - 1)) * W);
        }
        V res[4][4];
        if (VN == 2 && has_res) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) res[i][j] = wf_ldv<V>(p.R, ((yoff[i] + (unsigned)xo[j]) * (unsigned)Cout + (unsigned)k) * 4u);
        }
        V s[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            V col[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) col[i] = inv * (csv[i * 6 + j] * m[i][j]);      // both powers of two: exact
            V o[4];
            wino4_at(col, o);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i][j] = o[i];
        }
        if (VN == 4 && has_res) {
            // (four channels per lane: the residual's 64 registers only fit once M's are free)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) res[i][j] = wf_ldv<V>(p.R, ((yoff[i] + (unsigned)xo[j]) * (unsigned)Cout + (unsigned)k) * 4u);
        }
        const V bv = wf_ldv<V>(p.bias, (unsigned)k * 4u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            V o[4];
            wino4_at(s[i], o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                V v = o[j] + bv;
                if (has_res) v = v + res[i][j];
                if (p.relu) v = wino_relu(v);
                if ((okm >> j) & (okm >> (4 + i)) & 1u) {
                    wf_stv<V>(p.Y, ((yoff[i] + (unsigned)xo[j]) * (unsigned)Cout + (unsigned)k) * 4u, v);
                    mx = max(mx, WfVec<V>::amax(v));
                }
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
    const int tid = threadIdx.x;
    const unsigned long long cstart = p.dbg ? wall_clock64() : 0ull;
    if (tid == 0) {
        wf_s[1] = (unsigned)wf_xcc_id(); wf_s[2] = 0u; wf_s[3] = 0u; wf_s[4] = 1u;
        wf_s[0] = wf_pop(p.items, p.sync);
    }
    __syncthreads();
    unsigned it = wf_s[0];
    __syncthreads();
    while (it != WF_NONE) {
        const unsigned type = it & 3u;
        const int cb = (int)((it >> 2) & 3u), zp = (int)((it >> 4) & 63u), rb = (int)(it >> 10);
        const unsigned long long c0 = p.dbg ? wall_clock64() : 0ull;
        if (type == WF_MM) {
            it = wf_run_mm(p, it);             // a run of GEMM tiles; returns the item behind it
        } else {
            // the next item's pop runs on lane 0 while the other waves are already at work
            if (tid == 0) wf_s[0] = wf_pop(p.items, p.sync);
            if (type == WF_IN) {
                if (p.vec2 & 1) wf_item_in<float2>(p, rb, zp); else wf_item_in<float4>(p, rb, zp);
            } else {
                if (p.vec2 & 2) wf_item_out<float2>(p, rb, cb, zp); else wf_item_out<float4>(p, rb, cb, zp);
            }
            __syncthreads();
            it = wf_s[0];
            __syncthreads();
        }
        if (p.dbg && tid == 0) {
            atomicAdd(&p.dbg[2 * type], wall_clock64() - c0);
            atomicAdd(&p.dbg[2 * type + 1], 1ull);
        }
    }
    if (p.dbg && tid == 0) atomicAdd(&p.dbg[7], wall_clock64() - cstart);
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
    SPA_ARG(Cin % 32 == 0 && Cin >= 160 && Cout % 256 == 0 && Cout / 256 <= 4);      // (>= 5 K steps: the look-ahead pipeline)
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
    p.dbg = nullptr;
    if (getenv("SPA_WF_TIMING")) {
        if (!ctx->wf_dbg) SPA_HIP(hipMalloc((void **)&ctx->wf_dbg, 16 * 8));
        SPA_HIP(hipMemsetAsync(ctx->wf_dbg, 0, 16 * 8, s));
        p.dbg = ctx->wf_dbg;
    }

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
    if (p.dbg) {
        unsigned long long h[16];
        SPA_HIP(hipMemcpy(h, p.dbg, sizeof h, hipMemcpyDeviceToHost));
        const double us = 0.01;       // 100 MHz
        fprintf(stderr, "wf timing: IN %llu items %.1f us each | MM %llu items %.1f us each | OUT %llu items %.1f us each | pop %.1f us per item | "
                "waits IN->MM %.0f us OUT %.0f us total | workgroup lifetime %.0f us avg\n",
                h[1], h[1] ? h[0] * us / h[1] : 0., h[3], h[3] ? h[2] * us / h[3] : 0., h[5], h[5] ? h[4] * us / h[5] : 0.,
                (h[1] + h[3] + h[5]) ? h[6] * us / (h[1] + h[3] + h[5]) : 0., h[8] * us, h[9] * us, h[7] * us / ctx->n_cu);
    }
    return SPA_OK;
}
