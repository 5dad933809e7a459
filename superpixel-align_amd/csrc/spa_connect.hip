// Connectivity enforcement of SLIC labels on gfx950, bit exact with skimage's
// _enforce_label_connectivity_cython (a sequential scan-order breadth-first relabelling).
//
// The sequential algorithm, restated as order-free facts (DESIGN.md "Connectivity"):
//   * without a max_size cut, the components it discovers are exactly the 4-connected
//     components of equal input label, visited in order of their first pixel in raster
//     order (the "seed" = minimum raster index of the component);
//   * a component with >= min_size pixels is KEPT and receives the next free label, so its
//     label is the number of kept components with a smaller seed;
//   * a component C with < min_size pixels takes the label of `adjacent`: the LAST pixel, in
//     the BFS visiting order of C (neighbour order +x,-x,+y,-y), that lies outside C in a
//     component D with seed(D) < seed(C) — whatever label D ended up with — or 0 when there
//     is none, or when C precedes the first kept component (everything before it is 0).
//   So: union-find CCL for the components, a prefix sum over raster order for the kept
//   labels, one wavefront per small component replaying its BFS in the exact queue order
//   (64 queue entries per step; discoveries inside a step are ordered by an atomicMin on
//   (queue index, direction) keys), then pointer chasing D -> label.
//   A component that reaches max_size is cut by the sequential algorithm in BFS order;
//   that case is detected and reported (SPA_ST_CONN_OVERSIZE).
#include "spa_common.h"

#define INF_KEY 0xFFFFFFFFu

struct ConnMisc {
    int n_small;       // number of small component roots
    int first_kept;    // smallest kept root (npix if none)
    int qalloc;        // BFS queue allocation cursor
    int n_kept;        // number of kept components
};

__device__ __forceinline__ int ld_i32(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_i32(int *p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int uf_find(const int *parent, int i)
{
    int p;
    while ((p = ld_i32(parent + i)) != i) i = p;
    return i;
}

__device__ __forceinline__ void uf_merge(int *parent, int a, int b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a > b) { int t = a; a = b; b = t; }
        int old = atomicMin(parent + b, a);
        if (old == b) return;
        b = old;
    }
}

// parent[p] = first pixel of p's horizontal run inside its 64-pixel chunk
__global__ __launch_bounds__(256) void k_ccl_init(const int32_t *__restrict__ lab,
                                                  int *__restrict__ parent, int W, int npix)
{
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int32_t *L = lab + (long long)b * npix;
    bool in = p < npix;
    int x = in ? p % W : 0;
    int l = in ? L[p] : -2;
    int lprev = (in && x > 0 && lane > 0) ? L[p - 1] : -3;
    bool start = !in || lane == 0 || x == 0 || lprev != l;
    unsigned long long m = __ballot(start);
    unsigned long long below = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    int s = 63 - __clzll((long long)(m & below));
    if (in) parent[(long long)b * npix + p] = p - lane + s;
}

__global__ __launch_bounds__(256) void k_ccl_merge(const int32_t *__restrict__ lab,
                                                   int *__restrict__ parent, int W, int npix)
{
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const int lane = threadIdx.x & 63;
    const int32_t *L = lab + (long long)b * npix;
    int *P = parent + (long long)b * npix;
    const int x = p % W;
    const int l = L[p];
    const bool left_same = (x > 0) && (L[p - 1] == l);
    if (left_same && lane == 0) uf_merge(P, p, p - 1);          // run continues across a chunk
    if (p >= W && L[p - W] == l) {
        // the pair to the left already links the two runs when it is vertically connected too
        bool implied = left_same && (L[p - W - 1] == l);
        if (!implied) uf_merge(P, p, p - W);
    }
}

__global__ __launch_bounds__(256) void k_ccl_flatten(int *__restrict__ parent,
                                                     int *__restrict__ size, int npix)
{
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    int *P = parent + (long long)b * npix;
    int *S = size + (long long)b * npix;
    int r = -1;
    if (p < npix) {
        r = uf_find(P, p);
        st_i32(P + p, r);
    }
    // wave-aggregated histogram: one atomic per distinct root per wave
    unsigned long long todo = __ballot(r >= 0);
    while (todo) {
        int leader = __ffsll((long long)todo) - 1;
        int rr = __shfl(r, leader);
        unsigned long long same = __ballot(r == rr);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(S + rr, __popcll(same));
        todo &= ~same;
    }
}

// pass 1 of the raster-order prefix sum over kept roots (+ small-root list, status bits)
#define SCAN_PX 1024
__global__ __launch_bounds__(256) void k_conn_count(const int *__restrict__ parent,
                                                    const int *__restrict__ size, int npix,
                                                    int min_size, int max_size,
                                                    int *__restrict__ blk, int nblk,
                                                    int *__restrict__ small_list,
                                                    ConnMisc *__restrict__ misc,
                                                    uint32_t *__restrict__ status)
{
    __shared__ int wsum[4];
    const int b = blockIdx.y;
    const int *P = parent + (long long)b * npix;
    const int *S = size + (long long)b * npix;
    const int base = blockIdx.x * SCAN_PX + threadIdx.x * 4;
    int c = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int p = base + i;
        if (p < npix && P[p] == p) {
            int sz = S[p];
            if (sz >= min_size) {
                ++c;
                atomicMin(&misc[b].first_kept, p);
                if (sz >= max_size) atomicOr(status, SPA_ST_CONN_OVERSIZE);
            } else {
                int slot = atomicAdd(&misc[b].n_small, 1);
                small_list[(long long)b * npix + slot] = p;
            }
        }
    }
    // block sum
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk[(long long)b * nblk + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// pass 2: exclusive scan of the block counts of one image (single workgroup)
__global__ __launch_bounds__(256) void k_conn_scan(int *__restrict__ blk, int nblk,
                                                   ConnMisc *__restrict__ misc,
                                                   int32_t *__restrict__ n_labels)
{
    __shared__ int part[256];
    const int b = blockIdx.x;
    int *B_ = blk + (long long)b * nblk;
    const int per = (nblk + 255) / 256;
    const int lo = threadIdx.x * per, hi = min(nblk, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += B_[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) { int t = part[i]; part[i] = run; run += t; }
        misc[b].n_kept = run;
        n_labels[b] = run > 0 ? run : 1;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { int t = B_[i]; B_[i] = run; run += t; }
}

// pass 3: label of every kept root = number of kept roots before it in raster order
__global__ __launch_bounds__(256) void k_conn_number(const int *__restrict__ parent,
                                                     const int *__restrict__ size, int npix,
                                                     int min_size, const int *__restrict__ blk,
                                                     int nblk, int *__restrict__ final_)
{
    __shared__ int wsum[4];
    const int b = blockIdx.y;
    const int *P = parent + (long long)b * npix;
    const int *S = size + (long long)b * npix;
    int *F = final_ + (long long)b * npix;
    const int base = blockIdx.x * SCAN_PX + threadIdx.x * 4;
    bool k[4];
    int c = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int p = base + i;
        k[i] = (p < npix) && (P[p] == p) && (S[p] >= min_size);
        c += k[i] ? 1 : 0;
    }
    // exclusive prefix of c over the workgroup (thread order == raster order)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = c;
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int off = blk[(long long)b * nblk + blockIdx.x];
    for (int i = 0; i < wv; ++i) off += wsum[i];
    int rank = off + inc - c;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (k[i]) F[base + i] = rank++;
}

// one wavefront per small component: replay the BFS in queue order, find `adjacent`
__global__ __launch_bounds__(64) void k_conn_bfs(const int *__restrict__ parent,
                                                 const int *__restrict__ size,
                                                 const int *__restrict__ small_list,
                                                 ConnMisc *__restrict__ misc,
                                                 uint32_t *__restrict__ claim,
                                                 int *__restrict__ queue,
                                                 int *__restrict__ final_, int H, int W)
{
    const int b = blockIdx.y;
    const int npix = H * W;
    const int lane = threadIdx.x;
    const int *P = parent + (long long)b * npix;
    const int *S = size + (long long)b * npix;
    const int *SL = small_list + (long long)b * npix;
    uint32_t *CL = claim + (long long)b * npix;
    int *F = final_ + (long long)b * npix;
    const int n_small = misc[b].n_small;
    const int first_kept = misc[b].first_kept;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int ddx[4] = {1, -1, 0, 0};
    const int ddy[4] = {0, 0, 1, -1};

    for (int i = blockIdx.x; i < n_small; i += gridDim.x) {
        const int r = SL[i];
        if (r < first_kept) {           // before the first kept component everything is label 0
            if (lane == 0) F[r] = -1;
            continue;
        }
        const int sz = S[r];
        int qoff = 0;
        if (lane == 0) qoff = atomicAdd(&misc[b].qalloc, sz);
        qoff = __shfl(qoff, 0);
        int *Q = queue + (long long)b * npix + qoff;
        if (lane == 0) {
            st_i32(Q, r);
            __hip_atomic_store(CL + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int head = 0, tail = 1;
        long long best = -1;            // (key << 32) | root of the outside neighbour
        while (head < tail) {
            const int cnt = min(64, tail - head);
            const bool act = lane < cnt;
            const int uidx = head + lane;
            const int u = act ? ld_i32(Q + uidx) : 0;
            const int uy = u / W, ux = u - uy * W;
            int v[4];
            bool cand[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                int xx = ux + ddx[d], yy = uy + ddy[d];
                bool inb = act && xx >= 0 && xx < W && yy >= 0 && yy < H;
                v[d] = yy * W + xx;
                cand[d] = false;
                if (inb) {
                    int rv = P[v[d]];
                    if (rv == r) {
                        cand[d] = __hip_atomic_load(CL + v[d], __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT) == INF_KEY;
                    } else if (rv < r) {
                        long long key = ((long long)(uidx * 4 + d) << 32) | (unsigned)rv;
                        if (key > best) best = key;
                    }
                }
            }
            uint32_t keyd[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                keyd[d] = (uint32_t)(uidx * 4 + d);
                if (cand[d]) atomicMin(CL + v[d], keyd[d]);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_s_waitcnt(0);      // all atomics of the wave have been performed
            int mywins = 0, before = 0, total = 0;
            bool win[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                win[d] = cand[d] && (__hip_atomic_load(CL + v[d], __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT) == keyd[d]);
                unsigned long long m = __ballot(win[d]);
                before += __popcll(m & below);
                total += __popcll(m);
            }
            int pos = tail + before;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if (win[d]) { st_i32(Q + pos + mywins, v[d]); ++mywins; }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_s_waitcnt(0);
            head += cnt;
            tail += total;
        }
        // wave max of `best`
        for (int o = 32; o > 0; o >>= 1) {
            long long t = __shfl_xor(best, o);
            if (t > best) best = t;
        }
        if (lane == 0) F[r] = (best < 0) ? -1 : -2 - (int)(best & 0xFFFFFFFFll);
    }
}

// final label of small components: follow the `adjacent` pointers to a kept component
__global__ __launch_bounds__(256) void k_conn_resolve(const int *__restrict__ small_list,
                                                      const ConnMisc *__restrict__ misc,
                                                      int *__restrict__ final_, int npix)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= misc[b].n_small) return;
    int *F = final_ + (long long)b * npix;
    const int r = small_list[(long long)b * npix + i];
    int f = ld_i32(F + r);
    while (f < -1) f = ld_i32(F + (-2 - f));   // pointer to a component with a smaller seed
    st_i32(F + r, f == -1 ? 0 : f);
}

__global__ __launch_bounds__(256) void k_conn_relabel(const int *__restrict__ parent,
                                                      const int *__restrict__ final_,
                                                      int32_t *__restrict__ out, int npix)
{
    const int b = blockIdx.y;
    const long long o = (long long)b * npix;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        int f = final_[o + parent[o + p]];
        out[o + p] = f < 0 ? 0 : f;
    }
}

__global__ void k_conn_init_misc(ConnMisc *misc, int B, int npix)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { misc[b].n_small = 0; misc[b].first_kept = npix; misc[b].qalloc = 0; misc[b].n_kept = 0; }
}

extern "C" int spa_enforce_connectivity(spa_ctx *ctx, const int32_t *labels_in, int32_t B,
                                        int32_t H, int32_t W, int32_t min_size, int32_t max_size,
                                        int32_t *labels_out, int32_t *n_labels, void *stream)
{
    SPA_ARG(ctx && labels_in && labels_out && n_labels && B > 0 && H > 0 && W > 0);
    SPA_ARG((long long)H * W < (1ll << 29));
    hipStream_t s = spa_stream(stream);
    const int npix = H * W;
    const size_t img = (size_t)B * npix * 4;
    int *parent, *size, *final_, *queue, *blk, *small;
    uint32_t *claim;
    ConnMisc *misc;
    int rc;
    const int nblk = (npix + SCAN_PX - 1) / SCAN_PX;
    if ((rc = spa_ws_reserve(ctx, WS_PARENT, img, (void **)&parent)) != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_SIZE, img, (void **)&size)) != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_FINAL, img, (void **)&final_)) != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_CLAIM, img, (void **)&claim)) != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_QUEUE, img, (void **)&queue)) != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_SMALL, img, (void **)&small)) != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_BLK, (size_t)B * nblk * 4, (void **)&blk)) != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_CONNMISC, (size_t)B * sizeof(ConnMisc), (void **)&misc)) != SPA_OK) return rc;

    SpaProfScope prof_(ctx, PROF_CONNECT, s);
    SPA_HIP(hipMemsetAsync(size, 0, img, s));
    SPA_HIP(hipMemsetAsync(claim, 0xFF, img, s));
    hipLaunchKernelGGL(k_conn_init_misc, dim3((B + 63) / 64), dim3(64), 0, s, misc, B, npix);
    dim3 gp((npix + 255) / 256, B);
    hipLaunchKernelGGL(k_ccl_init, gp, dim3(256), 0, s, labels_in, parent, W, npix);
    hipLaunchKernelGGL(k_ccl_merge, gp, dim3(256), 0, s, labels_in, parent, W, npix);
    hipLaunchKernelGGL(k_ccl_flatten, gp, dim3(256), 0, s, parent, size, npix);
    hipLaunchKernelGGL(k_conn_count, dim3(nblk, B), dim3(256), 0, s, parent, size, npix, min_size,
                       max_size, blk, nblk, small, misc, ctx->d_status);
    hipLaunchKernelGGL(k_conn_scan, dim3(B), dim3(256), 0, s, blk, nblk, misc, n_labels);
    hipLaunchKernelGGL(k_conn_number, dim3(nblk, B), dim3(256), 0, s, parent, size, npix, min_size,
                       blk, nblk, final_);
    hipLaunchKernelGGL(k_conn_bfs, dim3(512, B), dim3(64), 0, s, parent, size, small, misc, claim,
                       queue, final_, H, W);
    // the number of small roots lives on the device: launch over the worst case (one root per
    // pixel) and let surplus workgroups exit at once
    hipLaunchKernelGGL(k_conn_resolve, dim3((npix + 255) / 256, B), dim3(256), 0, s, small, misc,
                       final_, npix);
    int gr = (npix + 255) / 256;
    if (gr > 2048) gr = 2048;
    hipLaunchKernelGGL(k_conn_relabel, dim3(gr, B), dim3(256), 0, s, parent, final_, labels_out, npix);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
