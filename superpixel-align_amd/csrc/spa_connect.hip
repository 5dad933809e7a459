// Connectivity enforcement of SLIC labels on gfx950, bit exact with skimage's
// _enforce_label_connectivity_cython (a sequential scan-order breadth-first relabelling).
//
// The sequential algorithm, restated as order-free facts (HISTORY.md section 4 has the long form):
//   * without a max_size cut, the components it discovers are exactly the 4-connected
//     components of equal input label, visited in order of their first pixel in raster
//     order (the "seed" = minimum raster index of the component);
//   * a component with >= min_size pixels is KEPT and receives the next free label, so its
//     label is the number of kept components with a smaller seed;
//   * a component C with < min_size pixels takes the label of `adjacent`: the LAST pixel, in
//     the BFS visiting order of C (neighbour order +x,-x,+y,-y), that lies outside C in a
//     component D with seed(D) < seed(C) — whatever label D ended up with — or 0 when there
//     is none, or when C precedes the first kept component (everything before it is 0).
//   So: connected components, a prefix sum over raster order for the kept labels, one wavefront
//   per small component replaying its BFS in the exact queue order, then pointer chasing D -> label.
//   A component that reaches max_size is cut by the sequential algorithm in BFS order: replayed
//   per component before the numbering (k_conn_split).
//
// Components are found on horizontal RUNS, not pixels: a workgroup per image row finds the runs of
// equal label (one streaming pass: 4 bytes read, 4 written per pixel), every pixel's parent is the first
// pixel of its run, and only the run starts (~25 per row on clean images, ~300 on noisy ones, against
// 2 048 pixels) are nodes of the union-find that links vertically adjacent runs of one label.  Sizes,
// the raster-order numbering, the classification and the bounding boxes of the small components are
// run-level passes too (one wave per row); a pixel's component root is parent[parent[pixel]].  The
// first round's pixel-level passes (k_conn_count / k_conn_number / k_small_bbox) remain for images that
// contain a component above max_size, whose pieces do not respect run boundaries.
#include "spa_common.h"

#define INF_KEY 0xFFFFFFFFu

struct ConnMisc {
    int n_small;       // number of tiny (<= LANE_MAX pixels) small component roots
    int first_kept;    // smallest kept root (npix if none)
    int qalloc;        // BFS queue allocation cursor
    int n_kept;        // number of kept components
    int n_todo1;       // big-small components of the 48 KB LDS tier
    int n_todo2;       // ... of the global-memory tier (too large for LDS, or a ring overflow)
    int n_big;         // small components with more than LANE_MAX pixels
    int n_over;        // components larger than max_size (cut in BFS order by the reference)
    int n_a;           // big-small components of the 16 KB LDS tier
    int n_c;           // ... of the 80 KB LDS tier (rare: boxes above 40 K pixels)
    int pad_[2];
    int nb[3][8];      // LDS tiers: components per size class (2^(c+5) .. 2^(c+6) pixels), replayed largest first
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

// find with path halving: every visited node is re-pointed at its grandparent (atomicMin: parents only
// ever move to smaller indices of the same set, whatever other waves do meanwhile)
__device__ __forceinline__ int uf_find_halve(int *parent, int i)
{
    for (;;) {
        const int p = ld_i32(parent + i);
        if (p == i) return i;
        const int g = ld_i32(parent + p);
        if (g == p) return p;
        atomicMin(parent + i, g);
        i = g;
    }
}

__device__ __forceinline__ void uf_merge(int *parent, int a, int b)
{
    for (;;) {
        a = uf_find_halve(parent, a);
        b = uf_find_halve(parent, b);
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

// ---------------------------------------------------------------------------------------
// Run-level connected components.  Everything a run needs lives in COMPACT tables at the front of its image
// row (run id = y * W + k for the k-th run of row y): start x, label, union-find parent (a run id), size
// (at root runs).  Run ids are ordered like the first pixels of the runs, so the smallest run id of a
// component is the run of its seed pixel, and comparing run ids compares seeds.  The per-pixel arrays are
// touched only by the streaming passes: k_run_rows reads the labels, k_run_expand writes parent[pixel] =
// seed pixel of the pixel's component for every pixel (the representation the BFS replays, the oversize
// split and the relabel pass work on).
// ---------------------------------------------------------------------------------------
#define RUN_CHUNK 2048          // pixels of a row handled per pass of the row's workgroup (8 per thread)

// one workgroup per (row, image): runs[rid] = start x, runlab[rid] = label, up[rid] = rid, rsz[rid] = 0,
// rowcnt[row] = number of runs
__global__ __launch_bounds__(256) void k_run_rows(const int32_t *__restrict__ lab, int *__restrict__ runs,
                                                  int *__restrict__ runlab, int *__restrict__ up,
                                                  int *__restrict__ rsz, int *__restrict__ rowcnt, int H, int W)
{
    __shared__ int w_cnt[4];
    __shared__ int carry_cnt;
    const int b = blockIdx.y, y = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long npix = (long long)H * W;
    const long long rowoff = b * npix + (long long)y * W;
    const int32_t *L = lab + rowoff;
    int *R = runs + rowoff, *RL = runlab + rowoff, *UP = up + rowoff, *SZ = rsz + rowoff;
    const int ridbase = y * W;
    const bool vec = (W & 3) == 0;
    if (tid == 0) carry_cnt = 0;
    __syncthreads();
    for (int x0 = 0; x0 < W; x0 += RUN_CHUNK) {
        const int xb = x0 + tid * 8;
        int l[8];
        if (vec && xb + 7 < W) {
            const int4 a = *(const int4 *)(L + xb), c = *(const int4 *)(L + xb + 4);
            l[0] = a.x; l[1] = a.y; l[2] = a.z; l[3] = a.w; l[4] = c.x; l[5] = c.y; l[6] = c.z; l[7] = c.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) l[i] = (xb + i < W) ? L[xb + i] : -1;
        }
        const int lprev = (xb > 0 && xb < W) ? L[xb - 1] : -2;
        int cnt = 0;
        bool st[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int x = xb + i;
            st[i] = x < W && (x == 0 || l[i] != (i == 0 ? lprev : l[i - 1]));
            cnt += st[i] ? 1 : 0;
        }
        int ps = cnt;                                     // inclusive prefix of the run-start counts
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int c = __shfl_up(ps, o);
            if (lane >= o) ps += c;
        }
        if (lane == 63) w_cnt[wv] = ps;
        __syncthreads();
        int before = carry_cnt;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < wv) before += w_cnt[i];
        int k = before + ps - cnt;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (st[i]) {
                R[k] = xb + i;
                RL[k] = l[i];
                UP[k] = ridbase + k;
                SZ[k] = 0;
                ++k;
            }
        }
        __syncthreads();
        if (tid == 255) carry_cnt = before + ps;
        __syncthreads();
    }
    if (tid == 0) rowcnt[(long long)b * H + y] = carry_cnt;
}

// ---- vertical links, two levels.  A strip of STRIP_ROWS rows holds a few thousand runs: its union-find lives
// in LDS (k_run_strip), where a hop costs an LDS access instead of an L2 round trip and the sizes and boxes
// of the strip-local components are summed with LDS atomics.  Only the row pairs that straddle two strips
// (one in STRIP_ROWS) are linked through global memory (k_run_border), on a forest that is already flat
// inside every strip, and only the strip-local roots carry size/box contributions to their final root
// (k_run_flatten).  Tables per run id, valid at root runs: rsz = pixels, ry1 / rx0 / rx1 = bounding box
// (its first row is the root run's own row: the root is the component's first run in raster order).
#define STRIP_ROWS 8
#define STRIP_CAP 4096          // runs per strip held in LDS; a busier strip links its rows through global memory

__device__ __forceinline__ int lds_uf_find(int *lp, int i)
{
    for (;;) {
        const int p = lp[i];
        if (p == i) return i;
        const int g = lp[p];
        if (g == p) return p;
        atomicMin(lp + i, g);
        i = g;
    }
}

__device__ __forceinline__ void lds_uf_merge(int *lp, int a, int b)
{
    for (;;) {
        a = lds_uf_find(lp, a);
        b = lds_uf_find(lp, b);
        if (a == b) return;
        if (a > b) { int t = a; a = b; b = t; }
        const int old = atomicMin(lp + b, a);
        if (old == b) return;
        b = old;
    }
}

// link run k of row y (xs..xe, label l) to the runs of the same label it touches in the row above (global tables)
__device__ __forceinline__ void run_link_up_global(int *UP, const int *Rp, const int *RLp, int cntp, int y, int k,
                                                   int xs, int xe, int l, int W)
{
    int lo = 0, hi = cntp - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (Rp[mid] <= xs) lo = mid; else hi = mid - 1;
    }
    for (int j = lo; j < cntp; ++j) {
        if (Rp[j] > xe) break;
        if (RLp[j] == l) uf_merge(UP, (y - 1) * W + j, y * W + k);
    }
}

__global__ __launch_bounds__(256) void k_run_strip(const int *__restrict__ runlab, int *__restrict__ up,
                                                   int *__restrict__ rsz, int *__restrict__ ry1,
                                                   int *__restrict__ rx0, int *__restrict__ rx1,
                                                   const int *__restrict__ runs, const int *__restrict__ rowcnt,
                                                   int H, int W)
{
    __shared__ int lp[STRIP_CAP];      // local parent
    __shared__ int la[STRIP_CAP];      // links: label          sums: pixels of the local component
    __shared__ int lb[STRIP_CAP];      // links: start x        sums: row << 28 | x0 << 14 | x1, kept current with a CAS
    __shared__ int off_s[STRIP_ROWS + 1];
    const int b = blockIdx.y, ys = blockIdx.x * STRIP_ROWS;
    const int nrows = min(STRIP_ROWS, H - ys);
    const int tid = threadIdx.x;
    const long long npix = (long long)H * W;
    const int *RC = rowcnt + (long long)b * H + ys;
    int off[STRIP_ROWS + 1];           // statically indexed copy (row_of); off_s for the dynamic look-ups
    off[0] = 0;
#pragma unroll
    for (int j = 0; j < STRIP_ROWS; ++j) off[j + 1] = off[j] + (j < nrows ? RC[j] : 0);
    if (tid <= STRIP_ROWS) {
        int v = 0;
#pragma unroll
        for (int j = 0; j <= STRIP_ROWS; ++j) v = (j == tid) ? off[j] : v;
        off_s[tid] = v;
    }
    const int total = off[STRIP_ROWS];
    auto row_of = [&](int i) {
        int j = 0;
#pragma unroll
        for (int t = 1; t < STRIP_ROWS; ++t) j += (i >= off[t]) ? 1 : 0;
        return j;
    };
    int *UP = up + b * npix;
    const long long base = b * npix + (long long)ys * W;
    const int *R0 = runs + base, *RL0 = runlab + base;
    int *SZ = rsz + base, *Y1 = ry1 + base, *X0 = rx0 + base, *X1 = rx1 + base;
    __syncthreads();

    if (total > STRIP_CAP) {
        // every run stays its own local component; the rows are linked through global memory
        for (int j = 0; j < nrows; ++j) {
            const int cnt = off_s[j + 1] - off_s[j];
            const int cntp = j > 0 ? off_s[j] - off_s[j - 1] : 0;
            const int *R = R0 + (long long)j * W, *RL = RL0 + (long long)j * W;
            for (int k = tid; k < cnt; k += 256) {
                const int xs = R[k];
                const int xe = (k + 1 < cnt ? R[k + 1] : W) - 1;
                SZ[j * W + k] = xe - xs + 1;
                Y1[j * W + k] = ys + j; X0[j * W + k] = xs; X1[j * W + k] = xe;
                if (j > 0) run_link_up_global(UP, R - W, RL - W, cntp, ys + j, k, xs, xe, RL[k], W);
            }
        }
        return;
    }
    // ---- stage the strip's runs
    for (int i = tid; i < total; i += 256) {
        const int j = row_of(i);
        const int k = i - off_s[j];
        lp[i] = i;
        la[i] = RL0[(long long)j * W + k];
        lb[i] = R0[(long long)j * W + k];
    }
    __syncthreads();
    // ---- links between the rows of the strip
    for (int i = off[1] + tid; i < total; i += 256) {
        const int j = row_of(i);
        const int o0 = off_s[j - 1], o1 = off_s[j], o2 = off_s[j + 1];
        const int xs = lb[i];
        const int xe = (i + 1 < o2 ? lb[i + 1] : W) - 1;
        const int l = la[i];
        int lo = o0, hi = o1 - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (lb[mid] <= xs) lo = mid; else hi = mid - 1;
        }
        for (int q = lo; q < o1; ++q) {
            if (lb[q] > xe) break;
            if (la[q] == l) lds_uf_merge(lp, q, i);
        }
    }
    __syncthreads();
    // ---- flatten; every thread keeps its runs' extents in registers while la / lb change their meaning
    constexpr int PER = STRIP_CAP / 256;
    int rt[PER], xs_[PER], xe_[PER];
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int i = tid + t * 256;
        rt[t] = -1; xs_[t] = 0; xe_[t] = 0;
        if (i < total) {
            const int j = row_of(i);
            rt[t] = lds_uf_find(lp, i);
            xs_[t] = lb[i];
            xe_[t] = (i + 1 < off_s[j + 1] ? lb[i + 1] : W) - 1;
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int i = tid + t * 256;
        if (i < total) { la[i] = 0; lb[i] = 0x3FFF << 14; }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int i = tid + t * 256;
        if (i >= total) continue;
        const unsigned j = (unsigned)row_of(i);
        const int r = rt[t];
        atomicAdd(la + r, xe_[t] - xs_[t] + 1);
        unsigned old = (unsigned)lb[r];
        for (;;) {
            const unsigned x0 = min((old >> 14) & 0x3FFFu, (unsigned)xs_[t]), x1 = max(old & 0x3FFFu, (unsigned)xe_[t]);
            const unsigned want = (max(old >> 28, j) << 28) | (x0 << 14) | x1;
            if (want == old) break;
            const unsigned seen = atomicCAS((unsigned *)lb + r, old, want);
            if (seen == old) break;
            old = seen;
        }
    }
    __syncthreads();
    // ---- out: every run points at its local root; the local roots carry the sums
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int i = tid + t * 256;
        if (i >= total) continue;
        const int j = row_of(i);
        const int r = rt[t];
        const int jr = row_of(r);
        const int o = j * W + (i - off_s[j]);
        UP[(long long)ys * W + o] = (ys + jr) * W + (r - off_s[jr]);
        if (r == i) {
            const unsigned bx = (unsigned)lb[i];
            SZ[o] = la[i];
            Y1[o] = ys + (int)(bx >> 28);
            X0[o] = (int)((bx >> 14) & 0x3FFFu);
            X1[o] = (int)(bx & 0x3FFFu);
        }
    }
}

// the row pairs between two strips: one workgroup per pair, both rows' run lists staged in LDS
__global__ __launch_bounds__(256) void k_run_border(const int *__restrict__ runlab, int *__restrict__ up,
                                                    const int *__restrict__ runs, const int *__restrict__ rowcnt,
                                                    int H, int W)
{
    extern __shared__ int lds_m[];                 // Rp [W] | Lp [W] | R [W]
    const int b = blockIdx.y, y = (blockIdx.x + 1) * STRIP_ROWS;
    const int tid = threadIdx.x;
    const long long npix = (long long)H * W;
    int *UP = up + b * npix;
    const int *RC = rowcnt + (long long)b * H;
    const int *R = runs + b * npix + (long long)y * W;
    const int *RL = runlab + b * npix + (long long)y * W;
    const int *Rp = R - W, *RLp = RL - W;
    const int cnt = RC[y], cntp = RC[y - 1];
    int *sRp = lds_m, *sLp = lds_m + W, *sR = lds_m + 2 * W;
    for (int k = tid; k < cntp; k += 256) {
        sRp[k] = Rp[k];
        sLp[k] = RLp[k];
    }
    for (int k = tid; k < cnt; k += 256) sR[k] = R[k];
    __syncthreads();
    for (int k = tid; k < cnt; k += 256) {
        const int xs = sR[k];
        const int xe = (k + 1 < cnt ? sR[k + 1] : W) - 1;
        run_link_up_global(UP, sRp, sLp, cntp, y, k, xs, xe, RL[k], W);
    }
}

// one wave per row: every run learns its root run; the strip-local roots hand their sums to the root
__global__ __launch_bounds__(256) void k_run_flatten(int *__restrict__ up, int *__restrict__ rsz,
                                                     int *__restrict__ ry1, int *__restrict__ rx0,
                                                     int *__restrict__ rx1, const int *__restrict__ rowcnt,
                                                     int H, int W)
{
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const long long npix = (long long)H * W;
    int *UP = up + b * npix;
    int *SZ = rsz + b * npix, *Y1 = ry1 + b * npix, *X0 = rx0 + b * npix, *X1 = rx1 + b * npix;
    const int *RC = rowcnt + (long long)b * H;
    for (int y = blockIdx.x * 4 + (threadIdx.x >> 6); y < H; y += gridDim.x * 4) {
        const int cnt = RC[y];
        for (int k = lane; k < cnt; k += 64) {
            const int rid = y * W + k;
            const int p = ld_i32(UP + rid);
            const int mine = SZ[rid];             // > 0: a strip-local root (nobody adds to it unless it is THE root)
            if (p == rid) continue;               // a root (of its strip and of the forest)
            const int r = uf_find_halve(UP, rid);
            st_i32(UP + rid, r);
            if (mine > 0) {
                atomicAdd(SZ + r, mine);
                atomicMax(Y1 + r, Y1[rid]);
                atomicMin(X0 + r, X0[rid]);
                atomicMax(X1 + r, X1[rid]);
            }
        }
    }
}

// one workgroup per (row, image): parent[pixel] = seed pixel of the pixel's component, for every pixel of the
// row (streaming write); size[seed pixel] = size of the component (written by the root run)
__global__ __launch_bounds__(256) void k_run_expand(const int *__restrict__ up, const int *__restrict__ rsz,
                                                    const int *__restrict__ runs, const int *__restrict__ rowcnt,
                                                    int *__restrict__ parent, int *__restrict__ size, int H, int W)
{
    extern __shared__ int lds_e[];                 // start x [W] | seed pixel [W]
    const int b = blockIdx.y, y = blockIdx.x;
    const int tid = threadIdx.x;
    const long long npix = (long long)H * W;
    const int *UP = up + b * npix, *SZ = rsz + b * npix, *RA = runs + b * npix;
    const int *R = RA + (long long)y * W;
    int *P = parent + b * npix + (long long)y * W;
    int *S = size + b * npix;
    const int cnt = rowcnt[(long long)b * H + y];
    int *sx = lds_e, *sp = lds_e + W;
    for (int k = tid; k < cnt; k += 256) {
        const int rid = y * W + k;
        const int r = UP[rid];                                  // root run id = ry * W + rk
        const int ry = r / W;
        const int seed = ry * W + RA[r];                         // its first pixel
        sx[k] = R[k];
        sp[k] = seed;
        if (r == rid) S[seed] = SZ[rid];
    }
    __syncthreads();
    const bool vec = (W & 3) == 0;
    for (int x0 = 0; x0 < W; x0 += RUN_CHUNK) {
        const int xb = x0 + tid * 8;
        if (xb >= W) continue;
        // run of pixel xb: last start <= xb
        int lo = 0, hi = cnt - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (sx[mid] <= xb) lo = mid; else hi = mid - 1;
        }
        int k = lo;
        int nxt = (k + 1 < cnt) ? sx[k + 1] : W;
        int cur = sp[k];
        int out[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int x = xb + i;
            if (x >= nxt && x < W) { ++k; cur = sp[k]; nxt = (k + 1 < cnt) ? sx[k + 1] : W; }
            out[i] = cur;
        }
        if (vec && xb + 7 < W) {
            *(int4 *)(P + xb) = make_int4(out[0], out[1], out[2], out[3]);
            *(int4 *)(P + xb + 4) = make_int4(out[4], out[5], out[6], out[7]);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (xb + i < W) P[xb + i] = out[i];
        }
    }
}

// ---------------------------------------------------------------------------------------
// Components larger than max_size.  The reference stops the breadth-first growth of a component
// at max_size pixels; the pixels it did not reach are found again later by the raster scan and
// start new components (again capped).  That decomposition depends only on the component itself,
// so it is replayed here, one wavefront per oversize component, before anything else looks at
// the roots: piece after piece (seed = first pixel of the component, in raster order, that no
// earlier piece took; growth in exact queue order, truncated at max_size), then the parents of
// the pixels are rewritten to their piece's seed and the sizes to the piece sizes.  Rare (a
// component must exceed three times the mean superpixel area), so it runs from global memory.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conn_find_oversize(const int *__restrict__ parent,
                                                            const int *__restrict__ size, int npix,
                                                            int max_size, int *__restrict__ over_list,
                                                            ConnMisc *__restrict__ misc)
{
    const int b = blockIdx.y;
    const int *P = parent + (long long)b * npix;
    const int *S = size + (long long)b * npix;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        if (P[p] == p && S[p] > max_size) {
            int k = atomicAdd(&misc[b].n_over, 1);
            over_list[(long long)b * npix + k] = p;
        }
    }
}

// run-level twin of k_conn_find_oversize: root runs above max_size -> their seed pixels
__global__ __launch_bounds__(256) void k_run_find_oversize(const int *__restrict__ up, const int *__restrict__ rsz,
                                                           const int *__restrict__ runs,
                                                           const int *__restrict__ rowcnt, int H, int W,
                                                           int max_size, int *__restrict__ over_list,
                                                           ConnMisc *__restrict__ misc)
{
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const long long npix = (long long)H * W;
    const int *UP = up + b * npix, *SZ = rsz + b * npix;
    const int *RC = rowcnt + (long long)b * H;
    for (int y = blockIdx.x * 4 + (threadIdx.x >> 6); y < H; y += gridDim.x * 4) {
        const int *R = runs + b * npix + (long long)y * W;
        const int cnt = RC[y];
        for (int k = lane; k < cnt; k += 64) {
            const int rid = y * W + k;
            if (UP[rid] == rid && SZ[rid] > max_size) {
                const int i = atomicAdd(&misc[b].n_over, 1);
                over_list[b * npix + i] = y * W + R[k];
            }
        }
    }
}

__global__ __launch_bounds__(64) void k_conn_split(int *__restrict__ parent, int *__restrict__ size,
                                                   const int *__restrict__ over_list,
                                                   ConnMisc *__restrict__ misc,
                                                   uint32_t *__restrict__ claim,
                                                   int *__restrict__ queue, int H, int W, int max_size)
{
    const int b = blockIdx.y;
    const int npix = H * W;
    const int lane = threadIdx.x;
    int *P = parent + (long long)b * npix;
    int *S = size + (long long)b * npix;
    uint32_t *CL = claim + (long long)b * npix;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int ddx[4] = {1, -1, 0, 0};
    const int ddy[4] = {0, 0, 1, -1};
    const int n_over = misc[b].n_over;
    for (int it = blockIdx.x; it < n_over; it += gridDim.x) {
        const int r0 = over_list[(long long)b * npix + it];
        const int total = S[r0];
        int qbase = 0;
        if (lane == 0) qbase = atomicAdd(&misc[b].qalloc, total);
        qbase = __shfl(qbase, 0);
        int *Q0 = queue + (long long)b * npix + qbase;     // all pieces of this component, back to back
        int done = 0;                                       // pixels assigned to pieces so far
        int cursor = r0;
        while (done < total) {
            // seed: first pixel >= cursor that belongs to the component and to no piece yet
            int seed = -1;
            for (int p0 = cursor; p0 < npix && seed < 0; p0 += 64) {
                const int p = p0 + lane;
                bool hit = false;
                if (p < npix && ld_i32(P + p) == r0)
                    hit = __hip_atomic_load(CL + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == INF_KEY;
                const unsigned long long m = __ballot(hit);
                if (m) seed = p0 + __ffsll((long long)m) - 1;
            }
            if (seed < 0) break;                           // cannot happen: sizes are exact
            int *Q = Q0 + done;
            if (lane == 0) {
                st_i32(Q, seed);
                __hip_atomic_store(CL + seed, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_s_waitcnt(0);
            int head = 0, tail = 1;
            while (head < tail && tail < max_size) {
                const int cnt = min(64, tail - head);
                const bool act = lane < cnt;
                const int uidx = head + lane;
                const int u = act ? ld_i32(Q + uidx) : 0;
                const int uy = u / W, ux = u - uy * W;
                int v[4];
                bool cand[4];
                uint32_t keyd[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    int xx = ux + ddx[d], yy = uy + ddy[d];
                    bool inb = act && xx >= 0 && xx < W && yy >= 0 && yy < H;
                    v[d] = yy * W + xx;
                    keyd[d] = (uint32_t)(uidx * 4 + d);
                    cand[d] = inb && ld_i32(P + v[d]) == r0 &&
                              __hip_atomic_load(CL + v[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == INF_KEY;
                }
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    if (cand[d]) atomicMin(CL + v[d], keyd[d]);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_s_waitcnt(0);
                int before = 0, tot = 0;
                bool win[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    win[d] = cand[d] && (__hip_atomic_load(CL + v[d], __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT) == keyd[d]);
                    unsigned long long m = __ballot(win[d]);
                    before += __popcll(m & below);
                    tot += __popcll(m);
                }
                const int room = max_size - tail;          // the reference stops at max_size pixels
                int pos = before;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    if (win[d]) {
                        if (pos < room) st_i32(Q + tail + pos, v[d]);
                        else __hip_atomic_store(CL + v[d], INF_KEY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ++pos;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_s_waitcnt(0);
                head += cnt;
                tail += min(tot, room);
            }
            // this piece: pixels Q[0..tail); they keep their claim (= taken) until the end
            for (int i = lane; i < tail; i += 64) st_i32(P + ld_i32(Q + i), -2 - seed);   // provisional
            if (lane == 0) st_i32(S + seed, tail);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_s_waitcnt(0);
            done += tail;
            cursor = seed + 1;
        }
        // finalise: parents -> piece seeds, claims released for the later global tier
        for (int i = lane; i < done; i += 64) {
            const int v = ld_i32(Q0 + i);
            st_i32(P + v, -2 - ld_i32(P + v));
            __hip_atomic_store(CL + v, INF_KEY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_s_waitcnt(0);
    }
    // queue allocations are released for the BFS tiers that follow (they allocate again)
}

__global__ void k_conn_reset_qalloc(ConnMisc *misc, int B)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) misc[b].qalloc = 0;
}

// ---------------------------------------------------------------------------------------
// Raster-order numbering.  Roots fall in three classes:
//   kept  (size >= min_size)            : label = number of kept roots before it (prefix sum)
//   tiny  (size <= LANE_MAX)            : listed in raster order (prefix sum), one LANE each
//   big-small (LANE_MAX < size < min)   : few; appended to `big_list` with an atomic counter
// A noisy image has ~10^5 tiny components (single boundary pixels), so nothing on their path
// may cost one atomic on a shared word per component.
// ---------------------------------------------------------------------------------------
#define TINY_DONE 0x40000000     // tiny-list entry whose `adjacent` is already in final_ (H*W < 2^29)
#define SCAN_PX 1024
#define SBOX_CAP 65536      // big-small components with an index below this get a bounding box
#define LANE_MAX 16

__device__ __forceinline__ void load4_roots(const int *P, int base, int npix, bool aligned, int pv[4])
{
    if (base + 3 < npix && aligned) {
        int4 t = *(const int4 *)(P + base);
        pv[0] = t.x; pv[1] = t.y; pv[2] = t.z; pv[3] = t.w;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) pv[i] = (base + i < npix) ? P[base + i] : -1;
    }
}

// pass 1: per-block counts of kept and tiny roots
__global__ __launch_bounds__(256) void k_conn_count(const int *__restrict__ parent,
                                                    const int *__restrict__ size, int npix,
                                                    int min_size, int max_size,
                                                    int *__restrict__ blk, int nblk,
                                                    ConnMisc *__restrict__ misc,
                                                    uint32_t *__restrict__ status)
{
    __shared__ int wk[4], wt[4], wf[4];
    const int b = blockIdx.y;
    if (misc[b].n_over == 0) return;             // run-level path (k_run_count)
    const int *P = parent + (long long)b * npix;
    const int *S = size + (long long)b * npix;
    const int base = blockIdx.x * SCAN_PX + threadIdx.x * 4;
    int pv[4];
    load4_roots(P, base, npix, (((long long)b * npix) & 3) == 0, pv);
    int ck = 0, ct = 0, fk = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (pv[i] != base + i) continue;
        const int sz = S[base + i];
        if (sz >= min_size) {
            ++ck;
            fk = min(fk, base + i);
        } else if (sz <= LANE_MAX) {
            ++ct;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        ck += __shfl_down(ck, o); ct += __shfl_down(ct, o); fk = min(fk, __shfl_down(fk, o));
    }
    if ((threadIdx.x & 63) == 0) { wk[threadIdx.x >> 6] = ck; wt[threadIdx.x >> 6] = ct; wf[threadIdx.x >> 6] = fk; }
    __syncthreads();
    if (threadIdx.x == 0) {
        blk[((long long)b * 2 + 0) * nblk + blockIdx.x] = wk[0] + wk[1] + wk[2] + wk[3];
        blk[((long long)b * 2 + 1) * nblk + blockIdx.x] = wt[0] + wt[1] + wt[2] + wt[3];
        int f = min(min(wf[0], wf[1]), min(wf[2], wf[3]));
        if (f != 0x7fffffff) atomicMin(&misc[b].first_kept, f);
    }
}

// pass 2: exclusive scans of the two block-count arrays of one image (single workgroup)
__global__ __launch_bounds__(256) void k_conn_scan(int *__restrict__ blk, int nblk,
                                                   ConnMisc *__restrict__ misc,
                                                   int32_t *__restrict__ n_labels, int run_path)
{
    __shared__ int part[256];
    const int b = blockIdx.x;
    if ((misc[b].n_over == 0) != (run_path != 0)) return;
    const int per = (nblk + 255) / 256;
    const int lo = threadIdx.x * per, hi = min(nblk, lo + per);
    for (int which = 0; which < 2; ++which) {
        int *B_ = blk + ((long long)b * 2 + which) * nblk;
        int s = 0;
        for (int i = lo; i < hi; ++i) s += B_[i];
        part[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0;
            for (int i = 0; i < 256; ++i) { int t = part[i]; part[i] = run; run += t; }
            if (which == 0) { misc[b].n_kept = run; n_labels[b] = run > 0 ? run : 1; }
            else misc[b].n_small = run;
        }
        __syncthreads();
        int run = part[threadIdx.x];
        for (int i = lo; i < hi; ++i) { int t = B_[i]; B_[i] = run; run += t; }
        __syncthreads();
    }
}

// pass 3: kept roots get their label, tiny roots their list position (both by prefix sums, so
// in raster order); big-small roots are appended to big_list and their bounding box is seeded
__global__ __launch_bounds__(256) void k_conn_number(const int *__restrict__ parent,
                                                     const int *__restrict__ size, int npix, int W,
                                                     int min_size, const int *__restrict__ blk,
                                                     int nblk, int *__restrict__ final_,
                                                     int *__restrict__ tiny_list,
                                                     int *__restrict__ big_list,
                                                     int *__restrict__ sbox,
                                                     ConnMisc *__restrict__ misc)
{
    __shared__ int wsk[4], wst[4];
    const int b = blockIdx.y;
    if (misc[b].n_over == 0) return;             // run-level path (k_run_number)
    const int *P = parent + (long long)b * npix;
    const int *S = size + (long long)b * npix;
    int *F = final_ + (long long)b * npix;
    const int base = blockIdx.x * SCAN_PX + threadIdx.x * 4;
    int pv[4];
    load4_roots(P, base, npix, (((long long)b * npix) & 3) == 0, pv);
    int cls[4];            // 0 none, 1 kept, 2 tiny, 3 big-small
    int ck = 0, ct = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        cls[i] = 0;
        if (pv[i] == base + i) {
            const int sz = S[base + i];
            cls[i] = sz >= min_size ? 1 : (sz <= LANE_MAX ? 2 : 3);
        }
        ck += cls[i] == 1; ct += cls[i] == 2;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int ik = ck, it = ct;
    for (int o = 1; o < 64; o <<= 1) {
        int a = __shfl_up(ik, o), c = __shfl_up(it, o);
        if (lane >= o) { ik += a; it += c; }
    }
    if (lane == 63) { wsk[wv] = ik; wst[wv] = it; }
    __syncthreads();
    int offk = blk[((long long)b * 2 + 0) * nblk + blockIdx.x];
    int offt = blk[((long long)b * 2 + 1) * nblk + blockIdx.x];
    for (int i = 0; i < wv; ++i) { offk += wsk[i]; offt += wst[i]; }
    int rk = offk + ik - ck, rt = offt + it - ct;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = base + i;
        if (cls[i] == 1) F[p] = rk++;
        else if (cls[i] == 2) tiny_list[(long long)b * npix + rt++] = p;
        else if (cls[i] == 3) {
            const int slot = atomicAdd(&misc[b].n_big, 1);
            big_list[(long long)b * npix + slot] = p;
            F[p] = slot;
            if (slot < SBOX_CAP) {
                int *bb = sbox + ((long long)b * SBOX_CAP + slot) * 4;
                const int y = p / W, x = p - y * W;
                bb[0] = y; bb[1] = y; bb[2] = x; bb[3] = x;
            }
        }
    }
}

// bounding boxes of the big-small components (only their pixels issue atomics, one set per
// distinct root per wave)
__global__ __launch_bounds__(256) void k_small_bbox(const int *__restrict__ parent,
                                                    const int *__restrict__ size,
                                                    const int *__restrict__ final_, int W, int npix,
                                                    int min_size, int *__restrict__ sbox,
                                                    const ConnMisc *__restrict__ misc)
{
    const int b = blockIdx.y;
    if (misc[b].n_over == 0) return;             // run-level path (k_run_bbox)
    const int lane = threadIdx.x & 63;
    const int *P = parent + (long long)b * npix;
    const int *S = size + (long long)b * npix;
    const int *F = final_ + (long long)b * npix;
    for (int p0 = blockIdx.x * 256; p0 < npix; p0 += gridDim.x * 256) {
        const int p = p0 + threadIdx.x;
        int r = -1, y = 0, x = 0;
        if (p < npix) {
            r = P[p];
            const int sz = S[r];
            if (sz >= min_size || sz <= LANE_MAX) r = -1;
            y = p / W; x = p - y * W;
        }
        // first pixel of the wave and whether its 64 pixels stay inside one row and the image
        const int pw = p0 + (threadIdx.x & ~63);
        const int wy = pw / W, wx = pw - wy * W;
        const bool one_row = (wx + 63 < W) && (pw + 63 < npix);
        unsigned long long todo = __ballot(r >= 0);
        while (todo) {
            int leader = __ffsll((long long)todo) - 1;
            int rr = __shfl(r, leader);
            unsigned long long same = __ballot(r == rr);
            bool mine = (r == rr);
            int ya, yb, xa, xb;
            if (one_row) {
                // the wave's 64 pixels lie in one image row: the extent of the root's lanes is the
                // extent of its pixels (no cross-lane reduction)
                const int first = __ffsll((long long)same) - 1, last = 63 - __clzll((long long)same);
                ya = yb = wy; xa = wx + first; xb = wx + last;
            } else {
                ya = mine ? y : 0x7fffffff; yb = mine ? y : -1; xa = mine ? x : 0x7fffffff; xb = mine ? x : -1;
                for (int o = 32; o > 0; o >>= 1) {
                    ya = min(ya, __shfl_xor(ya, o)); yb = max(yb, __shfl_xor(yb, o));
                    xa = min(xa, __shfl_xor(xa, o)); xb = max(xb, __shfl_xor(xb, o));
                }
            }
            if (lane == leader) {
                int slot = F[rr];
                if (slot < SBOX_CAP) {
                    int *bb = sbox + ((long long)b * SBOX_CAP + slot) * 4;
                    atomicMin(bb + 0, ya); atomicMax(bb + 1, yb);
                    atomicMin(bb + 2, xa); atomicMax(bb + 3, xb);
                }
            }
            todo &= ~same;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Run-level numbering (images without an oversize component): one wave per row; the roots of a row
// are met in raster order, rows are ordered by an exclusive scan of their counts (k_conn_scan).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_run_count(const int *__restrict__ up, const int *__restrict__ rsz,
                                                   const int *__restrict__ runs, const int *__restrict__ rowcnt,
                                                   int H, int W, int min_size, int *__restrict__ blk, int nblk,
                                                   ConnMisc *__restrict__ misc)
{
    const int b = blockIdx.y;
    if (misc[b].n_over != 0) return;
    const int lane = threadIdx.x & 63;
    const long long npix = (long long)H * W;
    const int *UP = up + b * npix, *SZ = rsz + b * npix;
    const int *RC = rowcnt + (long long)b * H;
    for (int y = blockIdx.x * 4 + (threadIdx.x >> 6); y < H; y += gridDim.x * 4) {
        const int *R = runs + b * npix + (long long)y * W;
        const int cnt = RC[y];
        int ck = 0, ct = 0, fk = 0x7fffffff;
        for (int k = lane; k < cnt; k += 64) {
            const int rid = y * W + k;
            if (UP[rid] != rid) continue;
            const int sz = SZ[rid];
            if (sz >= min_size) { ++ck; fk = min(fk, y * W + R[k]); }
            else if (sz <= LANE_MAX) ++ct;
        }
        for (int o = 32; o > 0; o >>= 1) {
            ck += __shfl_xor(ck, o); ct += __shfl_xor(ct, o); fk = min(fk, __shfl_xor(fk, o));
        }
        if (lane == 0) {
            blk[((long long)b * 2 + 0) * nblk + y] = ck;
            blk[((long long)b * 2 + 1) * nblk + y] = ct;
            if (fk != 0x7fffffff) atomicMin(&misc[b].first_kept, fk);
        }
    }
}

// roots in raster order: kept -> label, tiny -> list, big-small -> slot and bounding box
__global__ __launch_bounds__(256) void k_run_number(const int *__restrict__ up, const int *__restrict__ rsz,
                                                    const int *__restrict__ runs, const int *__restrict__ rowcnt,
                                                    int H, int W, int min_size, const int *__restrict__ blk, int nblk,
                                                    int *__restrict__ final_, int *__restrict__ tiny_list,
                                                    int *__restrict__ big_list, int *__restrict__ sbox,
                                                    const int *__restrict__ ry1, const int *__restrict__ rx0,
                                                    const int *__restrict__ rx1, const int *__restrict__ parent,
                                                    ConnMisc *__restrict__ misc)
{
    const int b = blockIdx.y;
    if (misc[b].n_over != 0) return;
    const int lane = threadIdx.x & 63;
    const long long npix = (long long)H * W;
    const int *UP = up + b * npix, *SZ = rsz + b * npix;
    const int *P = parent + b * npix;
    const int first_kept = misc[b].first_kept;
    int *F = final_ + b * npix;
    const int *RC = rowcnt + (long long)b * H;
    for (int y = blockIdx.x * 4 + (threadIdx.x >> 6); y < H; y += gridDim.x * 4) {
        const int *R = runs + b * npix + (long long)y * W;
        const int cnt = RC[y];
        int offk = blk[((long long)b * 2 + 0) * nblk + y];
        int offt = blk[((long long)b * 2 + 1) * nblk + y];
        for (int k0 = 0; k0 < cnt; k0 += 64) {
            const int k = k0 + lane;
            int cls = 0, p = 0, rid = 0;     // 0 none, 1 kept, 2 tiny, 3 big-small
            if (k < cnt) {
                rid = y * W + k;
                if (UP[rid] == rid) {
                    p = y * W + R[k];
                    const int sz = SZ[rid];
                    cls = sz >= min_size ? 1 : (sz <= LANE_MAX ? 2 : 3);
                }
            }
            const unsigned long long mk = __ballot(cls == 1), mt = __ballot(cls == 2);
            if (cls == 1) F[p] = offk + (int)spa_rank_in_mask(mk);
            else if (cls == 2) {
                // A tiny component that is ONE run (most noise specks) below the first image row: the search
                // visits p, p+1, ..., p+sz-1 and the last outside neighbour it looks at is the pixel above the
                // last one, which lies in the row above the seed and therefore in a component with a smaller
                // seed: its `adjacent` is known here, without a replay (flag in the list entry).
                const int sz = SZ[rid];
                const int len = (k + 1 < cnt ? R[k + 1] : W) - R[k];
                int entry = p;
                if (sz == len && y > 0) {
                    F[p] = p < first_kept ? -1 : -2 - P[p + sz - 1 - W];
                    entry |= TINY_DONE;
                }
                tiny_list[b * npix + offt + (int)spa_rank_in_mask(mt)] = entry;
            } else if (cls == 3) {
                const int slot = atomicAdd(&misc[b].n_big, 1);
                big_list[b * npix + slot] = p;
                F[p] = slot;
                if (slot < SBOX_CAP) {
                    // the root run is the component's first run in raster order: its row is the box's first row
                    int *bb = sbox + ((long long)b * SBOX_CAP + slot) * 4;
                    bb[0] = y; bb[1] = ry1[b * npix + rid]; bb[2] = rx0[b * npix + rid]; bb[3] = rx1[b * npix + rid];
                }
            }
            offk += __popcll(mk);
            offt += __popcll(mt);
        }
    }
}

// ---------------------------------------------------------------------------------------
// BFS replay, lane tier: one THREAD per tiny component (<= LANE_MAX pixels) runs the sequential
// BFS literally — queue in LDS, "already queued" by searching its own queue — and keeps the
// last outside neighbour that belongs to a component with a smaller seed.
// ---------------------------------------------------------------------------------------
#define LANE_CHUNK 1024
__global__ __launch_bounds__(256) void k_conn_bfs_lane(const int *__restrict__ parent,
                                                       const int *__restrict__ size,
                                                       const int *__restrict__ tiny_list,
                                                       const ConnMisc *__restrict__ misc,
                                                       int *__restrict__ final_, int H, int W)
{
    __shared__ int q[LANE_MAX * 256];
    __shared__ int cl[LANE_CHUNK];
    __shared__ int ccount;
    const int b = blockIdx.y;
    const int npix = H * W;
    const int tid = threadIdx.x;
    const int *P = parent + (long long)b * npix;
    const int *S = size + (long long)b * npix;
    int *F = final_ + (long long)b * npix;
    const int n = misc[b].n_small;
    const int first_kept = misc[b].first_kept;
    for (int base = blockIdx.x * LANE_CHUNK; base < n; base += gridDim.x * LANE_CHUNK) {
      // four in five list entries are single-run components k_run_number has settled: compact the rest first,
      // or every wave would run its replay loop for a dozen live lanes
      if (tid == 0) ccount = 0;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < LANE_CHUNK / 256; ++j) {
          const int i = base + j * 256 + tid;
          if (i >= n) continue;
          const int r = tiny_list[(long long)b * npix + i];
          if (r & TINY_DONE) continue;
          if (r < first_kept) { F[r] = -1; continue; }   // before the first kept component: label 0
          cl[atomicAdd(&ccount, 1)] = r;
      }
      __syncthreads();
      const int nc = ccount;
      for (int i = tid; i < nc; i += 256) {
        const int r = cl[i];
        const int sz = S[r];
        int best = -1;
        int head = 0, tail = 1;
        q[tid] = r;
        while (head < tail) {
            const int u = q[head * 256 + tid];
            const int uy = u / W, ux = u - uy * W;
            // neighbour order of the reference: +x, -x, +y, -y
            const int vx[4] = {ux + 1, ux - 1, ux, ux};
            const int vy[4] = {uy, uy, uy + 1, uy - 1};
            int rv[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const bool inb = vx[d] >= 0 && vx[d] < W && vy[d] >= 0 && vy[d] < H;
                rv[d] = inb ? P[vy[d] * W + vx[d]] : 0x7fffffff;
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if (rv[d] == r) {
                    if (sz > 1) {
                        const int v = vy[d] * W + vx[d];
                        bool seen = false;
                        for (int j = 0; j < tail; ++j) seen = seen || (q[j * 256 + tid] == v);
                        if (!seen) { q[tail * 256 + tid] = v; ++tail; }
                    }
                } else if (rv[d] < r) {
                    best = rv[d];
                }
            }
            ++head;
        }
        F[r] = best < 0 ? -1 : -2 - best;
      }
      __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// BFS replay of one small component, LDS tiers: one wavefront per component.  The component's bounding box
// with a BC_MARGIN-pixel margin on every side (image or not) is staged in LDS as one byte per pixel
//     BC_OTHER    not a candidate for `adjacent` (a component with a larger seed, or outside the image)
//     BC_EARLIER  pixel of a component with a smaller seed (a candidate for `adjacent`)
//     BC_MEMBER   pixel of this component, not yet discovered
//     BC_FOUND    pixel of this component, in the queue or already expanded
//     BC_CUR + l  pixel of this component that lane l expands in the current step
// and the queue is a ring of 16-bit box-local indices (only the current and the next BFS level are alive).
// A step expands 64 queue entries: queue read, mark the own pixel BC_CUR + lane, ONE batch of twelve byte
// reads (the 4 neighbours v and the 8 other pixels w that touch a v — thanks to the margin none needs a
// bounds test and none needs the pixel's row/column), ballots, queue/code writes.  A pixel reached from
// several pixels of the same step goes to the smallest (queue index, direction) key, as in the sequential
// BFS: another claimant w of v is a step-mate iff its code is BC_CUR + l, and it wins iff l < lane — no
// atomics.  The replay is a chain of dependent steps, and one wave issues an instruction every ~5 cycles:
// the step is ~200 instructions, ~1 100 cycles (it was 3 800 with per-neighbour bounds tests and divisions).
// LDS bytes x time is what the tiers compete for, hence bytes, not 16-bit codes, per pixel.
// Components whose box or frontier does not fit go to `todo` (next tier).
// ---------------------------------------------------------------------------------------
#define BC_OTHER 0u
#define BC_EARLIER 1u
#define BC_MEMBER 2u
#define BC_FOUND 3u
#define BC_CUR 64u
#define BC_MAXAREA 65535
#define BC_MARGIN 2

// tier of every big-small component, from its bounding box alone (the tiers then run concurrently
// on separate streams): 16 KB LDS, 80 KB LDS, or global memory
#define BFS_LDS_A (16 * 1024)
#define BFS_RING_A 1024
#define BFS_LDS_B (53 * 512)          // six per CU
#define BFS_RING_B 2048
#define BFS_LDS_C (80 * 1024)
#define BFS_RING_C 4096

__global__ __launch_bounds__(256) void k_conn_classify(const int *__restrict__ big_list,
                                                       const int *__restrict__ size,
                                                       const int *__restrict__ sbox,
                                                       ConnMisc *__restrict__ misc, int *__restrict__ final_,
                                                       int *__restrict__ list_a, int *__restrict__ list_b,
                                                       int *__restrict__ list_c, int *__restrict__ list_g,
                                                       int H, int W)
{
    const int b = blockIdx.y;
    const long long npix = (long long)H * W;
    const int n_big = misc[b].n_big;
    const int first_kept = misc[b].first_kept;
    for (int slot = blockIdx.x * 256 + threadIdx.x; slot < n_big; slot += gridDim.x * 256) {
        const int r = big_list[b * npix + slot];
        if (r < first_kept) { final_[b * npix + r] = -1; continue; }   // before the first kept component: label 0
        int tier = 3;
        if (slot < SBOX_CAP) {
            const int *bb = sbox + ((long long)b * SBOX_CAP + slot) * 4;
            const long long area = (long long)(bb[3] - bb[2] + 1 + 2 * BC_MARGIN) * (bb[1] - bb[0] + 1 + 2 * BC_MARGIN);
            const long long area4 = (area + 3) & ~3ll;
            if (area <= BC_MAXAREA) {
                if (area4 + BFS_RING_A * 2 <= BFS_LDS_A) tier = 0;
                else if (area4 + BFS_RING_B * 2 <= BFS_LDS_B) tier = 1;
                else if (area4 + BFS_RING_C * 2 <= BFS_LDS_C) tier = 2;
            }
        }
        // a replay is a chain of dependent steps whose length grows with the component: the longest ones must
        // start first, or the pass ends with a few workgroups finishing a 5 000-pixel component alone.
        // Lists per size class (an eighth of the list region each), consumed from the largest class down.
        const int sz = size[b * npix + r];
        int cls = 26 - __clz(sz);                      // sz 17..63 -> 0 (and below), 64..127 -> 1, ...
        cls = cls < 0 ? 0 : (cls > 7 ? 7 : cls);
        const long long sub = npix / 8 * cls;
        if (tier == 0) { atomicAdd(&misc[b].n_a, 1); list_a[b * npix + sub + atomicAdd(&misc[b].nb[0][cls], 1)] = slot; }
        else if (tier == 1) { atomicAdd(&misc[b].n_todo1, 1); list_b[b * npix + sub + atomicAdd(&misc[b].nb[1][cls], 1)] = slot; }
        else if (tier == 2) { atomicAdd(&misc[b].n_c, 1); list_c[b * npix + sub + atomicAdd(&misc[b].nb[2][cls], 1)] = slot; }
        else list_g[b * npix + atomicAdd(&misc[b].n_todo2, 1)] = slot;
    }
}

__device__ __forceinline__ void conn_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(64) void k_conn_bfs_lds(const int *__restrict__ parent,
                                                     const int *__restrict__ size,
                                                     const int *__restrict__ big_list,
                                                     const int *__restrict__ sbox,
                                                     const int *__restrict__ list,     // slots, or NULL = all
                                                     int *__restrict__ todo,           // slots that do not fit
                                                     ConnMisc *__restrict__ misc, int tier,
                                                     int *__restrict__ final_, int H, int W,
                                                     int lds_bytes, int ring)          // ring: power of two
{
    extern __shared__ uint32_t lds_u32[];
    const int b = blockIdx.y;
    const int npix = H * W;
    const int lane = threadIdx.x;
    const int *P = parent + (long long)b * npix;
    const int *BL = big_list + (long long)b * npix;
    int *F = final_ + (long long)b * npix;
    const int n_items = tier == 0 ? misc[b].n_a : (tier == 1 ? misc[b].n_todo1 : misc[b].n_c);
    int *todo_count = &misc[b].n_todo2;          // what fits no LDS tier after all goes to the global tier
    const int first_kept = misc[b].first_kept;
    const int rmask = ring - 1;
    // the replays of the larger tiers are the critical path of the pass (few, long, serial): let their
    // instructions issue ahead of the many short waves that share the SIMD
    if (tier > 0) __builtin_amdgcn_s_setprio(3);

    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        // item `it` in largest-class-first order
        int slot = it;
        if (list) {
            int rest = it, cls = 7;
            for (; cls > 0; --cls) {
                const int c = misc[b].nb[tier][cls];
                if (rest < c) break;
                rest -= c;
            }
            slot = list[(long long)b * npix + (long long)(npix / 8) * cls + rest];
        }
        const int r = BL[slot];
        if (r < first_kept) {           // before the first kept component everything is label 0
            if (lane == 0) F[r] = -1;
            continue;
        }
        bool fits = slot < SBOX_CAP;
        int y0 = 0, x0 = 0, bw = 0, bh = 0, area = 0, area4 = 0;
        if (fits) {
            // the staged box: the bounding box with a BC_MARGIN-pixel margin on every side, image or not
            const int *bb = sbox + ((long long)b * SBOX_CAP + slot) * 4;
            y0 = bb[0] - BC_MARGIN; x0 = bb[2] - BC_MARGIN;
            bw = bb[3] - bb[2] + 1 + 2 * BC_MARGIN; bh = bb[1] - bb[0] + 1 + 2 * BC_MARGIN;
            area = bw * bh;
            area4 = (area + 3) & ~3;
            fits = area <= BC_MAXAREA && area4 + ring * 2 <= lds_bytes;
        }
        bool overflow = false;
        int bk = -1, bv = 0;            // the latest (queue index * 4 + direction) that met an earlier component, and where
#ifdef SPA_CONN_TIMING
        const unsigned long long t0_ = wall_clock64(), c0_ = __builtin_readcyclecounter();
        unsigned long long t1_ = t0_, c1_ = c0_, itmax_ = 0;
        int steps_ = 0;
#endif
        if (fits) {
            unsigned char *code = (unsigned char *)lds_u32;
            unsigned short *Q = (unsigned short *)(code + area4);
            conn_wave_sync();
            // stage the box: 16 pixels per lane and pass (16 independent loads in flight per lane), four packed
            // 8-byte LDS stores.  Box coordinates and the global offset advance by additions only.
            const int qy = 1024 / bw, qx = 1024 - qy * bw;
            int px[4], py[4], po[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = g * 256 + lane * 4;
                py[g] = i / bw; px[g] = i - py[g] * bw;
                po[g] = (y0 + py[g]) * W + x0 + px[g];
            }
            const int wrap = W - bw, adv = qy * W + qx;
            for (int i0 = 0; i0 < area4; i0 += 1024) {
#ifdef SPA_CONN_TIMING
                { const unsigned long long n_ = wall_clock64(); if (i0 > 0 && n_ - c1_ > itmax_) itmax_ = n_ - c1_; c1_ = n_; }
#endif
                int rv[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        int xx = px[g] + u, yy = py[g], o = po[g] + u;
                        if (xx >= bw) { xx -= bw; ++yy; o += wrap; }         // bw >= 5 > u
                        const bool in = yy < bh && (unsigned)(y0 + yy) < (unsigned)H && (unsigned)(x0 + xx) < (unsigned)W;
                        const int q = P[(unsigned)(in ? o : r)];            // unconditional: no divergent branch per load
                        rv[g * 4 + u] = in ? q : 0x7fffffff;
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint32_t c[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int q = rv[g * 4 + u];
                        c[u] = q == r ? BC_MEMBER : (q < r ? BC_EARLIER : BC_OTHER);
                    }
                    if (i0 + g * 256 + lane * 4 < area4)
                        lds_u32[((i0 + g * 256) >> 2) + lane] = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
                    px[g] += qx; py[g] += qy; po[g] += adv;
                    if (px[g] >= bw) { px[g] -= bw; ++py[g]; po[g] += wrap; }
                }
            }
            const int ry = r / W, rx = r - ry * W;
            const int rloc = (ry - y0) * bw + (rx - x0);
            conn_wave_sync();
            if (lane == 0) { Q[0] = (unsigned short)rloc; code[rloc] = BC_FOUND; }
            conn_wave_sync();
            int head = 0, tail = 1;
#ifdef SPA_CONN_TIMING
            t1_ = wall_clock64();
            c1_ = __builtin_readcyclecounter();
#endif
            while (head < tail) {
#ifdef SPA_CONN_TIMING
                ++steps_;
#endif
                const int cnt = min(64, tail - head);
                const bool act = lane < cnt;
                const int uidx = head + lane;
                const int u = act ? (int)Q[uidx & rmask] : rloc;
                unsigned char *c = code + u;
                if (act) c[0] = (unsigned char)(BC_CUR + lane);
                conn_wave_sync();
                // directions in the order of the reference's BFS: (+1,0) (-1,0) (0,+1) (0,-1)
                const unsigned cE = c[1], cW = c[-1], cS = c[bw], cN = c[-bw];     // round trip 2
                const unsigned cEE = c[2], cWW = c[-2], cSS = c[2 * bw], cNN = c[-2 * bw];
                const unsigned cSE = c[bw + 1], cSW = c[bw - 1], cNE = c[1 - bw], cNW = c[-1 - bw];
                const unsigned uh = BC_CUR, ul = (unsigned)lane;             // "w is expanded in this step by an earlier lane"
                const bool kEE = cEE - uh < ul, kWW = cWW - uh < ul, kSS = cSS - uh < ul, kNN = cNN - uh < ul;
                const bool kSE = cSE - uh < ul, kSW = cSW - uh < ul, kNE = cNE - uh < ul, kNW = cNW - uh < ul;
                const bool wE = act && cE == BC_MEMBER && !(kEE || kNE || kSE);
                const bool wW = act && cW == BC_MEMBER && !(kWW || kNW || kSW);
                const bool wS = act && cS == BC_MEMBER && !(kSW || kSE || kSS);
                const bool wN = act && cN == BC_MEMBER && !(kNW || kNE || kNN);
                const int k4 = uidx * 4;
                if (act && cE == BC_EARLIER) { bk = k4; bv = u + 1; }
                if (act && cW == BC_EARLIER) { bk = k4 + 1; bv = u - 1; }
                if (act && cS == BC_EARLIER) { bk = k4 + 2; bv = u + bw; }
                if (act && cN == BC_EARLIER) { bk = k4 + 3; bv = u - bw; }
                const unsigned long long mE = __ballot(wE), mW = __ballot(wW), mS = __ballot(wS), mN = __ballot(wN);
                unsigned before = __builtin_amdgcn_mbcnt_hi((unsigned)(mE >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mE, 0u));
                before = __builtin_amdgcn_mbcnt_hi((unsigned)(mW >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mW, before));
                before = __builtin_amdgcn_mbcnt_hi((unsigned)(mS >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mS, before));
                before = __builtin_amdgcn_mbcnt_hi((unsigned)(mN >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mN, before));
                const int total = __popcll(mE) + __popcll(mW) + __popcll(mS) + __popcll(mN);
                if (tail + total - (head + cnt) > ring) { overflow = true; break; }
                int pos = tail + (int)before;
                if (wE) { Q[pos & rmask] = (unsigned short)(u + 1); c[1] = BC_FOUND; ++pos; }
                if (wW) { Q[pos & rmask] = (unsigned short)(u - 1); c[-1] = BC_FOUND; ++pos; }
                if (wS) { Q[pos & rmask] = (unsigned short)(u + bw); c[bw] = BC_FOUND; ++pos; }
                if (wN) { Q[pos & rmask] = (unsigned short)(u - bw); c[-bw] = BC_FOUND; }
                if (act) c[0] = BC_FOUND;
                conn_wave_sync();
                head += cnt;
                tail += total;
            }
        }
#ifdef SPA_CONN_TIMING
        {
            const unsigned long long t2_ = wall_clock64(), c2_ = __builtin_readcyclecounter();
            if (lane == 0 && b == 0 && t2_ - t0_ > 20000)    // > 200 us at 100 MHz
                printf("bfs tier %d item %d: size %d box %dx%d  steps %d  stage %.1f us (slowest 1024-pixel piece %.1f us)  replay %.1f us = %llu cycles  fits %d overflow %d\n", tier, it,
                       size[(long long)b * npix + r], bw, bh, steps_, (t1_ - t0_) * 0.01, itmax_ * 0.01, (t2_ - t1_) * 0.01, c2_ - c1_, (int)fits, (int)overflow);
        }
#endif
        if (!fits || overflow) {
            if (lane == 0) {
                const int k = atomicAdd(todo_count, 1);
                todo[(long long)b * npix + k] = slot;
            }
            continue;
        }
        long long best = bk < 0 ? -1ll : (((long long)bk << 32) | (unsigned)bv);
        for (int o = 32; o > 0; o >>= 1) {
            long long t = __shfl_xor(best, o);
            if (t > best) best = t;
        }
        if (lane == 0) {
            int f = -1;
            if (best >= 0) {
                const int loc = (int)(best & 0xFFFFFFFFll);
                const int gy = y0 + loc / bw, gx = x0 + loc % bw;
                f = -2 - P[gy * W + gx];
            }
            F[r] = f;
        }
    }
}

// one wavefront per small component: replay the BFS in queue order, find `adjacent`
__global__ __launch_bounds__(64) void k_conn_bfs(const int *__restrict__ parent,
                                                 const int *__restrict__ size,
                                                 const int *__restrict__ big_list,
                                                 const int *__restrict__ list,
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
    const int *BL = big_list + (long long)b * npix;
    uint32_t *CL = claim + (long long)b * npix;
    int *F = final_ + (long long)b * npix;
    const int n_small = misc[b].n_todo2;
    const int first_kept = misc[b].first_kept;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int ddx[4] = {1, -1, 0, 0};
    const int ddy[4] = {0, 0, 1, -1};

    for (int i = blockIdx.x; i < n_small; i += gridDim.x) {
        const int r = BL[list[(long long)b * npix + i]];
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
        // the claim words go back to INF: they are all-INF between calls (no per-call memset of the image)
        for (int i2 = lane; i2 < tail; i2 += 64)
            __hip_atomic_store(CL + ld_i32(Q + i2), INF_KEY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // wave max of `best`
        for (int o = 32; o > 0; o >>= 1) {
            long long t = __shfl_xor(best, o);
            if (t > best) best = t;
        }
        if (lane == 0) F[r] = (best < 0) ? -1 : -2 - (int)(best & 0xFFFFFFFFll);
    }
}

// final label of small components: follow the `adjacent` pointers to a kept component
__global__ __launch_bounds__(256) void k_conn_resolve(const int *__restrict__ list, int which,
                                                      const ConnMisc *__restrict__ misc,
                                                      int *__restrict__ final_, int npix)
{
    const int b = blockIdx.y;
    const int n = which == 0 ? misc[b].n_small : misc[b].n_big;
    int *F = final_ + (long long)b * npix;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int r = list[(long long)b * npix + i] & ~TINY_DONE;
        int f = ld_i32(F + r);
        while (f < -1) f = ld_i32(F + (-2 - f));   // pointer to a component with a smaller seed
        st_i32(F + r, f == -1 ? 0 : f);
    }
}

// the final labels of an image WITHOUT an oversize component, straight from the run tables (round 3): a workgroup per
// row looks up, per run, the final label of the run's component (root run -> its first pixel -> final_) and streams
// the row out — the label image is written once and nothing per-pixel is read (k_conn_relabel below reads
// parent[pixel] and gathers final_[parent] per pixel: 12 B per pixel instead of 4 + the row's run table)
__global__ __launch_bounds__(256) void k_run_relabel(const int *__restrict__ up, const int *__restrict__ runs,
                                                     const int *__restrict__ rowcnt, const int *__restrict__ final_,
                                                     const ConnMisc *__restrict__ misc, int32_t *__restrict__ out,
                                                     int H, int W)
{
    extern __shared__ int lds_r[];                 // start x [W] | final label [W]
    const int b = blockIdx.y, y = blockIdx.x;
    if (misc[b].n_over > 0) return;                // its components were cut at pixel level: k_conn_relabel
    const int tid = threadIdx.x;
    const long long npix = (long long)H * W;
    const int *UP = up + b * npix, *RA = runs + b * npix, *F = final_ + b * npix;
    const int *R = RA + (long long)y * W;
    int32_t *O = out + b * npix + (long long)y * W;
    const int cnt = rowcnt[(long long)b * H + y];
    int *sx = lds_r, *sl = lds_r + W;
    for (int k = tid; k < cnt; k += 256) {
        const int r = UP[y * W + k];                            // root run id = ry * W + rk
        const int f = F[(r / W) * W + RA[r]];                    // final_ of the component's first pixel
        sx[k] = R[k];
        sl[k] = f < 0 ? 0 : f;
    }
    __syncthreads();
    const bool vec = (W & 3) == 0;
    for (int x0 = 0; x0 < W; x0 += RUN_CHUNK) {
        const int xb = x0 + tid * 8;
        if (xb >= W) continue;
        int lo = 0, hi = cnt - 1;                               // run of pixel xb: last start <= xb
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (sx[mid] <= xb) lo = mid; else hi = mid - 1;
        }
        int k = lo;
        int nxt = (k + 1 < cnt) ? sx[k + 1] : W;
        int cur = sl[k];
        int o8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int x = xb + i;
            if (x >= nxt && x < W) { ++k; cur = sl[k]; nxt = (k + 1 < cnt) ? sx[k + 1] : W; }
            o8[i] = cur;
        }
        if (vec && xb + 7 < W) {
            *(int4 *)(O + xb) = make_int4(o8[0], o8[1], o8[2], o8[3]);
            *(int4 *)(O + xb + 4) = make_int4(o8[4], o8[5], o8[6], o8[7]);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (xb + i < W) O[xb + i] = o8[i];
        }
    }
}

__global__ __launch_bounds__(256) void k_conn_relabel(const int *__restrict__ parent,
                                                      const int *__restrict__ final_,
                                                      const ConnMisc *__restrict__ misc,
                                                      int32_t *__restrict__ out, int npix)
{
    const int b = blockIdx.y;
    if (misc[b].n_over == 0) return;               // relabelled from its run tables (k_run_relabel)
    const long long o = (long long)b * npix;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        int f = final_[o + parent[o + p]];
        out[o + p] = f < 0 ? 0 : f;
    }
}

__global__ void k_conn_init_misc(ConnMisc *misc, int B, int npix)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        misc[b].n_small = 0; misc[b].first_kept = npix; misc[b].qalloc = 0; misc[b].n_kept = 0;
        misc[b].n_todo1 = 0; misc[b].n_todo2 = 0; misc[b].n_big = 0; misc[b].n_over = 0; misc[b].n_a = 0; misc[b].n_c = 0;
        for (int t = 0; t < 3; ++t)
            for (int c = 0; c < 8; ++c) misc[b].nb[t][c] = 0;
    }
}

extern "C" int spa_enforce_connectivity(spa_ctx *ctx, const int32_t *labels_in, int32_t B,
                                        int32_t H, int32_t W, int32_t min_size, int32_t max_size,
                                        int32_t *labels_out, int32_t *n_labels, void *stream)
{
    SPA_ARG(ctx && labels_in && labels_out && n_labels && B > 0 && H > 0 && W > 0);
    SPA_ARG((long long)H * W < (1ll << 29));
    SPA_ARG((size_t)3 * W * 4 <= 150 * 1024);          // two rows of run tables in LDS (k_run_border, k_run_expand)
    hipStream_t s = spa_stream(stream);
    const int npix = H * W;
    const size_t img = (size_t)B * npix * 4;
    int *parent, *size, *final_, *queue, *blk, *tiny, *big, *sbox, *todo1, *todo2;
    uint32_t *claim;
    ConnMisc *misc;
    int rc;
    const int nblk = (npix + SCAN_PX - 1) / SCAN_PX;
    int *runs, *rowcnt, *blk_run, *rup, *rsz, *ry1, *rx0, *rx1;
    if ((rc = spa_ws_reserve(ctx, WS_PARENT, img, (void **)&parent)) != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_SIZE, img, (void **)&size)) != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_FINAL, img, (void **)&final_)) != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_CLAIM, img, (void **)&claim)) != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_QUEUE, img, (void **)&queue)) != SPA_OK) return rc;
    // compact run tables (front of every row): start x | union-find parent | size | box: last row, first x, last x
    if ((rc = spa_ws_reserve(ctx, WS_RUNS, 6 * img, (void **)&runs)) != SPA_OK) return rc;
    rup = runs + (size_t)B * npix;
    rsz = rup + (size_t)B * npix;
    ry1 = rsz + (size_t)B * npix;
    rx0 = ry1 + (size_t)B * npix;
    rx1 = rx0 + (size_t)B * npix;
    if ((rc = spa_ws_reserve(ctx, WS_SMALL, 2 * img, (void **)&tiny)) != SPA_OK) return rc;
    big = tiny + (size_t)B * npix;
    // per-block counts of the pixel-level scan | per-row counts of the run-level scan | runs per row
    if ((rc = spa_ws_reserve(ctx, WS_BLK, (size_t)B * (2 * nblk + 3 * (size_t)H) * 4, (void **)&blk)) != SPA_OK) return rc;
    blk_run = blk + (size_t)B * 2 * nblk;
    rowcnt = blk_run + (size_t)B * 2 * H;
    if ((rc = spa_ws_reserve(ctx, WS_SBOX, (size_t)B * SBOX_CAP * 16, (void **)&sbox)) != SPA_OK) return rc;
    int *todo0, *todo3;
    if ((rc = spa_ws_reserve(ctx, WS_TODO, 4 * img, (void **)&todo1)) != SPA_OK) return rc;
    todo2 = todo1 + (size_t)B * npix;
    todo0 = todo2 + (size_t)B * npix;
    todo3 = todo0 + (size_t)B * npix;
    if ((rc = spa_aux_streams(ctx)) != SPA_OK) return rc;
    if ((rc = spa_ws_reserve(ctx, WS_CONNMISC, (size_t)B * sizeof(ConnMisc), (void **)&misc)) != SPA_OK) return rc;

    SpaProfScope prof_(ctx, PROF_CONNECT, s);
    hipLaunchKernelGGL(k_conn_init_misc, dim3((B + 63) / 64), dim3(64), 0, s, misc, B, npix);
    // ---- components on runs: rows -> run tables, vertical links, roots + sizes, then one streaming write of
    // parent[pixel] = seed pixel (no per-pixel union-find, no per-pixel atomics)
    int gw = (H + 3) / 4;                       // one wave per row, four rows per workgroup
    if (gw > 1024) gw = 1024;
    // (the labels of the runs are parked in `final_`, which nothing else uses before the numbering)
    hipLaunchKernelGGL(k_run_rows, dim3(H, B), dim3(256), 0, s, labels_in, runs, final_, rup, rsz, rowcnt, H, W);
    if ((size_t)3 * W * 4 > 48 * 1024 && !(ctx->conn_attr_done & 2)) {
        SPA_HIP(hipFuncSetAttribute((const void *)k_run_border, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        SPA_HIP(hipFuncSetAttribute((const void *)k_run_expand, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        SPA_HIP(hipFuncSetAttribute((const void *)k_run_relabel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        ctx->conn_attr_done |= 2;
    }
    const int n_strips = (H + STRIP_ROWS - 1) / STRIP_ROWS;
    hipLaunchKernelGGL(k_run_strip, dim3(n_strips, B), dim3(256), 0, s, (const int *)final_, rup, rsz, ry1, rx0, rx1,
                       (const int *)runs, (const int *)rowcnt, H, W);
    if (n_strips > 1)
        hipLaunchKernelGGL(k_run_border, dim3(n_strips - 1, B), dim3(256), (size_t)3 * W * 4, s, (const int *)final_, rup,
                           (const int *)runs, (const int *)rowcnt, H, W);
    hipLaunchKernelGGL(k_run_flatten, dim3(gw, B), dim3(256), 0, s, rup, rsz, ry1, rx0, rx1, (const int *)rowcnt, H, W);
    hipLaunchKernelGGL(k_run_expand, dim3(H, B), dim3(256), (size_t)2 * W * 4, s, (const int *)rup, (const int *)rsz,
                       (const int *)runs, (const int *)rowcnt, parent, size, H, W);
    {
        // components above max_size are cut the way the reference cuts them (k_conn_split works on
        // parent[pixel] = seed); their images then take the pixel-level numbering kernels below
        if (!ctx->conn_claim_ready || ctx->conn_claim_bytes != ctx->ws_bytes[WS_CLAIM]) {
            // the claim words are all-INF between calls (every kernel that takes some releases them)
            SPA_HIP(hipMemsetAsync(claim, 0xFF, ctx->ws_bytes[WS_CLAIM], s));
            ctx->conn_claim_ready = 1;
            ctx->conn_claim_bytes = ctx->ws_bytes[WS_CLAIM];
        }
        hipLaunchKernelGGL(k_run_find_oversize, dim3(gw, B), dim3(256), 0, s, (const int *)rup, (const int *)rsz,
                           (const int *)runs, (const int *)rowcnt, H, W, max_size, todo1, misc);
        hipLaunchKernelGGL(k_conn_split, dim3(64, B), dim3(64), 0, s, parent, size, (const int *)todo1, misc,
                           claim, queue, H, W, max_size);
        hipLaunchKernelGGL(k_conn_reset_qalloc, dim3((B + 63) / 64), dim3(64), 0, s, misc, B);
    }
    int gb = (npix + 255) / 256;
    if (gb > 1024) gb = 1024;
    // ---- numbering: run level (images without an oversize component) ...
    hipLaunchKernelGGL(k_run_count, dim3(gw, B), dim3(256), 0, s, (const int *)rup, (const int *)rsz,
                       (const int *)runs, (const int *)rowcnt, H, W, min_size, blk_run, H, misc);
    hipLaunchKernelGGL(k_conn_scan, dim3(B), dim3(256), 0, s, blk_run, H, misc, n_labels, 1);
    hipLaunchKernelGGL(k_run_number, dim3(gw, B), dim3(256), 0, s, (const int *)rup, (const int *)rsz,
                       (const int *)runs, (const int *)rowcnt, H, W, min_size, (const int *)blk_run, H, final_, tiny,
                       big, sbox, (const int *)ry1, (const int *)rx0, (const int *)rx1, (const int *)parent, misc);
    // ---- ... and pixel level (early exit unless the image was expanded)
    hipLaunchKernelGGL(k_conn_count, dim3(nblk, B), dim3(256), 0, s, parent, size, npix, min_size,
                       max_size, blk, nblk, misc, ctx->d_status);
    hipLaunchKernelGGL(k_conn_scan, dim3(B), dim3(256), 0, s, blk, nblk, misc, n_labels, 0);
    hipLaunchKernelGGL(k_conn_number, dim3(nblk, B), dim3(256), 0, s, parent, size, npix, W, min_size,
                       blk, nblk, final_, tiny, big, sbox, misc);
    hipLaunchKernelGGL(k_small_bbox, dim3(gb, B), dim3(256), 0, s, parent, size, final_, W, npix,
                       min_size, sbox, misc);
    // BFS replay of the small components.  The tiers are independent of each other (a component only reads
    // the roots of its neighbours), so they run concurrently: the lane tier (<= 16 pixels, one thread each)
    // and the rare 80 KB LDS tier on a side stream, the six-per-CU LDS tier on another, the 16 KB LDS tier
    // here; what fits no LDS tier (or overflows a frontier ring) goes to the global-memory wave tier after the
    // join.  Measured alone each tier lasts about as long as its longest replay (~230 us: 370 steps of
    // ~1 100 cycles); together they are bound by LDS bytes x time and VALU issue (one replaying wave keeps
    // most of a SIMD's issue slots), ~1.1 ms for the bench images, whatever the stream assignment.
    if (!(ctx->conn_attr_done & 1)) {        // per context = per device
        SPA_HIP(hipFuncSetAttribute((const void *)k_conn_bfs_lds, hipFuncAttributeMaxDynamicSharedMemorySize, BFS_LDS_C));
        ctx->conn_attr_done |= 1;
    }
    hipLaunchKernelGGL(k_conn_classify, dim3(8, B), dim3(256), 0, s, big, (const int *)size, sbox, misc, final_, todo0, todo1, todo3, todo2, H, W);
    SPA_HIP(hipEventRecord(ctx->ev_fork, s));
    SPA_HIP(hipStreamWaitEvent(ctx->aux[0], ctx->ev_fork, 0));
    SPA_HIP(hipStreamWaitEvent(ctx->aux[1], ctx->ev_fork, 0));
    hipLaunchKernelGGL(k_conn_bfs_lds, dim3(256, B), dim3(64), BFS_LDS_B, ctx->aux[1], parent, size, big, sbox,
                       (const int *)todo1, todo2, misc, 1, final_, H, W, BFS_LDS_B, BFS_RING_B);
    hipLaunchKernelGGL(k_conn_bfs_lane, dim3(gb, B), dim3(256), 0, ctx->aux[0], parent, size, tiny, misc,
                       final_, H, W);
    hipLaunchKernelGGL(k_conn_bfs_lds, dim3(64, B), dim3(64), BFS_LDS_C, ctx->aux[0], parent, size, big, sbox,
                       (const int *)todo3, todo2, misc, 2, final_, H, W, BFS_LDS_C, BFS_RING_C);
    hipLaunchKernelGGL(k_conn_bfs_lds, dim3(1024, B), dim3(64), BFS_LDS_A, s, parent, size, big, sbox,
                       (const int *)todo0, todo2, misc, 0, final_, H, W, BFS_LDS_A, BFS_RING_A);
    SPA_HIP(hipEventRecord(ctx->ev_join[0], ctx->aux[0]));
    SPA_HIP(hipEventRecord(ctx->ev_join[1], ctx->aux[1]));
    SPA_HIP(hipStreamWaitEvent(s, ctx->ev_join[0], 0));
    SPA_HIP(hipStreamWaitEvent(s, ctx->ev_join[1], 0));
    // (the claim array is all-INF again: k_conn_split releases what it takes)
    hipLaunchKernelGGL(k_conn_bfs, dim3(256, B), dim3(64), 0, s, parent, size, big, (const int *)todo2,
                       misc, claim, queue, final_, H, W);
    hipLaunchKernelGGL(k_conn_resolve, dim3(gb, B), dim3(256), 0, s, tiny, 0, misc, final_, npix);
    hipLaunchKernelGGL(k_conn_resolve, dim3(64, B), dim3(256), 0, s, big, 1, misc, final_, npix);
    int gr = (npix + 255) / 256;
    if (gr > 2048) gr = 2048;
    hipLaunchKernelGGL(k_run_relabel, dim3(H, B), dim3(256), (size_t)2 * W * 4, s, (const int *)rup, (const int *)runs,
                       (const int *)rowcnt, (const int *)final_, (const ConnMisc *)misc, labels_out, H, W);
    hipLaunchKernelGGL(k_conn_relabel, dim3(gr, B), dim3(256), 0, s, parent, final_, (const ConnMisc *)misc, labels_out, npix);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}
