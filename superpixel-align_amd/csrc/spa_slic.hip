// SLIC on gfx950: rgb -> Lab, the Lloyd sweeps of skimage's _slic_cython, bit exact.
//
// What "bit exact" forces on the design (see DESIGN.md section "SLIC"):
//  * skimage instantiates the Cython core in float32 for a float32 image, and rounds after
//    every operation (no FMA in its x86-64 wheels).  This file is compiled with
//    -ffp-contract=off and spells every float32 operation in skimage's order.
//  * the assignment loop of skimage is centre-major ("for each centre, for each pixel of its
//    2S window, if dist < best") so among equal distances the LOWEST centre index that is
//    strictly better wins.  slic_assign is pixel-major and walks the candidate centres of a
//    tile in increasing index order with the same strict comparison.
//  * the centroid update of skimage is a float32 running sum over the pixels of a segment in
//    raster order.  float32 addition is not associative and the sums are large (10^4 terms
//    of magnitude 10^2..10^3), so any tree reduction changes the centroids by ~1e-5 relative
//    and the labels of hundreds of boundary pixels with them.  slic_update therefore keeps
//    the serial order: one wavefront per segment gathers the segment's pixels in raster order
//    (coalesced label reads + ballot/mbcnt compaction into an LDS ring) and five lanes carry
//    the five running sums (y, x, L, a, b) through the ring.  The chains of different segments
//    are independent, so the GPU runs thousands of them concurrently (B * n_centroids waves).
#include "spa_common.h"

// ------------------------------------------------------------------------------------
// rgb -> scaled Lab (skimage/color/colorconv.py:657-661, :950-969), float32 steps in the
// reference's order; x^2.4 and cbrt through the deterministic binary64 routines.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float lab_f(float s)
{
    if (s > (float)0.008856) return (float)spa_det_exp(spa_det_log_pos((double)s) / 3.0);
    return (float)7.787 * s + (float)(16.0 / 116.0);
}

__device__ __forceinline__ void rgb2lab_px(float r, float g, float b, float ratio, float &L,
                                           float &A, float &Bc)
{
    float v[3] = {r, g, b};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a = v[c];
        if (a > (float)0.04045) {
            float u = (a + (float)0.055) / (float)1.055;
            v[c] = (float)spa_det_exp(2.4 * spa_det_log_pos((double)u));
        } else {
            v[c] = a / (float)12.92;
        }
    }
    float X = (float)0.412453 * v[0];
    X = X + (float)0.357580 * v[1];
    X = X + (float)0.180423 * v[2];
    float Y = (float)0.212671 * v[0];
    Y = Y + (float)0.715160 * v[1];
    Y = Y + (float)0.072169 * v[2];
    float Z = (float)0.019334 * v[0];
    Z = Z + (float)0.119193 * v[1];
    Z = Z + (float)0.950227 * v[2];
    float fx = lab_f(X / (float)0.95047);
    float fy = lab_f(Y / (float)1.0);
    float fz = lab_f(Z / (float)1.08883);
    L = ((float)116.0 * fy - (float)16.0) * ratio;
    A = ((float)500.0 * (fx - fy)) * ratio;
    Bc = ((float)200.0 * (fy - fz)) * ratio;
}

__global__ __launch_bounds__(256) void k_rgb2lab(const float *__restrict__ rgb,
                                                 float *__restrict__ lab, long long npix,
                                                 float ratio, int vec4)
{
    const int b = blockIdx.y;
    const float *src = rgb + (long long)b * 3 * npix;
    float *dst = lab + (long long)b * 3 * npix;
    long long stride = (long long)gridDim.x * blockDim.x;
    if (vec4) {
        long long n4 = npix >> 2;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            float4 r = ((const float4 *)src)[i];
            float4 g = ((const float4 *)(src + npix))[i];
            float4 bl = ((const float4 *)(src + 2 * npix))[i];
            float4 L, A, Bc;
            rgb2lab_px(r.x, g.x, bl.x, ratio, L.x, A.x, Bc.x);
            rgb2lab_px(r.y, g.y, bl.y, ratio, L.y, A.y, Bc.y);
            rgb2lab_px(r.z, g.z, bl.z, ratio, L.z, A.z, Bc.z);
            rgb2lab_px(r.w, g.w, bl.w, ratio, L.w, A.w, Bc.w);
            ((float4 *)dst)[i] = L;
            ((float4 *)(dst + npix))[i] = A;
            ((float4 *)(dst + 2 * npix))[i] = Bc;
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
            float L, A, Bc;
            rgb2lab_px(src[i], src[npix + i], src[2 * npix + i], ratio, L, A, Bc);
            dst[i] = L;
            dst[npix + i] = A;
            dst[2 * npix + i] = Bc;
        }
    }
}

extern "C" int spa_rgb2lab(spa_ctx *ctx, const float *rgb, int32_t B, int32_t H, int32_t W,
                           float ratio, float *lab, void *stream)
{
    SPA_ARG(ctx && rgb && lab && B > 0 && H > 0 && W > 0);
    long long npix = (long long)H * W;
    int vec4 = (npix % 4 == 0) && (((uintptr_t)rgb | (uintptr_t)lab) % 16 == 0);
    long long work = vec4 ? npix / 4 : npix;
    int gx = (int)((work + 255) / 256);
    if (gx > 2048) gx = 2048;
    { SpaProfScope prof_(ctx, PROF_RGB2LAB, spa_stream(stream));
    hipLaunchKernelGGL(k_rgb2lab, dim3(gx, B), dim3(256), 0, spa_stream(stream), rgb, lab, npix,
                       ratio, vec4); }
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// ------------------------------------------------------------------------------------
// centre table: 12 words per (image, centre)
//   [0] cy [1] cx [2] cL [3] ca [4] cb [5] -  [6] y0 [7] y1 [8] x0 [9] x1 [10] count [11] -
//   [12] by0 [13] by1 [14] bx0 [15] bx1 : bounding box (inclusive) of the pixels the last
//   assignment sweep gave to this centre, accumulated by k_slic_assign with atomics and
//   consumed + reset by k_slic_update
// [y0,y1) x [x0,x1) is skimage's search window of the centre:
//   y_min = <Py_ssize_t>max(cy - 2*step_y, 0); y_max = <Py_ssize_t>min(cy + 2*step_y + 1, H)
// ------------------------------------------------------------------------------------
#define CEN_WORDS 16

__device__ __forceinline__ void slic_window(float cy, float cx, int s2y, int s2x, int H, int W,
                                            int &y0, int &y1, int &x0, int &x1)
{
    float fy0 = cy - (float)s2y;
    if (!(fy0 > 0.0f)) fy0 = 0.0f;
    float fy1 = (cy + (float)s2y) + 1.0f;
    if (!(fy1 < (float)H)) fy1 = (float)H;
    float fx0 = cx - (float)s2x;
    if (!(fx0 > 0.0f)) fx0 = 0.0f;
    float fx1 = (cx + (float)s2x) + 1.0f;
    if (!(fx1 < (float)W)) fx1 = (float)W;
    y0 = (int)fy0; y1 = (int)fy1; x0 = (int)fx0; x1 = (int)fx1;
}

__global__ void k_slic_init(uint32_t *__restrict__ cen, int nC, int grid_nx, int start_y,
                            int start_x, int step_y, int step_x, int s2y, int s2x, int H, int W)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    int b = blockIdx.y;
    if (k >= nC) return;
    int iy = k / grid_nx, ix = k % grid_nx;
    float cy = (float)(start_y + iy * step_y), cx = (float)(start_x + ix * step_x);
    int y0, y1, x0, x1;
    slic_window(cy, cx, s2y, s2x, H, W, y0, y1, x0, x1);
    uint32_t *c = cen + ((long long)b * nC + k) * CEN_WORDS;
    c[0] = __float_as_uint(cy); c[1] = __float_as_uint(cx);
    c[2] = 0u; c[3] = 0u; c[4] = 0u; c[5] = 0u;
    c[6] = (uint32_t)y0; c[7] = (uint32_t)y1; c[8] = (uint32_t)x0; c[9] = (uint32_t)x1;
    c[10] = 0u; c[11] = 0u;
    c[12] = 0x7fffffffu; c[13] = 0u; c[14] = 0x7fffffffu; c[15] = 0u;
}

// ------------------------------------------------------------------------------------
// assignment sweep: one 32x32 pixel tile per 256-thread workgroup, 4 pixels per thread.
// Candidate centres (windows intersecting the tile) are compacted in increasing index
// order into LDS, then every thread walks the list; LDS reads are wave-uniform broadcasts.
// Algorithmic HBM bytes per pixel per sweep: 12 (Lab) + 4 (label).
// ------------------------------------------------------------------------------------
#define TILE 32

__global__ __launch_bounds__(256) void k_slic_assign(const float *__restrict__ lab,
                                                     uint32_t *__restrict__ cen, int nC,
                                                     int H, int W, float sw,
                                                     int32_t *__restrict__ labels,
                                                     unsigned long long *__restrict__ rowmask, int HG, int PW,
                                                     uint32_t *__restrict__ status)
{
    __shared__ uint4 cand[256 * 3];
    __shared__ int wave_cnt[4];
    const int b = blockIdx.z;
    const int ty0 = blockIdx.y * TILE, tx0 = blockIdx.x * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long npix = (long long)H * W;
    const float *pl = lab + (long long)b * 3 * npix;
    uint32_t *cb = cen + (long long)b * nC * CEN_WORDS;

    const int y = ty0 + (tid >> 3);
    const int xb = tx0 + (tid & 7) * 4;
    const bool row_ok = y < H;
    float pL[4], pA[4], pB[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ok[i] = row_ok && (xb + i) < W;
    const long long base = (long long)y * W + xb;
    if (ok[3] && ((base & 3) == 0)) {
        float4 l4 = *(const float4 *)(pl + base);
        float4 a4 = *(const float4 *)(pl + npix + base);
        float4 b4 = *(const float4 *)(pl + 2 * npix + base);
        pL[0] = l4.x; pL[1] = l4.y; pL[2] = l4.z; pL[3] = l4.w;
        pA[0] = a4.x; pA[1] = a4.y; pA[2] = a4.z; pA[3] = a4.w;
        pB[0] = b4.x; pB[1] = b4.y; pB[2] = b4.z; pB[3] = b4.w;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pL[i] = ok[i] ? pl[base + i] : 0.0f;
            pA[i] = ok[i] ? pl[npix + base + i] : 0.0f;
            pB[i] = ok[i] ? pl[2 * npix + base + i] : 0.0f;
        }
    }
    float best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int bl[4] = {-1, -1, -1, -1};
    const float fy = (float)y;

    for (int kb = 0; kb < nC; kb += 256) {
        const int k = kb + tid;
        bool hit = false;
        uint4 w0, w1, w2;
        if (k < nC) {
            const uint4 *c = (const uint4 *)(cb + (long long)k * CEN_WORDS);
            w0 = c[0]; w1 = c[1]; w2 = c[2];
            int y0 = (int)w1.z, y1 = (int)w1.w, x0 = (int)w2.x, x1 = (int)w2.y;
            hit = (y0 < ty0 + TILE) && (y1 > ty0) && (x0 < tx0 + TILE) && (x1 > tx0);
        }
        unsigned long long m = __ballot(hit);
        if (lane == 0) wave_cnt[wv] = __popcll(m);
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = wave_cnt[i];
            if (i < wv) off += c;
            total += c;
        }
        if (hit) {
            int pos = off + (int)spa_rank_in_mask(m);
            // entry: (cy, cx, cL, ca) (cb, k, y0, y1) (x0, x1, -, -)
            cand[pos * 3 + 0] = w0;
            cand[pos * 3 + 1] = make_uint4(w1.x, (uint32_t)k, w1.z, w1.w);
            cand[pos * 3 + 2] = make_uint4(w2.x, w2.y, 0u, 0u);
        }
        __syncthreads();
        if (row_ok) {
            // straight-line body: the three 16-byte LDS reads of an entry are issued together and
            // the window test is folded into the final comparison (no divergent branches), so the
            // compiler can overlap the next entry's reads with this entry's arithmetic
#pragma unroll 2
            for (int j = 0; j < total; ++j) {
                const uint4 e0 = cand[j * 3 + 0], e1 = cand[j * 3 + 1], e2 = cand[j * 3 + 2];
                const int y0 = (int)e1.z, y1 = (int)e1.w, x0 = (int)e2.x, x1 = (int)e2.y;
                const float cy = __uint_as_float(e0.x), cx = __uint_as_float(e0.y);
                const float cl = __uint_as_float(e0.z), ca = __uint_as_float(e0.w);
                const float cbb = __uint_as_float(e1.x);
                const int kk = (int)e1.y;
                const bool rowin = (y >= y0) && (y < y1);
                const float ty = cy - fy;
                const float dy = ty * ty;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int x = xb + i;
                    const float tx = cx - (float)x;
                    const float dx = tx * tx;
                    float dc = (dy + dx) * sw;
                    const float t0 = pL[i] - cl, t1 = pA[i] - ca, t2 = pB[i] - cbb;
                    float col = t0 * t0;
                    col = col + t1 * t1;
                    col = col + t2 * t2;
                    dc = dc + col;
                    const bool take = rowin && ok[i] && (x >= x0) && (x < x1) && (best[i] > dc);
                    best[i] = take ? dc : best[i];
                    bl[i] = take ? kk : bl[i];
                }
            }
        }
        __syncthreads();
    }
    bool uncovered = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) uncovered = uncovered || (ok[i] && bl[i] < 0);
    if (uncovered) atomicOr(status, SPA_ST_SLIC_UNCOVERED);
    // occupancy masks of the new segments for the centroid update: the distinct labels of a wave
    // (8 rows x 32 columns, lane = row*8 + column group) are enumerated with ballots and each
    // sets one bit — scattered atomics run at a fixed chip-wide rate, so they are kept to one per
    // (wave, label)
    {
        // runs of equal label inside the thread's 4 pixels (static indexing only: a runtime
        // index into bl[]/ok[] would push the arrays to scratch memory)
        bool val[4], st[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) val[i] = ok[i] && bl[i] >= 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) st[i] = val[i] && (i == 0 || !val[i - 1] || bl[i] != bl[i - 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = st[i] ? bl[i] : -1;
            unsigned long long todo = __ballot(rl >= 0);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int ll = __shfl(rl, leader);
                const bool mine = (rl == ll);
                const unsigned long long same = __ballot(mine);
                // occupancy: one bit per (row, 64-pixel piece) of the centre.  A wave covers 8 rows of
                // one piece, i.e. one byte of the mask word of its (8-row group, piece octet), so one
                // atomicOr per (wave, label) records all of its rows.
                const unsigned long long rb =
                    __ballot(lane < 8 && ((same >> (lane * 8)) & 0xFFull) != 0ull) & 0xFFull;
                if (lane == leader) {
                    const int piece = tx0 >> 6;
                    atomicOr(rowmask + (((long long)b * nC + ll) * HG + ((ty0 >> 3) + wv)) * PW + (piece >> 3),
                             rb << ((piece & 7) * 8));
                }
                todo &= ~same;
            }
        }
    }
    int32_t *out = labels + (long long)b * npix;
    if (ok[3] && ((base & 3) == 0)) {
        *(int4 *)(out + base) = make_int4(bl[0], bl[1], bl[2], bl[3]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ok[i]) out[base + i] = bl[i];
    }
}

// ------------------------------------------------------------------------------------
// centroid update: one 256-thread workgroup per (image, centre).  Raster-order float32 sums.
//
// skimage adds the pixels of a segment to float32 accumulators in raster order; float32
// addition does not commute with regrouping, so the order is kept: lanes 0..4 of wave 0 carry
// the five running sums (y, x, L, a, b) through the segment's pixels one by one.  Everything
// around that serial chain is parallel and overlapped with it:
//   * the assignment sweep left one occupancy bit per (8-row group, 64-pixel piece) of the
//     segment; wave 1 expands the bits into a raster-ordered piece list in LDS;
//   * waves 1..3 (gatherers) take 4 pieces each per round: the labels and Lab values of their
//     pieces are requested together (16 independent loads in flight per lane), the matching
//     pixels are compacted with ballot + mbcnt, in raster order, into that wave's sub-ring;
//   * wave 0 (consumer) drains the three sub-rings of the PREVIOUS round in order while the
//     gatherers fill the other buffer (double buffering, one barrier per round).
// Thousands of these chains (B * n_centroids workgroups) run concurrently.
// ------------------------------------------------------------------------------------
#define UPD_NG 3           // gather waves
#define UPD_PPW 4          // pieces per gather wave per round
#define UPD_SUB (UPD_PPW * 64)
#define UPD_STRIDE (UPD_SUB + 16)   // floats per feature row of a sub-ring (+16: zero pad, read slack)
#define UPD_PLIST 4096     // piece-list capacity (entries of 16 bits: row in window << 5 | piece)

struct UpdRegs { int lv[UPD_PPW]; float vL[UPD_PPW], vA[UPD_PPW], vB[UPD_PPW]; unsigned pe[UPD_PPW]; };

__global__ __launch_bounds__(256) void k_slic_update(const float *__restrict__ lab,
                                                     const int32_t *__restrict__ labels,
                                                     uint32_t *__restrict__ cen, int nC, int H,
                                                     int W, int s2y, int s2x,
                                                     unsigned long long *__restrict__ rowmask, int HG, int PW,
                                                     uint32_t *__restrict__ status)
{
    __shared__ __attribute__((aligned(16))) float ring[2][UPD_NG][UPD_STRIDE * 5];   // [feature][entry]
    __shared__ int ring_cnt[2][UPD_NG];
    __shared__ unsigned short plist[UPD_PLIST];
    __shared__ int s_npieces, s_gtake;
    const int k = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long npix = (long long)H * W;
    const float *pl = lab + (long long)b * 3 * npix;
    const int32_t *lb = labels + (long long)b * npix;
    uint32_t *c = cen + ((long long)b * nC + k) * CEN_WORDS;
    const int wy0 = (int)c[6], wy1 = (int)c[7];   // the search window [wy0, wy1) bounds the segment
    unsigned long long *rm = rowmask + ((long long)b * nC + k) * HG * PW;

    float acc = 0.0f;          // wave 0, lanes 0..4: running sums of y, x, L, a, b
    unsigned n = 0;            // wave 0: pixel count

    const int g0 = wy0 >> 3, g1 = (wy1 - 1) >> 3;
    const int ybase = g0 << 3;                    // plist rows are relative to this
    int gb = g0;
    while (gb <= g1) {
        if (wv == 1) {
            // piece list in raster order: groups in order, rows in order, pieces in order.
            // lane g expands group gb+g: PW mask words, byte q of word i = rows of piece 8i+q.
            // As many whole groups as fit the list are taken (and their masks cleared).
            const int gg = gb + lane;
            const bool has = gg <= g1;
            unsigned long long mw[4] = {0ull, 0ull, 0ull, 0ull};
            int cntl = 0;
            unsigned long long *w = rm + (long long)gg * PW;
            if (has) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < PW) { mw[i] = w[i]; cntl += __popcll(mw[i]); }
            }
            int inc = cntl;
            for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
            const bool take = has && inc <= UPD_PLIST;
            const unsigned long long tm = __ballot(take);
            const int gtake = __popcll(tm);               // prefix property: lanes 0..gtake-1
            const int total = gtake ? __shfl(inc, gtake - 1) : 0;
            if (lane == 0) { s_npieces = total; s_gtake = gtake > 0 ? gtake : 1; }
            if (take) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < PW) w[i] = 0ull;                                      // consumed
                int pos = inc - cntl;
                if (cntl) {
                    const int yrel = (gg << 3) - ybase;
                    for (int r = 0; r < 8; ++r) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            unsigned long long bits = (mw[i] >> r) & 0x0101010101010101ull;
                            while (bits) {
                                const int q = (__ffsll((long long)bits) - 1) >> 3;
                                plist[pos++] = (unsigned short)(((yrel + r) << 5) | (i * 8 + q));
                                bits &= bits - 1ull;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        const int npieces = s_npieces;
        gb += s_gtake;
        const int rounds = (npieces + UPD_NG * UPD_PPW - 1) / (UPD_NG * UPD_PPW);
        const int gw = wv - 1;
        // three rounds in flight per gather wave: labels of round rd+2 are requested, the Lab values
        // of round rd+1 are requested for the lanes whose label matched (so pixels of neighbouring
        // segments inside a piece cost 4 bytes, not 16), round rd is compacted
        UpdRegs cur, nxt, nx2;
        auto fetch_labels = [&](UpdRegs &R, int rd) {
            const int t0 = (rd * UPD_NG + gw) * UPD_PPW;
#pragma unroll
            for (int u = 0; u < UPD_PPW; ++u) {
                const int t = t0 + u;
                R.pe[u] = t < npieces ? (unsigned)plist[t] : 0u;
                const int yy = ybase + (int)(R.pe[u] >> 5), xx = (int)((R.pe[u] & 31u) << 6) + lane;
                const bool in = (t < npieces) && (xx < W) && (yy < H);
                R.lv[u] = in ? lb[(long long)yy * W + xx] : -1;
            }
        };
        auto fetch_lab = [&](UpdRegs &R) {
#pragma unroll
            for (int u = 0; u < UPD_PPW; ++u) {
                const int yy = ybase + (int)(R.pe[u] >> 5), xx = (int)((R.pe[u] & 31u) << 6) + lane;
                const long long p = (long long)yy * W + xx;
                const bool m = (R.lv[u] == k);
                R.vL[u] = m ? pl[p] : 0.0f;
                R.vA[u] = m ? pl[npix + p] : 0.0f;
                R.vB[u] = m ? pl[2 * npix + p] : 0.0f;
            }
        };
        if (wv >= 1 && rounds > 0) {
            fetch_labels(cur, 0);
            if (rounds > 1) fetch_labels(nxt, 1);
            fetch_lab(cur);
        }
        for (int rd = 0; rd <= rounds; ++rd) {
            if (wv >= 1 && rd < rounds) {
                if (rd + 2 < rounds) fetch_labels(nx2, rd + 2);
                if (rd + 1 < rounds) fetch_lab(nxt);
                float *sub = ring[rd & 1][gw];
                int fill = 0;
#pragma unroll
                for (int u = 0; u < UPD_PPW; ++u) {
                    const bool match = (cur.lv[u] == k);
                    const unsigned long long mm = __ballot(match);
                    if (match) {
                        const int yy = ybase + (int)(cur.pe[u] >> 5), xx = (int)((cur.pe[u] & 31u) << 6) + lane;
                        const int ps = fill + (int)spa_rank_in_mask(mm);
                        sub[0 * UPD_STRIDE + ps] = (float)yy;
                        sub[1 * UPD_STRIDE + ps] = (float)xx;
                        sub[2 * UPD_STRIDE + ps] = cur.vL[u];
                        sub[3 * UPD_STRIDE + ps] = cur.vA[u];
                        sub[4 * UPD_STRIDE + ps] = cur.vB[u];
                    }
                    fill += __popcll(mm);
                }
                // pad to a multiple of 8 with +0.0f (x + 0.0f == x): the consumer needs no tail loop
                if (lane < 8) {
#pragma unroll
                    for (int f = 0; f < 5; ++f) sub[f * UPD_STRIDE + fill + lane] = 0.0f;
                }
                if (lane == 0) ring_cnt[rd & 1][gw] = fill;
                cur = nxt;
                nxt = nx2;
            } else if (wv == 0 && rd > 0) {
                // ---- chain: sub-rings of round rd-1, in order
#pragma unroll
                for (int g = 0; g < UPD_NG; ++g) {
                    const float *sub = ring[(rd - 1) & 1][g];
                    const int cnt = ring_cnt[(rd - 1) & 1][g];
                    if (lane < 5) {
                        // lane f walks feature row f: two 16-byte LDS reads per 8 pixels, issued one
                        // step ahead of the dependent float32 adds
                        const float *row = sub + lane * UPD_STRIDE;
                        const float4 *r4 = (const float4 *)row;
                        float4 a = r4[0], bq = r4[1];
                        int j = 0;
                        for (; j < cnt; j += 8) {
                            const float4 na = r4[(j >> 2) + 2], nb = r4[(j >> 2) + 3];
                            acc = acc + a.x; acc = acc + a.y; acc = acc + a.z; acc = acc + a.w;
                            acc = acc + bq.x; acc = acc + bq.y; acc = acc + bq.z; acc = acc + bq.w;
                            a = na; bq = nb;
                        }
                    }
                    n += (unsigned)cnt;
                }
            }
            __syncthreads();
        }
    }
    if (wv != 0) return;
    if (n == 0u) {
        if (lane == 0) atomicOr(status, SPA_ST_SLIC_EMPTY_SEGMENT);
        return;
    }
    float mean = acc / (float)n;      // segments[k, c] /= n_segment_elems[k]
    float cy = __shfl(mean, 0), cx = __shfl(mean, 1);
    float cl = __shfl(mean, 2), ca = __shfl(mean, 3), cbb = __shfl(mean, 4);
    if (lane == 0) {
        int ny0, ny1, nx0, nx1;
        slic_window(cy, cx, s2y, s2x, H, W, ny0, ny1, nx0, nx1);
        c[0] = __float_as_uint(cy); c[1] = __float_as_uint(cx);
        c[2] = __float_as_uint(cl); c[3] = __float_as_uint(ca); c[4] = __float_as_uint(cbb);
        c[6] = (uint32_t)ny0; c[7] = (uint32_t)ny1; c[8] = (uint32_t)nx0; c[9] = (uint32_t)nx1;
        c[10] = n;
    }
}

__global__ void k_slic_export_centres(const uint32_t *__restrict__ cen, float *__restrict__ out,
                                      long long total)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t *c = cen + i * CEN_WORDS;
    float *o = out + i * 6;
    o[0] = 0.0f;
    o[1] = __uint_as_float(c[0]); o[2] = __uint_as_float(c[1]);
    o[3] = __uint_as_float(c[2]); o[4] = __uint_as_float(c[3]); o[5] = __uint_as_float(c[4]);
}

extern "C" int spa_slic_core(spa_ctx *ctx, const float *lab, int32_t B, int32_t H, int32_t W,
                             int32_t n_segments, int32_t max_iter, int32_t *labels,
                             float *centres, void *stream)
{
    SPA_ARG(ctx && lab && labels && B > 0 && max_iter > 0);
    spa_slic_plan pl;
    int rc = spa_slic_make_plan(H, W, n_segments, &pl);
    if (rc != SPA_OK) return rc;
    const int nC = pl.n_centroids;
    hipStream_t s = spa_stream(stream);
    uint32_t *cen;
    rc = spa_ws_reserve(ctx, WS_CENTRES, (size_t)B * nC * CEN_WORDS * 4, (void **)&cen);
    if (rc != SPA_OK) return rc;
    const int s2y = 2 * pl.win_step_y, s2x = 2 * pl.win_step_x;
    SPA_ARG(W <= 2048);      // occupancy masks: <= 32 pieces of 64 pixels per row (piece-list encoding)
    SPA_ARG(4 * pl.win_step_y + 24 < 2048);   // piece-list rows are 11-bit offsets into the search window
    const int HG = (H + 7) / 8;
    const int PW = (((W + 63) / 64) + 7) / 8;      // mask words per (centre, 8-row group)
    unsigned long long *rowmask;
    rc = spa_ws_reserve(ctx, WS_ROWMASK, (size_t)B * nC * HG * PW * 8, (void **)&rowmask);
    if (rc != SPA_OK) return rc;
    SPA_HIP(hipMemsetAsync(rowmask, 0, (size_t)B * nC * HG * PW * 8, s));
    hipLaunchKernelGGL(k_slic_init, dim3((nC + 127) / 128, B), dim3(128), 0, s, cen, nC,
                       pl.grid_nx, pl.start_y, pl.start_x, pl.step_y, pl.step_x, s2y, s2x, H, W);
    SPA_LAUNCH_CHECK();
    // cdef floating spatial_weight = 1.0 / (step * step)
    const float sw = (float)(1.0 / (double)(pl.step * pl.step));
    dim3 ga((W + TILE - 1) / TILE, (H + TILE - 1) / TILE, B);
    for (int it = 0; it < max_iter; ++it) {
        { SpaProfScope prof_(ctx, PROF_SLIC_ASSIGN, s);
        hipLaunchKernelGGL(k_slic_assign, ga, dim3(256), 0, s, lab, cen, nC, H, W, sw, labels,
                           rowmask, HG, PW, ctx->d_status); }
        SPA_LAUNCH_CHECK();
        // the centroids computed after the last sweep never influence the labels
        if (it + 1 < max_iter || centres) {
            SpaProfScope prof_(ctx, PROF_SLIC_UPDATE, s);
            hipLaunchKernelGGL(k_slic_update, dim3(nC, B), dim3(256), 0, s, lab, labels, cen, nC,
                               H, W, s2y, s2x, rowmask, HG, PW, ctx->d_status);
            SPA_LAUNCH_CHECK();
        }
    }
    if (centres) {
        long long total = (long long)B * nC;
        hipLaunchKernelGGL(k_slic_export_centres, dim3((unsigned)((total + 255) / 256)), dim3(256),
                           0, s, cen, centres, total);
        SPA_LAUNCH_CHECK();
    }
    return SPA_OK;
}

extern "C" int spa_slic(spa_ctx *ctx, const float *rgb, int32_t B, int32_t H, int32_t W,
                        int32_t n_segments, float compactness, int32_t max_iter,
                        int32_t *labels, int32_t *n_labels, void *stream)
{
    SPA_ARG(ctx && rgb && labels && n_labels && compactness > 0.0f);
    spa_slic_plan pl;
    int rc = spa_slic_make_plan(H, W, n_segments, &pl);
    if (rc != SPA_OK) return rc;
    float *lab;
    int32_t *pre;
    size_t npix = (size_t)H * W;
    rc = spa_ws_reserve(ctx, WS_LAB, (size_t)B * 3 * npix * 4, (void **)&lab);
    if (rc != SPA_OK) return rc;
    rc = spa_ws_reserve(ctx, WS_PRE, (size_t)B * npix * 4, (void **)&pre);
    if (rc != SPA_OK) return rc;
    // ratio = 1.0 / compactness; image * ratio in float32 (slic_superpixels.py:303-305)
    float ratio = (float)(1.0 / (double)compactness);
    rc = spa_rgb2lab(ctx, rgb, B, H, W, ratio, lab, stream);
    if (rc != SPA_OK) return rc;
    rc = spa_slic_core(ctx, lab, B, H, W, n_segments, max_iter, pre, nullptr, stream);
    if (rc != SPA_OK) return rc;
    return spa_enforce_connectivity(ctx, pre, B, H, W, pl.min_size, pl.max_size, labels, n_labels,
                                    stream);
}
