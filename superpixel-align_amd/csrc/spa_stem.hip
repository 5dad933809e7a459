// The full-resolution stem of DRN-D (models/drn.py:134-145: layer0 = conv7x7(3->16)+BN+ReLU,
// layer1 = conv3x3(16->16)+BN+ReLU) as ONE kernel, float32 MFMA.
//
// Why it is not left to MIOpen like the rest of the network: with 3 / 16 channels the library's
// implicit-GEMM kernels reach 22-28 % of the float32 matrix peak (8.6 + 6.5 ms per 30 images at
// 1024x2048), each convolution is followed by a bias/ReLU pass over a 4 GB tensor, and the
// 16-channel full-resolution intermediate makes one extra trip through HBM.  Here a workgroup
//   A. stages the (16+8) x (32+8) x 3 input patch of its 16 x 32 output tile in LDS, normalised
//      on the way in (DRN.batch_predict, models/drn.py:319-321: x/255 float32, then (x - mean)
//      and (x / std) in float64 rounded to float32), zero outside the image (conv padding);
//   B. computes layer0 on the 18 x 34 pixels layer1 needs (v_mfma_f32_16x16x4_f32: 16 pixels x
//      16 channels per instruction, K = 7*7*3 = 147 padded to 148), adds the folded-BN bias,
//      applies ReLU and keeps the result in LDS (zeros outside the image: layer1's padding);
//   C. computes layer1 from that LDS tile (K = 3*3*16 = 144), bias, ReLU, and stores the
//      channels-last float32 map.
// The float32 MFMA is an exact k-ordered fmaf chain (no reduced precision); HBM traffic is the
// input once and the output once.
#include "spa_common.h"

#define ST_TH 16
#define ST_TW 32
#define ST_IH (ST_TH + 8)
#define ST_IW (ST_TW + 8)
#define ST_IPLANE (ST_IH * ST_IW + 8)        // floats per input channel plane (+8: bank spread)
#define ST_LH (ST_TH + 2)
#define ST_LW (ST_TW + 2)
#define ST_LP (ST_LH * ST_LW)                // layer0 pixels per tile (612)
#define ST_PS 17                             // floats per layer0 pixel in LDS (16 channels + 1: bank spread)
#define ST_K0 148                            // 147 padded to a multiple of 4
#define ST_S0 (ST_K0 / 4)                    // MFMA steps of layer0 (37)
#define ST_S1 (144 / 4)                      // MFMA steps of layer1 (36)
#define ST_PF 8                              // LDS read-ahead, in MFMA steps

typedef float stem_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short stem_bf16(float f)      // round to nearest even
{
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x0040u);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global store
// in flight (vmcnt(0)), i.e. for the previous tile's output to reach memory
__device__ __forceinline__ void stem_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__global__ __launch_bounds__(256) void k_drn_stem_d(const float *__restrict__ xn, int B, int H, int W,
                                                    const float *__restrict__ w0,    // [16][147]  (n, c*49+ky*7+kx)
                                                    const float *__restrict__ b0,    // [16]
                                                    const float *__restrict__ w1,    // [16][144]  (n, (ky*3+kx)*16+c)
                                                    const float *__restrict__ b1,    // [16]
                                                    void *__restrict__ yout, int out_bf16)
{
    float *__restrict__ y = (float *)yout;
    unsigned short *__restrict__ yh = (unsigned short *)yout;
    __shared__ float in_s[3 * ST_IPLANE];
    __shared__ float l0_s[ST_LP * ST_PS + 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const long long npix = (long long)H * W;
    const int tiles_x = (W + ST_TW - 1) / ST_TW, tiles_y = (H + ST_TH - 1) / ST_TH;
    const int n_tiles = tiles_x * tiles_y * B;

    // layer0 operands of this lane: B[k = 4s + g][n = m] and the LDS offset of input element k
    float wq[ST_S0];
    int koff[ST_S0];
#pragma unroll
    for (int s = 0; s < ST_S0; ++s) {
        const int k = 4 * s + g;
        const bool valid = k < 147;
        const int c = k / 49, rem = k - c * 49, ky = rem / 7, kx = rem - ky * 7;
        koff[s] = valid ? c * ST_IPLANE + ky * ST_IW + kx : 0;
        wq[s] = valid ? w0[m * 147 + k] : 0.0f;
    }
    const float bias0 = b0[m];
    // layer1 operands: B[k = 4s + g][n = m], k = (ky*3 + kx)*16 + c
    float wr[ST_S1];
#pragma unroll
    for (int s = 0; s < ST_S1; ++s) wr[s] = w1[m * 144 + 4 * s + g];
    const float bias1 = b1[m];

    // ---- A: input patch (already normalised, channels-last), origin (ty0 - 4, tx0 - 4); zero
    // outside the image (conv padding).  The loads of a tile are issued one tile ahead (while the
    // previous tile's layer1 runs) and straight-line: every thread loads from a clamped (always
    // valid) address and zeroes the value afterwards — a branch around a load would make each load
    // wait for the one before it.
    constexpr int NE = 3 * ST_IH * ST_IW, NU = (NE + 255) / 256;
    float raw[NU];
    // patch element e = tid + u * 256 = (iy * ST_IW + ix) * 3 + c of this thread: the same for every tile, so its
    // decomposition (two divisions) and its LDS offset are computed once
    int eiy[NU], eix[NU], elds[NU];          // eix: ix * 4 + c; elds: c * ST_IPLANE + pix, or -1 past the end
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        int e = tid + u * 256;
        const bool ok = e < NE;
        if (!ok) e = NE - 1;
        const int pix = e / 3, c = e - pix * 3;
        eiy[u] = pix / ST_IW;
        eix[u] = (pix - eiy[u] * ST_IW) * 4 + c;
        elds[u] = ok ? c * ST_IPLANE + pix : -1;
    }
    auto patch_load = [&](int tile) {
        const int b = tile / (tiles_x * tiles_y), tr = tile - b * (tiles_x * tiles_y);
        const int ty0 = (tr / tiles_x) * ST_TH, tx0 = (tr % tiles_x) * ST_TW;
        const float *src = xn + (long long)b * npix * 3;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int gy = ty0 - 4 + eiy[u], gx = tx0 - 4 + (eix[u] >> 2);
            const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
            const unsigned off = ((unsigned)cy * (unsigned)W + (unsigned)cx) * 12u + (unsigned)(eix[u] & 3) * 4u;
            raw[u] = *(const float *)((const char *)src + off);
        }
    };
    if ((int)blockIdx.x < n_tiles) patch_load(blockIdx.x);

    // persistent workgroups: the operand registers above are set up once, then tile after tile
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int b = tile / (tiles_x * tiles_y), tr = tile - b * (tiles_x * tiles_y);
    const int ty0 = (tr / tiles_x) * ST_TH, tx0 = (tr % tiles_x) * ST_TW;
    {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int gy = ty0 - 4 + eiy[u], gx = tx0 - 4 + (eix[u] >> 2);
            const bool in = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            if (elds[u] >= 0) in_s[elds[u]] = in ? raw[u] : 0.0f;
        }
    }
    stem_lds_barrier();

    // ---- B: layer0 on the 18 x 34 region (origin (ty0 - 1, tx0 - 1)), two 16-pixel tiles at a time
    for (int t = wv; t < (ST_LP + 15) / 16; t += 8) {
        const int t2 = t + 4;
        const bool two = t2 < (ST_LP + 15) / 16;
        int pa = t * 16 + m, pb = t2 * 16 + m;
        if (pa > ST_LP - 1) pa = ST_LP - 1;
        if (pb > ST_LP - 1) pb = ST_LP - 1;
        const int basea = (pa / ST_LW) * ST_IW + (pa % ST_LW);
        const int baseb = (pb / ST_LW) * ST_IW + (pb % ST_LW);
        stem_f4 acca = {0.0f, 0.0f, 0.0f, 0.0f}, accb = {0.0f, 0.0f, 0.0f, 0.0f};
        // the A operands come from LDS ST_PF steps ahead of the MFMA that uses them (a ring of
        // registers with compile-time indices), so the matrix pipe does not wait for LDS latency
        float ra[ST_PF], rb[ST_PF];
#pragma unroll
        for (int s = 0; s < ST_PF; ++s) { ra[s] = in_s[basea + koff[s]]; rb[s] = in_s[baseb + koff[s]]; }
#pragma unroll
        for (int s = 0; s < ST_S0; ++s) {
            const float aa = ra[s % ST_PF], ab = rb[s % ST_PF];
            if (s + ST_PF < ST_S0) {
                ra[s % ST_PF] = in_s[basea + koff[s + ST_PF]];
                rb[s % ST_PF] = in_s[baseb + koff[s + ST_PF]];
            }
            __builtin_amdgcn_sched_barrier(0);        // keep the reads ST_PF steps ahead of their use
            acca = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, wq[s], acca, 0, 0, 0);
            accb = __builtin_amdgcn_mfma_f32_16x16x4f32(ab, wq[s], accb, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // D[row = 4g + j][col = m]: row = pixel of the tile, col = channel
            const int qa = t * 16 + 4 * g + j;
            if (qa < ST_LP) {
                const int r = qa / ST_LW, q = qa - r * ST_LW;
                const int gy = ty0 - 1 + r, gx = tx0 - 1 + q;
                const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
                l0_s[qa * ST_PS + m] = in ? fmaxf(acca[j] + bias0, 0.0f) : 0.0f;
            }
            const int qb = t2 * 16 + 4 * g + j;
            if (two && qb < ST_LP) {
                const int r = qb / ST_LW, q = qb - r * ST_LW;
                const int gy = ty0 - 1 + r, gx = tx0 - 1 + q;
                const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
                l0_s[qb * ST_PS + m] = in ? fmaxf(accb[j] + bias0, 0.0f) : 0.0f;
            }
        }
    }
    stem_lds_barrier();
    if (tile + (int)gridDim.x < n_tiles) patch_load(tile + (int)gridDim.x);     // next tile's input travels under layer1

    // ---- C: layer1 on the 16 x 32 tile: 32 tiles of 16 pixels (row t>>1, columns (t&1)*16 ..)
    for (int t = wv; t < 32; t += 8) {
        const int t2 = t + 4;
        const int rowa = t >> 1, cola = (t & 1) * 16, rowb = t2 >> 1, colb = (t2 & 1) * 16;
        const float *pa = l0_s + (rowa * ST_LW + cola + m) * ST_PS + g;
        const float *pb = l0_s + (rowb * ST_LW + colb + m) * ST_PS + g;
        stem_f4 acca = {0.0f, 0.0f, 0.0f, 0.0f}, accb = {0.0f, 0.0f, 0.0f, 0.0f};
        float ra[ST_PF], rb[ST_PF];
#define ST_OFF1(s) ((((s) >> 2) / 3 * ST_LW + ((s) >> 2) % 3) * ST_PS + ((s) & 3) * 4)     // compile-time constant
#pragma unroll
        for (int s = 0; s < ST_PF; ++s) { ra[s] = pa[ST_OFF1(s)]; rb[s] = pb[ST_OFF1(s)]; }
#pragma unroll
        for (int s = 0; s < ST_S1; ++s) {
            const float aa = ra[s % ST_PF], ab = rb[s % ST_PF];
            if (s + ST_PF < ST_S1) {
                ra[s % ST_PF] = pa[ST_OFF1(s + ST_PF)];
                rb[s % ST_PF] = pb[ST_OFF1(s + ST_PF)];
            }
            __builtin_amdgcn_sched_barrier(0);
            acca = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, wr[s], acca, 0, 0, 0);
            accb = __builtin_amdgcn_mfma_f32_16x16x4f32(ab, wr[s], accb, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mm = 4 * g + j;
            int gy = ty0 + rowa, gx = tx0 + cola + mm;
            if (gy < H && gx < W) {
                const float v = fmaxf(acca[j] + bias1, 0.0f);
                const long long o = (((long long)b * H + gy) * W + gx) * 16 + m;
                if (out_bf16) yh[o] = stem_bf16(v); else y[o] = v;
            }
            gy = ty0 + rowb; gx = tx0 + colb + mm;
            if (gy < H && gx < W) {
                const float v = fmaxf(accb[j] + bias1, 0.0f);
                const long long o = (((long long)b * H + gy) * W + gx) * 16 + m;
                if (out_bf16) yh[o] = stem_bf16(v); else y[o] = v;
            }
        }
    }
    stem_lds_barrier();         // the next tile overwrites both LDS tiles
    }
}

// ---------------------------------------------------------------------------------------------------
// bfloat16 form of the same fused stem, for the bf16 network (BASELINE config 5): the operands are what
// the bf16 network defines them to be — the normalised image rounded to bfloat16, bfloat16 weights,
// float32 accumulation, layer0's output rounded to bfloat16 — on v_mfma_f32_16x16x32_bf16, i.e. 5-6 matrix
// instructions per 16 pixels instead of 37 + 36 float32 ones (the float32 kernel is bound by the float32
// matrix rate, 1/16 of this one).  The weights are the MFMA A operand (rows = output channels), the pixels
// the B operand, so a lane's accumulator is four consecutive channels of one pixel: 8-byte stores.
//   layer0: k = (ky*3 + c)*8 + kx, kx = 7 is a zero-weight pad: the eight k of a lane are eight consecutive
//           pixels of one patch row; two copies of the patch (the second shifted by one pixel) keep every
//           such read 4-byte aligned (ds_read_b32 x 4).  K = 168 padded to 192: 6 instructions.
//   layer1: k = tap*16 + c: eight consecutive channels of one layer0 pixel = one ds_read_b128 (48-byte pixel
//           stride in LDS: conflict-free).  K = 144 padded to 160: 5 instructions.
// ---------------------------------------------------------------------------------------------------
typedef __bf16 stem_bf8 __attribute__((ext_vector_type(8)));
// LDS bank mapping of layer0's fragment reads (one ds_read_b32 of a wave: 16 pixels x 4 k groups): the word
// offsets of a k group are c * plane + ky * row pitch, of the odd pixels' patch copy + copy stride.  With plane
// and copy strides that are multiples of 32 words the three channel planes (and both copies) fell on the same
// banks: 5.7 lanes per bank on average (PMC: SQ_LDS_BANK_CONFLICT = 4 x SQ_ACTIVE_INST_LDS).  Row pitch 25,
// plane stride 1 and copy stride 16 words (mod 32) — found by enumerating the strides — give 3.1 (2.0 is the
// floor: 64 lanes x 4 bytes over 32 banks).
#define SB_PW 50                               // patch row pitch in pixels (40 + the kx pad, even): 25 words
#define SB_PLANE (ST_IH * SB_PW + 18)          // pixels per patch plane: 609 words = 1 (mod 32)
#define SB_COPY (3 * SB_PLANE + 26)            // pixels per patch copy: 1 840 words = 16 (mod 32)
#define SB_L0_PITCH 24                         // bf16 per layer0 pixel in LDS (16 channels + 8: 48 bytes)

__device__ __forceinline__ unsigned short stem_bf16_rn(float f)
{
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
}

__global__ __launch_bounds__(256) void k_drn_stem_d_bf16(const unsigned short *__restrict__ xn, int B, int H, int W,   // normalised, bf16, (B,H,W,3)
                                                         const unsigned short *__restrict__ w0p,   // [6][64] x 8 bf16
                                                         const float *__restrict__ b0,
                                                         const unsigned short *__restrict__ w1p,   // [5][64] x 8 bf16
                                                         const float *__restrict__ b1,
                                                         unsigned short *__restrict__ yh,
                                                         unsigned short *__restrict__ y0h)   // DRN-C: layer0's output too (or NULL)
{
    // patch copy 0: pixel ix at [c][iy][ix]; copy 1: pixel ix at [c][iy][ix - 1] (so odd ix are 4-byte aligned)
    __shared__ __attribute__((aligned(16))) unsigned short in_s[2][SB_COPY];
    __shared__ __attribute__((aligned(16))) unsigned short l0_s[ST_LP * SB_L0_PITCH + 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const long long npix = (long long)H * W;
    const int tiles_x = (W + ST_TW - 1) / ST_TW, tiles_y = (H + ST_TH - 1) / ST_TH;
    const int n_tiles = tiles_x * tiles_y * B;

    // A operands (weights) of this lane, packed on the host in fragment order: [step][lane] x 8 bf16
    stem_bf8 wa0[6], wa1[5];
#pragma unroll
    for (int s = 0; s < 6; ++s) wa0[s] = *(const stem_bf8 *)(w0p + ((size_t)s * 64 + lane) * 8);
#pragma unroll
    for (int s = 0; s < 5; ++s) wa1[s] = *(const stem_bf8 *)(w1p + ((size_t)s * 64 + lane) * 8);
    // bias of the four channels this lane accumulates (rows 4g .. 4g+3)
    const float4 bias0 = *(const float4 *)(b0 + 4 * g), bias1 = *(const float4 *)(b1 + 4 * g);
    // layer0: patch offset (in pixels) of k group 4s + g: (ky, c) = divmod(group, 3); groups >= 21 are padding
    int goff[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int grp = 4 * s + g;
        const int ky = grp / 3, c = grp - ky * 3;
        goff[s] = grp < 21 ? c * SB_PLANE + ky * SB_PW : 0;
    }

    // layer1: offset of k group (step s, g): tap = 2s + g/2 (tap 9 is the zero-weight pad: any valid address),
    // channels 8 * (g & 1)
    int l1off[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        int tap = 2 * s + (g >> 1);
        if (tap > 8) tap = 8;
        const int ky = tap / 3, kx = tap - ky * 3;
        l1off[s] = (ky * ST_LW + kx) * SB_L0_PITCH + 8 * (g & 1);
    }
    constexpr int NE = 3 * ST_IH * ST_IW, NU = (NE + 255) / 256;
    unsigned short raw[NU];
    // patch element e = tid + u * 256 = (iy * ST_IW + ix) * 3 + c of this thread: the same for every tile, so its
    // decomposition (two divisions) and its LDS offset are computed once (PMC: the staging was 7 of the kernel's
    // 12 vector instructions per MFMA)
    int eiy[NU], eix[NU], elds[NU];          // elds: c * SB_PLANE + iy * SB_PW + ix, or -1 past the end
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        int e = tid + u * 256;
        const bool ok = e < NE;
        if (!ok) e = NE - 1;
        const int pix = e / 3, c = e - pix * 3;
        eiy[u] = pix / ST_IW; eix[u] = pix - eiy[u] * ST_IW;
        elds[u] = ok ? c * SB_PLANE + eiy[u] * SB_PW + eix[u] : -1;
        eix[u] = eix[u] * 4 + c;             // packed: ix and the channel
    }
    auto patch_load = [&](int tile) {
        const int b = tile / (tiles_x * tiles_y), tr = tile - b * (tiles_x * tiles_y);
        const int ty0 = (tr / tiles_x) * ST_TH, tx0 = (tr % tiles_x) * ST_TW;
        // the normalised image, already rounded to bfloat16 (spa_drn_normalise: the exact arithmetic of
        // DRN.batch_predict, then the bf16 network's rounding of its input), channels-last: 6 bytes per pixel
        const unsigned short *src = xn + (long long)b * npix * 3;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int gy = ty0 - 4 + eiy[u], gx = tx0 - 4 + (eix[u] >> 2);
            const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
            raw[u] = src[(unsigned)((cy * W + cx) * 3 + (eix[u] & 3))];
        }
    };
    // the pad columns (ix >= 40) of both copies are read by the kx = 7 lanes: keep them zero
    for (int i = tid; i < SB_COPY; i += 256) { in_s[0][i] = 0; in_s[1][i] = 0; }
    if ((int)blockIdx.x < n_tiles) patch_load(blockIdx.x);
    __syncthreads();

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / (tiles_x * tiles_y), tr = tile - b * (tiles_x * tiles_y);
        const int ty0 = (tr / tiles_x) * ST_TH, tx0 = (tr % tiles_x) * ST_TW;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int ix = eix[u] >> 2;
            const int gy = ty0 - 4 + eiy[u], gx = tx0 - 4 + ix;
            const bool in = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            if (elds[u] >= 0) {
                const unsigned short v = in ? raw[u] : (unsigned short)0;
                in_s[0][elds[u]] = v;
                if (ix >= 1) in_s[1][elds[u] - 1] = v;
            }
        }
        stem_lds_barrier();

        // ---- layer0 on the 18 x 34 region: 39 tiles of 16 pixels, pixel = B-operand column (lane & 15)
        for (int t = wv; t < (ST_LP + 15) / 16; t += 4) {
            int p = t * 16 + m;
            if (p > ST_LP - 1) p = ST_LP - 1;
            const int py = p / ST_LW, px = p - py * ST_LW;
            // eight consecutive patch pixels from x = px: from copy (px & 1) at an even index
            const unsigned short *base = in_s[px & 1] + py * SB_PW + (px & ~1);
            stem_f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const uint32_t *q = (const uint32_t *)(base + goff[s]);
                union { uint32_t u[4]; stem_bf8 v; } f;
                f.u[0] = q[0]; f.u[1] = q[1]; f.u[2] = q[2]; f.u[3] = q[3];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa0[s], f.v, acc, 0, 0, 0);
            }
            // D[row = 4g + j = channel][col = m = pixel]
            const int q0 = t * 16 + m;
            if (q0 < ST_LP) {
                const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
                const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
                const float v0 = in ? fmaxf(acc[0] + bias0.x, 0.0f) : 0.0f, v1 = in ? fmaxf(acc[1] + bias0.y, 0.0f) : 0.0f;
                const float v2 = in ? fmaxf(acc[2] + bias0.z, 0.0f) : 0.0f, v3 = in ? fmaxf(acc[3] + bias0.w, 0.0f) : 0.0f;
                uint2 o;
                o.x = (unsigned)stem_bf16_rn(v0) | ((unsigned)stem_bf16_rn(v1) << 16);
                o.y = (unsigned)stem_bf16_rn(v2) | ((unsigned)stem_bf16_rn(v3) << 16);
                *(uint2 *)(l0_s + q0 * SB_L0_PITCH + 4 * g) = o;
                if (y0h && in && py >= 1 && py <= ST_TH && px >= 1 && px <= ST_TW)       // the tile's own pixels (not its halo)
                    *(uint2 *)(y0h + ((((long long)b * H + gy) * W + gx) * 16 + 4 * g)) = o;
            }
        }
        stem_lds_barrier();
        if (tile + (int)gridDim.x < n_tiles) patch_load(tile + (int)gridDim.x);     // next tile's input travels under layer1

        // ---- layer1 on the 16 x 32 tile: 32 tiles of 16 pixels (row t >> 1, columns (t & 1) * 16 ..)
        for (int t = wv; t < 32; t += 4) {
            const int row = t >> 1, col = (t & 1) * 16 + m;
            stem_f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            const unsigned short *lb = l0_s + (row * ST_LW + col) * SB_L0_PITCH;
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const stem_bf8 f = *(const stem_bf8 *)(lb + l1off[s]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa1[s], f, acc, 0, 0, 0);
            }
            const int gy = ty0 + row, gx = tx0 + col;
            if (gy < H && gx < W) {
                uint2 o;
                o.x = (unsigned)stem_bf16_rn(fmaxf(acc[0] + bias1.x, 0.0f)) | ((unsigned)stem_bf16_rn(fmaxf(acc[1] + bias1.y, 0.0f)) << 16);
                o.y = (unsigned)stem_bf16_rn(fmaxf(acc[2] + bias1.z, 0.0f)) | ((unsigned)stem_bf16_rn(fmaxf(acc[3] + bias1.w, 0.0f)) << 16);
                *(uint2 *)(yh + ((((long long)b * H + gy) * W + gx) * 16 + 4 * g)) = o;
            }
        }
        stem_lds_barrier();         // the next tile overwrites both LDS tiles
    }
}

// ---------------------------------------------------------------------------------------------------
// float32 stem on the 16-bit matrix cores at float32 accuracy (round 3; the scheme of spa_gemm16.hip): every operand is
// two half-precision planes h = rn(s x), l = rn(s x - h) of an exactly scaled value and every product three
// v_mfma_f32_16x16x32_f16 (wl.xh + wh.xl + wh.xh, float32 accumulation) — 18 + 15 matrix instructions of 16 passes per
// 16 pixels instead of 37 + 36 float32 ones of 32 passes.  Geometry, k orders, the two shifted patch copies and the
// persistent tile loop are the bf16 kernel's above; the planes double the LDS (78 KB: two workgroups per CU).
// Scales (powers of two, all static): the normalised image is below 4 in magnitude -> 2^12; layer0's output is bounded
// by max_n (4 sum_k |w0[n][k]| + |b0[n]|) -> s1 = 2^(14 - ceil(log2 bound)); the weights by their largest magnitude
// (k_stem_pack_f16 computes them and leaves the two unscale factors next to the fragments).
// ---------------------------------------------------------------------------------------------------
typedef _Float16 stem_h8 __attribute__((ext_vector_type(8)));
// LDS images (round 4; MI355X_MICROARCH.md, LDS: a ds_read_b32 is served in two groups of 32 lanes, bank = word mod 32; a
// ds_read_b128 in four groups of 16 lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32, bank line = 16 slots of 16 bytes).
// Patch (layer 0 reads 4 words = 8 taps of one (channel, ky) row per lane): the 32 lanes of a group are 16 pixels x two k
// groups; even and odd pixels read the two copies, the two k groups two (channel, ky) rows.  With 24-word rows, planes of
// 584 words (8 mod 32; the wrap from channel 2 to the next ky is then 24 - 2 * 584 = 8 mod 32 as well) and copies of 1 776
// words (16 mod 32) the four sets of eight consecutive words fall on banks b + {0-7}, {8-15}, {16-23}, {24-31}.  The bf16
// kernel's geometry above (planes 1 mod 32) served these reads with 2-way conflicts.
// Layer 0's output (layer 1 reads 16 bytes of h and of l per lane): 64-byte pixels, the chunks [h 0-7 | h 8-15 | l 0-7 | l 8-15]
// of pixel q stored with the h and l pairs swapped when (q >> 2) is odd; a 16-lane group reads chunk c of pixels q + {0-3, 12-15}
// and chunk c ^ 1 of pixels q + {4-11}: 16 different slots for every q.  (80-byte pixels: 2-way conflicts, and 10 KB more.)
#define SH_PW 48                               // patch row pitch in pixels (40 + the kx pad): 24 words
#define SH_PLANE (ST_IH * SH_PW + 16)          // pixels per patch plane: 584 words = 8 (mod 32)
#define SH_COPY (3 * SH_PLANE + 48)            // pixels per patch copy: 1 776 words = 16 (mod 32)
#define SH_L0_PITCH 32                         // halfs per layer0 pixel in LDS: 16 h | 16 l, chunk pairs swapped by (q >> 2) & 1
#define SH_IN_S0 4096.0f                       // 2^12: |normalised input| < 4

__device__ __forceinline__ unsigned short stem_f16_bits(float f)
{
    const _Float16 h = (_Float16)f;
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float stem_f16_val(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }

// wp: [22][64] x 8 halfs (layer0 h: steps 0-5, layer0 l: 6-11, layer1 h: 12-16, layer1 l: 17-21), then 4 floats:
// unscale0 = 1 / (2^12 t0), s1, unscale1 = 1 / (s1 t1), -
__global__ __launch_bounds__(256, 2) void k_drn_stem_d_f16x3(const float *__restrict__ xn, int B, int H, int W,   // normalised, (B,H,W,3)
                                                             const unsigned short *__restrict__ wp,
                                                             const float *__restrict__ b0, const float *__restrict__ b1,
                                                             float *__restrict__ y, unsigned *__restrict__ amax_out,
                                                             float *__restrict__ y0)
{
    // y0 (DRN-C, models/drn.py:134-170): layer0's output relu(conv7x7 + bias) as a tensor of its own — the residual of the
    // BasicBlock whose first convolution is this kernel's second stage
    __shared__ __attribute__((aligned(16))) unsigned short in_h[2][SH_COPY], in_l[2][SH_COPY];
    unsigned amx = 0;                            // largest value stored (the scale of the layer that reads y)
    __shared__ __attribute__((aligned(16))) unsigned short l0_s[ST_LP * SH_L0_PITCH + 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const long long npix = (long long)H * W;
    const int tiles_x = (W + ST_TW - 1) / ST_TW, tiles_y = (H + ST_TH - 1) / ST_TH;
    const int n_tiles = tiles_x * tiles_y * B;

    stem_h8 w0h[6], w0l[6], w1h[5], w1l[5];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        w0h[s] = *(const stem_h8 *)(wp + ((size_t)s * 64 + lane) * 8);
        w0l[s] = *(const stem_h8 *)(wp + ((size_t)(6 + s) * 64 + lane) * 8);
    }
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        w1h[s] = *(const stem_h8 *)(wp + ((size_t)(12 + s) * 64 + lane) * 8);
        w1l[s] = *(const stem_h8 *)(wp + ((size_t)(17 + s) * 64 + lane) * 8);
    }
    const float *sc = (const float *)(wp + (size_t)22 * 64 * 8);
    const float unscale0 = sc[0], s1 = sc[1], unscale1 = sc[2];
    const float4 bias0 = *(const float4 *)(b0 + 4 * g), bias1 = *(const float4 *)(b1 + 4 * g);
    int goff[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int grp = 4 * s + g;
        const int ky = grp / 3, c = grp - ky * 3;
        goff[s] = grp < 21 ? c * SH_PLANE + ky * SH_PW : 0;
    }
    int l1off[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        int tap = 2 * s + (g >> 1);
        if (tap > 8) tap = 8;
        const int ky = tap / 3, kx = tap - ky * 3;
        l1off[s] = ky * ST_LW + kx;               // in pixels
    }
    constexpr int NE = 3 * ST_IH * ST_IW, NU = (NE + 255) / 256;
    float raw[NU];
    int eiy[NU], eix[NU], elds[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        int e = tid + u * 256;
        const bool ok = e < NE;
        if (!ok) e = NE - 1;
        const int pix = e / 3, c = e - pix * 3;
        eiy[u] = pix / ST_IW; eix[u] = pix - eiy[u] * ST_IW;
        elds[u] = ok ? c * SH_PLANE + eiy[u] * SH_PW + eix[u] : -1;
        eix[u] = eix[u] * 4 + c;
    }
    auto patch_load = [&](int tile) {
        const int b = tile / (tiles_x * tiles_y), tr = tile - b * (tiles_x * tiles_y);
        const int ty0 = (tr / tiles_x) * ST_TH, tx0 = (tr % tiles_x) * ST_TW;
        const float *src = xn + (long long)b * npix * 3;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int gy = ty0 - 4 + eiy[u], gx = tx0 - 4 + (eix[u] >> 2);
            const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
            raw[u] = src[(unsigned)((cy * W + cx) * 3 + (eix[u] & 3))];
        }
    };
    for (int i = tid; i < SH_COPY; i += 256) { in_h[0][i] = 0; in_h[1][i] = 0; in_l[0][i] = 0; in_l[1][i] = 0; }
    if ((int)blockIdx.x < n_tiles) patch_load(blockIdx.x);
    __syncthreads();

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / (tiles_x * tiles_y), tr = tile - b * (tiles_x * tiles_y);
        const int ty0 = (tr / tiles_x) * ST_TH, tx0 = (tr % tiles_x) * ST_TW;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int ix = eix[u] >> 2;
            const int gy = ty0 - 4 + eiy[u], gx = tx0 - 4 + ix;
            const bool in = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            if (elds[u] >= 0) {
                const float v = in ? raw[u] * SH_IN_S0 : 0.0f;
                const unsigned short vh = stem_f16_bits(v);
                const unsigned short vl = stem_f16_bits(v - stem_f16_val(vh));
                in_h[0][elds[u]] = vh; in_l[0][elds[u]] = vl;
                if (ix >= 1) { in_h[1][elds[u] - 1] = vh; in_l[1][elds[u] - 1] = vl; }
            }
        }
        stem_lds_barrier();

        // ---- layer0 on the 18 x 34 region
        for (int t = wv; t < (ST_LP + 15) / 16; t += 4) {
            int p = t * 16 + m;
            if (p > ST_LP - 1) p = ST_LP - 1;
            const int py = p / ST_LW, px = p - py * ST_LW;
            const int boff = py * SH_PW + (px & ~1);
            const unsigned short *bh = in_h[px & 1] + boff, *bl = in_l[px & 1] + boff;
            stem_f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const uint32_t *qh = (const uint32_t *)(bh + goff[s]), *ql = (const uint32_t *)(bl + goff[s]);
                union { uint32_t u[4]; stem_h8 v; } fh, fl;
                fh.u[0] = qh[0]; fh.u[1] = qh[1]; fh.u[2] = qh[2]; fh.u[3] = qh[3];
                fl.u[0] = ql[0]; fl.u[1] = ql[1]; fl.u[2] = ql[2]; fl.u[3] = ql[3];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0l[s], fh.v, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0h[s], fl.v, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0h[s], fh.v, acc, 0, 0, 0);
            }
            const int q0 = t * 16 + m;
            if (q0 < ST_LP) {
                const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
                const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
                float v[4] = {acc[0] * unscale0 + bias0.x, acc[1] * unscale0 + bias0.y, acc[2] * unscale0 + bias0.z, acc[3] * unscale0 + bias0.w};
                unsigned short vh[4], vl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float r = in ? fmaxf(v[j], 0.0f) * s1 : 0.0f;
                    vh[j] = stem_f16_bits(r);
                    vl[j] = stem_f16_bits(r - stem_f16_val(vh[j]));
                }
                if (y0 && in && py >= 1 && py <= ST_TH && px >= 1 && px <= ST_TW)       // the tile's own pixels (not its halo)
                    *(float4 *)(y0 + ((((long long)b * H + gy) * W + gx) * 16 + 4 * g)) =
                        make_float4(fmaxf(v[0], 0.0f), fmaxf(v[1], 0.0f), fmaxf(v[2], 0.0f), fmaxf(v[3], 0.0f));
                const int sw = (q0 << 2) & 16;                   // 16 halfs when (q0 >> 2) is odd: h and l pairs swapped
                unsigned short *o = l0_s + q0 * SH_L0_PITCH + 4 * g;
                *(uint2 *)(o + sw) = make_uint2((unsigned)vh[0] | ((unsigned)vh[1] << 16), (unsigned)vh[2] | ((unsigned)vh[3] << 16));
                *(uint2 *)(o + 16 - sw) = make_uint2((unsigned)vl[0] | ((unsigned)vl[1] << 16), (unsigned)vl[2] | ((unsigned)vl[3] << 16));
            }
        }
        stem_lds_barrier();
        if (tile + (int)gridDim.x < n_tiles) patch_load(tile + (int)gridDim.x);

        // ---- layer1 on the 16 x 32 tile
        for (int t = wv; t < 32; t += 4) {
            const int row = t >> 1, col = (t & 1) * 16 + m;
            stem_f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            const int qb = row * ST_LW + col;
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const int q = qb + l1off[s];
                const int sw = (q << 2) & 16;
                const unsigned short *lb = l0_s + q * SH_L0_PITCH + 8 * (g & 1);
                const stem_h8 fh = *(const stem_h8 *)(lb + sw);
                const stem_h8 fl = *(const stem_h8 *)(lb + 16 - sw);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1l[s], fh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h[s], fl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h[s], fh, acc, 0, 0, 0);
            }
            const int gy = ty0 + row, gx = tx0 + col;
            if (gy < H && gx < W) {
                const float4 o = make_float4(fmaxf(acc[0] * unscale1 + bias1.x, 0.0f), fmaxf(acc[1] * unscale1 + bias1.y, 0.0f),
                                             fmaxf(acc[2] * unscale1 + bias1.z, 0.0f), fmaxf(acc[3] * unscale1 + bias1.w, 0.0f));
                *(float4 *)(y + ((((long long)b * H + gy) * W + gx) * 16 + 4 * g)) = o;
                amx = max(amx, max(max(__float_as_uint(o.x), __float_as_uint(o.y)), max(__float_as_uint(o.z), __float_as_uint(o.w))));
            }
        }
        stem_lds_barrier();
    }
    if (amax_out) {
        for (int o = 32; o > 0; o >>= 1) amx = max(amx, (unsigned)__shfl_xor((int)amx, o));
        if (lane == 0 && amx > *(volatile unsigned *)amax_out) atomicMax(amax_out, amx);
    }
}

// ---------------------------------------------------------------------------------------------------
// Layer 2 of DRN-D (models/drn.py:134-145 `_make_conv_layers`: conv3x3(16 -> 32, stride 2, padding 1) + BN + ReLU at half
// resolution) on the 16-bit matrix cores at float32 accuracy — the last convolution that ran on MIOpen (its 16 input
// channels do not fit the 32-channel K steps of k_conv3x3_f32): ~2.3 ms + a 0.85 ms epilogue pass per 30 images.  Built
// like the stem's layer 1: a workgroup stages the 9 x 65 input pixels of its 4 x 32 output tile in LDS as two
// half-precision planes (80-byte pixels: conflict-free 16-byte fragment reads at the stride-2 pixel step), K = 9 taps x
// 16 channels = 144 padded to 160 (five steps of two taps), two 16-channel tiles of output channels, three matrix
// instructions per product; bias, ReLU, the largest stored magnitude (the next layer's scale) in the epilogue.  The input
// patch of the next tile is loaded while this tile computes.  HBM: the input once (9/8 of it), the output once.
// ---------------------------------------------------------------------------------------------------
#define L2_TH 4
#define L2_TW 32
#define L2_PH (2 * L2_TH + 1)
#define L2_PW (2 * L2_TW + 1)
#define L2_PITCH 40                            // halfs per staged pixel: 16 h | 16 l | 8 pad

// wp: [2 channel tiles][2 planes][5 steps][64 lanes] x 8 halfs (A fragments of t * w); x (B,H,W,16) float32; y (B,Ho,Wo,32)
__global__ __launch_bounds__(256, 2) void k_drn_layer2_f16x3(const float *__restrict__ x, int B, int H, int W, int Ho, int Wo,
                                                             const unsigned short *__restrict__ wp, const float *__restrict__ bias,
                                                             float inv_t, const unsigned *__restrict__ amax_in,
                                                             unsigned *__restrict__ amax_out, float *__restrict__ y)
{
    __shared__ __attribute__((aligned(16))) unsigned short patch[L2_PH * L2_PW * L2_PITCH + 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int tiles_x = (Wo + L2_TW - 1) / L2_TW, tiles_y = (Ho + L2_TH - 1) / L2_TH;
    const int n_tiles = tiles_x * tiles_y * B;
    stem_h8 wh[2][5], wl[2][5];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            wh[ct][s] = *(const stem_h8 *)(wp + ((size_t)((ct * 2 + 0) * 5 + s) * 64 + lane) * 8);
            wl[ct][s] = *(const stem_h8 *)(wp + ((size_t)((ct * 2 + 1) * 5 + s) * 64 + lane) * 8);
        }
    float sc, unscale;
    {
        const unsigned bits = *amax_in;
        int e = (int)(bits >> 23) - 127;
        e = bits == 0u ? 0 : (e < -100 ? -100 : (e > 100 ? 100 : e));
        sc = __uint_as_float((unsigned)(127 + 14 - e) << 23);
        unscale = __uint_as_float((unsigned)(127 - 14 + e) << 23) * inv_t;
    }
    const float4 bias_lo = *(const float4 *)(bias + 4 * g), bias_hi = *(const float4 *)(bias + 16 + 4 * g);
    int toff[5];                                   // patch offset (halfs) of k group (step s, g): tap = 2s + g/2, channels 8 (g & 1)
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        int tap = 2 * s + (g >> 1);
        if (tap > 8) tap = 8;                      // zero-weight pad: any valid address
        toff[s] = ((tap / 3) * L2_PW + tap % 3) * L2_PITCH + 8 * (g & 1);
    }
    constexpr int NE = L2_PH * L2_PW * 4, NU = (NE + 255) / 256;           // float4 elements of a patch
    float4 raw[NU];
    int epix[NU], eq[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        int e = tid + u * 256;
        if (e >= NE) e = -1;
        epix[u] = e < 0 ? -1 : e >> 2;
        eq[u] = e & 3;
    }
    auto patch_load = [&](int tile) {
        const int b = tile / (tiles_x * tiles_y), tr = tile - b * (tiles_x * tiles_y);
        const int ty0 = (tr / tiles_x) * L2_TH, tx0 = (tr % tiles_x) * L2_TW;
        const float *src = x + (long long)b * H * W * 16;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int pix = epix[u] < 0 ? 0 : epix[u];
            const int iy = pix / L2_PW, ix = pix - iy * L2_PW;
            const int gy = 2 * ty0 - 1 + iy, gx = 2 * tx0 - 1 + ix;
            const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
            raw[u] = *(const float4 *)(src + ((long long)cy * W + cx) * 16 + 4 * eq[u]);
        }
    };
    unsigned amx = 0;
    if ((int)blockIdx.x < n_tiles) patch_load(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / (tiles_x * tiles_y), tr = tile - b * (tiles_x * tiles_y);
        const int ty0 = (tr / tiles_x) * L2_TH, tx0 = (tr % tiles_x) * L2_TW;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (epix[u] < 0) continue;
            const int iy = epix[u] / L2_PW, ix = epix[u] - iy * L2_PW;
            const int gy = 2 * ty0 - 1 + iy, gx = 2 * tx0 - 1 + ix;
            const bool in = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            const float v[4] = {in ? raw[u].x * sc : 0.f, in ? raw[u].y * sc : 0.f, in ? raw[u].z * sc : 0.f, in ? raw[u].w * sc : 0.f};
            unsigned short vh[4], vl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { vh[j] = stem_f16_bits(v[j]); vl[j] = stem_f16_bits(v[j] - stem_f16_val(vh[j])); }
            unsigned short *o = patch + epix[u] * L2_PITCH + 4 * eq[u];
            *(uint2 *)o = make_uint2((unsigned)vh[0] | ((unsigned)vh[1] << 16), (unsigned)vh[2] | ((unsigned)vh[3] << 16));
            *(uint2 *)(o + 16) = make_uint2((unsigned)vl[0] | ((unsigned)vl[1] << 16), (unsigned)vl[2] | ((unsigned)vl[3] << 16));
        }
        stem_lds_barrier();
        if (tile + (int)gridDim.x < n_tiles) patch_load(tile + (int)gridDim.x);     // next tile's input travels under the matrix work
        // 8 pixel tiles of 16 (row t >> 1 of the output tile, columns (t & 1) * 16 ..): two per wave
        for (int t = wv; t < (L2_TH * L2_TW) / 16; t += 4) {
            const int row = t >> 1, col = (t & 1) * 16 + m;
            const unsigned short *pb = patch + ((2 * row) * L2_PW + 2 * col) * L2_PITCH;
            stem_f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const stem_h8 fh = *(const stem_h8 *)(pb + toff[s]);
                const stem_h8 fl = *(const stem_h8 *)(pb + toff[s] + 16);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[0][s], fh, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[1][s], fh, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[0][s], fl, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[1][s], fl, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[0][s], fh, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[1][s], fh, a1, 0, 0, 0);
            }
            const int gy = ty0 + row, gx = tx0 + col;
            if (gy < Ho && gx < Wo) {
                const float4 o0 = make_float4(fmaxf(a0[0] * unscale + bias_lo.x, 0.f), fmaxf(a0[1] * unscale + bias_lo.y, 0.f),
                                              fmaxf(a0[2] * unscale + bias_lo.z, 0.f), fmaxf(a0[3] * unscale + bias_lo.w, 0.f));
                const float4 o1 = make_float4(fmaxf(a1[0] * unscale + bias_hi.x, 0.f), fmaxf(a1[1] * unscale + bias_hi.y, 0.f),
                                              fmaxf(a1[2] * unscale + bias_hi.z, 0.f), fmaxf(a1[3] * unscale + bias_hi.w, 0.f));
                float *o = y + (((long long)b * Ho + gy) * Wo + gx) * 32 + 4 * g;
                *(float4 *)o = o0;
                *(float4 *)(o + 16) = o1;
                amx = max(amx, max(max(__float_as_uint(o0.x), __float_as_uint(o0.y)), max(__float_as_uint(o0.z), __float_as_uint(o0.w))));
                amx = max(amx, max(max(__float_as_uint(o1.x), __float_as_uint(o1.y)), max(__float_as_uint(o1.z), __float_as_uint(o1.w))));
            }
        }
        stem_lds_barrier();
    }
    if (amax_out) {
        for (int o = 32; o > 0; o >>= 1) amx = max(amx, (unsigned)__shfl_xor((int)amx, o));
        if (lane == 0 && amx > *(volatile unsigned *)amax_out) atomicMax(amax_out, amx);
    }
}

// x (B,H,W,16) float32 channels-last -> y (B,(H+1)/2,(W+1)/2,32) = relu(conv3x3 stride 2 padding 1 + bias); wp = the packed
// planes of t * w (layout above), inv_t = 1 / t; amax_in / amax_out as in spa_conv3x3_wino4_f16s
extern "C" int spa_drn_layer2_f16s(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, const void *wp, float inv_t,
                                   const float *bias, const void *amax_in, void *amax_out, float *y, void *stream)
{
    SPA_ARG(ctx && x && wp && bias && amax_in && y && B > 0 && H > 0 && W > 0 && inv_t > 0.f);
    SPA_ARG((((uintptr_t)x | (uintptr_t)wp | (uintptr_t)bias | (uintptr_t)y) & 15) == 0);
    hipStream_t s = spa_stream(stream);
    if (amax_out) spa_zero_word(amax_out, s);
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long n_tiles = (long long)((Wo + L2_TW - 1) / L2_TW) * ((Ho + L2_TH - 1) / L2_TH) * B;
    SPA_ARG(n_tiles < (1ll << 31));
    SpaProfScope prof_(ctx, PROF_DRN_CONV16_FRONT, s);
    long long grid = 3ll * ctx->n_cu;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(k_drn_layer2_f16x3, dim3((unsigned)grid), dim3(256), 0, s, x, B, H, W, Ho, Wo, (const unsigned short *)wp, bias,
                       inv_t, (const unsigned *)amax_in, (unsigned *)amax_out, y);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// Layer 2 in plain float32 (round 6: the strict float32 network, `--fp32_mfma_gemm`, on own kernels throughout): one thread per output
// pixel, 32 accumulators, the 9 x 16 x 32 weights in LDS read as wave-wide broadcasts, every output an fmaf chain in (tap, channel)
// order.  4 608 fused multiply-adds per pixel on the vector pipe: ~3 ms per 30 full-size images — the strict mode's step is 120 ms.
// w9: (9 taps, 16 input channels, 32 output channels) float32
__global__ __launch_bounds__(256) void k_drn_layer2_f32(const float *__restrict__ x, int B, int H, int W, int Ho, int Wo,
                                                       const float4 *__restrict__ w9, const float *__restrict__ bias, float *__restrict__ y)
{
    __shared__ float4 lw[9 * 16 * 8];
    for (int i = threadIdx.x; i < 9 * 16 * 8; i += 256) lw[i] = w9[i];
    __syncthreads();
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long long)B * Ho * Wo) return;
    const int xo = (int)(id % Wo);
    const long long t = id / Wo;
    const int yo = (int)(t % Ho), b = (int)(t / Ho);
    typedef float l2f2 __attribute__((ext_vector_type(2)));
    l2f2 acc[16];                                   // (packed fmas: two output channels per instruction, each an IEEE fma)
#pragma unroll
    for (int n = 0; n < 16; ++n) acc[n] = (l2f2){bias[2 * n], bias[2 * n + 1]};
    for (int tap = 0; tap < 9; ++tap) {
        const int yy = 2 * yo - 1 + tap / 3, xx = 2 * xo - 1 + tap % 3;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const float4 *px = (const float4 *)(x + (((long long)b * H + yy) * W + xx) * 16);
        const float4 v4[4] = {px[0], px[1], px[2], px[3]};
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float v = c & 2 ? (c & 1 ? v4[c >> 2].w : v4[c >> 2].z) : (c & 1 ? v4[c >> 2].y : v4[c >> 2].x);
            const l2f2 vv = {v, v};
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const float4 wv = lw[(tap * 16 + c) * 8 + n];
                acc[2 * n] = __builtin_elementwise_fma(vv, (l2f2){wv.x, wv.y}, acc[2 * n]);
                acc[2 * n + 1] = __builtin_elementwise_fma(vv, (l2f2){wv.z, wv.w}, acc[2 * n + 1]);
            }
        }
    }
    float4 *o = (float4 *)(y + id * 32);
#pragma unroll
    for (int n = 0; n < 8; ++n)
        o[n] = make_float4(fmaxf(acc[2 * n][0], 0.f), fmaxf(acc[2 * n][1], 0.f), fmaxf(acc[2 * n + 1][0], 0.f), fmaxf(acc[2 * n + 1][1], 0.f));
}

// x (B,H,W,16) float32 channels-last -> y (B,(H+1)/2,(W+1)/2,32) = relu(conv3x3 stride 2 padding 1 + bias), float32 throughout;
// w9 (9,16,32) float32 = the weights as (ky*3+kx, input channel, output channel)
extern "C" int spa_drn_layer2_f32(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, const float *w9,
                                  const float *bias, float *y, void *stream)
{
    SPA_ARG(ctx && x && w9 && bias && y && B > 0 && H > 0 && W > 0);
    SPA_ARG((((uintptr_t)x | (uintptr_t)w9 | (uintptr_t)bias | (uintptr_t)y) & 15) == 0);
    hipStream_t s = spa_stream(stream);
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long n = (long long)B * Ho * Wo;
    SPA_ARG((n + 255) / 256 < (1ll << 31));
    SpaProfScope prof_(ctx, PROF_DRN_CONV32, s);
    hipLaunchKernelGGL(k_drn_layer2_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, B, H, W, Ho, Wo, (const float4 *)w9, bias, y);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// weights -> scaled half-precision planes in MFMA A fragment order (one workgroup of 64 lanes per step, the k orders of
// k_stem_pack_bf16), and the scale factors.  Every block recomputes the three maxima (2 352 + 2 304 weights: cheap).
__global__ __launch_bounds__(64) void k_stem_pack_f16(const float *__restrict__ w0, const float *__restrict__ b0,
                                                      const float *__restrict__ w1, unsigned short *__restrict__ wp)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    float m0 = 0.f, m1 = 0.f, bound = 0.f;
    for (int i = lane; i < 16 * 147; i += 64) m0 = fmaxf(m0, fabsf(w0[i]));
    for (int i = lane; i < 16 * 144; i += 64) m1 = fmaxf(m1, fabsf(w1[i]));
    if (lane < 16) {
        float a = 0.f;
        for (int k = 0; k < 147; ++k) a += fabsf(w0[lane * 147 + k]);
        bound = 4.0f * a + fabsf(b0[lane]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        m0 = fmaxf(m0, __shfl_xor(m0, o)); m1 = fmaxf(m1, __shfl_xor(m1, o)); bound = fmaxf(bound, __shfl_xor(bound, o));
    }
    auto pow2_to_2_14 = [](float mx) {           // power of two t with t * mx in [2^14, 2^15)  (mx > 0)
        const int e = (int)(__float_as_uint(mx) >> 23) - 127;
        return __uint_as_float((unsigned)(127 + 14 - e) << 23);
    };
    const float t0 = pow2_to_2_14(fmaxf(m0, 1e-30f)), t1 = pow2_to_2_14(fmaxf(m1, 1e-30f));
    const float s1 = 0.5f * pow2_to_2_14(fmaxf(bound, 1e-30f));       // bound < 2^(e+1): s1 * bound < 2^15
    if (s == 22) {
        if (lane == 0) {
            float *sc = (float *)(wp + (size_t)22 * 64 * 8);
            sc[0] = 1.0f / (SH_IN_S0 * t0); sc[1] = s1; sc[2] = 1.0f / (s1 * t1); sc[3] = 0.f;
        }
        return;
    }
    const int n = lane & 15, g = lane >> 4;
    unsigned short *o = wp + ((size_t)s * 64 + lane) * 8;
    const bool lo = (s >= 6 && s < 12) || s >= 17;
    if (s < 12) {                                  // layer0: k = (ky*3 + c)*8 + kx
        const int s0 = s % 6, grp = 4 * s0 + g, ky = grp / 3, c = grp - ky * 3;
        for (int j = 0; j < 8; ++j) {
            const float v = (grp < 21 && j < 7) ? w0[n * 147 + c * 49 + ky * 7 + j] * t0 : 0.f;
            const unsigned short h = stem_f16_bits(v);
            o[j] = lo ? stem_f16_bits(v - stem_f16_val(h)) : h;
        }
    } else {                                       // layer1: k = tap*16 + c
        const int s1i = (s - 12) % 5, tap = 2 * s1i + (g >> 1), c0 = 8 * (g & 1);
        for (int j = 0; j < 8; ++j) {
            const float v = tap <= 8 ? w1[n * 144 + tap * 16 + c0 + j] * t1 : 0.f;
            const unsigned short h = stem_f16_bits(v);
            o[j] = lo ? stem_f16_bits(v - stem_f16_val(h)) : h;
        }
    }
}

// weights -> MFMA A fragments: [step][lane] x 8 bf16; lane = (channel n = lane & 15, k group g = lane >> 4)
__global__ void k_stem_pack_bf16(const float *__restrict__ w0, const float *__restrict__ w1, unsigned short *__restrict__ wp)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    const int n = lane & 15, g = lane >> 4;
    unsigned short *o = wp + ((size_t)s * 64 + lane) * 8;
    if (s < 6) {                                   // layer0: k = (ky*3 + c)*8 + kx
        const int grp = 4 * s + g, ky = grp / 3, c = grp - ky * 3;
        for (int j = 0; j < 8; ++j)
            o[j] = (grp < 21 && j < 7) ? stem_bf16_rn(w0[n * 147 + c * 49 + ky * 7 + j]) : (unsigned short)0;
    } else {                                       // layer1: k = tap*16 + c
        const int s1 = s - 6, tap = 2 * s1 + (g >> 1), c0 = 8 * (g & 1);
        for (int j = 0; j < 8; ++j)
            o[j] = tap <= 8 ? stem_bf16_rn(w1[n * 144 + tap * 16 + c0 + j]) : (unsigned short)0;
    }
}

static int stem_impl(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W,
                     const float *w0, const float *b0, const float *w1, const float *b1,
                     const double *mean3_host, const double *std3_host, void *y, int32_t out_dtype,
                     float *xn_scratch, void *amax_out, void *stream, float *y0 = nullptr);

extern "C" int spa_drn_stem_d(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W,
                              const float *w0, const float *b0, const float *w1, const float *b1,
                              const double *mean3_host, const double *std3_host, void *y, int32_t out_dtype,
                              float *xn_scratch, void *stream)
{
    return stem_impl(ctx, x, B, H, W, w0, b0, w1, b1, mean3_host, std3_host, y, out_dtype, xn_scratch, nullptr, stream);
}

// out_dtype 2 (float32 output on the 16-bit matrix cores) and amax_out[0] = bit pattern of the largest value stored: the
// scale input of spa_drn_layer2_f16s
extern "C" int spa_drn_stem_d_amax(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W,
                                   const float *w0, const float *b0, const float *w1, const float *b1,
                                   const double *mean3_host, const double *std3_host, float *y,
                                   float *xn_scratch, void *amax_out, void *stream)
{
    SPA_ARG(amax_out);
    return stem_impl(ctx, x, B, H, W, w0, b0, w1, b1, mean3_host, std3_host, y, 2, xn_scratch, amax_out, stream);
}

// the same for DRN-C (models/drn.py:134-170, 230-237): conv1 + bn1 + relu (= layer0) and the first convolution of layer1's
// BasicBlock (+ bn + relu) in one pass; y0 (B,H,W,16) also receives layer0's output, the block's residual
extern "C" int spa_drn_stem_c_amax(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W,
                                   const float *w0, const float *b0, const float *w1, const float *b1,
                                   const double *mean3_host, const double *std3_host, float *y, float *y0,
                                   float *xn_scratch, void *amax_out, void *stream)
{
    SPA_ARG(amax_out && y0 && ((uintptr_t)y0 & 15) == 0);
    return stem_impl(ctx, x, B, H, W, w0, b0, w1, b1, mean3_host, std3_host, y, 2, xn_scratch, amax_out, stream, y0);
}

// DRN-C in the bf16 network: y and y0 (B,H,W,16) bfloat16 — conv1's output and layer1's first convolution in one pass (round 6:
// the last library convolution of the bf16 arch-C forward)
extern "C" int spa_drn_stem_c_bf16(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W,
                                   const float *w0, const float *b0, const float *w1, const float *b1,
                                   const double *mean3_host, const double *std3_host, void *y, void *y0,
                                   float *xn_scratch, void *stream)
{
    SPA_ARG(y0 && ((uintptr_t)y0 & 15) == 0);
    return stem_impl(ctx, x, B, H, W, w0, b0, w1, b1, mean3_host, std3_host, y, 1, xn_scratch, nullptr, stream, (float *)y0);
}

static int stem_impl(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W,
                     const float *w0, const float *b0, const float *w1, const float *b1,
                     const double *mean3_host, const double *std3_host, void *y, int32_t out_dtype,
                     float *xn_scratch, void *amax_out, void *stream, float *y0)
{
    SPA_ARG(ctx && x && w0 && b0 && w1 && b1 && mean3_host && std3_host && y && B > 0 && H > 0 && W > 0);
    SPA_ARG(!y0 || out_dtype == 2 || out_dtype == 1);          // (out_dtype 1: y0 is bfloat16 as well)
    SPA_ARG(out_dtype == 0 || out_dtype == 1 || out_dtype == 2);      // 2: float32 output, 16-bit matrix cores (two planes)
    // exact input normalisation (models/drn.py:319-321) into a channels-last workspace, then the stem
    SpaProfScope prof_(ctx, PROF_DRN_STEM, spa_stream(stream));
    int rc = SPA_OK;
    const long long n_tiles = (long long)((W + ST_TW - 1) / ST_TW) * ((H + ST_TH - 1) / ST_TH) * B;
    SPA_ARG(n_tiles < (1ll << 31));
    SPA_ARG((long long)H * W * 12 < (1ll << 32));            // 32-bit byte offsets inside one image
    long long grid = 3ll * ctx->n_cu;                      // 3 resident workgroups per CU (LDS)
    if (grid > n_tiles) grid = n_tiles;
    if (out_dtype == 1) {
        // bf16 network: bf16 operands on the bf16 matrix cores (the weights are packed into fragment order
        // first); the normalised image is written once as bfloat16 (6 bytes per pixel)
        unsigned short *wp;
        if ((rc = spa_ws_reserve(ctx, WS_STEM_WPACK, (size_t)11 * 64 * 8 * 2, (void **)&wp)) != SPA_OK) return rc;
        hipLaunchKernelGGL(k_stem_pack_bf16, dim3(11), dim3(64), 0, spa_stream(stream), w0, w1, wp);
        unsigned short *xb = (unsigned short *)xn_scratch;
        if (!xb && (rc = spa_ws_reserve(ctx, WS_STEM_IN, (size_t)B * H * W * 3 * sizeof(float), (void **)&xb)) != SPA_OK) return rc;
        if ((rc = spa_drn_normalise(ctx, x, B, H, W, xb, 1, mean3_host, std3_host, stream)) != SPA_OK) return rc;
        hipLaunchKernelGGL(k_drn_stem_d_bf16, dim3((unsigned)grid), dim3(256), 0, spa_stream(stream), (const unsigned short *)xb, B,
                           H, W, (const unsigned short *)wp, b0, (const unsigned short *)(wp + 6 * 64 * 8), b1,
                           (unsigned short *)y, (unsigned short *)y0);
        SPA_LAUNCH_CHECK();
        return SPA_OK;
    }
    float *xn = xn_scratch;          // caller-owned scratch lets several calls run on different streams
    if (!xn) rc = spa_ws_reserve(ctx, WS_STEM_IN, (size_t)B * H * W * 3 * sizeof(float), (void **)&xn);
    if (rc != SPA_OK) return rc;
    rc = spa_drn_normalise(ctx, x, B, H, W, xn, 0, mean3_host, std3_host, stream);
    if (rc != SPA_OK) return rc;
    if (out_dtype == 2) {
        if (amax_out) spa_zero_word(amax_out, spa_stream(stream));
        unsigned short *wp;
        if ((rc = spa_ws_reserve(ctx, WS_STEM_WPACK, (size_t)22 * 64 * 8 * 2 + 16, (void **)&wp)) != SPA_OK) return rc;
        hipLaunchKernelGGL(k_stem_pack_f16, dim3(23), dim3(64), 0, spa_stream(stream), w0, b0, w1, wp);
        long long g2 = 2ll * ctx->n_cu;                    // two resident workgroups per CU (78 KB of LDS)
        if (g2 > n_tiles) g2 = n_tiles;
        hipLaunchKernelGGL(k_drn_stem_d_f16x3, dim3((unsigned)g2), dim3(256), 0, spa_stream(stream), (const float *)xn, B, H, W,
                           (const unsigned short *)wp, b0, b1, (float *)y, (unsigned *)amax_out, y0);
        SPA_LAUNCH_CHECK();
        return SPA_OK;
    }
    hipLaunchKernelGGL(k_drn_stem_d, dim3((unsigned)grid), dim3(256), 0, spa_stream(stream), (const float *)xn, B, H, W,
                       w0, b0, w1, b1, y, (int)out_dtype);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

