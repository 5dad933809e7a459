/*
 * spalign.h — C ABI of libspalign.so, the MI355X (gfx950) implementation of the
 * superpixel-align label-generation hot path.
 *
 * The reference (pfnet-research/superpixel-align) has no native boundary: its hot path is
 * Python calling CuPy/NumPy/scikit-image.  The entry points below are what a binding for
 * that path would call in place of the reference's Python bodies; each cites the reference
 * code it replaces (file:line in the reference repository).  INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns an int status: SPA_OK (0) or a negative SPA_ERR_*;
 *     spa_last_error() returns a thread-local message for the last failure.
 *   - no exceptions cross the boundary, no ownership is transferred: every array is
 *     allocated by the caller (device memory unless the name ends in _host); the library
 *     allocates only workspaces owned by the opaque spa_ctx.
 *   - `stream` is a hipStream_t passed as void*; all device entry points are asynchronous
 *     with respect to the host and ordered on that stream.
 *   - a spa_ctx is bound to one device and is not thread safe; distinct contexts are
 *     independent (one per GPU / process).
 *   - data-dependent failures detected on the device (a SLIC seed that lost all its pixels,
 *     a pixel outside every search window, ...) are latched in device status words, read
 *     back with spa_status() at a point where the caller synchronises anyway.
 *   - superpixels of a batch are laid out image after image: image b owns the descriptor
 *     rows [offsets[b], offsets[b+1]); N = offsets[B].
 */
#ifndef SPALIGN_H
#define SPALIGN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPA_OK 0
#define SPA_ERR_ARG (-1)      /* invalid argument                                       */
#define SPA_ERR_HIP (-2)      /* a HIP runtime call failed                              */
#define SPA_ERR_NOGPU (-3)    /* no usable gfx950 device                                */
#define SPA_ERR_LAYOUT (-4)   /* unsupported memory layout (e.g. feature map not NHWC)  */
#define SPA_ERR_CAPACITY (-5) /* caller-provided capacity too small                     */

/* bits of the device status word (spa_status) */
#define SPA_ST_SLIC_EMPTY_SEGMENT 0x01u  /* informational: a seed lost all pixels; NaN centre, dead afterwards (as skimage) */
#define SPA_ST_SLIC_UNCOVERED 0x02u      /* a pixel fell outside every 2S search window         */
#define SPA_ST_CONN_OVERSIZE 0x04u       /* a component reached max_size: handled by the slow exact path */
#define SPA_ST_POOL_SLOT_OVERFLOW 0x08u  /* retired: a feature pixel touched by > SPA_CELL_SLOTS superpixels is handled (its
                                            weights are recomputed on demand), the bit is never set */
#define SPA_ST_KMEANS_BARRIER 0x10u      /* grid barrier timed out (should never happen)        */
#define SPA_ST_LABEL_RANGE 0x20u         /* a label outside [0, S) was met                      */
#define SPA_ST_RNG_UNDERRUN 0x40u        /* the device random stream ran dry (spa_pyrandom_dev_generate too small) */
#define SPA_ST_NP_SHUFFLE 0x80u          /* spa_np_kmeans_init_dev: more points at or below the median weight than its LDS vector holds */

#define SPA_CELL_SLOTS 16

typedef struct spa_ctx spa_ctx;

/* ---- context ------------------------------------------------------------------------- */
int spa_version(void);
const char *spa_last_error(void);
/* device < 0: current device.  Fails with SPA_ERR_NOGPU when no gfx950 GPU is visible:
   there is no CPU fallback behind this ABI. */
int spa_ctx_create(int device, spa_ctx **out);
void spa_ctx_destroy(spa_ctx *ctx);
/* copies the latched status bits to *status_host and clears them; synchronises `stream`. */
int spa_status(spa_ctx *ctx, uint32_t *status_host, void *stream);
/* asynchronous variant for batch loops that never wait for the batch they have just enqueued: copies the
   latched bits to *status_pinned (pinned host memory) in stream order, clears nothing, synchronises nothing —
   the caller reads the word once an event recorded behind the call has completed.  (The reference has no
   counterpart: its per-image exceptions surface synchronously, batch_spalign_kmeans.py:538-548.) */
int spa_status_peek_async(spa_ctx *ctx, uint32_t *status_pinned, void *stream);
/* read AND clear as one atomic exchange on the device, in stream order, then the copy of the taken word to *status_pinned: the
   word holds the bits raised since the previous take — what a batch loop wants (an error is reported once, by the batch
   that raised it; informational bits of one batch do not stick to the next).  One exchange rather than copy + clear: the
   loops run the next batch's forward on another stream meanwhile, and a bit raised between a copy and a clear would be
   lost.  Up to 16 takes may be in flight. */
int spa_status_take_async(spa_ctx *ctx, uint32_t *status_pinned, void *stream);

/* Per-kernel timing for the roofline report (bench.py): when enabled, HIP events are recorded on
   the launch stream around each kernel family; spa_prof_read synchronises the device and
   returns the summed duration and the number of launches of one slot since spa_prof_enable. */
int spa_prof_enable(spa_ctx *ctx, int on);
int spa_prof_slots(void);
const char *spa_prof_name(int slot);
int spa_prof_read(spa_ctx *ctx, int slot, double *total_ms_host, int *launches_host);
/* Diagnostics only: copy `bytes` of internal workspace `which` (byte offset; -1 = the stamp buffer of diagnostic kernel builds) to the host. */
int spa_debug_peek(spa_ctx *ctx, int which, size_t offset, size_t bytes, void *host);
/* How many times a workspace of the context has been re-allocated (they grow with the largest shape seen).  A captured HIP graph holds
   the addresses of the workspaces its launches used: a replay is valid only while this number is what it was at capture time
   (drn.py keys its graph cache with it). */
int spa_ws_generation(spa_ctx *ctx);
/* Diagnostics only: kernel selection switches for A/B runs inside one process (tests/test_gpu_conv.py).  key 1: the narrow
   split-plane 3x3 layers (Cout 64 / 128) on the planes-in-LDS kernel (value 1, the default; environment SPA_CONVP at context
   creation) or on the round-3 kernel they replaced (value 0) — bit-identical outputs either way.  key 2 (libraries built with
   EXTRA=-DSPA_DIAG only): k_slic_assign's shelved LDS-table variant, the reproducer of the co-residency miscompare (DESIGN.md section 7,
   tools/race_probe8.py) — never use it for results.  No counterpart in the reference. */
int spa_debug_set(spa_ctx *ctx, int32_t key, int32_t value);
/* diagnostics (tools/lds_probe.py): an LDS table filled and read back per lane, n_wg workgroups of 256 threads, out[n_wg][256][4];
   mode 0: 16-byte reads, 1: 4-byte reads, 2: broadcast reads.  No counterpart in the reference. */
int spa_debug_lds_probe(spa_ctx *ctx, float *out, int32_t n_wg, int32_t steps, int32_t mode, void *stream);

/* ---- DRN forward glue (the convolutions themselves stay in PyTorch-ROCm / MIOpen) -----------
 * spa_drn_normalise: DRN.batch_predict input arithmetic (models/drn.py:319-321): x/255 in float32,
 *   then (x - mean) and (x / std) in float64 rounded to float32.  x (B,3,H,W) float32 planar ->
 *   out (B,H,W,3) channels-last, out_dtype 0 = float32, 1 = bfloat16.
 * spa_bias_act: y = relu?(y + bias [+ residual]) in place on a channels-last activation of `rows`
 *   pixels x C channels (dtype 0 = float32, 1 = bfloat16; bias/residual in the same dtype): the
 *   bias add left by folding BatchNorm, the residual add of a BasicBlock and the ReLU in one pass. */
int spa_drn_normalise(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, void *out,
                      int32_t out_dtype, const double *mean3_host, const double *std3_host, void *stream);
int spa_bias_act(spa_ctx *ctx, void *y, int32_t dtype, int64_t rows, int32_t C, const void *bias,
                 const void *residual, int32_t relu, void *stream);
/* float32 only: the same pass, and amax[0] (device word) = bit pattern of the largest magnitude stored — what spa_amax_f32
 * would return for y, without the extra pass (the scale input of the split-plane convolutions below). */
int spa_bias_act_amax(spa_ctx *ctx, float *y, int64_t rows, int32_t C, const float *bias,
                      const float *residual, int32_t relu, void *amax, void *stream);

/* The full-resolution stem of DRN-D as one float32-MFMA kernel: input normalisation
 * (models/drn.py:319-321), layer0 = conv7x7(3->16)+BN+ReLU and layer1 = conv3x3(16->16)+BN+ReLU
 * (models/drn.py:134-145) with BatchNorm folded into weights and biases.
 * x (B,3,H,W) float32 planar 0..255; w0 (16,147) = the (16,3,7,7) weight flattened; w1 (16,144) =
 * the (16,16,3,3) weight permuted to (n, ky, kx, c); b0, b1 (16); y (B,H,W,16) channels-last,
 * out_dtype 0 = float32, 1 = bfloat16 (the arithmetic is float32 either way).  xn_scratch: B*H*W*3
 * floats for the normalised image, or NULL to use the context's workspace (then one call at a time). */
int spa_drn_stem_d(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W,
                   const float *w0, const float *b0, const float *w1, const float *b1,
                   const double *mean3_host, const double *std3_host, void *y, int32_t out_dtype,
                   float *xn_scratch, void *stream);

/* The heavy 3x3 (dilated) convolutions of the DRN (models/drn.py:230-285, layers 5-8: 256/512 channels at 1/8
 * resolution) as a bfloat16 implicit GEMM on the matrix cores with the folded-BatchNorm bias, the BasicBlock's
 * residual add (models/drn.py:23-57) and the ReLU fused into the epilogue.  stride 1, padding = dilation.
 * x (B,H,W,Cin) bfloat16 channels-last; wt (Cout,9,Cin) bfloat16 = the (Cout,Cin,3,3) weight permuted to
 * (n, ky, kx, c); bias (Cout) float32; residual (B,H,W,Cout) bfloat16 or NULL; y (B,H,W,Cout) bfloat16.
 * Cin % 64 == 0, Cout % 64 == 0 (channel tiles of 256, 128 or 64), every pointer 16-byte aligned; float32
 * accumulation, one rounding to bfloat16 at the end. */
int spa_conv3x3_bf16(spa_ctx *ctx, const void *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                     const void *wt, int32_t Cout, const float *bias, const void *residual,
                     int32_t relu, int32_t dilation, void *y, void *stream);

/* The LIGHT layers of the bfloat16 DRN (models/drn.py:134-151 layer 2 of arch D and the 16 / 32-channel blocks of arch C,
 * :195-203 the stride-2 openers of layers 3 / 4 with their 1x1 stride-2 projections, the 1x1 projections of layers 5 / 6):
 * every convolution spa_conv3x3_bf16 does not take, so that the bfloat16 network has no library convolution left.
 * x (B,Hi,Wi,Cin) bfloat16 channels-last; wt (Cout,taps,Cin) bfloat16 (taps = 9: the 3x3 weight permuted to (n, ky, kx, c),
 * padding = dilation; taps = 1: 1x1, no padding); stride 1 or 2, output ((Hi + stride - 1) / stride, (Wi + stride - 1) /
 * stride); bias (Cout) float32; residual (B,Ho,Wo,Cout) bfloat16 or NULL; y (B,Ho,Wo,Cout) bfloat16.  Cin 16, 32 or 64
 * (3x3 and 1x1) or 128, 256 (1x1); Cout % 64 == 0 (Cin >= 32), % 32 == 0 (Cin 16 / 32) or % 16 == 0 (Cin 16); every
 * pointer 16-byte aligned; float32 accumulation, one rounding to bfloat16 at the end. */
int spa_conv_bf16_light(spa_ctx *ctx, const void *x, int32_t B, int32_t Hi, int32_t Wi, int32_t Cin, const void *wt,
                        int32_t taps, int32_t stride, int32_t Cout, const float *bias, const void *residual,
                        int32_t relu, int32_t dilation, void *y, void *stream);

/* The same layers of the float32 network on the float32 matrix cores (v_mfma_f32_16x16x4_f32), epilogue fused: every
 * stride-1 3x3 (dilated) convolution from 64 channels up (models/drn.py:230-285, the BasicBlocks' conv1 / conv2 and
 * the plain layers 7-8).  x (B,H,W,Cin) float32 channels-last; wt (Cout,9,Cin) float32 = the (Cout,Cin,3,3) weight
 * permuted to (n, ky, kx, c); bias (Cout) float32; residual (B,H,W,Cout) float32 or NULL; y (B,H,W,Cout) float32.
 * Cin % 32 == 0, Cout % 64 == 0, dilation <= 4, every pointer 16-byte aligned.  Float32 products and sums (an fmaf
 * chain per output): agrees with a float32 convolution to summation-order rounding. */
int spa_conv3x3_f32(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                    const float *wt, int32_t Cout, const float *bias, const float *residual,
                    int32_t relu, int32_t dilation, float *y, void *stream);
/* the 1x1 stride-1 projection of a BasicBlock whose channel count changes (models/drn.py:195-203 `downsample`,
 * layers 5 and 6): wt (Cout,Cin) float32; otherwise as spa_conv3x3_f32. */
int spa_conv1x1_f32(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                    const float *wt, int32_t Cout, const float *bias, const float *residual,
                    int32_t relu, float *y, void *stream);

/* The heaviest of those layers by the minimal filtering algorithm F(2x2,3x3) (Winograd): 2.25x fewer float32
 * multiplications than the direct form, float32 transforms and products — as close to a float64 convolution as
 * the direct float32 one (spa_wino.hip).  u (16,Cout,Cin) float32 = (G g G^T) of every (output, input) channel
 * pair, position-major; v_scratch / m_scratch: 16 * spa_wino_tiles(B,H,W,dilation) * Cin resp. Cout floats owned by
 * the caller; the rest as spa_conv3x3_f32 (any dilation >= 1). */
int64_t spa_wino_tiles(int32_t B, int32_t H, int32_t W, int32_t dilation);
int spa_conv3x3_wino_f32(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                         const float *u, int32_t Cout, const float *bias, const float *residual,
                         int32_t relu, int32_t dilation, float *v_scratch, float *m_scratch, float *y,
                         void *stream);

/* F(4x4,3x3): 4x fewer multiplications than the direct form, V / M 2.25x the activations.  Interpolation points
 * 0, 1, -1, 1/2, -2, infinity (spa_wino.hip gives the matrices and the measured float32 accuracy: through DRN-D-22 the
 * final map is as close to the float64 network as with direct float32 convolutions).  u (36,Cout,Cin) = (G g G^T)[6i+j];
 * scratch of 36 * spa_wino4_tiles(...) rows; the rest as spa_conv3x3_wino_f32. */
int64_t spa_wino4_tiles(int32_t B, int32_t H, int32_t W, int32_t dilation);
int spa_conv3x3_wino4_f32(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                          const float *u, int32_t Cout, const float *bias, const float *residual,
                          int32_t relu, int32_t dilation, float *v_scratch, float *m_scratch, float *y,
                          void *stream);

/* F(4x4,3x3) with its 36 GEMMs on the 16-bit matrix cores at float32 accuracy (spa_gemm16.hip): every float32 operand,
 * after an exact power-of-two scaling, is two half-precision planes h = rn(x), l = rn(x - h) (22 significand bits) and a
 * product is three matrix instructions ah.bh + ah.bl + al.bh accumulated in float32; through DRN-D-22 the final map is as
 * close to the float64 network as with float32 operands (tools/wino_network_error.py).
 *   u2      (36, Cout, Cin/32, 2, 32) half precision: the planes of t_ij * (G g G^T)[6i+j], t_ij a power of two
 *   cs      36 floats (host): 2^(p_i + p_j) / t_ij, p = 4 4 4 3 3 4 (the input transform's growth per row)
 *   amax_in device word: bit pattern of a bound on max |x| (spa_amax_f32, or the amax_out of the call that produced x)
 *   amax_out device word that receives that bound for y, or NULL
 * v_scratch 36 * spa_wino4_tiles(...) * Cin * 4 bytes, m_scratch 36 * spa_wino4_tiles(...) * Cout floats; Cout % 128 == 0;
 * the rest as spa_conv3x3_wino4_f32. */
int spa_amax_f32(spa_ctx *ctx, const float *x, int64_t n, void *amax, void *stream);
int spa_conv3x3_wino4_f16s(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                           const void *u2, const float *cs, int32_t Cout, const float *bias, const float *residual,
                           int32_t relu, int32_t dilation, const void *amax_in, void *amax_out, void *v_scratch,
                           float *m_scratch, float *y, void *stream);

/* spa_conv3x3_f32 / spa_conv1x1_f32 on the 16-bit matrix cores at float32 accuracy (the direct form of the scheme above,
 * for the 64-channel layers and the 1x1 projections): wt2 (Cout, taps, Cin/32, 2, 32) half precision = the planes of t * wt,
 * inv_t = 1 / t (t a power of two); amax_in / amax_out as in spa_conv3x3_wino4_f16s. */
int spa_conv3x3_f16s(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                     const void *wt2, float inv_t, int32_t Cout, const float *bias, const float *residual,
                     int32_t relu, int32_t dilation, const void *amax_in, void *amax_out, float *y, void *stream);
int spa_conv1x1_f16s(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                     const void *wt2, float inv_t, int32_t Cout, const float *bias, const float *residual,
                     int32_t relu, const void *amax_in, void *amax_out, float *y, void *stream);

/* the stride-2 3x3 convolution that opens layers 3 and 4 (models/drn.py:204-206; padding 1), on the 16-bit matrix cores at
 * float32 accuracy, together with the block's 1x1 stride-2 projection (models/drn.py:195-203) as output channels
 * [csplit, Cout) — wt2 rows csplit.. carry the projection's weights at the centre tap and zeros elsewhere — so the input is
 * read once: y (B,Ho,Wo,csplit) = relu?(conv + bias), y2 (B,Ho,Wo,Cout-csplit) = projection + bias (or y2 NULL and
 * csplit = Cout); Ho = (Hi+1)/2, Wo = (Wi+1)/2; amax_out tracks y.  Cout % 128 == 0, csplit % 64 == 0, Cin % 32 == 0. */
int spa_conv3x3_s2_f16s(spa_ctx *ctx, const float *x, int32_t B, int32_t Hi, int32_t Wi, int32_t Cin,
                        const void *wt2, float inv_t, int32_t Cout, int32_t csplit, const float *bias,
                        int32_t relu, const void *amax_in, void *amax_out, float *y, float *y2, void *stream);
/* the same opener (+ projection) with float32 matrix instructions, for the strict float32 network (`--fp32_mfma_gemm`): wt (Cout,9,Cin)
   float32, the projection's rows holding its weights at tap 4; no scales, no tracked maximum.  models/drn.py:195-206. */
int spa_conv3x3_s2_f32(spa_ctx *ctx, const float *x, int32_t B, int32_t Hi, int32_t Wi, int32_t Cin,
                       const float *wt, int32_t Cout, int32_t csplit, const float *bias,
                       int32_t relu, float *y, float *y2, void *stream);

/* spa_drn_stem_d with out_dtype 2 that also records the largest value it stores (device word amax_out), and layer 2 of
 * DRN-D (models/drn.py:134-145: conv3x3 16 -> 32, stride 2, padding 1, BN folded, ReLU) on the 16-bit matrix cores at float32
 * accuracy: x (B,H,W,16) float32 channels-last -> y (B,(H+1)/2,(W+1)/2,32); wp = the A fragments of the two planes of t * w,
 * [2 channel tiles][2 planes][5 steps][64 lanes] x 8 half-precision numbers (Engine.layer2_planes), inv_t = 1 / t. */
int spa_drn_stem_d_amax(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W,
                        const float *w0, const float *b0, const float *w1, const float *b1,
                        const double *mean3_host, const double *std3_host, float *y,
                        float *xn_scratch, void *amax_out, void *stream);
/* DRN-C's stem (models/drn.py:134-170, 230-237): spa_drn_stem_d_amax that also stores layer0's output y0 (B,H,W,16) — the
 * residual of layer1's BasicBlock, whose first convolution is the kernel's second stage. */
int spa_drn_stem_c_amax(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W,
                        const float *w0, const float *b0, const float *w1, const float *b1,
                        const double *mean3_host, const double *std3_host, float *y, float *y0,
                        float *xn_scratch, void *amax_out, void *stream);
/* the same for the bfloat16 network: y and y0 (B,H,W,16) bfloat16 (float32 arithmetic inside, bf16 matrix cores) */
int spa_drn_stem_c_bf16(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W,
                        const float *w0, const float *b0, const float *w1, const float *b1,
                        const double *mean3_host, const double *std3_host, void *y, void *y0,
                        float *xn_scratch, void *stream);
/* the thin 3x3 convolutions at the top of DRN-C (layer1's second convolution 16 -> 16, layer2's BasicBlock 16 -> 32 stride 2
 * with its 1x1 stride-2 projection, 32 -> 32) on the 16-bit matrix cores at float32 accuracy (csrc/spa_convs.hip):
 * x (B,H,W,Cin) float32 channels-last, Cin 16 or 32 -> y (B,Ho,Wo,Cout) = relu?(conv3x3(x; stride, padding 1) + bias [+ residual]),
 * Cout 16 or 32, stride 1 or 2, Ho = (H + stride - 1) / stride.  n_proj 0 or 32 (stride 2, Cin 16, Cout 32): the block's 1x1
 * stride-2 projection as y2 (B,Ho,Wo,32) = its convolution + bias[Cout ..], no ReLU — one pass over x, two outputs.
 * wp: the A fragments of the two planes of t * w, [Cout/16][2 planes][steps][64 lanes] x 8 half-precision numbers (steps = 5 of
 * two taps x 16 channels, or 9 of one tap x 32 channels), then [n_proj/16][2][64] x 8 for the projection (Engine.small_planes);
 * inv_t = 1 / t; bias Cout + n_proj floats; amax_in / amax_out as in spa_conv3x3_wino4_f16s (amax_out tracks y). */
int spa_conv_small_f16s(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, int32_t Cin, const void *wp,
                        float inv_t, int32_t Cout, int32_t stride, int32_t n_proj, const float *bias,
                        const float *residual, int32_t relu, const void *amax_in, void *amax_out, float *y, float *y2,
                        void *stream);
int spa_drn_layer2_f16s(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, const void *wp, float inv_t,
                        const float *bias, const void *amax_in, void *amax_out, float *y, void *stream);
/* layer 2 of DRN-D (models/drn.py:134-145) in plain float32 (fmaf chains in (tap, channel) order) for the strict float32 network:
   x (B,H,W,16) float32 channels-last, w9 (9,16,32) float32 = (ky*3+kx, input channel, output channel), y (B,(H+1)/2,(W+1)/2,32). */
int spa_drn_layer2_f32(spa_ctx *ctx, const float *x, int32_t B, int32_t H, int32_t W, const float *w9,
                       const float *bias, float *y, void *stream);

/* ---- input stage ---------------------------------------------------------------------------
 * replaces the host resize of ResizeImageDataset.get_example (datasets/resize_image_dataset.py:31-34:
 * chainercv.transforms.resize(image, resize_shape, 3)) as Pillow computes it on an 8-bit image, channel by
 * channel: PIL.Image.fromarray(ch).resize((w, h), PIL.Image.BICUBIC) — bit exact (integer arithmetic).
 * src (B,H,W,C) uint8 interleaved (a decoded PNG) -> out (B,C,dst_h,dst_w) float32 planar, values 0..255.
 * With dst == src size only the layout/dtype change is made. */
int spa_resize_bicubic_u8(spa_ctx *ctx, const uint8_t *src, int32_t B, int32_t H, int32_t W, int32_t C,
                          int32_t dst_h, int32_t dst_w, float *out, void *stream);
/* the OpenCV branch of the same resize (chainercv.transforms.resize with cv2 importable: cv2.resize(uint8 HWC image, (w, h),
 * interpolation=cv2.INTER_CUBIC) — the datasets resize the decoded uint8 image and convert to float32 afterwards,
 * datasets/resize_image_dataset.py:20-36, datasets/zipped_cityscapes_road_dataset.py:78-85), which is what the reference
 * environment ran — NOT PINNED (no cv2 in the build image, no fixture in the reference): OpenCV's published 8-bit scalar
 * algorithm (A = -0.75 taps as shorts of 1/2048, int32 sums, (v + 2^21) >> 22 saturated to 0..255, border replication),
 * bit-identical to the restatement oracle/resize_oracle.c:orc_resize_cvcubic_u8.  out holds the resized bytes as float32.
 * Same arguments as spa_resize_bicubic_u8. */
int spa_resize_cvcubic_u8(spa_ctx *ctx, const uint8_t *src, int32_t B, int32_t H, int32_t W, int32_t C,
                          int32_t dst_h, int32_t dst_w, float *out, void *stream);

/* ---- SLIC superpixels ------------------------------------------------------------------
 * replaces batch_superpixel(), SLIC branch: batch_spalign_kmeans.py:308-311, i.e.
 * skimage.segmentation.slic(img.transpose(1,2,0), n_segments) with every other argument
 * at its default (compactness 10, max_iter 10, sigma 0, enforce_connectivity True).        */

/* Host-side plan (skimage.util.regular_grid + slic_superpixels.py:322-327).               */
typedef struct {
    int32_t n_centroids;        /* seeds actually placed (<= n_segments)                   */
    int32_t grid_ny, grid_nx;   /* seed grid                                               */
    int32_t start_y, start_x;   /* first seed                                              */
    int32_t step_y, step_x;     /* seed spacing                                            */
    int32_t win_step_y, win_step_x; /* spacing used for the 2S search window (recomputed
                                   from n_centroids, as the Cython core does)              */
    float step;                 /* max spacing: spatial weight = 1/step^2                  */
    int32_t min_size, max_size; /* connectivity thresholds                                 */
    int32_t max_labels;         /* upper bound of labels after the connectivity pass       */
} spa_slic_plan;
int spa_slic_make_plan(int32_t H, int32_t W, int32_t n_segments, spa_slic_plan *plan_host);

/* rgb (B,3,H,W) float32, values 0..255, NOT normalised (the reference passes the raw image)
   -> CIE Lab (D65/2deg) * ratio, planar (B,3,H,W) float32.  skimage rgb2lab, float32.     */
int spa_rgb2lab(spa_ctx *ctx, const float *rgb, int32_t B, int32_t H, int32_t W,
                float ratio, float *lab, void *stream);

/* _slic_cython: max_iter Lloyd sweeps on a scaled Lab image.
   labels (B,H,W) int32 out; centres (B, n_centroids, 6) float32 out or NULL (z,y,x,L,a,b). */
int spa_slic_core(spa_ctx *ctx, const float *lab, int32_t B, int32_t H, int32_t W,
                  int32_t n_segments, int32_t max_iter, int32_t *labels, float *centres,
                  void *stream);

/* _enforce_label_connectivity_cython: scan-order relabelling, components < min_size merged
   into the last labelled neighbour met by the breadth-first search, growth capped at
   max_size.  labels_out (B,H,W) int32, n_labels (B) int32 (device).                       */
int spa_enforce_connectivity(spa_ctx *ctx, const int32_t *labels_in, int32_t B, int32_t H,
                             int32_t W, int32_t min_size, int32_t max_size,
                             int32_t *labels_out, int32_t *n_labels, void *stream);

/* ---- the float64 instantiation: superpixel_overlaps.py:301-304 calls slic(uint8 image, n_segments), and
 * scikit-image then runs rgb2lab and _slic_cython in float64 (slic_superpixels.py: dtype = image.dtype).
 * rgb (B,3,H,W) float32 holding the uint8 values 0..255 (anything else latches SPA_ST_LABEL_RANGE).      */
int spa_rgb2lab_u8_f64(spa_ctx *ctx, const float *rgb, int32_t B, int32_t H, int32_t W, double ratio,
                       double *lab, void *stream);            /* lab (B,3,H,W) float64 planar, x ratio */
int spa_slic_core_f64(spa_ctx *ctx, const double *lab, int32_t B, int32_t H, int32_t W,
                      int32_t n_segments, int32_t max_iter, int32_t *labels, double *centres,
                      void *stream);                          /* centres (B, n_centroids, 6) float64 or NULL */
int spa_slic_u8(spa_ctx *ctx, const float *rgb, int32_t B, int32_t H, int32_t W, int32_t n_segments,
                double compactness, int32_t max_iter, int32_t *labels, int32_t *n_labels, void *stream);

/* whole slic() call: rgb -> labels (B,H,W) int32 contiguous ids 0..n_labels[b]-1.         */
int spa_slic(spa_ctx *ctx, const float *rgb, int32_t B, int32_t H, int32_t W,
             int32_t n_segments, float compactness, int32_t max_iter,
             int32_t *labels, int32_t *n_labels, void *stream);

/* ---- Felzenszwalb superpixels ------------------------------------------------------------
 * replaces batch_superpixel(), felzenszwalb branch: batch_spalign_kmeans.py:301-307, i.e.
 * skimage.segmentation.felzenszwalb(img.transpose(1,2,0) / 255., scale, sigma, min_size) — the
 * branch every reference launcher uses (scale 300, sigma 0.8, min_size 20).
 * rgb (B,3,H,W) float32 0..255 -> labels (B,H,W) int32 contiguous ids, n_labels (B) int32.
 * Equal edge costs are ordered by edge index (numpy's argsort leaves that order unspecified). */
int spa_felzenszwalb(spa_ctx *ctx, const float *rgb, int32_t B, int32_t H, int32_t W,
                     double scale, double sigma, int32_t min_size, int32_t *labels,
                     int32_t *n_labels, void *stream);

/* the same call on a uint8 image, as superpixel_overlaps.py:294-300 makes it: the values (still
 * passed as float32 0..255) are divided by 255. in float64, not float32.                      */
int spa_felzenszwalb_u8(spa_ctx *ctx, const float *rgb, int32_t B, int32_t H, int32_t W,
                        double scale, double sigma, int32_t min_size, int32_t *labels,
                        int32_t *n_labels, void *stream);

/* ---- per-superpixel descriptors -------------------------------------------------------- */

/* offsets (B+1) int32 = exclusive prefix sum of n_labels (B).  (n_superpixels_per_image,
   batch_spalign_kmeans.py:321, kept on the device)                                        */
int spa_segment_offsets(spa_ctx *ctx, const int32_t *n_labels, int32_t B, int32_t *offsets,
                        void *stream);

/* counts, bounding boxes, centre of mass and the location prior of every superpixel.
     count    (Ncap) int32
     centroid (Ncap,2) float64 (y, x) image-pixel units — scipy.ndimage.center_of_mass, :229
     prior    (Ncap) float64 — create_prior, :111-129 / batch_create_prior, :333-344
   Any of centroid / prior may be NULL.  Ncap >= offsets[B].                               */
int spa_segment_stats(spa_ctx *ctx, const int32_t *labels, int32_t B, int32_t H, int32_t W,
                      const int32_t *offsets, int32_t Ncap,
                      double y_rel_pos, double x_rel_pos, double y_rel_sigma, double x_rel_sigma,
                      int32_t *count, double *centroid, double *prior, void *stream);

/* Anchor selection, host half: CPython random.shuffle(pixel list)[:n_anchors] for every
   superpixel in order (:231-234) needs only the superpixel sizes.  `rng` is an MT19937
   state seeded like random.seed(); ranks_host (N, n_anchors) int32 receives, per anchor, the
   raster-order rank of the chosen pixel inside its superpixel; n_valid_host (N).          */
typedef struct spa_pyrandom spa_pyrandom;
int spa_pyrandom_create(uint64_t seed, spa_pyrandom **out);
void spa_pyrandom_destroy(spa_pyrandom *rng);
int spa_pyrandom_shuffle_select_host(spa_pyrandom *rng, const int32_t *count_host, int32_t N,
                                     int32_t n_anchors, int32_t *ranks_host,
                                     int32_t *n_valid_host);
/* numpy legacy global RandomState (np.random.seed / np.random.shuffle), used by the k > 2
   k-means initialisation (:147-149).                                                      */
typedef struct spa_nprandom spa_nprandom;
int spa_nprandom_create(uint32_t seed, spa_nprandom **out);
void spa_nprandom_destroy(spa_nprandom *rng);
int spa_nprandom_shuffle_host(spa_nprandom *rng, int64_t *a_host, int64_t n);
/* numpy's rk_state of the generator as it stands: 624 state words + the position inside the current block (628 words written,
   the last three zero) — what spa_np_kmeans_init_dev keeps in device memory. */
int spa_nprandom_state(spa_nprandom *rng, uint32_t *state628_host);
/* the inverse: continue from a state taken with spa_nprandom_state or downloaded from the device copy spa_np_kmeans_init_dev
   advances (a batch too large for the device initialisation draws on the host and hands the stream back). */
int spa_nprandom_set_state(spa_nprandom *rng, const uint32_t *state628_host);
/* The k > 2 initial assignment (:141-149) drawn ON THE DEVICE from numpy's stream (csrc/spa_nprng.hip): thr = sort(w)[N // 2],
   m = #(w <= thr), idx = arange(m) % (k - 1) + 1, np.random.shuffle(idx) -> init_other[0 .. m) for spa_kmeans_weighted.
   state_dev: 628 words of device memory holding the generator (advanced by the call); n_ptr: device word N; gate: device word
   or NULL — with a gate the launch runs, and consumes outputs, only if *gate > 0 (speculatively enqueued retry runs,
   batch_spalign_kmeans.py:201-205); m_out: device word or NULL.  Ncap <= 65536, 2 < k <= 8. */
int spa_np_kmeans_init_dev(spa_ctx *ctx, void *state_dev, const double *w, const int32_t *n_ptr, int32_t Ncap,
                           int32_t k, const int32_t *gate, int64_t *init_other, int32_t *m_out, void *stream);
/* the retry bookkeeping of one k-means run (:201-205) on the device: n_fail = images whose superpixels assign[offsets[b] ..
   offsets[b+1]) hold no 0.  gate == NULL (the batch's first run): *pending = n_fail, *made = 0; a gated retry run: nothing
   unless *gate > 0, then *pending += n_fail - 1, *made += 1.  fail_out: B bytes (1 = image without cluster 0) or NULL. */
int spa_kmeans_retry_update(spa_ctx *ctx, const int32_t *assign, const int32_t *offsets, int32_t B, const int32_t *gate,
                            int32_t *pending, int32_t *made, uint8_t *fail_out, void *stream);

/* The same selection entirely on the device (no superpixel size ever visits the host): a device-resident
   CPython generator (seeded like random.seed()), its outputs produced ahead of use into a ring
   (spa_pyrandom_dev_generate: `want` outputs available; asynchronous, data independent), then for the N =
   *n_ptr superpixels in order the rejection sampling of every shuffle swap and the first n_anchors places of
   every shuffled list.  total_pixels >= sum of the sizes.  ranks (Ncap, n_anchors), n_valid (Ncap) int32 out;
   SPA_ST_RNG_UNDERRUN is latched if the ring runs dry (about 1.4 outputs are used per pixel). */
int spa_pyrandom_dev_seed(spa_ctx *ctx, uint64_t seed, void *stream);
int spa_pyrandom_dev_generate(spa_ctx *ctx, int64_t want, void *stream);
int spa_anchor_ranks_dev(spa_ctx *ctx, const int32_t *count, const int32_t *n_ptr, int32_t Ncap,
                         int32_t n_anchors, int64_t total_pixels, int32_t *ranks, int32_t *n_valid,
                         void *stream);

/* Anchor selection, device half: rank -> pixel.  ranks (Ncap, n_anchors) int32,
   n_valid (Ncap) int32 -> anchors (Ncap, n_anchors, 2) int32 (y, x).                      */
int spa_select_anchor_pixels(spa_ctx *ctx, const int32_t *labels, int32_t B, int32_t H,
                             int32_t W, const int32_t *offsets, int32_t Ncap,
                             const int32_t *ranks, const int32_t *n_valid, int32_t n_anchors,
                             int32_t *anchors, void *stream);

/* Feature-map addressing: element (b, c, y, x) lives at
   fmap[b*stride_b + c*stride_c + y*stride_y + x*stride_x] (strides in elements).
   The kernels require channels-last storage (stride_c == 1): SPA_ERR_LAYOUT otherwise.
   fmap_dtype: 0 = float32, 1 = bfloat16.                                                  */
typedef struct {
    int32_t C, fh, fw;
    int64_t stride_b, stride_c, stride_y, stride_x;
    int32_t dtype;
} spa_fmap_desc;

/* Output descriptor matrix X: row s at X + s*ld; x_dtype 0 = float32, 1 = float64.
   Columns [0, C) receive the pooled features; with append_pos the centroid (y, x) goes to
   columns C, C+1 (hstack, :269-270).                                                      */

/* superpixel_align(), anchor mode (:210-276) given the selected anchors.                  */
int spa_pool_anchor(spa_ctx *ctx, const void *fmap, const spa_fmap_desc *desc,
                    int32_t B, int32_t img_h, const int32_t *offsets, int32_t Ncap,
                    const int32_t *anchors, const int32_t *n_valid, int32_t n_anchors,
                    int32_t n_neighbors, const double *centroid, int32_t append_pos,
                    void *X, int32_t x_dtype, int64_t ld, void *stream);

/* dense per-superpixel mean (mean mode, notebooks/Superpixel_Align.ipynb cell 4).
   sampling 0 = nearest, 1 = bilinear (corners aligned, chainer F.resize_images).          */
int spa_pool_mean(spa_ctx *ctx, const void *fmap, const spa_fmap_desc *desc,
                  const int32_t *labels, int32_t B, int32_t H, int32_t W,
                  const int32_t *offsets, int32_t Ncap, const int32_t *count,
                  int32_t sampling, const double *centroid, int32_t append_pos,
                  void *X, int32_t x_dtype, int64_t ld, void *stream);

/* ---- weighted k-means + painting -------------------------------------------------------
 * kmeans(), :136-183, on all superpixels of the batch at once.
 *   X (Ncap, D) row stride ld, x_dtype as above; w (Ncap) float64; n_ptr: device int32
 *   holding N (= offsets + B), so no host synchronisation is needed;
 *   init_other: device int64 array, the shuffled `idx` of :147-149 for the points with
 *   w <= threshold, or NULL for its unshuffled value arange(M) % (k-1) + 1 (exact for k = 2);
 *   assign (Ncap) int32 out; info (4) int32 out: {iterations, status, N, reserved},
 *   status 0 converged / 1 hit max_iter / 2 stopped on an empty cluster.                  */
int spa_kmeans_weighted(spa_ctx *ctx, const void *X, int32_t x_dtype, int64_t ld, int32_t D,
                        const double *w, const int32_t *n_ptr, int32_t Ncap, int32_t k,
                        int32_t max_iter, const int64_t *init_other, int32_t *assign,
                        int32_t *info, void *stream);
/* the same launch behind a device-side gate: nothing runs (assign / info untouched) unless *gate > 0 — the retry runs of
   weighted_kmeans (:201-205), enqueued speculatively by the pipeline and decided on the device.  gate == NULL: always runs. */
int spa_kmeans_weighted_gated(spa_ctx *ctx, const void *X, int32_t x_dtype, int64_t ld, int32_t D,
                              const double *w, const int32_t *n_ptr, int32_t Ncap, int32_t k,
                              int32_t max_iter, const int64_t *init_other, const int32_t *gate, int32_t *assign,
                              int32_t *info, void *stream);

/* weighted_kmeans() paint loop (:193-199) and `clustering_result == 0` (:207):
   cluster[p] = assign[offsets[b] + labels[p]], road[p] = cluster[p] == 0; (B,H,W) uint8.  */
int spa_paint(spa_ctx *ctx, const int32_t *labels, const int32_t *assign,
              const int32_t *offsets, int32_t B, int32_t H, int32_t W,
              uint8_t *cluster, uint8_t *road, void *stream);

/* replaces the refinement loop of superpixel_overlaps.py:353-361: refined[p] = 1 where the
 * superpixel of p holds more than `threshold` of the image's predicted road pixels
 * (overlap / float(n_road) > threshold in float64; all zeros when no road was predicted).
 * labels (B,npix) int32 in [0,max_labels), road (B,npix) uint8 -> refined (B,npix) uint8.     */
int spa_overlap_refine(spa_ctx *ctx, const int32_t *labels, const uint8_t *road, int32_t B,
                       int64_t npix, int32_t max_labels, double threshold, uint8_t *refined,
                       void *stream);

/* save_info() scoring (:398-405): per image confusion of road (B,npix) uint8 against
   gt (B,npix) int32 in {-1 ignore, 0, 1} -> out (B,4) int64 {TN, FP, FN, TP}.             */
int spa_confusion(spa_ctx *ctx, const uint8_t *road, const int32_t *gt, int32_t B,
                  int64_t npix, int64_t *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SPALIGN_H */
