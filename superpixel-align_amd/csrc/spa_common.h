// Internal header of libspalign.so (gfx950 only; no CUDA/portability layer by design).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/spalign.h"

#define SPA_WAVE 64

// ---------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------
void spa_set_error(const char *fmt, ...);

#define SPA_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            spa_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),      \
                          __FILE__, __LINE__);                                         \
            return SPA_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

#define SPA_ARG(cond)                                                                  \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            spa_set_error("invalid argument: %s (%s:%d)", #cond, __FILE__, __LINE__);  \
            return SPA_ERR_ARG;                                                        \
        }                                                                              \
    } while (0)

#define SPA_LAUNCH_CHECK()                                                             \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            spa_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_),   \
                          __FILE__, __LINE__);                                         \
            return SPA_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

// ---------------------------------------------------------------------------------------
// context: device binding + grow-on-demand workspaces (never freed between calls, so the
// steady state of a batch loop performs no hipMalloc)
// ---------------------------------------------------------------------------------------
enum {
    WS_LAB = 0,      // scaled Lab image (B,3,H,W) f32
    WS_CENTRES,      // SLIC centre table (B,nC,12) words
    WS_PRE,          // labels before the connectivity pass (B,H,W) i32
    WS_PARENT,       // union-find parents / component roots (B,H,W) i32
    WS_SIZE,         // component sizes, indexed by root pixel (B,H,W) i32
    WS_FINAL,        // final label of a component, indexed by root pixel (B,H,W) i32
    WS_CLAIM,        // BFS claim keys (B,H,W) u32
    WS_QUEUE,        // BFS queues (B,H,W) i32
    WS_BLK,          // per-block scan counters
    WS_SMALL,        // compacted list of small component roots (B,H*W/?) i32
    WS_CONNMISC,     // per-image counters of the connectivity pass
    WS_BBOX,         // per-superpixel bounding boxes (Ncap,4) i32
    WS_CELLSLOT,     // mean pooling: per feature pixel (label, weight) slots
    WS_KM_PART,      // k-means partial sums
    WS_KM_MISC,      // k-means centres, counters, barrier words
    WS_NLABELS,      // n_labels (B) for spa_slic
    WS_SBOX,         // bounding boxes of small components
    WS_TODO,         // BFS tier hand-over lists
    WS_ROWMASK,      // SLIC per (centre, row) occupancy bits
    WS_FINEMASK,     // SLIC per (centre, window row, 64-pixel piece) pixel masks
    WS_SLIC_ORDER,   // SLIC update: per-XCD longest-first segment lists + queue heads
    WS_OVERLAP,      // superpixel_overlaps baseline: road pixels per (image, superpixel)
    WS_STEM_IN,      // DRN-D stem: normalised channels-last input image
    WS_SRGB_LUT,     // sRGB companding of the 256 integer channel values
    WS_FZ_KEYS,      // felzenszwalb: edge cost keys (in/out of the radix sort)
    WS_FZ_VALS,      // felzenszwalb: edge indices (in/out of the radix sort)
    WS_FZ_STATE,     // felzenszwalb: internal costs + reservation marks
    WS_FZ_TMP,       // felzenszwalb: radix sort temporary storage
    WS_STEM_WPACK,   // bf16 stem: weights packed in MFMA fragment order
    WS_ZERO_LINE,    // convolution: zero line read for padding pixels
    WS_RESIZE_TAB,   // input stage: bicubic tap tables of the current (input, output) size
    WS_RESIZE_TMP,   // input stage: horizontally resized images (B,H,w,C) u8
    WS_RNG_STATE,    // device MT19937 (CPython `random`) state + ring positions
    WS_RNG_RING,     // ring of generated 32-bit outputs
    WS_RNG_JBUF,     // accepted swap partners of every shuffle (one int per pixel)
    WS_RNG_JOFF,     // offsets of the superpixels' swap lists
    WS_RUNS,         // connectivity: per-row lists of run starts (B,H,W) i32, used from the front of each row
    WS_DEBUG,        // diagnostic builds: in-kernel stamps (tools read it with spa_debug_peek)
    WS_COUNT
};

// optional per-kernel timing (spa_prof_*): HIP events recorded on the launch stream around the
// kernels bench.py prices against the roofline
enum {
    PROF_RGB2LAB = 0, PROF_SLIC_ASSIGN, PROF_SLIC_UPDATE, PROF_CONNECT, PROF_STATS,
    PROF_CELL_WEIGHTS, PROF_POOL_MEAN, PROF_POOL_ANCHOR, PROF_KMEANS, PROF_PAINT,
    PROF_DRN_STEM, PROF_DRN_BIAS_ACT, PROF_DRN_CONV, PROF_DRN_CONV32, PROF_DRN_GEMM32, PROF_DRN_GEMM32_N, PROF_WINO_IN, PROF_WINO_OUT, PROF_DRN_GEMM16, PROF_DRN_GEMM16_N, PROF_DRN_CONV16, PROF_DRN_CONV16_128, PROF_DRN_CONV16_256, PROF_DRN_CONV16_1X1, PROF_DRN_CONV16_FRONT, PROF_DRN_CONV_LIGHT, PROF_SLOTS
};

struct spa_ctx {
    int device;
    int n_cu;
    uint32_t *d_status;      // latched status bits (word 0; words 16-31: spa_status_take_async's ring)
    unsigned status_takes;
    void *ws[WS_COUNT];
    size_t ws_bytes[WS_COUNT];
    int prof_on;
    hipEvent_t *prof_ev[PROF_SLOTS];   // pairs (start, stop)
    int prof_cap[PROF_SLOTS], prof_used[PROF_SLOTS];
    // side streams for independent kernels of one call (fork/join with events; created on first use)
    hipStream_t aux[2];
    hipEvent_t ev_fork, ev_join[2];
    int aux_ready;
    // per-context (= per-device) one-time kernel attributes and cached occupancy answers
    int km_attr_done[2];
    int conn_attr_done;
    int conn_claim_ready;          // the BFS claim words are all-INF (set once per allocation)
    size_t conn_claim_bytes;
    int upd_wg_per_cu, upd_wg_per_cu8;
    int slic_force_general;      // SPA_SLIC_GENERAL=1 at context creation: spa_slic_core takes the general kernels
    int zero_line_ready, conv_attr_done, conv32_attr_done, fz_attr_done, gemm16_attr_done, gemm16s_attr_done, nprng_attr_done, conv_stag_attr_done;
    int rng_seeded;
    int ws_generation;             // counts workspace re-allocations (spa_ws_generation: captured graphs hold workspace addresses)
    int dbg_slic_ldsx;             // spa_debug_set key 2 (diagnostic builds: the reproducer variant of k_slic_assign)
    int convp_on;                  // spa_debug_set key 1 (spa_convp.hip instead of spa_conv32.hip's narrow tiles)
    int rs_key[4], rs_ks[2];       // bicubic tables held in WS_RESIZE_TAB: (H, W, h, w) and tap counts
};

#ifdef __HIPCC__
// Two float32 values scaled by a power of two and split into half-precision planes: h = rn16(x * sc), l = rn16(x * sc - h), each
// ONE mixed-precision fma (v_fma_mixlo/hi_f16: the product by a power of two and the difference are exact in float32, so the
// only rounding is the conversion — the same bits as multiply, convert, convert back, subtract, convert).  Returns h as the
// packed pair (x0 low half, x1 high half), l likewise through `l`.  Single-issue vector instructions: no packed float32
// operation (slow beside matrix instructions, MI355X_MICROARCH.md).
__device__ __forceinline__ unsigned spa_split16_pair(float x0, float x1, float sc, unsigned &l)
{
    unsigned h, lo;
    asm("v_fma_mixlo_f16 %0, %2, %4, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(lo) : "v"(x0), "v"(x1), "v"(sc));
    l = lo;
    return h;
}
#endif

int spa_aux_streams(spa_ctx *ctx);

void spa_prof_mark(spa_ctx *ctx, int slot, int end, hipStream_t s);
struct SpaProfScope {
    spa_ctx *c; int slot; hipStream_t s;
    SpaProfScope(spa_ctx *c_, int slot_, hipStream_t s_) : c(c_), slot(slot_), s(s_) { if (c->prof_on && slot >= 0) spa_prof_mark(c, slot, 0, s); }
    ~SpaProfScope() { if (c->prof_on && slot >= 0) spa_prof_mark(c, slot, 1, s); }      // slot < 0: part of an enclosing scope
};

int spa_ws_reserve(spa_ctx *ctx, int which, size_t bytes, void **out);
// one 32-bit word set to zero by a KERNEL on stream s (the tracked maxima of the DRN path).  Not hipMemsetAsync: as memset nodes
// of a captured graph (drn.py replays small forwards as HIP graphs) these 4-byte fills were observed to run out of order with
// the graph's kernels while a second stream was busy — a replay then scaled a layer by a maximum of 0 and returned zeros.
void spa_zero_word(void *word, hipStream_t s);

static inline hipStream_t spa_stream(void *s) { return (hipStream_t)s; }

// ---------------------------------------------------------------------------------------
// deterministic elementary functions (device).  Same definition as oracle/detmath.h: binary64
// evaluation with +,-,*,/ only (each correctly rounded, contraction off), one final rounding.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double spa_det_log_pos(double x)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __longlong_as_double((long long)b);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 1.0 / 25.0;
    p = p * z + 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    double lnm = 2.0 * s * p;
    double ed = (double)e;
    return ed * 6.93147180369123816490e-01 + (ed * 1.90821492927058770002e-10 + lnm);
}

__device__ __forceinline__ double spa_det_exp(double t)
{
    double kf = floor(t * 1.44269504088896338700e+00 + 0.5);
    double r = (t - kf * 6.93147180369123816490e-01) - kf * 1.90821492927058770002e-10;
    double p = 1.0 / 87178291200.0;
    p = p * r + 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    int k = (int)kf;
    unsigned long long b = (unsigned long long)(k + 1023) << 52;
    return p * __longlong_as_double((long long)b);
}

// ---------------------------------------------------------------------------------------
// wave helpers (wave64)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned spa_lane() { return __lane_id(); }
// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ unsigned spa_rank_in_mask(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
