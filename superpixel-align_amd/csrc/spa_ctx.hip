// Context, error state, workspaces and the host-side planning code of libspalign.so.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include "spa_common.h"

static thread_local char g_err[512] = "";

void spa_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *spa_last_error(void) { return g_err; }
extern "C" int spa_version(void) { return 100; }

extern "C" int spa_ctx_create(int device, spa_ctx **out)
{
    SPA_ARG(out != nullptr);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        spa_set_error("no HIP device visible: libspalign has no CPU fallback");
        return SPA_ERR_NOGPU;
    }
    if (device < 0) SPA_HIP(hipGetDevice(&device));
    SPA_ARG(device < n);
    SPA_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SPA_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        spa_set_error("device %d is %s; libspalign is built for gfx950 (MI355X) only", device,
                      prop.gcnArchName);
        return SPA_ERR_NOGPU;
    }
    spa_ctx *ctx = (spa_ctx *)calloc(1, sizeof(spa_ctx));
    ctx->device = device;
    ctx->n_cu = prop.multiProcessorCount;
    if (const char *e = getenv("SPA_SLIC_GENERAL")) ctx->slic_force_general = atoi(e) != 0;
    ctx->convp_on = getenv("SPA_CONVP") ? atoi(getenv("SPA_CONVP")) != 0 : 1;
    SPA_HIP(hipMalloc((void **)&ctx->d_status, 32 * sizeof(uint32_t)));        // [0] the bits, [16..31] ring of taken words
    SPA_HIP(hipMemset(ctx->d_status, 0, 32 * sizeof(uint32_t)));
    *out = ctx;
    return SPA_OK;
}

extern "C" void spa_ctx_destroy(spa_ctx *ctx)
{
    if (!ctx) return;
    for (int i = 0; i < PROF_SLOTS; ++i) {
        for (int j = 0; j < ctx->prof_cap[i]; ++j) (void)hipEventDestroy(ctx->prof_ev[i][j]);
        free(ctx->prof_ev[i]);
    }
    for (int i = 0; i < WS_COUNT; ++i)
        if (ctx->ws[i]) (void)hipFree(ctx->ws[i]);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->aux_ready) {
        for (int i = 0; i < 2; ++i) { (void)hipStreamDestroy(ctx->aux[i]); (void)hipEventDestroy(ctx->ev_join[i]); }
        (void)hipEventDestroy(ctx->ev_fork);
    }
    free(ctx);
}

__global__ void k_zero_word(unsigned *w) { *w = 0u; }
void spa_zero_word(void *word, hipStream_t s) { hipLaunchKernelGGL(k_zero_word, dim3(1), dim3(1), 0, s, (unsigned *)word); }

int spa_ws_reserve(spa_ctx *ctx, int which, size_t bytes, void **out)
{
    if (bytes == 0) bytes = 16;
    if (ctx->ws_bytes[which] < bytes) {
        // growing a workspace is a rare, synchronising event (first batch of a new shape)
        SPA_HIP(hipDeviceSynchronize());
        if (ctx->ws[which]) SPA_HIP(hipFree(ctx->ws[which]));
        ctx->ws[which] = nullptr;
        ctx->ws_bytes[which] = 0;
        size_t want = bytes + bytes / 8;
        SPA_HIP(hipMalloc(&ctx->ws[which], want));
        ctx->ws_bytes[which] = want;
        ++ctx->ws_generation;
    }
    *out = ctx->ws[which];
    return SPA_OK;
}

extern "C" int spa_status(spa_ctx *ctx, uint32_t *status_host, void *stream)
{
    SPA_ARG(ctx && status_host);
    hipStream_t s = spa_stream(stream);
    SPA_HIP(hipMemcpyAsync(status_host, ctx->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    SPA_HIP(hipMemsetAsync(ctx->d_status, 0, sizeof(uint32_t), s));
    SPA_HIP(hipStreamSynchronize(s));
    return SPA_OK;
}

extern "C" int spa_status_peek_async(spa_ctx *ctx, uint32_t *status_pinned, void *stream)
{
    SPA_ARG(ctx && status_pinned);
    SPA_HIP(hipMemcpyAsync(status_pinned, ctx->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, spa_stream(stream)));
    return SPA_OK;
}

// the same copy followed, in stream order, by the clear: every batch of an asynchronous loop reads ITS bits (an error
// once, by the batch that raised it; informational bits do not stick to later batches)
// (round 4: the batch loops run the next batch's DRN forward on another stream while this batch's tail finishes, so the read
// and the clear are ONE atomic exchange — a bit raised by the other stream between a copy and a clear would be lost)
__global__ void k_status_take(uint32_t *status, uint32_t *taken) { *taken = atomicExch(status, 0u); }

extern "C" int spa_status_take_async(spa_ctx *ctx, uint32_t *status_pinned, void *stream)
{
    SPA_ARG(ctx && status_pinned);
    uint32_t *slot = ctx->d_status + 16 + (ctx->status_takes++ & 15);           // 16 takes may be in flight
    hipLaunchKernelGGL(k_status_take, dim3(1), dim3(1), 0, spa_stream(stream), ctx->d_status, slot);
    SPA_HIP(hipMemcpyAsync(status_pinned, slot, sizeof(uint32_t), hipMemcpyDeviceToHost, spa_stream(stream)));
    return SPA_OK;
}

// ---------------------------------------------------------------------------------------
// skimage.util.regular_grid((1, H, W), n) — host integer/double logic of slic()
// (skimage/util/_regular_grid.py:61-83; call sites slic_superpixels.py:91 and inside the
// Cython core).  Returns 0 or -1 where the Python code raises.
// ---------------------------------------------------------------------------------------
static int regular_grid_1hw(int64_t H, int64_t W, int64_t n_points, int64_t start[3],
                            int64_t step[3], int has_step[3])
{
    int64_t shape[3] = {1, H, W};
    int order[3] = {0, 1, 2};
    for (int i = 1; i < 3; ++i) {   // stable argsort of three values
        int o = order[i], j = i - 1;
        while (j >= 0 && shape[order[j]] > shape[o]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = o;
    }
    double sorted[3];
    for (int i = 0; i < 3; ++i) sorted[i] = (double)shape[order[i]];
    double space = (double)H * (double)W;
    if (space <= (double)n_points) {
        for (int i = 0; i < 3; ++i) { start[i] = 0; step[i] = 1; has_step[i] = 0; }
        return 0;
    }
    double ss[3];
    ss[0] = ss[1] = ss[2] = pow(space / (double)n_points, 1.0 / 3.0);
    bool small = false;
    for (int i = 0; i < 3; ++i) small = small || (sorted[i] < ss[i]);
    if (small) {
        for (int dim = 0; dim < 3; ++dim) {
            ss[dim] = sorted[dim];
            double sp = 1.0;
            for (int j = dim + 1; j < 3; ++j) sp *= sorted[j];
            if (dim == 2) return -1;
            double v = pow(sp / (double)n_points, 1.0 / (double)(2 - dim));
            for (int j = dim + 1; j < 3; ++j) ss[j] = v;
            bool ok = true;
            for (int j = 0; j < 3; ++j) ok = ok && (sorted[j] >= ss[j]);
            if (ok) break;
        }
    }
    for (int p = 0; p < 3; ++p) {
        int d = order[p];
        start[d] = (int64_t)floor(ss[p] / 2.0);
        step[d] = (int64_t)nearbyint(ss[p]);
        has_step[d] = 1;
    }
    return 0;
}

static int64_t range_len(int64_t n, int64_t start, int64_t step)
{
    return start >= n ? 0 : (n - start + step - 1) / step;
}

extern "C" int spa_slic_make_plan(int32_t H, int32_t W, int32_t n_segments, spa_slic_plan *plan)
{
    SPA_ARG(plan && H > 0 && W > 0 && n_segments > 0);
    int64_t st[3], sp[3];
    int has[3];
    if (regular_grid_1hw(H, W, n_segments, st, sp, has) != 0) {
        spa_set_error("n_segments=%d too large for a %dx%d image", n_segments, H, W);
        return SPA_ERR_ARG;
    }
    int64_t ny = range_len(H, st[1], sp[1]), nx = range_len(W, st[2], sp[2]);
    SPA_ARG(ny > 0 && nx > 0);
    int64_t n = ny * nx;
    plan->n_centroids = (int32_t)n;
    plan->grid_ny = (int32_t)ny;
    plan->grid_nx = (int32_t)nx;
    plan->start_y = (int32_t)st[1];
    plan->start_x = (int32_t)st[2];
    plan->step_y = (int32_t)sp[1];
    plan->step_x = (int32_t)sp[2];
    double stmax = 1.0;   // float(s.step) if s.step is not None else 1.0; z step is 1
    if (has[1] && (double)sp[1] > stmax) stmax = (double)sp[1];
    if (has[2] && (double)sp[2] > stmax) stmax = (double)sp[2];
    plan->step = (float)stmax;
    int64_t st2[3], sp2[3];
    int has2[3];
    if (regular_grid_1hw(H, W, n, st2, sp2, has2) != 0) return SPA_ERR_ARG;
    plan->win_step_y = has2[1] ? (int32_t)sp2[1] : 1;
    plan->win_step_x = has2[2] ? (int32_t)sp2[2] : 1;
    double seg = ((double)H * (double)W) / (double)n;
    plan->min_size = (int32_t)(0.5 * seg);
    plan->max_size = (int32_t)(3.0 * seg);
    // every kept component has >= min_size pixels; min_size 0 keeps every component
    int64_t ml = plan->min_size > 0 ? ((int64_t)H * W) / plan->min_size + 1 : (int64_t)H * W;
    if (ml > (int64_t)H * W) ml = (int64_t)H * W;
    plan->max_labels = (int32_t)ml;
    return SPA_OK;
}

// ---------------------------------------------------------------------------------------
// per-kernel timing with HIP events on the launch stream
// ---------------------------------------------------------------------------------------
void spa_prof_mark(spa_ctx *ctx, int slot, int end, hipStream_t s)
{
    int idx = ctx->prof_used[slot] * 2 + end;
    if (idx >= ctx->prof_cap[slot]) {
        int ncap = ctx->prof_cap[slot] ? ctx->prof_cap[slot] * 2 : 256;
        ctx->prof_ev[slot] = (hipEvent_t *)realloc(ctx->prof_ev[slot], (size_t)ncap * sizeof(hipEvent_t));
        for (int j = ctx->prof_cap[slot]; j < ncap; ++j) (void)hipEventCreate(&ctx->prof_ev[slot][j]);
        ctx->prof_cap[slot] = ncap;
    }
    (void)hipEventRecord(ctx->prof_ev[slot][idx], s);
    if (end) ctx->prof_used[slot] += 1;
}

extern "C" int spa_prof_enable(spa_ctx *ctx, int on)
{
    SPA_ARG(ctx);
    ctx->prof_on = on ? 1 : 0;
    for (int i = 0; i < PROF_SLOTS; ++i) ctx->prof_used[i] = 0;
    return SPA_OK;
}

extern "C" int spa_prof_slots(void) { return PROF_SLOTS; }

extern "C" const char *spa_prof_name(int slot)
{
    static const char *names[PROF_SLOTS] = {"k_rgb2lab", "k_slic_assign", "k_slic_update",
        "connectivity(all)", "segment_stats(all)", "k_cell_weights", "k_pool_mean", "k_pool_anchor",
        "k_kmeans", "k_paint", "k_drn_stem_d(+normalise)", "k_bias_act(all)", "k_conv3x3_bf16(all)", "k_conv3x3_f32<taps 9>(all)", "k_conv3x3_f32<0, 256, 1, 256>", "k_conv3x3_f32<taps 1, narrow tiles>(all)", "k_wino_in", "k_wino_out", "k_gemm_f16x3_stag<256, 256>", "k_gemm_f16x3<128, 128>", "k_conv3x3_p16<64, 256> (split planes, 64-channel layers)", "k_conv3x3_p16<128, 128> (split planes, 128-channel layers)", "k_conv3x3_f32<split, 256-channel tile>", "k_conv3x3_f32<split, 1x1>", "split-plane front (stride-2 openers + projections, layer 2, DRN-C layers 1-2)", "k_conv_bf16_light(all)"};
    return (slot >= 0 && slot < PROF_SLOTS) ? names[slot] : "";
}

// total_ms / launches of one slot since spa_prof_enable; synchronises the device
extern "C" int spa_prof_read(spa_ctx *ctx, int slot, double *total_ms, int *launches)
{
    SPA_ARG(ctx && total_ms && launches && slot >= 0 && slot < PROF_SLOTS);
    SPA_HIP(hipDeviceSynchronize());
    double t = 0.0;
    for (int j = 0; j < ctx->prof_used[slot]; ++j) {
        float ms = 0.0f;
        SPA_HIP(hipEventElapsedTime(&ms, ctx->prof_ev[slot][2 * j], ctx->prof_ev[slot][2 * j + 1]));
        t += ms;
    }
    *total_ms = t;
    *launches = ctx->prof_used[slot];
    return SPA_OK;
}

// side streams: created once per context; spa_ctx_destroy releases them
int spa_aux_streams(spa_ctx *ctx)
{
    if (ctx->aux_ready) return SPA_OK;
    // aux[1] carries the few long, LDS-hungry replays of a call: highest priority, so that its
    // workgroups are placed before the many small ones of the other streams fill the CUs
    int prio_lo = 0, prio_hi = 0;
    SPA_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    for (int i = 0; i < 2; ++i) {
        SPA_HIP(hipStreamCreateWithPriority(&ctx->aux[i], hipStreamNonBlocking, i == 1 ? prio_hi : prio_lo));
        SPA_HIP(hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming));
    }
    SPA_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    ctx->aux_ready = 1;
    return SPA_OK;
}


// ---------------------------------------------------------------------------------------
// diagnostics (tools/lds_probe.py): a kernel shaped like the k_slic_assign variant of HISTORY.md section 5 (open items) — every workgroup
// fills an LDS table with values computed by its threads, then every thread reads four consecutive floats of a row per loop
// step (address in a vector register advanced by a vector add) and folds them with packed adds — whose result is a pure
// function of (workgroup, thread).  Run beside another kernel on a second stream and compared with a run alone.
// mode 0: 16-byte reads; 1: four 4-byte reads; 2: the same address for every lane of a row group (broadcast)
// ---------------------------------------------------------------------------------------
typedef float dbg_f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_debug_lds_probe(float *__restrict__ out, int steps, int mode, float seed)
{
    __shared__ __attribute__((aligned(16))) float tab[32 * 32];
    __shared__ uint4 other[256 * 3];
    const int tid = threadIdx.x;
    other[tid * 3] = make_uint4(tid, blockIdx.x, 0u, 0u);
    for (int e = tid; e < steps * 32; e += 256) {
        const float t = seed * (float)(blockIdx.x % 977) - (float)(e & 31) * 1.25f - (float)(e >> 5);
        tab[e] = ((e * 7 + (int)blockIdx.x) % 11 == 0) ? INFINITY : t * t;
    }
    __syncthreads();
    dbg_f32x2 acc0 = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f};
    float best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int bl[4] = {-1, -1, -1, -1};
    const float fy = (float)(tid >> 3);
#pragma unroll 1
    for (int jj = 0; jj < steps; ++jj) {
        const uint4 e0 = other[(jj * 3) % 768];
        const float *q = tab + jj * 32 + (mode == 2 ? 0 : (tid & 7) * 4);
        float4 d;
        if (mode == 1) { d.x = ((const volatile float *)q)[0]; d.y = ((const volatile float *)q)[1]; d.z = ((const volatile float *)q)[2]; d.w = ((const volatile float *)q)[3]; }
        else d = *(const float4 *)q;
        const float ty = (float)(e0.x & 31u) - fy;
        const float dy = (jj & 3) == 3 ? INFINITY : ty * ty;
        const dbg_f32x2 p0 = (dbg_f32x2{dy, dy} + dbg_f32x2{d.x, d.y}) * dbg_f32x2{0.37f, 0.37f} + acc0 * dbg_f32x2{0.001f, 0.001f};
        const dbg_f32x2 p1 = (dbg_f32x2{dy, dy} + dbg_f32x2{d.z, d.w}) * dbg_f32x2{0.37f, 0.37f} + acc1 * dbg_f32x2{0.001f, 0.001f};
        const float dd[4] = {p0.x, p0.y, p1.x, p1.y};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool take = best[i] > dd[i];
            best[i] = take ? dd[i] : best[i];
            bl[i] = take ? jj : bl[i];
        }
        if (p0.x < 1e30f) acc0 = p0;
        if (p1.x < 1e30f) acc1 = p1;
    }
    float4 r = make_float4(best[0] + (float)bl[0], best[1] + (float)bl[1], best[2] + (float)bl[2], best[3] + (float)bl[3]);
    *(float4 *)(out + ((long long)blockIdx.x * 256 + tid) * 4) = r;
}

extern "C" int spa_debug_lds_probe(spa_ctx *ctx, float *out, int32_t n_wg, int32_t steps, int32_t mode, void *stream)
{
    SPA_ARG(ctx && out && n_wg > 0 && steps > 0 && steps <= 32 && mode >= 0 && mode <= 2);
    hipLaunchKernelGGL(k_debug_lds_probe, dim3((unsigned)n_wg), dim3(256), 0, spa_stream(stream), out, steps, mode, 0.731f);
    SPA_LAUNCH_CHECK();
    return SPA_OK;
}

// diagnostics: copy `bytes` of workspace `which` (offset in bytes) to the host; synchronises
extern "C" int spa_ws_generation(spa_ctx *ctx) { return ctx ? ctx->ws_generation : -1; }

extern "C" int spa_debug_set(spa_ctx *ctx, int32_t key, int32_t value)
{
    SPA_ARG(ctx && (key == 1 || key == 2) && (value == 0 || value == 1));
    if (key == 1) ctx->convp_on = value;
    else {
#ifdef SPA_DIAG
        ctx->dbg_slic_ldsx = value;
#else
        SPA_ARG(!"spa_debug_set key 2 (the reproducer variant of k_slic_assign) exists in diagnostic builds only: make EXTRA=-DSPA_DIAG");
#endif
    }
    return SPA_OK;
}

extern "C" int spa_debug_peek(spa_ctx *ctx, int which, size_t offset, size_t bytes, void *host)
{
    if (which == -1) which = WS_DEBUG;           // the stamp buffer of diagnostic kernel builds
    SPA_ARG(ctx && host && which >= 0 && which < WS_COUNT && offset + bytes <= ctx->ws_bytes[which]);
    SPA_HIP(hipDeviceSynchronize());
    SPA_HIP(hipMemcpy(host, (const char *)ctx->ws[which] + offset, bytes, hipMemcpyDeviceToHost));
    return SPA_OK;
}
