// microbenchmark: cycles per dependent v_add_f32 in one wave (alone on its SIMD), with 1..4
// independent chains interleaved, and for the LDS-fed chain loop shape of k_slic_update.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CH>
__global__ void k_dep(float *out, unsigned long long *cyc, int n, float seed)
{
    float a[CH];
    for (int c = 0; c < CH; ++c) a[c] = seed + c;
    float x = seed * 0.5f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = a[c] + x;
        }
        asm volatile("" : "+v"(x));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < CH; ++c) s += a[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_lds_chain(float *out, unsigned long long *cyc, int n)
{
    __shared__ __attribute__((aligned(16))) float st[5 * 164];
    for (int i = threadIdx.x; i < 5 * 164; i += 64) st[i] = 1.0f + i;
    __syncthreads();
    float acc = 0;
    const int lane = threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (lane < 5) {
        const float4 *r4 = (const float4 *)(st + lane * 164);
        for (int rep = 0; rep < n; ++rep) {
            for (int j = 0; j < 128; j += 16) {
                const float4 q0 = r4[(j >> 2)], q1 = r4[(j >> 2) + 1], q2 = r4[(j >> 2) + 2], q3 = r4[(j >> 2) + 3];
                acc = acc + q0.x; acc = acc + q0.y; acc = acc + q0.z; acc = acc + q0.w;
                acc = acc + q1.x; acc = acc + q1.y; acc = acc + q1.z; acc = acc + q1.w;
                acc = acc + q2.x; acc = acc + q2.y; acc = acc + q2.z; acc = acc + q2.w;
                acc = acc + q3.x; acc = acc + q3.y; acc = acc + q3.z; acc = acc + q3.w;
            }
            asm volatile("" : "+v"(acc));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    const int n = 4096;
#define RUN(CH) { hipLaunchKernelGGL(k_dep<CH>, dim3(1), dim3(64), 0, 0, out, cyc, n, 1.5f); hipDeviceSynchronize(); \
    hipLaunchKernelGGL(k_dep<CH>, dim3(1), dim3(64), 0, 0, out, cyc, n, 1.5f); hipDeviceSynchronize(); \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
    printf("chains %d: %.2f cycles per add instruction, %.2f per step of all chains\n", CH, (double)h / (n * 16.0 * CH), (double)h / (n * 16.0)); }
    RUN(1) RUN(2) RUN(3) RUN(4)
    hipLaunchKernelGGL(k_lds_chain, dim3(1), dim3(64), 0, 0, out, cyc, 1000); hipDeviceSynchronize();
    hipLaunchKernelGGL(k_lds_chain, dim3(1), dim3(64), 0, 0, out, cyc, 1000); hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("LDS-fed chain (non-pipelined reads): %.2f cycles per pixel\n", (double)h / (1000.0 * 128));
    return 0;
}
