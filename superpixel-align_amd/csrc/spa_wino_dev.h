// Device helpers of the Winograd kernels (spa_wino.hip).
#pragma once
#include "spa_common.h"

struct WinoGeom { int B, H, W, d, th, tw; long long T; };
typedef float wino_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 wino_nt_load(const float4 *p) { const wino_v4 v = __builtin_nontemporal_load((const wino_v4 *)p); return make_float4(v[0], v[1], v[2], v[3]); }

// tile id -> (image, sub-grid, tile row, tile column)
__device__ __forceinline__ void wino_tile(const WinoGeom &g, long long t, int &b, int &sy, int &sx, int &ty, int &tx)
{
    tx = (int)(t % g.tw); t /= g.tw;
    ty = (int)(t % g.th); t /= g.th;
    sx = (int)(t % g.d); t /= g.d;
    sy = (int)(t % g.d);
    b = (int)(t / g.d);
}

typedef float wino_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 wino_nt_load(const float2 *p) { const wino_v2 v = __builtin_nontemporal_load((const wino_v2 *)p); return make_float2(v[0], v[1]); }
__device__ __forceinline__ unsigned wino_absmax_bits(float2 v) { return max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu); }
__device__ __forceinline__ unsigned wino_absmax_bits(float4 v) { return max(max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu), max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)); }
__device__ __forceinline__ float2 operator+(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 operator-(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 operator*(float s, float2 a) { return make_float2(s * a.x, s * a.y); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
template <typename V> __device__ __forceinline__ V wino_zero();
template <> __device__ __forceinline__ float2 wino_zero<float2>() { return make_float2(0.f, 0.f); }
template <> __device__ __forceinline__ float4 wino_zero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float2 wino_relu(float2 v) { return make_float2(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)); }
__device__ __forceinline__ float4 wino_relu(float4 v) { return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)); }

// y = B^T x for a 6-vector of float2
template <typename V>
__device__ __forceinline__ void wino4_bt(const V (&x)[6], V (&y)[6])
{
    y[0] = 2.0f * (x[0] + x[4]) + 3.0f * (x[3] - x[1]) - 4.0f * x[2];
    y[1] = 2.0f * (x[4] - x[1]) + x[2] + 5.0f * x[3];
    y[2] = 5.0f * x[2] - 2.0f * (x[1] + x[4]) - x[3];
    y[3] = 2.0f * (x[1] - x[3]) + x[2] - x[4];
    y[4] = (x[1] - x[3]) + 2.0f * (x[4] - x[2]);
    y[5] = 2.0f * (x[1] + x[5]) + 3.0f * (x[4] - x[2]) - 4.0f * x[3];
}

// y = A^T x: 4 outputs from 6
template <typename V>
__device__ __forceinline__ void wino4_at(const V (&x)[6], V (&y)[4])
{
    y[0] = ((x[0] + x[1]) + x[2]) + (x[3] + x[4]);
    y[1] = (x[1] - x[2]) + (0.5f * x[3] - 2.0f * x[4]);
    y[2] = (x[1] + x[2]) + (0.25f * x[3] + 4.0f * x[4]);
    y[3] = ((x[1] - x[2]) + (0.125f * x[3] - 8.0f * x[4])) + x[5];
}

struct WinoScale { float c[36]; };

__device__ __forceinline__ int wino_amax_exp(unsigned bits)
{
    int e = (int)(bits >> 23) - 127;
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    return bits == 0u ? 0 : e;
}
__device__ __forceinline__ float wino_pow2(int k) { return __uint_as_float((unsigned)(127 + k) << 23); }

// F(4x4,3x3) tiling of the sub-grids of a dilation (host)
static inline void wino4_geom(int B, int H, int W, int d, WinoGeom *g)
{
    g->B = B; g->H = H; g->W = W; g->d = d;
    const int hs = (H + d - 1) / d, ws = (W + d - 1) / d;
    g->th = (hs + 3) / 4; g->tw = (ws + 3) / 4;
    g->T = (long long)B * d * d * g->th * g->tw;
}
