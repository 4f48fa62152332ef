// Device-side helpers shared by the MRLA HIP kernels (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrla {

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
constexpr int kThreads = 256;          // 4 waves per workgroup everywhere
constexpr int kWaves = kThreads / kWave;

// Moment slots written by the forward statistics pass, per (image, channel).
enum { M_SX = 0, M_SV = 1, M_SO = 2, M_SVV = 3, M_SVO = 4, M_SOO = 5, M_N = 6 };
// Moment slots written by the backward statistics pass.
enum { D_D = 0, D_DV = 1, D_DO = 2, D_N = 3 };

template <typename T> __device__ __forceinline__ float to_f(T v) { return static_cast<float>(v); }
template <typename T> __device__ __forceinline__ T from_f(float v) { return static_cast<T>(v); }

// lane i receives the value of lane i-1 / i+1 of the wave (DPP wave_shr:1 / wave_shl:1, GFX9 family).
// Edge lanes receive 0; callers mask plane edges themselves.
__device__ __forceinline__ float lane_prev(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_next(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}

// Sum over the `w` consecutive lanes [lane-col, lane-col+w) that hold one plane row; the result is
// valid in the lane with col == 0.  All 64 lanes must call it.
__device__ __forceinline__ float seg_sum(float v, int col, int w) {
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    float t = __shfl_down(v, off, kWave);
    if (off < w && col + off < w) v += t;
  }
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;   // valid in lane 0
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}

__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float u) {
  const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * u * u);
  return cdf + u * pdf;
}

// Cooperative linear copy of `n` contiguous elements global -> LDS (16 B per lane when aligned).
template <typename T>
__device__ __forceinline__ void slab_load(T* __restrict__ dst, const T* __restrict__ src, int n, int tid) {
  constexpr int VEC = 16 / sizeof(T);
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    const int nv = n / VEC;
    const u32x4* __restrict__ s4 = reinterpret_cast<const u32x4*>(src);
    u32x4* __restrict__ d4 = reinterpret_cast<u32x4*>(dst);
#pragma unroll 2
    for (int i = tid; i < nv; i += kThreads) d4[i] = __builtin_nontemporal_load(&s4[i]);
    for (int i = nv * VEC + tid; i < n; i += kThreads) dst[i] = src[i];
  } else {
    for (int i = tid; i < n; i += kThreads) dst[i] = src[i];
  }
}

// Cooperative linear copy LDS -> global.
template <typename T>
__device__ __forceinline__ void slab_store(T* __restrict__ dst, const T* __restrict__ src, int n, int tid) {
  constexpr int VEC = 16 / sizeof(T);
  if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    const int nv = n / VEC;
    const u32x4* __restrict__ s4 = reinterpret_cast<const u32x4*>(src);
    u32x4* __restrict__ d4 = reinterpret_cast<u32x4*>(dst);
#pragma unroll 2
    for (int i = tid; i < nv; i += kThreads) d4[i] = s4[i];
    for (int i = nv * VEC + tid; i < n; i += kThreads) dst[i] = src[i];
  } else {
    for (int i = tid; i < n; i += kThreads) dst[i] = src[i];
  }
}

}  // namespace mrla
