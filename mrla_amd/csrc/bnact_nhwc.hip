// Fused BatchNorm2d (+ReLU) passes for channels_last (NHWC) activations; same roles as bnact_nchw.hip.
// Lane = 8 consecutive channels (one 16-byte access), 8 lanes = 64 channels, a wave = 8 pixels x 64 channels.
//   nhwc_moments   MODE 0: (sum x, sum x^2)   MODE 1: (sum dz, sum dz*x), dz = dy*[sc*x+sh > 0 or !relu]
//                  -> partial sums [b * nsplit, c, 2]   (consumed by plain_bn_{fwd,bwd}_kernel with B = b*nsplit)
//   nhwc_affine    forward y = relu?(sc*x + sh);  backward dx = e*dz + f*x + h
#include <algorithm>

#include "mrla_device.h"
#include "mrla_kernels.h"

namespace mrla {

template <typename T> struct V16 { typedef T type __attribute__((ext_vector_type(16 / sizeof(T)))); };

template <typename T>
__device__ __forceinline__ void ld16(const T* __restrict__ p, float (&v)[16 / sizeof(T)]) {
  typedef typename V16<T>::type VT;
  const VT t = *reinterpret_cast<const VT*>(p);
#pragma unroll
  for (int i = 0; i < (int)(16 / sizeof(T)); ++i) v[i] = static_cast<float>(t[i]);
}
template <typename T>
__device__ __forceinline__ void st16(T* __restrict__ p, const float (&v)[16 / sizeof(T)]) {
  typedef typename V16<T>::type VT;
  VT t;
#pragma unroll
  for (int i = 0; i < (int)(16 / sizeof(T)); ++i) t[i] = static_cast<T>(v[i]);
  *reinterpret_cast<VT*>(p) = t;
}

// grid: (channel chunks of 64*?, b * nsplit).  Requires C % VEC == 0.
template <typename T, int MODE>
__global__ __launch_bounds__(kThreads) void nhwc_moments_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                const float* __restrict__ sc, const float* __restrict__ sh,
                                                                int relu, float* __restrict__ out /*[b*nsplit, c, 2]*/,
                                                                int C, int HW, int nsplit) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int LPC = 64 / VEC;                  // lanes per 64-channel row (8 for 16-bit types, 16 for fp32)
  constexpr int PPW = kWave / LPC;               // pixels per wave-instruction
  __shared__ float red[kWaves][2][64];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int cl = (lane % LPC) * VEC;             // first channel of this lane inside the chunk
  const int c0 = blockIdx.x * 64 + cl;
  const bool cv = c0 < C;
  const int bs = blockIdx.y, b = bs / nsplit, sp = bs - b * nsplit;
  const int npix = HW / nsplit;
  const size_t base = ((size_t)b * HW + (size_t)sp * npix) * C;
  float scv[VEC], shv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { scv[i] = (MODE && relu && cv) ? sc[c0 + i] : 0.f; shv[i] = (MODE && relu && cv) ? sh[c0 + i] : 0.f; }
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  if (cv) {
    for (int p = wave * PPW + lane / LPC; p < npix; p += kWaves * PPW) {
      float xv[VEC];
      ld16<T>(x + base + (size_t)p * C + c0, xv);
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) { s1[i] += xv[i]; s2[i] = fmaf(xv[i], xv[i], s2[i]); }
      } else {
        float gv[VEC];
        ld16<T>(dy + base + (size_t)p * C + c0, gv);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float dz = (!relu || fmaf(scv[i], xv[i], shv[i]) > 0.f) ? gv[i] : 0.f;
          s1[i] += dz;
          s2[i] = fmaf(dz, xv[i], s2[i]);
        }
      }
    }
  }
  // lanes l, l + LPC, l + 2*LPC, ... hold the same channels: butterfly over the pixel sub-index
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    for (int off = LPC; off < kWave; off <<= 1) {
      s1[i] += __shfl_xor(s1[i], off, kWave);
      s2[i] += __shfl_xor(s2[i], off, kWave);
    }
  }
  if (lane < LPC) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { red[wave][0][cl + i] = s1[i]; red[wave][1][cl + i] = s2[i]; }
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int k = threadIdx.x / 64, ch = threadIdx.x % 64;
    const int c = blockIdx.x * 64 + ch;
    if (c < C) {
      float s = 0.f;
#pragma unroll
      for (int wv = 0; wv < kWaves; ++wv) s += red[wv][k][ch];
      out[((size_t)bs * C + c) * 2 + k] = s;
    }
  }
}

template <typename T, bool BWD>
__global__ __launch_bounds__(kThreads) void nhwc_affine_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                               const float* __restrict__ a, const float* __restrict__ sc,
                                                               const float* __restrict__ sh, int relu, T* __restrict__ out,
                                                               size_t total, int C) {
  constexpr int VEC = 16 / sizeof(T);
  const size_t stride = (size_t)gridDim.x * kThreads * VEC;
  const size_t first = ((size_t)blockIdx.x * kThreads + threadIdx.x) * VEC;
  const bool fixed = (stride % C) == 0;          // every vector of this thread then covers the same channels
  float scv[VEC], shv[VEC], ev[VEC], fv[VEC], hv[VEC];
  auto load_coef = [&](int c0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      scv[i] = sc[c0 + i]; shv[i] = sh[c0 + i];
      if (BWD) { ev[i] = a[(c0 + i) * 3]; fv[i] = a[(c0 + i) * 3 + 1]; hv[i] = a[(c0 + i) * 3 + 2]; }
    }
  };
  if (fixed && first < total) load_coef((int)(first % C));
  for (size_t e0 = first; e0 < total; e0 += stride) {
    if (!fixed) load_coef((int)(e0 % C));
    float xv[VEC], gv[VEC], y[VEC];
    ld16<T>(x + e0, xv);
    if (BWD) ld16<T>(dy + e0, gv);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float z = fmaf(scv[i], xv[i], shv[i]);
      if (!BWD) y[i] = relu ? fmaxf(z, 0.f) : z;
      else {
        const float dz = (!relu || z > 0.f) ? gv[i] : 0.f;
        y[i] = fmaf(ev[i], dz, fmaf(fv[i], xv[i], hv[i]));
      }
    }
    st16<T>(out + e0, y);
  }
}

// ------------------------------------------------------------------------------------------------
int nhwc_bn_splits(int B, int C, int HW) {
  const long chunks = (C + 63) / 64;
  static const int cand[] = {1, 2, 4, 7, 8, 14, 16, 28, 32, 49, 56, 64};
  int best = 1;
  for (int s : cand) {
    if (HW % s) continue;
    best = s;
    if ((long)B * chunks * s >= 2048) break;
  }
  return best;
}

#define MRLA_DISPATCH_N(DT, CALL)        \
  switch (DT) {                          \
    case MRLA_F32:  CALL(float); break;  \
    case MRLA_BF16: CALL(bf16_t); break; \
    case MRLA_F16:  CALL(f16_t); break;  \
    default: return MRLA_EINVAL;         \
  }

int launch_nhwc_moments(const void* x, const void* dy, const float* sc, const float* sh, int relu, float* out, int B,
                        int C, int HW, int dtype, int mode, hipStream_t st) {
  const int vec = 16 / (int)dtype_size(dtype);
  if (C % vec) return MRLA_EUNSUPPORTED;
  const int ns = nhwc_bn_splits(B, C, HW);
  const dim3 grid((C + 63) / 64, B * ns);
#define CALL(TT)                                                                                                     \
  if (mode) hipLaunchKernelGGL((nhwc_moments_kernel<TT, 1>), grid, dim3(kThreads), 0, st, (const TT*)x, (const TT*)dy, \
                               sc, sh, relu, out, C, HW, ns);                                                        \
  else      hipLaunchKernelGGL((nhwc_moments_kernel<TT, 0>), grid, dim3(kThreads), 0, st, (const TT*)x, (const TT*)dy, \
                               sc, sh, relu, out, C, HW, ns);
  MRLA_DISPATCH_N(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_nhwc_affine(const void* x, const void* dy, const float* a, const float* sc, const float* sh, int relu,
                       void* out, int B, int C, int HW, int dtype, int bwd, hipStream_t st) {
  const size_t total = (size_t)B * C * HW;
  const int vec = 16 / (int)dtype_size(dtype);
  if (C % vec) return MRLA_EUNSUPPORTED;
  const size_t want = (total / vec + kThreads - 1) / kThreads;
  const int grid = (int)std::max<size_t>(1, std::min<size_t>(want, 256 * 16));
#define CALL(TT)                                                                                                   \
  if (bwd) hipLaunchKernelGGL((nhwc_affine_kernel<TT, true>), dim3(grid), dim3(kThreads), 0, st, (const TT*)x,     \
                              (const TT*)dy, a, sc, sh, relu, (TT*)out, total, C);                                 \
  else     hipLaunchKernelGGL((nhwc_affine_kernel<TT, false>), dim3(grid), dim3(kThreads), 0, st, (const TT*)x,    \
                              (const TT*)dy, a, sc, sh, relu, (TT*)out, total, C);
  MRLA_DISPATCH_N(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

}  // namespace mrla
