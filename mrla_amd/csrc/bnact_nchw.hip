// Fused BatchNorm2d (+ReLU) for NCHW activations: the producer-side epilogue of the MRLA blocks (SURVEY.md 8f rank 1:
// `bn -> relu` in front of the path; reference call sites resnet/models/resnet_mrla_light.py:93-102).
//
//   forward : plane_moments  (sum x, sum x^2 per (image, channel))      -> mrla_bn_stats_fwd (base_nchw.hip) -> sc, sh
//             affine_act_fwd  y = relu?(sc[c]*x + sh[c])
//   backward: plane_dmoments (sum dz, sum dz*x, dz = dy*[y > 0])         -> mrla_bn_stats_bwd -> (e, f, h)
//             affine_act_bwd  dx = e[c]*dz + f[c]*x + h[c]
// HBM traffic: forward 1N + 2N, backward 2N + 3N elements (the compulsory minimum for a train-mode BatchNorm whose
// statistics are a global dependency); MIOpen's kernels move the same bytes at ~1.2 TB/s on this model.
#include <algorithm>

#include "mrla_device.h"
#include "mrla_kernels.h"

namespace mrla {

template <typename T, int VW> struct VecB { typedef T type __attribute__((ext_vector_type(VW))); };

template <typename T, int VW>
__device__ __forceinline__ void ldv(const T* __restrict__ p, float (&v)[VW]) {
  if constexpr (VW == 1) {
    v[0] = to_f(p[0]);
  } else {
    typedef typename VecB<T, VW>::type VT;
    const VT t = *reinterpret_cast<const VT*>(p);
#pragma unroll
    for (int i = 0; i < VW; ++i) v[i] = static_cast<float>(t[i]);
  }
}
template <typename T, int VW>
__device__ __forceinline__ void stv(T* __restrict__ p, const float (&v)[VW]) {
  if constexpr (VW == 1) {
    p[0] = from_f<T>(v[0]);
  } else {
    typedef typename VecB<T, VW>::type VT;
    VT t;
#pragma unroll
    for (int i = 0; i < VW; ++i) t[i] = static_cast<T>(v[i]);
    *reinterpret_cast<VT*>(p) = t;
  }
}
template <typename T> __device__ __forceinline__ bool al16(const T* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ------------------------------------------------------------------------------------------------
// per-plane moments.  One wave per plane; a workgroup walks over planes with a grid stride.
// MODE 0: (sum x, sum x^2)          MODE 1: (sum dz, sum dz*x), dz = dy * [sc*x + sh > 0 or !relu]
// ------------------------------------------------------------------------------------------------
template <typename T, int MODE, int VW>
__device__ __forceinline__ void plane_sums(const T* __restrict__ xp, const T* __restrict__ gp, int HW, int lane, float s,
                                           float h, bool relu, float pv, float& s1, float& s2) {
  for (int e = lane * VW; e < HW; e += kWave * VW) {
    float xv[VW];
    ldv<T, VW>(xp + e, xv);
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < VW; ++i) { const float d = xv[i] - pv; s1 += d; s2 = fmaf(d, d, s2); }
    } else {
      float gv[VW];
      ldv<T, VW>(gp + e, gv);
#pragma unroll
      for (int i = 0; i < VW; ++i) {
        const float dz = (!relu || fmaf(s, xv[i], h) > 0.f) ? gv[i] : 0.f;
        s1 += dz;
        s2 = fmaf(dz, xv[i] - pv, s2);                   // pv: the center (MODE 1)
      }
    }
  }
}

template <typename T, int MODE>
__global__ __launch_bounds__(kThreads) void plane_moments_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                 const float* __restrict__ sc, const float* __restrict__ sh,
                                                                 int relu, float* __restrict__ out /*[planes,2]*/,
                                                                 float* __restrict__ pivot, int planes, int C, int HW) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  constexpr int VEC = 16 / sizeof(T);
  const bool vec_ok = (HW % VEC == 0) && al16(x) && (MODE == 0 || al16(dy));
  for (int p = blockIdx.x * kWaves + wave; p < planes; p += gridDim.x * kWaves) {
    const T* xp = x + (size_t)p * HW;
    const T* gp = MODE ? dy + (size_t)p * HW : nullptr;
    const int c = p % C;
    const float s = (MODE && relu) ? sc[c] : 0.f, h = (MODE && relu) ? sh[c] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    // MODE 0 with `pivot`: sums of (x - p), p = the channel's first element of the first image (see bnact_nhwc.hip)
    // MODE 1: `pivot` is the INPUT center (the saved batch mean): sum dz*(x - center)
    const float pv = !pivot ? 0.f : (MODE == 0 ? to_f(x[(size_t)c * HW]) : pivot[c]);
    if (MODE == 0 && pivot && p < C && lane == 0) pivot[c] = pv;
    if (vec_ok) plane_sums<T, MODE, VEC>(xp, gp, HW, lane, s, h, relu != 0, pv, s1, s2);
    else        plane_sums<T, MODE, 1>(xp, gp, HW, lane, s, h, relu != 0, pv, s1, s2);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) { out[(size_t)p * 2] = s1; out[(size_t)p * 2 + 1] = s2; }
  }
}

// Small planes (HW < 512): several planes per wave would leave most lanes idle with the kernel above, so a
// workgroup stages a contiguous run of planes in LDS (16 B per lane from HBM) and reduces from there.
template <typename T, int MODE>
__global__ __launch_bounds__(kThreads) void plane_moments_small_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                       const float* __restrict__ sc,
                                                                       const float* __restrict__ sh, int relu,
                                                                       float* __restrict__ out, float* __restrict__ pivot,
                                                                       int planes, int C, int HW,
                                                                       int PP /*planes per workgroup chunk*/) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* xs = reinterpret_cast<T*>(smem);
  const int stride = ((PP * HW * (int)sizeof(T) + 15) / 16) * 16 / (int)sizeof(T);
  T* gs = xs + stride;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  for (int p0 = blockIdx.x * PP; p0 < planes; p0 += gridDim.x * PP) {
    const int np = min(PP, planes - p0);
    slab_load(xs, x + (size_t)p0 * HW, np * HW, tid);
    if (MODE) slab_load(gs, dy + (size_t)p0 * HW, np * HW, tid);
    __syncthreads();
    // 16 lanes per plane, 4 planes per wave step
    const int sub = lane >> 4, l16 = lane & 15;
    for (int q = wave * 4 + sub; q < np; q += kWaves * 4) {
      const int c = (p0 + q) % C;
      const float s = (MODE && relu) ? sc[c] : 0.f, h = (MODE && relu) ? sh[c] : 0.f;
      float s1 = 0.f, s2 = 0.f;
      const float pv = !pivot ? 0.f : (MODE == 0 ? to_f(x[(size_t)c * HW]) : pivot[c]);
      if (MODE == 0 && pivot && p0 + q < C && l16 == 0) pivot[c] = pv;
      for (int e = l16; e < HW; e += 16) {
        const float xv = to_f(xs[q * HW + e]);
        if (MODE == 0) { const float d = xv - pv; s1 += d; s2 = fmaf(d, d, s2); }
        else {
          const float dz = (!relu || fmaf(s, xv, h) > 0.f) ? to_f(gs[q * HW + e]) : 0.f;
          s1 += dz;
          s2 = fmaf(dz, xv - pv, s2);
        }
      }
      s1 = seg_sum(s1, lane, 16);
      s2 = seg_sum(s2, lane, 16);
      if (l16 == 15) { out[(size_t)(p0 + q) * 2] = s1; out[(size_t)(p0 + q) * 2 + 1] = s2; }
    }
    // lanes whose q >= np still have to take part in the DPP steps above: they do (loop bound is per sub-group,
    // DPP row ops never cross 16-lane rows), nothing to fix up here
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// elementwise passes
// ------------------------------------------------------------------------------------------------
template <typename T, int VW, bool BWD>
__device__ __forceinline__ void affine_body(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ a,
                                            const float* __restrict__ sc, const float* __restrict__ sh, int relu,
                                            T* __restrict__ out, size_t total, int C, int HW) {
  const size_t stride = (size_t)gridDim.x * kThreads * VW;
  for (size_t e0 = ((size_t)blockIdx.x * kThreads + threadIdx.x) * VW; e0 < total; e0 += stride) {
    float xv[VW], gv[VW], y[VW];
    ldv<T, VW>(x + e0, xv);
    if (BWD) ldv<T, VW>(dy + e0, gv);
    const size_t pl0 = e0 / HW, pl1 = (e0 + VW - 1) / HW;
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      const int c = (int)(((pl0 == pl1) ? pl0 : (e0 + i) / HW) % C);
      if (!BWD) {
        const float z = fmaf(sc[c], xv[i], sh[c]);
        y[i] = relu ? fmaxf(z, 0.f) : z;
      } else {
        // a[c,3] = (e, f, h):  dx = e*dz + f*x + h,  dz = dy*[sc*x + sh > 0]
        const float dz = (!relu || fmaf(sc[c], xv[i], sh[c]) > 0.f) ? gv[i] : 0.f;
        y[i] = fmaf(a[c * 3], dz, fmaf(a[c * 3 + 1], xv[i], a[c * 3 + 2]));
      }
    }
    stv<T, VW>(out + e0, y);
  }
}

template <typename T, bool BWD>
__global__ __launch_bounds__(kThreads) void affine_act_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                              const float* __restrict__ a, const float* __restrict__ sc,
                                                              const float* __restrict__ sh, int relu, T* __restrict__ out,
                                                              size_t total, int C, int HW) {
  constexpr int VEC = 16 / sizeof(T);
  if (total % VEC == 0 && al16(x) && al16(out) && (!BWD || al16(dy)))
    affine_body<T, VEC, BWD>(x, dy, a, sc, sh, relu, out, total, C, HW);
  else
    affine_body<T, 1, BWD>(x, dy, a, sc, sh, relu, out, total, C, HW);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
#define MRLA_DISPATCH_B(DT, CALL)        \
  switch (DT) {                          \
    case MRLA_F32:  CALL(float); break;  \
    case MRLA_BF16: CALL(bf16_t); break; \
    case MRLA_F16:  CALL(f16_t); break;  \
    default: return MRLA_EINVAL;         \
  }

static int launch_moments(const void* x, const void* dy, const float* sc, const float* sh, int relu, float* out,
                          float* pivot, int B, int C, int HW, int dtype, int mode, hipStream_t st) {
  const int planes = B * C;
  const size_t es = dtype_size(dtype);
  if (HW >= 512) {
    const int grid = std::max(1, std::min((planes + kWaves - 1) / kWaves, 256 * 8));
#define CALL(TT)                                                                                                  \
  if (mode) hipLaunchKernelGGL((plane_moments_kernel<TT, 1>), dim3(grid), dim3(kThreads), 0, st, (const TT*)x,   \
                               (const TT*)dy, sc, sh, relu, out, pivot, planes, C, HW);                           \
  else      hipLaunchKernelGGL((plane_moments_kernel<TT, 0>), dim3(grid), dim3(kThreads), 0, st, (const TT*)x,   \
                               (const TT*)dy, sc, sh, relu, out, pivot, planes, C, HW);
    MRLA_DISPATCH_B(dtype, CALL)
#undef CALL
  } else {
    int PP = std::max(1, std::min(planes, 4096 / HW));
    const size_t per = (((size_t)PP * HW * es + 15) / 16) * 16;
    const size_t lds = per * (mode ? 2 : 1);
    const int grid = std::max(1, std::min((planes + PP - 1) / PP, 256 * 8));
#define CALL(TT)                                                                                                      \
  if (mode) hipLaunchKernelGGL((plane_moments_small_kernel<TT, 1>), dim3(grid), dim3(kThreads), lds, st, (const TT*)x, \
                               (const TT*)dy, sc, sh, relu, out, pivot, planes, C, HW, PP);                           \
  else      hipLaunchKernelGGL((plane_moments_small_kernel<TT, 0>), dim3(grid), dim3(kThreads), lds, st, (const TT*)x, \
                               (const TT*)dy, sc, sh, relu, out, pivot, planes, C, HW, PP);
    MRLA_DISPATCH_B(dtype, CALL)
#undef CALL
  }
  return hip_status(hipGetLastError());
}

int launch_plane_moments(const void* x, float* amom, float* pivot, int B, int C, int HW, int dtype, hipStream_t st) {
  return launch_moments(x, nullptr, nullptr, nullptr, 0, amom, pivot, B, C, HW, dtype, 0, st);
}

int launch_plane_dmoments(const void* dy, const void* x, const float* sc, const float* sh, const float* center, int relu,
                          float* tmom, int B, int C, int HW, int dtype, hipStream_t st) {
  return launch_moments(x, dy, sc, sh, relu, tmom, const_cast<float*>(center), B, C, HW, dtype, 1, st);
}

int launch_affine_act(const void* x, const void* dy, const float* a, const float* sc, const float* sh, int relu,
                      void* out, int B, int C, int HW, int dtype, int bwd, hipStream_t st) {
  const size_t total = (size_t)B * C * HW;
  const size_t vec = 16 / dtype_size(dtype);
  const size_t want = (total / vec + kThreads - 1) / kThreads;
  const int grid = (int)std::max<size_t>(1, std::min<size_t>(want, 256 * 16));
#define CALL(TT)                                                                                                \
  if (bwd) hipLaunchKernelGGL((affine_act_kernel<TT, true>), dim3(grid), dim3(kThreads), 0, st, (const TT*)x,   \
                              (const TT*)dy, a, sc, sh, relu, (TT*)out, total, C, HW);                          \
  else     hipLaunchKernelGGL((affine_act_kernel<TT, false>), dim3(grid), dim3(kThreads), 0, st, (const TT*)x,  \
                              (const TT*)dy, a, sc, sh, relu, (TT*)out, total, C, HW);
  MRLA_DISPATCH_B(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

}  // namespace mrla
