// Fused BatchNorm2d (+ReLU) passes for channels_last (NHWC) activations; same roles as bnact_nchw.hip.
// Lane = 8 consecutive channels (one 16-byte access), 8 lanes = 64 channels, a wave = 8 pixels x 64 channels.
//   nhwc_moments   MODE 0: (sum x, sum x^2)   MODE 1: (sum dz, sum dz*(x - center)), dz = dy*[sc*x+sh > 0 or !relu];
//                  `pivot` is an OUTPUT in MODE 0 (the shift it picked) and the INPUT `center` in MODE 1 (the batch mean
//                  the BatchNorm saved: dgamma = inv_std * sum dz*(x - mean) then needs no cancelling subtraction)
//                  -> partial sums [b * nsplit, c, 2]   (consumed by plain_bn_{fwd,bwd}_kernel with B = b*nsplit)
//   nhwc_affine    forward y = relu?(sc*x + sh);  backward dx = e*dz + f*x + h
#include <algorithm>

#include "mrla_device.h"
#include "mrla_kernels.h"

#ifndef MRLA_REVERSE_AFFINE
#define MRLA_REVERSE_AFFINE 1
#endif

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
                                                                float* __restrict__ pivot, int C, int HW, int nsplit) {
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
  float s1[VEC], s2[VEC], pv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; pv[i] = 0.f; }
  // MODE 0 with `pivot`: sums of (x - p), p = the channel's value at the first pixel of the first image -- a sample of
  // the distribution, so the second moment does not cancel when |mean| >> sigma
  if (MODE == 0 && pivot && cv) {
    ld16<T>(x + c0, pv);
    if (bs == 0 && wave == 0 && lane < LPC) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) pivot[c0 + i] = pv[i];
    }
  }
  if (MODE == 1 && pivot && cv) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) pv[i] = pivot[c0 + i];
  }
  if (cv) {
    for (int p = wave * PPW + lane / LPC; p < npix; p += kWaves * PPW) {
      float xv[VEC];
      ld16<T>(x + base + (size_t)p * C + c0, xv);
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) { const float d = xv[i] - pv[i]; s1[i] += d; s2[i] = fmaf(d, d, s2[i]); }
      } else {
        float gv[VEC];
        ld16<T>(dy + base + (size_t)p * C + c0, gv);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float dz = (!relu || fmaf(scv[i], xv[i], shv[i]) > 0.f) ? gv[i] : 0.f;
          s1[i] += dz;
          s2[i] = fmaf(dz, xv[i] - pv[i], s2[i]);
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
// "flat" forms for C % 64 == 0 (every BatchNorm of the networks here): a workgroup iteration covers 256 consecutive
// 16-byte vectors = PPI whole pixels x CW channels (CW = largest power of two <= 256*VEC dividing C), so the
// accesses of a workgroup are contiguous 4 KiB pieces when CW == C, a thread keeps the same channels from iteration to
// iteration, and UNR independent loads per lane are in flight.
// ------------------------------------------------------------------------------------------------
constexpr int kUnr = 4;

template <int N>
__device__ __forceinline__ void ldf(const float* __restrict__ p, float (&v)[N]) {
  static_assert(N % 4 == 0, "vector loads of 4 floats");
#pragma unroll
  for (int i = 0; i < N; i += 4) {
    const float4 t = *reinterpret_cast<const float4*>(p + i);
    v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
  }
}

// grid: (C / CW, b * nsplit).  out[b*nsplit, c, 2]
template <typename T, int MODE>
__global__ __launch_bounds__(kThreads) void nhwc_moments_flat_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                     const float* __restrict__ sc,
                                                                     const float* __restrict__ sh, int relu,
                                                                     const float* __restrict__ dp /*[b] or null*/,
                                                                     float* __restrict__ out, float* __restrict__ pivot,
                                                                     int C, int HW, int nsplit, int CW) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[2][kThreads * VEC];
  const int t = threadIdx.x;
  const int lpc = CW / VEC;                      // lanes per pixel piece (power of two, <= 256)
  const int ppi = kThreads / lpc;                // pixels per workgroup iteration
  const int cl = (t & (lpc - 1)) * VEC, ps = t / lpc;
  const int c0 = blockIdx.x * CW + cl;
  const int bs = blockIdx.y, b = bs / nsplit, sp = bs - b * nsplit;
  const int npix = HW / nsplit;
  const float dpb = (MODE && dp) ? dp[b] : 1.f;        // MRLA-base tail: dz = dp[b] * dOut * [sc*attn + sh > 0]
  const T* xp = x + ((size_t)b * HW + (size_t)sp * npix) * C + c0;
  const T* gp = MODE ? dy + ((size_t)b * HW + (size_t)sp * npix) * C + c0 : nullptr;
  float scv[VEC], shv[VEC];
  if ((MODE == 1 && relu) || (MODE == 2 && sc)) { ldf<VEC>(sc + c0, scv); ldf<VEC>(sh + c0, shv); }
  else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { scv[i] = 0.f; shv[i] = 0.f; }
  }
  float s1[VEC], s2[VEC], pv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; pv[i] = 0.f; }
  if (MODE == 0 && pivot) {                      // shifted sums, see nhwc_moments_kernel
    ld16<T>(x + c0, pv);
    if (bs == 0 && ps == 0) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) pivot[c0 + i] = pv[i];
    }
  }
  if (MODE == 1 && pivot) ldf<VEC>(pivot + c0, pv);      // the center (input)
  auto accumulate = [&](const float (&xv)[VEC], const float (&gv)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      if (MODE == 0) { const float d = xv[i] - pv[i]; s1[i] += d; s2[i] = fmaf(d, d, s2[i]); }
      else if (MODE == 2) {     // pooling of x_t = relu(round(sc*pre + sh) + o) for the inference path (x = pre, dy = o)
        const float z = sc ? to_f(from_f<T>(fmaf(scv[i], xv[i], shv[i]))) : xv[i];
        s1[i] += fmaxf(to_f(from_f<T>(z + gv[i])), 0.f);
      } else {
        const float dz = (!relu || fmaf(scv[i], xv[i], shv[i]) > 0.f) ? dpb * gv[i] : 0.f;
        s1[i] += dz;
        s2[i] = fmaf(dz, xv[i] - pv[i], s2[i]);
      }
    }
  };
  constexpr int UN = MODE ? kUnr : 2 * kUnr;       // independent 16-byte loads in flight per lane: 8 either way
  int p = ps;
  for (; p + (UN - 1) * ppi < npix; p += UN * ppi) {
    float xv[UN][VEC], gv[UN][VEC];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      ld16<T>(xp + (size_t)(p + u * ppi) * C, xv[u]);
      if (MODE) ld16<T>(gp + (size_t)(p + u * ppi) * C, gv[u]);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) accumulate(xv[u], gv[u]);
  }
  for (; p < npix; p += ppi) {
    float xv[VEC], gv[VEC];
    ld16<T>(xp + (size_t)p * C, xv);
    if (MODE) ld16<T>(gp + (size_t)p * C, gv);
    accumulate(xv, gv);
  }
  // thread t holds channels cl.. of pixel sub-index ps: element t*VEC + i of the [ppi][CW] tile
#pragma unroll
  for (int i = 0; i < VEC; ++i) { red[0][t * VEC + i] = s1[i]; red[1][t * VEC + i] = s2[i]; }
  __syncthreads();
  for (int j = t; j < 2 * CW; j += kThreads) {
    const int k = j / CW, ch = j - k * CW;
    float sum = 0.f;
    for (int r = 0; r < ppi; ++r) sum += red[k][r * CW + ch];
    out[((size_t)bs * C + blockIdx.x * CW + ch) * 2 + k] = sum;
  }
}

// mom[b, c, slot 0 of 6] = sum over the nsplit partial rows of an image (the pooled x_t of the inference path)
__global__ __launch_bounds__(kThreads) void nhwc_pool_finish_kernel(const float* __restrict__ part /*[b*ns, c, 2]*/,
                                                                    float* __restrict__ mom /*[b, c, M_REC]*/, int BC, int C,
                                                                    int ns) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= BC) return;
  const int b = i / C, c = i - b * C;
  float s = 0.f;
  for (int k = 0; k < ns; ++k) s += part[(((size_t)b * ns + k) * C + c) * 2];
  mom[(size_t)i * M_REC] = s;
}

// Elementwise over the flat tensor; requires (kThreads * VEC) % C == 0 (then a thread's channels never change).
template <typename T, bool BWD>
__global__ __launch_bounds__(kThreads) void nhwc_affine_flat_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                    const float* __restrict__ a,
                                                                    const float* __restrict__ sc,
                                                                    const float* __restrict__ sh, int relu,
                                                                    T* __restrict__ out, size_t total, int C, int iters) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr size_t STEP = (size_t)kThreads * VEC;
  const int c0 = (int)((threadIdx.x * VEC) & (unsigned)(C - 1));       // C is a power of two here
  float scv[VEC], shv[VEC], cbv[BWD ? 3 * VEC : 4];
  ldf<VEC>(sc + c0, scv);
  ldf<VEC>(sh + c0, shv);
  if constexpr (BWD) ldf<3 * VEC>(a + (size_t)c0 * 3, cbv);
  // the workgroups walk the tensor from its END: the statistics pass that ran just before left the end in the cache
  const size_t bx = MRLA_REVERSE_AFFINE ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  size_t e = bx * iters * STEP + (size_t)threadIdx.x * VEC;
  auto one = [&](const float (&xv)[VEC], const float (&gv)[VEC], size_t at) {
    float y[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float z = fmaf(scv[i], xv[i], shv[i]);
      if (!BWD) y[i] = relu ? fmaxf(z, 0.f) : z;
      else {
        const float dz = (!relu || z > 0.f) ? gv[i] : 0.f;
        y[i] = fmaf(cbv[3 * i], dz, fmaf(cbv[3 * i + 1], xv[i], cbv[3 * i + 2]));
      }
    }
    st16<T>(out + at, y);
  };
  int it = 0;
  for (; it + kUnr <= iters && e + (kUnr - 1) * STEP < total; it += kUnr, e += kUnr * STEP) {
    float xv[kUnr][VEC], gv[kUnr][VEC];
#pragma unroll
    for (int u = 0; u < kUnr; ++u) {
      ld16<T>(x + e + u * STEP, xv[u]);
      if (BWD) ld16<T>(dy + e + u * STEP, gv[u]);
    }
#pragma unroll
    for (int u = 0; u < kUnr; ++u) one(xv[u], gv[u], e + u * STEP);
  }
  for (; it < iters && e < total; ++it, e += STEP) {
    float xv[VEC], gv[VEC];
    ld16<T>(x + e, xv);
    if (BWD) ld16<T>(dy + e, gv);
    one(xv, gv, e);
  }
}

// ------------------------------------------------------------------------------------------------
// partial-moment rows per image: independent of the storage type (the caller sizes its buffers from it)
int nhwc_bn_splits(int B, int C, int HW) {
  (void)C;
  static const int cand[] = {1, 2, 4, 7, 8, 14, 16, 28, 32, 49, 56, 64};
  int best = 1;
  for (int s : cand) {
    if (HW % s || (s > 1 && HW / s < 16)) continue;     // keep >= 16 pixels per workgroup: the partial sums are traffic too
    best = s;
    if ((long)B * s >= 1024) break;
  }
  return best;
}

// widest power-of-two channel piece a workgroup iteration can cover, or 0 when the flat kernels do not apply
static int flat_cw(int C, int vec) {
  if (C % 64) return 0;
  int cw = 64;
  while (cw * 2 <= kThreads * vec && C % (cw * 2) == 0) cw *= 2;
  return cw;
}

#define MRLA_DISPATCH_N(DT, CALL)        \
  switch (DT) {                          \
    case MRLA_F32:  CALL(float); break;  \
    case MRLA_BF16: CALL(bf16_t); break; \
    case MRLA_F16:  CALL(f16_t); break;  \
    default: return MRLA_EINVAL;         \
  }

int launch_nhwc_moments(const void* x, const void* dy, const float* sc, const float* sh, int relu, const float* dp,
                        float* out, float* pivot, int B, int C, int HW, int dtype, int mode, hipStream_t st) {
  const int vec = 16 / (int)dtype_size(dtype);
  if (C % vec) return MRLA_EUNSUPPORTED;
  const int ns = nhwc_bn_splits(B, C, HW);
  const int cw = flat_cw(C, vec);
  if (cw) {
    const dim3 fgrid(C / cw, B * ns);
#define CALL(TT)                                                                                                       \
  if (mode) hipLaunchKernelGGL((nhwc_moments_flat_kernel<TT, 1>), fgrid, dim3(kThreads), 0, st, (const TT*)x,          \
                               (const TT*)dy, sc, sh, relu, dp, out, pivot, C, HW, ns, cw);                            \
  else      hipLaunchKernelGGL((nhwc_moments_flat_kernel<TT, 0>), fgrid, dim3(kThreads), 0, st, (const TT*)x,          \
                               (const TT*)dy, sc, sh, relu, dp, out, pivot, C, HW, ns, cw);
    MRLA_DISPATCH_N(dtype, CALL)
#undef CALL
    return hip_status(hipGetLastError());
  }
  if (dp) return MRLA_EUNSUPPORTED;                 // per-image scaling exists in the flat form only
  const dim3 grid((C + 63) / 64, B * ns);
#define CALL(TT)                                                                                                     \
  if (mode) hipLaunchKernelGGL((nhwc_moments_kernel<TT, 1>), grid, dim3(kThreads), 0, st, (const TT*)x, (const TT*)dy, \
                               sc, sh, relu, out, pivot, C, HW, ns);                                                 \
  else      hipLaunchKernelGGL((nhwc_moments_kernel<TT, 0>), grid, dim3(kThreads), 0, st, (const TT*)x, (const TT*)dy, \
                               sc, sh, relu, out, pivot, C, HW, ns);
  MRLA_DISPATCH_N(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

// mom[b,c,0] = sum_hw relu(round(sc*pre + sh) + o)  (sc null: relu(pre + o)); part: [b*nhwc_bn_splits, c, 2] workspace
int launch_nhwc_pool_fused(const void* pre, const float* sc, const float* sh, const void* o, float* part, float* mom,
                           int B, int C, int HW, int dtype, hipStream_t st) {
  const int vec = 16 / (int)dtype_size(dtype);
  const int cw = flat_cw(C, vec);
  if (!cw) return MRLA_EUNSUPPORTED;
  const int ns = nhwc_bn_splits(B, C, HW);
  const dim3 fgrid(C / cw, B * ns);
#define CALL(TT)                                                                                                 \
  hipLaunchKernelGGL((nhwc_moments_flat_kernel<TT, 2>), fgrid, dim3(kThreads), 0, st, (const TT*)pre, (const TT*)o, \
                     sc, sh, 0, (const float*)nullptr, part, (float*)nullptr, C, HW, ns, cw);
  MRLA_DISPATCH_N(dtype, CALL)
#undef CALL
  hipLaunchKernelGGL(nhwc_pool_finish_kernel, dim3((B * C + kThreads - 1) / kThreads), dim3(kThreads), 0, st, part, mom,
                     B * C, C, ns);
  return hip_status(hipGetLastError());
}

int launch_nhwc_affine(const void* x, const void* dy, const float* a, const float* sc, const float* sh, int relu,
                       void* out, int B, int C, int HW, int dtype, int bwd, hipStream_t st) {
  const size_t total = (size_t)B * C * HW;
  const int vec = 16 / (int)dtype_size(dtype);
  if (C % vec) return MRLA_EUNSUPPORTED;
  const size_t want = (total / vec + kThreads - 1) / kThreads;
  if ((C & (C - 1)) == 0 && C >= vec && C <= kThreads * vec) {        // power-of-two channel count: flat form
    const int iters = (int)std::max<size_t>(1, std::min<size_t>((want + 4095) / 4096, 64));
    const int fgrid = (int)((want + iters - 1) / iters);
#define CALL(TT)                                                                                                       \
  if (bwd) hipLaunchKernelGGL((nhwc_affine_flat_kernel<TT, true>), dim3(fgrid), dim3(kThreads), 0, st, (const TT*)x,   \
                              (const TT*)dy, a, sc, sh, relu, (TT*)out, total, C, iters);                              \
  else     hipLaunchKernelGGL((nhwc_affine_flat_kernel<TT, false>), dim3(fgrid), dim3(kThreads), 0, st, (const TT*)x,  \
                              (const TT*)dy, a, sc, sh, relu, (TT*)out, total, C, iters);
    MRLA_DISPATCH_N(dtype, CALL)
#undef CALL
    return hip_status(hipGetLastError());
  }
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
