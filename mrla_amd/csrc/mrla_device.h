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

// Moment record written by the forward statistics pass, per (image, channel): M_REC floats.  The sums over V and o are
// taken about the pivots (pV, po) stored beside them -- samples of V / o from the same plane -- so that the second
// moments do not cancel when |mean| >> sigma; producers without pivots store pV = po = 0 (raw sums).
//   [M_SX] sum x   [M_SV] sum (V-pV)   [M_SO] sum (o-po)   [M_SVV] sum (V-pV)^2   [M_SVO] sum (V-pV)(o-po)
//   [M_SOO] sum (o-po)^2   [M_PV] pV   [M_PO] po
enum { M_SX = 0, M_SV = 1, M_SO = 2, M_SVV = 3, M_SVO = 4, M_SOO = 5, M_N = 6, M_PV = 6, M_PO = 7, M_REC = 8 };
// Moment slots written by the backward statistics pass.
enum { D_D = 0, D_DV = 1, D_DO = 2, D_N = 3 };
// token path (tokens*.hip): slots of the per-(image, channel) parameter partials that the gate backward completes with
// the pooled-descriptor gradient (dlnx_w += dy * sum xhat, dlnx_b += dy * pixels), and the record size
constexpr int kTokPartLnxW = 10, kTokPartLnxB = 11, kTokPartHat = 14, kTokParts = 15;

// Raw moments of V and o over the n pixels of a plane, in double, from a (pivot-shifted) moment record.
struct RawMoments { double sv, so, svv, svo, soo; };
__device__ __forceinline__ RawMoments raw_moments(const float* __restrict__ m, double n) {
  const double pv = m[M_PV], po = m[M_PO], a = m[M_SV], b = m[M_SO];
  RawMoments r;
  r.sv = a + n * pv;
  r.so = b + n * po;
  r.svv = (double)m[M_SVV] + 2.0 * pv * a + n * pv * pv;
  r.svo = (double)m[M_SVO] + pv * b + po * a + n * pv * po;
  r.soo = (double)m[M_SOO] + 2.0 * po * b + n * po * po;
  return r;
}
// sums of a record about (pv, po) over n pixels -> the same sums about (qv, qo)
__device__ __forceinline__ void rebase_moments(float (&a)[M_N], float n, float pv, float po, float qv, float qo) {
  const float dv = pv - qv, d_o = po - qo;
  a[M_SVV] += 2.f * dv * a[M_SV] + n * dv * dv;
  a[M_SOO] += 2.f * d_o * a[M_SO] + n * d_o * d_o;
  a[M_SVO] += d_o * a[M_SV] + dv * a[M_SO] + n * dv * d_o;
  a[M_SV] += n * dv;
  a[M_SO] += n * d_o;
}

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

// ---- cross-lane sums with DPP (no LDS traffic, one VALU op per step) ------------------------------------------
// row_shr:n shifts within a 16-lane row (zero fill), row_bcast:15 / :31 carry a row's last lane into the next
// row(s); after the steps the LAST lane of every aligned power-of-two segment holds the segment's sum.
#define MRLA_DPP_ADD(v, ctrl, row_mask, bank_mask) \
  (v) += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), (row_mask), (bank_mask), true))

// Sum over aligned segments of `ws` lanes (ws = 8, 16, 32 or 64; smaller segments use the generic path).
// The result is valid in the last lane of each segment: (lane & (ws-1)) == ws-1.  All 64 lanes must call it.
__device__ __forceinline__ float seg_sum(float v, int lane, int ws) {
  if (ws >= 8) {
    MRLA_DPP_ADD(v, 0x111, 0xf, 0xf);                       // row_shr:1
    MRLA_DPP_ADD(v, 0x112, 0xf, 0xf);                       // row_shr:2
    MRLA_DPP_ADD(v, 0x114, 0xf, 0xf);                       // row_shr:4
    if (ws >= 16) MRLA_DPP_ADD(v, 0x118, 0xf, 0xf);         // row_shr:8
    if (ws >= 32) MRLA_DPP_ADD(v, 0x142, 0xa, 0xf);         // row_bcast:15 -> rows 1, 3
    if (ws >= 64) MRLA_DPP_ADD(v, 0x143, 0xc, 0xf);         // row_bcast:31 -> rows 2, 3
    return v;
  }
  for (int off = 1; off < ws; off <<= 1) {                  // ws = 1, 2, 4: xor butterfly, every lane gets the sum
    const float t = __shfl_xor(v, off, kWave);
    v += t;
  }
  return v;
}

// Full-wave sum, returned in every lane.
__device__ __forceinline__ float wave_sum(float v) {
  MRLA_DPP_ADD(v, 0x111, 0xf, 0xf);
  MRLA_DPP_ADD(v, 0x112, 0xf, 0xf);
  MRLA_DPP_ADD(v, 0x114, 0xf, 0xf);
  MRLA_DPP_ADD(v, 0x118, 0xf, 0xf);
  MRLA_DPP_ADD(v, 0x142, 0xa, 0xf);
  MRLA_DPP_ADD(v, 0x143, 0xc, 0xf);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// erf to < 1 ulp without the library call: two minimax polynomials (|a| <= 0.927734375: a + a*P(a^2); beyond:
// 1 - exp(Q(|a|))), both evaluated and selected -- lanes of a wave straddle the boundary anyway.  ~22 VALU instructions
// where ocml's erff + expf cost the token kernels >100 per GELU (they were the bulk of those kernels' instruction count).
__device__ __forceinline__ float erf_f(float a) {
  const float t = fminf(fabsf(a), 10.0f), s = t * t;       // erf(10) == 1 in fp32
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float q = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, q);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  const float big = copysignf(1.0f - __expf(r), a);
  float p = -5.96761703e-4f;
  p = fmaf(p, s, 4.99119423e-3f);
  p = fmaf(p, s, -2.67681349e-2f);
  p = fmaf(p, s, 1.12819925e-1f);
  p = fmaf(p, s, -3.76125336e-1f);
  p = fmaf(p, s, 1.28379166e-1f);
  const float small = fmaf(p, a, a);
  return t > 0.927734375f ? big : small;
}

// exact (erf) GELU of nn.GELU() and its derivative
__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.0f + erf_f(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float u) {
  const float cdf = 0.5f * (1.0f + erf_f(u * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * u * u);
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

// Asynchronous HBM -> LDS copy of a slab with LDS-DMA (`global_load_lds_dwordx4`: 1 KiB per wave-instruction, no
// VGPR staging).  The caller later executes wait_async_copies() + __syncthreads() before reading `dst`.
// Falls back to the synchronous register-staged copy when the slab is not 16-byte granular / aligned.
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <typename T>
__device__ __forceinline__ void slab_prefetch(T* __restrict__ dst, const T* __restrict__ src, int n, int tid) {
  const int bytes = n * (int)sizeof(T);
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (bytes & 15) == 0) {
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    const char* s = reinterpret_cast<const char*>(src);
    char* d = reinterpret_cast<char*>(dst);
    const int nchunks = (bytes + 1023) >> 10;
    for (int ch = wave; ch < nchunks; ch += kWaves) {
      const int off = (ch << 10) + (lane << 4);
      if (off < bytes)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(s + off),
                                         (lds_void_ptr)(d + (ch << 10)), 16, 0, 0);
    }
  } else {
    slab_load(dst, src, n, tid);
  }
}

__device__ __forceinline__ void wait_async_copies() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

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
