// Shared pieces of the channels_last (NHWC) MRLA-light kernels: row gathers / scatters through the wave-private LDS
// scratch, the workgroup prologue and the launch geometry.  Included by light_nhwc.hip and light_nhwc_bwd.hip.
#pragma once
#include <algorithm>

#include "mrla_device.h"
#include "mrla_kernels.h"

namespace mrla {

constexpr int kS = 7;          // owned columns per strip
constexpr int kMaxStrips = 8;  // waves per workgroup (wider images loop strips inside a wave)

template <typename T>
__device__ __forceinline__ float ldpix(const T* __restrict__ img, int r, int col, int H, int W, int C, int c) {
  // wave-uniform predicate: all lanes look at the same pixel
  if (r < 0 || r >= H || col < 0 || col >= W) return 0.f;
  return to_f(img[((size_t)r * W + col) * C + c]);
}

// ---- wide row access -------------------------------------------------------------------------------------------
// A 2-byte-per-lane access costs the texture-addresser as much as a 16-byte one, so rows are fetched with 16 B per
// lane (lane = (pixel, 8-channel group): one wave-instruction = 8 pixels x 64 channels) and re-distributed to the
// LANE = CHANNEL compute mapping through a wave-private LDS scratch (ds_write_b128, then one ds_read_u16 per pixel).
// Needs C % 64 == 0 (the chunk is 64 real channels, 16-byte aligned); otherwise the kernels use ldpix().
// bytes per gather / scatter buffer: the widest row piece is kS+4 = 11 pixels x 64 channels, in whole 1 KiB loads
template <typename T> constexpr int scratch_bytes() { return 1024 * ((11 * 64 * (int)sizeof(T) + 1023) / 1024); }

// A row piece in flight: the 16-byte loads have been issued, nothing has been waited for yet.
template <typename T, int NPX>
struct RowLoad {
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int UPP = 64 / VEC;          // lanes per pixel
  static constexpr int PPL = 64 / UPP;          // pixels per wave-instruction
  static constexpr int NL = (NPX + PPL - 1) / PPL;
  u32x4 regs[NL];
  bool live;                                    // wave-uniform: row inside the image
};

// Per-strip lane addressing of a row piece: element offset of this lane's 16 bytes inside an image row and whether
// its pixel exists.  Computed once per strip; per row only a wave-uniform row pointer is added.
template <typename T, int NPX>
struct RowAddr {
  int off[RowLoad<T, NPX>::NL];
  bool ok[RowLoad<T, NPX>::NL];
};
template <typename T, int NPX>
__device__ __forceinline__ void make_row_addr(RowAddr<T, NPX>& a, int col0, int W, int C, int cbase, int lane) {
  typedef RowLoad<T, NPX> Q;
  const int px = lane / Q::UPP, part = lane - px * Q::UPP;
#pragma unroll
  for (int l = 0; l < Q::NL; ++l) {
    const int p = l * Q::PPL + px, col = col0 + p;
    a.ok[l] = p < NPX && col >= 0 && col < W;
    a.off[l] = a.ok[l] ? col * C + cbase + part * Q::VEC : 0;
  }
}

template <typename T, int NPX>
__device__ __forceinline__ void issue_row(RowLoad<T, NPX>& q, const T* __restrict__ img, int r, int H, int rowstride,
                                          const RowAddr<T, NPX>& a) {
  typedef RowLoad<T, NPX> Q;
  static_assert(Q::NL * 1024 <= scratch_bytes<T>(), "scratch too small");
  q.live = r >= 0 && r < H;
  const T* rowp = img + (size_t)(q.live ? r : 0) * rowstride;          // wave-uniform
#pragma unroll
  for (int l = 0; l < Q::NL; ++l) {
    q.regs[l] = (u32x4){0u, 0u, 0u, 0u};
    if (q.live && a.ok[l]) q.regs[l] = *reinterpret_cast<const u32x4*>(rowp + (unsigned)a.off[l]);
  }
}

template <typename T, int NPX>
__device__ __forceinline__ void finish_row(const RowLoad<T, NPX>& q, int lane, T* __restrict__ scratch, float (&out)[NPX]) {
  typedef RowLoad<T, NPX> Q;
  if (!q.live) {                                // wave-uniform
#pragma unroll
    for (int j = 0; j < NPX; ++j) out[j] = 0.f;
    return;
  }
  u32x4* s4 = reinterpret_cast<u32x4*>(scratch);
#pragma unroll
  for (int l = 0; l < Q::NL; ++l) s4[l * kWave + lane] = q.regs[l];
#pragma unroll
  for (int j = 0; j < NPX; ++j) out[j] = to_f(scratch[j * kWave + lane]);
}

template <typename T, int NPX>
__device__ __forceinline__ void gather_row(const T* __restrict__ img, int r, int col0, int H, int W, int C, int cbase,
                                           int lane, T* __restrict__ scratch, float (&out)[NPX]) {
  RowLoad<T, NPX> q;
  RowAddr<T, NPX> a;
  make_row_addr<T, NPX>(a, col0, W, C, cbase, lane);
  issue_row<T, NPX>(q, img, r, H, W * C, a);
  finish_row<T, NPX>(q, lane, scratch, out);
}

// lane = channel values v[j] of pixels col0 .. col0+npx-1 of row r -> global, 16 B per lane
template <typename T, int NPX>
__device__ __forceinline__ void scatter_row(T* __restrict__ img, int r, int col0, int npx, int W, int C, int cbase,
                                            int lane, T* __restrict__ scratch, const float (&v)[NPX]) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int UPP = 64 / VEC;
  constexpr int PPL = 64 / UPP;
  constexpr int NL = (NPX + PPL - 1) / PPL;
#pragma unroll
  for (int j = 0; j < NPX; ++j) scratch[j * kWave + lane] = from_f<T>(v[j]);
  const u32x4* s4 = reinterpret_cast<const u32x4*>(scratch);
  const int px = lane / UPP, part = lane - px * UPP;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const int p = l * PPL + px;
    if (p < npx)
      *reinterpret_cast<u32x4*>(img + ((size_t)r * W + col0 + p) * C + cbase + part * VEC) = s4[l * kWave + lane];
  }
}

__device__ __forceinline__ float conv_at(const float (&w)[9], const float* __restrict__ ra, const float* __restrict__ rb,
                                         const float* __restrict__ rc, int j) {
  float s = w[0] * ra[j];
  s = fmaf(w[1], ra[j + 1], s); s = fmaf(w[2], ra[j + 2], s);
  s = fmaf(w[3], rb[j], s); s = fmaf(w[4], rb[j + 1], s); s = fmaf(w[5], rb[j + 2], s);
  s = fmaf(w[6], rc[j], s); s = fmaf(w[7], rc[j + 1], s); s = fmaf(w[8], rc[j + 2], s);
  return s;
}

// Sum per-lane accumulators over the waves of the workgroup that hold the same channels -- waves v with equal v % wc (wc = 1:
// all of them) -- in a fixed order; result valid in waves 0 .. wc-1.
template <int K>
__device__ __forceinline__ void wg_reduce(float (&acc)[K], float* __restrict__ red, int lane, int wave, int nwaves,
                                          int wc = 1) {
  if (nwaves == wc) return;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) red[(wave * K + k) * kWave + lane] = acc[k];
  __syncthreads();
  if (wave < wc) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float s = 0.f;
      for (int v = wave; v < nwaves; v += wc) s += red[(v * K + k) * kWave + lane];
      acc[k] = s;
    }
  }
}

// Row readers / writers used by the kernels: WIDE -> 16-byte accesses through the wave-private scratch, else ldpix().
template <typename T, bool WIDE, int NPX>
__device__ __forceinline__ void read_row(const T* __restrict__ img, int r, int col0, int H, int W, int C, int cbase,
                                         int cc, int lane, T* __restrict__ scratch, float (&out)[NPX]) {
  if constexpr (WIDE) {
    gather_row<T, NPX>(img, r, col0, H, W, C, cbase, lane, scratch, out);
  } else {
#pragma unroll
    for (int j = 0; j < NPX; ++j) out[j] = ldpix(img, r, col0 + j, H, W, C, cc);
  }
}
template <typename T, bool WIDE, int NPX>
__device__ __forceinline__ void write_row(T* __restrict__ img, int r, int col0, int npx, int W, int C, int cbase, int c,
                                          bool cv, int lane, T* __restrict__ scratch, const float (&v)[NPX]) {
  if constexpr (WIDE) {
    scatter_row<T, NPX>(img, r, col0, npx, W, C, cbase, lane, scratch, v);
  } else {
#pragma unroll
    for (int j = 0; j < NPX; ++j)
      if (j < npx && cv) img[((size_t)r * W + col0 + j) * C + c] = from_f<T>(v[j]);
  }
}

#define MRLA_NHWC_PROLOGUE(NRED)                                                                          \
  extern __shared__ __align__(16) unsigned char smem_raw[];                                               \
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;    \
  float* red = reinterpret_cast<float*>(smem_raw);                                                        \
  T* scr = reinterpret_cast<T*>(smem_raw + (size_t)nwaves * (NRED) * kWave * sizeof(float) +              \
                                (size_t)wave * kScrBufs * scratch_bytes<T>());                                 \
  const int cbase = blockIdx.x * kWave;                                                                   \
  const int c = cbase + lane;                                                                             \
  const bool cv = c < C;                                                                                  \
  const int cc = cv ? c : C - 1;                                                                          \
  const int nstrips = (W + kS - 1) / kS;                                                                  \
  (void)red; (void)scr;
// scratch buffers per wave: one for gathers, one for scatters (LDS operations of a wave execute in order, so a buffer
// can be re-filled right after its previous contents were read back)
constexpr int kScrBufs = 2;
#define SCR(i) (scr + ((i) >= 3 ? 1 : 0) * (scratch_bytes<T>() / (int)sizeof(T)))

#define MRLA_DISPATCH_AO_N(TT, ACT, HASO, CALL)                          \
  if (ACT) { if (HASO) { CALL(TT, true, true); } else { CALL(TT, true, false); } } \
  else     { if (HASO) { CALL(TT, false, true); } else { CALL(TT, false, false); } }
#define MRLA_DISPATCH_T_N(DT, ACT, HASO, CALL)                       \
  switch (DT) {                                                      \
    case MRLA_F32:  MRLA_DISPATCH_AO_N(float, ACT, HASO, CALL) break;  \
    case MRLA_BF16: MRLA_DISPATCH_AO_N(bf16_t, ACT, HASO, CALL) break; \
    case MRLA_F16:  MRLA_DISPATCH_AO_N(f16_t, ACT, HASO, CALL) break;  \
    default: return MRLA_EINVAL;                                     \
  }

struct NhwcLaunch { dim3 grid, block; size_t lds; int BG; bool wide; };
static NhwcLaunch nhwc_launch(int B, int C, int W, int nred, int dtype) {
  NhwcLaunch L;
  const int nstrips = (W + kS - 1) / kS;
  const int nwaves = std::min(nstrips, kMaxStrips);
  L.BG = nhwc_images_per_group(B, C, 0);
  L.grid = dim3((C + kWave - 1) / kWave, (B + L.BG - 1) / L.BG);
  L.block = dim3(nwaves * kWave);
  L.wide = (C % kWave) == 0;
  const size_t sb = dtype == MRLA_F32 ? scratch_bytes<float>() : scratch_bytes<bf16_t>();
  L.lds = (size_t)nwaves * nred * kWave * sizeof(float) + (L.wide ? (size_t)nwaves * kScrBufs * sb : 0);
  return L;
}
template <typename K>
static hipError_t set_lds_n(K kernel, size_t bytes) {
  return lds_opt_in(reinterpret_cast<const void*>(kernel), bytes);
}

}  // namespace mrla
