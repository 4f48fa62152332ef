// 1x1 stride-1 convolution with a wide output (N % 256 == 0) as a streaming MFMA GEMM, BatchNorm-statistics epilogue:
//   Y[M, N] = X[M, K] * W[N, K]^T (+ A[M, N])      K in {64, 128, 256}; bf16 in/out, fp32 accumulate
//   part[row, n, 0..3] = per-workgroup moment record of the bf16-ROUNDED outputs of channel n (MRLA_GEMM_MOMENTS):
//                        sum (y - p), sum (y - p)^2, the pivot p (the workgroup's first output of the channel), pixel count
// Reference: the bottleneck's conv3 / bn3 (resnet/models/resnet_mrla_light.py:100-101) and, with the operands swapped
// by the caller (x = dY, w = W^T), the input gradient of conv1 (:93) -- there `A` is the gradient that reaches the
// block input through the shortcut, so the autograd accumulation (a separate 3N elementwise pass) happens in this
// epilogue.
//
// conv1x1.hip fetches every lane's X fragment straight from global memory.  For wide outputs that re-fetches X once
// per 64 output channels, and with K = 256 (512-byte pixel rows, 32 bytes used per row and instruction) the L1 thrashes:
// 1.7 TB/s on 256 -> 1024.  Here:
//   * a workgroup = 8 waves = 256 output channels; wave w keeps ITS 32 rows of W in registers for the whole kernel
//     (A operand, K/4 VGPRs) and walks a contiguous range of 32-pixel blocks;
//   * the X block [32 px][K] goes global -> LDS with LDS-DMA (16 B per lane, whole rows, every byte fetched once per
//     256 output channels); a ring of blocks is in flight; the B operand is one ds_read_b128 per k-step (chunk c of
//     pixel row r sits at chunk position c ^ f(r): conflict-free for the 16-lane groups of ds_read_b128);
//   * workgroup ids are re-mapped so that the channel groups of one pixel range run side by side on one XCD (L2 hits);
//   * the 32 x 256 bf16 output tile is assembled in LDS and leaves as whole 512-byte rows; with `A` the DMA drops the
//     addend tile into that staging buffer first and the epilogue adds in fp32 before the one rounding;
//   * one barrier per pixel block; LDS traffic of the pipeline is inline asm with explicit waits (the compiler would put
//     `s_waitcnt vmcnt(0)` in front of every LDS access it sees after an LDS-DMA load).
#include <algorithm>

#include "mrla_device.h"
#include "mrla_kernels.h"

namespace mrla {
namespace {

typedef __bf16 cw_bf16x8 __attribute__((ext_vector_type(8)));
typedef float cw_f32x16 __attribute__((ext_vector_type(16)));

#define MRLA_CW_FLAGS 0x00020000          /* raw buffer descriptor word 3 (as nhwc_rows.h) */
constexpr int kCwWaves = 8;
constexpr int kCwTN = 256;                // output channels per workgroup
constexpr int kOutB = 32 * kCwTN * 2;     // bytes of an output / addend tile [32 px][256 ch]

template <int KS, bool ADD>
struct CwGeo {
  static constexpr int K = KS * 16;
  static constexpr int XB = 32 * K * 2;                         // X block bytes (4 / 8 / 16 KB)
  static constexpr int NIX = XB / 1024 >= kCwWaves ? XB / 1024 / kCwWaves : 1;   // X DMA instructions per wave and unit
  static constexpr int NIA = ADD ? kOutB / 1024 / kCwWaves : 0;                  // addend DMA instructions (2)
  static constexpr int NI = NIX + NIA;
  // ring depth: ~64 KB of X in flight without the addend, as much as LDS allows with it
  static constexpr int UST = ADD ? (KS == 16 ? 3 : KS == 8 ? 4 : 5) : (KS == 16 ? 5 : KS == 8 ? 8 : 16);
  static constexpr int OST = ADD ? UST : 2;                      // output tiles (with the addend: one per ring slot)
  static constexpr int kXRing = UST * XB;
  static constexpr int kDummy = kXRing + OST * kOutB;            // 1 KB target of the idle waves' dummy DMA (K = 64)
  static constexpr int kLds = kDummy + 1024;
};

__device__ __forceinline__ unsigned cw_lds_addr(const void* p) {
  return (unsigned)(size_t)((__attribute__((address_space(3))) const char*)p);
}
__device__ __forceinline__ void cw_read16(u32x4& v, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
}
__device__ __forceinline__ void cw_write16(unsigned addr, const u32x4& v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int N>
__device__ __forceinline__ void cw_fence(u32x4& v, bool wait) {
  if (wait) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N) : "memory");
  else asm volatile("" : "+v"(v)::"memory");
}

// chunk swizzle of X rows (CPR 16-byte chunks per row) and of output-tile rows (32 chunks)
template <int CPR>
__device__ __forceinline__ int cw_swz(int row) { return CPR >= 16 ? (row & 15) : ((row >> 1) & 7); }
__device__ __forceinline__ int cw_oswz(int row) { return row & 7; }

template <int KS, bool MOM, bool ADD>
__global__ __launch_bounds__(kCwWaves* kWave) void conv1x1_wide_kernel(const bf16_t* __restrict__ X,
                                                                       const bf16_t* __restrict__ W,
                                                                       const bf16_t* __restrict__ A, bf16_t* __restrict__ Y,
                                                                       float* __restrict__ part, int M, int N,
                                                                       int units_per_wg, int nsplits, int rows_total) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef CwGeo<KS, ADD> G;
  constexpr int K = G::K, CPR = K / 8, UST = G::UST;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware order (see conv1x1_wgrad.hip): the channel groups of one pixel range are neighbours on one XCD
  const int groups = N / kCwTN;
  const int per = gridDim.x >> 3;
  const int vid = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (vid >= groups * nsplits) return;
  const int split = vid / groups, cg = vid - split * groups;
  const int nblk = (M + 31) / 32;
  const int u_begin = split * units_per_wg;
  const int nun = min(units_per_wg, nblk - u_begin);            // >= 1 by construction of the grid
  const int n0 = cg * kCwTN + wave * 32;

  // ---- this wave's 32 rows of W, for the whole kernel ----
  cw_bf16x8 wf[KS];
  {
    const u32x4* wp = reinterpret_cast<const u32x4*>(W + (size_t)(n0 + r) * K + h * 8);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[ks] = __builtin_bit_cast(cw_bf16x8, wp[ks * 2]);
  }

  // ---- DMA plan ----
  const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(X), 0, (int)((size_t)M * K * 2), MRLA_CW_FLAGS);
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(ADD ? A : X), 0,
                                                     ADD ? (int)((size_t)M * N * 2) : 0, MRLA_CW_FLAGS);
  unsigned voffX[G::NIX], voffA[ADD ? G::NIA : 1];
  constexpr bool kIdleWaves = G::XB / 1024 < kCwWaves;          // K = 64: four real X instructions, waves 4-7 issue a dummy
  const bool idle = kIdleWaves && wave >= G::XB / 1024;
#pragma unroll
  for (int i = 0; i < G::NIX; ++i) {
    const int u = wave + kCwWaves * i, row = u * (64 / CPR) + lane / CPR, cp = lane % CPR;
    voffX[i] = idle ? 0x80000000u
                    : (unsigned)(((size_t)u_begin * 32 + row) * K * 2 + ((cp ^ cw_swz<CPR>(row)) << 4));
  }
  if (ADD) {
#pragma unroll
    for (int i = 0; i < G::NIA; ++i) {
      const int u = wave + kCwWaves * i, row = u * 2 + lane / 32, cp = lane % 32;
      voffA[i] = (unsigned)(((size_t)u_begin * 32 + row) * N * 2 + cg * (kCwTN * 2) + ((cp ^ cw_oswz(row)) << 4));
    }
  }
  const unsigned advX = 32u * K * 2, advA = 32u * (unsigned)N * 2;
  int issued = 0;
  auto issue = [&](int slot) {
    const unsigned kill = issued++ < nun ? 0u : 0x80000000u;     // past the range: out of bounds, no memory traffic
#pragma unroll
    for (int i = 0; i < G::NIX; ++i) {
      unsigned char* dst = idle ? smem_raw + G::kDummy : smem_raw + slot * G::XB + (wave + kCwWaves * i) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_ptr)dst, 16, voffX[i] | kill, 0, 0, 0);
      voffX[i] += advX;
    }
    if (ADD) {
#pragma unroll
      for (int i = 0; i < G::NIA; ++i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_ptr)(smem_raw + G::kXRing + slot * kOutB + (wave + kCwWaves * i) * 1024),
                                                 16, voffA[i] | kill, 0, 0, 0);
        voffA[i] += advA;
      }
    }
  };

  // ---- addresses ----
  const unsigned lds0 = cw_lds_addr(smem_raw);
  const int fx = cw_swz<CPR>(r);
  const unsigned xrow = lds0 + r * (K * 2);                                  // + slot*XB + (((2ks+h) ^ fx) << 4)
  // this lane's two 16-byte pieces of its pixel's output row: channels wave*32 + h*8 .. and wave*32 + 16 + h*8 ..
  const int oc0 = wave * 4 + h, oc1 = oc0 + 2;
  const unsigned orow = lds0 + G::kXRing + r * (kCwTN * 2);
  const unsigned opc0 = orow + ((oc0 ^ cw_oswz(r)) << 4), opc1 = orow + ((oc1 ^ cw_oswz(r)) << 4);
  // cooperative store: thread -> (row, chunk) of the tile, two rows 16 apart
  const int srow = threadIdx.x >> 5, schunk = threadIdx.x & 31;
  const unsigned sld0 = lds0 + G::kXRing + srow * (kCwTN * 2) + ((schunk ^ cw_oswz(srow)) << 4);
  const unsigned sld1 = sld0 + 16 * (kCwTN * 2);                             // (row + 16: same swizzle, (row+16)&7 == row&7)
  bf16_t* yrow = Y + ((size_t)u_begin * 32 + srow) * N + cg * kCwTN + schunk * 8;

  // BatchNorm moments of the rounded outputs, per lane (register = channel, lane = pixel), taken about a pivot: the
  // workgroup's first output of the channel (lane r = 0 of the half-wave in the first unit).  The one-pass variance
  // E[(y-p)^2] - E[y-p]^2 then stays well conditioned when |mean| >> sigma; the per-channel kernel merges the
  // workgroups' records by re-basing them (in double).
  float s1[MOM ? 16 : 1], s2[MOM ? 16 : 1], pv[MOM ? 16 : 1];
  if (MOM) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { s1[i] = 0.f; s2[i] = 0.f; pv[i] = 0.f; }
  }
  const bool ragged_tail = (u_begin + nun) * 32 > M;             // the last unit of this range has pixels past the end

  // ---- pipeline: UST-1 units issued ahead; unit u+1 is waited for in front of unit u's barrier ----
  constexpr int KC = KS < 8 ? KS : 8, NCH = KS / KC;
  constexpr int C_EARLY = (UST - 3) * G::NI > 0 ? (UST - 3) * G::NI : 0;   // DMA instructions allowed in flight
  constexpr int C_STEADY = C_EARLY + (UST - 2) * 2;                        // ... plus the row stores issued since
  static_assert(UST >= 3 && C_STEADY < 64, "vmcnt immediate");
#pragma unroll
  for (int j = 0; j < UST - 1; ++j) issue(j);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((UST - 2) * G::NI) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  int slot = 0, nxt = UST - 1;
  for (int u = 0; u < nun; ++u) {
    const unsigned xs = xrow + slot * G::XB;
    const unsigned os = (ADD ? slot : (u & 1)) * kOutB;
    cw_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    u32x4 xf[2][KC];
#pragma unroll
    for (int ks = 0; ks < KC; ++ks) cw_read16(xf[0][ks], xs + (((2 * ks + h) ^ fx) << 4));
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc) {
      const int cur = kc & 1;
      if (kc + 1 < NCH) {
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) cw_read16(xf[cur ^ 1][ks], xs + (((2 * ((kc + 1) * KC + ks) + h) ^ fx) << 4));
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) cw_fence<KC>(xf[cur][ks], ks == 0);
      } else {
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) cw_fence<0>(xf[cur][ks], ks == 0);
      }
#pragma unroll
      for (int ks = 0; ks < KC; ++ks)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kc * KC + ks], __builtin_bit_cast(cw_bf16x8, xf[cur][ks]), acc, 0, 0, 0);
    }
    // lane = pixel r; register e = channel 8*(e/4) + 4*h + e%4 of the wave's 32.  Round pairs of neighbours to bf16.
    // (in piece order after the exchange below: piece g = channels 16g + 8h .. 16g + 8h + 7, 16 bytes of the pixel's row)
    unsigned p[8];
    float av[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) av[i] = acc[i];
    if (ADD) {
      // the addend tile is in LDS in piece order: bring the fp32 values into piece order first (swap register quads
      // between the half-waves), add, and round once
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(av[8 * g + q]), __float_as_uint(av[8 * g + 4 + q]), false, false);
          av[8 * g + q] = __uint_as_float(sw[0]);
          av[8 * g + 4 + q] = __uint_as_float(sw[1]);
        }
      u32x4 a0, a1;
      cw_read16(a0, opc0 + os);
      cw_read16(a1, opc1 + os);
      cw_fence<0>(a0, true);
      cw_fence<0>(a1, false);
      const unsigned aw[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        av[2 * i] += __uint_as_float(aw[i] << 16);
        av[2 * i + 1] += __uint_as_float(aw[i] & 0xffff0000u);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
      bf16x2 pr;
      pr[0] = from_f<bf16_t>(av[2 * i]);
      pr[1] = from_f<bf16_t>(av[2 * i + 1]);
      p[i] = __builtin_bit_cast(unsigned, pr);
    }
    if (MOM) {
      float val[16];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        val[2 * i] = __uint_as_float(p[i] << 16);
        val[2 * i + 1] = __uint_as_float(p[i] & 0xffff0000u);
      }
      if (u == 0) {              // (uniform) the pivots: this half-wave's pixel 0, which always exists
#pragma unroll
        for (int i = 0; i < 16; ++i) pv[i] = __shfl(val[i], h * 32, kWave);
      }
      if (!(ragged_tail && u == nun - 1)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float dlt = val[i] - pv[i];
          s1[i] += dlt;
          s2[i] = fmaf(dlt, dlt, s2[i]);
        }
      } else {                   // (uniform) the one unit with pixels past the end: those do not count
        const bool live = (u_begin + u) * 32 + r < M;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float dlt = live ? val[i] - pv[i] : 0.f;
          s1[i] += dlt;
          s2[i] = fmaf(dlt, dlt, s2[i]);
        }
      }
    }
    if (!ADD) {
      // half 0 holds channels {0-3, 8-11, 16-19, 24-27}, half 1 the other four groups; after the swaps half 0 holds
      // {0-7, 16-23} and half 1 {8-15, 24-31}: two 16-byte pieces per lane
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const auto sw = __builtin_amdgcn_permlane32_swap(p[4 * g + q], p[4 * g + 2 + q], false, false);
          p[4 * g + q] = sw[0];
          p[4 * g + 2 + q] = sw[1];
        }
    }
    cw_write16(opc0 + os, (u32x4){p[0], p[1], p[2], p[3]});
    cw_write16(opc1 + os, (u32x4){p[4], p[5], p[6], p[7]});
    // hand-over: my part of unit u+1 has landed, my tile pieces are written; after the barrier everybody's are, and
    // unit u's X block is free for the DMA of unit u+UST-1
    if (u < UST) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C_EARLY) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C_STEADY) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(nxt);
    nxt = nxt + 1 == UST ? 0 : nxt + 1;
    slot = slot + 1 == UST ? 0 : slot + 1;
    // whole 512-byte rows out: 32 lanes per pixel row, rows srow and srow + 16
    {
      u32x4 v0, v1;
      cw_read16(v0, sld0 + os);
      cw_read16(v1, sld1 + os);
      cw_fence<0>(v0, true);
      cw_fence<0>(v1, false);
      const int pix = (u_begin + u) * 32 + srow;
      bf16_t* yp = yrow + (size_t)u * 32 * N;
      if (pix < M) *reinterpret_cast<u32x4*>(yp) = v0;
      if (pix + 16 < M) *reinterpret_cast<u32x4*>(yp + (size_t)16 * N) = v1;
    }
  }

  if (MOM) {
    // rows beyond the workgroups' own only exist to make the row count divide M: empty records (written by the first range)
    if (split == 0) {
      for (int i = threadIdx.x; i < (rows_total - nsplits) * kCwTN * 4; i += kCwWaves * kWave) {
        const int row = nsplits + i / (kCwTN * 4), j = i % (kCwTN * 4);
        part[((size_t)row * N + cg * kCwTN) * 4 + j] = 0.f;
      }
    }
    const float npix = (float)(min(M, (u_begin + nun) * 32) - u_begin * 32);
    // register e of half h = channel: see the piece layout the values were accumulated in
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float a = s1[i], b = s2[i];
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, kWave);
        b += __shfl_xor(b, off, kWave);
      }
      if (r == 0) {
        const int ch = ADD ? ((i < 8 ? 0 : 16) + h * 8 + (i & 7)) : ((i & 3) + 8 * (i >> 2) + 4 * h);
        float* dst = part + ((size_t)split * N + n0 + ch) * 4;
        dst[0] = a;
        dst[1] = b;
        dst[2] = pv[i];
        dst[3] = npix;
      }
    }
  }
#endif
}

struct CwPlan {
  int groups = 0, splits = 0, units_per_wg = 0, rows = 0;
};

CwPlan cw_plan(int M, int K, int N) {
  CwPlan p;
  if (M <= 0 || (K != 64 && K != 128 && K != 256) || N % kCwTN || (size_t)M * std::max(N, K) * 2 >= (size_t)1 << 31) return p;
  const int groups = N / kCwTN, nblk = (M + 31) / 32;
  const int want = std::max(1, std::min(nblk, (256 + groups - 1) / groups));          // one workgroup per CU
  const int upw = (nblk + want - 1) / want;
  const int splits = (nblk + upw - 1) / upw;
  // the statistics kernel takes (rows, M / rows): pad the partial buffer with zero rows up to the next divisor of M
  int rows = splits;
  while (rows <= 2 * splits + 64 && M % rows) ++rows;
  if (M % rows) return p;
  p.groups = groups; p.splits = splits; p.units_per_wg = upw; p.rows = rows;
  return p;
}

template <int KS, bool MOM, bool ADD>
int cw_launch(const CwPlan& p, const void* x, const void* w, const void* a, void* y, float* part, int M, int N, hipStream_t st) {
  typedef CwGeo<KS, ADD> G;
  static_assert(G::kLds <= 160 * 1024, "LDS");
  if (lds_opt_in(reinterpret_cast<const void*>(conv1x1_wide_kernel<KS, MOM, ADD>), G::kLds) != hipSuccess) return MRLA_EHIP;
  hipLaunchKernelGGL((conv1x1_wide_kernel<KS, MOM, ADD>), dim3((p.groups * p.splits + 7) / 8 * 8), dim3(kCwWaves * kWave),
                     G::kLds, st, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)a, (bf16_t*)y, part, M, N,
                     p.units_per_wg, p.splits, p.rows);
  return hip_status(hipGetLastError());
}

}  // namespace

int conv1x1_wide_rows(int M, int K, int N) {
  const CwPlan p = cw_plan(M, K, N);
  return p.groups ? p.rows : MRLA_EUNSUPPORTED;
}

// {32-pixel blocks per workgroup, LDS ring depth in blocks, workgroups, moment rows}
int conv1x1_wide_plan(int M, int K, int N, int add, int* out) {
  const CwPlan p = cw_plan(M, K, N);
  if (!p.groups) return MRLA_EUNSUPPORTED;
  const int ks = K / 16;
  out[0] = p.units_per_wg;
  out[1] = add ? (ks == 16 ? CwGeo<16, true>::UST : ks == 8 ? CwGeo<8, true>::UST : CwGeo<4, true>::UST)
               : (ks == 16 ? CwGeo<16, false>::UST : ks == 8 ? CwGeo<8, false>::UST : CwGeo<4, false>::UST);
  out[2] = p.groups * p.splits;
  out[3] = p.rows;
  return MRLA_OK;
}

int launch_conv1x1_wide(const void* x, const void* w, const void* addend, void* y, float* part, int M, int K, int N,
                        hipStream_t st) {
  const CwPlan p = cw_plan(M, K, N);
  if (!p.groups) return MRLA_EUNSUPPORTED;
#define MRLA_CW_CALL(KS)                                                                         \
  {                                                                                              \
    if (addend) return part ? cw_launch<KS, true, true>(p, x, w, addend, y, part, M, N, st)      \
                            : cw_launch<KS, false, true>(p, x, w, addend, y, part, M, N, st);    \
    return part ? cw_launch<KS, true, false>(p, x, w, addend, y, part, M, N, st)                 \
                : cw_launch<KS, false, false>(p, x, w, addend, y, part, M, N, st);               \
  }
  switch (K) {
    case 64: MRLA_CW_CALL(4)
    case 128: MRLA_CW_CALL(8)
    case 256: MRLA_CW_CALL(16)
    default: return MRLA_EUNSUPPORTED;
  }
#undef MRLA_CW_CALL
}

}  // namespace mrla
