// 1x1 stride-1 convolution of a channels_last bf16 activation as an MFMA GEMM with a BatchNorm-statistics epilogue:
//   Y[M, N] = X[M, K] * W[N, K]^T      (M = b*h*w pixels, K = in-channels, N = out-channels; both operands K-contiguous)
//   part[row, n, 0..1] = per-workgroup partial (sum, sum of squares) of the bf16-ROUNDED outputs, channel n
// Reference: the bottleneck's conv1 / bn1 and conv3 / bn3 (resnet/models/resnet_mrla_light.py:93-102): the statistics pass
// of the BatchNorm that follows the convolution (1N read of the large conv3 output) disappears into this epilogue, and
// the output is written exactly once (MIOpen's implicit-GEMM solver memsets it first).
//
// These GEMMs are memory-bound in the early stages (K = 64 .. 128: 50-100 flop/byte) and skinny everywhere, so the
// kernel is organised around the streams, not around a big LDS tile:
//   * MFMA 32x32x16 bf16 with A = W tile (rows = out-channels), B = X tile (columns = pixels): each lane's B fragment is
//     16 contiguous bytes of ITS pixel's row of X, loaded straight from global memory into registers (no LDS, no
//     transposition), kept for all out-channels of the wave and double-buffered against the next pixel block;
//   * the W slice of the workgroup stays in LDS for the whole kernel (rows padded by 16 B: conflict-free b128 reads);
//     a workgroup is persistent over pixel blocks, its 8 waves split (pixel blocks) x (64-channel pairs);
//   * the accumulator tile has the pixel on the lane and 16 channels in registers; after rounding to bf16 a
//     v_permlane32_swap between the two half-waves leaves every lane with 2 x 16 contiguous bytes of its pixel's
//     output row: 16-byte stores without an LDS round trip;
//   * the BatchNorm moments are per-lane running sums over the wave's pixels (register = channel), reduced across
//     lanes once at the end of the kernel and written as one partial row per (workgroup, pixel-wave).
#include <algorithm>

#include "mrla_device.h"
#include "mrla_kernels.h"

namespace mrla {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kGemmWaves = 8;

// channel of accumulator register `reg` inside its 32-channel tile, for lane half h
__device__ __forceinline__ int acc_channel(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

template <int KS, bool MOM>
__global__ __launch_bounds__(kGemmWaves * kWave) void conv1x1_fwd_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, bf16_t* __restrict__ Y, float* __restrict__ part,
    int M, int N, int NS, int WN) {
  constexpr int K = KS * 16;
  constexpr int ROWB = K * 2 + 16;                   // padded LDS row of W, bytes
  constexpr int KC = KS < 16 ? KS : 16;              // k-steps whose X fragments are in registers at a time
  constexpr int NCH = KS / KC;
  constexpr bool DB = KS <= 4;                       // double-buffer the X fragments when they are small
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int r = lane & 31, h = lane >> 5;
  const int WM = kGemmWaves / WN, wn = wave % WN, wm = wave / WN;
  const int n_slice0 = blockIdx.y * NS;

  // stage the W slice [NS][K] into LDS (16-byte pieces, coalesced)
  {
    constexpr int PPR = K / 8;                       // 16-byte pieces per row
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(W + (size_t)n_slice0 * K);
    for (int i = threadIdx.x; i < NS * PPR; i += kGemmWaves * kWave) {
      const int row = i / PPR, pc = i - row * PPR;
      *reinterpret_cast<u32x4*>(smem_raw + (size_t)row * ROWB + pc * 16) = wsrc[i];
    }
  }
  __syncthreads();

  float s1[2][16], s2[2][16];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) { s1[t][i] = 0.f; s2[t][i] = 0.f; }

  const int nblk = M / 32;
  const int stride = gridDim.x * WM;
  int blk = blockIdx.x * WM + wm;
  u32x4 xf[KC], xn[DB ? KC : 1];
  // K chunk `kc` (KC k-steps) of the X fragments of pixel block b_
  auto load_x = [&](u32x4 (&dst)[KC], int b_, int kc) {
    const u32x4* xp = reinterpret_cast<const u32x4*>(X + ((size_t)b_ * 32 + r) * K + kc * KC * 16 + h * 8);
#pragma unroll
    for (int ks = 0; ks < KC; ++ks) dst[ks] = xp[ks * 2];
  };
  if (blk < nblk) load_x(xf, blk, 0);
  for (; blk < nblk; blk += stride) {
    if constexpr (DB) {
      if (blk + stride < nblk) load_x(reinterpret_cast<u32x4(&)[KC]>(xn), blk + stride, 0);
    }
    const size_t m = (size_t)blk * 32 + r;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll 1
    for (int kc = 0; kc < NCH; ++kc) {             // (not unrolled: the chunks' fragments must not be live together)
      if (kc > 0) load_x(xf, blk, kc);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const unsigned char* wrow = smem_raw + (size_t)(wn * 64 + t * 32 + r) * ROWB + kc * KC * 32 + h * 16;
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) {
          const u32x4 wf = *reinterpret_cast<const u32x4*>(wrow + ks * 32);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf),
                                                           __builtin_bit_cast(bf16x8, xf[ks]), acc[t], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int nloc = wn * 64 + t * 32;                               // first channel of the tile inside the slice
      // round to bf16 (pairs of neighbouring channels), statistics of the rounded values
      unsigned p[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 pr;
        pr[0] = from_f<bf16_t>(acc[t][2 * i]);
        pr[1] = from_f<bf16_t>(acc[t][2 * i + 1]);
        p[i] = __builtin_bit_cast(unsigned, pr);
        if (MOM) {
          const float lo = __uint_as_float(p[i] << 16), hi = __uint_as_float(p[i] & 0xffff0000u);
          s1[t][2 * i] += lo;     s2[t][2 * i] = fmaf(lo, lo, s2[t][2 * i]);
          s1[t][2 * i + 1] += hi; s2[t][2 * i + 1] = fmaf(hi, hi, s2[t][2 * i + 1]);
        }
      }
      // lane half 0 holds channels {0-3, 8-11, 16-19, 24-27} of its pixel, half 1 the other four groups; after the
      // swaps half 0 holds {0-7, 16-23} and half 1 {8-15, 24-31}: two 16-byte pieces per lane
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const auto sw = __builtin_amdgcn_permlane32_swap(p[4 * g + q], p[4 * g + 2 + q], false, false);
          p[4 * g + q] = sw[0];
          p[4 * g + 2 + q] = sw[1];
        }
      }
#endif
      bf16_t* yp = Y + m * N + n_slice0 + nloc + h * 8;
      *reinterpret_cast<u32x4*>(yp) = (u32x4){p[0], p[1], p[2], p[3]};
      *reinterpret_cast<u32x4*>(yp + 16) = (u32x4){p[4], p[5], p[6], p[7]};
    }
    if constexpr (DB) {
#pragma unroll
      for (int ks = 0; ks < KC; ++ks) xf[ks] = xn[ks];
    } else {
      if (blk + stride < nblk) load_x(xf, blk + stride, 0);
    }
  }

  if (MOM) {
    // sum over the 32 pixel-lanes of each half (fixed order), one partial row per (workgroup, pixel-wave)
    float* dst = part + ((size_t)(blockIdx.x * WM + wm) * N + n_slice0 + wn * 64) * 2;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float a = s1[t][i], b = s2[t][i];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
          a += __shfl_xor(a, off, kWave);
          b += __shfl_xor(b, off, kWave);
        }
        if (r == 0) {
          const int ch = t * 32 + acc_channel(i, h);
          dst[ch * 2 + 0] = a;
          dst[ch * 2 + 1] = b;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------
// geometry: N slice per workgroup (LDS), waves along N, persistent grid
// ------------------------------------------------------------------------------------------------
struct GemmGeo { int NS, WN, WM, gx, gy, rows; size_t lds; };

static bool conv1x1_geo(GemmGeo* g, int M, int K, int N) {
  if (K != 64 && K != 128 && K != 256 && K != 512) return false;
  if (N % 64 || M % 32 || M <= 0) return false;
  const int rowb = K * 2 + 16;
  int ns = std::min(N, (80 * 1024 / rowb) / 64 * 64);
  ns = std::min(ns, 256);                            // at most 4 channel pairs per workgroup (8 waves: >= 2 pixel waves)
  while (ns > 64 && (N % ns || (ns / 64 != 1 && ns / 64 != 2 && ns / 64 != 4))) ns -= 64;
  if (N % ns) return false;
  g->NS = ns;
  g->WN = ns / 64;
  g->WM = kGemmWaves / g->WN;
  g->gy = N / ns;
  g->lds = (size_t)ns * rowb;
  const int nblk = M / 32;
  // persistent workgroups: about two per CU over the whole grid; rows must divide M for the statistics kernel
  int gx = std::max(1, std::min((nblk + g->WM - 1) / g->WM, 512 / g->gy > 0 ? 512 / g->gy : 1));
  while (gx > 1 && (M % (gx * g->WM)) != 0) --gx;
  g->gx = gx;
  g->rows = gx * g->WM;
  return true;
}

int conv1x1_rows(int M, int K, int N) {
  GemmGeo g;
  return conv1x1_geo(&g, M, K, N) ? g.rows : MRLA_EUNSUPPORTED;
}

int launch_conv1x1_fwd(const void* x, const void* w, void* y, float* part, int M, int K, int N, hipStream_t st) {
  GemmGeo g;
  if (!conv1x1_geo(&g, M, K, N)) return MRLA_EUNSUPPORTED;
  const dim3 grid(g.gx, g.gy), block(kGemmWaves * kWave);
#define CALL_M(KS, MO)                                                                                           \
  {                                                                                                              \
    if (g.lds > 48 * 1024 &&                                                                                     \
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_fwd_kernel<KS, MO>),                           \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds) != hipSuccess)               \
      return MRLA_EHIP;                                                                                          \
    hipLaunchKernelGGL((conv1x1_fwd_kernel<KS, MO>), grid, block, g.lds, st, (const bf16_t*)x, (const bf16_t*)w, \
                       (bf16_t*)y, part, M, N, g.NS, g.WN);                                                      \
  }
#define CALL(KS) { if (part) CALL_M(KS, true) else CALL_M(KS, false) }
  switch (K) {
    case 64:  CALL(4) break;
    case 128: CALL(8) break;
    case 256: CALL(16) break;
    case 512: CALL(32) break;
    default: return MRLA_EUNSUPPORTED;
  }
#undef CALL
#undef CALL_M
  return hip_status(hipGetLastError());
}

}  // namespace mrla
