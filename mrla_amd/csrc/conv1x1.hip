// 1x1 stride-1 convolution of a channels_last bf16 activation as an MFMA GEMM with a BatchNorm-statistics epilogue:
//   Y[M, N] = X[M, K] * W[N, K]^T      (M = b*h*w pixels, K = in-channels, N = out-channels; both operands K-contiguous)
//   part[row, n, 0..3] = per-workgroup moment record of the bf16-ROUNDED outputs of channel n (MRLA_GEMM_MOMENTS):
//                        sum (y - p), sum (y - p)^2, the pivot p (a first output of the channel), pixel count
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
//     v_permlane32_swap between the two half-waves leaves every lane with 16-byte pieces of its pixel's output row;
//     a wave-private LDS tile [32 pixels][64 channels] then turns them into stores of whole 128-byte lines (8 lanes
//     per pixel) -- 32-byte fragments per pixel straight from the lanes cost the L2 four requests per line;
//   * the BatchNorm moments are per-lane running sums over the wave's pixels (register = channel), reduced across
//     lanes and pixel-waves once at the end of the kernel and written as one partial row per workgroup.
#include <algorithm>

#include "mrla_device.h"
#include "mrla_kernels.h"

namespace mrla {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kOutRowB = 64 * 2 + 16;               // padded row of the per-wave output tile (64 channels of one pixel)
constexpr int kOutTileB = 32 * kOutRowB;

// channel of accumulator register `reg` inside its 32-channel tile, for lane half h
__device__ __forceinline__ int acc_channel(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

template <int KS, bool MOM, int NW>
__global__ __launch_bounds__(NW * kWave) void conv1x1_fwd_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, bf16_t* __restrict__ Y, float* __restrict__ part,
    int M, int N, int NS, int WN, int rows_total) {
  constexpr int K = KS * 16;
  constexpr int ROWB = K * 2 + 16;                   // padded LDS row of W, bytes
  constexpr int KC = KS < 16 ? KS : 16;              // k-steps whose X fragments are in registers at a time
  constexpr int NCH = KS / KC;
  constexpr bool DB = KS <= 16;                      // the next pixel block's X fragments are fetched during this one's MFMAs
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int r = lane & 31, h = lane >> 5;
  const int WM = NW / WN, wn = wave % WN, wm = wave / WN;
  const int n_slice0 = blockIdx.y * NS;
  unsigned char* otile = smem_raw + (size_t)NS * ROWB + (size_t)wave * kOutTileB;      // after the W slice

  // stage the W slice [NS][K] into LDS (16-byte pieces, coalesced)
  {
    constexpr int PPR = K / 8;                       // 16-byte pieces per row
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(W + (size_t)n_slice0 * K);
    for (int i = threadIdx.x; i < NS * PPR; i += NW * kWave) {
      const int row = i / PPR, pc = i - row * PPR;
      *reinterpret_cast<u32x4*>(smem_raw + (size_t)row * ROWB + pc * 16) = wsrc[i];
    }
  }
  __syncthreads();

  // moments about a pivot (this wave's first output of the channel), see conv1x1_wide.hip
  float s1[2][16], s2[2][16], pv[2][16];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) { s1[t][i] = 0.f; s2[t][i] = 0.f; pv[t][i] = 0.f; }
  bool have_pivot = false;
  int npix = 0;                                     // pixels this wave accumulated (wave-uniform)

  const int nblk = (M + 31) / 32;
  const int stride = gridDim.x * WM;
  int blk = blockIdx.x * WM + wm;
  u32x4 xf[KC], xn[DB ? KC : 1];
  // K chunk `kc` (KC k-steps) of the X fragments of pixel block b_
  auto load_x = [&](u32x4 (&dst)[KC], int b_, int kc) {
    const int row = min(b_ * 32 + r, M - 1);        // (rows past the end are loaded clamped and never used)
    const u32x4* xp = reinterpret_cast<const u32x4*>(X + (size_t)row * K + kc * KC * 16 + h * 8);
#pragma unroll
    for (int ks = 0; ks < KC; ++ks) dst[ks] = xp[ks * 2];
  };
  if (blk < nblk) load_x(xf, blk, 0);
  for (; blk < nblk; blk += stride) {
    if constexpr (DB) {
      if (blk + stride < nblk) load_x(reinterpret_cast<u32x4(&)[KC]>(xn), blk + stride, 0);
    }
    const bool live = blk * 32 + r < M;                      // this lane's pixel exists (ragged last block)
    npix += min(32, M - blk * 32);
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
      // round to bf16 (pairs of neighbouring channels), statistics of the rounded values
      unsigned p[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 pr;
        pr[0] = from_f<bf16_t>(acc[t][2 * i]);
        pr[1] = from_f<bf16_t>(acc[t][2 * i + 1]);
        p[i] = __builtin_bit_cast(unsigned, pr);
      }
      if (MOM) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float lo = __uint_as_float(p[i] << 16), hi = __uint_as_float(p[i] & 0xffff0000u);
          if (!have_pivot) {       // (wave-uniform) first block of this wave: pixel 0 of the half-wave always exists
            pv[t][2 * i] = __shfl(lo, h * 32, kWave);
            pv[t][2 * i + 1] = __shfl(hi, h * 32, kWave);
          }
          const float dl = live ? lo - pv[t][2 * i] : 0.f, dh = live ? hi - pv[t][2 * i + 1] : 0.f;
          s1[t][2 * i] += dl;     s2[t][2 * i] = fmaf(dl, dl, s2[t][2 * i]);
          s1[t][2 * i + 1] += dh; s2[t][2 * i + 1] = fmaf(dh, dh, s2[t][2 * i + 1]);
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
      // this lane's pieces of its pixel's row in the wave's output tile: channels t*32 + h*8 .. and t*32 + 16 + h*8 ..
      unsigned char* orow = otile + r * kOutRowB + (t * 32 + h * 8) * 2;
      *reinterpret_cast<u32x4*>(orow) = (u32x4){p[0], p[1], p[2], p[3]};
      *reinterpret_cast<u32x4*>(orow + 32) = (u32x4){p[4], p[5], p[6], p[7]};
    }
    have_pivot = true;
    // whole lines out: 8 lanes per pixel (64 channels = 128 bytes), 8 pixels per store instruction
    {
      const int px = lane >> 3, piece = lane & 7;
      bf16_t* ybase = Y + (size_t)blk * 32 * N + n_slice0 + wn * 64 + piece * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pr = i * 8 + px;
        const u32x4 v = *reinterpret_cast<const u32x4*>(otile + pr * kOutRowB + piece * 16);
        if (blk * 32 + pr < M) *reinterpret_cast<u32x4*>(ybase + (size_t)pr * N) = v;
      }
    }
    if constexpr (DB) {
#pragma unroll
      for (int ks = 0; ks < KC; ++ks) xf[ks] = xn[ks];
    } else {
      if (blk + stride < nblk) load_x(xf, blk + stride, 0);
    }
  }

  if (MOM) {
    // rows beyond the active workgroups only exist to make the row count divide M: empty records
    if (blockIdx.x == 0) {
      for (int i = threadIdx.x; i < (rows_total - (int)gridDim.x) * NS * 4; i += NW * kWave) {
        const int row = gridDim.x + i / (NS * 4), j = i % (NS * 4);
        part[((size_t)row * N + n_slice0) * 4 + j] = 0.f;
      }
    }
    // sum over the 32 pixel-lanes of each half (they share the pivot), then merge the workgroup's pixel-waves by
    // re-basing them onto the first one's pivot (fixed order, through the LDS of the output tiles, which are done
    // with): one record per workgroup and channel
    __syncthreads();
    float* sums = reinterpret_cast<float*>(smem_raw + (size_t)NS * ROWB);          // [WM][NS][4]
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
          const int ch = wn * 64 + t * 32 + acc_channel(i, h);
          float* rec = sums + ((size_t)wm * NS + ch) * 4;
          rec[0] = a; rec[1] = b; rec[2] = pv[t][i]; rec[3] = (float)npix;
        }
      }
    __syncthreads();
    float* dst = part + ((size_t)blockIdx.x * N + n_slice0) * 4;
    for (int ch = threadIdx.x; ch < NS; ch += NW * kWave) {
      float S1 = 0.f, S2 = 0.f, P = 0.f, n = 0.f;
      for (int v = 0; v < WM; ++v) {
        const float* rec = sums + ((size_t)v * NS + ch) * 4;
        const float a = rec[0], b = rec[1], nv = rec[3];
        if (nv == 0.f) continue;                     // a pixel-wave without a block
        if (n == 0.f) P = rec[2];
        const float dlt = rec[2] - P;                // sums about rec[2] -> about P
        S2 += b + 2.f * dlt * a + nv * dlt * dlt;
        S1 += a + nv * dlt;
        n += nv;
      }
      dst[ch * 4 + 0] = S1; dst[ch * 4 + 1] = S2; dst[ch * 4 + 2] = P; dst[ch * 4 + 3] = n;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// geometry: N slice per workgroup (LDS), waves along N, persistent grid
// ------------------------------------------------------------------------------------------------
struct GemmGeo { int NS, WN, WM, NW, gx, gy, rows; size_t lds; };

static bool conv1x1_geo(GemmGeo* g, int M, int K, int N) {
  // K = 512 (two fragment chunks per pixel block, no prefetch across them) is slower than the stock convolution: not taken
  if (K != 64 && K != 128 && K != 256) return false;
  if (N % 64 || M <= 0) return false;
  const int rowb = K * 2 + 16;
  int ns = std::min(N, 256);                         // at most 4 channel pairs (= waves along N) per workgroup
  while (ns > 64 && (N % ns || (ns / 64 != 1 && ns / 64 != 2 && ns / 64 != 4) || (size_t)ns * rowb > 72 * 1024)) ns -= 64;
  if (N % ns) return false;
  g->NS = ns;
  g->WN = ns / 64;
  // 4-wave workgroups while the X fragments are small (more workgroups per CU under the register budget), else 8
  g->NW = K <= 128 ? 4 : 8;
  if (g->NW < g->WN) g->NW = g->WN;
  g->WM = g->NW / g->WN;
  g->gy = N / ns;
  g->lds = (size_t)ns * rowb + (size_t)g->NW * kOutTileB;
  const int nblk = (M + 31) / 32;
  // persistent workgroups: a few per CU over the whole grid; rows must divide M for the statistics kernel
  const int want = std::max(1, (256 * (g->NW == 4 ? 3 : 2)) / g->gy);
  int gx = std::max(1, std::min((nblk + g->WM - 1) / g->WM, want));
  // the statistics kernel takes (rows, M / rows): prefer a grid whose workgroup count divides M, else pad
  // the partial buffer with zero rows up to the next divisor of M
  for (int t = gx; t >= std::max(1, gx - gx / 4); --t)
    if (M % t == 0) { gx = t; break; }
  int rows = gx;
  while (rows <= 2 * gx + 64 && M % rows) ++rows;
  if (M % rows) return false;
  g->gx = gx;
  g->rows = rows;
  return true;
}

int conv1x1_rows(int M, int K, int N) {
  const int wide = conv1x1_wide_rows(M, K, N);                 // N % 256 == 0: the streaming kernel of conv1x1_wide.hip
  if (wide > 0) return wide;
  GemmGeo g;
  if (conv1x1_geo(&g, M, K, N)) return g.rows;
  return conv1x1_kstream_supported(M, K, N) ? conv1x1_kstream_rows(M, K, N) : MRLA_EUNSUPPORTED;   // K >= 512: one record row per pixel tile
}

// {32-pixel blocks per workgroup (wide form) / per pixel-wave (narrow form), pipeline depth in blocks, workgroups, rows}
int conv1x1_plan(int M, int K, int N, int add, int* out) {
  if (conv1x1_wide_plan(M, K, N, add, out) == MRLA_OK) return MRLA_OK;
  GemmGeo g;
  if (!add && !conv1x1_geo(&g, M, K, N) && conv1x1_kstream_supported(M, K, N)) {
    out[0] = K / 32; out[1] = conv1x1_kstream_stages(M, K, N); out[2] = 0; out[3] = conv1x1_kstream_rows(M, K, N);   // 32-deep chunks per tile, LDS stages, -, record rows
    return MRLA_OK;
  }
  if (add || !conv1x1_geo(&g, M, K, N)) return MRLA_EUNSUPPORTED;
  const int nblk = (M + 31) / 32;
  out[0] = (nblk + g.gx * g.WM - 1) / (g.gx * g.WM);
  out[1] = 2;                               // the next block's X fragments are fetched during this block's MFMAs
  out[2] = g.gx * g.gy;
  out[3] = g.rows;
  return MRLA_OK;
}

int launch_conv1x1_fwd(const void* x, const void* w, void* y, float* part, int M, int K, int N, hipStream_t st) {
  if (conv1x1_wide_rows(M, K, N) > 0) return launch_conv1x1_wide(x, w, nullptr, y, part, M, K, N, st);
  GemmGeo g;
  if (!conv1x1_geo(&g, M, K, N)) {
    if (!conv1x1_kstream_supported(M, K, N)) return MRLA_EUNSUPPORTED;
    return launch_conv1x1_kstream(x, w, y, part, M, K, N, st);
  }
  const dim3 grid(g.gx, g.gy), block(g.NW * kWave);
#define CALL_W(KS, MO, NWV)                                                                                          \
  {                                                                                                                  \
    if (lds_opt_in(reinterpret_cast<const void*>(conv1x1_fwd_kernel<KS, MO, NWV>), g.lds) != hipSuccess)              \
      return MRLA_EHIP;                                                                                              \
    hipLaunchKernelGGL((conv1x1_fwd_kernel<KS, MO, NWV>), grid, block, g.lds, st, (const bf16_t*)x, (const bf16_t*)w, \
                       (bf16_t*)y, part, M, N, g.NS, g.WN, g.rows);                                                  \
  }
#define CALL_M(KS, MO) { if (g.NW == 4) CALL_W(KS, MO, 4) else CALL_W(KS, MO, 8) }
#define CALL(KS) { if (part) CALL_M(KS, true) else CALL_M(KS, false) }
  switch (K) {
    case 64:  CALL(4) break;
    case 128: CALL(8) break;
    case 256: CALL(16) break;
    default: return MRLA_EUNSUPPORTED;
  }
#undef CALL
#undef CALL_M
#undef CALL_W
  return hip_status(hipGetLastError());
}

}  // namespace mrla
