// Weight gradient of a 1x1 stride-1 convolution of channels_last bf16 activations as a split-M MFMA GEMM:
//   dW[n, k] = sum_m dY[m, n] * X[m, k]        (M = b*h*w pixels; dY [M, N] and X [M, K] both channel-contiguous)
// Reference: the backward of the bottleneck's conv1 / conv3 (resnet/models/resnet_mrla_light.py:93,100, nn.Conv2d 1x1).
//
// The product sums over the SLOW index of both operands, N x K is small (4 K .. 1 M outputs) and M is huge (12 K .. 800 K):
// the kernel is a stream over pixels -- every byte of dY and X is read once when one workgroup tile covers N x K
// (stages 1-2), and the arithmetic is 10-30 % of the MFMA rate.  So:
//   * a workgroup (4 waves) owns an output tile TN x TK (<= 32 K accumulators, 64 x 64 .. 128 x 256) and a contiguous
//     range of 32-pixel chunks; tiles x splits = one workgroup per CU; every workgroup writes its fp32 partial tile to
//     part[split][N][K] and a second kernel sums the splits in a fixed order (no atomics, no memset);
//   * workgroup ids are spread over the 8 XCDs round-robin, so the ids are re-mapped to make the tiles of ONE pixel
//     range neighbours in time on ONE XCD: the operand chunk they share comes out of that XCD's L2;
//   * the chunk [32 px][TN] of dY and [32 px][TK] of X goes from global memory straight into LDS with LDS-DMA buffer
//     loads (16 B per lane, 1 KB per wave-instruction, no VGPR staging); four stages rotate, one barrier per chunk;
//     pixels beyond M arrive as zeros from the buffer bounds check;
//   * both MFMA operands are COLUMNS of those row-major tiles (lane = channel, 8 consecutive pixels per lane):
//     gfx950's transposing LDS read `ds_read_b64_tr_b16` delivers exactly that (4 pixels x 16 channels per 16 lanes),
//     two reads per 32x32x16 operand.  The DMA places 16-byte chunk c of tile row r at chunk position c ^ f(r)
//     (f = (r & 3) << 2 for rows of >= 256 B, ((r >> 1) & 1) << 2 for 128-B rows), which spreads the four rows a
//     half-wave reads over the four 64-B bank groups: conflict-free;
//   * LDS reads and the barrier are inline asm / bare `s_barrier`: the compiler orders every LDS read it can see (and
//     __syncthreads()) behind ALL outstanding LDS-DMA loads with `s_waitcnt vmcnt(0)`, which would serialise the
//     stages (measured: 1 us per chunk whatever the depth).
#include <algorithm>

#include "mrla_device.h"
#include "mrla_kernels.h"

namespace mrla {
namespace {

typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x4 __attribute__((ext_vector_type(4)));
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
typedef float wg_f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) wg_s16x4* lds_s16x4_ptr;

constexpr int kPC = 32;                 // pixels per stage
constexpr int kWgStages = 4;            // LDS stages: one being read, one being refilled, two in flight
constexpr int kWgWaves = 4;
#define MRLA_WG_FLAGS 0x00020000        /* raw buffer descriptor word 3 (as nhwc_rows.h) */
constexpr int kWgTarget = 256;          // workgroups: one per CU (the stages, not a second workgroup, hide the DMA latency;
                                        // every workgroup costs a partial tile of up to 128 KB written and re-read)

template <int TN, int TK, int PC>
struct WgGeo {
  static constexpr int WN = (TN >= 256 && TK <= 64) ? 4 : (TK >= 256 && TN <= 64) ? 1 : 2;   // waves along n
  static constexpr int WK = kWgWaves / WN;
  static constexpr int WTN = TN / WN, WTK = TK / WK;       // wave tile
  static constexpr int BN = WTN / 32, BK = WTK / 32;       // 32x32 accumulator blocks
  static constexpr int NIY = TN / 64 * (PC / 32), NIX = TK / 64 * (PC / 32);       // DMA wave-instructions per wave and stage
  static constexpr int YB = PC * TN * 2, XB = PC * TK * 2;
  static constexpr int SB = YB + XB;                       // bytes of one stage
  static_assert(BN >= 1 && BK >= 1 && BN * BK <= 8, "wave tile");
};

// chunk swizzle of tile row `row` for rows of CPR 16-byte chunks
template <int CPR>
__device__ __forceinline__ int swz(int row) { return CPR >= 16 ? ((row & 3) << 2) : (((row >> 1) & 1) << 2); }

// LDS byte offset (inside a [kPC][CPR*8] bf16 tile) this lane hands to the transposing read of the 4-pixel x 16-channel
// block it takes part in: pixels row0 .. row0+3 (row0 % 4 == 0), channels ch0 + 16*(group & 1) .. +15 -- so that the
// wave's two 32-lane halves get pixels row0.. and row0+8.. of channels ch0 .. ch0+31 (the 32x32x16 operand map).
template <int CPR>
__device__ __forceinline__ int tr_offset(int lane, int row0, int ch0) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int row = row0 + 8 * (g >> 1) + q;
  const int col = ch0 + 16 * (g & 1) + 4 * p;
  return row * (CPR * 16) + ((((col >> 3) ^ swz<CPR>(row))) << 4) + ((col >> 2) & 1) * 8;
}

// 32x32x16 operand of channels ch0 .. ch0+31 over 16 pixels of a tile: two transposing reads (pixels +0..3 and +4..7 of
// this half-wave's eight) from LDS byte address `addr` = tile + tr_offset(lane, row0, ch0).
// The reads are inline asm on purpose: the compiler treats an LDS-DMA load as a write to all of LDS and puts
// `s_waitcnt vmcnt(0)` in front of every LDS read it can see -- which would drain the chunks still in flight and turn
// the multi-stage pipeline into one DMA round trip per chunk.  Ordering is explicit instead: frag_issue only starts
// the reads, frag_fence<N> waits until at most N newer LDS reads are outstanding and ties the registers to that wait.
struct Frag {
  wg_s16x4 lo, hi;
};
template <int CPR>
__device__ __forceinline__ void frag_issue(Frag& f, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
               : "=&v"(f.lo), "=&v"(f.hi)
               : "v"(addr), "n"(4 * CPR * 16));
}
template <int N>
__device__ __forceinline__ void frag_fence(Frag& f, bool wait) {
  if (wait) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.lo), "+v"(f.hi) : "n"(N) : "memory");
  else asm volatile("" : "+v"(f.lo), "+v"(f.hi)::"memory");      // rides on the wait of the fragment fenced just before
}
__device__ __forceinline__ wg_bf16x8 frag_value(const Frag& f) {
  return __builtin_bit_cast(wg_bf16x8, __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ unsigned wg_lds_addr(const void* p) {
  return (unsigned)(size_t)((__attribute__((address_space(3))) const char*)p);
}

// grid: tiles * splits workgroups (tiles = (N/TN)*(K/TK)) padded to a multiple of 8; 256 threads; LDS = ST * SB
template <int TN, int TK, int ST, int PC>
__global__ __launch_bounds__(kWgWaves* kWave) void conv1x1_wgrad_kernel(const bf16_t* __restrict__ dY,
                                                                        const bf16_t* __restrict__ X,
                                                                        float* __restrict__ part, int M, int N, int K,
                                                                        int chunks_per_wg, int nsplits) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef WgGeo<TN, TK, PC> G;
  constexpr int CPRY = TN / 8, CPRX = TK / 8;            // 16-byte chunks per tile row
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  // Workgroups go to the 8 XCDs round-robin (id % 8), each XCD with its own L2.  The tiles of one pixel range re-read
  // the same dY / X chunks, so they are made neighbours in time ON ONE XCD: XCD x takes the virtual ids
  // x*per .. (x+1)*per-1 in dispatch order, virtual id = split * tiles + tile.
  const int tiles_k = K / TK, tiles = (N / TN) * tiles_k;
  const int per = gridDim.x >> 3;
  const int vid = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (vid >= tiles * nsplits) return;                     // (the grid is padded to a multiple of 8)
  const int split = vid / tiles, tile = vid - split * tiles;
  const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
  const int n0 = tn * TN, k0 = tk * TK;
  const int chunks_total = (M + PC - 1) / PC;
  const int c_begin = split * chunks_per_wg;
  const int nch = min(chunks_per_wg, chunks_total - c_begin);      // >= 1 by construction of the grid

  // ---- DMA plan: wave-instruction u of a stage moves 1 KB = 64/CPR tile rows; this wave issues u = wave + 4*i ----
  const auto rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dY), 0, (int)((size_t)M * N * 2), MRLA_WG_FLAGS);
  const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(X), 0, (int)((size_t)M * K * 2), MRLA_WG_FLAGS);
  unsigned voffY[G::NIY], voffX[G::NIX];
#pragma unroll
  for (int i = 0; i < G::NIY; ++i) {
    const int u = wave + kWgWaves * i, row = u * (64 / CPRY) + lane / CPRY, cp = lane % CPRY;
    voffY[i] = (unsigned)(((size_t)c_begin * PC + row) * N * 2 + (n0 + ((cp ^ swz<CPRY>(row)) << 3)) * 2);
  }
#pragma unroll
  for (int i = 0; i < G::NIX; ++i) {
    const int u = wave + kWgWaves * i, row = u * (64 / CPRX) + lane / CPRX, cp = lane % CPRX;
    voffX[i] = (unsigned)(((size_t)c_begin * PC + row) * K * 2 + (k0 + ((cp ^ swz<CPRX>(row)) << 3)) * 2);
  }
  const unsigned advY = (unsigned)PC * N * 2, advX = (unsigned)PC * K * 2;
  int issued = 0;
  auto issue = [&](int stage) {
    unsigned char* sb = smem_raw + stage * G::SB;
    const unsigned kill = issued++ < nch ? 0u : 0x80000000u;      // past the range: out of bounds, no memory traffic
#pragma unroll
    for (int i = 0; i < G::NIY; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_void_ptr)(sb + (wave + kWgWaves * i) * 1024), 16, voffY[i] | kill, 0, 0, 0);
      voffY[i] += advY;
    }
#pragma unroll
    for (int i = 0; i < G::NIX; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_ptr)(sb + G::YB + (wave + kWgWaves * i) * 1024), 16, voffX[i] | kill, 0, 0, 0);
      voffX[i] += advX;
    }
  };

  // ---- this wave's operand addresses (stage- and k-step-relative) ----
  const int wn = wave / G::WK, wk = wave - wn * G::WK;
  const unsigned lds0 = wg_lds_addr(smem_raw);
  unsigned offA[G::BN], offB[G::BK];
#pragma unroll
  for (int i = 0; i < G::BN; ++i) offA[i] = lds0 + tr_offset<CPRY>(lane, 0, wn * G::WTN + i * 32);
#pragma unroll
  for (int j = 0; j < G::BK; ++j) offB[j] = lds0 + G::YB + tr_offset<CPRX>(lane, 0, wk * G::WTK + j * 32);

  wg_f32x16 acc[G::BN][G::BK];
#pragma unroll
  for (int i = 0; i < G::BN; ++i)
#pragma unroll
    for (int j = 0; j < G::BK; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ST stages rotate.  Chunks past the range are issued all the same, with an out-of-bounds offset (zeros from the
  // bounds check, no memory traffic): the wait counts stay compile-time constants.
  // The hand-over to the next chunk sits in front of the LAST k-step of a chunk, not between chunks: the wave waits
  // for its part of chunk c+1, meets the others at the barrier (after which chunk c-1's buffer is free for the DMA of
  // chunk c+ST-1), starts the LDS reads of chunk c+1's first k-step, and only then issues the MFMAs of chunk c's
  // last k-step -- barrier skew and LDS latency are covered by MFMAs already queued.
  constexpr int KSN = PC / 16, NF = 2 * (G::BN + G::BK), NI = G::NIY + G::NIX;
  static_assert(KSN % 2 == 0 && ST >= 3, "fragment double buffer / stages");
  Frag fa[2][G::BN], fb[2][G::BK];
  auto read_step = [&](int buf, unsigned sbo, int ks) {
#pragma unroll
    for (int i = 0; i < G::BN; ++i) frag_issue<CPRY>(fa[buf][i], offA[i] + sbo + ks * 16 * CPRY * 16);
#pragma unroll
    for (int j = 0; j < G::BK; ++j) frag_issue<CPRX>(fb[buf][j], offB[j] + sbo + ks * 16 * CPRX * 16);
  };
  auto mfma_step = [&](int buf) {
#pragma unroll
    for (int i = 0; i < G::BN; ++i)
#pragma unroll
      for (int j = 0; j < G::BK; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_value(fa[buf][i]), frag_value(fb[buf][j]), acc[i][j], 0, 0, 0);
  };
#pragma unroll
  for (int j = 0; j < ST - 1; ++j) issue(j);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * NI) : "memory");
  __builtin_amdgcn_s_barrier();                         // (bare: __syncthreads() would drain vmcnt as well)
  asm volatile("" ::: "memory");
  read_step(0, 0, 0);
  int stage = 0, nxt = ST - 1;
  for (int c = 0; c < nch; ++c) {
    const unsigned sbo = stage * G::SB;
    stage = stage + 1 == ST ? 0 : stage + 1;
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) {
      const int cur = ks & 1;
      int pending = NF;                                 // LDS reads issued after this step's own
      if (ks + 1 < KSN) {
        read_step(cur ^ 1, sbo, ks + 1);
      } else if (c + 1 < nch) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 3) * NI) : "memory");      // my part of chunk c+1 has landed
        __builtin_amdgcn_s_barrier();                   // everybody's has; everybody is done reading chunk c-1
        asm volatile("" ::: "memory");
        issue(nxt);
        nxt = nxt + 1 == ST ? 0 : nxt + 1;
        read_step(0, stage * G::SB, 0);
      } else {
        pending = 0;
      }
#pragma unroll
      for (int i = 0; i < G::BN; ++i) {
        if (pending) frag_fence<NF>(fa[cur][i], i == 0); else frag_fence<0>(fa[cur][i], i == 0);
      }
#pragma unroll
      for (int j = 0; j < G::BK; ++j) frag_fence<0>(fb[cur][j], false);
      mfma_step(cur);
    }
  }

  // ---- partial tile: D[i][j], j = lane % 32 on the lane, i = 8*(e/4) + 4*(lane/32) + e%4 in register e ----
  float* po = part + ((size_t)split * N + n0 + wn * G::WTN) * K + k0 + wk * G::WTK + (lane & 31);
  const int h = lane >> 5;
#pragma unroll
  for (int i = 0; i < G::BN; ++i)
#pragma unroll
    for (int j = 0; j < G::BK; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        po[(size_t)(i * 32 + 8 * (e >> 2) + 4 * h + (e & 3)) * K + j * 32] = acc[i][j][e];
#endif
}

// dW[e] = sum over splits of part[s][e] in a fixed order: a workgroup takes 64 outputs, its 16 groups of 16 lanes take
// every 16th split with 16-byte loads, LDS folds the groups
template <typename TO>
__global__ __launch_bounds__(256) void conv1x1_wgrad_reduce_kernel(const float* __restrict__ part, TO* __restrict__ dW,
                                                                   int splits, int NK) {
  __shared__ float4 red[16][16];
  const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int e = blockIdx.x * 64 + q * 4;                 // NK % 64 == 0
  const float4* src = reinterpret_cast<const float4*>(part + e);
  const size_t pitch = (size_t)NK / 4;
  float4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
  int s = g;
  for (; s + 16 < splits; s += 32) {
    const float4 u = src[(size_t)s * pitch], v = src[(size_t)(s + 16) * pitch];
    a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
  }
  if (s < splits) {
    const float4 u = src[(size_t)s * pitch];
    a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
  }
  red[g][q] = float4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
  __syncthreads();
  if (threadIdx.x < 64) {
    const int qq = threadIdx.x >> 2, cc = threadIdx.x & 3;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += reinterpret_cast<const float*>(&red[i][qq])[cc];
    dW[blockIdx.x * 64 + threadIdx.x] = from_f<TO>(t);
  }
}

struct WgPlan {
  int tn = 0, tk = 0, tiles = 0, splits = 0, chunks_per_wg = 0;
};

// tile = the largest of {256, 128, 64} dividing each extent, shrunk (the larger side first) to <= 32 K accumulators;
// splits so that the grid is one workgroup per CU
WgPlan wgrad_plan(int M, int K, int N) {
  WgPlan p;
  if (M <= 0 || N % 64 || K % 64 || (size_t)M * std::max(N, K) * 2 >= (size_t)1 << 31) return p;
  int tn = N % 256 == 0 ? 256 : N % 128 == 0 ? 128 : 64;
  int tk = K % 256 == 0 ? 256 : K % 128 == 0 ? 128 : 64;
  // (16 K / 8 K accumulators per workgroup on the deep shapes -- fewer fp32 partial tiles -- were measured: no gain, the tile loop
  // bounds them, profiles/r03_notes.md section 9)
  while (tn * tk > 32768) {
    if (tn >= tk) tn /= 2; else tk /= 2;
  }
  p.tn = tn; p.tk = tk;
  p.tiles = (N / tn) * (K / tk);
  const int chunks = (M + kPC - 1) / kPC;
  const int want = std::max(1, std::min(chunks, (kWgTarget + p.tiles - 1) / p.tiles));
  p.chunks_per_wg = (chunks + want - 1) / want;
  p.splits = (chunks + p.chunks_per_wg - 1) / p.chunks_per_wg;
  return p;
}

template <int TN, int TK, int ST, int PC>
int launch_tile(const WgPlan& p, const void* dy, const void* x, float* part, int M, int K, int N, hipStream_t st) {
  typedef WgGeo<TN, TK, PC> G;
  const size_t lds = (size_t)ST * G::SB;
  if (lds_opt_in(reinterpret_cast<const void*>(conv1x1_wgrad_kernel<TN, TK, ST, PC>), lds) != hipSuccess) return MRLA_EHIP;
  hipLaunchKernelGGL((conv1x1_wgrad_kernel<TN, TK, ST, PC>), dim3((p.tiles * p.splits + 7) / 8 * 8), dim3(kWgWaves * kWave), lds, st,
                     (const bf16_t*)dy, (const bf16_t*)x, part, M, N, K, p.chunks_per_wg, p.splits);
  return MRLA_OK;
}

}  // namespace

int conv1x1_wgrad_rows(int M, int K, int N) {
  const WgPlan p = wgrad_plan(M, K, N);
  return p.tiles ? p.splits : MRLA_EUNSUPPORTED;
}

// {32-pixel chunks per workgroup, LDS stages, tile n, tile k, splits, output tiles}
int conv1x1_wgrad_plan(int M, int K, int N, int* out) {
  const WgPlan p = wgrad_plan(M, K, N);
  if (!p.tiles) return MRLA_EUNSUPPORTED;
  out[0] = p.chunks_per_wg; out[1] = kWgStages; out[2] = p.tn; out[3] = p.tk; out[4] = p.splits; out[5] = p.tiles;
  return MRLA_OK;
}

int launch_conv1x1_wgrad(const void* dy, const void* x, float* part, void* dw, int dw_f32, int M, int K, int N,
                         hipStream_t st) {
  const WgPlan p = wgrad_plan(M, K, N);
  if (!p.tiles) return MRLA_EUNSUPPORTED;
  int rc = MRLA_EUNSUPPORTED;
#define MRLA_WG_TILE(A, B) \
  if (p.tn == A && p.tk == B) rc = launch_tile<A, B, kWgStages, kPC>(p, dy, x, part, M, K, N, st);
  MRLA_WG_TILE(64, 64) MRLA_WG_TILE(64, 128) MRLA_WG_TILE(64, 256) MRLA_WG_TILE(128, 64) MRLA_WG_TILE(128, 128)
  MRLA_WG_TILE(128, 256) MRLA_WG_TILE(256, 64) MRLA_WG_TILE(256, 128)
#undef MRLA_WG_TILE
  if (rc != MRLA_OK) return rc;
  const int NK = N * K;
  if (dw_f32)      // the fp32 master weight's gradient directly (no cast kernel behind an autocast convolution)
    hipLaunchKernelGGL(conv1x1_wgrad_reduce_kernel<float>, dim3(NK / 64), dim3(256), 0, st, part, (float*)dw, p.splits, NK);
  else
    hipLaunchKernelGGL(conv1x1_wgrad_reduce_kernel<bf16_t>, dim3(NK / 64), dim3(256), 0, st, part, (bf16_t*)dw, p.splits, NK);
  return hip_status(hipGetLastError());
}

}  // namespace mrla
