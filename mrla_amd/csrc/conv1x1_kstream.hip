// 1x1 stride-1 convolution with a WIDE REDUCTION (K = 512 .. 2048 input channels) as a K-streaming MFMA GEMM:
//   Y[M, N] = X[M, K] * W[N, K]^T        bf16 in / out, fp32 accumulate;  M = b*h*w pixels, K % 32 == 0, N % 128 == 0
// Reference: conv1 of the bottlenecks of stages 2-4 (resnet/models/resnet_mrla_light.py:93: 512 / 1024 / 2048 -> width),
// conv3 of stage 4 (:100: 512 -> 2048) and, with the operands swapped by the caller (x = dY, w = W^T), the input
// gradients of conv3 everywhere but stage 1 and of conv1 in stage 4.
//
// conv1x1.hip / conv1x1_wide.hip keep the whole [N or 256][K] weight slice of a workgroup in LDS / registers, which
// stops at K = 256.  Here neither operand is resident: a workgroup (8 waves) owns an output tile of TM pixels x TN
// channels and walks K in chunks of 32:
//   * chunk = X[TM][32] and W[TN][32] (64-byte rows), both global -> LDS by LDS-DMA (16 B per lane, 16 rows per
//     wave-instruction), three stages; the weight chunk comes out of L2 (every workgroup reads the same W), X is read
//     once per TN output channels -- the channel groups of one pixel tile are neighbours in time on one XCD;
//   * a wave owns 64 channels x (32 or 64) pixels of the tile: MFMA 32x32x16 with A = W rows, B = X rows, both operand
//     fragments one ds_read_b128 per k-step (16-byte slot s of row r sits at slot s ^ ((r >> 2) & 3): conflict-free for
//     the 16-lane groups of ds_read_b128);
//   * 72 KB of LDS and <= 128 VGPRs: two workgroups per CU, so that one's DMA round trips and barriers are covered by
//     the other's MFMAs (a K = 512 tile is only 16 chunks long: a lone workgroup would spend a third of its life
//     filling and draining its pipeline);
//   * LDS reads are inline asm with explicit lgkmcnt fences, the barrier is a bare s_barrier and the DMA waits are
//     constant-count vmcnt (see conv1x1_wgrad.hip: the compiler would drain all DMA stages at every LDS access);
//   * the fp32 tile is rounded once, transposed to pixel-major 16-byte pieces with v_permlane32_swap (as
//     conv1x1_wide.hip) through the (then idle) stage memory and leaves as whole rows.
// These products need 0.5 - 2 PFLOP/s to run at the HBM rate (N*K/(N+K) = 100 - 400 flop per byte), so unlike their
// K <= 256 siblings they are bound by the matrix pipe as much as by memory; MIOpen's kernels reach 25 - 30 % of it.
#include <algorithm>

#include "mrla_device.h"
#include "mrla_kernels.h"

namespace mrla {
namespace {

typedef __bf16 ks_bf16x8 __attribute__((ext_vector_type(8)));
typedef float ks_f32x16 __attribute__((ext_vector_type(16)));

#define MRLA_KS_FLAGS 0x00020000          /* raw buffer descriptor word 3 (as nhwc_rows.h) */
constexpr int kKsWaves = 8;
constexpr int kKsKC = 32;                 // reduction elements per chunk (64-byte rows)
constexpr int kKsStages = 3;
constexpr int kKsNI = 3;                  // DMA wave-instructions per wave and chunk (24 x 16 rows = 384 rows)

template <int WN, int PB>
struct KsGeo {
  static constexpr int WM = kKsWaves / WN;              // waves along the pixels
  static constexpr int TM = WM * PB * 32, TN = WN * 64; // output tile
  static constexpr int ROWS = TM + TN;                  // 64-byte rows per stage (<= 384)
  static constexpr int SB = 384 * 64;                   // stage bytes (the DMA plan always covers 384 rows; the rest is dummy)
  static constexpr int kLds = kKsStages * SB;           // 72 KB
  static constexpr int CPR = TN / 8;                    // 16-byte chunks per output-tile row
  static_assert(ROWS <= 384 && TM * TN * 2 <= kLds - 8192, "stage memory doubles as the output tile");
};

__device__ __forceinline__ unsigned ks_lds_addr(const void* p) {
  return (unsigned)(size_t)((__attribute__((address_space(3))) const char*)p);
}
__device__ __forceinline__ void ks_read16(u32x4& v, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
}
__device__ __forceinline__ void ks_write16(unsigned addr, const u32x4& v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int N>
__device__ __forceinline__ void ks_fence(u32x4& v, bool wait) {
  if (wait) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N) : "memory");
  else asm volatile("" : "+v"(v)::"memory");
}
__device__ __forceinline__ int ks_swz(int row) { return (row >> 2) & 3; }      // slot swizzle of a 64-byte row
__device__ __forceinline__ int ks_oswz(int row) { return row & 7; }            // chunk swizzle of an output-tile row


// Copy the staged bf16 tile [TM][TN] (16-byte chunks swizzled by ks_oswz) out as whole rows; MOM: also the BatchNorm moment
// record of this pixel tile per channel (MRLA_GEMM_MOMENTS: sum (y - p), sum (y - p)^2, p, count; y = the ROUNDED outputs,
// p = the tile's first pixel) -- thread (srow, chunk) keeps eight channels over its rows, the RPI row-threads of a chunk are
// folded through LDS in a fixed order once the tile has been read.
template <int TM, int TN, bool MOM>
__device__ __forceinline__ void ks_store_tile(unsigned char* smem_raw, unsigned lds0, bf16_t* __restrict__ Y,
                                              float* __restrict__ mom_part, int M, int N, int m0, int n0, int tile) {
  constexpr int CPR = TN / 8, RPI = (kKsWaves * kWave) / CPR;
  const int srow = threadIdx.x / CPR, chunk = threadIdx.x % CPR;
  float piv[8], s1[8], s2[8];
  if (MOM) {
    u32x4 pv;
    ks_read16(pv, lds0 + ((chunk ^ ks_oswz(0)) << 4));               // tile row 0
    ks_fence<0>(pv, true);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned wv = j == 0 ? pv.x : j == 1 ? pv.y : j == 2 ? pv.z : pv.w;
      piv[2 * j] = __uint_as_float(wv << 16);
      piv[2 * j + 1] = __uint_as_float(wv & 0xffff0000u);
      s1[2 * j] = s1[2 * j + 1] = s2[2 * j] = s2[2 * j + 1] = 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < TM / RPI; ++i) {
    const int row = srow + RPI * i;
    u32x4 v;
    ks_read16(v, lds0 + row * (TN * 2) + ((chunk ^ ks_oswz(row)) << 4));
    ks_fence<0>(v, true);
    if (m0 + row < M) {
      *reinterpret_cast<u32x4*>(Y + (size_t)(m0 + row) * N + n0 + chunk * 8) = v;
      if (MOM) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned wv = j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w;
          const float d0 = __uint_as_float(wv << 16) - piv[2 * j], d1 = __uint_as_float(wv & 0xffff0000u) - piv[2 * j + 1];
          s1[2 * j] += d0; s2[2 * j] = fmaf(d0, d0, s2[2 * j]);
          s1[2 * j + 1] += d1; s2[2 * j + 1] = fmaf(d1, d1, s2[2 * j + 1]);
        }
      }
    }
  }
  if (MOM) {
    float* red = reinterpret_cast<float*>(smem_raw);              // [RPI][TN][2] + [TN] pivots, over the (now read) tile
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[((srow * TN) + chunk * 8 + j) * 2 + 0] = s1[j];
      red[((srow * TN) + chunk * 8 + j) * 2 + 1] = s2[j];
    }
    if (srow == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[RPI * TN * 2 + chunk * 8 + j] = piv[j];
    }
    __syncthreads();
    if ((int)threadIdx.x < TN) {
      const int ch = threadIdx.x;
      float a = 0.f, b = 0.f;
      for (int q = 0; q < RPI; ++q) { a += red[((q * TN) + ch) * 2 + 0]; b += red[((q * TN) + ch) * 2 + 1]; }
      float4 rec;
      rec.x = a; rec.y = b; rec.z = red[RPI * TN * 2 + ch]; rec.w = (float)min(TM, M - m0);
      *reinterpret_cast<float4*>(mom_part + ((size_t)tile * N + n0 + ch) * 4) = rec;
    }
  }
}

template <int WN, int PB, bool MOM>
__global__ __launch_bounds__(kKsWaves* kWave, 4) void conv1x1_kstream_kernel(const bf16_t* __restrict__ X,
                                                                             const bf16_t* __restrict__ W,
                                                                             bf16_t* __restrict__ Y,
                                                                             float* __restrict__ mom_part, int M, int N,
                                                                             int K, int tiles_m, int groups_n) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef KsGeo<WN, PB> G;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware order: the channel groups of one pixel tile are neighbours in time on one XCD (they share the X chunks)
  const int per = gridDim.x >> 3;
  const int vid = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (vid >= tiles_m * groups_n) return;
  const int tile = vid / groups_n, cg = vid - tile * groups_n;
  const int m0 = tile * G::TM, n0 = cg * G::TN;
  const int wn = wave % WN, wm = wave / WN;
  const int nchunks = K / kKsKC;

  // ---- DMA plan: instruction u = wave + 8*i covers stage rows 16u .. 16u+15 (4 lanes per 64-byte row);
  //      rows [0, TM) are X pixels, [TM, TM+TN) W channels, the rest does not exist (out-of-bounds offset: no traffic) ----
  const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(X), 0, (int)((size_t)M * K * 2), MRLA_KS_FLAGS);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(W), 0, (int)((size_t)N * K * 2), MRLA_KS_FLAGS);
  unsigned voff[kKsNI];
  bool isw[kKsNI];
#pragma unroll
  for (int i = 0; i < kKsNI; ++i) {
    const int u = wave + kKsWaves * i, row = u * 16 + (lane >> 2), slot = (lane & 3) ^ ks_swz(row);
    isw[i] = row >= G::TM;                                     // wave-uniform (TM is a multiple of 16)
    if (row < G::TM) voff[i] = (unsigned)(((size_t)(m0 + row) * K) * 2 + slot * 16);      // (pixels past M: beyond num_records)
    else if (row < G::ROWS) voff[i] = (unsigned)(((size_t)(n0 + row - G::TM) * K) * 2 + slot * 16);
    else voff[i] = 0x80000000u;
  }
  int issued = 0;
  auto issue = [&](int stage) {
    const unsigned kill = issued++ < nchunks ? 0u : 0x80000000u;   // past the reduction: out of bounds, no memory traffic
#pragma unroll
    for (int i = 0; i < kKsNI; ++i) {
      unsigned char* dst = smem_raw + stage * G::SB + (wave + kKsWaves * i) * 1024;
      if (isw[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_ptr)dst, 16, voff[i] | kill, 0, 0, 0);
      else        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_ptr)dst, 16, voff[i] | kill, 0, 0, 0);
      voff[i] += kKsKC * 2;
    }
  };

  // ---- operand addresses: A = W rows of this wave's two 32-channel blocks, B = X rows of its PB pixel blocks ----
  const unsigned lds0 = ks_lds_addr(smem_raw);
  unsigned offA[2], offB[PB];
  int fA[2], fB[PB];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int row = G::TM + wn * 64 + c * 32 + r;
    offA[c] = lds0 + row * 64;
    fA[c] = ks_swz(row);
  }
#pragma unroll
  for (int p = 0; p < PB; ++p) {
    const int row = (wm * PB + p) * 32 + r;
    offB[p] = lds0 + row * 64;
    fB[p] = ks_swz(row);
  }

  ks_f32x16 acc[PB][2];
#pragma unroll
  for (int p = 0; p < PB; ++p)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[p][c][e] = 0.f;

  // ---- pipeline: chunks c+1 (in flight) and c+2 (issued after the barrier of chunk c) ahead of chunk c ----
#pragma unroll
  for (int j = 0; j < kKsStages - 1; ++j) issue(j);
  int stage = 0, nxt = kKsStages - 1;
  for (int c = 0; c < nchunks; ++c) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kKsNI) : "memory");       // my part of chunk c has landed (c+1 may be in flight)
    __builtin_amdgcn_s_barrier();                                     // everybody's has; everybody is done reading chunk c-1
    asm volatile("" ::: "memory");
    issue(nxt);
    nxt = nxt + 1 == kKsStages ? 0 : nxt + 1;
    const unsigned sbo = stage * G::SB;
    stage = stage + 1 == kKsStages ? 0 : stage + 1;
    u32x4 a[2][2], b[2][PB];                                          // [k-step][block]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) ks_read16(a[ks][cb], offA[cb] + sbo + (((2 * ks + h) ^ fA[cb]) << 4));
#pragma unroll
      for (int p = 0; p < PB; ++p) ks_read16(b[ks][p], offB[p] + sbo + (((2 * ks + h) ^ fB[p]) << 4));
    }
    // k-step 0 is complete when at most the (2 + PB) reads of k-step 1 are outstanding
    ks_fence<2 + PB>(a[0][0], true);
    ks_fence<2 + PB>(a[0][1], false);
#pragma unroll
    for (int p = 0; p < PB; ++p) ks_fence<2 + PB>(b[0][p], false);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < PB; ++p)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        acc[p][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ks_bf16x8, a[0][cb]),
                                                             __builtin_bit_cast(ks_bf16x8, b[0][p]), acc[p][cb], 0, 0, 0);
    ks_fence<0>(a[1][0], true);
    ks_fence<0>(a[1][1], false);
#pragma unroll
    for (int p = 0; p < PB; ++p) ks_fence<0>(b[1][p], false);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < PB; ++p)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        acc[p][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ks_bf16x8, a[1][cb]),
                                                             __builtin_bit_cast(ks_bf16x8, b[1][p]), acc[p][cb], 0, 0, 0);
  }
  // the two dummy chunks issued past the end carry no data, but their LDS writes must be over before the tile is staged
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- epilogue: round once, 16-byte pieces of each pixel's row, staged in LDS, whole rows out ----
  // acc[p][cb]: lane = pixel r of block p; register e = channel 8*(e/4) + 4*h + e%4 of the 32-channel block cb
#pragma unroll
  for (int p = 0; p < PB; ++p) {
    const int trow = (wm * PB + p) * 32 + r;
    const unsigned orow = lds0 + trow * (G::TN * 2);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      unsigned q[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 pr;
        pr[0] = from_f<bf16_t>(acc[p][cb][2 * i]);
        pr[1] = from_f<bf16_t>(acc[p][cb][2 * i + 1]);
        q[i] = __builtin_bit_cast(unsigned, pr);
      }
      // half 0 holds channels {0-3, 8-11, 16-19, 24-27}, half 1 the other four groups; after the swaps half 0 holds
      // {0-7, 16-23} and half 1 {8-15, 24-31}: two 16-byte pieces per lane
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const auto sw = __builtin_amdgcn_permlane32_swap(q[4 * g + t], q[4 * g + 2 + t], false, false);
          q[4 * g + t] = sw[0];
          q[4 * g + 2 + t] = sw[1];
        }
      const int ch0 = wn * 8 + cb * 4 + h;                     // 16-byte chunk index of channels 32*cb + 8*h .. (+7)
      ks_write16(orow + (((ch0) ^ ks_oswz(trow)) << 4), (u32x4){q[0], q[1], q[2], q[3]});
      ks_write16(orow + (((ch0 + 2) ^ ks_oswz(trow)) << 4), (u32x4){q[4], q[5], q[6], q[7]});
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  ks_store_tile<G::TM, G::TN, MOM>(smem_raw, lds0, Y, mom_part, M, N, m0, n0, tile);
#endif
}

// ------------------------------------------------------------------------------------------------
// The 256 x 256 tile (round 3, second form).  In the kernel above a wave owns 64 channels x 64 pixels: one ds_read_b128
// per MFMA, and since the CU's LDS port delivers one such read in the time its four matrix pipes take for four MFMAs, LDS
// reads alone cap it at half the MFMA rate (it measures 22 - 31 %).  Here a wave owns 64 channels x 128 pixels (0.75 reads
// per MFMA, 128 accumulator registers), a workgroup 256 x 256 outputs with four 32 KB stages (one workgroup per CU, three
// chunks of DMA in flight), and the operand fragments are double-buffered per k-step: the reads of k-step u+1 are issued
// before the MFMAs of k-step u and fenced after them, so that the LDS latency -- and the barrier that opens a new chunk --
// sit under matrix work instead of in front of it.
// ------------------------------------------------------------------------------------------------
struct Ks256 {
  static constexpr int TM = 256, TN = 256, ROWS = TM + TN, SB = ROWS * 64, ST = 4, NI = 4, PB = 4;
  static constexpr int kLds = ST * SB;                  // 128 KB: also exactly the bf16 output tile
};

template <bool MOM>
__global__ __launch_bounds__(kKsWaves* kWave, 2) void conv1x1_kstream256_kernel(const bf16_t* __restrict__ X,
                                                                               const bf16_t* __restrict__ W,
                                                                               bf16_t* __restrict__ Y,
                                                                               float* __restrict__ mom_part, int M, int N,
                                                                               int K, int tiles_m, int groups_n) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef Ks256 G;
  constexpr int PB = G::PB;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int r = lane & 31, h = lane >> 5;
  const int per = gridDim.x >> 3;
  const int vid = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (vid >= tiles_m * groups_n) return;
  const int tile = vid / groups_n, cg = vid - tile * groups_n;
  const int m0 = tile * G::TM, n0 = cg * G::TN;
  const int wn = wave & 3, wm = wave >> 2;
  const int nchunks = K / kKsKC;

  const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(X), 0, (int)((size_t)M * K * 2), MRLA_KS_FLAGS);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(W), 0, (int)((size_t)N * K * 2), MRLA_KS_FLAGS);
  unsigned voff[G::NI];
#pragma unroll
  for (int i = 0; i < G::NI; ++i) {       // instruction u = wave + 8*i: stage rows 16u .. 16u+15; i < 2: X pixels, else W channels
    const int u = wave + kKsWaves * i, row = u * 16 + (lane >> 2), slot = (lane & 3) ^ ks_swz(row);
    if (i < 2) voff[i] = (unsigned)(((size_t)(m0 + row) * K) * 2 + slot * 16);            // (pixels past M: beyond num_records)
    else voff[i] = (unsigned)(((size_t)(n0 + row - G::TM) * K) * 2 + slot * 16);
  }
  int issued = 0;
  auto issue = [&](int stage) {
    const unsigned kill = issued++ < nchunks ? 0u : 0x80000000u;
#pragma unroll
    for (int i = 0; i < G::NI; ++i) {
      unsigned char* dst = smem_raw + stage * G::SB + (wave + kKsWaves * i) * 1024;
      if (i < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_ptr)dst, 16, voff[i] | kill, 0, 0, 0);
      else       __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_ptr)dst, 16, voff[i] | kill, 0, 0, 0);
      voff[i] += kKsKC * 2;
    }
  };

  const unsigned lds0 = ks_lds_addr(smem_raw);
  unsigned offA[2], offB[PB];
  int fA[2], fB[PB];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int row = G::TM + wn * 64 + c * 32 + r;
    offA[c] = lds0 + row * 64;
    fA[c] = ks_swz(row);
  }
#pragma unroll
  for (int p = 0; p < PB; ++p) {
    const int row = (wm * PB + p) * 32 + r;
    offB[p] = lds0 + row * 64;
    fB[p] = ks_swz(row);
  }
  ks_f32x16 acc[PB][2];
#pragma unroll
  for (int p = 0; p < PB; ++p)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[p][c][e] = 0.f;

  u32x4 fa[2][2], fb[2][PB];                       // [buffer][block]: the fragments of k-step u live in buffer u & 1
  auto read_frags = [&](int buf, unsigned sbo, int ks) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) ks_read16(fa[buf][cb], offA[cb] + sbo + (((2 * ks + h) ^ fA[cb]) << 4));
#pragma unroll
    for (int p = 0; p < PB; ++p) ks_read16(fb[buf][p], offB[p] + sbo + (((2 * ks + h) ^ fB[p]) << 4));
  };
  auto fence_frags = [&](int buf) {
    ks_fence<0>(fa[buf][0], true);
    ks_fence<0>(fa[buf][1], false);
#pragma unroll
    for (int p = 0; p < PB; ++p) ks_fence<0>(fb[buf][p], false);
  };
  auto mfmas = [&](int buf) {
#pragma unroll
    for (int p = 0; p < PB; ++p)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        acc[p][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ks_bf16x8, fa[buf][cb]),
                                                             __builtin_bit_cast(ks_bf16x8, fb[buf][p]), acc[p][cb], 0, 0, 0);
  };

  // chunks 0 .. 2 in flight, chunk 3 follows once chunk 0 is complete everywhere
#pragma unroll
  for (int j = 0; j < G::ST - 1; ++j) issue(j);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G::NI) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  issue(G::ST - 1);
  read_frags(0, 0u, 0);
  fence_frags(0);
  int stage = 0;                                   // stage of the chunk being multiplied
  for (int c = 0; c < nchunks; ++c) {
    const unsigned sbo = stage * G::SB;
    // k-step 0 of chunk c: its fragments are in buffer 0; fetch k-step 1's
    read_frags(1, sbo, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(0);
    __builtin_amdgcn_sched_barrier(0);
    fence_frags(1);
    // k-step 1: every wave has read all of chunk c by now.  Open chunk c+1 (mine has landed when at most the two chunks
    // behind it are in flight; after the barrier everybody's has, and the stage of chunk c is free for chunk c+4), fetch
    // its first fragments, multiply under their latency
    const int nstage = stage + 1 == G::ST ? 0 : stage + 1;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G::NI) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(stage);
    read_frags(0, nstage * G::SB, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(1);
    __builtin_amdgcn_sched_barrier(0);
    fence_frags(0);
    stage = nstage;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (dummy chunks past the end: no data, but their LDS writes)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- epilogue: as above (round once, v_permlane32_swap to 16-byte pieces, whole rows out) ----
#pragma unroll
  for (int p = 0; p < PB; ++p) {
    const int trow = (wm * PB + p) * 32 + r;
    const unsigned orow = lds0 + trow * (G::TN * 2);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      unsigned q[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 pr;
        pr[0] = from_f<bf16_t>(acc[p][cb][2 * i]);
        pr[1] = from_f<bf16_t>(acc[p][cb][2 * i + 1]);
        q[i] = __builtin_bit_cast(unsigned, pr);
      }
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const auto sw = __builtin_amdgcn_permlane32_swap(q[4 * g + t], q[4 * g + 2 + t], false, false);
          q[4 * g + t] = sw[0];
          q[4 * g + 2 + t] = sw[1];
        }
      const int ch0 = wn * 8 + cb * 4 + h;
      ks_write16(orow + (((ch0) ^ ks_oswz(trow)) << 4), (u32x4){q[0], q[1], q[2], q[3]});
      ks_write16(orow + (((ch0 + 2) ^ ks_oswz(trow)) << 4), (u32x4){q[4], q[5], q[6], q[7]});
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  ks_store_tile<G::TM, G::TN, MOM>(smem_raw, lds0, Y, mom_part, M, N, m0, n0, tile);
#endif
}

struct KsPlan {
  int wn = 0, pb = 0, tiles_m = 0, groups_n = 0;
  bool big = false;        // the 256 x 256 tile kernel
};

KsPlan ks_plan(int M, int K, int N) {
  KsPlan p;
  if (M <= 0 || K < 512 || K % kKsKC || N % 128 || (size_t)M * std::max(N, K) * 2 >= (size_t)1 << 31) return p;
  p.wn = N % 256 == 0 ? 4 : 2;
  if (N % 256 == 0 && K >= 1024) {
    // 256 x 256 tiles, one workgroup per CU: when they come close to filling the chip's rounds and the reduction is long
    // enough to amortise a lone workgroup's pipeline fill and 128 KB epilogue (measured, b = 256, us old -> new: 1024 -> 256
    // @14 38.4 -> 35.7, 1024 -> 512 @14 77.5 -> 70.9; but K = 512: 512 -> 1024 @14 66.6 -> 82.9, 512 -> 2048 @7 39.2 -> 42.5)
    const int tiles = ((M + 255) / 256) * (N / 256);
    const int rounds = (tiles + 255) / 256;
    if (tiles >= 160 && tiles * 100 >= rounds * 256 * 70) {
      p.big = true; p.pb = 4; p.tiles_m = (M + 255) / 256; p.groups_n = N / 256;
      return p;
    }
  }
  const int tn = p.wn * 64, wm = kKsWaves / p.wn;
  p.groups_n = N / tn;
  // the larger pixel tile (less weight traffic out of L2 per X byte) when it still gives every CU a workgroup
  const int tm2 = wm * 64;
  p.pb = ((M + tm2 - 1) / tm2) * p.groups_n >= 300 ? 2 : 1;
  const int tm = wm * p.pb * 32;
  p.tiles_m = (M + tm - 1) / tm;
  return p;
}

template <int WN, int PB, bool MOM>
int ks_launch_m(const KsPlan& p, const void* x, const void* w, void* y, float* part, int M, int K, int N, hipStream_t st) {
  typedef KsGeo<WN, PB> G;
  if (lds_opt_in(reinterpret_cast<const void*>(conv1x1_kstream_kernel<WN, PB, MOM>), G::kLds) != hipSuccess) return MRLA_EHIP;
  hipLaunchKernelGGL((conv1x1_kstream_kernel<WN, PB, MOM>), dim3((p.tiles_m * p.groups_n + 7) / 8 * 8), dim3(kKsWaves * kWave),
                     G::kLds, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, part, M, N, K, p.tiles_m, p.groups_n);
  return hip_status(hipGetLastError());
}
template <int WN, int PB>
int ks_launch(const KsPlan& p, const void* x, const void* w, void* y, float* part, int M, int K, int N, hipStream_t st) {
  return part ? ks_launch_m<WN, PB, true>(p, x, w, y, part, M, K, N, st) : ks_launch_m<WN, PB, false>(p, x, w, y, part, M, K, N, st);
}
template <bool MOM>
int ks_launch256(const KsPlan& p, const void* x, const void* w, void* y, float* part, int M, int K, int N, hipStream_t st) {
  if (lds_opt_in(reinterpret_cast<const void*>(conv1x1_kstream256_kernel<MOM>), Ks256::kLds) != hipSuccess) return MRLA_EHIP;
  hipLaunchKernelGGL(conv1x1_kstream256_kernel<MOM>, dim3((p.tiles_m * p.groups_n + 7) / 8 * 8), dim3(kKsWaves * kWave),
                     Ks256::kLds, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, part, M, N, K, p.tiles_m, p.groups_n);
  return hip_status(hipGetLastError());
}

}  // namespace

int conv1x1_kstream_supported(int M, int K, int N) { return ks_plan(M, K, N).wn ? 1 : 0; }
int conv1x1_kstream_rows(int M, int K, int N) { return ks_plan(M, K, N).tiles_m; }      // pixel tiles = record rows
int conv1x1_kstream_stages(int M, int K, int N) { return ks_plan(M, K, N).big ? Ks256::ST : kKsStages; }

// part != null: moment records [conv1x1_kstream_rows()][N][MRLA_GEMM_MOMENTS] of the rounded outputs (one row per pixel tile)
int launch_conv1x1_kstream(const void* x, const void* w, void* y, float* part, int M, int K, int N, hipStream_t st) {
  const KsPlan p = ks_plan(M, K, N);
  if (!p.wn) return MRLA_EUNSUPPORTED;
  if (p.big) return part ? ks_launch256<true>(p, x, w, y, part, M, K, N, st) : ks_launch256<false>(p, x, w, y, part, M, K, N, st);
  if (p.wn == 4) return p.pb == 2 ? ks_launch<4, 2>(p, x, w, y, part, M, K, N, st) : ks_launch<4, 1>(p, x, w, y, part, M, K, N, st);
  return p.pb == 2 ? ks_launch<2, 2>(p, x, w, y, part, M, K, N, st) : ks_launch<2, 1>(p, x, w, y, part, M, K, N, st);
}

}  // namespace mrla
