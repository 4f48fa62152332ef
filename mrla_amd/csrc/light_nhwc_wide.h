// Shared pieces of the row-pipeline kernels for channels_last activations with C % 64 == 0 (light_nhwc_wide.hip: the passes
// that read x_t; light_nhwc_lean.hip: the backward passes that re-form x_t from conv3's output and the shortcut): build
// flags, the workgroup prologue, the fused producer's x_t formula, LDS budgets and the launch geometry.
#pragma once
#include "light_nhwc.h"
#include "nhwc_rows.h"

#ifndef MRLA_REVERSE_APPLY
#define MRLA_REVERSE_APPLY 1
#endif
// 1: the instances of light_apply_bwd_wide without GELU run in packed FP32 (experiments/light_apply_bwd_pk.h: 36 instead of 50
// VALU instructions per element at the SAME time per launch -- the pass is not issue-bound; measured, not adopted:
// profiles/r06_notes.md section 2); 0 (product): the plain kernel
#ifndef MRLA_APPLY_BWD_PK
#define MRLA_APPLY_BWD_PK 0
#endif
// row sets in flight per wave in the packed apply_bwd, 16-bit types (experiments/light_apply_bwd_pk.h)
#ifndef MRLA_APPLY_BWD_DEPTH
#define MRLA_APPLY_BWD_DEPTH 2
#endif
#ifndef MRLA_STREAM_MB
#define MRLA_STREAM_MB 128
#endif
// Channel groups side by side in apply_bwd's workgroups on 28-wide maps (4 strips, >= 8 channel groups).  4 (round 4): two
// strips at a time, the other two walked afterwards -- the halo columns between strips 1 and 2 are then fetched twice, far
// apart in time: PMC 1 390 MB per launch at 512 x 28^2, b = 256 = 1.13 x its 6N (the 56-wide stage, whose eight strips run
// side by side: 1.04 x; 14-wide: 1.07 x; 7-wide: 1.03 x).  2 (round 5): all four strips side by side: 1 291 MB = 1.05 x, at
// the same 275 - 282 us per launch (profiles/r05_notes.md section 9, profiles/r05_sq_counters_row_pipeline.md).  The lighter passes keep 4 (their row pieces overlap
// by two columns, not four, and they gained 9 - 16 % from the wider contiguous request).
#ifndef MRLA_APPLY_BWD_WC_4STRIPS
#define MRLA_APPLY_BWD_WC_4STRIPS 2
#endif
// Experiment switch (scripts/build_variant.sh only; the product builds with 0): leave the dWv sums out of the backward apply
// passes -- 63 of their ~350 / ~427 vector instructions per row step -- to see which of them the vector pipe bounds
// (profiles/r06_notes.md section 2).  The results are WRONG with 1.
#ifndef MRLA_EXP_SKIP_WG
#define MRLA_EXP_SKIP_WG 0
#endif
// Occupancy of the fused forward statistics pass (round 4, profiles/r04_notes.md section 5): at 150 - 162 VGPRs three waves
// fit a SIMD, i.e. ONE eight-wave workgroup per CU on the 56-wide stage -- half the waves apply_fwd keeps in flight on the
// same 3N bytes (SQ counters side by side: 37 % of its wave cycles wait against apply_fwd's 71 %: too few waves, 2.4 x the
// vector instructions).  An instance capped at 128 registers (four waves per SIMD; 19 - 26 values spilled) ran the 56-wide
// launch 305 -> 280 us ALONE, but inside the training step it bought nothing measurable (123.0 us per launch on average
// with it, 122.8 us without) and cost 20 MB of HBM traffic per launch (558 -> 578 MB: the spills are traffic) -- measured, not
// kept; four-wave workgroups walking two strips each: 3 % slower.
// Cache policy of the row fetches (template AUX: 0 = default, 2 = nt / streaming) and image order, measured in the
// training step (b = 256, same box, GB/s):            stats_fwd_fused  apply_fwd  stats_bwd  apply_bwd
//   default policy, images in launch order                  4069         4681       4773       4846
//   nt fetches in the three 3N passes (tensors >= 128 MB)   4273         4789       5034       4636
//   ... and apply_* walking the images in reverse order     4258         4991       5017       4667
// `nt` keeps a pass from competing for the Infinity Cache with the write-back of its predecessor's lines (+5 %), and the
// pass that follows the statistics pass finds the END of the tensors in the cache, so it starts there (apply_fwd +4 %).
// But apply_bwd loses 4 % when stats_bwd streams (it lives off what stats_bwd leaves behind), and its own 9- / 11-pixel
// row pieces overlap between neighbouring strips, which `nt` re-fetches from HBM.  Hence: nt for the two FORWARD passes
// on tensors far beyond the cache, default policy for both backward passes, reverse image order in both apply passes.

namespace mrla {

// Per-wave LDS row buffers follow the cross-wave reduction area.
// The waves of a workgroup are `wc` NEIGHBOURING channel groups x (waves / wc) strips side by side (wave = strip slot * wc +
// channel-group slot); gridDim.z splits the strips further in the passes that keep no sums over them.  See wide_shape().
// (ZS of NZS: this workgroup's strip range -- blockIdx.z of gridDim.z, unless the rows are cut as well: RowCut below)
#define MRLA_WIDE_PROLOGUE_Z(NRED, WAVE_BYTES, ZS, NZS)                                                   \
  extern __shared__ __align__(16) unsigned char smem_raw[];                                               \
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;    \
  float* red = reinterpret_cast<float*>(smem_raw);                                                        \
  unsigned char* wbuf = smem_raw + (size_t)nwaves * (NRED) * kWave * sizeof(float) + (size_t)wave * (WAVE_BYTES); \
  const int cbase = (blockIdx.x * wc + wave % wc) * kWave;                                                \
  const int c = cbase + lane;                                                                             \
  const int nstrips = (W + kS - 1) / kS;                                                                  \
  const int sfirst = (ZS) * (nwaves / wc) + wave / wc, sstep = (NZS) * (nwaves / wc);                     \
  const int rowelems = W * C;                                                                             \
  (void)red;
#define MRLA_WIDE_PROLOGUE(NRED, WAVE_BYTES) MRLA_WIDE_PROLOGUE_Z(NRED, WAVE_BYTES, blockIdx.z, gridDim.z)

// Row ranges (round 6).  The row pipeline's parallelism is images x channel groups x strips; a detection batch (2 images of
// 256 x 200 x 336) has 384 waves of that kind for 2 048 slots.  Where a launch would leave most CUs without a workgroup even
// after its strip rounds went to gridDim.z, the ROWS of an image are cut into `nzr` ranges as well: gridDim.z = strip ranges x
// row ranges, blockIdx.z = row range * strip ranges + strip range.  A workgroup then walks rows [r0, r0 + n) of its images in
// a frame that starts at r0: the halo rows above and below are fetched from the neighbouring ranges' rows (the image's rows
// are [lo, hi) = [-r0, H - r0) in that frame) instead of arriving as zeros, sums are kept over the owned rows only, and every
// range leaves its own partial record / row, exactly as the strip ranges do.  CUT = false is the kernel as it was: one range,
// r0 = 0, every expression below a constant or H itself.
template <bool CUT> struct RowCut;
template <> struct RowCut<false> {
  int H;
  __device__ __forceinline__ RowCut(int H_, int, int) : H(H_) {}
  __device__ __forceinline__ int zs() const { return blockIdx.z; }
  __device__ __forceinline__ int nzs() const { return gridDim.z; }
  __device__ __forceinline__ int r0() const { return 0; }
  __device__ __forceinline__ int n() const { return H; }
  __device__ __forceinline__ int lo() const { return 0; }
  __device__ __forceinline__ int hi() const { return H; }
  __device__ __forceinline__ bool above() const { return false; }      // rows above the range that belong to the image
};
template <> struct RowCut<true> {
  int zs_, nzs_, r0_, n_, hi_;
  __device__ __forceinline__ RowCut(int H, int nzr, int) {
    nzs_ = gridDim.z / nzr;
    zs_ = blockIdx.z % nzs_;
    const int per = (H + nzr - 1) / nzr;
    r0_ = (int)(blockIdx.z / nzs_) * per;
    n_ = min(per, H - r0_);
    hi_ = H - r0_;
  }
  __device__ __forceinline__ int zs() const { return zs_; }
  __device__ __forceinline__ int nzs() const { return nzs_; }
  __device__ __forceinline__ int r0() const { return r0_; }
  __device__ __forceinline__ int n() const { return n_; }
  __device__ __forceinline__ int lo() const { return -r0_; }
  __device__ __forceinline__ int hi() const { return hi_; }
  __device__ __forceinline__ bool above() const { return r0_ > 0; }
};

// Runs step(r, A, B, C) for r = 0 .. n-1 with the three row windows rotating by name.
#define MRLA_ROTATE3(n, step, A, B, C)                       \
  {                                                          \
    int r_ = 0;                                              \
    for (; r_ + 3 <= (n); r_ += 3) {                         \
      step(r_, A, B, C);                                     \
      step(r_ + 1, B, C, A);                                 \
      step(r_ + 2, C, A, B);                                 \
    }                                                        \
    if (r_ < (n)) {                                          \
      step(r_, A, B, C);                                     \
      if (r_ + 1 < (n)) step(r_ + 1, B, C, A);               \
    }                                                        \
  }

// The shift per window column: `ash` inside the image, 0 outside.  For whole strips only the two halo columns can be
// outside, so three values stand for the nine (RAGGED strips keep the full table).
template <bool RAGGED> struct ColumnShifts;
template <> struct ColumnShifts<true> {
  float v[kS + 2];
  __device__ __forceinline__ void set(float ash, int s0, int W) {
#pragma unroll
    for (int j = 0; j < kS + 2; ++j) {
      const int col = s0 - 1 + j;
      v[j] = (col >= 0 && col < W) ? ash : 0.f;      // wave-uniform predicate
    }
  }
  __device__ __forceinline__ float at(int j) const { return v[j]; }
};
template <> struct ColumnShifts<false> {
  float first, mid, last;
  __device__ __forceinline__ void set(float ash, int s0, int W) {
    first = s0 > 0 ? ash : 0.f;
    mid = ash;
    last = s0 + kS < W ? ash : 0.f;
  }
  __device__ __forceinline__ float at(int j) const { return j == 0 ? first : (j == kS + 1 ? last : mid); }
};

template <typename T, bool AFF, bool RAGGED>
__device__ __forceinline__ void form_x_row(const RawRow<kS + 2>& pre, const RawRow<kS + 2>& o, float asc,
                                           const ColumnShifts<RAGGED>& sh, float (&dst)[kS + 2]) {
#pragma unroll
  for (int j = 0; j < kS + 2; ++j) {
    const float z = AFF ? to_f(from_f<T>(fmaf(asc, pre.v[j], sh.at(j)))) : pre.v[j];
    dst[j] = fmaxf(to_f(from_f<T>(z + o.v[j])), 0.f);
  }
}

// per wave: pre row, two o rows (row r is read again when V[r] is paired with it), one store buffer
template <typename T> constexpr int fused_wave_bytes() { return 3 * RowIO<T, kS + 2>::kBytes + RowIO<T, kS>::kBytes; }

template <typename T, bool PRE = false> constexpr int apply_bwd_wave_bytes() {
  return RowIO<T, kS + 4>::kBytes + 3 * RowIO<T, kS + 2>::kBytes + (PRE ? 3 : 2) * RowIO<T, kS>::kBytes;
}

// (Measured and rejected, profiles/r02_notes.md: 4-wave workgroups under a 168-register budget -- three waves per SIMD, a
// wave walking two strips of the 56-wide stage -- 25 % slower; the 168-register cap alone on these 8-wave workgroups 8 %
// slower than the 171 registers the compiler picks by itself.)
constexpr int kBwdWaves = kMaxStrips;
// Launch geometry.  A workgroup's waves are `wc` NEIGHBOURING channel groups x `ws` strips side by side: the row pieces a CU
// has in flight at one time are then contiguous wc x 128 B per pixel, and a plain copy in this geometry runs 5 - 13 % faster
// than with the waves spread over the strips of ONE channel group (scripts/micro/stream.hip; the passes themselves:
// profiles/r04_notes.md section 9).  The strips a workgroup's waves do not cover side by side are walked by the same waves
// one after the other (passes with sums over the plane: the sums stay inside the workgroup) or go to gridDim.z (`split`).
struct WideLaunch { dim3 grid, block; size_t lds; int BG, wc; };
enum WidePass { P_STATS_FUSED, P_STATS_FWD, P_APPLY_FWD, P_STATS_BWD, P_APPLY_BWD };   // (P_APPLY_BWD: also the MRLA-base value backward)
// Waves side by side in a workgroup: wc channel groups x ws strips.  Measured per pass and stage shape (b = 256, bf16,
// scripts/wc_sweep.sh, profiles/r04_notes.md section 9):
//   * wc = 4 wherever there are >= 8 channel groups (>= 512 channels): -9 ... -16 % per launch; wc = 8 is no better;
//   * 256 channels (56-wide maps): -3 % at wc = 2 for the passes without a dWv reduction, nothing for apply_bwd;
//   * the fused forward statistics pass keeps 3 waves per SIMD (150 - 162 registers): FOUR-wave workgroups fill the CU
//     (three of them) where one eight-wave workgroup leaves a third of the slots empty, so below 8 strips it takes
//     4 / strips channel groups (14-wide: 81 -> 66 us, 7-wide: 37 -> 34 us; 28-wide: 4 strips of one group, as before).
static inline void wide_shape(WidePass pass, int C, int W, int* wc_out, int* ws_out) {
  const int ncg = C / kWave, nstrips = (W + kS - 1) / kS;
  int wc, waves = kMaxStrips;
  switch (pass) {
    case P_STATS_FUSED:
      if (nstrips >= kMaxStrips) { wc = 1; break; }
      waves = 4;
      wc = nstrips >= 4 ? 1 : nstrips >= 2 ? 2 : 4;
      break;
    case P_APPLY_BWD: wc = ncg >= 8 ? (nstrips == 4 ? MRLA_APPLY_BWD_WC_4STRIPS : 4) : 1; break;
    default:          wc = ncg >= 8 ? 4 : 2; break;
  }
  wc = std::max(1, std::min(wc, waves));
  while (ncg % wc) wc >>= 1;
  *wc_out = wc;
  *ws_out = std::max(1, std::min(waves / wc, nstrips));
}

struct ZRanges { int strips, rows; };              // gridDim.z = strips x rows
int nhwc_images_per_group(int B, int C, int W);
// gridDim.z of the passes that keep sums (strip ranges x row ranges; the public queries return strips * rows):
ZRanges nhwc_wgrad_zranges(int B, int C, int H, int W);     // the dWv-producing backward passes (partial rows = image groups x z)
ZRanges nhwc_bmom_zranges(int B, int C, int H, int W);      // the backward statistics pass (partial records)
ZRanges nhwc_mom_zranges(int B, int C, int H, int W);       // the forward statistics passes (records mom[z], merged into mom[0])
int nhwc_wgrad_ranges(int B, int C, int H, int W);
int nhwc_bmom_ranges(int B, int C, int H, int W);
int nhwc_mom_ranges(int B, int C, int H, int W);

// split: every strip round of an image gets its own workgroup (gridDim.z; passes that keep no sums over the plane);
// nz > 1: the strips are cut into that many ranges (passes WITH sums: every range leaves its own partial record / row).
// nz_rows: row ranges (RowCut; the kernel gets the same number as `nzr`).
static inline WideLaunch wide_launch(WidePass pass, int B, int C, int W, int nred, size_t wave_bytes, int bg, bool split,
                                     int nz_ranges = 1, int nz_rows = 1) {
  WideLaunch L;
  const int ncg = C / kWave, nstrips = (W + kS - 1) / kS;
  int wc, ws;
  wide_shape(pass, C, W, &wc, &ws);
  const int nz = split ? (nstrips + ws - 1) / ws : std::max(1, nz_ranges);
  if (bg <= 0) bg = (int)std::max(1L, std::min(8L, (long)B * (ncg / wc) * nz / 2048));
  L.wc = wc;
  L.BG = bg;
  L.grid = dim3(ncg / wc, (B + bg - 1) / bg, nz * std::max(1, nz_rows));
  L.block = dim3(wc * ws * kWave);
  L.lds = (size_t)wc * ws * ((size_t)nred * kWave * sizeof(float) + wave_bytes);
  return L;
}

// Strip ranges (gridDim.z) of the passes that keep sums over the plane.  At the classification batches a workgroup walks all
// strips of its images and the grid still fills the chip several times (1 range).  A detection batch does not: 2 images of
// 256 x 200 x 336 are 4 channel groups x 2 images = 8 workgroups of eight waves walking 48 strips six rounds each on 256 CUs
// (round 6: mrla_light_apply_bwd 781 us per launch = 0.44 TB/s, mrla_light_stats_bwd 0.15 TB/s at 2 x 3 x 800 x 1344).  So when
// the launch would leave most CUs without a workgroup, the strip rounds are spread over gridDim.z -- each range writes its
// own partial row (dWv / bn3 sums: mrla_light_wgrad_rows counts them) or partial record (mrla_light_bmom_splits), summed
// by the consumers in range order.
static inline int wide_strip_ranges(WidePass pass, int B, int C, int W, int bg) {
  int wc, ws;
  wide_shape(pass, C, W, &wc, &ws);
  const int nstrips = (W + kS - 1) / kS, rounds = (nstrips + ws - 1) / ws;
  if (rounds <= 1) return 1;
  if (bg <= 0) bg = (int)std::max(1L, std::min(8L, (long)B * (C / kWave / wc) / 2048));
  const long wgs = (long)(C / kWave / wc) * ((B + bg - 1) / bg);
  if (wgs >= 256) return 1;                            // a workgroup per CU already
  return (int)std::min<long>(rounds, (512 + wgs - 1) / wgs);
}

// Row ranges of a launch that has `wgs` workgroups after its strip rounds were spread (see RowCut): none when every CU has
// a workgroup, or when the tensor is too small for the launch to matter (below kMinCutElems elements a pass moves < 100 MB);
// otherwise towards kCutTargetWgs workgroups, ranges of at least kMinCutRows rows (each range re-fetches two halo rows and
// waits for its first rows alone, and the backward apply pass re-computes one row of dU), every range non-empty.
// Measured on the detection backbone's step (2 x 3 x 800 x 1344, scripts/r06_cut_sweep.sh, ms per step, two rounds):
// (rows >= 8, 512 workgroups) 9.05 / 9.03, (12, 384) 8.88 / 8.88, (16, 256) 9.22 / 9.31, (16, 512) 9.21 / 9.22,
// (24, 512) 9.02 / 9.08, (8, 768) 9.37 / 9.32; uncut: 11.4.
// nhwc_row_cut_mode(): 0 = as described, 1 = never, 2 = wherever an image has >= 16 rows, ranges of >= 8 (the parity tests
// run small shapes through the cut kernels with it: mrla_tuning_row_ranges()).
#ifndef MRLA_CUT_MIN_ROWS
#define MRLA_CUT_MIN_ROWS 12
#endif
#ifndef MRLA_CUT_TARGET_WGS
#define MRLA_CUT_TARGET_WGS 384
#endif
constexpr int kMinCutRows = MRLA_CUT_MIN_ROWS;
constexpr long kCutTargetWgs = MRLA_CUT_TARGET_WGS;
constexpr long kMinCutElems = 8L << 20;
int nhwc_row_cut_mode();
static inline int wide_row_ranges(long wgs, int H, long elems) {
  const int mode = nhwc_row_cut_mode();
  const int min_rows = mode == 2 ? 8 : kMinCutRows;
  const long target = mode == 2 ? 512 : kCutTargetWgs;
  if (mode == 1 || H < 2 * min_rows) return 1;
  if (mode == 0 && (wgs >= 256 || elems < kMinCutElems)) return 1;
  const int n = (int)std::min<long>(std::max(2L, (target + wgs - 1) / wgs), H / min_rows);
  if (n <= 1) return 1;
  const int per = (H + n - 1) / n;
  return (H + per - 1) / per;
}
// workgroups of a launch of `pass` with `nzs` strip ranges (bg <= 0: wide_launch()'s own choice)
static inline long wide_workgroups(WidePass pass, int B, int C, int W, int bg, int nzs) {
  int wc, ws;
  wide_shape(pass, C, W, &wc, &ws);
  const int ncg = C / kWave;
  if (bg <= 0) bg = (int)std::max(1L, std::min(8L, (long)B * (ncg / wc) * nzs / 2048));
  return (long)(ncg / wc) * ((B + bg - 1) / bg) * nzs;
}
// strip rounds of a pass without sums (wide_launch(split = true) gives each its own workgroup)
static inline int wide_strip_rounds(WidePass pass, int C, int W) {
  int wc, ws;
  wide_shape(pass, C, W, &wc, &ws);
  return ((W + kS - 1) / kS + ws - 1) / ws;
}

// tensors of this size and beyond are fetched `nt` by the 3N passes (see the note at the top of the file)
static inline bool stream_fetches(int B, int C, int H, int W, size_t elem) {
  return (size_t)B * C * H * W * elem >= ((size_t)MRLA_STREAM_MB << 20);
}

}  // namespace mrla
