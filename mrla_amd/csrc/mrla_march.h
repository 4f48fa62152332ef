// Row-marching helpers shared by the NCHW stencil kernels: a wave walks down the rows of PW side-by-side
// channel planes held in LDS (lanes = columns), keeping vertical neighbours in registers and fetching
// horizontal neighbours with DPP wave shifts.
#pragma once
#include "mrla_device.h"
#include "mrla_kernels.h"

namespace mrla {

// Per-lane description of the plane row segment a lane works on inside one wave task.
struct LaneTask {
  int p;        // plane index inside the slab
  int col;      // column
  int r0, r1;   // row band [r0, r1)
  int pl;       // plane slot inside the wave group
  bool valid;
  bool live;     // wave-uniform: this (plane group, band) holds at least one real plane
  bool last;     // last lane of the plane's lane segment (receives seg_sum results)
  float lmask, rmask;   // 0 at the left / right plane edge, else 1
};

// Image- and task-independent part of the lane mapping (one integer division per kernel, not per task).
struct LaneMap { int pl, col; float lmask, rmask; };
__device__ __forceinline__ LaneMap make_lane_map(const SlabGeo& g, int lane) {
  LaneMap m;
  m.pl = lane / g.WS;                 // WS = power of two >= W: plane segments are aligned for the DPP reductions
  m.col = lane & (g.WS - 1);
  m.lmask = (m.col > 0) ? 1.f : 0.f;
  m.rmask = (m.col < g.W - 1) ? 1.f : 0.f;
  return m;
}
__device__ __forceinline__ LaneTask make_task(const SlabGeo& g, const LaneMap& m, int task, int np) {
  LaneTask t;
  const int grp = task / g.NB;
  const int band = task - grp * g.NB;
  t.r0 = band * g.RB;
  t.r1 = min(g.H, t.r0 + g.RB);
  t.pl = m.pl;
  t.col = m.col;
  t.p = grp * g.PW + t.pl;
  t.live = grp * g.PW < np;
  t.valid = (t.col < g.W) && (t.p < np);
  t.lmask = m.lmask;
  t.rmask = m.rmask;
  if (!t.valid) { t.p = t.live ? grp * g.PW : 0; t.col = 0; t.lmask = 0.f; t.rmask = 1.f; }
  t.last = m.col == g.WS - 1;          // lane that holds the segment sums after seg_sum()
  return t;
}

// One row of x for this lane's column plus its two horizontal neighbours.
struct Row3 { float l, c, r; };

template <typename T>
__device__ __forceinline__ Row3 load_row3(const T* __restrict__ plane, int r, const SlabGeo& g, const LaneTask& t) {
  Row3 v;
  v.c = (r >= 0 && r < g.H) ? to_f(plane[r * g.W + t.col]) : 0.f;
  v.l = lane_prev(v.c) * t.lmask;
  v.r = lane_next(v.c) * t.rmask;
  return v;
}

// Branch-free variant used by the pipelined kernels: the centre value only (clamped address, zero outside the
// plane); horizontal neighbours are taken with DPP by the caller and plane edges are handled by MASKED WEIGHTS
// (mask_conv / mask_convT below), not by per-row mask multiplies.
template <typename T>
__device__ __forceinline__ float ld_centre(const T* __restrict__ plane_col, int r, int H, int W) {
  const int rc = min(max(r, 0), H - 1);
  const float v = to_f(plane_col[rc * W]);
  return (r >= 0 && r < H) ? v : 0.f;
}
// Same, for a row cursor: `idx` = r*W (element offset of row r), `last` = (H-1)*W.
template <typename T>
__device__ __forceinline__ float ld_centre_at(const T* __restrict__ plane_col, int idx, int last) {
  const float v = to_f(plane_col[min(max(idx, 0), last)]);
  return (idx >= 0 && idx <= last) ? v : 0.f;
}
__device__ __forceinline__ Row3 row_of(float c) {
  Row3 v;
  v.c = c;
  v.l = lane_prev(c);
  v.r = lane_next(c);
  return v;
}
// conv:   out[w] = sum w[i][j] * x[..][w + j - 1]  -> left taps (j = 0) vanish at col 0, right taps at col W-1
__device__ __forceinline__ void mask_conv(float (&w)[9], const LaneTask& t) {
  w[0] *= t.lmask; w[3] *= t.lmask; w[6] *= t.lmask;
  w[2] *= t.rmask; w[5] *= t.rmask; w[8] *= t.rmask;
}
// conv^T: dx[w] = sum w[i][j] * dU[..][w - j + 1] -> taps j = 0 read the RIGHT neighbour, j = 2 the LEFT one
__device__ __forceinline__ void mask_convT(float (&w)[9], const LaneTask& t) {
  w[0] *= t.rmask; w[3] *= t.rmask; w[6] *= t.rmask;
  w[2] *= t.lmask; w[5] *= t.lmask; w[8] *= t.lmask;
}

__device__ __forceinline__ float conv9(const float (&w)[9], const Row3& a, const Row3& b, const Row3& c) {
  float s = w[0] * a.l;
  s = fmaf(w[1], a.c, s); s = fmaf(w[2], a.r, s);
  s = fmaf(w[3], b.l, s); s = fmaf(w[4], b.c, s); s = fmaf(w[5], b.r, s);
  s = fmaf(w[6], c.l, s); s = fmaf(w[7], c.c, s); s = fmaf(w[8], c.r, s);
  return s;
}

__device__ __forceinline__ void load_w9(float (&w)[9], const float* __restrict__ wv, int c) {
#pragma unroll
  for (int i = 0; i < 9; ++i) w[i] = wv[c * 9 + i];
}

}  // namespace mrla
