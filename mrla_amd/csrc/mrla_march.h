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
  float lmask, rmask;   // 0 at the left / right plane edge, else 1
};

__device__ __forceinline__ LaneTask make_task(const SlabGeo& g, int task, int np, int lane) {
  LaneTask t;
  const int grp = task / g.NB;
  const int band = task - grp * g.NB;
  t.r0 = band * g.RB;
  t.r1 = min(g.H, t.r0 + g.RB);
  t.pl = lane / g.W;
  t.col = lane - t.pl * g.W;
  t.p = grp * g.PW + t.pl;
  t.live = grp * g.PW < np;
  t.valid = (t.pl < g.PW) && (t.p < np);
  if (!t.valid) { t.p = t.live ? grp * g.PW : 0; t.col = 0; }   // park on a real plane (parameters are indexed by it)
  t.lmask = (t.col > 0) ? 1.f : 0.f;
  t.rmask = (t.col < g.W - 1) ? 1.f : 0.f;
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
