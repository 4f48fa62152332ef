// MRLA-light streaming kernels for channels_last activations with C % 64 == 0 -- every stage of the networks here --
// on the LDS-DMA row pipeline of nhwc_rows.h.  Same passes and math as light_nhwc.hip (which keeps serving other channel
// counts with plain per-lane accesses); reference: resnet/models/modules/mrla_light_module.py:52-74,
// resnet/models/resnet_mrla_light.py:40-43,113-116.
//
// The loops are written for instruction count (the register-staged kernels of round 1 were bound by vector issue,
// profiles/r02_notes.md): row windows rotate by NAME (three steps per trip, no register copies), everything outside the
// image arrives as zeros from the buffer bounds check (no predicates in the arithmetic), bf16 rows need no conversion
// instruction, and the file is compiled without the SLP vectoriser (its v_pk_fma_f32 pairs cost more moves than they save).
// What bounds them on this pipeline is WHICH BYTES A CU REQUESTS AT ONE TIME (profiles/r04_notes.md section 9): hence the
// workgroups of neighbouring channel groups, wide_shape() below.
#include <atomic>
#include "light_nhwc_wide.h"

namespace mrla {

// ------------------------------------------------------------------------------------------------
// Pivoted forward moments (mrla_device.h: the M_REC record).  A strip accumulates its sums over V and o about ITS first
// values (pV, po) -- samples of the plane, so nothing cancels when |mean| >> sigma; strips of a wave and then the waves of
// the workgroup are merged by re-basing onto the first one's pivots (exact algebra on the shifted sums).
// ------------------------------------------------------------------------------------------------
constexpr int kMomRed = M_N + 3;                     // per-wave reduction record: the six sums, pV, po, pixel count
struct WaveMoments {
  float s[M_N], pv, po, n;
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int k = 0; k < M_N; ++k) s[k] = 0.f;
    pv = 0.f; po = 0.f; n = 0.f;
  }
  // fold in `a` (sums about (apv, apo) over an pixels)
  __device__ __forceinline__ void merge(float (&a)[M_N], float apv, float apo, float an) {
    if (n == 0.f) { pv = apv; po = apo; }            // (lanes agree: n is wave-uniform)
    rebase_moments(a, an, apv, apo, pv, po);
#pragma unroll
    for (int k = 0; k < M_N; ++k) s[k] += a[k];
    n += an;
  }
};
// Workgroup reduction over the waves that hold the same channels (v % wc equal), in wave order; waves 0 .. wc-1 write the
// records of (image b, their channels c).
__device__ __forceinline__ void store_moments(WaveMoments& w, float* __restrict__ red, float* __restrict__ mom, int lane,
                                              int wave, int nwaves, int wc) {
  if (nwaves > wc) {
    __syncthreads();
    float* mine = red + (size_t)wave * kMomRed * kWave;
#pragma unroll
    for (int k = 0; k < M_N; ++k) mine[k * kWave + lane] = w.s[k];
    mine[M_N * kWave + lane] = w.pv; mine[(M_N + 1) * kWave + lane] = w.po; mine[(M_N + 2) * kWave + lane] = w.n;
    __syncthreads();
    if (wave < wc) {
      for (int v = wave + wc; v < nwaves; v += wc) {
        const float* o = red + (size_t)v * kMomRed * kWave;
        float a[M_N];
#pragma unroll
        for (int k = 0; k < M_N; ++k) a[k] = o[k * kWave + lane];
        const float an = o[(M_N + 2) * kWave + lane];
        if (an > 0.f) w.merge(a, o[M_N * kWave + lane], o[(M_N + 1) * kWave + lane], an);
      }
    }
  }
  if (wave < wc) {
#pragma unroll
    for (int k = 0; k < M_N; ++k) mom[k] = w.s[k];
    mom[M_PV] = w.pv;
    mom[M_PO] = w.po;
  }
}

// The forward statistics passes spread an image's strip rounds -- and, beyond that, its rows (RowCut) -- over gridDim.z ranges
// when the launch would otherwise leave most of the chip idle (detection batches); range z leaves its record at mom[z][b][c].
// This kernel folds ranges 1 .. nz-1 into record 0 -- the one every consumer reads -- by the algebra of WaveMoments::merge
// (re-based onto range 0's pivots, in range order); z = row range * nzs + strip range, n_z = pixels of range z = (strip rounds
// r with r % nzs == strip range, `ws` strips each) x (rows of the row range).
__global__ __launch_bounds__(kThreads) void light_mom_merge_kernel(float* __restrict__ mom, int B, int C, int H, int W, int ws,
                                                                   int nzs, int nzr) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= B * C) return;
  const int nstrips = (W + kS - 1) / kS;
  const int per = (H + nzr - 1) / nzr;
  float* m0 = mom + (size_t)i * M_REC;
  WaveMoments w;
  w.clear();
  float sx = 0.f;
  for (int zs = 0; zs < nzs; ++zs) {               // (record 0 = strip range 0 of row range 0 comes first: its pivots are the base)
    int cols = 0;                                    // pixels per row of this strip range
    for (int r = zs; r * ws < nstrips; r += nzs) {
      const int s_lo = r * ws, s_hi = min(nstrips, s_lo + ws);
      cols += min(W, s_hi * kS) - s_lo * kS;
    }
    if (cols == 0) continue;
    for (int zr = 0; zr < nzr; ++zr) {
      const int rows = min(per, H - zr * per);
      if (rows <= 0) continue;
      const float* mz = m0 + (size_t)(zr * nzs + zs) * B * C * M_REC;
      float a[M_N];
#pragma unroll
      for (int k = 0; k < M_N; ++k) a[k] = mz[k];
      sx += a[M_SX];
      w.merge(a, mz[M_PV], mz[M_PO], (float)(cols * rows));
    }
  }
  w.s[M_SX] = sx;
#pragma unroll
  for (int k = 0; k < M_N; ++k) m0[k] = w.s[k];
  m0[M_PV] = w.pv;
  m0[M_PO] = w.po;
}

// ------------------------------------------------------------------------------------------------
// backward statistics: per (image, channel) sums of dOut, dOut*(V - pV), dOut*(o - pO) -- about the pivots the forward
// statistics pass recorded for the plane (mom[.., M_PV / M_PO]; mom == null: raw sums).  The per-channel kernels put
// pV * sum dOut back in double: accumulated raw in fp32, sum dOut*V loses eps * |mean V| / sigma_V of what survives
// the cancellations of the BatchNorm backward (percent-level parameter gradients at |mean| / sigma = 10^3).
// ------------------------------------------------------------------------------------------------
template <typename T> constexpr int stats_bwd_wave_bytes() { return RowIO<T, kS + 2>::kBytes + 2 * RowIO<T, kS>::kBytes; }

template <typename T, bool GELU, bool HAS_O, int AUX, bool CUT>
__global__ __launch_bounds__(kMaxStrips * kWave) void light_stats_bwd_wide(
    const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv,
    const float* __restrict__ mom, float* __restrict__ bmom, int B, int C, int H, int W, int BG, int wc, int nzr) {
  const RowCut<CUT> rc(H, nzr, 0);
  MRLA_WIDE_PROLOGUE_Z(D_N, stats_bwd_wave_bytes<T>(), rc.zs(), rc.nzs())
  T* bufX = reinterpret_cast<T*>(wbuf);
  T* bufG = reinterpret_cast<T*>(wbuf + RowIO<T, kS + 2>::kBytes);
  T* bufO = reinterpret_cast<T*>(wbuf + RowIO<T, kS + 2>::kBytes + RowIO<T, kS>::kBytes);
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[c * 9 + k];
  const int b_end = min(B, (int)(blockIdx.y + 1) * BG);
  for (int b = blockIdx.y * BG; b < b_end; ++b) {
    const size_t ioff = ((size_t)b * H + rc.r0()) * rowelems;
    const T* xi = x + ioff;
    const T* gi = dout + ioff;
    const T* oi = HAS_O ? o + ioff : nullptr;
    float acc[D_N] = {0.f, 0.f, 0.f};
    const float pV = mom ? mom[((size_t)b * C + c) * M_REC + M_PV] : 0.f;
    const float pO = (mom && HAS_O) ? mom[((size_t)b * C + c) * M_REC + M_PO] : 0.f;
    for (int s = sfirst; s < nstrips; s += sstep) {
      const int s0 = s * kS;
      RowIO<T, kS + 2> ax;
      RowIO<T, kS> ag;
      make_row_io<T, kS + 2>(ax, s0 - 1, kS + 2, W, C, cbase, lane);
      make_row_io<T, kS>(ag, s0, kS, W, C, cbase, lane);
      RawRow<kS + 2> xa, xb, xc;                     // x rows r-1, r, r+1 on columns s0-1 .. s0+kS
      RawRow<kS> gv, ov;
      xa.clear(); xb.clear(); xc.clear(); gv.clear(); ov.clear();
      if (CUT && rc.above()) {                       // the row above the range belongs to the image
        row_fetch_in<T, kS + 2, AUX>(ax, xi, -1, rc.lo(), rc.hi(), rowelems, bufX);
        rows_landed();
        row_read<T, kS + 2>(bufX, lane, xa);
      }
      row_fetch_in<T, kS + 2, AUX>(ax, xi, 0, rc.lo(), rc.hi(), rowelems, bufX);
      rows_landed();
      row_read<T, kS + 2>(bufX, lane, xb);
      row_fetch_in<T, kS + 2, AUX>(ax, xi, 1, rc.lo(), rc.hi(), rowelems, bufX);
      row_fetch_in<T, kS, AUX>(ag, gi, 0, rc.lo(), rc.hi(), rowelems, bufG);
      if (HAS_O) row_fetch_in<T, kS, AUX>(ag, oi, 0, rc.lo(), rc.hi(), rowelems, bufO);
      // dOut (and o) are zero beyond the image, so columns of a ragged last strip drop out of every sum by themselves
      auto step = [&](int r, RawRow<kS + 2>& XA, RawRow<kS + 2>& XB, RawRow<kS + 2>& XC) {
        rows_landed();
        row_read_issue<T, kS + 2>(bufX, lane, XC);
        row_read_issue<T, kS>(bufG, lane, gv);
        if (HAS_O) row_read_issue<T, kS>(bufO, lane, ov);
        row_read_fence(XC, true);
        row_read_fence(gv, false);
        if (HAS_O) row_read_fence(ov, false);
        row_fetch_in<T, kS + 2, AUX>(ax, xi, r + 2, rc.lo(), rc.hi(), rowelems, bufX);
        row_fetch_in<T, kS, AUX>(ag, gi, r + 1, rc.lo(), rc.hi(), rowelems, bufG);
        if (HAS_O) row_fetch_in<T, kS, AUX>(ag, oi, r + 1, rc.lo(), rc.hi(), rowelems, bufO);
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          float v = conv_at(w, XA.v, XB.v, XC.v, j);
          if (GELU) v = gelu_f(v);
          acc[D_D] += gv.v[j];
          acc[D_DV] = fmaf(gv.v[j], v - pV, acc[D_DV]);
          if (HAS_O) acc[D_DO] = fmaf(gv.v[j], ov.v[j] - pO, acc[D_DO]);
        }
      };
      MRLA_ROTATE3(rc.n(), step, xa, xb, xc)
      rows_landed();                                 // the look-ahead rows of the last step (zeros) are still in flight
    }
    wg_reduce<D_N>(acc, red, lane, wave, nwaves, wc);
    if (wave < wc) {
#pragma unroll
      for (int k = 0; k < D_N; ++k) bmom[(((size_t)blockIdx.z * B + b) * C + c) * D_N + k] = acc[k];      // (range z's record)
    }
  }
}


// ------------------------------------------------------------------------------------------------
// forward statistics, x given: per (image, channel) sums of x, V, o, V^2, V*o, o^2 (V = act(dwconv3x3(x)));
// MRLA-base also keeps V (vout = the stage's value-history slot)
// ------------------------------------------------------------------------------------------------
template <typename T> constexpr int stats_fwd_wave_bytes() { return RowIO<T, kS + 2>::kBytes + 2 * RowIO<T, kS>::kBytes; }

template <typename T, bool GELU, bool HAS_O, bool RAGGED, bool CUT>
__global__ __launch_bounds__(kMaxStrips * kWave) void light_stats_fwd_wide(
    const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv, float* __restrict__ mom,
    T* __restrict__ vout, int B, int C, int H, int W, int BG, int wc, int nzr) {
  const RowCut<CUT> rc(H, nzr, 0);
  MRLA_WIDE_PROLOGUE_Z(kMomRed, stats_fwd_wave_bytes<T>(), rc.zs(), rc.nzs())
  T* bufX = reinterpret_cast<T*>(wbuf);
  T* bufO = reinterpret_cast<T*>(wbuf + RowIO<T, kS + 2>::kBytes);
  T* bufS = reinterpret_cast<T*>(wbuf + RowIO<T, kS + 2>::kBytes + RowIO<T, kS>::kBytes);
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[c * 9 + k];
  const int b_end = min(B, (int)(blockIdx.y + 1) * BG);
  for (int b = blockIdx.y * BG; b < b_end; ++b) {
    const size_t ioff = ((size_t)b * H + rc.r0()) * rowelems;
    const T* xi = x + ioff;
    const T* oi = HAS_O ? o + ioff : nullptr;
    T* vo = vout ? vout + ioff : nullptr;
    WaveMoments wm;
    wm.clear();
    for (int s = sfirst; s < nstrips; s += sstep) {
      float acc[M_N] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float pV = 0.f, pO = 0.f;                       // this strip's pivots: its first V and o
      const int s0 = s * kS, nc = min(kS, W - s0);
      RowIO<T, kS + 2> ax;
      RowIO<T, kS> ao, as;
      make_row_io<T, kS + 2>(ax, s0 - 1, kS + 2, W, C, cbase, lane);
      make_row_io<T, kS>(ao, s0, kS, W, C, cbase, lane);
      make_row_io<T, kS>(as, s0, nc, W, C, cbase, lane);
      RawRow<kS + 2> xa, xb, xc;
      RawRow<kS> ov;
      xa.clear(); xb.clear(); xc.clear(); ov.clear();
      if (CUT && rc.above()) {                       // the row above the range belongs to the image
        row_fetch_in<T, kS + 2>(ax, xi, -1, rc.lo(), rc.hi(), rowelems, bufX);
        rows_landed();
        row_read<T, kS + 2>(bufX, lane, xa);
      }
      row_fetch_in<T, kS + 2>(ax, xi, 0, rc.lo(), rc.hi(), rowelems, bufX);
      rows_landed();
      row_read<T, kS + 2>(bufX, lane, xb);
      row_fetch_in<T, kS + 2>(ax, xi, 1, rc.lo(), rc.hi(), rowelems, bufX);
      if (HAS_O) row_fetch_in<T, kS>(ao, oi, 0, rc.lo(), rc.hi(), rowelems, bufO);
      auto step = [&](int r, RawRow<kS + 2>& XA, RawRow<kS + 2>& XB, RawRow<kS + 2>& XC) {
        rows_landed();
        row_read_issue<T, kS + 2>(bufX, lane, XC);
        if (HAS_O) row_read_issue<T, kS>(bufO, lane, ov);
        row_read_fence(XC, true);
        if (HAS_O) row_read_fence(ov, false);
        row_fetch_in<T, kS + 2>(ax, xi, r + 2, rc.lo(), rc.hi(), rowelems, bufX);
        if (HAS_O) row_fetch_in<T, kS>(ao, oi, r + 1, rc.lo(), rc.hi(), rowelems, bufO);
        float vrow[kS];
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          float v = conv_at(w, XA.v, XB.v, XC.v, j);
          if (GELU) v = gelu_f(v);
          vrow[j] = v;
        }
        if (r == 0) { pV = vrow[0]; pO = HAS_O ? ov.v[0] : 0.f; }
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          float dv = vrow[j] - pV, d_o = HAS_O ? ov.v[j] - pO : 0.f;
          if (RAGGED) { dv = j < nc ? dv : 0.f; d_o = j < nc ? d_o : 0.f; }     // columns beyond the image do not count
          acc[M_SX] += XB.v[j + 1];
          acc[M_SV] += dv;
          acc[M_SVV] = fmaf(dv, dv, acc[M_SVV]);
          if (HAS_O) {
            acc[M_SO] += d_o;
            acc[M_SVO] = fmaf(dv, d_o, acc[M_SVO]);
            acc[M_SOO] = fmaf(d_o, d_o, acc[M_SOO]);
          }
        }
        if (vo) row_store<T, kS>(as, vo, r, rowelems, lane, bufS, vrow);
      };
      MRLA_ROTATE3(rc.n(), step, xa, xb, xc)
      rows_landed();
      wm.merge(acc, pV, pO, (float)(rc.n() * nc));
    }
    store_moments(wm, red, mom + (((size_t)blockIdx.z * B + b) * C + c) * M_REC, lane, wave, nwaves, wc);      // (range z's record)
  }
}

// ------------------------------------------------------------------------------------------------
// The fused producer: x_t = relu(round(round(psc*pre + psh) + o)) formed from conv3's output and the shortcut
// (resnet_mrla_light.py:101-114; AFF = bn3's affine handed over instead of applied in a pass of its own).
// `pre` and `o` arrive as zeros outside the image; the shift is masked there so that the padding of the 3x3 stays zero.
// ------------------------------------------------------------------------------------------------
// STORE_X = false (round 6): x_t is formed for the statistics and NOT written -- the training tail that re-forms it from `pre`
// and `o` in every later pass (apply_fwd_pre, light_nhwc_lean.hip) moves 2N here instead of 3N.
template <typename T, bool AFF, bool RAGGED, int AUX, bool STORE_X, bool CUT>
__device__ __forceinline__ void light_stats_fwd_fused_body(
    const T* __restrict__ pre, const T* __restrict__ o, const float* __restrict__ wv, float* __restrict__ mom,
    T* __restrict__ xout, const float* __restrict__ psc, const float* __restrict__ psh, T* __restrict__ vout, int B,
    int C, int H, int W, int BG, int wc, int nzr) {
  const RowCut<CUT> rc(H, nzr, 0);
  MRLA_WIDE_PROLOGUE_Z(kMomRed, fused_wave_bytes<T>(), rc.zs(), rc.nzs())
  constexpr int RB = RowIO<T, kS + 2>::kBytes;
  T* bufP = reinterpret_cast<T*>(wbuf);
  unsigned char* bufO2 = wbuf + RB;                  // o row r lives in half (r & 1)
  T* bufS = reinterpret_cast<T*>(wbuf + 3 * RB);
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[c * 9 + k];
  const float asc = AFF ? psc[c] : 1.f, ash = AFF ? psh[c] : 0.f;
  const int b_end = min(B, (int)(blockIdx.y + 1) * BG);
  for (int b = blockIdx.y * BG; b < b_end; ++b) {
    const size_t ioff = ((size_t)b * H + rc.r0()) * rowelems;
    const T* pi = pre + ioff;
    const T* oi = o + ioff;
    T* xo = STORE_X ? xout + ioff : nullptr;
    T* vo = vout ? vout + ioff : nullptr;
    WaveMoments wm;
    wm.clear();
    for (int s = sfirst; s < nstrips; s += sstep) {
      float acc[M_N] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float pV = 0.f, pO = 0.f;                       // this strip's pivots: its first V and o
      const int s0 = s * kS, nc = min(kS, W - s0);
      RowIO<T, kS + 2> ax;
      RowIO<T, kS> as;
      make_row_io<T, kS + 2>(ax, s0 - 1, kS + 2, W, C, cbase, lane);
      make_row_io<T, kS>(as, s0, nc, W, C, cbase, lane);
      ColumnShifts<RAGGED> shj;
      shj.set(ash, s0, W);
      RawRow<kS + 2> praw, oraw;
      RawRow<kS> ov;
      praw.clear(); oraw.clear(); ov.clear();
      float xa[kS + 2], xb[kS + 2], xc[kS + 2];      // x_t rows r-1, r, r+1 on columns s0-1 .. s0+kS
      auto obuf = [&](int r) { return reinterpret_cast<T*>(bufO2 + (r & 1) * RB); };
      auto store_owned = [&](T* dst, int r, const float (&row)[kS + 2]) {
        float own[kS];
#pragma unroll
        for (int j = 0; j < kS; ++j) own[j] = row[j + 1];
        row_store<T, kS>(as, dst, r, rowelems, lane, bufS, own);
      };
      if (CUT && rc.above()) {                       // x_t of the row above the range (it belongs to the image)
        row_fetch_in<T, kS + 2, AUX>(ax, pi, -1, rc.lo(), rc.hi(), rowelems, bufP);
        row_fetch_in<T, kS + 2, AUX>(ax, oi, -1, rc.lo(), rc.hi(), rowelems, obuf(0));
        rows_landed();
        row_read<T, kS + 2>(bufP, lane, praw);
        row_read<T, kS + 2>(obuf(0), lane, oraw);
        form_x_row<T, AFF, RAGGED>(praw, oraw, asc, shj, xa);
      } else {
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) xa[j] = 0.f;
      }
      row_fetch_in<T, kS + 2, AUX>(ax, pi, 0, rc.lo(), rc.hi(), rowelems, bufP);
      row_fetch_in<T, kS + 2, AUX>(ax, oi, 0, rc.lo(), rc.hi(), rowelems, obuf(0));
      rows_landed();
      row_read<T, kS + 2>(bufP, lane, praw);
      row_read<T, kS + 2>(obuf(0), lane, oraw);
      row_fetch_in<T, kS + 2, AUX>(ax, pi, 1, rc.lo(), rc.hi(), rowelems, bufP);
      row_fetch_in<T, kS + 2, AUX>(ax, oi, 1, rc.lo(), rc.hi(), rowelems, obuf(1));
      form_x_row<T, AFF, RAGGED>(praw, oraw, asc, shj, xb);
      if (STORE_X) store_owned(xo, 0, xb);
      auto step = [&](int r, float (&XA)[kS + 2], float (&XB)[kS + 2], float (&XC)[kS + 2]) {
        // step r-1 stored x_t row r (STORE_X; always: r <= H-1) and, for MRLA-base, V row r-1 after its fetches
        if (r == 0) rows_landed();
        else if (vo) rows_landed_keep<(STORE_X ? 2 : 1) * RowIO<T, kS>::NL>();
        else rows_landed_keep<(STORE_X ? 1 : 0) * RowIO<T, kS>::NL>();
        row_read_issue<T, kS + 2>(bufP, lane, praw);         // row r+1
        row_read_issue<T, kS + 2>(obuf(r + 1), lane, oraw);
        row_read_issue<T, kS>(obuf(r), lane, ov, 1);         // owned pixels of row r
        row_read_fence(praw, true);
        row_read_fence(oraw, false);
        row_read_fence(ov, false);
        row_fetch_in<T, kS + 2, AUX>(ax, pi, r + 2, rc.lo(), rc.hi(), rowelems, bufP);
        row_fetch_in<T, kS + 2, AUX>(ax, oi, r + 2, rc.lo(), rc.hi(), rowelems, obuf(r));
        if (r + 1 < rc.hi()) {
          form_x_row<T, AFF, RAGGED>(praw, oraw, asc, shj, XC);
          if (STORE_X && (!CUT || r + 1 < rc.n())) store_owned(xo, r + 1, XC);      // (the row below the range: its owner's)
        } else {
#pragma unroll
          for (int j = 0; j < kS + 2; ++j) XC[j] = 0.f;
        }
        float vrow[kS];
#pragma unroll
        for (int j = 0; j < kS; ++j) vrow[j] = conv_at(w, XA, XB, XC, j);
        if (r == 0) { pV = vrow[0]; pO = ov.v[0]; }
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          float dv = vrow[j] - pV, d_o = ov.v[j] - pO;
          if (RAGGED) { dv = j < nc ? dv : 0.f; d_o = j < nc ? d_o : 0.f; }
          acc[M_SX] += XB[j + 1];
          acc[M_SV] += dv;
          acc[M_SVV] = fmaf(dv, dv, acc[M_SVV]);
          acc[M_SO] += d_o;
          acc[M_SVO] = fmaf(dv, d_o, acc[M_SVO]);
          acc[M_SOO] = fmaf(d_o, d_o, acc[M_SOO]);
        }
        if (vo) row_store<T, kS>(as, vo, r, rowelems, lane, bufS, vrow);
      };
      MRLA_ROTATE3(rc.n(), step, xa, xb, xc)
      rows_landed();
      wm.merge(acc, pV, pO, (float)(rc.n() * nc));
    }
    store_moments(wm, red, mom + (((size_t)blockIdx.z * B + b) * C + c) * M_REC, lane, wave, nwaves, wc);      // (range z's record)
  }
}

template <typename T, bool AFF, bool RAGGED, int AUX, bool STORE_X, bool CUT>
__global__ __launch_bounds__(kMaxStrips * kWave) void light_stats_fwd_fused_wide(
    const T* __restrict__ pre, const T* __restrict__ o, const float* __restrict__ wv, float* __restrict__ mom,
    T* __restrict__ xout, const float* __restrict__ psc, const float* __restrict__ psh, T* __restrict__ vout, int B,
    int C, int H, int W, int BG, int wc, int nzr) {
  light_stats_fwd_fused_body<T, AFF, RAGGED, AUX, STORE_X, CUT>(pre, o, wv, mom, xout, psc, psh, vout, B, C, H, W, BG, wc, nzr);
}

// ------------------------------------------------------------------------------------------------
// forward apply:  out = res*x + A*V + B*o + C   (A, B, C per (image, channel); the gate, the BatchNorm scale and the
// residual are folded into the nine taps)
// ------------------------------------------------------------------------------------------------
template <typename T> constexpr int apply_fwd_wave_bytes() { return RowIO<T, kS + 2>::kBytes + 2 * RowIO<T, kS>::kBytes; }

template <typename T, bool GELU, bool HAS_O, int AUX, bool CUT>
__global__ __launch_bounds__(kMaxStrips * kWave) void light_apply_fwd_wide(
    const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv, const float* __restrict__ gate,
    const float* __restrict__ sc, const float* __restrict__ sh, const float* __restrict__ lam,
    const float* __restrict__ dp, T* __restrict__ out, int B, int C, int H, int W, int BG, int d, int res, int wc, int nzr) {
  const RowCut<CUT> rc(H, nzr, 0);
  MRLA_WIDE_PROLOGUE_Z(0, apply_fwd_wave_bytes<T>(), rc.zs(), rc.nzs())
  T* bufX = reinterpret_cast<T*>(wbuf);
  T* bufO = reinterpret_cast<T*>(wbuf + RowIO<T, kS + 2>::kBytes);
  T* bufS = reinterpret_cast<T*>(wbuf + RowIO<T, kS + 2>::kBytes + RowIO<T, kS>::kBytes);
  const int G = C / d;
  float w0[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w0[k] = wv[c * 9 + k];
  const float scc = sc ? sc[c] : 1.f, shc = sh ? sh[c] : 0.f, lmc = (HAS_O && lam) ? lam[c] : 0.f;
  const float resf = res ? 1.f : 0.f;
  const int yy = MRLA_REVERSE_APPLY ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;   // (see the note at the top)
  const int b_end = min(B, (yy + 1) * BG);
  for (int b = yy * BG; b < b_end; ++b) {
    const size_t ioff = ((size_t)b * H + rc.r0()) * rowelems;
    const T* xi = x + ioff;
    const T* oi = HAS_O ? o + ioff : nullptr;
    T* yo = out + ioff;
    const float dpb = dp ? dp[b] : 1.f;
    const float scale = dpb * scc;
    const float A = scale * gate[(size_t)b * G + c / d];
    const float Bc = scale * lmc, Cc = dpb * shc;
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = GELU ? w0[k] : w0[k] * A;
    if (!GELU) w[4] += resf;
    for (int s = sfirst; s < nstrips; s += sstep) {
      const int s0 = s * kS, nc = min(kS, W - s0);
      RowIO<T, kS + 2> ax;
      RowIO<T, kS> ao, as;
      make_row_io<T, kS + 2>(ax, s0 - 1, kS + 2, W, C, cbase, lane);
      make_row_io<T, kS>(ao, s0, kS, W, C, cbase, lane);
      make_row_io<T, kS>(as, s0, nc, W, C, cbase, lane);
      RawRow<kS + 2> xa, xb, xc;
      RawRow<kS> ov;
      xa.clear(); xb.clear(); xc.clear(); ov.clear();
      if (CUT && rc.above()) {                       // the row above the range belongs to the image
        row_fetch_in<T, kS + 2, AUX>(ax, xi, -1, rc.lo(), rc.hi(), rowelems, bufX);
        rows_landed();
        row_read<T, kS + 2>(bufX, lane, xa);
      }
      row_fetch_in<T, kS + 2, AUX>(ax, xi, 0, rc.lo(), rc.hi(), rowelems, bufX);
      rows_landed();
      row_read<T, kS + 2>(bufX, lane, xb);
      row_fetch_in<T, kS + 2, AUX>(ax, xi, 1, rc.lo(), rc.hi(), rowelems, bufX);
      if (HAS_O) row_fetch_in<T, kS, AUX>(ao, oi, 0, rc.lo(), rc.hi(), rowelems, bufO);
      auto step = [&](int r, RawRow<kS + 2>& XA, RawRow<kS + 2>& XB, RawRow<kS + 2>& XC) {
        // the previous step's output row (its newest memory instructions) may stay in flight
        if (r == 0) rows_landed(); else rows_landed_keep<RowIO<T, kS>::NL>();
        row_read_issue<T, kS + 2>(bufX, lane, XC);
        if (HAS_O) row_read_issue<T, kS>(bufO, lane, ov);
        row_read_fence(XC, true);
        if (HAS_O) row_read_fence(ov, false);
        row_fetch_in<T, kS + 2, AUX>(ax, xi, r + 2, rc.lo(), rc.hi(), rowelems, bufX);
        if (HAS_O) row_fetch_in<T, kS, AUX>(ao, oi, r + 1, rc.lo(), rc.hi(), rowelems, bufO);
        float y[kS];
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          if (GELU) y[j] = fmaf(A, gelu_f(conv_at(w, XA.v, XB.v, XC.v, j)), fmaf(resf, XB.v[j + 1], Cc));
          else      y[j] = conv_at(w, XA.v, XB.v, XC.v, j) + Cc;
          if (HAS_O) y[j] = fmaf(Bc, ov.v[j], y[j]);
        }
        row_store<T, kS>(as, yo, r, rowelems, lane, bufS, y);
      };
      MRLA_ROTATE3(rc.n(), step, xa, xb, xc)
      rows_landed();
    }
  }
}

// Inference form of the apply pass: x_t is re-formed from conv3's output and the shortcut while it is convolved (it is
// neither needed again nor saved when nothing is differentiated): the block tail moves 5N instead of 6N elements.
template <typename T, bool AFF, bool CUT>
__global__ __launch_bounds__(kMaxStrips * kWave) void light_apply_fwd_pre_wide(
    const T* __restrict__ pre, const T* __restrict__ o, const float* __restrict__ psc, const float* __restrict__ psh,
    const float* __restrict__ wv, const float* __restrict__ gate, const float* __restrict__ sc,
    const float* __restrict__ sh, const float* __restrict__ lam, const float* __restrict__ dp, T* __restrict__ out, int B,
    int C, int H, int W, int BG, int d, int res, int wc, int nzr) {
  const RowCut<CUT> rc(H, nzr, 0);
  MRLA_WIDE_PROLOGUE_Z(0, fused_wave_bytes<T>(), rc.zs(), rc.nzs())
  constexpr int RB = RowIO<T, kS + 2>::kBytes;
  T* bufP = reinterpret_cast<T*>(wbuf);
  unsigned char* bufO2 = wbuf + RB;
  T* bufS = reinterpret_cast<T*>(wbuf + 3 * RB);
  const int G = C / d;
  float w0[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w0[k] = wv[c * 9 + k];
  const float scc = sc ? sc[c] : 1.f, shc = sh ? sh[c] : 0.f, lmc = lam ? lam[c] : 0.f;
  const float resf = res ? 1.f : 0.f;
  const float asc = AFF ? psc[c] : 1.f, ash = AFF ? psh[c] : 0.f;
  const int b_end = min(B, (int)(blockIdx.y + 1) * BG);
  for (int b = blockIdx.y * BG; b < b_end; ++b) {
    const size_t ioff = ((size_t)b * H + rc.r0()) * rowelems;
    const T* pi = pre + ioff;
    const T* oi = o + ioff;
    T* yo = out + ioff;
    const float dpb = dp ? dp[b] : 1.f;
    const float scale = dpb * scc;
    const float A = scale * gate[(size_t)b * G + c / d];
    const float Bc = scale * lmc, Cc = dpb * shc;
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = w0[k] * A;
    w[4] += resf;
    for (int s = sfirst; s < nstrips; s += sstep) {
      const int s0 = s * kS, nc = min(kS, W - s0);
      RowIO<T, kS + 2> ax;
      RowIO<T, kS> as;
      make_row_io<T, kS + 2>(ax, s0 - 1, kS + 2, W, C, cbase, lane);
      make_row_io<T, kS>(as, s0, nc, W, C, cbase, lane);
      ColumnShifts<true> shj;
      shj.set(ash, s0, W);
      RawRow<kS + 2> praw, oraw;
      RawRow<kS> ov;
      praw.clear(); oraw.clear(); ov.clear();
      float xa[kS + 2], xb[kS + 2], xc[kS + 2];
      auto obuf = [&](int r) { return reinterpret_cast<T*>(bufO2 + (r & 1) * RB); };
      if (CUT && rc.above()) {                       // x_t of the row above the range (it belongs to the image)
        row_fetch_in<T, kS + 2>(ax, pi, -1, rc.lo(), rc.hi(), rowelems, bufP);
        row_fetch_in<T, kS + 2>(ax, oi, -1, rc.lo(), rc.hi(), rowelems, obuf(0));
        rows_landed();
        row_read<T, kS + 2>(bufP, lane, praw);
        row_read<T, kS + 2>(obuf(0), lane, oraw);
        form_x_row<T, AFF, true>(praw, oraw, asc, shj, xa);
      } else {
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) xa[j] = 0.f;
      }
      row_fetch_in<T, kS + 2>(ax, pi, 0, rc.lo(), rc.hi(), rowelems, bufP);
      row_fetch_in<T, kS + 2>(ax, oi, 0, rc.lo(), rc.hi(), rowelems, obuf(0));
      rows_landed();
      row_read<T, kS + 2>(bufP, lane, praw);
      row_read<T, kS + 2>(obuf(0), lane, oraw);
      row_fetch_in<T, kS + 2>(ax, pi, 1, rc.lo(), rc.hi(), rowelems, bufP);
      row_fetch_in<T, kS + 2>(ax, oi, 1, rc.lo(), rc.hi(), rowelems, obuf(1));
      form_x_row<T, AFF, true>(praw, oraw, asc, shj, xb);
      auto step = [&](int r, float (&XA)[kS + 2], float (&XB)[kS + 2], float (&XC)[kS + 2]) {
        if (r == 0) rows_landed(); else rows_landed_keep<RowIO<T, kS>::NL>();
        row_read_issue<T, kS + 2>(bufP, lane, praw);
        row_read_issue<T, kS + 2>(obuf(r + 1), lane, oraw);
        row_read_issue<T, kS>(obuf(r), lane, ov, 1);
        row_read_fence(praw, true);
        row_read_fence(oraw, false);
        row_read_fence(ov, false);
        row_fetch_in<T, kS + 2>(ax, pi, r + 2, rc.lo(), rc.hi(), rowelems, bufP);
        row_fetch_in<T, kS + 2>(ax, oi, r + 2, rc.lo(), rc.hi(), rowelems, obuf(r));
        if (r + 1 < rc.hi()) {
          form_x_row<T, AFF, true>(praw, oraw, asc, shj, XC);
        } else {
#pragma unroll
          for (int j = 0; j < kS + 2; ++j) XC[j] = 0.f;
        }
        float y[kS];
#pragma unroll
        for (int j = 0; j < kS; ++j) y[j] = fmaf(Bc, ov.v[j], conv_at(w, XA, XB, XC, j) + Cc);
        row_store<T, kS>(as, yo, r, rowelems, lane, bufS, y);
      };
      MRLA_ROTATE3(rc.n(), step, xa, xb, xc)
      rows_landed();
    }
  }
}


// ------------------------------------------------------------------------------------------------
// backward apply:  dm = E*dOut + F*V + G*o + H (BatchNorm backward in closed form), dU = a*dm*act'(U),
//   dx = dwconv3x3^T(dU) + res*dOut + dy/(hw)   [RELU: masked by x > 0 -- the fused relu(pre + o) producer],
//   do = lam*dm  [RELU: + dx],   dWv partial sums per image group.
// Strip-local windows (columns relative to s0): x rows rr-1..rr+1 over cols -2..kS+1 (kS+4 wide), dU rows rr-2..rr over
// cols -1..kS (kS+2 wide).  Step rr: U[rr] on cols -1..kS -> dU[rr]; then dx[rr-1] on the owned cols from dU rows
// rr-2..rr.  dOut of row rr-1 is read a second time from LDS (two dOut row buffers) instead of being kept in registers.
// ------------------------------------------------------------------------------------------------
// PRE (with RELU only): the caller deferred the BatchNorm in front of the fused producer (bn3, resnet_mrla_light.py:101-102)
// and its backward needs sum(dpre) and sum(dpre * y3) per channel, y3 = conv3's raw output `pre`.  dpre = dx is formed
// here, so the two sums are taken here as well (one more kS-wide row fetch per step, two accumulators): the separate
// 2N statistics pass over (dpre, y3) disappears.  y3 cannot be reconstructed from x_t - o instead: bn3's scale is zero
// at initialisation (zero_init_last_bn) and tiny early in training.
template <typename T, bool GELU, bool HAS_O, bool RELU, bool RAGGED, bool PRE, bool CUT>
__global__ __launch_bounds__(kBwdWaves * kWave) void light_apply_bwd_wide(
    const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv,
    const float* __restrict__ gate, const float* __restrict__ cb, const float* __restrict__ lam,
    const float* __restrict__ dp, const float* __restrict__ dyx, T* __restrict__ dx, T* __restrict__ dprev,
    float* __restrict__ dwv_part, const T* __restrict__ pre, const float* __restrict__ pre_center,
    float* __restrict__ pre_tmom, int B, int C, int H, int W, int BG, int d, int res, int wc, int nzr) {
  static_assert(!PRE || RELU, "the deferred-BatchNorm sums belong to the fused relu(pre + o) producer");
  const RowCut<CUT> rc(H, nzr, 0);
  MRLA_WIDE_PROLOGUE_Z(9, (apply_bwd_wave_bytes<T, PRE>()), rc.zs(), rc.nzs())
  constexpr int XB_ = RowIO<T, kS + 4>::kBytes, GB = RowIO<T, kS + 2>::kBytes, SB = RowIO<T, kS>::kBytes;
  T* bufX = reinterpret_cast<T*>(wbuf);
  unsigned char* bufG2 = wbuf + XB_;                 // dOut row rr lives in half (rr & 1)
  T* bufO = reinterpret_cast<T*>(wbuf + XB_ + 2 * GB);
  T* bufS1 = reinterpret_cast<T*>(wbuf + XB_ + 3 * GB);
  T* bufS2 = reinterpret_cast<T*>(wbuf + XB_ + 3 * GB + SB);
  T* bufP = reinterpret_cast<T*>(wbuf + XB_ + 3 * GB + 2 * SB);      // (PRE) y3 row rr-1 on the owned columns
  float pm[2] = {0.f, 0.f};                          // (PRE) sum dpre, sum dpre * (y3 - center) over this workgroup's images
  const float pcen = (PRE && pre_center) ? pre_center[c] : 0.f;      // bn3's batch mean: no cancelling subtraction later
  const int G = C / d;
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[c * 9 + k];
  const float e_ = cb ? cb[c * 4 + 0] : 1.f, f_ = cb ? cb[c * 4 + 1] : 0.f;
  const float Gc = cb ? cb[c * 4 + 2] : 0.f, Hc = cb ? cb[c * 4 + 3] : 0.f;
  const float lm = (HAS_O && lam) ? lam[c] : 1.f;
  const float resf = res ? 1.f : 0.f;
  float wg[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int yy = MRLA_REVERSE_APPLY ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;   // (see the note at the top)
  const int b_end = min(B, (yy + 1) * BG);
  for (int b = yy * BG; b < b_end; ++b) {
    const size_t ioff = ((size_t)b * H + rc.r0()) * rowelems;
    const T* xi = x + ioff;
    const T* gi = dout + ioff;
    const T* oi = HAS_O ? o + ioff : nullptr;
    T* dxo = dx + ioff;
    T* doo = HAS_O ? dprev + ioff : nullptr;
    const T* pri = PRE ? pre + ioff : nullptr;
    const float dpb = dp ? dp[b] : 1.f;
    const float a = gate[(size_t)b * G + c / d];
    const float E = e_ * dpb, F = f_ * a;
    const float dy = dyx[(size_t)b * C + c];
    for (int s = sfirst; s < nstrips; s += sstep) {
      const int s0 = s * kS, nc = min(kS, W - s0);
      RowIO<T, kS + 4> ax;
      RowIO<T, kS + 2> ag;
      RowIO<T, kS> as;
      make_row_io<T, kS + 4>(ax, s0 - 2, kS + 4, W, C, cbase, lane);
      make_row_io<T, kS + 2>(ag, s0 - 1, kS + 2, W, C, cbase, lane);
      make_row_io<T, kS>(as, s0, nc, W, C, cbase, lane);
      RawRow<kS + 4> xa, xb, xc;                     // x rows rr-1, rr, rr+1
      RawRow<kS + 2> gv, ov;                         // dOut / o of row rr on columns -1 .. kS
      RawRow<kS> gp;                                 // dOut of row rr-1 on the owned columns
      RawRow<kS> pv;                                 // (PRE) y3 of row rr-1 on the owned columns
      xa.clear(); xb.clear(); xc.clear(); gv.clear(); ov.clear(); gp.clear(); pv.clear();
      float ua[kS + 2], ub[kS + 2], uc[kS + 2];      // dU rows rr-2, rr-1, rr
      float d0[kS], d1[kS], d2[kS];                  // lam*dm of rows rr-1 / rr (RELU: do = lam*dm + dx one step later)
#pragma unroll
      for (int j = 0; j < kS + 2; ++j) { ua[j] = 0.f; ub[j] = 0.f; }
#pragma unroll
      for (int j = 0; j < kS; ++j) d0[j] = 0.f;
      auto gbuf = [&](int r) { return reinterpret_cast<T*>(bufG2 + (r & 1) * GB); };
      // CUT, rows above the range: the walk starts one row early -- dU of the row above the range is re-computed (the upper
      // neighbour computes it too), its dx / do rows and its share of the sums are the neighbour's
      const int st = (CUT && rc.above()) ? -1 : 0;
      if (CUT && rc.above()) {
        row_fetch_in<T, kS + 4>(ax, xi, st - 1, rc.lo(), rc.hi(), rowelems, bufX);
        rows_landed();
        row_read<T, kS + 4>(bufX, lane, xa);
      }
      row_fetch_in<T, kS + 4>(ax, xi, st, rc.lo(), rc.hi(), rowelems, bufX);
      rows_landed();
      row_read<T, kS + 4>(bufX, lane, xb);
      row_fetch_in<T, kS + 4>(ax, xi, st + 1, rc.lo(), rc.hi(), rowelems, bufX);
      row_fetch_in<T, kS + 2>(ag, gi, st, rc.lo(), rc.hi(), rowelems, gbuf(st));
      if (HAS_O) row_fetch_in<T, kS + 2>(ag, oi, st, rc.lo(), rc.hi(), rowelems, bufO);
      auto step = [&](int rr, RawRow<kS + 4>& XA, RawRow<kS + 4>& XB, RawRow<kS + 4>& XC, float (&UA)[kS + 2],
                      float (&UB)[kS + 2], float (&UC)[kS + 2], float (&DP)[kS], float (&DC)[kS]) {
        // steps rr-1 >= 1 stored two rows (dx and do) after their fetches; those may stay in flight
        if (rr <= 1 || !HAS_O) rows_landed(); else rows_landed_keep<2 * RowIO<T, kS>::NL>();
        row_read_issue<T, kS + 4>(bufX, lane, XC);
        row_read_issue<T, kS + 2>(gbuf(rr), lane, gv);
        if (HAS_O) row_read_issue<T, kS + 2>(bufO, lane, ov);
        if (rr >= 1) row_read_issue<T, kS>(gbuf(rr + 1), lane, gp, 1);      // row rr-1, owned pixels
        if (PRE && rr >= 1) row_read_issue<T, kS>(bufP, lane, pv);
        row_read_fence(XC, true);
        row_read_fence(gv, false);
        if (HAS_O) row_read_fence(ov, false);
        row_read_fence(gp, false);
        if (PRE) row_read_fence(pv, false);
        row_fetch_in<T, kS + 4>(ax, xi, rr + 2, rc.lo(), rc.hi(), rowelems, bufX);
        row_fetch_in<T, kS + 2>(ag, gi, rr + 1, rc.lo(), rc.hi(), rowelems, gbuf(rr + 1));
        if (HAS_O) row_fetch_in<T, kS + 2>(ag, oi, rr + 1, rc.lo(), rc.hi(), rowelems, bufO);
        if (PRE) row_fetch_in<T, kS>(as, pri, rr, rc.lo(), rc.hi(), rowelems, bufP);      // for step rr+1 (pixels past the image: zeros)
        // CUT: dU of the rows just above and below the range is formed for dx of the range's first and last row only
        const bool own = !CUT || (rr >= 0 && rr < rc.n());
        float dorow[kS];
        if (rr >= rc.hi()) {       // the step past the image's last row only finishes dx[H-1]
#pragma unroll
          for (int j = 0; j < kS + 2; ++j) UC[j] = 0.f;
#pragma unroll
          for (int j = 0; j < kS; ++j) { DC[j] = 0.f; dorow[j] = 0.f; }
        } else {
#pragma unroll
          for (int j = 0; j < kS + 2; ++j) {
            const int col = s0 - 1 + j;
            // dU exists inside the image only (wave-uniform; for whole strips only the two halo columns can be outside)
            const bool in = RAGGED ? (col >= 0 && col < W) : (j == 0 ? s0 > 0 : (j == kS + 1 ? s0 + kS < W : true));
            const float u = conv_at(w, XA.v, XB.v, XC.v, j);             // window cols j..j+2 <-> image cols col-1..col+1
            const float v = GELU ? gelu_f(u) : u;
            float dm = fmaf(E, gv.v[j], Hc);
            dm = fmaf(F, v, dm);
            if (HAS_O) dm = fmaf(Gc, ov.v[j], dm);
            float du = a * dm;
            if (GELU) du *= gelu_grad_f(u);
            if (RAGGED || j == 0 || j == kS + 1) du = in ? du : 0.f;
            if (j >= 1 && j <= kS) {                                        // owned column (compile-time after unroll)
              // (kept a rounded product: left to the compiler, lam*dm is contracted into next step's `+ dx` where both sit in one
              // unrolled loop body and not across its back edge -- do[r] would depend on r % 3 and on where a row range starts)
              float ldm = lm * dm;
              asm("" : "+v"(ldm));
              DC[j - 1] = ldm;                       // (columns beyond the image are dropped by the store)
              dorow[j - 1] = DC[j - 1];
              // dWv[i][k] += dU[rr][col] * x[rr+i-1][col+k-1]
              if (!MRLA_EXP_SKIP_WG) {
              const float dw = (CUT && !own) ? 0.f : du;
              wg[0] = fmaf(dw, XA.v[j], wg[0]); wg[1] = fmaf(dw, XA.v[j + 1], wg[1]); wg[2] = fmaf(dw, XA.v[j + 2], wg[2]);
              wg[3] = fmaf(dw, XB.v[j], wg[3]); wg[4] = fmaf(dw, XB.v[j + 1], wg[4]); wg[5] = fmaf(dw, XB.v[j + 2], wg[5]);
              wg[6] = fmaf(dw, XC.v[j], wg[6]); wg[7] = fmaf(dw, XC.v[j + 1], wg[7]); wg[8] = fmaf(dw, XC.v[j + 2], wg[8]);
              }
            }
            UC[j] = du;
          }
          if (HAS_O && !RELU && own) row_store<T, kS>(as, doo, rr, rowelems, lane, bufS2, dorow);
        }
        // dx[rr-1] on the owned columns:  dx[ro][col] = sum_{i,k} w[i][k] * dU[ro-i+1][col-k+1]
        if (rr >= 1) {
          float yrow[kS], dsum[kS];
#pragma unroll
          for (int j = 0; j < kS; ++j) {
            // window index of column (col + 1 - k) in the dU arrays (which start at col -1): j + 2 - k
            float s9 = w[0] * UC[j + 2];
            s9 = fmaf(w[1], UC[j + 1], s9); s9 = fmaf(w[2], UC[j], s9);
            s9 = fmaf(w[3], UB[j + 2], s9); s9 = fmaf(w[4], UB[j + 1], s9); s9 = fmaf(w[5], UB[j], s9);
            s9 = fmaf(w[6], UA[j + 2], s9); s9 = fmaf(w[7], UA[j + 1], s9); s9 = fmaf(w[8], UA[j], s9);
            float y = fmaf(resf, gp.v[j], s9 + dy);
            if (RELU) y = (XA.v[j + 2] > 0.f) ? y : 0.f;                  // XA = x[rr-1]; owned col j <-> window j+2
            yrow[j] = y;
            dsum[j] = DP[j] + y;
            if (PRE) {             // (columns past the image have x = 0, hence y = 0: nothing to mask)
              // the sums are of dpre AS STORED (rounded to T): where the MRLA branch's contribution is below half an ulp
              // of dOut it is lost in the stored tensor, systematically -- the BatchNorm backward must see the same values
              const float yq = to_f(from_f<T>(y));
              pm[0] += yq;
              pm[1] = fmaf(yq, pv.v[j] - pcen, pm[1]);
            }
          }
          row_store<T, kS>(as, dxo, rr - 1, rowelems, lane, bufS1, yrow);
          if (RELU && HAS_O) row_store<T, kS>(as, doo, rr - 1, rowelems, lane, bufS2, dsum);
        }
      };
      // steps rr = 0 .. H; after three steps every array is back in its starting role
      int rr = st;
      for (; rr + 2 <= rc.n(); rr += 3) {
        step(rr,     xa, xb, xc, ua, ub, uc, d0, d1);
        step(rr + 1, xb, xc, xa, ub, uc, ua, d1, d2);
        step(rr + 2, xc, xa, xb, uc, ua, ub, d2, d0);
      }
      if (rr <= rc.n()) {
        step(rr, xa, xb, xc, ua, ub, uc, d0, d1);
        if (rr + 1 <= rc.n()) step(rr + 1, xb, xc, xa, ub, uc, ua, d1, d2);
      }
      rows_landed();
    }
  }
  wg_reduce<9>(wg, red, lane, wave, nwaves, wc);
  if (wave < wc) {
#pragma unroll
    for (int k = 0; k < 9; ++k) dwv_part[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * C + c) * 9 + k] = wg[k];
  }
  if (PRE) {
    wg_reduce<2>(pm, red, lane, wave, nwaves, wc);
    if (wave < wc) {
      pre_tmom[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * C + c) * 2 + 0] = pm[0];
      pre_tmom[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * C + c) * 2 + 1] = pm[1];
    }
  }
}

}  // namespace mrla
#if MRLA_APPLY_BWD_PK
#include "experiments/light_apply_bwd_pk.h"      // the packed-FP32 form of the kernel above (measured, not adopted)
#endif
namespace mrla {

// ------------------------------------------------------------------------------------------------
// MRLA-base value backward on the row pipeline (round 4; base_nhwc.hip keeps the register-staged form for C % 64 != 0):
//   dx = [x > 0 if res&2] * ((res&1) * dOut + dwconv3x3^T(dV) + dyx);  dWv partials;  PRE: bn3's backward sums
//   dx[r][col]  = sum_{i,k} w[i][k] * dV[r-i+1][col-k+1]
//   dWv[i][k]  += x[r][col] * dV[r-i+1][col-k+1]      (the same window, centred on x)
// dV rows r-1 .. r+1 on columns s0-1 .. s0+kS rotate by name; x / dOut / y3 rows of the owned columns arrive one step ahead.
// ------------------------------------------------------------------------------------------------
template <typename T, bool PRE> constexpr int base_vbwd_wave_bytes() {
  return RowIO<T, kS + 2>::kBytes + (PRE ? 4 : 3) * RowIO<T, kS>::kBytes;
}

template <typename T, bool PRE>
__global__ __launch_bounds__(kMaxStrips * kWave) void base_value_bwd_wide(
    const T* __restrict__ dout, const T* __restrict__ x, const float* __restrict__ wv, const T* __restrict__ dv,
    const float* __restrict__ dyx, T* __restrict__ dx, float* __restrict__ dwv_part, const T* __restrict__ pre,
    const float* __restrict__ pre_center, float* __restrict__ pre_tmom, int B, int C, int H, int W, int BG, int res, int wc) {
  MRLA_WIDE_PROLOGUE(9, (base_vbwd_wave_bytes<T, PRE>()))
  constexpr int UB_ = RowIO<T, kS + 2>::kBytes, SB = RowIO<T, kS>::kBytes;
  T* bufU = reinterpret_cast<T*>(wbuf);
  T* bufX = reinterpret_cast<T*>(wbuf + UB_);
  T* bufG = reinterpret_cast<T*>(wbuf + UB_ + SB);
  T* bufS = reinterpret_cast<T*>(wbuf + UB_ + 2 * SB);
  T* bufP = reinterpret_cast<T*>(wbuf + UB_ + 3 * SB);               // (PRE) y3 row on the owned columns
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[c * 9 + k];
  float wg[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float pm[2] = {0.f, 0.f};                          // (PRE) sum dpre, sum dpre * (y3 - center) over this workgroup's images
  const float pcen = (PRE && pre_center) ? pre_center[c] : 0.f;
  const float resf = (res & 1) ? 1.f : 0.f;
  const bool mask = (res & 2) != 0;
  const int b_end = min(B, (int)(blockIdx.y + 1) * BG);
  for (int b = blockIdx.y * BG; b < b_end; ++b) {
    const size_t ioff = (size_t)b * H * rowelems;
    const T* xi = x + ioff;
    const T* gi = dout + ioff;
    const T* ui = dv + ioff;
    const T* pri = PRE ? pre + ioff : nullptr;
    T* dxo = dx + ioff;
    const float dy = dyx[(size_t)b * C + c];
    for (int s = sfirst; s < nstrips; s += sstep) {
      const int s0 = s * kS, nc = min(kS, W - s0);
      RowIO<T, kS + 2> au;
      RowIO<T, kS> ag, as;
      make_row_io<T, kS + 2>(au, s0 - 1, kS + 2, W, C, cbase, lane);
      make_row_io<T, kS>(ag, s0, kS, W, C, cbase, lane);
      make_row_io<T, kS>(as, s0, nc, W, C, cbase, lane);
      RawRow<kS + 2> ua, ub, uc;                     // dV rows r-1, r, r+1 on columns s0-1 .. s0+kS
      RawRow<kS> xr, gr, pr;
      ua.clear(); ub.clear(); uc.clear(); xr.clear(); gr.clear(); pr.clear();
      row_fetch<T, kS + 2>(au, ui, 0, H, rowelems, bufU);
      rows_landed();
      row_read<T, kS + 2>(bufU, lane, ub);
      row_fetch<T, kS + 2>(au, ui, 1, H, rowelems, bufU);
      row_fetch<T, kS>(ag, xi, 0, H, rowelems, bufX);
      row_fetch<T, kS>(ag, gi, 0, H, rowelems, bufG);
      if (PRE) row_fetch<T, kS>(ag, pri, 0, H, rowelems, bufP);
      auto step = [&](int r, RawRow<kS + 2>& UA, RawRow<kS + 2>& UB, RawRow<kS + 2>& UC) {
        // step r-1 stored dx row r-1 after its fetches: that store may stay in flight
        if (r == 0) rows_landed(); else rows_landed_keep<RowIO<T, kS>::NL>();
        row_read_issue<T, kS + 2>(bufU, lane, UC);
        row_read_issue<T, kS>(bufX, lane, xr);
        row_read_issue<T, kS>(bufG, lane, gr);
        if (PRE) row_read_issue<T, kS>(bufP, lane, pr);
        row_read_fence(UC, true);
        row_read_fence(xr, false);
        row_read_fence(gr, false);
        if (PRE) row_read_fence(pr, false);
        row_fetch<T, kS + 2>(au, ui, r + 2, H, rowelems, bufU);
        row_fetch<T, kS>(ag, xi, r + 1, H, rowelems, bufX);
        row_fetch<T, kS>(ag, gi, r + 1, H, rowelems, bufG);
        if (PRE) row_fetch<T, kS>(ag, pri, r + 1, H, rowelems, bufP);
        float yrow[kS];
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          // window index of column (col + 1 - k) in the dV rows (which start at col - 1): j + 2 - k
          float s9 = w[0] * UC.v[j + 2];
          s9 = fmaf(w[1], UC.v[j + 1], s9); s9 = fmaf(w[2], UC.v[j], s9);
          s9 = fmaf(w[3], UB.v[j + 2], s9); s9 = fmaf(w[4], UB.v[j + 1], s9); s9 = fmaf(w[5], UB.v[j], s9);
          s9 = fmaf(w[6], UA.v[j + 2], s9); s9 = fmaf(w[7], UA.v[j + 1], s9); s9 = fmaf(w[8], UA.v[j], s9);
          float y = fmaf(resf, gr.v[j], s9 + dy);
          if (mask) y = (xr.v[j] > 0.f) ? y : 0.f;
          yrow[j] = y;
          // (x is zero beyond the image, so columns of a ragged last strip drop out of the sums by themselves; with the
          // mask their y is zero as well)
          if (PRE) {                                 // of dpre AS STORED (rounded to T)
            const float yq = to_f(from_f<T>(y));
            pm[0] += yq;
            pm[1] = fmaf(yq, pr.v[j] - pcen, pm[1]);
          }
          const float xv = xr.v[j];
          wg[0] = fmaf(xv, UC.v[j + 2], wg[0]); wg[1] = fmaf(xv, UC.v[j + 1], wg[1]); wg[2] = fmaf(xv, UC.v[j], wg[2]);
          wg[3] = fmaf(xv, UB.v[j + 2], wg[3]); wg[4] = fmaf(xv, UB.v[j + 1], wg[4]); wg[5] = fmaf(xv, UB.v[j], wg[5]);
          wg[6] = fmaf(xv, UA.v[j + 2], wg[6]); wg[7] = fmaf(xv, UA.v[j + 1], wg[7]); wg[8] = fmaf(xv, UA.v[j], wg[8]);
        }
        row_store<T, kS>(as, dxo, r, rowelems, lane, bufS, yrow);
      };
      MRLA_ROTATE3(H, step, ua, ub, uc)
      rows_landed();
    }
  }
  wg_reduce<9>(wg, red, lane, wave, nwaves, wc);
  if (wave < wc) {
#pragma unroll
    for (int k = 0; k < 9; ++k) dwv_part[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * C + c) * 9 + k] = wg[k];
  }
  if (PRE) {
    wg_reduce<2>(pm, red, lane, wave, nwaves, wc);
    if (wave < wc) {
      pre_tmom[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * C + c) * 2 + 0] = pm[0];
      pre_tmom[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * C + c) * 2 + 1] = pm[1];
    }
  }
}

// Images a workgroup walks through, at most 8.
// W > 0 (the dWv-producing backward kernels, 2 waves per SIMD = 2048 wave slots): about one round of resident waves, so
// that the prologue, the dWv reduction and its partial rows are amortised over several images (14x14 stage: +25 %).
// W == 0 (the lighter passes, twice the occupancy): >= 2048 workgroups; they are faster with more, shorter workgroups.
int nhwc_images_per_group(int B, int C, int W) {
  const long wgs = (long)B * ((C + kWave - 1) / kWave);
  if (W <= 0) return (int)std::max(1L, std::min(8L, wgs / 2048));
  int wc = 1, ws = std::min((W + kS - 1) / kS, kMaxStrips);
  if (C % kWave == 0) wide_shape(P_APPLY_BWD, C, W, &wc, &ws);        // (the row pipeline's workgroups)
  return (int)std::max(1L, std::min(8L, wgs * ws / 2048));
}

// mrla_tuning_row_ranges(): process-wide, read at every query and launch (set it before the buffers are sized)
static std::atomic<int> g_row_cut_mode{0};
int nhwc_row_cut_mode() { return g_row_cut_mode.load(std::memory_order_relaxed); }
int nhwc_set_row_cut_mode(int mode) { return g_row_cut_mode.exchange(mode, std::memory_order_relaxed); }

// gridDim.z of the passes that keep sums: strip ranges (wide_strip_ranges()) x row ranges (wide_row_ranges())
static ZRanges sum_pass_zranges(WidePass pass, int B, int C, int H, int W, int bg) {
  ZRanges z = {1, 1};
  if (C % kWave) return z;                             // (the row pipeline only)
  z.strips = wide_strip_ranges(pass, B, C, W, bg);
  z.rows = wide_row_ranges(wide_workgroups(pass, B, C, W, bg, z.strips), H, (long)B * C * H * W);
  return z;
}
ZRanges nhwc_wgrad_zranges(int B, int C, int H, int W) {
  return sum_pass_zranges(P_APPLY_BWD, B, C, H, W, nhwc_images_per_group(B, C, W));
}
ZRanges nhwc_mom_zranges(int B, int C, int H, int W) { return sum_pass_zranges(P_STATS_FUSED, B, C, H, W, 0); }
ZRanges nhwc_bmom_zranges(int B, int C, int H, int W) { return sum_pass_zranges(P_STATS_BWD, B, C, H, W, 0); }
int nhwc_wgrad_ranges(int B, int C, int H, int W) { const ZRanges z = nhwc_wgrad_zranges(B, C, H, W); return z.strips * z.rows; }
int nhwc_mom_ranges(int B, int C, int H, int W) { const ZRanges z = nhwc_mom_zranges(B, C, H, W); return z.strips * z.rows; }
int nhwc_bmom_ranges(int B, int C, int H, int W) { const ZRanges z = nhwc_bmom_zranges(B, C, H, W); return z.strips * z.rows; }

int launch_light_stats_fwd_wide(const void* x, const void* o, const float* wv, float* mom, void* xout, const float* psc,
                                const float* psh, void* vout, int B, int C, int H, int W, int dtype, int act,
                                hipStream_t st, bool fused_no_x, int mom_ranges) {
  const bool ragged = (W % kS) != 0;
  const int bg = 0;               // (wide_launch(): >= 2048 workgroups)
  // mom_ranges = the records the caller holds (mrla_light_mom_splits(); 1: the callers that never split): strip ranges x row
  // ranges, each with its own record in mom[z], merged below
  ZRanges zr = {1, 1};
  if (mom_ranges > 1) {
    zr = nhwc_mom_zranges(B, C, H, W);
    if (zr.strips * zr.rows != mom_ranges) return MRLA_EINVAL;
  }
  const int nz = zr.strips, nzr = zr.rows;
  auto merged = [&](int ws) {
    if (nz * nzr > 1)
      hipLaunchKernelGGL(light_mom_merge_kernel, dim3((B * C + kThreads - 1) / kThreads), dim3(kThreads), 0, st, mom, B, C, H, W,
                         ws, nz, nzr);
    return hip_status(hipGetLastError());
  };
  int ws_used = 1;
  if (xout || fused_no_x) {       // the fused producer (needs o, no activation on V); fused_no_x: x_t is not written
    if (!o || act) return MRLA_EINVAL;
#define CALL_C(KERNEL, T, AF, RG, NT, SX, CT)                                                                       \
  {                                                                                                                 \
    if (set_lds_n(KERNEL<T, AF, RG, NT, SX, CT>, L.lds) != hipSuccess) return MRLA_EHIP;                             \
    hipLaunchKernelGGL((KERNEL<T, AF, RG, NT, SX, CT>), L.grid, L.block, L.lds, st, (const T*)x, (const T*)o, wv, mom, \
                       (T*)xout, psc, psh, (T*)vout, B, C, H, W, L.BG, L.wc, nzr);                                  \
  }
#define CALL_K(KERNEL, T, AF, RG, NT, SX) { if (nzr > 1) CALL_C(KERNEL, T, AF, RG, NT, SX, true) else CALL_C(KERNEL, T, AF, RG, NT, SX, false) }
#define CALL_N(T, AF, RG, NT)                                                                                       \
  {                                                                                                                 \
    const WideLaunch L = wide_launch(P_STATS_FUSED, B, C, W, kMomRed, fused_wave_bytes<T>(), bg, false, nz, nzr);   \
    ws_used = (int)(L.block.x / kWave) / L.wc;                                                                      \
    if (xout) CALL_K(light_stats_fwd_fused_wide, T, AF, RG, NT, true)                                               \
    else CALL_K(light_stats_fwd_fused_wide, T, AF, RG, NT, false)                                                   \
  }
#define CALL_R(T, AF, RG) { if (stream_fetches(B, C, H, W, sizeof(T))) CALL_N(T, AF, RG, 2) else CALL_N(T, AF, RG, 0) }
#define CALL_A(T, AF) { if (ragged) CALL_R(T, AF, true) else CALL_R(T, AF, false) }
#define CALL(T) { if (psc) CALL_A(T, true) else CALL_A(T, false) }
    switch (dtype) {
      case MRLA_F32:  CALL(float) break;
      case MRLA_BF16: CALL(bf16_t) break;
      case MRLA_F16:  CALL(f16_t) break;
      default: return MRLA_EINVAL;
    }
#undef CALL
#undef CALL_A
#undef CALL_R
#undef CALL_N
#undef CALL_K
#undef CALL_C
    return merged(ws_used);
  }
#define CALL_C(T, A, O, RG, CT)                                                                                     \
  {                                                                                                                 \
    const WideLaunch L = wide_launch(P_STATS_FWD, B, C, W, kMomRed, stats_fwd_wave_bytes<T>(), bg, false, nz, nzr); \
    ws_used = (int)(L.block.x / kWave) / L.wc;                                                                      \
    if (set_lds_n(light_stats_fwd_wide<T, A, O, RG, CT>, L.lds) != hipSuccess) return MRLA_EHIP;                     \
    hipLaunchKernelGGL((light_stats_fwd_wide<T, A, O, RG, CT>), L.grid, L.block, L.lds, st, (const T*)x, (const T*)o, wv, \
                       mom, (T*)vout, B, C, H, W, L.BG, L.wc, nzr);                                                 \
  }
#define CALL_R(T, A, O, RG) { if (nzr > 1) CALL_C(T, A, O, RG, true) else CALL_C(T, A, O, RG, false) }
#define CALL(T, A, O) { if (ragged) CALL_R(T, A, O, true) else CALL_R(T, A, O, false) }
  MRLA_DISPATCH_T_N(dtype, act, o != nullptr, CALL)
#undef CALL
#undef CALL_R
#undef CALL_C
  return merged(ws_used);
}

int launch_light_apply_fwd_wide(const void* x, const void* o, const float* wv, const float* gate, const float* sc,
                                const float* sh, const float* lam, const float* dp, void* out, int B, int C, int H,
                                int W, int d, int res, int dtype, int act, hipStream_t st) {
  // (no sums: every strip round has its own workgroup; the rows are cut as well where that still leaves CUs idle)
  const int nzr = wide_row_ranges(wide_workgroups(P_APPLY_FWD, B, C, W, 0, wide_strip_rounds(P_APPLY_FWD, C, W)), H, (long)B * C * H * W);
#define CALL(T, A, O) { if (stream_fetches(B, C, H, W, sizeof(T))) CALL_N(T, A, O, 2) else CALL_N(T, A, O, 0) }
#define CALL_N(T, A, O, NT) { if (nzr > 1) CALL_C(T, A, O, NT, true) else CALL_C(T, A, O, NT, false) }
#define CALL_C(T, A, O, NT, CT)                                                                                    \
  {                                                                                                                \
    const WideLaunch L = wide_launch(P_APPLY_FWD, B, C, W, 0, apply_fwd_wave_bytes<T>(), 0, true, 1, nzr);         \
    if (set_lds_n(light_apply_fwd_wide<T, A, O, NT, CT>, L.lds) != hipSuccess) return MRLA_EHIP;                    \
    hipLaunchKernelGGL((light_apply_fwd_wide<T, A, O, NT, CT>), L.grid, L.block, L.lds, st, (const T*)x, (const T*)o, wv, \
                       gate, sc, sh, lam, dp, (T*)out, B, C, H, W, L.BG, d, res, L.wc, nzr);                       \
  }
  MRLA_DISPATCH_T_N(dtype, act, o != nullptr, CALL)
#undef CALL
#undef CALL_N
#undef CALL_C
  return hip_status(hipGetLastError());
}

int launch_light_apply_fwd_pre_wide(const void* pre, const void* o, const float* psc, const float* psh, const float* wv,
                                    const float* gate, const float* sc, const float* sh, const float* lam,
                                    const float* dp, void* out, int B, int C, int H, int W, int d, int res, int dtype,
                                    hipStream_t st) {
  const int nzr = wide_row_ranges(wide_workgroups(P_APPLY_FWD, B, C, W, 0, wide_strip_rounds(P_APPLY_FWD, C, W)), H, (long)B * C * H * W);
#define CALL_C(T, AF, CT)                                                                                            \
  {                                                                                                                  \
    const WideLaunch L = wide_launch(P_APPLY_FWD, B, C, W, 0, fused_wave_bytes<T>(), 0, true, 1, nzr);               \
    if (set_lds_n(light_apply_fwd_pre_wide<T, AF, CT>, L.lds) != hipSuccess) return MRLA_EHIP;                        \
    hipLaunchKernelGGL((light_apply_fwd_pre_wide<T, AF, CT>), L.grid, L.block, L.lds, st, (const T*)pre, (const T*)o, psc, \
                       psh, wv, gate, sc, sh, lam, dp, (T*)out, B, C, H, W, L.BG, d, res, L.wc, nzr);                \
  }
#define CALL_A(T, AF) { if (nzr > 1) CALL_C(T, AF, true) else CALL_C(T, AF, false) }
#define CALL(T) { if (psc) CALL_A(T, true) else CALL_A(T, false) }
  switch (dtype) {
    case MRLA_F32:  CALL(float) break;
    case MRLA_BF16: CALL(bf16_t) break;
    case MRLA_F16:  CALL(f16_t) break;
    default: return MRLA_EINVAL;
  }
#undef CALL
#undef CALL_A
#undef CALL_C
  return hip_status(hipGetLastError());
}


int launch_light_apply_bwd_wide(const void* dout, const void* x, const void* o, const float* wv, const float* gate,
                                const float* cb, const float* lam, const float* dp, const float* dyx, void* dx,
                                void* dprev, float* dwv_part, const void* pre, const float* pre_center, float* pre_tmom,
                                int B, int C, int H, int W, int d, int res, int relu, int dtype, int act, hipStream_t st) {
  const bool ragged = (W % kS) != 0;
  const int bg = nhwc_images_per_group(B, C, W);          // image groups x strip ranges = the rows mrla_light_wgrad_rows() promised
  const ZRanges zr = nhwc_wgrad_zranges(B, C, H, W);
  const int nz = zr.strips, nzr = zr.rows;
  if (pre_tmom && (!pre || !relu)) return MRLA_EINVAL;
  if (pre_tmom && dtype == MRLA_F32) return MRLA_EUNSUPPORTED;      // (LDS: see mrla_light_apply_bwd_pre_sums)
#define CALL_CUT(T, A, O, R, RG, PR, CT)                                                                             \
  {                                                                                                                  \
    const WideLaunch L = wide_launch(P_APPLY_BWD, B, C, W, 9, apply_bwd_wave_bytes<T, PR>(), bg, false, nz, nzr);    \
    if (set_lds_n(light_apply_bwd_wide<T, A, O, R, RG, PR, CT>, L.lds) != hipSuccess) return MRLA_EHIP;               \
    hipLaunchKernelGGL((light_apply_bwd_wide<T, A, O, R, RG, PR, CT>), L.grid, L.block, L.lds, st, (const T*)dout,    \
                       (const T*)x, (const T*)o, wv, gate, cb, lam, dp, dyx, (T*)dx, (T*)dprev, dwv_part,             \
                       (const T*)pre, pre_center, pre_tmom, B, C, H, W, L.BG, d, res, L.wc, nzr);                     \
  }
#define CALL_PLAIN(T, A, O, R, RG, PR) { if (nzr > 1) CALL_CUT(T, A, O, R, RG, PR, true) else CALL_CUT(T, A, O, R, RG, PR, false) }
#if MRLA_APPLY_BWD_PK
#define CALL_G(T, A, O, R, RG, PR)                                                                                   \
  {                                                                                                                  \
    if (!(A) && nzr == 1) {                                                                                          \
      constexpr int DP = sizeof(T) == 2 ? MRLA_APPLY_BWD_DEPTH : 1;                                                  \
      const WideLaunch L = wide_launch(P_APPLY_BWD, B, C, W, 0, apply_bwd_pk_wave_bytes<T, PR, DP>(), bg, false, nz); \
      if (set_lds_n(light_apply_bwd_wide_pk<T, O, R, RG, PR, DP>, L.lds) != hipSuccess) return MRLA_EHIP;             \
      hipLaunchKernelGGL((light_apply_bwd_wide_pk<T, O, R, RG, PR, DP>), L.grid, L.block, L.lds, st, (const T*)dout,  \
                         (const T*)x, (const T*)o, wv, gate, cb, lam, dp, dyx, (T*)dx, (T*)dprev, dwv_part,           \
                         (const T*)pre, pre_center, pre_tmom, B, C, H, W, L.BG, d, res, L.wc);                        \
    } else CALL_PLAIN(T, A, O, R, RG, PR)                                                                            \
  }
#else
#define CALL_G(T, A, O, R, RG, PR) CALL_PLAIN(T, A, O, R, RG, PR)
#endif
#define CALL_R(T, A, O, R) { if (ragged) CALL_G(T, A, O, R, true, false) else CALL_G(T, A, O, R, false, false) }
#define CALL_P(T)                                                                                       \
  {                                                                                                     \
    if constexpr (sizeof(T) == 2) {                                                                     \
      if (ragged) CALL_G(T, false, true, true, true, true) else CALL_G(T, false, true, true, false, true) \
    }                                                                                                   \
  }
#define CALL(T, A, O)                                                                        \
  {                                                                                          \
    if (relu) {                                                                              \
      if (!(O && !(A))) return MRLA_EINVAL;                                                  \
      if (pre_tmom) CALL_P(T) else CALL_R(T, false, true, true)                              \
    } else CALL_R(T, A, O, false)                                                            \
  }
  MRLA_DISPATCH_T_N(dtype, act, o != nullptr, CALL)
#undef CALL
#undef CALL_P
#undef CALL_R
#undef CALL_G
#undef CALL_PLAIN
#undef CALL_CUT
  return hip_status(hipGetLastError());
}

int launch_light_stats_bwd_wide(const void* dout, const void* x, const void* o, const float* wv, const float* mom,
                                float* bmom, int B, int C, int H, int W, int dtype, int act, hipStream_t st) {
#define CALL(T, A, O) CALL_N(T, A, O, 0)      /* default policy: apply_bwd re-reads these tensors right after */
  const ZRanges zr = nhwc_bmom_zranges(B, C, H, W);
#define CALL_N(T, A, O, NT) { if (zr.rows > 1) CALL_C(T, A, O, NT, true) else CALL_C(T, A, O, NT, false) }
#define CALL_C(T, A, O, NT, CT)                                                                                    \
  {                                                                                                                \
    const WideLaunch L = wide_launch(P_STATS_BWD, B, C, W, D_N, stats_bwd_wave_bytes<T>(), 0, false, zr.strips, zr.rows); \
    if (set_lds_n(light_stats_bwd_wide<T, A, O, NT, CT>, L.lds) != hipSuccess) return MRLA_EHIP;                    \
    hipLaunchKernelGGL((light_stats_bwd_wide<T, A, O, NT, CT>), L.grid, L.block, L.lds, st, (const T*)dout, (const T*)x, \
                       (const T*)o, wv, mom, bmom, B, C, H, W, L.BG, L.wc, zr.rows);                               \
  }
  MRLA_DISPATCH_T_N(dtype, act, o != nullptr, CALL)
#undef CALL
#undef CALL_N
#undef CALL_C
  if (hipGetLastError() != hipSuccess) return MRLA_EHIP;
  // the ranges' partial records -> record 0, the one mrla_light_bn_bwd / mrla_light_gate_bwd read
  return launch_fold_rows(bmom, zr.strips * zr.rows, B * C * D_N, st);
}

__global__ __launch_bounds__(kThreads) void zero_rows_kernel(float* __restrict__ a, size_t na, float* __restrict__ b, size_t nb) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < na) a[i] = 0.f;
  else if (i - na < nb) b[i - na] = 0.f;
}

int launch_base_value_bwd_wide(const void* dout, const void* x, const float* wv, const void* dv, const float* dyx, void* dx,
                               float* dwv_part, const void* pre, const float* pre_center, float* pre_tmom, int B, int C,
                               int H, int W, int res, int dtype, hipStream_t st) {
  if (C % kWave) return MRLA_EUNSUPPORTED;
  if (pre_tmom && (!pre || !(res & 2))) return MRLA_EINVAL;
  const int bg = nhwc_images_per_group(B, C, W);
  const ZRanges zr = nhwc_wgrad_zranges(B, C, H, W);
  const int nz = zr.strips;
  if (zr.rows > 1) {      // this kernel walks whole images: the partial rows of the other row ranges (mrla_light_wgrad_rows) are zero
    // (a kernel, not hipMemsetAsync: inside a captured HIP graph the memset node left these rows unwritten on ROCm 7.2 --
    // tests/test_graph_replay_gpu.py, resnet101_mrlab at batch 32)
    const size_t groups = (size_t)((B + bg - 1) / bg), used = groups * nz, all = used * zr.rows;
    const size_t n1 = (all - used) * C * 9, n2 = pre_tmom ? (all - used) * C * 2 : 0;
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((n1 + n2 + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                       dwv_part + used * C * 9, n1, pre_tmom ? pre_tmom + used * C * 2 : nullptr, n2);
  }
#define CALL_P(T, PR)                                                                                               \
  {                                                                                                                 \
    const WideLaunch L = wide_launch(P_APPLY_BWD, B, C, W, 9, base_vbwd_wave_bytes<T, PR>(), bg, false, nz);                             \
    if (set_lds_n(base_value_bwd_wide<T, PR>, L.lds) != hipSuccess) return MRLA_EHIP;                                \
    hipLaunchKernelGGL((base_value_bwd_wide<T, PR>), L.grid, L.block, L.lds, st, (const T*)dout, (const T*)x, wv,    \
                       (const T*)dv, dyx, (T*)dx, dwv_part, (const T*)pre, pre_center, pre_tmom, B, C, H, W, L.BG, res, L.wc); \
  }
#define CALL(T) { if (pre_tmom) CALL_P(T, true) else CALL_P(T, false) }
  switch (dtype) {
    case MRLA_F32:  CALL(float) break;
    case MRLA_BF16: CALL(bf16_t) break;
    case MRLA_F16:  CALL(f16_t) break;
    default: return MRLA_EINVAL;
  }
#undef CALL
#undef CALL_P
  return hip_status(hipGetLastError());
}

}  // namespace mrla
