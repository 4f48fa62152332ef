// light_apply_bwd_wide in packed FP32 (round 6).  Same pass, same buffers, same launch geometry, same sums as
// light_apply_bwd_wide<.., GELU = false, ..> of light_nhwc_wide.hip -- the step is re-cut so that two neighbouring COLUMNS
// share an instruction wherever their operands sit in one even-aligned register pair (nhwc_rows_pk.h):
//   U = dwconv3x3(x)          per row of taps: k = 0 and k = 2 packed, k = 1 (operands straddle two pairs) as two plain FMAs
//   dm, dU, lam*dm            packed (element-wise in the column index)
//   dWv partial sums          k = 0 / k = 2 as packed accumulators (one per column parity, added once at the end), k = 1 plain
//   dx = dwconv3x3^T(dU)      all nine taps packed: the dU rows are ALSO kept cut at even columns (four v_pk_mov per step)
//   relu mask                 per element (v_cmp + v_cndmask have no packed form)
//   bn3's backward sums       packed; the bf16 rounding of dpre is one v_cvt_pk per pair, shared with the store
// U, dm, dU and dx keep the operation ORDER of the plain kernel (a packed FMA is two independent FMAs): bit-identical.  The
// dWv and bn3 sums run in two accumulators per tap instead of one: same sums, different rounding (fp32 accumulation noise).
// VALU instructions per element (ISA count of the bf16 / relu / bn3-sums instance): 49.9 -> see profiles/r06_notes.md.
#pragma once
#include "nhwc_rows_pk.h"

namespace mrla {

// DEPTH = row sets in flight per wave (round 6).  What a step reads from LDS -- "set k" = x row k+1, dOut row k, o row k, y3
// row k-1 -- is fetched DEPTH steps ahead into rings of DEPTH (x, o, y3) / DEPTH + 1 (dOut: row k-1 is read again) row
// buffers, and the wait at the top of a step is COUNTED: the newest (DEPTH - 1) sets and the previous step's stores stay in
// flight.  With one set in flight (rounds 2 - 5) a CU has 8 waves x 4.6 KB = 37 KB on its way, and 256 CUs x 37 KB / ~2 us of
// loaded HBM latency is the 4.5 - 4.7 TB/s every variant of this pass measured -- the plain kernel at 49.9 VALU instructions
// per element and this one at 36 alike (profiles/r06_notes.md section 2): bytes in flight, not issue slots.  The cross-wave
// reduction area ALIASES the row buffers (it is used after the last row has landed), so the 18 KB per wave of DEPTH = 2 fit:
// 8 waves x 18 KB = 144 KB of the CU's 160 KB.  fp32 rows are twice as large: DEPTH = 1 there.
template <typename T, bool PRE, int DEPTH> constexpr int apply_bwd_pk_wave_bytes() {
  constexpr int b = DEPTH * RowIO<T, kS + 4>::kBytes + (2 * DEPTH + 1) * RowIO<T, kS + 2>::kBytes +
                    (2 + (PRE ? DEPTH : 0)) * RowIO<T, kS>::kBytes;
  return b > 9 * kWave * (int)sizeof(float) ? b : 9 * kWave * (int)sizeof(float);
}

template <typename T, bool HAS_O, bool RELU, bool RAGGED, bool PRE, int DEPTH>
__global__ __launch_bounds__(kBwdWaves * kWave) void light_apply_bwd_wide_pk(
    const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv,
    const float* __restrict__ gate, const float* __restrict__ cb, const float* __restrict__ lam,
    const float* __restrict__ dp, const float* __restrict__ dyx, T* __restrict__ dx, T* __restrict__ dprev,
    float* __restrict__ dwv_part, const T* __restrict__ pre, const float* __restrict__ pre_center,
    float* __restrict__ pre_tmom, int B, int C, int H, int W, int BG, int d, int res, int wc) {
  static_assert(kS == 7, "the pair cuts below are written out for 7 owned columns");
  static_assert(!PRE || RELU, "the deferred-BatchNorm sums belong to the fused relu(pre + o) producer");
  static_assert(DEPTH == 1 || DEPTH == 2, "row sets in flight");
  MRLA_WIDE_PROLOGUE(0, (apply_bwd_pk_wave_bytes<T, PRE, DEPTH>()))      // (red == smem_raw: aliases the row buffers)
  constexpr int XB_ = RowIO<T, kS + 4>::kBytes, GB = RowIO<T, kS + 2>::kBytes, SB = RowIO<T, kS>::kBytes;
  // vector-memory instructions of one fetched set, and of the stores a step >= 1 issues after its fetches
  constexpr int K_LOADS = RowIO<T, kS + 4>::NL + (HAS_O ? 2 : 1) * RowIO<T, kS + 2>::NL + (PRE ? RowIO<T, kS>::NL : 0);
  constexpr int K_STORES = (HAS_O ? 2 : 1) * RowIO<T, kS>::NL;
  static_assert((DEPTH - 1) * K_LOADS + K_STORES < 64, "vmcnt immediate");
  typedef PkRow<kS + 4, 1> XRow;     // x window, columns s0-2 .. s0+kS+1:   head | (1,2) (3,4) (5,6) (7,8) (9,10)
  typedef PkRow<kS + 2, 1> URow;     // dOut / o / dU window, columns s0-1 .. s0+kS:   head | (1,2) (3,4) (5,6) (7,8)
  typedef PkRow<kS + 2, 0> SRow;     // the dU window cut at even columns:   (0,1) (2,3) (4,5) (6,7) | [8 = URow.p[3].y]
  typedef PkRow<kS, 0> ORow;         // owned columns:   (0,1) (2,3) (4,5) | 6     (owned j = window column j + 1)
  // rings: x row r in slot r % DEPTH, dOut row r in slot r % (DEPTH + 1), o row r in slot r % DEPTH, y3 row r in slot r % DEPTH
  unsigned char* ringX = wbuf;
  unsigned char* ringG = ringX + DEPTH * XB_;
  unsigned char* ringO = ringG + (DEPTH + 1) * GB;
  T* bufS1 = reinterpret_cast<T*>(ringO + DEPTH * GB);
  T* bufS2 = reinterpret_cast<T*>(ringO + DEPTH * GB + SB);
  unsigned char* ringP = ringO + DEPTH * GB + 2 * SB;                  // (PRE)
  auto xbuf = [&](int r) { return reinterpret_cast<T*>(ringX + (r & (DEPTH - 1)) * XB_); };
  auto gbuf = [&](int r) { return reinterpret_cast<T*>(ringG + (DEPTH == 1 ? (r & 1) : (r % 3)) * GB); };      // (r >= 0)
  auto obuf = [&](int r) { return reinterpret_cast<T*>(ringO + (r & (DEPTH - 1)) * GB); };
  auto pbuf = [&](int r) { return reinterpret_cast<T*>(ringP + (r & (DEPTH - 1)) * SB); };
  v2f pm0 = splat2(0.f), pm1 = splat2(0.f);          // (PRE) sum dpre, sum dpre * (y3 - center), per column parity
  const float pcen = (PRE && pre_center) ? pre_center[c] : 0.f;
  const int G = C / d;
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[c * 9 + k];
  const float e_ = cb ? cb[c * 4 + 0] : 1.f, f_ = cb ? cb[c * 4 + 1] : 0.f;
  const float Gc = cb ? cb[c * 4 + 2] : 0.f, Hc = cb ? cb[c * 4 + 3] : 0.f;
  const float lm = (HAS_O && lam) ? lam[c] : 1.f;
  const float resf = res ? 1.f : 0.f;
  v2f wgA[3], wgB[3];                                // dWv taps k = 0 / k = 2 of row i, by column parity
  float wgm[3];                                      // dWv tap k = 1 of row i
#pragma unroll
  for (int i = 0; i < 3; ++i) { wgA[i] = splat2(0.f); wgB[i] = splat2(0.f); wgm[i] = 0.f; }
  const int yy = MRLA_REVERSE_APPLY ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
  const int b_end = min(B, (yy + 1) * BG);
  for (int b = yy * BG; b < b_end; ++b) {
    const size_t ioff = (size_t)b * H * rowelems;
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
      // dU exists inside the image only: the gate per window column, 0 outside (wave-uniform; whole strips: only the two
      // halo columns can be outside).  dm is finite there (dOut and o arrive as zeros), so the product is the mask.
      URow am;
      {
        float t[kS + 2];
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) {
          const int col = s0 - 1 + j;
          const bool in = RAGGED ? (col >= 0 && col < W) : (j == 0 ? s0 > 0 : (j == kS + 1 ? s0 + kS < W : true));
          t[j] = in ? a : 0.f;
        }
        am.head = t[0];
#pragma unroll
        for (int q = 0; q < 4; ++q) am.p[q] = (v2f){t[2 * q + 1], t[2 * q + 2]};
      }
      XRow xa, xb, xc;                               // x rows rr-1, rr, rr+1
      URow gv, ov;                                   // dOut / o of row rr
      ORow gp, pv;                                   // dOut / (PRE) y3 of row rr-1 on the owned columns
      xa.clear(); xb.clear(); xc.clear(); gv.clear(); ov.clear(); gp.clear(); pv.clear();
      URow ua, ub, uc;                               // dU rows rr-2, rr-1, rr
      SRow sa, sb, sc;                               // ... cut at even columns
      ORow d0, d1, d2;                               // lam*dm of rows rr-1 / rr (RELU: do = lam*dm + dx one step later)
      ua.clear(); ub.clear(); sa.clear(); sb.clear(); d0.clear();
      // set k = what step k reads from LDS (rows outside the image arrive as zeros: every set has the same instruction count)
      auto fetch_set = [&](int k) {
        row_fetch<T, kS + 4>(ax, xi, k + 1, H, rowelems, xbuf(k + 1));
        row_fetch<T, kS + 2>(ag, gi, k, H, rowelems, gbuf(k));
        if (HAS_O) row_fetch<T, kS + 2>(ag, oi, k, H, rowelems, obuf(k));
        if (PRE) row_fetch<T, kS>(as, pri, k - 1, H, rowelems, pbuf(k - 1));
      };
      row_fetch<T, kS + 4>(ax, xi, 0, H, rowelems, xbuf(0));
      rows_landed();
      pk_read_issue<T, kS + 4, 1>(xbuf(0), lane, xb);
      pk_read_fence(xb, true);
      fetch_set(0);
      if (DEPTH == 2) fetch_set(1);
      auto step = [&](int rr, XRow& XA, XRow& XB, XRow& XC, URow& UA, URow& UB, URow& UC, SRow& SA, SRow& SB, SRow& SC,
                      ORow& DP, ORow& DC) {
        // set rr has landed; the (DEPTH - 1) sets fetched since and the rows step rr-1 >= 1 stored after its fetches may stay
        // in flight (vmcnt counts in issue order)
        if (rr <= 1) rows_landed_keep<(DEPTH - 1) * K_LOADS>(); else rows_landed_keep<(DEPTH - 1) * K_LOADS + K_STORES>();
        pk_read_issue<T, kS + 4, 1>(xbuf(rr + 1), lane, XC);
        pk_read_issue<T, kS + 2, 1>(gbuf(rr), lane, gv);
        if (HAS_O) pk_read_issue<T, kS + 2, 1>(obuf(rr), lane, ov);
        if (rr >= 1) pk_read_issue<T, kS, 0>(gbuf(rr - 1), lane, gp, 1);   // row rr-1, owned pixels
        if (PRE && rr >= 1) pk_read_issue<T, kS, 0>(pbuf(rr - 1), lane, pv);
        pk_read_fence(XC, true);
        pk_read_fence(gv, false);
        if (HAS_O) pk_read_fence(ov, false);
        pk_read_fence(gp, false);
        if (PRE) pk_read_fence(pv, false);
        fetch_set(rr + DEPTH);     // into the slots whose rows were just read
        if (rr >= H) {             // the step past the last row only finishes dx[H-1]
          UC.clear(); SC.clear(); DC.clear();
        } else {
          const v2f E2 = splat2(E), F2 = splat2(F), G2 = splat2(Gc), H2 = splat2(Hc), lm2 = splat2(lm);
          // U on window column 0 (x columns 0 .. 2 of the three rows)
          float u0 = w[0] * XA.head;
          u0 = fmaf(w[1], XA.p[0].x, u0); u0 = fmaf(w[2], XA.p[0].y, u0);
          u0 = fmaf(w[3], XB.head, u0); u0 = fmaf(w[4], XB.p[0].x, u0); u0 = fmaf(w[5], XB.p[0].y, u0);
          u0 = fmaf(w[6], XC.head, u0); u0 = fmaf(w[7], XC.p[0].x, u0); u0 = fmaf(w[8], XC.p[0].y, u0);
          float dm0 = fmaf(E, gv.head, Hc);
          dm0 = fmaf(F, u0, dm0);
          if (HAS_O) dm0 = fmaf(Gc, ov.head, dm0);
          UC.head = am.head * dm0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // U on window columns (2q+1, 2q+2): x columns (2q+1, 2q+2) | (2q+2, 2q+3) | (2q+3, 2q+4)
            v2f u = splat2(w[0]) * XA.p[q];
            u.x = fmaf(w[1], XA.p[q].y, u.x); u.y = fmaf(w[1], XA.p[q + 1].x, u.y);
            u = pk_fma(splat2(w[2]), XA.p[q + 1], u);
            u = pk_fma(splat2(w[3]), XB.p[q], u);
            u.x = fmaf(w[4], XB.p[q].y, u.x); u.y = fmaf(w[4], XB.p[q + 1].x, u.y);
            u = pk_fma(splat2(w[5]), XB.p[q + 1], u);
            u = pk_fma(splat2(w[6]), XC.p[q], u);
            u.x = fmaf(w[7], XC.p[q].y, u.x); u.y = fmaf(w[7], XC.p[q + 1].x, u.y);
            u = pk_fma(splat2(w[8]), XC.p[q + 1], u);
            v2f dm = pk_fma(E2, gv.p[q], H2);
            dm = pk_fma(F2, u, dm);
            if (HAS_O) dm = pk_fma(G2, ov.p[q], dm);
            UC.p[q] = am.p[q] * dm;
            if (q < 3) DC.p[q] = lm2 * dm; else DC.tail = lm * dm.x;      // (columns beyond the image are dropped by the store)
          }
          // the same row cut at even columns
          SC.p[0] = (v2f){UC.head, UC.p[0].x};
#pragma unroll
          for (int m = 1; m < 4; ++m) SC.p[m] = (v2f){UC.p[m - 1].y, UC.p[m].x};
          // dWv[i][k] += dU[rr][col] * x[rr+i-1][col+k-1] over the owned columns (window columns 1 .. 7)
          auto wgrad = [&](int i, const XRow& X) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              const v2f du = UC.p[q];
              wgA[i] = pk_fma(du, X.p[q], wgA[i]);
              wgm[i] = fmaf(du.x, X.p[q].y, wgm[i]);
              wgm[i] = fmaf(du.y, X.p[q + 1].x, wgm[i]);
              wgB[i] = pk_fma(du, X.p[q + 1], wgB[i]);
            }
            const float d7 = UC.p[3].x;
            wgA[i].x = fmaf(d7, X.p[3].x, wgA[i].x);
            wgm[i] = fmaf(d7, X.p[3].y, wgm[i]);
            wgB[i].x = fmaf(d7, X.p[4].x, wgB[i].x);
          };
          wgrad(0, XA); wgrad(1, XB); wgrad(2, XC);
          if (HAS_O && !RELU) pk_row_store<T>(as, doo, rr, rowelems, lane, bufS2, DC);
        }
        // dx[rr-1] on the owned columns:  dx[ro][col] = sum_{i,k} w[i][k] * dU[ro-i+1][col-k+1]
        if (rr >= 1) {
          ORow yrow, dsum;
          const v2f dy2 = splat2(dy), res2 = splat2(resf), ncen2 = splat2(-pcen);
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            // owned columns (2q, 2q+1) = window columns (2q+1, 2q+2): tap k reads window columns shifted by 1 - k
            v2f s9 = splat2(w[0]) * SC.p[q + 1];
            s9 = pk_fma(splat2(w[1]), UC.p[q], s9); s9 = pk_fma(splat2(w[2]), SC.p[q], s9);
            s9 = pk_fma(splat2(w[3]), SB.p[q + 1], s9); s9 = pk_fma(splat2(w[4]), UB.p[q], s9); s9 = pk_fma(splat2(w[5]), SB.p[q], s9);
            s9 = pk_fma(splat2(w[6]), SA.p[q + 1], s9); s9 = pk_fma(splat2(w[7]), UA.p[q], s9); s9 = pk_fma(splat2(w[8]), SA.p[q], s9);
            v2f y = pk_fma(res2, gp.p[q], s9 + dy2);
            if (RELU) {                                 // XA = x[rr-1]; owned column j <-> x window column j + 2
              y.x = (XA.p[q].y > 0.f) ? y.x : 0.f;
              y.y = (XA.p[q + 1].x > 0.f) ? y.y : 0.f;
            }
            yrow.p[q] = y;
            dsum.p[q] = DP.p[q] + y;
            if (PRE) {             // (columns past the image have x = 0, hence y = 0: nothing to mask)
              // the sums are of dpre AS STORED (rounded to T): where the MRLA branch's contribution is below half an ulp
              // of dOut it is lost in the stored tensor, systematically -- the BatchNorm backward must see the same values
              const v2f yq = pk_as_stored<T>(y);
              pm0 += yq;
              pm1 = pk_fma(yq, pv.p[q] + ncen2, pm1);
            }
          }
          {                                             // owned column 6 = window column 7
            float s9 = w[0] * UC.p[3].y;
            s9 = fmaf(w[1], UC.p[3].x, s9); s9 = fmaf(w[2], UC.p[2].y, s9);
            s9 = fmaf(w[3], UB.p[3].y, s9); s9 = fmaf(w[4], UB.p[3].x, s9); s9 = fmaf(w[5], UB.p[2].y, s9);
            s9 = fmaf(w[6], UA.p[3].y, s9); s9 = fmaf(w[7], UA.p[3].x, s9); s9 = fmaf(w[8], UA.p[2].y, s9);
            float y = fmaf(resf, gp.tail, s9 + dy);
            if (RELU) y = (XA.p[3].y > 0.f) ? y : 0.f;
            yrow.tail = y;
            dsum.tail = DP.tail + y;
            if (PRE) {
              const float yq = to_f(from_f<T>(y));
              pm0.x += yq;
              pm1.x = fmaf(yq, pv.tail - pcen, pm1.x);
            }
          }
          pk_row_store<T>(as, dxo, rr - 1, rowelems, lane, bufS1, yrow);
          if (RELU && HAS_O) pk_row_store<T>(as, doo, rr - 1, rowelems, lane, bufS2, dsum);
        }
      };
      // steps rr = 0 .. H; after three steps every array is back in its starting role
      int rr = 0;
      for (; rr + 2 <= H; rr += 3) {
        step(rr,     xa, xb, xc, ua, ub, uc, sa, sb, sc, d0, d1);
        step(rr + 1, xb, xc, xa, ub, uc, ua, sb, sc, sa, d1, d2);
        step(rr + 2, xc, xa, xb, uc, ua, ub, sc, sa, sb, d2, d0);
      }
      if (rr <= H) {
        step(rr, xa, xb, xc, ua, ub, uc, sa, sb, sc, d0, d1);
        if (rr + 1 <= H) step(rr + 1, xb, xc, xa, ub, uc, ua, sb, sc, sa, d1, d2);
      }
      rows_landed();
    }
  }
  float wg[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    wg[3 * i] = wgA[i].x + wgA[i].y;
    wg[3 * i + 1] = wgm[i];
    wg[3 * i + 2] = wgB[i].x + wgB[i].y;
  }
  wg_reduce<9>(wg, red, lane, wave, nwaves, wc);
  if (wave < wc) {
#pragma unroll
    for (int k = 0; k < 9; ++k) dwv_part[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * C + c) * 9 + k] = wg[k];
  }
  if (PRE) {
    float pm[2] = {pm0.x + pm0.y, pm1.x + pm1.y};
    wg_reduce<2>(pm, red, lane, wave, nwaves, wc);
    if (wave < wc) {
      pre_tmom[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * C + c) * 2 + 0] = pm[0];
      pre_tmom[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * C + c) * 2 + 1] = pm[1];
    }
  }
}

}  // namespace mrla
