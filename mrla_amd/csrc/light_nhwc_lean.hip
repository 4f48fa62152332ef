// The backward passes of the MRLA-light block tail WITHOUT a stored x_t (round 6; channels_last, C % 64 == 0, 16-bit types).
//
// x_t = relu(round(round(psc*y3 + psh) + o)) -- bn3's affine on conv3's raw output y3, the shortcut add and the ReLU
// (resnet_mrla_light.py:101-114) -- is what the depthwise 3x3 of the MRLA branch convolves (mrla_light_module.py:61).  Rounds
// 2 - 5 wrote it once (fused forward statistics pass) and read it three more times (apply_fwd, stats_bwd, apply_bwd): 15N
// elements per block and training step against section 8(d)'s compulsory 8N.  Every one of those passes also reads `o` (and
// apply_bwd reads y3 for bn3's backward sums), so x_t can be RE-FORMED from rows that are in LDS anyway -- the same
// form_x_row() the forward used, hence the same bits -- and never touch HBM:
//   forward   statistics: reads y3, o            (2N, was 3N: no x_t store)      apply: reads y3, o, writes out  (3N)
//   backward  statistics: reads dOut, y3, o      (3N)                             apply: reads dOut, y3, o; writes dx, do (5N, was 6N)
// = 13N.  The arithmetic this adds (one fma, two roundings, an add and a max per window element) goes into issue slots the
// passes do not use: they stream at the rate of their access pattern with the vector unit about half busy
// (profiles/r06_notes.md section 2: 36 instead of 50 VALU instructions per element changed nothing).
// Same launch geometry, same sums, same partial-row layout as the x_t-reading kernels of light_nhwc_wide.hip.
#include "light_nhwc_wide.h"

namespace mrla {

// The shift per window column of an NW-wide window with HALO columns either side of the kS owned ones: `ash` inside the image,
// 0 outside (so that the padding of the 3x3 stays zero).  Whole strips: only the halo columns can be outside.
template <bool RAGGED, int NW, int HALO> struct WindowShifts;
template <int NW, int HALO> struct WindowShifts<true, NW, HALO> {
  float v[NW];
  __device__ __forceinline__ void set(float ash, int s0, int W) {
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int col = s0 - HALO + j;
      v[j] = (col >= 0 && col < W) ? ash : 0.f;      // wave-uniform predicate
    }
  }
  __device__ __forceinline__ float at(int j) const { return v[j]; }
};
template <int NW, int HALO> struct WindowShifts<false, NW, HALO> {
  float first, mid, last;
  __device__ __forceinline__ void set(float ash, int s0, int W) {
    first = s0 > 0 ? ash : 0.f;
    mid = ash;
    last = s0 + kS < W ? ash : 0.f;
  }
  __device__ __forceinline__ float at(int j) const { return j < HALO ? first : (j >= NW - HALO ? last : mid); }
};

// form_x_row() of light_nhwc_wide.h for any window width
template <typename T, bool AFF, int NW, typename SH>
__device__ __forceinline__ void form_x_window(const RawRow<NW>& pre, const RawRow<NW>& o, float asc, const SH& sh,
                                              float (&dst)[NW]) {
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const float z = AFF ? to_f(from_f<T>(fmaf(asc, pre.v[j], sh.at(j)))) : pre.v[j];
    dst[j] = fmaxf(to_f(from_f<T>(z + o.v[j])), 0.f);
  }
}

// ------------------------------------------------------------------------------------------------
// backward statistics (light_stats_bwd_wide with x_t re-formed): per (image, channel) sums of dOut, dOut*(V - pV),
// dOut*(o - pO) about the pivots the forward statistics pass recorded.
// per wave: y3 row, two o rows (row r is read again, owned columns, when dOut[r] is paired with it), dOut row
// ------------------------------------------------------------------------------------------------
template <typename T> constexpr int stats_bwd_lean_wave_bytes() { return 3 * RowIO<T, kS + 2>::kBytes + RowIO<T, kS>::kBytes; }

template <typename T, bool AFF, bool RAGGED>
__global__ __launch_bounds__(kMaxStrips * kWave) __attribute__((amdgpu_waves_per_eu(3))) void light_stats_bwd_lean_wide(
    const T* __restrict__ dout, const T* __restrict__ pre, const T* __restrict__ o, const float* __restrict__ wv,
    const float* __restrict__ psc, const float* __restrict__ psh, const float* __restrict__ mom,
    float* __restrict__ bmom, int B, int C, int H, int W, int BG, int wc) {
  MRLA_WIDE_PROLOGUE(D_N, stats_bwd_lean_wave_bytes<T>())
  constexpr int RB = RowIO<T, kS + 2>::kBytes;
  T* bufP = reinterpret_cast<T*>(wbuf);
  unsigned char* bufO2 = wbuf + RB;                  // o row r lives in half (r & 1)
  T* bufG = reinterpret_cast<T*>(wbuf + 3 * RB);
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[c * 9 + k];
  const float asc = AFF ? psc[c] : 1.f, ash = AFF ? psh[c] : 0.f;
  const int b_end = min(B, (int)(blockIdx.y + 1) * BG);
  for (int b = blockIdx.y * BG; b < b_end; ++b) {
    const size_t ioff = (size_t)b * H * rowelems;
    const T* pi = pre + ioff;
    const T* oi = o + ioff;
    const T* gi = dout + ioff;
    float acc[D_N] = {0.f, 0.f, 0.f};
    const float pV = mom ? mom[((size_t)b * C + c) * M_REC + M_PV] : 0.f;
    const float pO = mom ? mom[((size_t)b * C + c) * M_REC + M_PO] : 0.f;
    for (int s = sfirst; s < nstrips; s += sstep) {
      const int s0 = s * kS;
      RowIO<T, kS + 2> ax;
      RowIO<T, kS> ag;
      make_row_io<T, kS + 2>(ax, s0 - 1, kS + 2, W, C, cbase, lane);
      make_row_io<T, kS>(ag, s0, kS, W, C, cbase, lane);
      WindowShifts<RAGGED, kS + 2, 1> shj;
      shj.set(ash, s0, W);
      RawRow<kS + 2> praw, oraw;
      RawRow<kS> ov, gv;
      praw.clear(); oraw.clear(); ov.clear(); gv.clear();
      float xa[kS + 2], xb[kS + 2], xc[kS + 2];      // x_t rows r-1, r, r+1 on columns s0-1 .. s0+kS
      auto obuf = [&](int r) { return reinterpret_cast<T*>(bufO2 + (r & 1) * RB); };
#pragma unroll
      for (int j = 0; j < kS + 2; ++j) xa[j] = 0.f;
      row_fetch<T, kS + 2>(ax, pi, 0, H, rowelems, bufP);
      row_fetch<T, kS + 2>(ax, oi, 0, H, rowelems, obuf(0));
      rows_landed();
      row_read<T, kS + 2>(bufP, lane, praw);
      row_read<T, kS + 2>(obuf(0), lane, oraw);
      row_fetch<T, kS + 2>(ax, pi, 1, H, rowelems, bufP);
      row_fetch<T, kS + 2>(ax, oi, 1, H, rowelems, obuf(1));
      row_fetch<T, kS>(ag, gi, 0, H, rowelems, bufG);
      form_x_window<T, AFF, kS + 2>(praw, oraw, asc, shj, xb);
      // dOut is zero beyond the image, so columns of a ragged last strip drop out of every sum by themselves
      auto step = [&](int r, float (&XA)[kS + 2], float (&XB)[kS + 2], float (&XC)[kS + 2]) {
        rows_landed();
        row_read_issue<T, kS + 2>(bufP, lane, praw);         // row r+1
        row_read_issue<T, kS + 2>(obuf(r + 1), lane, oraw);
        row_read_issue<T, kS>(obuf(r), lane, ov, 1);         // owned pixels of row r
        row_read_issue<T, kS>(bufG, lane, gv);
        row_read_fence(praw, true);
        row_read_fence(oraw, false);
        row_read_fence(ov, false);
        row_read_fence(gv, false);
        row_fetch<T, kS + 2>(ax, pi, r + 2, H, rowelems, bufP);
        row_fetch<T, kS + 2>(ax, oi, r + 2, H, rowelems, obuf(r));
        row_fetch<T, kS>(ag, gi, r + 1, H, rowelems, bufG);
        if (r + 1 < H) {
          form_x_window<T, AFF, kS + 2>(praw, oraw, asc, shj, XC);
        } else {
#pragma unroll
          for (int j = 0; j < kS + 2; ++j) XC[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          const float v = conv_at(w, XA, XB, XC, j);
          acc[D_D] += gv.v[j];
          acc[D_DV] = fmaf(gv.v[j], v - pV, acc[D_DV]);
          acc[D_DO] = fmaf(gv.v[j], ov.v[j] - pO, acc[D_DO]);
        }
      };
      MRLA_ROTATE3(H, step, xa, xb, xc)
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
// backward apply (light_apply_bwd_wide<.., RELU, PRE> with x_t re-formed):
//   dm = E*dOut + F*V + G*o + H,  dU = a*dm,  dx = [x_t > 0] * (dwconv3x3^T(dU) + res*dOut + dy/(hw)),  do = lam*dm + dx,
//   dWv partial sums per image group,  PRE: bn3's backward sums (sum dpre, sum dpre * (y3 - center)) of dpre = dx as stored.
// Strip-local windows (columns relative to s0): y3 and o rows over cols -2..kS+1 (kS+4 wide) -> x_t rows rr-1..rr+1;
// dOut rows over cols -1..kS.  Step rr forms x_t[rr+1] from y3[rr+1] and o[rr+1]; o[rr] (cols -1..kS), dOut[rr-1] and
// y3[rr-1] (owned cols) are read a second time from their LDS rings instead of being kept in registers.
// per wave: three y3 rows (slot r % 3), two o rows, two dOut rows, two store buffers
// ------------------------------------------------------------------------------------------------
template <typename T> constexpr int apply_bwd_lean_wave_bytes() {
  return 5 * RowIO<T, kS + 4>::kBytes + 2 * RowIO<T, kS + 2>::kBytes + 2 * RowIO<T, kS>::kBytes;
}

template <typename T, bool AFF, bool RAGGED, bool PRE>
__global__ __launch_bounds__(kBwdWaves * kWave) void light_apply_bwd_lean_wide(
    const T* __restrict__ dout, const T* __restrict__ pre, const T* __restrict__ o, const float* __restrict__ wv,
    const float* __restrict__ psc, const float* __restrict__ psh, const float* __restrict__ gate,
    const float* __restrict__ cb, const float* __restrict__ lam, const float* __restrict__ dp,
    const float* __restrict__ dyx, T* __restrict__ dx, T* __restrict__ dprev, float* __restrict__ dwv_part,
    const float* __restrict__ pre_center, float* __restrict__ pre_tmom, int B, int C, int H, int W, int BG, int d, int res,
    int wc) {
  MRLA_WIDE_PROLOGUE(9, (apply_bwd_lean_wave_bytes<T>()))
  constexpr int XB_ = RowIO<T, kS + 4>::kBytes, GB = RowIO<T, kS + 2>::kBytes, SB = RowIO<T, kS>::kBytes;
  unsigned char* ringP = wbuf;                       // y3 row r in slot r % 3
  unsigned char* ringO = wbuf + 3 * XB_;             // o row r in slot r & 1
  unsigned char* ringG = wbuf + 5 * XB_;             // dOut row r in slot r & 1
  T* bufS1 = reinterpret_cast<T*>(wbuf + 5 * XB_ + 2 * GB);
  T* bufS2 = reinterpret_cast<T*>(wbuf + 5 * XB_ + 2 * GB + SB);
  auto pbuf = [&](int r) { return reinterpret_cast<T*>(ringP + ((r + 3) % 3) * XB_); };
  auto obuf = [&](int r) { return reinterpret_cast<T*>(ringO + (r & 1) * XB_); };
  auto gbuf = [&](int r) { return reinterpret_cast<T*>(ringG + (r & 1) * GB); };
  float pm[2] = {0.f, 0.f};                          // (PRE) sum dpre, sum dpre * (y3 - center) over this workgroup's images
  const float pcen = (PRE && pre_center) ? pre_center[c] : 0.f;      // bn3's batch mean: no cancelling subtraction later
  const int G = C / d;
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[c * 9 + k];
  const float asc = AFF ? psc[c] : 1.f, ash = AFF ? psh[c] : 0.f;
  const float e_ = cb ? cb[c * 4 + 0] : 1.f, f_ = cb ? cb[c * 4 + 1] : 0.f;
  const float Gc = cb ? cb[c * 4 + 2] : 0.f, Hc = cb ? cb[c * 4 + 3] : 0.f;
  const float lm = lam ? lam[c] : 1.f;
  const float resf = res ? 1.f : 0.f;
  float wg[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int yy = MRLA_REVERSE_APPLY ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;   // (see light_nhwc_wide.h)
  const int b_end = min(B, (yy + 1) * BG);
  for (int b = yy * BG; b < b_end; ++b) {
    const size_t ioff = (size_t)b * H * rowelems;
    const T* pi = pre + ioff;
    const T* gi = dout + ioff;
    const T* oi = o + ioff;
    T* dxo = dx + ioff;
    T* doo = dprev + ioff;
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
      WindowShifts<RAGGED, kS + 4, 2> shj;
      shj.set(ash, s0, W);
      RawRow<kS + 4> praw, oraw;                     // y3 / o of row rr+1 on columns -2 .. kS+1
      RawRow<kS + 2> gv, ov;                         // dOut / o of row rr on columns -1 .. kS
      RawRow<kS> gp;                                 // dOut of row rr-1 on the owned columns
      RawRow<kS> pv;                                 // (PRE) y3 of row rr-1 on the owned columns
      praw.clear(); oraw.clear(); gv.clear(); ov.clear(); gp.clear(); pv.clear();
      float xa[kS + 4], xb[kS + 4], xc[kS + 4];      // x_t rows rr-1, rr, rr+1
      float ua[kS + 2], ub[kS + 2], uc[kS + 2];      // dU rows rr-2, rr-1, rr
      float d0[kS], d1[kS], d2[kS];                  // lam*dm of rows rr-1 / rr (do = lam*dm + dx one step later)
#pragma unroll
      for (int j = 0; j < kS + 4; ++j) xa[j] = 0.f;
#pragma unroll
      for (int j = 0; j < kS + 2; ++j) { ua[j] = 0.f; ub[j] = 0.f; }
#pragma unroll
      for (int j = 0; j < kS; ++j) d0[j] = 0.f;
      row_fetch<T, kS + 4>(ax, pi, 0, H, rowelems, pbuf(0));
      row_fetch<T, kS + 4>(ax, oi, 0, H, rowelems, obuf(0));
      rows_landed();
      row_read<T, kS + 4>(pbuf(0), lane, praw);
      row_read<T, kS + 4>(obuf(0), lane, oraw);
      row_fetch<T, kS + 4>(ax, pi, 1, H, rowelems, pbuf(1));
      row_fetch<T, kS + 4>(ax, oi, 1, H, rowelems, obuf(1));
      row_fetch<T, kS + 2>(ag, gi, 0, H, rowelems, gbuf(0));
      form_x_window<T, AFF, kS + 4>(praw, oraw, asc, shj, xb);
      auto step = [&](int rr, float (&XA)[kS + 4], float (&XB)[kS + 4], float (&XC)[kS + 4], float (&UA)[kS + 2],
                      float (&UB)[kS + 2], float (&UC)[kS + 2], float (&DP)[kS], float (&DC)[kS]) {
        // steps rr-1 >= 1 stored two rows (dx and do) after their fetches; those may stay in flight
        if (rr <= 1) rows_landed(); else rows_landed_keep<2 * RowIO<T, kS>::NL>();
        row_read_issue<T, kS + 4>(pbuf(rr + 1), lane, praw);
        row_read_issue<T, kS + 4>(obuf(rr + 1), lane, oraw);
        row_read_issue<T, kS + 2>(gbuf(rr), lane, gv);
        row_read_issue<T, kS + 2>(obuf(rr), lane, ov, 1);                    // row rr, columns -1 .. kS
        if (rr >= 1) row_read_issue<T, kS>(gbuf(rr - 1), lane, gp, 1);       // row rr-1, owned pixels
        if (PRE && rr >= 1) row_read_issue<T, kS>(pbuf(rr - 1), lane, pv, 2);
        row_read_fence(praw, true);
        row_read_fence(oraw, false);
        row_read_fence(gv, false);
        row_read_fence(ov, false);
        row_read_fence(gp, false);
        if (PRE) row_read_fence(pv, false);
        row_fetch<T, kS + 4>(ax, pi, rr + 2, H, rowelems, pbuf(rr + 2));     // (the slot of row rr-1)
        row_fetch<T, kS + 4>(ax, oi, rr + 2, H, rowelems, obuf(rr));
        row_fetch<T, kS + 2>(ag, gi, rr + 1, H, rowelems, gbuf(rr + 1));
        if (rr + 1 < H) {
          form_x_window<T, AFF, kS + 4>(praw, oraw, asc, shj, XC);
        } else {
#pragma unroll
          for (int j = 0; j < kS + 4; ++j) XC[j] = 0.f;
        }
        if (rr >= H) {             // the step past the last row only finishes dx[H-1]
#pragma unroll
          for (int j = 0; j < kS + 2; ++j) UC[j] = 0.f;
#pragma unroll
          for (int j = 0; j < kS; ++j) DC[j] = 0.f;
        } else {
#pragma unroll
          for (int j = 0; j < kS + 2; ++j) {
            const int col = s0 - 1 + j;
            // dU exists inside the image only (wave-uniform; for whole strips only the two halo columns can be outside)
            const bool in = RAGGED ? (col >= 0 && col < W) : (j == 0 ? s0 > 0 : (j == kS + 1 ? s0 + kS < W : true));
            const float u = conv_at(w, XA, XB, XC, j);                    // window cols j..j+2 <-> image cols col-1..col+1
            float dm = fmaf(E, gv.v[j], Hc);
            dm = fmaf(F, u, dm);
            dm = fmaf(Gc, ov.v[j], dm);
            float du = a * dm;
            if (RAGGED || j == 0 || j == kS + 1) du = in ? du : 0.f;
            if (j >= 1 && j <= kS) {                                        // owned column (compile-time after unroll)
              float ldm = lm * dm;                   // (a rounded product, as in light_apply_bwd_wide: see the note there)
              asm("" : "+v"(ldm));
              DC[j - 1] = ldm;                       // (columns beyond the image are dropped by the store)
              // dWv[i][k] += dU[rr][col] * x[rr+i-1][col+k-1]
              if (!MRLA_EXP_SKIP_WG) {
              wg[0] = fmaf(du, XA[j], wg[0]); wg[1] = fmaf(du, XA[j + 1], wg[1]); wg[2] = fmaf(du, XA[j + 2], wg[2]);
              wg[3] = fmaf(du, XB[j], wg[3]); wg[4] = fmaf(du, XB[j + 1], wg[4]); wg[5] = fmaf(du, XB[j + 2], wg[5]);
              wg[6] = fmaf(du, XC[j], wg[6]); wg[7] = fmaf(du, XC[j + 1], wg[7]); wg[8] = fmaf(du, XC[j + 2], wg[8]);
              }
            }
            UC[j] = du;
          }
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
            y = (XA[j + 2] > 0.f) ? y : 0.f;                              // XA = x_t[rr-1]; owned col j <-> window j+2
            yrow[j] = y;
            dsum[j] = DP[j] + y;
            if (PRE) {             // (columns past the image have x = 0, hence y = 0: nothing to mask)
              // the sums are of dpre AS STORED (rounded to T): the BatchNorm backward must see the values it will read
              const float yq = to_f(from_f<T>(y));
              pm[0] += yq;
              pm[1] = fmaf(yq, pv.v[j] - pcen, pm[1]);
            }
          }
          row_store<T, kS>(as, dxo, rr - 1, rowelems, lane, bufS1, yrow);
          row_store<T, kS>(as, doo, rr - 1, rowelems, lane, bufS2, dsum);
        }
      };
      // steps rr = 0 .. H; after three steps every array is back in its starting role
      int rr = 0;
      for (; rr + 2 <= H; rr += 3) {
        step(rr,     xa, xb, xc, ua, ub, uc, d0, d1);
        step(rr + 1, xb, xc, xa, ub, uc, ua, d1, d2);
        step(rr + 2, xc, xa, xb, uc, ua, ub, d2, d0);
      }
      if (rr <= H) {
        step(rr, xa, xb, xc, ua, ub, uc, d0, d1);
        if (rr + 1 <= H) step(rr + 1, xb, xc, xa, ub, uc, ua, d1, d2);
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

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
int light_lean_supported(int B, int C, int H, int W, int dtype) {
  if (C % kWave || !(dtype == MRLA_BF16 || dtype == MRLA_F16)) return 0;
  // (these passes walk whole images: where the x_t-storing passes cut the ROWS as well -- a few very large images, RowCut in
  // light_nhwc_wide.h -- the partial records would not match; the stored form runs there)
  return (nhwc_mom_zranges(B, C, H, W).rows == 1 && nhwc_bmom_zranges(B, C, H, W).rows == 1 &&
          nhwc_wgrad_zranges(B, C, H, W).rows == 1) ? 1 : 0;
}

int launch_light_stats_bwd_lean_wide(const void* dout, const void* pre, const void* o, const float* wv, const float* psc,
                                     const float* psh, const float* mom, float* bmom, int B, int C, int H, int W, int dtype,
                                     hipStream_t st) {
  if (!light_lean_supported(B, C, H, W, dtype)) return MRLA_EUNSUPPORTED;
  const bool ragged = (W % kS) != 0;
#define CALL_K(T, AF, RG)                                                                                          \
  {                                                                                                                \
    const WideLaunch L = wide_launch(P_STATS_BWD, B, C, W, D_N, stats_bwd_lean_wave_bytes<T>(), 0, false, nhwc_bmom_zranges(B, C, H, W).strips); \
    if (set_lds_n(light_stats_bwd_lean_wide<T, AF, RG>, L.lds) != hipSuccess) return MRLA_EHIP;                     \
    hipLaunchKernelGGL((light_stats_bwd_lean_wide<T, AF, RG>), L.grid, L.block, L.lds, st, (const T*)dout,          \
                       (const T*)pre, (const T*)o, wv, psc, psh, mom, bmom, B, C, H, W, L.BG, L.wc);               \
  }
#define CALL_A(T, AF) { if (ragged) CALL_K(T, AF, true) else CALL_K(T, AF, false) }
#define CALL(T) { if (psc) CALL_A(T, true) else CALL_A(T, false) }
  if (dtype == MRLA_BF16) CALL(bf16_t) else CALL(f16_t)
#undef CALL
#undef CALL_A
#undef CALL_K
  if (hipGetLastError() != hipSuccess) return MRLA_EHIP;
  return launch_fold_rows(bmom, nhwc_bmom_zranges(B, C, H, W).strips, B * C * D_N, st);      // partial records -> record 0
}

int launch_light_apply_bwd_lean_wide(const void* dout, const void* pre, const void* o, const float* wv, const float* psc,
                                     const float* psh, const float* gate, const float* cb, const float* lam,
                                     const float* dp, const float* dyx, void* dx, void* dprev, float* dwv_part,
                                     const float* pre_center, float* pre_tmom, int B, int C, int H, int W, int d, int res,
                                     int dtype, hipStream_t st) {
  if (!light_lean_supported(B, C, H, W, dtype)) return MRLA_EUNSUPPORTED;
  const bool ragged = (W % kS) != 0;
  const int bg = nhwc_images_per_group(B, C, W);          // image groups x strip ranges = the rows mrla_light_wgrad_rows() promised
  const int nz = nhwc_wgrad_zranges(B, C, H, W).strips;
#define CALL_K(T, AF, RG, PR)                                                                                      \
  {                                                                                                                \
    const WideLaunch L = wide_launch(P_APPLY_BWD, B, C, W, 9, apply_bwd_lean_wave_bytes<T>(), bg, false, nz);      \
    if (set_lds_n(light_apply_bwd_lean_wide<T, AF, RG, PR>, L.lds) != hipSuccess) return MRLA_EHIP;                 \
    hipLaunchKernelGGL((light_apply_bwd_lean_wide<T, AF, RG, PR>), L.grid, L.block, L.lds, st, (const T*)dout,      \
                       (const T*)pre, (const T*)o, wv, psc, psh, gate, cb, lam, dp, dyx, (T*)dx, (T*)dprev,         \
                       dwv_part, pre_center, pre_tmom, B, C, H, W, L.BG, d, res, L.wc);                            \
  }
#define CALL_P(T, AF, RG) { if (pre_tmom) CALL_K(T, AF, RG, true) else CALL_K(T, AF, RG, false) }
#define CALL_A(T, AF) { if (ragged) CALL_P(T, AF, true) else CALL_P(T, AF, false) }
#define CALL(T) { if (psc) CALL_A(T, true) else CALL_A(T, false) }
  if (dtype == MRLA_BF16) CALL(bf16_t) else CALL(f16_t)
#undef CALL
#undef CALL_A
#undef CALL_P
#undef CALL_K
  return hip_status(hipGetLastError());
}

}  // namespace mrla
