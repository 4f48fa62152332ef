// MRLA-light backward apply pass for channels_last activations (see light_nhwc.hip for the layout and the passes).
// A translation unit of its own: this kernel is register-bound, and pairing its FMAs into v_pk_fma_f32 (the SLP
// vectoriser) costs it ~40 VGPRs and a move per pair, so the Makefile compiles this file with -fno-slp-vectorize;
// the lighter passes in light_nhwc.hip gain from the pairing and keep it.
#include "light_nhwc.h"

namespace mrla {

// ------------------------------------------------------------------------------------------------
// backward apply
// ------------------------------------------------------------------------------------------------
// Strip-local windows (columns relative to s0):  x rows rr-1..rr+1 over cols -2..kS+1 (kS+4 wide),
// dU rows rr-2..rr over cols -1..kS (kS+2 wide).  At step rr: U[rr] on cols -1..kS -> dU[rr]; then dx[rr-1] on the
// owned cols from dU rows rr-2..rr.
template <typename T, bool GELU, bool HAS_O, bool RELU, bool WIDE>
__global__ __launch_bounds__(kMaxStrips * kWave) void light_apply_bwd_nhwc(
    const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv,
    const float* __restrict__ gate, const float* __restrict__ cb, const float* __restrict__ lam,
    const float* __restrict__ dp, const float* __restrict__ dyx, T* __restrict__ dx, T* __restrict__ dprev,
    float* __restrict__ dwv_part, int B, int C, int H, int W, int BG, int d, int res) {
  MRLA_NHWC_PROLOGUE(9)
  const int G = C / d;
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[cc * 9 + k];
  const float e_ = cb ? cb[cc * 4 + 0] : 1.f, f_ = cb ? cb[cc * 4 + 1] : 0.f;
  const float Gc = cb ? cb[cc * 4 + 2] : 0.f, Hc = cb ? cb[cc * 4 + 3] : 0.f;
  const float lm = (HAS_O && lam) ? lam[cc] : 1.f;
  const float resf = res ? 1.f : 0.f;
  float wg[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int b_end = min(B, (int)(blockIdx.y + 1) * BG);
  for (int b = blockIdx.y * BG; b < b_end; ++b) {
    const size_t ioff = (size_t)b * H * W * C;
    const T* xi = x + ioff;
    const T* gi = dout + ioff;
    const T* oi = HAS_O ? o + ioff : nullptr;
    T* dxo = dx + ioff;
    T* doo = HAS_O ? dprev + ioff : nullptr;
    const float dpb = dp ? dp[b] : 1.f;
    const float a = gate[(size_t)b * G + cc / d];
    const float E = e_ * dpb, F = f_ * a;
    const float dy = dyx[(size_t)b * C + cc];
    for (int s = wave; s < nstrips; s += nwaves) {
      const int s0 = s * kS, nc = min(kS, W - s0);
      float xa[kS + 4], xb[kS + 4], xc[kS + 4];      // x rows rr-1, rr, rr+1
      float ua[kS + 2], ub[kS + 2], uc[kS + 2];      // dU rows rr-2, rr-1, rr
#pragma unroll
      for (int j = 0; j < kS + 4; ++j) xa[j] = 0.f;
      read_row<T, WIDE, kS + 4>(xi, 0, s0 - 2, H, W, C, cbase, cc, lane, SCR(0), xb);
#pragma unroll
      for (int j = 0; j < kS + 2; ++j) { ua[j] = 0.f; ub[j] = 0.f; }
      // software pipeline (WIDE): the row pieces of step rr+1 are in flight while step rr computes
      RowLoad<T, kS + 4> qx;
      RowLoad<T, kS + 2> qg, qo;
      RowAddr<T, kS + 4> ax;
      RowAddr<T, kS + 2> ag, ao;
      if (WIDE) {
        make_row_addr<T, kS + 4>(ax, s0 - 2, W, C, cbase, lane);
        make_row_addr<T, kS + 2>(ag, s0 - 1, W, C, cbase, lane);
        make_row_addr<T, kS + 2>(ao, s0 - 1, W, C, cbase, lane);
        issue_row<T, kS + 4>(qx, xi, 1, H, W * C, ax);
        issue_row<T, kS + 2>(qg, gi, 0, H, W * C, ag);
        if (HAS_O) issue_row<T, kS + 2>(qo, oi, 0, H, W * C, ao);
      }
      // One row step.  The window arrays rotate by NAME (XA/XB/XC, UA/UB/UC, G*/D* below), three steps per loop trip,
      // so no register copies are spent on shifting the windows.
      auto step = [&](int rr, float (&XA)[kS + 4], float (&XB)[kS + 4], float (&XC)[kS + 4], float (&UA)[kS + 2],
                      float (&UB)[kS + 2], float (&UC)[kS + 2], float (&GP)[kS], float (&GC)[kS], float (&DP)[kS],
                      float (&DC)[kS]) {
        float gv[kS + 2], ov[kS + 2];                // dOut / o of row rr on columns -1 .. kS (zero outside the image)
        if (WIDE) {
          finish_row<T, kS + 4>(qx, lane, SCR(0), XC);
          finish_row<T, kS + 2>(qg, lane, SCR(1), gv);
          if (HAS_O) finish_row<T, kS + 2>(qo, lane, SCR(2), ov);
          issue_row<T, kS + 4>(qx, xi, rr + 2, H, W * C, ax);
          issue_row<T, kS + 2>(qg, gi, rr + 1, H, W * C, ag);
          if (HAS_O) issue_row<T, kS + 2>(qo, oi, rr + 1, H, W * C, ao);
        } else {
          read_row<T, false, kS + 4>(xi, rr + 1, s0 - 2, H, W, C, cbase, cc, lane, SCR(0), XC);
          read_row<T, false, kS + 2>(gi, rr, s0 - 1, H, W, C, cbase, cc, lane, SCR(1), gv);
          if (HAS_O) read_row<T, false, kS + 2>(oi, rr, s0 - 1, H, W, C, cbase, cc, lane, SCR(2), ov);
        }
        float dorow[kS];
        // dU[rr] on columns -1 .. kS (zero outside the image; nothing to compute on the step past the last row)
        if (rr >= H) {
#pragma unroll
          for (int j = 0; j < kS + 2; ++j) UC[j] = 0.f;
#pragma unroll
          for (int j = 0; j < kS; ++j) { GC[j] = 0.f; DC[j] = 0.f; dorow[j] = 0.f; }
        } else
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) {
          const int col = s0 - 1 + j;
          const bool in = col >= 0 && col < W;                              // wave-uniform
          const float u = conv_at(w, XA, XB, XC, j);                        // window cols j..j+2 <-> image cols col-1..col+1
          const float v = GELU ? gelu_f(u) : u;
          float dm = fmaf(E, gv[j], Hc);
          dm = fmaf(F, v, dm);
          if (HAS_O) dm = fmaf(Gc, ov[j], dm);
          float du = a * dm;
          if (GELU) du *= gelu_grad_f(u);
          du = in ? du : 0.f;
          if (j >= 1 && j <= kS) {                                          // owned column (compile-time after unroll)
            GC[j - 1] = gv[j];
            DC[j - 1] = in ? lm * dm : 0.f;
            dorow[j - 1] = DC[j - 1];
            if (j - 1 < nc) {
              // dWv[i][k] += dU[rr][col] * x[rr+i-1][col+k-1]
              wg[0] = fmaf(du, XA[j], wg[0]); wg[1] = fmaf(du, XA[j + 1], wg[1]); wg[2] = fmaf(du, XA[j + 2], wg[2]);
              wg[3] = fmaf(du, XB[j], wg[3]); wg[4] = fmaf(du, XB[j + 1], wg[4]); wg[5] = fmaf(du, XB[j + 2], wg[5]);
              wg[6] = fmaf(du, XC[j], wg[6]); wg[7] = fmaf(du, XC[j + 1], wg[7]); wg[8] = fmaf(du, XC[j + 2], wg[8]);
            }
          }
          UC[j] = du;
        }
        if (HAS_O && !RELU && rr < H) write_row<T, WIDE, kS>(doo, rr, s0, nc, W, C, cbase, c, cv, lane, SCR(4), dorow);
        // dx[rr-1] on the owned columns:  dx[ro][col] = sum_{i,k} w[i][k] * dU[ro-i+1][col-k+1]
        if (rr >= 1) {
          const int ro = rr - 1;
          float yrow[kS], dsum[kS];
#pragma unroll
          for (int j = 0; j < kS; ++j) {
            // window index of column (col + 1 - k) in the dU arrays (which start at col -1): j + 2 - k
            float s9 = w[0] * UC[j + 2];
            s9 = fmaf(w[1], UC[j + 1], s9); s9 = fmaf(w[2], UC[j], s9);
            s9 = fmaf(w[3], UB[j + 2], s9); s9 = fmaf(w[4], UB[j + 1], s9); s9 = fmaf(w[5], UB[j], s9);
            s9 = fmaf(w[6], UA[j + 2], s9); s9 = fmaf(w[7], UA[j + 1], s9); s9 = fmaf(w[8], UA[j], s9);
            float y = fmaf(resf, GP[j], s9 + dy);
            if (RELU) y = (XA[j + 2] > 0.f) ? y : 0.f;                      // XA = x[rr-1] = x[ro]; owned col j <-> window j+2
            yrow[j] = y;
            dsum[j] = DP[j] + y;
          }
          write_row<T, WIDE, kS>(dxo, ro, s0, nc, W, C, cbase, c, cv, lane, SCR(3), yrow);
          if (RELU && HAS_O) write_row<T, WIDE, kS>(doo, ro, s0, nc, W, C, cbase, c, cv, lane, SCR(4), dsum);
        }
      };
      float g0[kS], g1[kS], g2[kS], d0[kS], d1[kS], d2[kS];
#pragma unroll
      for (int j = 0; j < kS; ++j) { g0[j] = 0.f; d0[j] = 0.f; }
      // steps rr = 0 .. H; after three steps every array is back in its starting role
      int rr = 0;
      for (; rr + 2 <= H; rr += 3) {
        step(rr,     xa, xb, xc, ua, ub, uc, g0, g1, d0, d1);
        step(rr + 1, xb, xc, xa, ub, uc, ua, g1, g2, d1, d2);
        step(rr + 2, xc, xa, xb, uc, ua, ub, g2, g0, d2, d0);
      }
      if (rr <= H) {
        step(rr, xa, xb, xc, ua, ub, uc, g0, g1, d0, d1);
        if (rr + 1 <= H) step(rr + 1, xb, xc, xa, ub, uc, ua, g1, g2, d1, d2);
      }
    }
  }
  wg_reduce<9>(wg, red, lane, wave, nwaves);
  if (wave == 0 && cv) {
#pragma unroll
    for (int k = 0; k < 9; ++k) dwv_part[((size_t)blockIdx.y * C + c) * 9 + k] = wg[k];
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// (nhwc_images_per_group(): light_nhwc_wide.hip, next to the launch geometry it belongs to)

int launch_light_apply_bwd_nhwc(const void* dout, const void* x, const void* o, const float* wv, const float* gate,
                                const float* cb, const float* lam, const float* dp, const float* dyx, void* dx,
                                void* dprev, float* dwv_part, const void* pre, const float* pre_center, float* pre_tmom,
                                int B, int C, int H, int W, int d, int res, int relu, int dtype, int act, hipStream_t st) {
  NhwcLaunch L = nhwc_launch(B, C, W, 9, dtype);
  if (L.wide)          // C % 64 == 0: the LDS-DMA row pipeline (light_nhwc_wide.hip)
    return launch_light_apply_bwd_wide(dout, x, o, wv, gate, cb, lam, dp, dyx, dx, dprev, dwv_part, pre, pre_center, pre_tmom,
                                       B, C, H, W, d, res, relu, dtype, act, st);
  if (pre_tmom) return MRLA_EUNSUPPORTED;       // (the deferred-BatchNorm sums exist on the row pipeline only)
  L.BG = nhwc_images_per_group(B, C, W);                  // = the rows mrla_light_wgrad_rows() promised
  L.grid = dim3(L.grid.x, (B + L.BG - 1) / L.BG);
#define CALL_W(T, A, O, R, WD)                                                                                       \
  {                                                                                                                  \
    if (set_lds_n(light_apply_bwd_nhwc<T, A, O, R, WD>, L.lds) != hipSuccess) return MRLA_EHIP;                        \
    hipLaunchKernelGGL((light_apply_bwd_nhwc<T, A, O, R, WD>), L.grid, L.block, L.lds, st, (const T*)dout,            \
                       (const T*)x, (const T*)o, wv, gate, cb, lam, dp, dyx, (T*)dx, (T*)dprev, dwv_part, B, C, H, W, \
                       L.BG, d, res);                                                                                \
  }
#define CALL_R(T, A, O, R) CALL_W(T, A, O, R, false)
#define CALL(T, A, O)                                                                        \
  {                                                                                          \
    if (relu) { if (O && !(A)) CALL_R(T, false, true, true) else return MRLA_EINVAL; }       \
    else CALL_R(T, A, O, false)                                                              \
  }
  MRLA_DISPATCH_T_N(dtype, act, o != nullptr, CALL)
#undef CALL
#undef CALL_R
#undef CALL_W
  return hip_status(hipGetLastError());
}

}  // namespace mrla
