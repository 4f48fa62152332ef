// MRLA-light streaming kernels for channels_last (NHWC) activations: x[b, h, w, c] with c contiguous (gfx950).
//
// Same four passes and the same math as light_nchw.hip (stats_fwd / apply_fwd / stats_bwd / apply_bwd; reference:
// resnet/models/modules/mrla_light_module.py:52-74, resnet/models/resnet_mrla_light.py:40-43,113-116), but the
// channel axis is the contiguous one, so the natural mapping is LANE = CHANNEL:
//   * a wave owns 64 consecutive channels and a strip of 7 image columns, and walks down the rows;
//   * every 3x3 neighbour of a pixel lives in the SAME lane (another pixel of the same channel), so the stencil
//     is plain register arithmetic on a rolling row window -- no LDS tile, no DPP, no edge masks, all 64 lanes busy;
//   * a pixel access is 64 lanes x 2 B = one 128-byte line, addressed scalar-base + lane;
//   * per-(image, channel) sums are per-lane accumulators: the only reduction is over the <= 8 strip-waves of a
//     workgroup through LDS, in a fixed order (bitwise reproducible).
// A workgroup = (image group, 64-channel chunk); its waves = column strips.  ResNet stage widths 56/28/14/7 give
// 8/4/2/1 strips of exactly 7 columns.
#include "light_nhwc.h"

namespace mrla {

// ------------------------------------------------------------------------------------------------
// forward statistics (+ optional fused producer x = relu(pre + o), or x = relu((psc*pre + psh) + o) when the
// per-channel affine of the BatchNorm in front (bn3) is handed over instead of being applied in a pass of its own)
// ------------------------------------------------------------------------------------------------
template <typename T, bool GELU, bool HAS_O, bool FUSE, bool WIDE>
__global__ __launch_bounds__(kMaxStrips * kWave) void light_stats_fwd_nhwc(
    const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv, float* __restrict__ mom,
    T* __restrict__ xout, const float* __restrict__ psc, const float* __restrict__ psh, T* __restrict__ vout, int B,
    int C, int H, int W, int BG) {
  MRLA_NHWC_PROLOGUE(M_N)
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[cc * 9 + k];
  const bool aff = FUSE && psc != nullptr;
  const float asc = aff ? psc[cc] : 1.f, ash = aff ? psh[cc] : 0.f;
  const int b_end = min(B, (int)(blockIdx.y + 1) * BG);
  for (int b = blockIdx.y * BG; b < b_end; ++b) {
    const size_t ioff = (size_t)b * H * W * C;
    const T* xi = x + ioff;
    const T* oi = HAS_O ? o + ioff : nullptr;
    T* xo = FUSE ? xout + ioff : nullptr;
    T* vo = vout ? vout + ioff : nullptr;          // MRLA-base: V = dwconv3x3(x) is kept (the stage's value history)
    float acc[M_N] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = wave; s < nstrips; s += nwaves) {
      const int s0 = s * kS, nc = min(kS, W - s0);
      float ra[kS + 2], rb[kS + 2], rc[kS + 2];        // x rows r-1, r, r+1 over columns s0-1 .. s0+kS
      float ob[kS + 2], oc[kS + 2];                    // o rows r, r+1 (same columns)
      // row pieces of the NEXT load_row() call are already in flight (WIDE): software pipeline of depth one row
      RowLoad<T, kS + 2> qx, qo;
      RowAddr<T, kS + 2> ax, ao;
      if (WIDE) {
        make_row_addr<T, kS + 2>(ax, s0 - 1, W, C, cbase, lane);
        make_row_addr<T, kS + 2>(ao, s0 - 1, W, C, cbase, lane);
        issue_row<T, kS + 2>(qx, xi, 0, H, W * C, ax);
        if (HAS_O) issue_row<T, kS + 2>(qo, oi, 0, H, W * C, ao);
      }
      // loads row r into (dst, odst); FUSE forms x = relu(pre + o) and stores the owned pixels to xout
      auto load_row = [&](int r, float (&dst)[kS + 2], float (&odst)[kS + 2]) {
        if (WIDE) {
          finish_row<T, kS + 2>(qx, lane, SCR(0), dst);
          if (HAS_O) finish_row<T, kS + 2>(qo, lane, SCR(1), odst);
          issue_row<T, kS + 2>(qx, xi, r + 1, H, W * C, ax);
          if (HAS_O) issue_row<T, kS + 2>(qo, oi, r + 1, H, W * C, ao);
        } else {
          read_row<T, false, kS + 2>(xi, r, s0 - 1, H, W, C, cbase, cc, lane, SCR(0), dst);
          if (HAS_O) read_row<T, false, kS + 2>(oi, r, s0 - 1, H, W, C, cbase, cc, lane, SCR(1), odst);
        }
        if (FUSE) {
          if (aff) {      // BatchNorm affine of the pre-activation, rounded to T as the stand-alone pass stores it;
                          // pixels outside the image must stay zero (the padding of the 3x3 taps)
            const bool rowok = r >= 0 && r < H;
            const int lim = W - s0;
#pragma unroll
            for (int j = 0; j < kS + 2; ++j) {
              const bool ok = rowok && (j == 0 ? s0 > 0 : j <= lim);
              dst[j] = ok ? to_f(from_f<T>(fmaf(asc, dst[j], ash))) : 0.f;
            }
          }
#pragma unroll
          for (int j = 0; j < kS + 2; ++j) dst[j] = fmaxf(to_f(from_f<T>(dst[j] + odst[j])), 0.f);   // zeros stay zeros
          if (r >= 0 && r < H) {
            float own[kS];
#pragma unroll
            for (int j = 0; j < kS; ++j) own[j] = dst[j + 1];
            write_row<T, WIDE, kS>(xo, r, s0, nc, W, C, cbase, c, cv, lane, SCR(3), own);
          }
        }
      };
#pragma unroll
      for (int j = 0; j < kS + 2; ++j) ra[j] = 0.f;
      load_row(0, rb, ob);
      for (int r = 0; r < H; ++r) {
        load_row(r + 1, rc, oc);
        float vrow[kS];
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          vrow[j] = 0.f;
          if (j < nc) {
            float v = conv_at(w, ra, rb, rc, j);
            if (GELU) v = gelu_f(v);
            vrow[j] = v;
            acc[M_SX] += rb[j + 1];
            acc[M_SV] += v;
            acc[M_SVV] = fmaf(v, v, acc[M_SVV]);
            if (HAS_O) {
              const float ov = ob[j + 1];
              acc[M_SO] += ov;
              acc[M_SVO] = fmaf(v, ov, acc[M_SVO]);
              acc[M_SOO] = fmaf(ov, ov, acc[M_SOO]);
            }
          }
        }
        if (vo) write_row<T, WIDE, kS>(vo, r, s0, nc, W, C, cbase, c, cv, lane, SCR(3), vrow);
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) { ra[j] = rb[j]; rb[j] = rc[j]; ob[j] = oc[j]; }
      }
    }
    wg_reduce<M_N>(acc, red, lane, wave, nwaves);
    if (wave == 0 && cv) {
#pragma unroll
      for (int k = 0; k < M_N; ++k) mom[((size_t)b * C + c) * M_REC + k] = acc[k];
      mom[((size_t)b * C + c) * M_REC + M_PV] = 0.f;          // raw sums (no pivots on this path)
      mom[((size_t)b * C + c) * M_REC + M_PO] = 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward apply:  out = res*x + A*V + B*o + C
// ------------------------------------------------------------------------------------------------
template <typename T, bool GELU, bool HAS_O, bool WIDE>
__global__ __launch_bounds__(kMaxStrips * kWave) void light_apply_fwd_nhwc(
    const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv, const float* __restrict__ gate,
    const float* __restrict__ sc, const float* __restrict__ sh, const float* __restrict__ lam,
    const float* __restrict__ dp, T* __restrict__ out, int B, int C, int H, int W, int BG, int d, int res) {
  MRLA_NHWC_PROLOGUE(0)
  const int G = C / d;
  float w0[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w0[k] = wv[cc * 9 + k];
  const float scc = sc ? sc[cc] : 1.f, shc = sh ? sh[cc] : 0.f, lmc = (HAS_O && lam) ? lam[cc] : 0.f;
  const float resf = res ? 1.f : 0.f;
  const int b_end = min(B, (int)(blockIdx.y + 1) * BG);
  for (int b = blockIdx.y * BG; b < b_end; ++b) {
    const size_t ioff = (size_t)b * H * W * C;
    const T* xi = x + ioff;
    const T* oi = HAS_O ? o + ioff : nullptr;
    T* yo = out + ioff;
    const float dpb = dp ? dp[b] : 1.f;
    const float scale = dpb * scc;
    const float A = scale * gate[(size_t)b * G + cc / d];
    const float Bc = scale * lmc, Cc = dpb * shc;
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = GELU ? w0[k] : w0[k] * A;       // fold gate / BN scale into the taps
    if (!GELU) w[4] += resf;                                           // ... and the residual into the centre tap
    for (int s = wave; s < nstrips; s += nwaves) {
      const int s0 = s * kS, nc = min(kS, W - s0);
      float ra[kS + 2], rb[kS + 2], rc[kS + 2];
#pragma unroll
      for (int j = 0; j < kS + 2; ++j) ra[j] = 0.f;
      read_row<T, WIDE, kS + 2>(xi, 0, s0 - 1, H, W, C, cbase, cc, lane, SCR(0), rb);
      RowLoad<T, kS + 2> qx;
      RowLoad<T, kS> qo;
      RowAddr<T, kS + 2> ax;
      RowAddr<T, kS> ao;
      if (WIDE) {
        make_row_addr<T, kS + 2>(ax, s0 - 1, W, C, cbase, lane);
        make_row_addr<T, kS>(ao, s0, W, C, cbase, lane);
        issue_row<T, kS + 2>(qx, xi, 1, H, W * C, ax);
        if (HAS_O) issue_row<T, kS>(qo, oi, 0, H, W * C, ao);
      }
      for (int r = 0; r < H; ++r) {
        float ov[kS], y[kS];
        if (WIDE) {
          finish_row<T, kS + 2>(qx, lane, SCR(0), rc);
          if (HAS_O) finish_row<T, kS>(qo, lane, SCR(1), ov);
          issue_row<T, kS + 2>(qx, xi, r + 2, H, W * C, ax);
          if (HAS_O) issue_row<T, kS>(qo, oi, r + 1, H, W * C, ao);
        } else {
          read_row<T, false, kS + 2>(xi, r + 1, s0 - 1, H, W, C, cbase, cc, lane, SCR(0), rc);
          if (HAS_O) read_row<T, false, kS>(oi, r, s0, H, W, C, cbase, cc, lane, SCR(1), ov);
        }
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          if (GELU) y[j] = fmaf(A, gelu_f(conv_at(w, ra, rb, rc, j)), fmaf(resf, rb[j + 1], Cc));
          else      y[j] = conv_at(w, ra, rb, rc, j) + Cc;
          if (HAS_O) y[j] = fmaf(Bc, ov[j], y[j]);
        }
        write_row<T, WIDE, kS>(yo, r, s0, nc, W, C, cbase, c, cv, lane, SCR(3), y);
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) { ra[j] = rb[j]; rb[j] = rc[j]; }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// inference form of the apply pass: x_t = relu(round(psc*pre + psh) + o) is formed on the fly from conv3's output and the
// shortcut (it is neither needed again nor saved when no gradient is asked for), so the block tail moves 5N instead of 6N
// elements:  pooling pass (bnact_nhwc.hip, reads pre and o) + this pass (reads pre and o, writes out).
// ------------------------------------------------------------------------------------------------
template <typename T, bool WIDE>
__global__ __launch_bounds__(kMaxStrips * kWave) void light_apply_fwd_pre_nhwc(
    const T* __restrict__ pre, const T* __restrict__ o, const float* __restrict__ psc, const float* __restrict__ psh,
    const float* __restrict__ wv, const float* __restrict__ gate, const float* __restrict__ sc,
    const float* __restrict__ sh, const float* __restrict__ lam, const float* __restrict__ dp, T* __restrict__ out, int B,
    int C, int H, int W, int BG, int d, int res) {
  MRLA_NHWC_PROLOGUE(0)
  const int G = C / d;
  float w0[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w0[k] = wv[cc * 9 + k];
  const float scc = sc ? sc[cc] : 1.f, shc = sh ? sh[cc] : 0.f, lmc = lam ? lam[cc] : 0.f;
  const float resf = res ? 1.f : 0.f;
  const bool aff = psc != nullptr;
  const float asc = aff ? psc[cc] : 1.f, ash = aff ? psh[cc] : 0.f;
  const int b_end = min(B, (int)(blockIdx.y + 1) * BG);
  for (int b = blockIdx.y * BG; b < b_end; ++b) {
    const size_t ioff = (size_t)b * H * W * C;
    const T* xi = pre + ioff;
    const T* oi = o + ioff;
    T* yo = out + ioff;
    const float dpb = dp ? dp[b] : 1.f;
    const float scale = dpb * scc;
    const float A = scale * gate[(size_t)b * G + cc / d];
    const float Bc = scale * lmc, Cc = dpb * shc;
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = w0[k] * A;
    w[4] += resf;
    for (int s = wave; s < nstrips; s += nwaves) {
      const int s0 = s * kS, nc = min(kS, W - s0);
      const int lim = W - s0;
      float ra[kS + 2], rb[kS + 2], rc[kS + 2];          // x_t rows r-1, r, r+1 on columns s0-1 .. s0+kS
      float ob[kS + 2], oc[kS + 2];                      // o rows r, r+1
      RowLoad<T, kS + 2> qx, qo;
      RowAddr<T, kS + 2> ax;
      if (WIDE) {
        make_row_addr<T, kS + 2>(ax, s0 - 1, W, C, cbase, lane);
        issue_row<T, kS + 2>(qx, xi, 0, H, W * C, ax);
        issue_row<T, kS + 2>(qo, oi, 0, H, W * C, ax);
      }
      auto form_row = [&](int r, float (&dst)[kS + 2], float (&odst)[kS + 2]) {
        if (WIDE) {
          finish_row<T, kS + 2>(qx, lane, SCR(0), dst);
          finish_row<T, kS + 2>(qo, lane, SCR(1), odst);
          issue_row<T, kS + 2>(qx, xi, r + 1, H, W * C, ax);
          issue_row<T, kS + 2>(qo, oi, r + 1, H, W * C, ax);
        } else {
          read_row<T, false, kS + 2>(xi, r, s0 - 1, H, W, C, cbase, cc, lane, SCR(0), dst);
          read_row<T, false, kS + 2>(oi, r, s0 - 1, H, W, C, cbase, cc, lane, SCR(1), odst);
        }
        if (aff) {
          const bool rowok = r >= 0 && r < H;
#pragma unroll
          for (int j = 0; j < kS + 2; ++j) {
            const bool ok = rowok && (j == 0 ? s0 > 0 : j <= lim);
            dst[j] = ok ? to_f(from_f<T>(fmaf(asc, dst[j], ash))) : 0.f;
          }
        }
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) dst[j] = fmaxf(to_f(from_f<T>(dst[j] + odst[j])), 0.f);
      };
#pragma unroll
      for (int j = 0; j < kS + 2; ++j) ra[j] = 0.f;
      form_row(0, rb, ob);
      for (int r = 0; r < H; ++r) {
        form_row(r + 1, rc, oc);
        float y[kS];
#pragma unroll
        for (int j = 0; j < kS; ++j) y[j] = fmaf(Bc, ob[j + 1], conv_at(w, ra, rb, rc, j) + Cc);
        write_row<T, WIDE, kS>(yo, r, s0, nc, W, C, cbase, c, cv, lane, SCR(3), y);
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) { ra[j] = rb[j]; rb[j] = rc[j]; ob[j] = oc[j]; }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward statistics
// ------------------------------------------------------------------------------------------------
template <typename T, bool GELU, bool HAS_O, bool WIDE>
__global__ __launch_bounds__(kMaxStrips * kWave) void light_stats_bwd_nhwc(
    const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv,
    float* __restrict__ bmom, int B, int C, int H, int W, int BG) {
  MRLA_NHWC_PROLOGUE(D_N)
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[cc * 9 + k];
  const int b_end = min(B, (int)(blockIdx.y + 1) * BG);
  for (int b = blockIdx.y * BG; b < b_end; ++b) {
    const size_t ioff = (size_t)b * H * W * C;
    const T* xi = x + ioff;
    const T* gi = dout + ioff;
    const T* oi = HAS_O ? o + ioff : nullptr;
    float acc[D_N] = {0.f, 0.f, 0.f};
    for (int s = wave; s < nstrips; s += nwaves) {
      const int s0 = s * kS, nc = min(kS, W - s0);
      float ra[kS + 2], rb[kS + 2], rc[kS + 2];
#pragma unroll
      for (int j = 0; j < kS + 2; ++j) ra[j] = 0.f;
      read_row<T, WIDE, kS + 2>(xi, 0, s0 - 1, H, W, C, cbase, cc, lane, SCR(0), rb);
      RowLoad<T, kS + 2> qx;
      RowLoad<T, kS> qg, qo;
      RowAddr<T, kS + 2> ax;
      RowAddr<T, kS> ag, ao;
      if (WIDE) {
        make_row_addr<T, kS + 2>(ax, s0 - 1, W, C, cbase, lane);
        make_row_addr<T, kS>(ag, s0, W, C, cbase, lane);
        make_row_addr<T, kS>(ao, s0, W, C, cbase, lane);
        issue_row<T, kS + 2>(qx, xi, 1, H, W * C, ax);
        issue_row<T, kS>(qg, gi, 0, H, W * C, ag);
        if (HAS_O) issue_row<T, kS>(qo, oi, 0, H, W * C, ao);
      }
      for (int r = 0; r < H; ++r) {
        float gv[kS], ov[kS];
        if (WIDE) {
          finish_row<T, kS + 2>(qx, lane, SCR(0), rc);
          finish_row<T, kS>(qg, lane, SCR(1), gv);
          if (HAS_O) finish_row<T, kS>(qo, lane, SCR(2), ov);
          issue_row<T, kS + 2>(qx, xi, r + 2, H, W * C, ax);
          issue_row<T, kS>(qg, gi, r + 1, H, W * C, ag);
          if (HAS_O) issue_row<T, kS>(qo, oi, r + 1, H, W * C, ao);
        } else {
          read_row<T, false, kS + 2>(xi, r + 1, s0 - 1, H, W, C, cbase, cc, lane, SCR(0), rc);
          read_row<T, false, kS>(gi, r, s0, H, W, C, cbase, cc, lane, SCR(1), gv);
          if (HAS_O) read_row<T, false, kS>(oi, r, s0, H, W, C, cbase, cc, lane, SCR(2), ov);
        }
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          if (j < nc) {
            float v = conv_at(w, ra, rb, rc, j);
            if (GELU) v = gelu_f(v);
            acc[D_D] += gv[j];
            acc[D_DV] = fmaf(gv[j], v, acc[D_DV]);
            if (HAS_O) acc[D_DO] = fmaf(gv[j], ov[j], acc[D_DO]);
          }
        }
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) { ra[j] = rb[j]; rb[j] = rc[j]; }
      }
    }
    wg_reduce<D_N>(acc, red, lane, wave, nwaves);
    if (wave == 0 && cv) {
#pragma unroll
      for (int k = 0; k < D_N; ++k) bmom[((size_t)b * C + c) * D_N + k] = acc[k];
    }
  }
}

int launch_light_stats_fwd_nhwc(const void* x, const void* o, const float* wv, float* mom, void* xout,
                                const float* psc, const float* psh, void* vout, int B, int C, int H, int W, int dtype,
                                int act, hipStream_t st, bool fused_no_x, int mom_ranges) {
  const NhwcLaunch L = nhwc_launch(B, C, W, M_N, dtype);
  if (L.wide)          // C % 64 == 0: the LDS-DMA row pipeline (light_nhwc_wide.hip)
    return launch_light_stats_fwd_wide(x, o, wv, mom, xout, psc, psh, vout, B, C, H, W, dtype, act, st, fused_no_x, mom_ranges);
  if (fused_no_x) return MRLA_EUNSUPPORTED;       // (x_t stays unmaterialised on the row pipeline only)
#define CALL_W(T, A, O, F, WD)                                                                                       \
  {                                                                                                                  \
    if (set_lds_n(light_stats_fwd_nhwc<T, A, O, F, WD>, L.lds) != hipSuccess) return MRLA_EHIP;                        \
    hipLaunchKernelGGL((light_stats_fwd_nhwc<T, A, O, F, WD>), L.grid, L.block, L.lds, st, (const T*)x, (const T*)o,  \
                       wv, mom, (T*)xout, psc, psh, (T*)vout, B, C, H, W, L.BG);                                     \
  }
#define CALL_F(T, A, O, F) CALL_W(T, A, O, F, false)
#define CALL(T, A, O)                                                        \
  {                                                                          \
    if (xout) { if (O) CALL_F(T, A, true, true) else return MRLA_EINVAL; }   \
    else CALL_F(T, A, O, false)                                              \
  }
  MRLA_DISPATCH_T_N(dtype, act, o != nullptr, CALL)
#undef CALL
#undef CALL_F
#undef CALL_W
  return hip_status(hipGetLastError());
}

int launch_light_apply_fwd_nhwc(const void* x, const void* o, const float* wv, const float* gate, const float* sc,
                                const float* sh, const float* lam, const float* dp, void* out, int B, int C, int H,
                                int W, int d, int res, int dtype, int act, hipStream_t st) {
  const NhwcLaunch L = nhwc_launch(B, C, W, 0, dtype);
  if (L.wide)          // C % 64 == 0: the LDS-DMA row pipeline (light_nhwc_wide.hip)
    return launch_light_apply_fwd_wide(x, o, wv, gate, sc, sh, lam, dp, out, B, C, H, W, d, res, dtype, act, st);
#define CALL_W(T, A, O, WD)                                                                                          \
  {                                                                                                                  \
    if (set_lds_n(light_apply_fwd_nhwc<T, A, O, WD>, L.lds) != hipSuccess) return MRLA_EHIP;                           \
    hipLaunchKernelGGL((light_apply_fwd_nhwc<T, A, O, WD>), L.grid, L.block, L.lds, st, (const T*)x, (const T*)o, wv, \
                       gate, sc, sh, lam, dp, (T*)out, B, C, H, W, L.BG, d, res);                                    \
  }
#define CALL(T, A, O) CALL_W(T, A, O, false)
  MRLA_DISPATCH_T_N(dtype, act, o != nullptr, CALL)
#undef CALL
#undef CALL_W
  return hip_status(hipGetLastError());
}

int launch_light_apply_fwd_pre_nhwc(const void* pre, const void* o, const float* psc, const float* psh, const float* wv,
                                    const float* gate, const float* sc, const float* sh, const float* lam,
                                    const float* dp, void* out, int B, int C, int H, int W, int d, int res, int dtype,
                                    hipStream_t st) {
  const NhwcLaunch L = nhwc_launch(B, C, W, 0, dtype);
  if (L.wide)          // C % 64 == 0: the LDS-DMA row pipeline (light_nhwc_wide.hip)
    return launch_light_apply_fwd_pre_wide(pre, o, psc, psh, wv, gate, sc, sh, lam, dp, out, B, C, H, W, d, res, dtype, st);
#define CALL_W(T, WD)                                                                                                 \
  {                                                                                                                   \
    if (set_lds_n(light_apply_fwd_pre_nhwc<T, WD>, L.lds) != hipSuccess) return MRLA_EHIP;                              \
    hipLaunchKernelGGL((light_apply_fwd_pre_nhwc<T, WD>), L.grid, L.block, L.lds, st, (const T*)pre, (const T*)o, psc, \
                       psh, wv, gate, sc, sh, lam, dp, (T*)out, B, C, H, W, L.BG, d, res);                            \
  }
#define CALL(T) CALL_W(T, false)
  switch (dtype) {
    case MRLA_F32:  CALL(float) break;
    case MRLA_BF16: CALL(bf16_t) break;
    case MRLA_F16:  CALL(f16_t) break;
    default: return MRLA_EINVAL;
  }
#undef CALL
#undef CALL_W
  return hip_status(hipGetLastError());
}

int launch_light_stats_bwd_nhwc(const void* dout, const void* x, const void* o, const float* wv, const float* mom,
                                float* bmom, int B, int C, int H, int W, int dtype, int act, hipStream_t st) {
  const NhwcLaunch L = nhwc_launch(B, C, W, D_N, dtype);
  if (L.wide) return launch_light_stats_bwd_wide(dout, x, o, wv, mom, bmom, B, C, H, W, dtype, act, st);
  // (this path's forward statistics carry zero pivots -- light_stats_fwd_nhwc -- so its raw sums ARE the shifted ones)
#define CALL_W(T, A, O, WD)                                                                                          \
  {                                                                                                                  \
    if (set_lds_n(light_stats_bwd_nhwc<T, A, O, WD>, L.lds) != hipSuccess) return MRLA_EHIP;                           \
    hipLaunchKernelGGL((light_stats_bwd_nhwc<T, A, O, WD>), L.grid, L.block, L.lds, st, (const T*)dout, (const T*)x,  \
                       (const T*)o, wv, bmom, B, C, H, W, L.BG);                                                     \
  }
#define CALL(T, A, O) CALL_W(T, A, O, false)
  MRLA_DISPATCH_T_N(dtype, act, o != nullptr, CALL)
#undef CALL
#undef CALL_W
  return hip_status(hipGetLastError());
}

}  // namespace mrla
