// MRLA-light on token sequences, map kernels in the row-marching LANE = CHANNEL form of light_nhwc.hip.
//
// The map tokens x[b, 1 + r*side + col, c] ARE a channels_last image (pixel pitch C, row pitch side*C, image pitch n*C,
// first pixel at offset C), so the strips / rolling row windows / 16-byte row gathers of light_nhwc.h apply unchanged;
// what is token-specific is the LayerNorm applied to every fetched pixel (its (mean, rstd) are wave-uniform and come
// through the scalar cache) and the exact-GELU on V.  Used when C % 64 == 0 (DeiT widths 192 / 384 / 768); tokens.hip
// keeps the LDS-tile kernels for other widths, the LayerNorm statistics / pooling and the LayerNorm backward.
//   forward : out[b,i>=1] = res*x + a*gelu(dwconv3x3(LN_x(x) map)) + lam*LN_o(o_prev)      (the cls row: token_cls_fwd)
//   backward: ONE pass: bmom = sum dOut*gelu(U), dxn' = dwconv3x3^T(a*dOut*gelu'(U)) (fp32) and 15 partials / channel; the
//             pooled-descriptor gradient dy (known only after the gate backward has seen bmom) is a per-(image, channel)
//             constant on the map rows: mrla_token_gate_bwd folds it into the partials, mrla_token_ln_bwd into dxn
// Reference: deit/deit_mrla_light.py:157-180,194-209,234.
#include <algorithm>

#include "light_nhwc.h"
#include "nhwc_rows.h"

namespace mrla {

enum { TS_MX = 0, TS_RX = 1, TS_MO = 2, TS_RO = 3, TS_N = 4 };
enum { TQ_WV = 0, TQ_LAM = 9, TQ_LNXW = 10, TQ_LNXB = 11, TQ_LNOW = 12, TQ_LNOB = 13, TQ_H = 14, TQ_N = 15 };
static_assert(TQ_N == kTokParts && TQ_LNXW == kTokPartLnxW && TQ_LNXB == kTokPartLnxB && TQ_H == kTokPartHat,
              "the gate backward (gate.hip) patches these slots");

// cls row: out[b, 0, :] = res*x + LN_x(x)   (one thread per channel; grid (ceil(C/256), b)) -- MRLA-base on tokens, whose
// map rows come from a flat kernel
template <typename T>
__global__ __launch_bounds__(kThreads) void token_cls_fwd_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                                 const float* __restrict__ wx, const float* __restrict__ bx,
                                                                 T* __restrict__ out, int n, int C, int res) {
  const int b = blockIdx.y, c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= C) return;
  const size_t g = (size_t)b * n * C + c;
  const float* s = stats + (size_t)b * n * TS_N;
  const float xv = to_f(x[g]);
  float y = fmaf((xv - s[TS_MX]) * s[TS_RX], wx[c], bx[c]);
  if (res) y += xv;
  out[g] = from_f<T>(y);
}

// ------------------------------------------------------------------------------------------------
// backward apply: dxn' (fp32, map rows and the cls row), bmom and the 15 parameter partials per (image, channel)
//   dU[i]   = a * dOut[i] * gelu'(U[i])            (zero outside the map)
//   dxn'[i] = sum_{di,dj} wv[di][dj] * dU[i - (di,dj)]          (the caller's later passes add dy, see the file header)
//   bmom    = sum_i dOut[i] * gelu(U[i]);  partial TQ_H = sum_i xhat[i] (what dy multiplies in the LN_x weight gradient)
// Windows per strip: xn rows rr-1..rr+1 on columns s0-2..s0+kS+1; dU rows rr-2..rr on columns s0-1..s0+kS.
// On the LDS-DMA row pipeline of nhwc_rows.h (as the ResNet kernels of light_nhwc_wide.hip): rows arrive by
// `buffer_load ... lds` with everything outside the map as zeros from the bounds check, the three row windows rotate by
// NAME (three steps per trip, no register copies), and nothing in the arithmetic is predicated -- dOut is zero outside the
// map and in the two extra steps, so every gradient term vanishes there by itself; LayerNorm of a pixel outside the map is
// switched off through its wave-uniform (rstd, bias) pair instead of a branch.  (The first form of this kernel --
// register-staged row gathers, windows rotated by copies, exec-mask branches -- spent ~3 000 instructions per row step, 450
// of them register moves and 160 SGPR-spill lane moves: profiles/r03_notes.md section 8; this one ~1 150.)
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// LayerNorm statistics of a window row through LDS (round 4).  The (mean, rstd) pairs of the window's tokens are
// wave-uniform; fetched one by one through the scalar cache (18 dependent `s_load_dwordx2` + `s_waitcnt` per
// row step) they were what the backward kernel's waves waited for -- SQ counters: 34 % of its wave cycles in waits that a
// second row of DMA look-ahead did not move (profiles/r04_notes.md).  Here the window's 11 records [mean_x, rstd_x, mean_o,
// rstd_o] travel with the rows: ONE `buffer_load_dword ... lds` per step (lane l -> float l of the window's records;
// tokens outside the map arrive as zeros from the bounds check: rstd = 0 switches their LayerNorm off), read back as
// broadcast `ds_read_b64` in the same issue / fence block as the row reads.  A row's records serve twice: LN_x of window
// row rr+1 in one step, LN_o of the owned pixels of row rr in the next (two 256-byte buffers per wave, parity = row & 1).
// ------------------------------------------------------------------------------------------------
constexpr int kStatBufBytes = 256;
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N> struct StatRow { f32x2 v[N]; };                  // (mean, rstd) per token

// this lane's byte offset inside a row of stats records for window start column col0 (kRowOob: no such float)
__device__ __forceinline__ unsigned stat_voff(int col0, int ntok, int side, int lane) {
  const int j = lane >> 2, comp = lane & 3, col = col0 + j;
  return (j < ntok && col >= 0 && col < side) ? (unsigned)((col * TS_N + comp) * 4) : kRowOob;
}
// DMA of map row r's records (tokens tok0 + r*side ..) into `sbuf`; rows outside [0, side): zeros
__device__ __forceinline__ void stat_fetch(const float* __restrict__ stats, int tok0, int side, int r, unsigned voff,
                                           float* sbuf) {
#if defined(__HIP_DEVICE_COMPILE__)
  const bool live = r >= 0 && r < side;                 // wave-uniform
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(stats) + (size_t)(tok0 + (live ? r : 0) * side) * TS_N,
                                                    0, live ? side * TS_N * 4 : 0, kBufFlags);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)sbuf, 4, voff, 0, 0, 0);
#endif
}
#define MRLA_B64(i, off) "ds_read_b64 %" #i ", %[a] offset:" #off "\n\t"
// LN_x records of the 11 window tokens (floats 0..1 of each 16-byte record)
__device__ __forceinline__ void stat_read_issue_x(const float* sbuf, StatRow<kS + 4>& o) {
  static_assert(kS + 4 == 11, "window width");
  const unsigned a = lds_addr_of(sbuf);
  asm volatile(MRLA_B64(0, 0) MRLA_B64(1, 16) MRLA_B64(2, 32) MRLA_B64(3, 48) MRLA_B64(4, 64) MRLA_B64(5, 80) MRLA_B64(6, 96)
               MRLA_B64(7, 112) MRLA_B64(8, 128) MRLA_B64(9, 144) MRLA_B64(10, 160) ""
               : "=&v"(o.v[0]), "=&v"(o.v[1]), "=&v"(o.v[2]), "=&v"(o.v[3]), "=&v"(o.v[4]), "=&v"(o.v[5]), "=&v"(o.v[6]),
                 "=&v"(o.v[7]), "=&v"(o.v[8]), "=&v"(o.v[9]), "=&v"(o.v[10])
               : [a] "v"(a) : "memory");
}
// LN_o records of the 7 owned tokens = window tokens 2..8 (floats 2..3 of each record)
__device__ __forceinline__ void stat_read_issue_o(const float* sbuf, StatRow<kS>& o) {
  static_assert(kS == 7, "strip width");
  const unsigned a = lds_addr_of(sbuf);
  asm volatile(MRLA_B64(0, 40) MRLA_B64(1, 56) MRLA_B64(2, 72) MRLA_B64(3, 88) MRLA_B64(4, 104) MRLA_B64(5, 120) MRLA_B64(6, 136) ""
               : "=&v"(o.v[0]), "=&v"(o.v[1]), "=&v"(o.v[2]), "=&v"(o.v[3]), "=&v"(o.v[4]), "=&v"(o.v[5]), "=&v"(o.v[6])
               : [a] "v"(a) : "memory");
}
#undef MRLA_B64
// ties the registers to the (already executed) s_waitcnt of the step's first fence
__device__ __forceinline__ void stat_fence(StatRow<kS + 4>& o) {
  asm volatile("" : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]), "+v"(o.v[4]), "+v"(o.v[5]), "+v"(o.v[6]),
                    "+v"(o.v[7]), "+v"(o.v[8]), "+v"(o.v[9]), "+v"(o.v[10]) : : "memory");
}
__device__ __forceinline__ void stat_fence_wait(StatRow<kS>& o) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]), "+v"(o.v[4]), "+v"(o.v[5]), "+v"(o.v[6]) : : "memory");
}

// (A second row of DMA look-ahead -- double row buffers, counted `s_waitcnt vmcnt(N)`, asm LDS reads -- was built and measured
// in round 4: bit-identical, 57.6 -> 58.1 us; the waits were the scalar LayerNorm-statistics loads, see above.  Removed.)
template <typename T> constexpr int tok_bwd_wave_bytes() {
  return RowIO<T, kS + 4>::kBytes + RowIO<T, kS + 2>::kBytes + RowIO<T, kS>::kBytes + RowIO<float, kS>::kBytes +
         3 * kStatBufBytes;
}

// BASE: the MRLA-base token module's value backward (mrla_token_base_value_bwd): dU is READ -- the dense dV_t image
// `dv` [b, side, side, c] -- instead of formed from dOut; no o_{t-1} / lambda / gate terms, no bmom.
template <typename T, bool RAGGED, bool BASE>
__global__ __launch_bounds__(kMaxStrips * kWave) void token_apply_bwd_rows(
    const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ stats,
    const float* __restrict__ wx, const float* __restrict__ bx, const float* __restrict__ wo,
    const float* __restrict__ bo, const float* __restrict__ wv, const float* __restrict__ gate,
    const float* __restrict__ lam, const T* __restrict__ dv, float* __restrict__ dxn, float* __restrict__ part,
    float* __restrict__ bmom, int n, int C, int side, int d) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;
  float* red = reinterpret_cast<float*>(smem_raw);
  unsigned char* wbuf = smem_raw + (size_t)nwaves * (TQ_N + 1) * kWave * sizeof(float) + (size_t)wave * tok_bwd_wave_bytes<T>();
  constexpr int XB_ = RowIO<T, kS + 4>::kBytes, GB_ = RowIO<T, kS + 2>::kBytes, OB_ = RowIO<T, kS>::kBytes;
  T* bufX = reinterpret_cast<T*>(wbuf);
  T* bufG = reinterpret_cast<T*>(wbuf + XB_);
  T* bufO = reinterpret_cast<T*>(wbuf + XB_ + GB_);
  float* bufS = reinterpret_cast<float*>(wbuf + XB_ + GB_ + OB_);
  // LayerNorm records of map row r live in stat buffer r mod 3: rows rr and rr + 1 are alive in a step, and the row fetched
  // during the step (rr + 2) must not land on row rr's, whose LN_o records are read late in the step
  auto sbuf = [&](int r) {
    return reinterpret_cast<float*>(wbuf + XB_ + GB_ + OB_ + RowIO<float, kS>::kBytes) + ((r + 3) % 3) * (kStatBufBytes / 4);   // r >= -1
  };
  const int cbase = blockIdx.x * kWave, c = cbase + lane;
  const int b = blockIdx.y, W = side, H = side;
  const int nstrips = (W + kS - 1) / kS;
  const int tok0 = b * n + 1;
  const size_t ioff = (size_t)tok0 * C;
  const int rowelems = W * C;
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[c * 9 + k];
  const float a = BASE ? 1.f : gate[(size_t)b * (C / d) + c / d];
  const float wxc = wx[c], bxc = bx[c];
  const float lm = BASE ? 0.f : lam[c], woc = BASE ? 0.f : wo[c], boc = BASE ? 0.f : bo[c];
  const T* xi = x + ioff;
  const T* gi = BASE ? dv + (size_t)b * H * W * C : dout + ioff;       // rows of dV_t (dense image) / of dOut (token rows)
  const T* oi = BASE ? nullptr : o + ioff;
  float* dxo = dxn + ioff;
  float q[TQ_N + 1];
#pragma unroll
  for (int k = 0; k < TQ_N + 1; ++k) q[k] = 0.f;
  for (int s = wave; s < nstrips; s += nwaves) {
    const int s0 = s * kS, nc = min(kS, W - s0);
    RowIO<T, kS + 4> ax;
    RowIO<T, kS + 2> ag;
    RowIO<T, kS> ao;
    RowIO<float, kS> as;
    make_row_io<T, kS + 4>(ax, s0 - 2, kS + 4, W, C, cbase, lane);
    make_row_io<T, kS + 2>(ag, s0 - 1, kS + 2, W, C, cbase, lane);
    make_row_io<T, kS>(ao, s0, nc, W, C, cbase, lane);
    make_row_io<float, kS>(as, s0, nc, W, C, cbase, lane);
    const unsigned svoff = stat_voff(s0 - 2, kS + 4, side, lane);
    RawRow<kS + 4> xr;
    RawRow<kS + 2> gv;
    RawRow<kS> ov;
    StatRow<kS + 4> sx;                              // LN_x (mean, rstd) of window row rr+1
    StatRow<kS> so;                                  // LN_o (mean, rstd) of the owned pixels of row rr
    xr.clear(); gv.clear(); ov.clear();
    float xa[kS + 4], xb[kS + 4], xc[kS + 4];        // xn rows rr-1, rr, rr+1 on columns s0-2 .. s0+kS+1
    float ua[kS + 2], ub[kS + 2], uc[kS + 2];        // dU rows rr-2, rr-1, rr on columns s0-1 .. s0+kS
    float h0[kS], h1[kS], h2[kS];                    // xhat of the owned pixels, rows rr-1 .. rr+1
#pragma unroll
    for (int j = 0; j < kS + 4; ++j) { xa[j] = 0.f; xb[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < kS + 2; ++j) { ua[j] = 0.f; ub[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < kS; ++j) { h0[j] = 0.f; h1[j] = 0.f; }
    // step rr (= -1 .. H) consumes x row rr+1 and dOut / o rows rr
    row_fetch<T, kS + 4>(ax, xi, 0, H, rowelems, bufX);
    row_fetch<T, kS + 2>(ag, gi, -1, H, rowelems, bufG);
    if (!BASE) row_fetch<T, kS>(ao, oi, -1, H, rowelems, bufO);
    if (!BASE) stat_fetch(stats, tok0, side, -1, svoff, sbuf(-1));      // (zeros: LN_o of the row above the map)
    stat_fetch(stats, tok0, side, 0, svoff, sbuf(0));
    auto step = [&](int rr, float (&XA)[kS + 4], float (&XB)[kS + 4], float (&XC)[kS + 4], float (&UA)[kS + 2],
                    float (&UB)[kS + 2], float (&UC)[kS + 2], float (&H0)[kS], float (&H1)[kS], float (&H2)[kS]) {
      rows_landed();
      row_read_issue<T, kS + 4>(bufX, lane, xr);
      row_read_issue<T, kS + 2>(bufG, lane, gv);
      if (!BASE) row_read_issue<T, kS>(bufO, lane, ov);
      stat_read_issue_x(sbuf(rr + 1), sx);
      row_read_fence(xr, true);
      row_read_fence(gv, false);
      if (!BASE) row_read_fence(ov, false);
      if (sizeof(T) == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (generic fp32 row reads carry no fence)
      stat_fence(sx);
      // the buffers just read are free: the next step's rows go there
      row_fetch<T, kS + 4>(ax, xi, rr + 2, H, rowelems, bufX);
      row_fetch<T, kS + 2>(ag, gi, rr + 1, H, rowelems, bufG);
      if (!BASE) row_fetch<T, kS>(ao, oi, rr + 1, H, rowelems, bufO);
      stat_fetch(stats, tok0, side, rr + 2, svoff, sbuf(rr + 2));
      // LN_x on the way in (row rr+1); pixels outside the map come out as exact zeros
#pragma unroll
      for (int j = 0; j < kS + 4; ++j) {
        const bool ok = rr + 1 >= 0 && rr + 1 < side && s0 - 2 + j >= 0 && s0 - 2 + j < side;      // wave-uniform
        const float on = ok ? 1.f : 0.f;
        const float hat = (xr.v[j] - sx.v[j].x) * sx.v[j].y;          // (rstd arrives as 0 outside the map)
        XC[j] = fmaf(hat, wxc, on * bxc);
        if (j >= 2 && j < kS + 2) H2[j - 2] = hat;
      }
      // LN_o's records of row rr: read only now that LN_x's have been consumed (14 registers less at the peak); the wait
      // sits in front of the first owned column, behind column 0's arithmetic
      if (!BASE) stat_read_issue_o(sbuf(rr), so);
      // dU of row rr on columns s0-1 .. s0+kS (dOut is zero outside the map and in the steps rr = -1 and rr = H)
#pragma unroll
      for (int j = 0; j < kS + 2; ++j) {
        if constexpr (BASE) {
          const float du = gv.v[j];
          UC[j] = du;
          if (j >= 1 && j <= kS) {
            q[0] = fmaf(du, XA[j], q[0]); q[1] = fmaf(du, XA[j + 1], q[1]); q[2] = fmaf(du, XA[j + 2], q[2]);
            q[3] = fmaf(du, XB[j], q[3]); q[4] = fmaf(du, XB[j + 1], q[4]); q[5] = fmaf(du, XB[j + 2], q[5]);
            q[6] = fmaf(du, XC[j], q[6]); q[7] = fmaf(du, XC[j + 1], q[7]); q[8] = fmaf(du, XC[j + 2], q[8]);
          }
          continue;
        }
        const float u = conv_at(w, XA, XB, XC, j);
        const float go = gv.v[j];
        const float du = a * go * gelu_grad_f(u);
        UC[j] = du;
        if (j == 1) stat_fence_wait(so);             // LN_o's records (issued above, behind column 0's arithmetic)
        if (j >= 1 && j <= kS) {                     // owned column (compile-time after unroll)
          q[TQ_N] = fmaf(go, gelu_f(u), q[TQ_N]);
          const float ohat = (ov.v[j - 1] - so.v[j - 1].x) * so.v[j - 1].y;   // (go is zero where the token does not exist)
          q[TQ_LAM] = fmaf(go, fmaf(ohat, woc, boc), q[TQ_LAM]);
          q[TQ_LNOW] = fmaf(lm * go, ohat, q[TQ_LNOW]);
          q[TQ_LNOB] = fmaf(lm, go, q[TQ_LNOB]);
          q[0] = fmaf(du, XA[j], q[0]); q[1] = fmaf(du, XA[j + 1], q[1]); q[2] = fmaf(du, XA[j + 2], q[2]);
          q[3] = fmaf(du, XB[j], q[3]); q[4] = fmaf(du, XB[j + 1], q[4]); q[5] = fmaf(du, XB[j + 2], q[5]);
          q[6] = fmaf(du, XC[j], q[6]); q[7] = fmaf(du, XC[j + 1], q[7]); q[8] = fmaf(du, XC[j + 2], q[8]);
        }
      }
      if (rr >= 1) {                                 // dxn' of row rr-1 from dU rows rr-2 .. rr
        float yrow[kS];
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          float s9 = w[0] * UC[j + 2];
          s9 = fmaf(w[1], UC[j + 1], s9); s9 = fmaf(w[2], UC[j], s9);
          s9 = fmaf(w[3], UB[j + 2], s9); s9 = fmaf(w[4], UB[j + 1], s9); s9 = fmaf(w[5], UB[j], s9);
          s9 = fmaf(w[6], UA[j + 2], s9); s9 = fmaf(w[7], UA[j + 1], s9); s9 = fmaf(w[8], UA[j], s9);
          yrow[j] = s9;
          q[TQ_LNXW] = fmaf(s9, H0[j], q[TQ_LNXW]);                          // (xhat is zero beyond the map)
          q[TQ_H] += H0[j];
          if (!RAGGED || j < nc) q[TQ_LNXB] += s9;
        }
        row_store<float, kS>(as, dxo, rr - 1, rowelems, lane, bufS, yrow);
      }
    };
    // steps rr = -1 .. H: after three steps every array is back in its starting role
    int rr = -1;
    for (; rr + 2 <= H; rr += 3) {
      step(rr,     xa, xb, xc, ua, ub, uc, h0, h1, h2);
      step(rr + 1, xb, xc, xa, ub, uc, ua, h1, h2, h0);
      step(rr + 2, xc, xa, xb, uc, ua, ub, h2, h0, h1);
    }
    if (rr <= H) {
      step(rr, xa, xb, xc, ua, ub, uc, h0, h1, h2);
      if (rr + 1 <= H) step(rr + 1, xb, xc, xa, ub, uc, ua, h1, h2, h0);
    }
    rows_landed();
  }
  wg_reduce<TQ_N + 1>(q, red, lane, wave, nwaves);
  if (wave == 0) {
    if (!BASE) {
      float* bm = bmom + ((size_t)b * C + c) * D_N;
      bm[D_D] = 0.f; bm[D_DV] = q[TQ_N]; bm[D_DO] = 0.f;
    }
    const size_t g = (size_t)b * n * C + c;          // cls row: the module output there is LN_x(x) itself
    const float* st = stats + (size_t)b * n * TS_N;
    const float dn = to_f(dout[g]);
    dxn[g] = dn;
    q[TQ_LNXW] = fmaf(dn, (to_f(x[g]) - st[TS_MX]) * st[TS_RX], q[TQ_LNXW]);
    q[TQ_LNXB] += dn;
#pragma unroll
    for (int k = 0; k < TQ_N; ++k) part[((size_t)b * C + c) * TQ_N + k] = q[k];
  }
}

// ------------------------------------------------------------------------------------------------
// MRLA-base on tokens (deit/deit_mrla_base.py:224-243): the value map V_t = dwconv3x3(LN_x(x) map) goes straight into the
// stage's slot-major NHWC ring (a dense [b, side, side, c] image per slot) -- LN_x(x) itself is never materialised --
// and the backward (token_apply_bwd_rows<.., BASE = true> above) turns dV_t (dense, from mrla_base_dv_combine) into dxn'
// on the token rows.  Its forward is the VALUE form of the kernel below.
// ------------------------------------------------------------------------------------------------
// forward on the LDS-DMA row pipeline (round 4; it replaced the register-staged row gathers of rounds 1 - 3: token_apply_fwd
// 0.387 -> 0.323 ms, token_base_value_fwd 0.269 -> 0.216 ms per step): rows by `buffer_load ... lds` with zeros outside the
// map, the LayerNorm records of a window row through LDS (one dword DMA per step, broadcast ds_read_b64), the three LN_x rows
// rotating by name.
//   VALUE = false (mrla_token_apply_fwd):  out[b,i>=1] = res*x + a*gelu(dwconv3x3(LN_x(x) map)) + lam*LN_o(o_prev); cls row
//   VALUE = true  (mrla_token_base_value_fwd): V_t = dwconv3x3(LN_x(x) map) -> the dense [b, side, side, c] ring slot
// Step r (r0-2 .. r1-1 of the workgroup's row band) normalises x row r+1 and, from r0 on, writes output row r.
// ------------------------------------------------------------------------------------------------
// LN_x records of the 9 window tokens s0-1 .. s0+7 / LN_o records of the owned tokens = window tokens 1 .. 7
#define MRLA_B64(i, off) "ds_read_b64 %" #i ", %[a] offset:" #off "\n\t"
__device__ __forceinline__ void stat_read_issue_x9(const float* sbuf, StatRow<kS + 2>& o) {
  static_assert(kS + 2 == 9, "window width");
  const unsigned a = lds_addr_of(sbuf);
  asm volatile(MRLA_B64(0, 0) MRLA_B64(1, 16) MRLA_B64(2, 32) MRLA_B64(3, 48) MRLA_B64(4, 64) MRLA_B64(5, 80) MRLA_B64(6, 96)
               MRLA_B64(7, 112) MRLA_B64(8, 128) ""
               : "=&v"(o.v[0]), "=&v"(o.v[1]), "=&v"(o.v[2]), "=&v"(o.v[3]), "=&v"(o.v[4]), "=&v"(o.v[5]), "=&v"(o.v[6]),
                 "=&v"(o.v[7]), "=&v"(o.v[8])
               : [a] "v"(a) : "memory");
}
__device__ __forceinline__ void stat_read_issue_o9(const float* sbuf, StatRow<kS>& o) {
  const unsigned a = lds_addr_of(sbuf);
  asm volatile(MRLA_B64(0, 24) MRLA_B64(1, 40) MRLA_B64(2, 56) MRLA_B64(3, 72) MRLA_B64(4, 88) MRLA_B64(5, 104) MRLA_B64(6, 120) ""
               : "=&v"(o.v[0]), "=&v"(o.v[1]), "=&v"(o.v[2]), "=&v"(o.v[3]), "=&v"(o.v[4]), "=&v"(o.v[5]), "=&v"(o.v[6])
               : [a] "v"(a) : "memory");
}
#undef MRLA_B64
__device__ __forceinline__ void stat_fence_wait(StatRow<kS + 2>& o) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]), "+v"(o.v[4]), "+v"(o.v[5]), "+v"(o.v[6]),
                                        "+v"(o.v[7]), "+v"(o.v[8]) : : "memory");
}
__device__ __forceinline__ void stat_fence(StatRow<kS>& o) {
  asm volatile("" : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]), "+v"(o.v[4]), "+v"(o.v[5]), "+v"(o.v[6]) : : "memory");
}

template <typename T> constexpr int tok_fwd_wave_bytes() {
  return RowIO<T, kS + 2>::kBytes + 2 * RowIO<T, kS>::kBytes + 3 * kStatBufBytes;
}

template <typename T, bool VALUE>
__global__ __launch_bounds__(kMaxStrips * kWave) void token_fwd_rows(
    const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ stats, const float* __restrict__ wx,
    const float* __restrict__ bx, const float* __restrict__ wo, const float* __restrict__ bo,
    const float* __restrict__ wv, const float* __restrict__ gate, const float* __restrict__ lam, T* __restrict__ out,
    int n, int C, int side, int d, int res) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;
  unsigned char* wbuf = smem_raw + (size_t)wave * tok_fwd_wave_bytes<T>();
  constexpr int XB_ = RowIO<T, kS + 2>::kBytes, OB_ = RowIO<T, kS>::kBytes;
  T* bufX = reinterpret_cast<T*>(wbuf);
  T* bufO = reinterpret_cast<T*>(wbuf + XB_);
  T* bufS = reinterpret_cast<T*>(wbuf + XB_ + OB_);
  auto sbuf = [&](int r) { return reinterpret_cast<float*>(wbuf + XB_ + 2 * OB_) + ((r + 6) % 3) * (kStatBufBytes / 4); };   // r >= -2
  const int cbase = blockIdx.x * kWave, c = cbase + lane;
  const int b = blockIdx.y, W = side, H = side;
  const int nstrips = (W + kS - 1) / kS;
  const int tok0 = b * n + 1;
  const size_t ioff = (size_t)tok0 * C;
  const int rowelems = W * C;
  const int r0 = (int)((H * blockIdx.z) / gridDim.z), r1 = (int)((H * (blockIdx.z + 1)) / gridDim.z);   // this workgroup's row band
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[c * 9 + k];
  const float a = VALUE ? 1.f : gate[(size_t)b * (C / d) + c / d];
  const float wxc = wx[c], bxc = bx[c];
  const float lm = VALUE ? 0.f : lam[c], woc = VALUE ? 0.f : wo[c], boc = VALUE ? 0.f : bo[c];
  const float resf = (!VALUE && res) ? 1.f : 0.f;
  const T* xi = x + ioff;
  const T* oi = VALUE ? nullptr : o + ioff;
  T* yo = VALUE ? out + (size_t)b * H * W * C : out + ioff;       // the dense ring slot / the token rows of `out`
  for (int s = wave; s < nstrips; s += nwaves) {
    const int s0 = s * kS, nc = min(kS, W - s0);
    RowIO<T, kS + 2> ax;
    RowIO<T, kS> ao, as;
    make_row_io<T, kS + 2>(ax, s0 - 1, kS + 2, W, C, cbase, lane);
    make_row_io<T, kS>(ao, s0, nc, W, C, cbase, lane);
    make_row_io<T, kS>(as, s0, nc, W, C, cbase, lane);
    const unsigned svoff = stat_voff(s0 - 1, kS + 2, side, lane);
    RawRow<kS + 2> xr;
    RawRow<kS> ov;
    StatRow<kS + 2> sx;
    StatRow<kS> so;
    xr.clear(); ov.clear();
    float xa[kS + 2], xb[kS + 2], xc[kS + 2];        // LN_x rows r-1, r, r+1 on columns s0-1 .. s0+kS
    float raw0[kS], raw1[kS];                        // raw x of the owned pixels of rows r / r+1 (the block residual)
#pragma unroll
    for (int j = 0; j < kS + 2; ++j) { xa[j] = 0.f; xb[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < kS; ++j) raw0[j] = 0.f;
    // step r consumes x row r+1 and o row r
    row_fetch<T, kS + 2>(ax, xi, r0 - 1, H, rowelems, bufX);
    if (!VALUE) row_fetch<T, kS>(ao, oi, r0 - 2, H, rowelems, bufO);
    stat_fetch(stats, tok0, side, r0 - 2, svoff, sbuf(r0 - 2));
    stat_fetch(stats, tok0, side, r0 - 1, svoff, sbuf(r0 - 1));
    auto step = [&](int r, float (&XA)[kS + 2], float (&XB)[kS + 2], float (&XC)[kS + 2]) {
      // step r-1 (if it wrote a row) stored after its fetches: that store may stay in flight
      if (r <= r0) rows_landed(); else rows_landed_keep<RowIO<T, kS>::NL>();
      row_read_issue<T, kS + 2>(bufX, lane, xr);
      if (!VALUE) row_read_issue<T, kS>(bufO, lane, ov);
      stat_read_issue_x9(sbuf(r + 1), sx);
      if (!VALUE) stat_read_issue_o9(sbuf(r), so);
      row_read_fence(xr, true);
      if (!VALUE) row_read_fence(ov, false);
      stat_fence_wait(sx);
      if (!VALUE) stat_fence(so);
      row_fetch<T, kS + 2>(ax, xi, r + 2, H, rowelems, bufX);
      if (!VALUE) row_fetch<T, kS>(ao, oi, r + 1, H, rowelems, bufO);
      stat_fetch(stats, tok0, side, r + 2, svoff, sbuf(r + 2));
      // LN_x on the way in (row r+1); pixels outside the map come out as exact zeros
#pragma unroll
      for (int j = 0; j < kS + 2; ++j) {
        const bool ok = r + 1 >= 0 && r + 1 < side && s0 - 1 + j >= 0 && s0 - 1 + j < side;      // wave-uniform
        const float hat = (xr.v[j] - sx.v[j].x) * sx.v[j].y;        // (rstd arrives as 0 outside the map)
        XC[j] = fmaf(hat, wxc, ok ? bxc : 0.f);
        if (j >= 1 && j <= kS) raw1[j - 1] = xr.v[j];
      }
      if (r >= r0) {
        float y[kS];
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          const float u = conv_at(w, XA, XB, XC, j);
          if (VALUE) {
            y[j] = u;
          } else {
            const float on = fmaf((ov.v[j] - so.v[j].x) * so.v[j].y, woc, boc);
            y[j] = fmaf(a, gelu_f(u), fmaf(lm, on, resf * raw0[j]));
          }
        }
        row_store<T, kS>(as, yo, r, rowelems, lane, bufS, y);
      }
#pragma unroll
      for (int j = 0; j < kS; ++j) raw0[j] = raw1[j];
    };
    // steps r0-2 .. r1-1: after three steps every array is back in its starting role
    int r = r0 - 2;
    for (; r + 3 <= r1; r += 3) {
      step(r,     xa, xb, xc);
      step(r + 1, xb, xc, xa);
      step(r + 2, xc, xa, xb);
    }
    if (r < r1) {
      step(r, xa, xb, xc);
      if (r + 1 < r1) step(r + 1, xb, xc, xa);
    }
    rows_landed();
  }
  if (!VALUE && wave == 0 && blockIdx.z == 0) {      // cls row of this (image, 64 channels): out = res*x + LN_x(x)
    const size_t g = (size_t)b * n * C + c;
    const float* st = stats + (size_t)b * n * TS_N;
    const float xv = to_f(x[g]);
    out[g] = from_f<T>(fmaf(resf, xv, fmaf((xv - st[TS_MX]) * st[TS_RX], wxc, bxc)));
  }
}

// ------------------------------------------------------------------------------------------------
// launchers (C % 64 == 0)
// ------------------------------------------------------------------------------------------------
#define MRLA_DISPATCH_TN(DT, CALL)       \
  switch (DT) {                          \
    case MRLA_F32:  CALL(float); break;  \
    case MRLA_BF16: CALL(bf16_t); break; \
    case MRLA_F16:  CALL(f16_t); break;  \
    default: return MRLA_EINVAL;         \
  }

bool token_nhwc_applies(int C) { return C % kWave == 0; }

// Row bands of the forward apply kernel: a 14 x 14 map gives a wave only 14 dependent row steps and the grid only
// (C/64) * B two-wave workgroups; splitting the rows (one halo row of recompute) doubles the waves in flight: 0.59 ->
// 0.45 ms per step on deit_mrlal_tiny.  The backward kernel supports bands as well but is bound by its arithmetic
// (exact GELU and its derivative per pixel): the two halo rows per band made it 7 % slower, so it runs unbanded (and its
// bmom sum relies on that).
int token_bands_bwd(int, int, int) { return 1; }
int token_bands(int B, int C, int side) {
  if (!token_nhwc_applies(C)) return 1;
  const int wgs = (C / kWave) * B;
  int bands = 1;
  while (bands < 4 && wgs * bands < 1536 && side / (bands * 2) >= 4) bands *= 2;
  return bands;
}

int launch_token_apply_fwd_nhwc(const void* x, const void* o, const float* stats, const float* wx, const float* bx,
                                const float* wo, const float* bo, const float* wv, const float* gate, const float* lam,
                                void* out, int B, int n, int C, int side, int d, int res, int dtype, hipStream_t st) {
  const int nwaves = std::min((side + kS - 1) / kS, kMaxStrips);
  const dim3 grid(C / kWave, B, token_bands(B, C, side)), block(nwaves * kWave);
#define CALL(TT)                                                                                                    \
  {                                                                                                                 \
    const size_t lds = (size_t)nwaves * tok_fwd_wave_bytes<TT>();                                                   \
    if (set_lds_n(token_fwd_rows<TT, false>, lds) != hipSuccess) return MRLA_EHIP;                                   \
    hipLaunchKernelGGL((token_fwd_rows<TT, false>), grid, block, lds, st, (const TT*)x, (const TT*)o, stats, wx, bx, wo, \
                       bo, wv, gate, lam, (TT*)out, n, C, side, d, res);                                            \
  }
  MRLA_DISPATCH_TN(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_token_apply_bwd_nhwc(const void* dout, const void* x, const void* o, const float* stats, const float* wx,
                                const float* bx, const float* wo, const float* bo, const float* wv, const float* gate,
                                const float* lam, float* dxn, float* part, float* bmom, int B, int n, int C, int side,
                                int d, int dtype, hipStream_t st) {
  const int nwaves = std::min((side + kS - 1) / kS, kMaxStrips);
  const dim3 grid(C / kWave, B), block(nwaves * kWave);
  const bool ragged = side % kS != 0;
#define CALL_R(TT, RG)                                                                                             \
  {                                                                                                                \
    const size_t lds = (size_t)nwaves * (TQ_N + 1) * kWave * sizeof(float) + (size_t)nwaves * tok_bwd_wave_bytes<TT>(); \
    if (set_lds_n(token_apply_bwd_rows<TT, RG, false>, lds) != hipSuccess) return MRLA_EHIP;                         \
    hipLaunchKernelGGL((token_apply_bwd_rows<TT, RG, false>), grid, block, lds, st, (const TT*)dout, (const TT*)x,     \
                       (const TT*)o, stats, wx, bx, wo, bo, wv, gate, lam, (const TT*)nullptr, dxn, part, bmom, n, C, side, d); \
  }
#define CALL(TT) { if (ragged) CALL_R(TT, true) else CALL_R(TT, false) }
  MRLA_DISPATCH_TN(dtype, CALL)
#undef CALL
#undef CALL_R
  return hip_status(hipGetLastError());
}

int launch_token_value_fwd_nhwc(const void* x, const float* stats, const float* wx, const float* bx, const float* wv,
                                void* vslot, int B, int n, int C, int side, int dtype, hipStream_t st) {
  if (!token_nhwc_applies(C)) return MRLA_EUNSUPPORTED;
  const int nwaves = std::min((side + kS - 1) / kS, kMaxStrips);
  const dim3 grid(C / kWave, B, token_bands(B, C, side)), block(nwaves * kWave);
#define CALL(TT)                                                                                                    \
  {                                                                                                                 \
    const size_t lds = (size_t)nwaves * tok_fwd_wave_bytes<TT>();                                                   \
    if (set_lds_n(token_fwd_rows<TT, true>, lds) != hipSuccess) return MRLA_EHIP;                                    \
    hipLaunchKernelGGL((token_fwd_rows<TT, true>), grid, block, lds, st, (const TT*)x, (const TT*)nullptr, stats, wx, bx, \
                       (const float*)nullptr, (const float*)nullptr, wv, (const float*)nullptr, (const float*)nullptr, \
                       (TT*)vslot, n, C, side, 1, 0);                                                               \
  }
  MRLA_DISPATCH_TN(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

// out[b, 0, :] = LN_x(x)[b, 0, :] (the cls token passes through the module, deit_mrla_base.py:241)
int launch_token_cls_fwd(const void* x, const float* stats, const float* wx, const float* bx, void* out, int B, int n,
                         int C, int dtype, hipStream_t st) {
#define CALL(TT)                                                                                                    \
  hipLaunchKernelGGL((token_cls_fwd_kernel<TT>), dim3((C + kThreads - 1) / kThreads, B), dim3(kThreads), 0, st,      \
                     (const TT*)x, stats, wx, bx, (TT*)out, n, C, 0);
  MRLA_DISPATCH_TN(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_token_value_bwd_nhwc(const void* dout, const void* x, const float* stats, const float* wx, const float* bx,
                                const float* wv, const void* dv, float* dxn, float* part, int B, int n, int C, int side,
                                int dtype, hipStream_t st) {
  if (!token_nhwc_applies(C)) return MRLA_EUNSUPPORTED;
  const int nwaves = std::min((side + kS - 1) / kS, kMaxStrips);
  const dim3 grid(C / kWave, B), block(nwaves * kWave);
  const bool ragged = side % kS != 0;
#define CALL_R(TT, RG)                                                                                             \
  {                                                                                                                \
    const size_t lds = (size_t)nwaves * (TQ_N + 1) * kWave * sizeof(float) + (size_t)nwaves * tok_bwd_wave_bytes<TT>(); \
    if (set_lds_n(token_apply_bwd_rows<TT, RG, true>, lds) != hipSuccess) return MRLA_EHIP;                          \
    hipLaunchKernelGGL((token_apply_bwd_rows<TT, RG, true>), grid, block, lds, st, (const TT*)dout, (const TT*)x,      \
                       (const TT*)nullptr, stats, wx, bx, (const float*)nullptr, (const float*)nullptr, wv,        \
                       (const float*)nullptr, (const float*)nullptr, (const TT*)dv, dxn, part, (float*)nullptr, n, C, side, 1); \
  }
#define CALL(TT) { if (ragged) CALL_R(TT, true) else CALL_R(TT, false) }
  MRLA_DISPATCH_TN(dtype, CALL)
#undef CALL
#undef CALL_R
  return hip_status(hipGetLastError());
}

}  // namespace mrla
