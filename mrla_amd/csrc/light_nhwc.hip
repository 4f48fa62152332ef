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
#include "mrla_device.h"
#include "mrla_kernels.h"

namespace mrla {

constexpr int kS = 7;          // owned columns per strip
constexpr int kMaxStrips = 8;  // waves per workgroup (wider images loop strips inside a wave)

template <typename T>
__device__ __forceinline__ float ldpix(const T* __restrict__ img, int r, int col, int H, int W, int C, int c) {
  // wave-uniform predicate: all lanes look at the same pixel
  if (r < 0 || r >= H || col < 0 || col >= W) return 0.f;
  return to_f(img[((size_t)r * W + col) * C + c]);
}

__device__ __forceinline__ float conv_at(const float (&w)[9], const float* __restrict__ ra, const float* __restrict__ rb,
                                         const float* __restrict__ rc, int j) {
  float s = w[0] * ra[j];
  s = fmaf(w[1], ra[j + 1], s); s = fmaf(w[2], ra[j + 2], s);
  s = fmaf(w[3], rb[j], s); s = fmaf(w[4], rb[j + 1], s); s = fmaf(w[5], rb[j + 2], s);
  s = fmaf(w[6], rc[j], s); s = fmaf(w[7], rc[j + 1], s); s = fmaf(w[8], rc[j + 2], s);
  return s;
}

// Sum per-lane accumulators over the waves of the workgroup (fixed order); result valid in wave 0.
template <int K>
__device__ __forceinline__ void wg_reduce(float (&acc)[K], float* __restrict__ red, int lane, int wave, int nwaves) {
  if (nwaves == 1) return;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) red[(wave * K + k) * kWave + lane] = acc[k];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float s = 0.f;
      for (int v = 0; v < nwaves; ++v) s += red[(v * K + k) * kWave + lane];
      acc[k] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward statistics (+ optional fused producer x = relu(pre + o))
// ------------------------------------------------------------------------------------------------
template <typename T, bool GELU, bool HAS_O, bool FUSE>
__global__ __launch_bounds__(kMaxStrips * kWave) void light_stats_fwd_nhwc(const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv,
                                     float* __restrict__ mom, T* __restrict__ xout, int B, int C, int H, int W, int BG) {
  extern __shared__ float red[];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;
  const int c = blockIdx.x * kWave + lane;
  const bool cv = c < C;
  const int cc = cv ? c : C - 1;
  const int nstrips = (W + kS - 1) / kS;
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[cc * 9 + k];
  const int b_end = min(B, (int)(blockIdx.y + 1) * BG);
  for (int b = blockIdx.y * BG; b < b_end; ++b) {
    const size_t ioff = (size_t)b * H * W * C;
    const T* xi = x + ioff;
    const T* oi = HAS_O ? o + ioff : nullptr;
    T* xo = FUSE ? xout + ioff : nullptr;
    float acc[M_N] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = wave; s < nstrips; s += nwaves) {
      const int s0 = s * kS, nc = min(kS, W - s0);
      float ra[kS + 2], rb[kS + 2], rc[kS + 2];        // x rows r-1, r, r+1 over columns s0-1 .. s0+kS
      auto load_row = [&](int r, float* dst) {
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) {
          const int col = s0 - 1 + j;
          float v = ldpix(xi, r, col, H, W, C, cc);
          if (FUSE) {
            const bool in = r >= 0 && r < H && col >= 0 && col < W;
            v = in ? fmaxf(to_f(from_f<T>(v + ldpix(oi, r, col, H, W, C, cc))), 0.f) : 0.f;
            if (in && j >= 1 && j <= nc && cv) xo[((size_t)r * W + col) * C + c] = from_f<T>(v);
          }
          dst[j] = v;
        }
      };
#pragma unroll
      for (int j = 0; j < kS + 2; ++j) ra[j] = 0.f;
      load_row(0, rb);
      for (int r = 0; r < H; ++r) {
        load_row(r + 1, rc);
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          if (j < nc) {
            float v = conv_at(w, ra, rb, rc, j);
            if (GELU) v = gelu_f(v);
            acc[M_SX] += rb[j + 1];
            acc[M_SV] += v;
            acc[M_SVV] = fmaf(v, v, acc[M_SVV]);
            if (HAS_O) {
              const float ov = to_f(oi[((size_t)r * W + s0 + j) * C + cc]);
              acc[M_SO] += ov;
              acc[M_SVO] = fmaf(v, ov, acc[M_SVO]);
              acc[M_SOO] = fmaf(ov, ov, acc[M_SOO]);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) { ra[j] = rb[j]; rb[j] = rc[j]; }
      }
    }
    wg_reduce<M_N>(acc, red, lane, wave, nwaves);
    if (wave == 0 && cv) {
#pragma unroll
      for (int k = 0; k < M_N; ++k) mom[((size_t)b * C + c) * M_N + k] = acc[k];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward apply:  out = res*x + A*V + B*o + C
// ------------------------------------------------------------------------------------------------
template <typename T, bool GELU, bool HAS_O>
__global__ __launch_bounds__(kMaxStrips * kWave) void light_apply_fwd_nhwc(const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv,
                                     const float* __restrict__ gate, const float* __restrict__ sc,
                                     const float* __restrict__ sh, const float* __restrict__ lam,
                                     const float* __restrict__ dp, T* __restrict__ out, int B, int C, int H, int W,
                                     int BG, int d, int res) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;
  const int c = blockIdx.x * kWave + lane;
  const bool cv = c < C;
  const int cc = cv ? c : C - 1;
  const int nstrips = (W + kS - 1) / kS;
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
      for (int j = 0; j < kS + 2; ++j) { ra[j] = 0.f; rb[j] = ldpix(xi, 0, s0 - 1 + j, H, W, C, cc); }
      for (int r = 0; r < H; ++r) {
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) rc[j] = ldpix(xi, r + 1, s0 - 1 + j, H, W, C, cc);
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          if (j < nc) {
            const size_t e = ((size_t)r * W + s0 + j) * C + cc;
            float y;
            if (GELU) y = fmaf(A, gelu_f(conv_at(w, ra, rb, rc, j)), fmaf(resf, rb[j + 1], Cc));
            else      y = conv_at(w, ra, rb, rc, j) + Cc;
            if (HAS_O) y = fmaf(Bc, to_f(oi[e]), y);
            if (cv) yo[e] = from_f<T>(y);
          }
        }
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) { ra[j] = rb[j]; rb[j] = rc[j]; }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward statistics
// ------------------------------------------------------------------------------------------------
template <typename T, bool GELU, bool HAS_O>
__global__ __launch_bounds__(kMaxStrips * kWave) void light_stats_bwd_nhwc(const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ o,
                                     const float* __restrict__ wv, float* __restrict__ bmom, int B, int C, int H, int W,
                                     int BG) {
  extern __shared__ float red[];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;
  const int c = blockIdx.x * kWave + lane;
  const bool cv = c < C;
  const int cc = cv ? c : C - 1;
  const int nstrips = (W + kS - 1) / kS;
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
      for (int j = 0; j < kS + 2; ++j) { ra[j] = 0.f; rb[j] = ldpix(xi, 0, s0 - 1 + j, H, W, C, cc); }
      for (int r = 0; r < H; ++r) {
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) rc[j] = ldpix(xi, r + 1, s0 - 1 + j, H, W, C, cc);
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          if (j < nc) {
            const size_t e = ((size_t)r * W + s0 + j) * C + cc;
            float v = conv_at(w, ra, rb, rc, j);
            if (GELU) v = gelu_f(v);
            const float gv = to_f(gi[e]);
            acc[D_D] += gv;
            acc[D_DV] = fmaf(gv, v, acc[D_DV]);
            if (HAS_O) acc[D_DO] = fmaf(gv, to_f(oi[e]), acc[D_DO]);
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

// ------------------------------------------------------------------------------------------------
// backward apply
// ------------------------------------------------------------------------------------------------
// Strip-local windows (columns relative to s0):  x rows rr-1..rr+1 over cols -2..kS+1 (kS+4 wide),
// dU rows rr-2..rr over cols -1..kS (kS+2 wide).  At step rr: U[rr] on cols -1..kS -> dU[rr]; then dx[rr-1] on the
// owned cols from dU rows rr-2..rr.
template <typename T, bool GELU, bool HAS_O, bool RELU>
__global__ __launch_bounds__(kMaxStrips * kWave) void light_apply_bwd_nhwc(const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ o,
                                     const float* __restrict__ wv, const float* __restrict__ gate,
                                     const float* __restrict__ cb, const float* __restrict__ lam,
                                     const float* __restrict__ dp, const float* __restrict__ dyx, T* __restrict__ dx,
                                     T* __restrict__ dprev, float* __restrict__ dwv_part, int B, int C, int H, int W,
                                     int BG, int d, int res) {
  extern __shared__ float red[];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;
  const int c = blockIdx.x * kWave + lane;
  const bool cv = c < C;
  const int cc = cv ? c : C - 1;
  const int nstrips = (W + kS - 1) / kS;
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
      float gprev[kS], dmprev[kS];                   // dOut[rr-1], lam*dm[rr-1] on the owned columns
#pragma unroll
      for (int j = 0; j < kS + 4; ++j) { xa[j] = 0.f; xb[j] = ldpix(xi, 0, s0 - 2 + j, H, W, C, cc); }
#pragma unroll
      for (int j = 0; j < kS + 2; ++j) { ua[j] = 0.f; ub[j] = 0.f; }
#pragma unroll
      for (int j = 0; j < kS; ++j) { gprev[j] = 0.f; dmprev[j] = 0.f; }
      for (int rr = 0; rr <= H; ++rr) {
#pragma unroll
        for (int j = 0; j < kS + 4; ++j) xc[j] = ldpix(xi, rr + 1, s0 - 2 + j, H, W, C, cc);
        float gcur[kS], dmcur[kS];
        // dU[rr] on columns -1 .. kS (zero outside the image)
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) {
          const int col = s0 - 1 + j;
          float du = 0.f;
          if (rr < H && col >= 0 && col < W) {                              // wave-uniform
            const size_t e = ((size_t)rr * W + col) * C + cc;
            const float u = conv_at(w, xa, xb, xc, j);                      // window cols j..j+2 <-> image cols col-1..col+1
            const float v = GELU ? gelu_f(u) : u;
            const float gv = to_f(gi[e]);
            float dm = fmaf(E, gv, Hc);
            dm = fmaf(F, v, dm);
            if (HAS_O) dm = fmaf(Gc, to_f(oi[e]), dm);
            du = a * dm;
            if (GELU) du *= gelu_grad_f(u);
            if (j >= 1 && j <= kS) {                                        // owned column (compile-time after unroll)
              if (j - 1 < nc) {
                gcur[j - 1] = gv;
                dmcur[j - 1] = lm * dm;
                if (HAS_O && !RELU && cv) doo[e] = from_f<T>(lm * dm);
                // dWv[i][k] += dU[rr][col] * x[rr+i-1][col+k-1]
                wg[0] = fmaf(du, xa[j], wg[0]); wg[1] = fmaf(du, xa[j + 1], wg[1]); wg[2] = fmaf(du, xa[j + 2], wg[2]);
                wg[3] = fmaf(du, xb[j], wg[3]); wg[4] = fmaf(du, xb[j + 1], wg[4]); wg[5] = fmaf(du, xb[j + 2], wg[5]);
                wg[6] = fmaf(du, xc[j], wg[6]); wg[7] = fmaf(du, xc[j + 1], wg[7]); wg[8] = fmaf(du, xc[j + 2], wg[8]);
              }
            }
          } else if (j >= 1 && j <= kS) {
            gcur[j - 1] = 0.f;
            dmcur[j - 1] = 0.f;
          }
          uc[j] = du;
        }
        // dx[rr-1] on the owned columns:  dx[ro][col] = sum_{i,k} w[i][k] * dU[ro-i+1][col-k+1]
        if (rr >= 1) {
          const int ro = rr - 1;
#pragma unroll
          for (int j = 0; j < kS; ++j) {
            if (j < nc) {
              // window index of column (col + 1 - k) in the dU arrays (which start at col -1): j + 2 - k
              float s9 = w[0] * uc[j + 2];
              s9 = fmaf(w[1], uc[j + 1], s9); s9 = fmaf(w[2], uc[j], s9);
              s9 = fmaf(w[3], ub[j + 2], s9); s9 = fmaf(w[4], ub[j + 1], s9); s9 = fmaf(w[5], ub[j], s9);
              s9 = fmaf(w[6], ua[j + 2], s9); s9 = fmaf(w[7], ua[j + 1], s9); s9 = fmaf(w[8], ua[j], s9);
              float y = fmaf(resf, gprev[j], s9 + dy);
              const size_t e = ((size_t)ro * W + s0 + j) * C + cc;
              if (RELU) {
                y = (xa[j + 2] > 0.f) ? y : 0.f;                            // xa = x[rr-1] = x[ro]; owned col j <-> window j+2
                if (HAS_O && cv) doo[e] = from_f<T>(dmprev[j] + y);
              }
              if (cv) dxo[e] = from_f<T>(y);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < kS + 4; ++j) { xa[j] = xb[j]; xb[j] = xc[j]; }
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) { ua[j] = ub[j]; ub[j] = uc[j]; }
#pragma unroll
        for (int j = 0; j < kS; ++j) { gprev[j] = gcur[j]; dmprev[j] = dmcur[j]; }
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
int nhwc_images_per_group(int B, int C) {
  const long wgs = (long)B * ((C + kWave - 1) / kWave);
  return (int)std::max(1L, std::min(8L, wgs / 2048));
}

#define MRLA_DISPATCH_AO_N(TT, ACT, HASO, CALL)                          \
  if (ACT) { if (HASO) { CALL(TT, true, true); } else { CALL(TT, true, false); } } \
  else     { if (HASO) { CALL(TT, false, true); } else { CALL(TT, false, false); } }
#define MRLA_DISPATCH_T_N(DT, ACT, HASO, CALL)                       \
  switch (DT) {                                                      \
    case MRLA_F32:  MRLA_DISPATCH_AO_N(float, ACT, HASO, CALL) break;  \
    case MRLA_BF16: MRLA_DISPATCH_AO_N(bf16_t, ACT, HASO, CALL) break; \
    case MRLA_F16:  MRLA_DISPATCH_AO_N(f16_t, ACT, HASO, CALL) break;  \
    default: return MRLA_EINVAL;                                     \
  }

struct NhwcLaunch { dim3 grid, block; size_t lds; int BG; };
static NhwcLaunch nhwc_launch(int B, int C, int W, int nred) {
  NhwcLaunch L;
  const int nstrips = (W + kS - 1) / kS;
  const int nwaves = std::min(nstrips, kMaxStrips);
  L.BG = nhwc_images_per_group(B, C);
  L.grid = dim3((C + kWave - 1) / kWave, (B + L.BG - 1) / L.BG);
  L.block = dim3(nwaves * kWave);
  L.lds = nwaves > 1 ? (size_t)nwaves * nred * kWave * sizeof(float) : 0;
  return L;
}

int launch_light_stats_fwd_nhwc(const void* x, const void* o, const float* wv, float* mom, void* xout, int B, int C,
                                int H, int W, int dtype, int act, hipStream_t st) {
  const NhwcLaunch L = nhwc_launch(B, C, W, M_N);
#define CALL_F(T, A, O, F)                                                                                          \
  hipLaunchKernelGGL((light_stats_fwd_nhwc<T, A, O, F>), L.grid, L.block, L.lds, st, (const T*)x, (const T*)o, wv, mom, \
                     (T*)xout, B, C, H, W, L.BG);
#define CALL(T, A, O)                                                        \
  {                                                                          \
    if (xout) { if (O) { CALL_F(T, A, true, true) } else return MRLA_EINVAL; } \
    else { CALL_F(T, A, O, false) }                                          \
  }
  MRLA_DISPATCH_T_N(dtype, act, o != nullptr, CALL)
#undef CALL
#undef CALL_F
  return hip_status(hipGetLastError());
}

int launch_light_apply_fwd_nhwc(const void* x, const void* o, const float* wv, const float* gate, const float* sc,
                                const float* sh, const float* lam, const float* dp, void* out, int B, int C, int H,
                                int W, int d, int res, int dtype, int act, hipStream_t st) {
  const NhwcLaunch L = nhwc_launch(B, C, W, 1);
#define CALL(T, A, O)                                                                                               \
  hipLaunchKernelGGL((light_apply_fwd_nhwc<T, A, O>), L.grid, L.block, 0, st, (const T*)x, (const T*)o, wv, gate, sc, \
                     sh, lam, dp, (T*)out, B, C, H, W, L.BG, d, res);
  MRLA_DISPATCH_T_N(dtype, act, o != nullptr, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_light_stats_bwd_nhwc(const void* dout, const void* x, const void* o, const float* wv, float* bmom, int B,
                                int C, int H, int W, int dtype, int act, hipStream_t st) {
  const NhwcLaunch L = nhwc_launch(B, C, W, D_N);
#define CALL(T, A, O)                                                                                               \
  hipLaunchKernelGGL((light_stats_bwd_nhwc<T, A, O>), L.grid, L.block, L.lds, st, (const T*)dout, (const T*)x,       \
                     (const T*)o, wv, bmom, B, C, H, W, L.BG);
  MRLA_DISPATCH_T_N(dtype, act, o != nullptr, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_light_apply_bwd_nhwc(const void* dout, const void* x, const void* o, const float* wv, const float* gate,
                                const float* cb, const float* lam, const float* dp, const float* dyx, void* dx,
                                void* dprev, float* dwv_part, int B, int C, int H, int W, int d, int res, int relu,
                                int dtype, int act, hipStream_t st) {
  const NhwcLaunch L = nhwc_launch(B, C, W, 9);
#define CALL_R(T, A, O, R)                                                                                          \
  hipLaunchKernelGGL((light_apply_bwd_nhwc<T, A, O, R>), L.grid, L.block, L.lds, st, (const T*)dout, (const T*)x,    \
                     (const T*)o, wv, gate, cb, lam, dp, dyx, (T*)dx, (T*)dprev, dwv_part, B, C, H, W, L.BG, d, res);
#define CALL(T, A, O)                                                                        \
  {                                                                                          \
    if (relu) { if (O && !(A)) { CALL_R(T, false, true, true) } else return MRLA_EINVAL; }   \
    else { CALL_R(T, A, O, false) }                                                          \
  }
  MRLA_DISPATCH_T_N(dtype, act, o != nullptr, CALL)
#undef CALL
#undef CALL_R
  return hip_status(hipGetLastError());
}

}  // namespace mrla
