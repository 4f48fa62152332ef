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
#include <algorithm>

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

// ---- wide row access -------------------------------------------------------------------------------------------
// A 2-byte-per-lane access costs the texture-addresser as much as a 16-byte one, so rows are fetched with 16 B per
// lane (lane = (pixel, 8-channel group): one wave-instruction = 8 pixels x 64 channels) and re-distributed to the
// LANE = CHANNEL compute mapping through a wave-private LDS scratch (ds_write_b128, then one ds_read_u16 per pixel).
// Needs C % 64 == 0 (the chunk is 64 real channels, 16-byte aligned); otherwise the kernels use ldpix().
// bytes per gather / scatter buffer: the widest row piece is kS+4 = 11 pixels x 64 channels, in whole 1 KiB loads
template <typename T> constexpr int scratch_bytes() { return 1024 * ((11 * 64 * (int)sizeof(T) + 1023) / 1024); }

// A row piece in flight: the 16-byte loads have been issued, nothing has been waited for yet.
template <typename T, int NPX>
struct RowLoad {
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int UPP = 64 / VEC;          // lanes per pixel
  static constexpr int PPL = 64 / UPP;          // pixels per wave-instruction
  static constexpr int NL = (NPX + PPL - 1) / PPL;
  u32x4 regs[NL];
  bool live;                                    // wave-uniform: row inside the image
};

// Per-strip lane addressing of a row piece: element offset of this lane's 16 bytes inside an image row and whether
// its pixel exists.  Computed once per strip; per row only a wave-uniform row pointer is added.
template <typename T, int NPX>
struct RowAddr {
  int off[RowLoad<T, NPX>::NL];
  bool ok[RowLoad<T, NPX>::NL];
};
template <typename T, int NPX>
__device__ __forceinline__ void make_row_addr(RowAddr<T, NPX>& a, int col0, int W, int C, int cbase, int lane) {
  typedef RowLoad<T, NPX> Q;
  const int px = lane / Q::UPP, part = lane - px * Q::UPP;
#pragma unroll
  for (int l = 0; l < Q::NL; ++l) {
    const int p = l * Q::PPL + px, col = col0 + p;
    a.ok[l] = p < NPX && col >= 0 && col < W;
    a.off[l] = a.ok[l] ? col * C + cbase + part * Q::VEC : 0;
  }
}

template <typename T, int NPX>
__device__ __forceinline__ void issue_row(RowLoad<T, NPX>& q, const T* __restrict__ img, int r, int H, int rowstride,
                                          const RowAddr<T, NPX>& a) {
  typedef RowLoad<T, NPX> Q;
  static_assert(Q::NL * 1024 <= scratch_bytes<T>(), "scratch too small");
  q.live = r >= 0 && r < H;
  const T* rowp = img + (size_t)(q.live ? r : 0) * rowstride;          // wave-uniform
#pragma unroll
  for (int l = 0; l < Q::NL; ++l) {
    q.regs[l] = (u32x4){0u, 0u, 0u, 0u};
    if (q.live && a.ok[l]) q.regs[l] = *reinterpret_cast<const u32x4*>(rowp + (unsigned)a.off[l]);
  }
}

template <typename T, int NPX>
__device__ __forceinline__ void finish_row(const RowLoad<T, NPX>& q, int lane, T* __restrict__ scratch, float (&out)[NPX]) {
  typedef RowLoad<T, NPX> Q;
  if (!q.live) {                                // wave-uniform
#pragma unroll
    for (int j = 0; j < NPX; ++j) out[j] = 0.f;
    return;
  }
  u32x4* s4 = reinterpret_cast<u32x4*>(scratch);
#pragma unroll
  for (int l = 0; l < Q::NL; ++l) s4[l * kWave + lane] = q.regs[l];
#pragma unroll
  for (int j = 0; j < NPX; ++j) out[j] = to_f(scratch[j * kWave + lane]);
}

template <typename T, int NPX>
__device__ __forceinline__ void gather_row(const T* __restrict__ img, int r, int col0, int H, int W, int C, int cbase,
                                           int lane, T* __restrict__ scratch, float (&out)[NPX]) {
  RowLoad<T, NPX> q;
  RowAddr<T, NPX> a;
  make_row_addr<T, NPX>(a, col0, W, C, cbase, lane);
  issue_row<T, NPX>(q, img, r, H, W * C, a);
  finish_row<T, NPX>(q, lane, scratch, out);
}

// lane = channel values v[j] of pixels col0 .. col0+npx-1 of row r -> global, 16 B per lane
template <typename T, int NPX>
__device__ __forceinline__ void scatter_row(T* __restrict__ img, int r, int col0, int npx, int W, int C, int cbase,
                                            int lane, T* __restrict__ scratch, const float (&v)[NPX]) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int UPP = 64 / VEC;
  constexpr int PPL = 64 / UPP;
  constexpr int NL = (NPX + PPL - 1) / PPL;
#pragma unroll
  for (int j = 0; j < NPX; ++j) scratch[j * kWave + lane] = from_f<T>(v[j]);
  const u32x4* s4 = reinterpret_cast<const u32x4*>(scratch);
  const int px = lane / UPP, part = lane - px * UPP;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const int p = l * PPL + px;
    if (p < npx)
      *reinterpret_cast<u32x4*>(img + ((size_t)r * W + col0 + p) * C + cbase + part * VEC) = s4[l * kWave + lane];
  }
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

// Row readers / writers used by the kernels: WIDE -> 16-byte accesses through the wave-private scratch, else ldpix().
template <typename T, bool WIDE, int NPX>
__device__ __forceinline__ void read_row(const T* __restrict__ img, int r, int col0, int H, int W, int C, int cbase,
                                         int cc, int lane, T* __restrict__ scratch, float (&out)[NPX]) {
  if constexpr (WIDE) {
    gather_row<T, NPX>(img, r, col0, H, W, C, cbase, lane, scratch, out);
  } else {
#pragma unroll
    for (int j = 0; j < NPX; ++j) out[j] = ldpix(img, r, col0 + j, H, W, C, cc);
  }
}
template <typename T, bool WIDE, int NPX>
__device__ __forceinline__ void write_row(T* __restrict__ img, int r, int col0, int npx, int W, int C, int cbase, int c,
                                          bool cv, int lane, T* __restrict__ scratch, const float (&v)[NPX]) {
  if constexpr (WIDE) {
    scatter_row<T, NPX>(img, r, col0, npx, W, C, cbase, lane, scratch, v);
  } else {
#pragma unroll
    for (int j = 0; j < NPX; ++j)
      if (j < npx && cv) img[((size_t)r * W + col0 + j) * C + c] = from_f<T>(v[j]);
  }
}

#define MRLA_NHWC_PROLOGUE(NRED)                                                                          \
  extern __shared__ __align__(16) unsigned char smem_raw[];                                               \
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;    \
  float* red = reinterpret_cast<float*>(smem_raw);                                                        \
  T* scr = reinterpret_cast<T*>(smem_raw + (size_t)nwaves * (NRED) * kWave * sizeof(float) +              \
                                (size_t)wave * kScrBufs * scratch_bytes<T>());                                 \
  const int cbase = blockIdx.x * kWave;                                                                   \
  const int c = cbase + lane;                                                                             \
  const bool cv = c < C;                                                                                  \
  const int cc = cv ? c : C - 1;                                                                          \
  const int nstrips = (W + kS - 1) / kS;                                                                  \
  (void)red; (void)scr;
// scratch buffers per wave: one for gathers, one for scatters (LDS operations of a wave execute in order, so a buffer
// can be re-filled right after its previous contents were read back)
constexpr int kScrBufs = 2;
#define SCR(i) (scr + ((i) >= 3 ? 1 : 0) * (scratch_bytes<T>() / (int)sizeof(T)))

// ------------------------------------------------------------------------------------------------
// forward statistics (+ optional fused producer x = relu(pre + o), or x = relu((psc*pre + psh) + o) when the
// per-channel affine of the BatchNorm in front (bn3) is handed over instead of being applied in a pass of its own)
// ------------------------------------------------------------------------------------------------
template <typename T, bool GELU, bool HAS_O, bool FUSE, bool WIDE>
__global__ __launch_bounds__(kMaxStrips * kWave) void light_stats_fwd_nhwc(
    const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv, float* __restrict__ mom,
    T* __restrict__ xout, const float* __restrict__ psc, const float* __restrict__ psh, int B, int C, int H, int W,
    int BG) {
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
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          if (j < nc) {
            float v = conv_at(w, ra, rb, rc, j);
            if (GELU) v = gelu_f(v);
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
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) { ra[j] = rb[j]; rb[j] = rc[j]; ob[j] = oc[j]; }
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
        // dU[rr] on columns -1 .. kS (zero outside the image)
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) {
          const int col = s0 - 1 + j;
          const bool in = rr < H && col >= 0 && col < W;                    // wave-uniform
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

struct NhwcLaunch { dim3 grid, block; size_t lds; int BG; bool wide; };
static NhwcLaunch nhwc_launch(int B, int C, int W, int nred, int dtype) {
  NhwcLaunch L;
  const int nstrips = (W + kS - 1) / kS;
  const int nwaves = std::min(nstrips, kMaxStrips);
  L.BG = nhwc_images_per_group(B, C);
  L.grid = dim3((C + kWave - 1) / kWave, (B + L.BG - 1) / L.BG);
  L.block = dim3(nwaves * kWave);
  L.wide = (C % kWave) == 0;
  const size_t sb = dtype == MRLA_F32 ? scratch_bytes<float>() : scratch_bytes<bf16_t>();
  L.lds = (size_t)nwaves * nred * kWave * sizeof(float) + (L.wide ? (size_t)nwaves * kScrBufs * sb : 0);
  return L;
}
template <typename K>
static hipError_t set_lds_n(K kernel, size_t bytes) {
  if (bytes <= 48 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

int launch_light_stats_fwd_nhwc(const void* x, const void* o, const float* wv, float* mom, void* xout,
                                const float* psc, const float* psh, int B, int C, int H, int W, int dtype, int act,
                                hipStream_t st) {
  const NhwcLaunch L = nhwc_launch(B, C, W, M_N, dtype);
#define CALL_W(T, A, O, F, WD)                                                                                       \
  {                                                                                                                  \
    if (set_lds_n(light_stats_fwd_nhwc<T, A, O, F, WD>, L.lds) != hipSuccess) return MRLA_EHIP;                        \
    hipLaunchKernelGGL((light_stats_fwd_nhwc<T, A, O, F, WD>), L.grid, L.block, L.lds, st, (const T*)x, (const T*)o,  \
                       wv, mom, (T*)xout, psc, psh, B, C, H, W, L.BG);                                               \
  }
#define CALL_F(T, A, O, F) { if (L.wide) CALL_W(T, A, O, F, true) else CALL_W(T, A, O, F, false) }
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
#define CALL_W(T, A, O, WD)                                                                                          \
  {                                                                                                                  \
    if (set_lds_n(light_apply_fwd_nhwc<T, A, O, WD>, L.lds) != hipSuccess) return MRLA_EHIP;                           \
    hipLaunchKernelGGL((light_apply_fwd_nhwc<T, A, O, WD>), L.grid, L.block, L.lds, st, (const T*)x, (const T*)o, wv, \
                       gate, sc, sh, lam, dp, (T*)out, B, C, H, W, L.BG, d, res);                                    \
  }
#define CALL(T, A, O) { if (L.wide) CALL_W(T, A, O, true) else CALL_W(T, A, O, false) }
  MRLA_DISPATCH_T_N(dtype, act, o != nullptr, CALL)
#undef CALL
#undef CALL_W
  return hip_status(hipGetLastError());
}

int launch_light_stats_bwd_nhwc(const void* dout, const void* x, const void* o, const float* wv, float* bmom, int B,
                                int C, int H, int W, int dtype, int act, hipStream_t st) {
  const NhwcLaunch L = nhwc_launch(B, C, W, D_N, dtype);
#define CALL_W(T, A, O, WD)                                                                                          \
  {                                                                                                                  \
    if (set_lds_n(light_stats_bwd_nhwc<T, A, O, WD>, L.lds) != hipSuccess) return MRLA_EHIP;                           \
    hipLaunchKernelGGL((light_stats_bwd_nhwc<T, A, O, WD>), L.grid, L.block, L.lds, st, (const T*)dout, (const T*)x,  \
                       (const T*)o, wv, bmom, B, C, H, W, L.BG);                                                     \
  }
#define CALL(T, A, O) { if (L.wide) CALL_W(T, A, O, true) else CALL_W(T, A, O, false) }
  MRLA_DISPATCH_T_N(dtype, act, o != nullptr, CALL)
#undef CALL
#undef CALL_W
  return hip_status(hipGetLastError());
}

int launch_light_apply_bwd_nhwc(const void* dout, const void* x, const void* o, const float* wv, const float* gate,
                                const float* cb, const float* lam, const float* dp, const float* dyx, void* dx,
                                void* dprev, float* dwv_part, int B, int C, int H, int W, int d, int res, int relu,
                                int dtype, int act, hipStream_t st) {
  const NhwcLaunch L = nhwc_launch(B, C, W, 9, dtype);
#define CALL_W(T, A, O, R, WD)                                                                                       \
  {                                                                                                                  \
    if (set_lds_n(light_apply_bwd_nhwc<T, A, O, R, WD>, L.lds) != hipSuccess) return MRLA_EHIP;                        \
    hipLaunchKernelGGL((light_apply_bwd_nhwc<T, A, O, R, WD>), L.grid, L.block, L.lds, st, (const T*)dout,            \
                       (const T*)x, (const T*)o, wv, gate, cb, lam, dp, dyx, (T*)dx, (T*)dprev, dwv_part, B, C, H, W, \
                       L.BG, d, res);                                                                                \
  }
#define CALL_R(T, A, O, R) { if (L.wide) CALL_W(T, A, O, R, true) else CALL_W(T, A, O, R, false) }
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
