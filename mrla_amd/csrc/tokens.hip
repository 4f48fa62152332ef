// MRLA-light for token sequences (DeiT): x[b, n, c] with n = 1 + side*side, channel-contiguous (NHWC).
//
// Reference: deit/deit_mrla_light.py:157-180 (mrlal_layer: GAP -> Wq/Wk conv1d -> sigmoid gate -> GELU(dwconv3x3))
// and :194-209 (mrlal_module: LayerNorm of x_t and o_{t-1}, cls split, token <-> map permutes, lambda_t, cat),
// :234 (block residual x + mrla(x, o)).
//
//   xn = LN_x(x), on = LN_o(o)                       per token over c (eps = 1e-6)
//   y[b,c] = mean_{i>=1} xn[b,i,c] ; a[b,g] = sigmoid(<Wq*y, Wk*y>_g / sqrt(d))       (gate.hip kernels)
//   out[b,0,:]  = res*x[b,0,:] + xn[b,0,:]                                              cls row
//   out[b,i,:]  = res*x[b,i,:] + a[b,g]*gelu(dwconv3x3(xn map)[i,:]) + lam*on[b,i,:]    i >= 1
//
// Kernels: token_norm_pool (per image: LN statistics of both inputs, pooled descriptor), token_apply_fwd,
// token_apply_bwd (per image x 16-channel chunk: the normalised map lives in LDS as fp32, lanes = channels so the 3x3
// neighbours are plain LDS reads), token_ln_bwd (per token: both LayerNorm backward passes).  The permutes / split / cat
// of the reference never touch memory: they are index arithmetic here.
// The backward is ONE pass over the map: token_apply_bwd also takes bmom = sum dOut*gelu(U) for the gate backward, whose
// result dy (a per-(image, channel) constant on the map rows of dxn) enters afterwards: the gate backward completes the
// LayerNorm parameter partials with it and token_ln_bwd adds it to dxn as it reads it.
#include <algorithm>

#include "mrla_device.h"
#include "mrla_kernels.h"

namespace mrla {

// stats[b, i, 0..3] = mean_x, rstd_x, mean_o, rstd_o
enum { S_MX = 0, S_RX = 1, S_MO = 2, S_RO = 3, S_N = 4 };
// per-(image, channel) parameter-gradient partials written by token_apply_bwd
enum { Q_WV = 0, Q_LAM = 9, Q_LNXW = 10, Q_LNXB = 11, Q_LNOW = 12, Q_LNOB = 13, Q_H = 14, Q_N = 15 };
static_assert(Q_N == kTokParts && Q_LNXW == kTokPartLnxW && Q_LNXB == kTokPartLnxB && Q_H == kTokPartHat,
              "the gate backward (gate.hip) patches these slots");

// ------------------------------------------------------------------------------------------------
// LayerNorm statistics of x and o (one wave per token, the row held in registers between the two passes of the
// two-pass variance), then the pooled descriptor of LN_x(x) over the map tokens (one thread per channel)
// ------------------------------------------------------------------------------------------------
constexpr int kTokRowRegs = 16;      // channels per lane held in registers: C <= 64 * 16

template <typename T>
__global__ __launch_bounds__(kThreads) void token_stats_kernel(const T* __restrict__ x, const T* __restrict__ o, float eps,
                                                               float* __restrict__ stats, int ntok, int C) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const float invc = 1.0f / (float)C;
  for (int tok = blockIdx.x * kWaves + wave; tok < ntok; tok += gridDim.x * kWaves) {
    const T* xr = x + (size_t)tok * C;
    float xv[kTokRowRegs], ov[kTokRowRegs];
    float sx = 0.f, so = 0.f;
    if (!o) {                           // MRLA-base on tokens has no o_{t-1}: only x is read, its statistics fill both pairs
#pragma unroll
      for (int k = 0; k < kTokRowRegs; ++k) {
        const int c = lane + k * kWave;
        xv[k] = c < C ? to_f(xr[c]) : 0.f;
        sx += xv[k];
      }
      const float mx = wave_sum(sx) * invc;
      float vx = 0.f;
#pragma unroll
      for (int k = 0; k < kTokRowRegs; ++k) {
        const float dxv = lane + k * kWave < C ? xv[k] - mx : 0.f;
        vx = fmaf(dxv, dxv, vx);
      }
      const float rx = rsqrtf(wave_sum(vx) * invc + eps);
      if (lane == 0) *reinterpret_cast<float4*>(stats + (size_t)tok * S_N) = make_float4(mx, rx, mx, rx);
      continue;
    }
    const T* orow = o + (size_t)tok * C;
#pragma unroll
    for (int k = 0; k < kTokRowRegs; ++k) {
      const int c = lane + k * kWave;
      xv[k] = c < C ? to_f(xr[c]) : 0.f;
      ov[k] = c < C ? to_f(orow[c]) : 0.f;
      sx += xv[k]; so += ov[k];
    }
    const float mx = wave_sum(sx) * invc, mo = wave_sum(so) * invc;
    float vx = 0.f, vo = 0.f;
#pragma unroll
    for (int k = 0; k < kTokRowRegs; ++k) {
      const bool in = lane + k * kWave < C;
      const float dxv = in ? xv[k] - mx : 0.f, dov = in ? ov[k] - mo : 0.f;
      vx = fmaf(dxv, dxv, vx);
      vo = fmaf(dov, dov, vo);
    }
    const float rx = rsqrtf(wave_sum(vx) * invc + eps), ro = rsqrtf(wave_sum(vo) * invc + eps);
    if (lane == 0) *reinterpret_cast<float4*>(stats + (size_t)tok * S_N) = make_float4(mx, rx, mo, ro);
  }
}

// 16 bytes per lane (round 4; it was one element per lane and load: the address unit pays per instruction, not per byte):
// a lane holds VEC = 16 / sizeof(T) consecutive channels per pass, a wave-pass covers 64 * VEC channels (DeiT-tiny's 192 fp32
// channels: one pass, 48 lanes).  C % VEC != 0 keeps the element-wise form below.
constexpr int kTokVecPasses = 4;     // C <= 64 * VEC * 4 (1024 fp32 / 2048 16-bit channels)

template <typename T>
__global__ __launch_bounds__(kThreads) void token_stats_vec_kernel(const T* __restrict__ x, const T* __restrict__ o, float eps,
                                                                   float* __restrict__ stats, int ntok, int C) {
  constexpr int VEC = 16 / sizeof(T);
  typedef T VT __attribute__((ext_vector_type(VEC)));
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const float invc = 1.0f / (float)C;
  const int nv = C / VEC;                               // 16-byte vectors per token row
  for (int tok = blockIdx.x * kWaves + wave; tok < ntok; tok += gridDim.x * kWaves) {
    const VT* xr = reinterpret_cast<const VT*>(x + (size_t)tok * C);
    const VT* orow = o ? reinterpret_cast<const VT*>(o + (size_t)tok * C) : nullptr;
    float xv[kTokVecPasses][VEC], ov[kTokVecPasses][VEC];
    float sx = 0.f, so = 0.f;
#pragma unroll
    for (int k = 0; k < kTokVecPasses; ++k) {
      const int v = lane + k * kWave;
      const bool in = v < nv;
      VT a, b;
      if (in) a = xr[v];
      if (in && o) b = orow[v];
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        xv[k][i] = in ? to_f(a[i]) : 0.f;
        ov[k][i] = (in && o) ? to_f(b[i]) : 0.f;
        sx += xv[k][i]; so += ov[k][i];
      }
    }
    const float mx = wave_sum(sx) * invc, mo = o ? wave_sum(so) * invc : 0.f;
    float vx = 0.f, vo = 0.f;
#pragma unroll
    for (int k = 0; k < kTokVecPasses; ++k) {
      const bool in = lane + k * kWave < nv;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float dxv = in ? xv[k][i] - mx : 0.f, dov = in ? ov[k][i] - mo : 0.f;
        vx = fmaf(dxv, dxv, vx);
        vo = fmaf(dov, dov, vo);
      }
    }
    const float rx = rsqrtf(wave_sum(vx) * invc + eps);
    const float ro = o ? rsqrtf(wave_sum(vo) * invc + eps) : rx;
    // (MRLA-base on tokens has no o_{t-1}: x's statistics fill both pairs)
    if (lane == 0) *reinterpret_cast<float4*>(stats + (size_t)tok * S_N) = make_float4(mx, rx, o ? mo : mx, ro);
  }
}

// mom[b,c,0] = wx[c] * sum_{i>=1} (x[b,i,c] - mean_i) * rstd_i + (n-1) * bx[c]   (other slots 0).  grid (C/64.., b);
// the four waves of a workgroup take every fourth token each (a single wave per (image, 64 channels) walked 196 tokens
// in 49 dependent steps: 20 us for a 39 MB read), partial sums combined in wave order.
template <typename T>
__global__ __launch_bounds__(kThreads) void token_pool_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                              const float* __restrict__ wx, const float* __restrict__ bx,
                                                              float* __restrict__ mom, int n, int C) {
  __shared__ float part[kWaves][kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int b = blockIdx.y, c = blockIdx.x * kWave + lane;
  const bool live = c < C;
  const T* xb = x + (size_t)b * n * C + (live ? c : 0);
  const float* sb = stats + (size_t)b * n * S_N;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // four independent chains per wave, summed in a fixed order
  int i = 1 + wave;
  for (; i + 3 * kWaves < n; i += 4 * kWaves) {
    const int i1 = i + kWaves, i2 = i + 2 * kWaves, i3 = i + 3 * kWaves;
    s0 = fmaf(to_f(xb[(size_t)i * C]) - sb[i * S_N + S_MX], sb[i * S_N + S_RX], s0);
    s1 = fmaf(to_f(xb[(size_t)i1 * C]) - sb[i1 * S_N + S_MX], sb[i1 * S_N + S_RX], s1);
    s2 = fmaf(to_f(xb[(size_t)i2 * C]) - sb[i2 * S_N + S_MX], sb[i2 * S_N + S_RX], s2);
    s3 = fmaf(to_f(xb[(size_t)i3 * C]) - sb[i3 * S_N + S_MX], sb[i3 * S_N + S_RX], s3);
  }
  for (; i < n; i += kWaves) s0 = fmaf(to_f(xb[(size_t)i * C]) - sb[i * S_N + S_MX], sb[i * S_N + S_RX], s0);
  part[wave][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave != 0 || !live) return;
  float s = part[0][lane];
#pragma unroll
  for (int k = 1; k < kWaves; ++k) s += part[k][lane];
  float* m = mom + ((size_t)b * C + c) * M_REC;
  m[M_PV] = 0.f; m[M_PO] = 0.f;
  // slot 0 holds hw * y so that the shared gate kernels' y = Sx / hw is the LN-affine pooled value
  m[M_SX] = fmaf(wx[c], s, bx[c] * (float)(n - 1));
  m[M_SV] = 0.f; m[M_SO] = 0.f; m[M_SVV] = 0.f; m[M_SVO] = 0.f; m[M_SOO] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// shared tile code: normalised map of one (image, channel chunk) in LDS, fp32, ZERO-PADDED by one pixel on every side:
// layout [(side+2)][(side+2)][CC], so the nine taps need no edge test.  A thread walks tokens i = tg, tg+ntg, ...; its
// (row, column) is advanced without divisions by TokenPos.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tile_elems(int side, int CC) { return (side + 2) * (side + 2) * CC; }

struct TokenPos {
  int r, col, step_r, step_c, side;
  __device__ __forceinline__ TokenPos(int first, int stride, int side_) : side(side_) {
    r = first / side_; col = first - r * side_;
    step_r = stride / side_; step_c = stride - step_r * side_;
  }
  __device__ __forceinline__ void advance() {
    r += step_r; col += step_c;
    if (col >= side) { col -= side; ++r; }
  }
  // index of the pixel in the padded tile (channels innermost, CC of them)
  __device__ __forceinline__ int at(int CC, int cc) const { return ((r + 1) * (side + 2) + col + 1) * CC + cc; }
};

__device__ __forceinline__ void zero_tile_border(float* __restrict__ tile, int side, int CC, int tid) {
  const int P = side + 2;
  for (int k = tid; k < 4 * P * CC; k += kThreads) {          // two rows and two columns of P pixels each
    const int which = k / (P * CC), rem = k - which * P * CC, p = rem / CC, cc = rem - p * CC;
    const int pix = which == 0 ? p : which == 1 ? (P - 1) * P + p : which == 2 ? p * P : p * P + P - 1;
    tile[pix * CC + cc] = 0.f;
  }
}

template <typename T>
__device__ __forceinline__ void load_xn_tile(float* __restrict__ tile, const T* __restrict__ x,
                                             const float* __restrict__ stats, const float* __restrict__ wx,
                                             const float* __restrict__ bx, int b, int n, int C, int c0, int CC, int side,
                                             int tid) {
  const int hw = n - 1;
  const int cc = tid % CC, tg = tid / CC, ntg = kThreads / CC;      // lanes = channels: coalesced along c
  const float w = wx[c0 + cc], bb = bx[c0 + cc];
  zero_tile_border(tile, side, CC, tid);
  TokenPos pos(tg, ntg, side);
  for (int i = tg; i < hw; i += ntg, pos.advance()) {
    const float2 s = *reinterpret_cast<const float2*>(stats + ((size_t)b * n + i + 1) * S_N);   // (mean, rstd): uniform per row
    const float xv = to_f(x[((size_t)b * n + i + 1) * C + c0 + cc]);
    tile[pos.at(CC, cc)] = fmaf((xv - s.x) * s.y, w, bb);
  }
}

// sum_k w[k] * tile[centre + offset_k]: `centre` = TokenPos::at(), rs = (side + 2) * CC
__device__ __forceinline__ float conv9_tile(const float* __restrict__ tile, const float (&w)[9], int centre, int rs, int CC) {
  const float* t = tile + centre;
  float u = w[0] * t[-rs - CC];
  u = fmaf(w[1], t[-rs], u); u = fmaf(w[2], t[-rs + CC], u);
  u = fmaf(w[3], t[-CC], u); u = fmaf(w[4], t[0], u); u = fmaf(w[5], t[CC], u);
  u = fmaf(w[6], t[rs - CC], u); u = fmaf(w[7], t[rs], u); u = fmaf(w[8], t[rs + CC], u);
  return u;
}

// ------------------------------------------------------------------------------------------------
// forward apply
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void token_apply_fwd_kernel(
    const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ stats, const float* __restrict__ wx,
    const float* __restrict__ bx, const float* __restrict__ wo, const float* __restrict__ bo,
    const float* __restrict__ wv, const float* __restrict__ gate, const float* __restrict__ lam, T* __restrict__ out,
    int n, int C, int side, int d, int CC, int res) {
  extern __shared__ float tile[];
  const int b = blockIdx.y, c0 = blockIdx.x * CC, tid = threadIdx.x;
  const int hw = n - 1;
  load_xn_tile(tile, x, stats, wx, bx, b, n, C, c0, CC, side, tid);
  __syncthreads();
  const int cc = tid % CC, tg = tid / CC, ntg = kThreads / CC;
  const int c = c0 + cc;
  const int rs = (side + 2) * CC;
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[c * 9 + k];
  const float a = gate[(size_t)b * (C / d) + c / d];
  const float lm = lam[c], wo_c = wo[c], bo_c = bo[c];
  TokenPos pos(tg, ntg, side);
  for (int i = tg; i < hw; i += ntg, pos.advance()) {
    const float u = conv9_tile(tile, w, pos.at(CC, cc), rs, CC);
    const size_t g = ((size_t)b * n + i + 1) * C + c;
    const float* s = stats + ((size_t)b * n + i + 1) * S_N;
    const float on = fmaf((to_f(o[g]) - s[S_MO]) * s[S_RO], wo_c, bo_c);
    float y = fmaf(a, gelu_f(u), lm * on);
    if (res) y += to_f(x[g]);
    out[g] = from_f<T>(y);
  }
  if (tg == 0) {                                          // cls row passes LN_x(x) through
    const size_t g = (size_t)b * n * C + c;
    const float* s = stats + (size_t)b * n * S_N;
    const float xv = to_f(x[g]);
    float y = fmaf((xv - s[S_MX]) * s[S_RX], wx[c], bx[c]);
    if (res) y += xv;
    out[g] = from_f<T>(y);
  }
}

// ------------------------------------------------------------------------------------------------
// backward apply: dxn' (gradient wrt LN_x(x) without the pooled-descriptor term, all rows), bmom and the per-(image,
// channel) parameter partials
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void token_apply_bwd_kernel(
    const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ stats,
    const float* __restrict__ wx, const float* __restrict__ bx, const float* __restrict__ wo,
    const float* __restrict__ bo, const float* __restrict__ wv, const float* __restrict__ gate,
    const float* __restrict__ lam, float* __restrict__ dxn /*[b,n,c] fp32*/, float* __restrict__ part /*[b,c,Q_N]*/,
    float* __restrict__ bmom, int n, int C, int side, int d, int CC) {
  extern __shared__ float tile[];
  const int hw = n - 1;
  float* dus = tile + tile_elems(side, CC);          // dU, same padded layout
  float* red = dus + tile_elems(side, CC);           // [ntg][CC][Q_N + 1]
  const int b = blockIdx.y, c0 = blockIdx.x * CC, tid = threadIdx.x;
  load_xn_tile(tile, x, stats, wx, bx, b, n, C, c0, CC, side, tid);
  zero_tile_border(dus, side, CC, tid);
  __syncthreads();
  const int cc = tid % CC, tg = tid / CC, ntg = kThreads / CC;
  const int c = c0 + cc;
  const int rs = (side + 2) * CC;
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[c * 9 + k];
  const float a = gate[(size_t)b * (C / d) + c / d];
  const float lm = lam[c], wo_c = wo[c], bo_c = bo[c];
  float q[Q_N], qb = 0.f;
#pragma unroll
  for (int k = 0; k < Q_N; ++k) q[k] = 0.f;
  {
    TokenPos pos(tg, ntg, side);
    for (int i = tg; i < hw; i += ntg, pos.advance()) {
      const int ctr = pos.at(CC, cc);
      const float u = conv9_tile(tile, w, ctr, rs, CC);
      const size_t g = ((size_t)b * n + i + 1) * C + c;
      const float go = to_f(dout[g]);
      const float du = a * go * gelu_grad_f(u);
      qb = fmaf(go, gelu_f(u), qb);
      dus[ctr] = du;
      const float* s = stats + ((size_t)b * n + i + 1) * S_N;
      const float ohat = (to_f(o[g]) - s[S_MO]) * s[S_RO];
      q[Q_LAM] = fmaf(go, fmaf(ohat, wo_c, bo_c), q[Q_LAM]);
      q[Q_LNOW] = fmaf(lm * go, ohat, q[Q_LNOW]);
      q[Q_LNOB] = fmaf(lm, go, q[Q_LNOB]);
      // dWv[di][dj] += dU[i] * xn[i + (di, dj)]   (the padded border of xn is zero)
      const float* t = tile + ctr;
      q[Q_WV + 0] = fmaf(du, t[-rs - CC], q[Q_WV + 0]); q[Q_WV + 1] = fmaf(du, t[-rs], q[Q_WV + 1]);
      q[Q_WV + 2] = fmaf(du, t[-rs + CC], q[Q_WV + 2]); q[Q_WV + 3] = fmaf(du, t[-CC], q[Q_WV + 3]);
      q[Q_WV + 4] = fmaf(du, t[0], q[Q_WV + 4]);        q[Q_WV + 5] = fmaf(du, t[CC], q[Q_WV + 5]);
      q[Q_WV + 6] = fmaf(du, t[rs - CC], q[Q_WV + 6]);  q[Q_WV + 7] = fmaf(du, t[rs], q[Q_WV + 7]);
      q[Q_WV + 8] = fmaf(du, t[rs + CC], q[Q_WV + 8]);
    }
  }
  __syncthreads();
  {
    TokenPos pos(tg, ntg, side);
    for (int i = tg; i < hw; i += ntg, pos.advance()) {
      // dxn'[i] = sum_{di,dj} wv[di][dj] * dU[i - (di, dj)]   (the padded border of dU is zero)
      const float* t = dus + pos.at(CC, cc);
      float s9 = w[0] * t[rs + CC]; s9 = fmaf(w[1], t[rs], s9); s9 = fmaf(w[2], t[rs - CC], s9);
      s9 = fmaf(w[3], t[CC], s9);      s9 = fmaf(w[4], t[0], s9);  s9 = fmaf(w[5], t[-CC], s9);
      s9 = fmaf(w[6], t[-rs + CC], s9); s9 = fmaf(w[7], t[-rs], s9); s9 = fmaf(w[8], t[-rs - CC], s9);
      const size_t g = ((size_t)b * n + i + 1) * C + c;
      dxn[g] = s9;
      const float* s = stats + ((size_t)b * n + i + 1) * S_N;
      const float xhat = (to_f(x[g]) - s[S_MX]) * s[S_RX];
      q[Q_LNXW] = fmaf(s9, xhat, q[Q_LNXW]);
      q[Q_LNXB] += s9;
      q[Q_H] += xhat;
    }
  }
  if (tg == 0) {                                              // cls row: module output is LN_x(x) itself
    const size_t g = (size_t)b * n * C + c;
    const float* s = stats + (size_t)b * n * S_N;
    const float dn = to_f(dout[g]);
    dxn[g] = dn;
    q[Q_LNXW] = fmaf(dn, (to_f(x[g]) - s[S_MX]) * s[S_RX], q[Q_LNXW]);
    q[Q_LNXB] += dn;
  }
#pragma unroll
  for (int k = 0; k < Q_N; ++k) red[(tg * CC + cc) * (Q_N + 1) + k] = q[k];
  red[(tg * CC + cc) * (Q_N + 1) + Q_N] = qb;
  __syncthreads();
  if (tg == 0) {
#pragma unroll
    for (int k = 0; k < Q_N + 1; ++k) {
      float s = 0.f;
      for (int t2 = 0; t2 < ntg; ++t2) s += red[(t2 * CC + cc) * (Q_N + 1) + k];
      if (k < Q_N) {
        part[((size_t)b * C + c) * Q_N + k] = s;
      } else {
        float* bm = bmom + ((size_t)b * C + c) * D_N;
        bm[D_D] = 0.f; bm[D_DV] = s; bm[D_DO] = 0.f;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// per token: backward of both LayerNorms.  dx = LN_x^T(dxn' + dy on the map rows) + res*dOut ;  do = LN_o^T(lam*dOut)
// (cls row: 0).  dyx [b, c]: the pooled-descriptor gradient / hw from the gate backward (null: dxn is complete);
// o == null (MRLA-base on tokens): only dx
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void token_ln_bwd_kernel(
    const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ dxn,
    const float* __restrict__ dyx, const float* __restrict__ stats, const float* __restrict__ wx,
    const float* __restrict__ wo, const float* __restrict__ lam, T* __restrict__ dx, T* __restrict__ dprev, int ntok, int n,
    int C, int res) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int tok = blockIdx.x * kWaves + wave;
  if (tok >= ntok) return;
  const int i = tok % n;
  const size_t base = (size_t)tok * C;
  const float* dyb = (dyx && i >= 1) ? dyx + (size_t)(tok / n) * C : nullptr;       // (wave-uniform)
  const float* s = stats + (size_t)tok * S_N;
  const float mx = s[S_MX], rx = s[S_RX], mo = s[S_MO], ro = s[S_RO];
  float s1 = 0.f, s2 = 0.f, t1 = 0.f, t2 = 0.f;
  for (int c = lane; c < C; c += kWave) {
    const float dh = (dxn[base + c] + (dyb ? dyb[c] : 0.f)) * wx[c];
    const float xh = (to_f(x[base + c]) - mx) * rx;
    s1 += dh;
    s2 = fmaf(dh, xh, s2);
    if (o && i >= 1) {
      const float dho = lam[c] * to_f(dout[base + c]) * wo[c];
      const float oh = (to_f(o[base + c]) - mo) * ro;
      t1 += dho;
      t2 = fmaf(dho, oh, t2);
    }
  }
  const float invc = 1.0f / (float)C;
  s1 = __shfl(wave_sum(s1), 0, kWave) * invc;
  s2 = __shfl(wave_sum(s2), 0, kWave) * invc;
  t1 = __shfl(wave_sum(t1), 0, kWave) * invc;
  t2 = __shfl(wave_sum(t2), 0, kWave) * invc;
  for (int c = lane; c < C; c += kWave) {
    const float go = to_f(dout[base + c]);
    const float dh = (dxn[base + c] + (dyb ? dyb[c] : 0.f)) * wx[c];
    const float xh = (to_f(x[base + c]) - mx) * rx;
    float y = rx * (dh - s1 - xh * s2);
    if (res) y += go;
    dx[base + c] = from_f<T>(y);
    if (!o) continue;
    float z = 0.f;
    if (i >= 1) {
      const float dho = lam[c] * go * wo[c];
      const float oh = (to_f(o[base + c]) - mo) * ro;
      z = ro * (dho - t1 - oh * t2);
    }
    dprev[base + c] = from_f<T>(z);
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
#define MRLA_DISPATCH_TT(DT, CALL)       \
  switch (DT) {                          \
    case MRLA_F32:  CALL(float); break;  \
    case MRLA_BF16: CALL(bf16_t); break; \
    case MRLA_F16:  CALL(f16_t); break;  \
    default: return MRLA_EINVAL;         \
  }

// 16-channel chunks: the zero-padded fp32 map tiles are 16 KB each, so 4 (backward: two tiles) to 8 workgroups share a
// CU; measured faster than 32-channel chunks (2-4 per CU) although a row of lanes then moves 64 instead of 128 bytes
static int chunk_for(int C) { return C % 16 == 0 ? 16 : 0; }

template <typename K>
static hipError_t set_lds3(K kernel, size_t bytes) {
  return lds_opt_in(reinterpret_cast<const void*>(kernel), bytes);
}

int launch_token_norm_pool(const void* x, const void* o, const float* wx, const float* bx, float eps, float* stats,
                           float* mom, int B, int n, int C, int dtype, hipStream_t st) {
  if (C > kWave * kTokRowRegs) return MRLA_EUNSUPPORTED;
  const int ntok = B * n;
  const int wgs = std::max(1, std::min((ntok + kWaves - 1) / kWaves, 256 * 32));
  const int vec = 16 / (int)dtype_size(dtype);
  const bool vec_ok = C % vec == 0 && C <= kWave * vec * kTokVecPasses &&
                      ((uintptr_t)x & 15) == 0 && (!o || ((uintptr_t)o & 15) == 0);
#define CALL(TT)                                                                                                     \
  if (vec_ok)                                                                                                        \
    hipLaunchKernelGGL((token_stats_vec_kernel<TT>), dim3(wgs), dim3(kThreads), 0, st, (const TT*)x, (const TT*)o, eps, \
                       stats, ntok, C);                                                                              \
  else                                                                                                               \
    hipLaunchKernelGGL((token_stats_kernel<TT>), dim3(wgs), dim3(kThreads), 0, st, (const TT*)x, (const TT*)o, eps, stats, \
                       ntok, C);                                                                                     \
  hipLaunchKernelGGL((token_pool_kernel<TT>), dim3((C + kWave - 1) / kWave, B), dim3(kThreads), 0, st, (const TT*)x, stats, \
                     wx, bx, mom, n, C);
  MRLA_DISPATCH_TT(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_token_apply_fwd(const void* x, const void* o, const float* stats, const float* wx, const float* bx,
                           const float* wo, const float* bo, const float* wv, const float* gate, const float* lam,
                           void* out, int B, int n, int C, int side, int d, int res, int dtype, hipStream_t st) {
  if (token_nhwc_applies(C))                                   // row-marching kernels (tokens_nhwc.hip)
    return launch_token_apply_fwd_nhwc(x, o, stats, wx, bx, wo, bo, wv, gate, lam, out, B, n, C, side, d, res, dtype, st);
  const int CC = chunk_for(C);
  if (!CC) return MRLA_EUNSUPPORTED;
  const size_t lds = (size_t)(side + 2) * (side + 2) * CC * sizeof(float);
  if (lds > 150 * 1024) return MRLA_EUNSUPPORTED;
  const dim3 grid(C / CC, B);
#define CALL(TT)                                                                                              \
  {                                                                                                           \
    if (set_lds3(token_apply_fwd_kernel<TT>, lds) != hipSuccess) return MRLA_EHIP;                              \
    hipLaunchKernelGGL((token_apply_fwd_kernel<TT>), grid, dim3(kThreads), lds, st, (const TT*)x, (const TT*)o, \
                       stats, wx, bx, wo, bo, wv, gate, lam, (TT*)out, n, C, side, d, CC, res);               \
  }
  MRLA_DISPATCH_TT(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_token_apply_bwd(const void* dout, const void* x, const void* o, const float* stats, const float* wx,
                           const float* bx, const float* wo, const float* bo, const float* wv, const float* gate,
                           const float* lam, float* dxn, float* part, float* bmom, int B, int n, int C, int side, int d,
                           int dtype, hipStream_t st) {
  if (token_nhwc_applies(C))                                   // row-marching kernels (tokens_nhwc.hip)
    return launch_token_apply_bwd_nhwc(dout, x, o, stats, wx, bx, wo, bo, wv, gate, lam, dxn, part, bmom, B, n, C, side, d, dtype, st);
  const int CC = chunk_for(C);
  if (!CC) return MRLA_EUNSUPPORTED;
  const size_t lds = ((size_t)2 * (side + 2) * (side + 2) * CC + (size_t)kThreads * (Q_N + 1)) * sizeof(float);
  if (lds > 150 * 1024) return MRLA_EUNSUPPORTED;
  const dim3 grid(C / CC, B);
#define CALL(TT)                                                                                              \
  {                                                                                                           \
    if (set_lds3(token_apply_bwd_kernel<TT>, lds) != hipSuccess) return MRLA_EHIP;                              \
    hipLaunchKernelGGL((token_apply_bwd_kernel<TT>), grid, dim3(kThreads), lds, st, (const TT*)dout,           \
                       (const TT*)x, (const TT*)o, stats, wx, bx, wo, bo, wv, gate, lam, dxn, part, bmom, n, C, \
                       side, d, CC);                                                                          \
  }
  MRLA_DISPATCH_TT(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_token_ln_bwd(const void* dout, const void* x, const void* o, const float* dxn, const float* dyx,
                        const float* stats, const float* wx, const float* wo, const float* lam, void* dx, void* dprev, int B,
                        int n, int C, int res, int dtype, hipStream_t st) {
  const int ntok = B * n;
  const dim3 grid((ntok + kWaves - 1) / kWaves);
#define CALL(TT)                                                                                              \
  hipLaunchKernelGGL((token_ln_bwd_kernel<TT>), grid, dim3(kThreads), 0, st, (const TT*)dout, (const TT*)x,    \
                     (const TT*)o, dxn, dyx, stats, wx, wo, lam, (TT*)dx, (TT*)dprev, ntok, n, C, res);
  MRLA_DISPATCH_TT(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

}  // namespace mrla
