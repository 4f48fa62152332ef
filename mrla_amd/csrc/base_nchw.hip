// MRLA-base (softmax over depth) kernels for NCHW-contiguous activations (gfx950).
//
// Reference: resnet/models/modules/mrla_base_module.py:54-89 (layer), resnet/models/resnet_mrla_base.py:
// 44-51,120-129 (module + block tail  out = x + DropPath(relu(bn_mrla(attn)))), history loop :254-259.
//
// Stage-resident state (owned by the caller, never re-allocated per layer, no torch.cat):
//   V ring  [b, T, c, h, w]  activation dtype   v_j = dwconv3x3(x_j)            (slot j-1 written by layer j)
//   K ring  [b, T, c]        float32            k_j = corr1d(GAP(x_j), wk)
//   P all   [b, g, T, T]     float32            row t-1 = softmax_j(s * <q_t, k_j>), j < t
//   dA ring [b, T, c, h, w]  activation dtype   dL/d attn_t, written by layer t's backward
//   dK ring [b, T, c]        float32            accumulated dL/d k_j
//
// Forward of layer t (history length t):        HBM traffic in units of N = b*c*h*w elements
//   pool      (light stats kernel, o = null)     read x                                    1
//   gate      q_t, k_t -> K ring, P row          tiny
//   attend    v_t -> ring; attn = sum_j P_j V_j  read x, V_1..t-1; write v_t, attn       t+2
//   bn        batch statistics of attn           tiny
//   tail      out = x + dp*relu(sc*attn + sh)    read x, attn; write out                   3
// Backward of layer t (run for t = Tc .. 1; every later layer has already run):
//   tail stats   sum dz, sum dz*attn             read dOut, attn                           2
//   bn bwd       constants e, f, h               tiny
//   attend bwd   dA_t -> ring; sum_hw dA_t*V_j   read dOut, attn, V_1..t; write dA_t      t+3
//   gate bwd     softmax bwd, dq, dK ring, dy    tiny
//   value bwd    dV_t = sum_{t'>=t} P_t'[t] dA_t'; dx = dOut + dwconv^T(dV_t) + dyx; dWv
//                                                read dA_t..Tc, x, dOut; write dx      Tc-t+4
#include "mrla_device.h"
#include <algorithm>

#include "mrla_kernels.h"
#include "mrla_march.h"

namespace mrla {

// ------------------------------------------------------------------------------------------------
// vector helpers for the "linear domain" (element order of the slab, 16 bytes per lane when aligned)
// ------------------------------------------------------------------------------------------------
template <typename T, int VW> struct Vec { typedef T type __attribute__((ext_vector_type(VW))); };
template <typename T> struct Vec<T, 1> { typedef T type; };

template <typename T, int VW>
__device__ __forceinline__ void ld_vec(const T* __restrict__ p, float (&v)[VW]) {
  if constexpr (VW == 1) {
    v[0] = to_f(p[0]);
  } else {
    typedef typename Vec<T, VW>::type VT;
    const VT t = *reinterpret_cast<const VT*>(p);
#pragma unroll
    for (int i = 0; i < VW; ++i) v[i] = static_cast<float>(t[i]);
  }
}
template <typename T, int VW>
__device__ __forceinline__ void st_vec(T* __restrict__ p, const float (&v)[VW]) {
  if constexpr (VW == 1) {
    p[0] = from_f<T>(v[0]);
  } else {
    typedef typename Vec<T, VW>::type VT;
    VT t;
#pragma unroll
    for (int i = 0; i < VW; ++i) t[i] = static_cast<T>(v[i]);
    *reinterpret_cast<VT*>(p) = t;
  }
}
template <typename T> __device__ __forceinline__ bool aligned16(const T* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct RingGeo { int T; int t; };   // ring capacity, current history length (this layer = slot t-1)

__device__ __forceinline__ size_t ring_off(const SlabGeo& g, int T, int b, int j, int c0) {
  return (((size_t)b * T + j) * g.C + c0) * (size_t)g.HW;
}

// lanes that share one head's softmax row: the largest power of two <= 256 / G, at most a wave
__device__ __forceinline__ int lanes_per_head(int G) {
  int l = kWave;
  while (l > 1 && l * G > kThreads) l >>= 1;
  return l;
}

// ------------------------------------------------------------------------------------------------
// gate forward: one workgroup per image.  q_t, k_t, softmax over the history.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void base_gate_fwd_kernel(
    const float* __restrict__ mom, const float* __restrict__ wq, const float* __restrict__ wk, int ks,
    float* __restrict__ Kring, float* __restrict__ Pall, float* __restrict__ qout, int C, int HW, int d, int T, int t) {
  extern __shared__ float sm[];
  const int p = (ks - 1) / 2;
  float* ys = sm;                 // [C + 2p]
  float* qs = ys + C + 2 * p;     // [C]
  float* kts = qs + C;            // [C] k_t (also written to the ring)
  const int b = blockIdx.x, tid = threadIdx.x;
  const int G = C / d;
  const float inv_hw = 1.0f / (float)HW;
  for (int i = tid; i < C + 2 * p; i += kThreads) {
    const int c = i - p;
    ys[i] = (c >= 0 && c < C) ? mom[((size_t)b * C + c) * M_REC + M_SX] * inv_hw : 0.f;
  }
  __syncthreads();
  float* Kb = Kring + (size_t)b * T * C;
  for (int c = tid; c < C; c += kThreads) {
    float q = 0.f, k = 0.f;
    for (int j = 0; j < ks; ++j) {
      q = fmaf(wq[j], ys[c + j], q);
      k = fmaf(wk[j], ys[c + j], k);
    }
    qs[c] = q;
    qout[(size_t)b * C + c] = q;
    kts[c] = k;
    Kb[(size_t)(t - 1) * C + c] = k;
  }
  __syncthreads();
  const float s = rsqrtf((float)d);
  // logits: one (head, slot) pair per thread (G*t pairs: the whole workgroup loads the key history) into LDS
  float* lg = kts + C;            // [G][t]
  for (int idx = tid; idx < G * t; idx += kThreads) {
    const int g = idx / t, j = idx - g * t;
    const float* kj = (j == t - 1) ? kts + g * d : Kb + (size_t)j * C + g * d;
    float acc = 0.f;
    for (int i = 0; i < d; ++i) acc = fmaf(qs[g * d + i], kj[i], acc);
    lg[idx] = acc * s;
  }
  __syncthreads();
  // softmax over the depth: LPH lanes per head (all heads of the image at once when G <= 256), the slots strided over
  // them, lane-group reductions.  (One head per wave at a time serialises 16 global round trips per wave; one thread
  // per head serialises 3 t of them: both measured slower.)
  const int lph = lanes_per_head(G), hpp = kThreads / lph;
  const int sub = tid & (lph - 1);
  for (int g0 = 0; g0 < G; g0 += hpp) {
    const int g = g0 + tid / lph;
    const bool live = g < G;
    float* Prow = Pall + (((size_t)b * G + (live ? g : 0)) * T + (t - 1)) * T;
    const float* lr = lg + (live ? g : 0) * t;
    float mx = -INFINITY;
    for (int j = sub; j < t; j += lph) mx = fmaxf(mx, lr[j]);
    for (int off = lph >> 1; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, kWave));
    float den = 0.f;
    for (int j = sub; j < t; j += lph) den += expf(lr[j] - mx);
    for (int off = lph >> 1; off > 0; off >>= 1) den += __shfl_xor(den, off, kWave);
    const float r = 1.0f / den;
    if (live)
      for (int j = sub; j < t; j += lph) Prow[j] = expf(lr[j] - mx) * r;
  }
}

// ------------------------------------------------------------------------------------------------
// attend forward
// ------------------------------------------------------------------------------------------------
template <typename T, int VW>
__device__ __forceinline__ void attend_fwd_linear(const T* __restrict__ vs, float* __restrict__ as,
                                                  const float* __restrict__ coef /*[np][t]*/,
                                                  const T* __restrict__ Vring, T* __restrict__ attn_out,
                                                  const SlabGeo& g, int T_, int t, int b, int c0, int n, int tid) {
  const size_t out_off = ((size_t)b * g.C + c0) * g.HW;
  for (int e0 = tid * VW; e0 < n; e0 += kThreads * VW) {
    const int p0 = e0 / g.HW;
    const int p1 = (e0 + VW - 1) / g.HW;
    float acc[VW], v[VW];
    ld_vec<T, VW>(vs + e0, v);
    if (p0 == p1) {
      const float* cf = coef + p0 * t;
      const float ct = cf[t - 1];
#pragma unroll
      for (int i = 0; i < VW; ++i) acc[i] = ct * v[i];
      for (int j = 0; j < t - 1; ++j) {
        ld_vec<T, VW>(Vring + ring_off(g, T_, b, j, c0) + e0, v);
        const float cj = cf[j];
#pragma unroll
        for (int i = 0; i < VW; ++i) acc[i] = fmaf(cj, v[i], acc[i]);
      }
    } else {                                  // vector straddles planes: per-element coefficients
#pragma unroll
      for (int i = 0; i < VW; ++i) acc[i] = coef[((e0 + i) / g.HW) * t + t - 1] * v[i];
      for (int j = 0; j < t - 1; ++j) {
        ld_vec<T, VW>(Vring + ring_off(g, T_, b, j, c0) + e0, v);
#pragma unroll
        for (int i = 0; i < VW; ++i) acc[i] = fmaf(coef[((e0 + i) / g.HW) * t + j], v[i], acc[i]);
      }
    }
    st_vec<T, VW>(attn_out + out_off + e0, acc);
    // statistics are taken on the rounded values, which is what the tail pass will read back
#pragma unroll
    for (int i = 0; i < VW; ++i) as[e0 + i] = to_f(from_f<T>(acc[i]));
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void base_attend_fwd_nchw(
    const T* __restrict__ x, const float* __restrict__ wv, T* __restrict__ Vring, const float* __restrict__ Pall,
    T* __restrict__ attn_out, float* __restrict__ amom /*[b,c,2]*/, SlabGeo g, int d, int T_, int t) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* xs = reinterpret_cast<T*>(smem);
  T* vs = xs + g.astride;
  float* as = reinterpret_cast<float*>(vs + g.astride);     // [astride] fp32 attn staging
  float* coef = as + g.astride;                             // [CP][t]
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const LaneMap lmap = make_lane_map(g, lane);
  const int c0 = blockIdx.x * g.CP;
  const int np = min(g.CP, g.C - c0);
  const int n = np * g.HW;
  const int ntasks = g.NG * g.NB;
  const int G = g.C / d;
  constexpr int VEC = 16 / sizeof(T);
  const int b_end = min(g.B, (int)(blockIdx.y + 1) * g.BG);
  for (int b = blockIdx.y * g.BG; b < b_end; ++b) {
    const size_t off = ((size_t)b * g.C + c0) * g.HW;
    slab_load(xs, x + off, n, tid);
    for (int i = tid; i < np * t; i += kThreads) {
      const int p = i / t, j = i - p * t;
      coef[i] = Pall[(((size_t)b * G + (c0 + p) / d) * T_ + (t - 1)) * T_ + j];
    }
    __syncthreads();
    for (int task = wave; task < ntasks; task += kWaves) {
      const LaneTask tk = make_task(g, lmap, task, np);
      if (!tk.live) continue;
      const T* xp = xs + tk.p * g.HW;
      T* vp = vs + tk.p * g.HW;
      float w[9];
      load_w9(w, wv, c0 + tk.p);
      Row3 ra = load_row3(xp, tk.r0 - 1, g, tk);
      Row3 rb = load_row3(xp, tk.r0, g, tk);
      for (int r = tk.r0; r < tk.r1; ++r) {
        const Row3 rc = load_row3(xp, r + 1, g, tk);
        const float v = conv9(w, ra, rb, rc);
        if (tk.valid) vp[r * g.W + tk.col] = from_f<T>(v);
        ra = rb; rb = rc;
      }
    }
    __syncthreads();
    T* slot = Vring + ring_off(g, T_, b, t - 1, c0);
    slab_store(slot, vs, n, tid);                             // v_t joins the history
    const bool vec_ok = (n % VEC == 0) && aligned16(Vring + ring_off(g, T_, b, 0, c0)) &&
                        ((((size_t)g.C * g.HW * sizeof(T)) & 15) == 0) && aligned16(attn_out + off);
    if (vec_ok) attend_fwd_linear<T, VEC>(vs, as, coef, Vring, attn_out, g, T_, t, b, c0, n, tid);
    else        attend_fwd_linear<T, 1>(vs, as, coef, Vring, attn_out, g, T_, t, b, c0, n, tid);
    __syncthreads();
    for (int p = wave; p < np; p += kWaves) {                 // per-plane sum / sum of squares
      float s1 = 0.f, s2 = 0.f;
      for (int e = lane; e < g.HW; e += kWave) {
        const float a = as[p * g.HW + e];
        s1 += a;
        s2 = fmaf(a, a, s2);
      }
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      if (lane == 0) {
        amom[((size_t)b * g.C + c0 + p) * 2 + 0] = s1;
        amom[((size_t)b * g.C + c0 + p) * 2 + 1] = s2;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// plain BatchNorm statistics from per-(image, channel) (sum, sum of squares)
// ------------------------------------------------------------------------------------------------
// 4 channels x 64 row lanes: the NHWC moment passes leave up to b*nsplit (thousands of) partial rows per channel
constexpr int kBnCh2 = 4;
constexpr int kBnLanes2 = kThreads / kBnCh2;

__global__ __launch_bounds__(kThreads) void plain_bn_fwd_kernel(
    const float* __restrict__ amom, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ run_mean, float* __restrict__ run_var, int training, float momentum, float eps,
    float* __restrict__ sc, float* __restrict__ sh, float* __restrict__ save_mean, float* __restrict__ save_inv,
    const float* __restrict__ pivot, int B, int C, int HW) {
  __shared__ double r1[kBnLanes2][kBnCh2], r2[kBnLanes2][kBnCh2];
  const int cc = threadIdx.x % kBnCh2, bl = threadIdx.x / kBnCh2;
  const int c = blockIdx.x * kBnCh2 + cc;
  const bool live = c < C;
  double s1 = 0.0, s2 = 0.0;
  if (training && live) {
    // (one 8-byte load per row, four rows in flight: with ~1000 partial rows a lane walks 16 of them, and a loop that
    // waited for every load took 14 us of dependent round trips)
    const float2* src = reinterpret_cast<const float2*>(amom) + c;
#pragma unroll 4
    for (int b = bl; b < B; b += kBnLanes2) {
      const float2 v = src[(size_t)b * C];
      s1 += v.x;
      s2 += v.y;
    }
  }
  r1[bl][cc] = s1; r2[bl][cc] = s2;
  __syncthreads();
  if (bl != 0 || !live) return;
  double mean, var;
  if (training) {
    s1 = 0.0; s2 = 0.0;
    for (int i = 0; i < kBnLanes2; ++i) { s1 += r1[i][cc]; s2 += r2[i][cc]; }
    const double M = (double)B * HW;
    // the sums are of (x - pivot[c]) when the producer shifted them: the variance is shift-invariant, the mean is not
    mean = s1 / M;
    var = s2 / M - mean * mean;
    if (var < 0.0) var = 0.0;
    if (pivot) mean += (double)pivot[c];
    run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mean);
    run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * var * (M / (M > 1.0 ? M - 1.0 : 1.0)));
  } else {
    mean = run_mean[c];
    var = run_var[c];
  }
  const double inv = 1.0 / sqrt(var + (double)eps);
  const double scale = gamma[c] * inv;
  sc[c] = (float)scale;
  sh[c] = (float)(beta[c] - scale * mean);
  save_mean[c] = (float)mean;
  save_inv[c] = (float)inv;
}

// The same from moment RECORDS rec[rows, c, 4] = (sum (x - p), sum (x - p)^2, p, n) with a pivot p and a pixel count n
// per row (the 1x1-convolution GEMM epilogue, MRLA_GEMM_MOMENTS): rows are merged by re-basing them onto the first
// non-empty row's pivot, in double.
__global__ __launch_bounds__(kThreads) void plain_bn_fwd_rec_kernel(
    const float* __restrict__ rec, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ run_mean, float* __restrict__ run_var, int training, float momentum, float eps,
    float* __restrict__ sc, float* __restrict__ sh, float* __restrict__ save_mean, float* __restrict__ save_inv, int R,
    int C) {
  __shared__ double r1[kBnLanes2][kBnCh2], r2[kBnLanes2][kBnCh2], rn[kBnLanes2][kBnCh2];
  __shared__ double piv[kBnCh2];
  const int cc = threadIdx.x % kBnCh2, bl = threadIdx.x / kBnCh2;
  const int c = blockIdx.x * kBnCh2 + cc;
  const bool live = c < C;
  if (bl == 0) {                      // the common pivot: the first non-empty row's
    double P = 0.0;
    if (training && live)
      for (int b = 0; b < R; ++b)
        if (rec[((size_t)b * C + c) * 4 + 3] > 0.f) { P = rec[((size_t)b * C + c) * 4 + 2]; break; }
    piv[cc] = P;
  }
  __syncthreads();
  const double P = piv[cc];
  double s1 = 0.0, s2 = 0.0, n = 0.0;
  if (training && live)
#pragma unroll 4
    for (int b = bl; b < R; b += kBnLanes2) {
      const float4 q = *reinterpret_cast<const float4*>(rec + ((size_t)b * C + c) * 4);
      const double a = q.x, b2 = q.y, d = (double)q.z - P, nb = q.w;
      if (nb > 0.0) {
        s1 += a + nb * d;
        s2 += b2 + 2.0 * d * a + nb * d * d;
        n += nb;
      }
    }
  r1[bl][cc] = s1; r2[bl][cc] = s2; rn[bl][cc] = n;
  __syncthreads();
  if (bl != 0 || !live) return;
  double mean, var;
  if (training) {
    s1 = 0.0; s2 = 0.0; n = 0.0;
    for (int i = 0; i < kBnLanes2; ++i) { s1 += r1[i][cc]; s2 += r2[i][cc]; n += rn[i][cc]; }
    const double M = n > 0.0 ? n : 1.0;
    const double dm = s1 / M;
    mean = dm + P;
    var = s2 / M - dm * dm;
    if (var < 0.0) var = 0.0;
    run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mean);
    run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * var * (M / (M > 1.0 ? M - 1.0 : 1.0)));
  } else {
    mean = run_mean[c];
    var = run_var[c];
  }
  const double inv = 1.0 / sqrt(var + (double)eps);
  const double scale = gamma[c] * inv;
  sc[c] = (float)scale;
  sh[c] = (float)(beta[c] - scale * mean);
  save_mean[c] = (float)mean;
  save_inv[c] = (float)inv;
}

// tmom[b,c,2] = (sum dz, sum dz*attn)  ->  cb[c,3] = (e, f, h) with dattn = e*dz + f*attn + h; dgamma, dbeta
__global__ __launch_bounds__(kThreads) void plain_bn_bwd_kernel(
    const float* __restrict__ tmom, const float* __restrict__ gamma, const float* __restrict__ save_mean,
    const float* __restrict__ save_inv, int training, int centered, float* __restrict__ cb, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int B, int C, int HW) {
  __shared__ double r1[kBnLanes2][kBnCh2], r2[kBnLanes2][kBnCh2];
  const int cc = threadIdx.x % kBnCh2, bl = threadIdx.x / kBnCh2;
  const int c = blockIdx.x * kBnCh2 + cc;
  const bool live = c < C;
  double s1 = 0.0, s2 = 0.0;
  if (live) {
    const float2* src = reinterpret_cast<const float2*>(tmom) + c;
#pragma unroll 4
    for (int b = bl; b < B; b += kBnLanes2) {
      const float2 v = src[(size_t)b * C];
      s1 += v.x;
      s2 += v.y;
    }
  }
  r1[bl][cc] = s1; r2[bl][cc] = s2;
  __syncthreads();
  if (bl != 0 || !live) return;
  s1 = 0.0; s2 = 0.0;
  for (int i = 0; i < kBnLanes2; ++i) { s1 += r1[i][cc]; s2 += r2[i][cc]; }
  const double mean = save_mean[c], inv = save_inv[c];
  // centered: the producer summed dz * (x - mean) already (no cancelling subtraction when |mean| >> sigma)
  const double dbe = s1, dga = centered ? inv * s2 : inv * (s2 - mean * s1);
  const double e = gamma[c] * inv;
  double f = 0.0, h = 0.0;
  if (training) {
    const double M = (double)B * HW;
    const double c1 = dbe / M, c2 = dga / M;
    f = -e * inv * c2;
    h = e * (-c1 + inv * mean * c2);
  }
  cb[c * 3 + 0] = (float)e; cb[c * 3 + 1] = (float)f; cb[c * 3 + 2] = (float)h;
  dgamma[c] = (float)dga;
  dbeta[c] = (float)dbe;
}

// ------------------------------------------------------------------------------------------------
// tail forward: out = x + dp[b]*relu(sc[c]*attn + sh[c])       (pure streaming, 16 B per lane)
// ------------------------------------------------------------------------------------------------
template <typename T, int VW>
__device__ __forceinline__ void tail_fwd_body(const T* __restrict__ x, const T* __restrict__ attn,
                                              const float* __restrict__ sc, const float* __restrict__ sh,
                                              const float* __restrict__ dp, T* __restrict__ out, size_t total, int C,
                                              int HW) {
  const size_t stride = (size_t)gridDim.x * kThreads * VW;
  for (size_t e0 = ((size_t)blockIdx.x * kThreads + threadIdx.x) * VW; e0 < total; e0 += stride) {
    float xv[VW], av[VW], y[VW];
    ld_vec<T, VW>(x + e0, xv);
    ld_vec<T, VW>(attn + e0, av);
    const size_t pl0 = e0 / HW, pl1 = (e0 + VW - 1) / HW;
    if (pl0 == pl1) {
      const int c = (int)(pl0 % C);
      const float dpb = dp ? dp[pl0 / C] : 1.f;
      const float s = sc[c], h = sh[c];
#pragma unroll
      for (int i = 0; i < VW; ++i) y[i] = fmaf(dpb, fmaxf(fmaf(s, av[i], h), 0.f), xv[i]);
    } else {
#pragma unroll
      for (int i = 0; i < VW; ++i) {
        const size_t pl = (e0 + i) / HW;
        const int c = (int)(pl % C);
        const float dpb = dp ? dp[pl / C] : 1.f;
        y[i] = fmaf(dpb, fmaxf(fmaf(sc[c], av[i], sh[c]), 0.f), xv[i]);
      }
    }
    st_vec<T, VW>(out + e0, y);
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void base_tail_fwd_nchw(const T* __restrict__ x, const T* __restrict__ attn,
                                                               const float* __restrict__ sc, const float* __restrict__ sh,
                                                               const float* __restrict__ dp, T* __restrict__ out,
                                                               size_t total, int C, int HW) {
  constexpr int VEC = 16 / sizeof(T);
  if (total % VEC == 0 && aligned16(x) && aligned16(attn) && aligned16(out))
    tail_fwd_body<T, VEC>(x, attn, sc, sh, dp, out, total, C, HW);
  else
    tail_fwd_body<T, 1>(x, attn, sc, sh, dp, out, total, C, HW);
}

// ------------------------------------------------------------------------------------------------
// tail backward statistics: tmom[b,c,2] = (sum dz, sum dz*attn), dz = dp*dOut*[sc*attn+sh > 0]
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void base_tail_stats_bwd_nchw(
    const T* __restrict__ dout, const T* __restrict__ attn, const float* __restrict__ sc, const float* __restrict__ sh,
    const float* __restrict__ dp, float* __restrict__ tmom, SlabGeo g) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* gs = reinterpret_cast<T*>(smem);
  T* as = gs + g.astride;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int c0 = blockIdx.x * g.CP;
  const int np = min(g.CP, g.C - c0);
  const int n = np * g.HW;
  const int b_end = min(g.B, (int)(blockIdx.y + 1) * g.BG);
  for (int b = blockIdx.y * g.BG; b < b_end; ++b) {
    const size_t off = ((size_t)b * g.C + c0) * g.HW;
    slab_load(gs, dout + off, n, tid);
    slab_load(as, attn + off, n, tid);
    __syncthreads();
    const float dpb = dp ? dp[b] : 1.f;
    for (int p = wave; p < np; p += kWaves) {
      const float s = sc[c0 + p], h = sh[c0 + p];
      float s1 = 0.f, s2 = 0.f;
      for (int e = lane; e < g.HW; e += kWave) {
        const float a = to_f(as[p * g.HW + e]);
        const float dz = (fmaf(s, a, h) > 0.f) ? dpb * to_f(gs[p * g.HW + e]) : 0.f;
        s1 += dz;
        s2 = fmaf(dz, a, s2);
      }
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      if (lane == 0) {
        tmom[((size_t)b * g.C + c0 + p) * 2 + 0] = s1;
        tmom[((size_t)b * g.C + c0 + p) * 2 + 1] = s2;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// attend backward: dA_t = e*dz + f*attn + h -> dA ring; pmom[b,c,j] = sum_hw dA_t * V_j  (j < t)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void base_attend_bwd_nchw(
    const T* __restrict__ dout, const T* __restrict__ attn, const float* __restrict__ sc, const float* __restrict__ sh,
    const float* __restrict__ dp, const float* __restrict__ cb /*[c,3]*/, const T* __restrict__ Vring,
    T* __restrict__ dAring, float* __restrict__ pmom /*[b,c,t]*/, SlabGeo g, int T_, int t) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* gs = reinterpret_cast<T*>(smem);          // dOut, then V_j staging
  T* as = gs + g.astride;                      // attn, then dA (rounded) for the ring store
  float* da = reinterpret_cast<float*>(as + g.astride);   // [astride] dA in fp32 (rounded values)
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int c0 = blockIdx.x * g.CP;
  const int np = min(g.CP, g.C - c0);
  const int n = np * g.HW;
  const int b_end = min(g.B, (int)(blockIdx.y + 1) * g.BG);
  for (int b = blockIdx.y * g.BG; b < b_end; ++b) {
    const size_t off = ((size_t)b * g.C + c0) * g.HW;
    slab_load(gs, dout + off, n, tid);
    if (attn) slab_load(as, attn + off, n, tid);
    __syncthreads();
    const float dpb = dp ? dp[b] : 1.f;
    for (int p = wave; p < np; p += kWaves) {
      const int c = c0 + p;
      const float s = sc ? sc[c] : 0.f, h = sc ? sh[c] : 1.f;       // sc == null: no tail, dA = dOut
      const float e_ = sc ? cb[c * 3 + 0] : 1.f, f_ = sc ? cb[c * 3 + 1] : 0.f, h_ = sc ? cb[c * 3 + 2] : 0.f;
      for (int e = lane; e < g.HW; e += kWave) {
        const float a = attn ? to_f(as[p * g.HW + e]) : 0.f;
        const float dz = (fmaf(s, a, h) > 0.f) ? dpb * to_f(gs[p * g.HW + e]) : 0.f;
        const T r = from_f<T>(fmaf(e_, dz, fmaf(f_, a, h_)));
        as[p * g.HW + e] = r;
        da[p * g.HW + e] = to_f(r);
      }
    }
    __syncthreads();
    slab_store(dAring + ring_off(g, T_, b, t - 1, c0), as, n, tid);
    for (int j = 0; j < t; ++j) {
      slab_load(gs, Vring + ring_off(g, T_, b, j, c0), n, tid);
      __syncthreads();
      for (int p = wave; p < np; p += kWaves) {
        float s1 = 0.f;
        for (int e = lane; e < g.HW; e += kWave) s1 = fmaf(da[p * g.HW + e], to_f(gs[p * g.HW + e]), s1);
        s1 = wave_sum(s1);
        if (lane == 0) pmom[((size_t)b * g.C + c0 + p) * t + j] = s1;
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// gate backward: one workgroup per image
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum2(float v, float* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kWaves; ++i) s += scratch[i];
  return s;
}

__global__ __launch_bounds__(kThreads) void base_gate_bwd_kernel(
    const float* __restrict__ mom, const float* __restrict__ pmom /*[b,c,t]*/, const float* __restrict__ Pall,
    const float* __restrict__ q, const float* __restrict__ Kring, float* __restrict__ dKring,
    const float* __restrict__ wq, const float* __restrict__ wk, int ks, float* __restrict__ dyx,
    float* __restrict__ dwqk_part, int C, int HW, int d, int T, int t, int first_touch, int stage_pmom,
    float* __restrict__ tok_part, int tok_bands) {
  extern __shared__ float sm[];
  const int p = (ks - 1) / 2;
  const int CPD = C + 2 * p;
  const int G = C / d;
  float* ys = sm;              // [CPD]
  float* dqs = ys + CPD;       // [CPD]
  float* dks = dqs + CPD;      // [CPD]
  float* dlg = dks + CPD;      // [G][t] dlogit
  float* scratch = dlg + G * t;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float inv_hw = 1.0f / (float)HW;
  const float s = rsqrtf((float)d);
  for (int i = tid; i < CPD; i += kThreads) {
    const int c = i - p;
    ys[i] = (c >= 0 && c < C) ? mom[((size_t)b * C + c) * M_REC + M_SX] * inv_hw : 0.f;
    dqs[i] = 0.f;
    dks[i] = 0.f;
  }
  // dP[g,j] = sum_{c in g} pmom[b,c,j].  The image's [C][t] block of pmom is contiguous: every thread fetches float4s of it
  // with all its loads in flight at once and parks them in LDS (94 KB at C = 1024, t = 23; the grid is one workgroup per
  // image, so LDS occupancy is not a concern), then one (head, slot) pair per thread adds its d entries from LDS -- a chain
  // of d dependent-latency global loads per thread was most of this kernel's time.
  if (stage_pmom) {
    float* pm = sm + ((3 * CPD + G * t + kWaves + 3) & ~3);        // 16-byte aligned
    const float4* src = reinterpret_cast<const float4*>(pmom + (size_t)b * C * t);
    const int n4 = (C * t) >> 2;                       // C % 4 == 0
    for (int i = tid; i < n4; i += kThreads) reinterpret_cast<float4*>(pm)[i] = src[i];
    __syncthreads();
    for (int idx = tid; idx < G * t; idx += kThreads) {
      const int g = idx / t, j = idx - g * t;
      float dP = 0.f;
      for (int i = 0; i < d; ++i) dP += pm[(g * d + i) * t + j];
      dlg[idx] = dP;
    }
  } else {
    for (int idx = tid; idx < G * t; idx += kThreads) {
      const int g = idx / t, j = idx - g * t;
      float dP = 0.f;
      for (int i = 0; i < d; ++i) dP += pmom[((size_t)b * C + g * d + i) * t + j];
      dlg[idx] = dP;
    }
  }
  __syncthreads();
  {   // softmax backward: LPH lanes per head, all heads at once (see base_gate_fwd_kernel)
    const int lph = lanes_per_head(G), hpp = kThreads / lph;
    const int sub = tid & (lph - 1);
    for (int g0 = 0; g0 < G; g0 += hpp) {
      const int g = g0 + tid / lph;
      const bool live = g < G;
      const float* Prow = Pall + (((size_t)b * G + (live ? g : 0)) * T + (t - 1)) * T;
      float* dr = dlg + (live ? g : 0) * t;
      float dot = 0.f;
      for (int j = sub; j < t; j += lph) dot = fmaf(Prow[j], dr[j], dot);
      for (int off = lph >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, kWave);
      if (live)
        for (int j = sub; j < t; j += lph) dr[j] = Prow[j] * (dr[j] - dot) * s;      // (the row is in L1 by now)
    }
  }
  __syncthreads();
  const float* Kb = Kring + (size_t)b * T * C;
  float* dKb = dKring + (size_t)b * T * C;
  // four channels per thread (one head when d % 4 == 0), the slots in batches of kJB so that 2*kJB 16-byte loads are in flight
  constexpr int kJB = 8;
  const bool one_head = (d & 3) == 0;
  for (int c = tid * 4; c < C; c += kThreads * 4) {
    const float* dl = dlg + (c / d) * t;
    const float4 qc = *reinterpret_cast<const float4*>(q + (size_t)b * C + c);
    float4 dq = {0.f, 0.f, 0.f, 0.f}, last = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < t; j0 += kJB) {
      float4 kv[kJB], pv[kJB];
#pragma unroll
      for (int u = 0; u < kJB; ++u) {
        const int j = min(j0 + u, t - 1);
        kv[u] = *reinterpret_cast<const float4*>(Kb + (size_t)j * C + c);
        // the first backward call of a stage (its last layer) touches every slot first: start from zero
        pv[u] = first_touch ? float4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const float4*>(dKb + (size_t)j * C + c);
      }
#pragma unroll
      for (int u = 0; u < kJB; ++u) {
        if (j0 + u < t) {
          const int j = j0 + u;
          const float l = dl[j];
          const float ly = one_head ? l : dlg[((c + 1) / d) * t + j], lz = one_head ? l : dlg[((c + 2) / d) * t + j],
                      lw = one_head ? l : dlg[((c + 3) / d) * t + j];
          dq.x = fmaf(l, kv[u].x, dq.x); dq.y = fmaf(ly, kv[u].y, dq.y);
          dq.z = fmaf(lz, kv[u].z, dq.z); dq.w = fmaf(lw, kv[u].w, dq.w);
          last.x = fmaf(l, qc.x, pv[u].x); last.y = fmaf(ly, qc.y, pv[u].y);
          last.z = fmaf(lz, qc.z, pv[u].z); last.w = fmaf(lw, qc.w, pv[u].w);
          *reinterpret_cast<float4*>(dKb + (size_t)(j0 + u) * C + c) = last;
        }
      }
    }
    // dL/dk_t (slot t-1) is complete now
    dks[p + c] = last.x; dks[p + c + 1] = last.y; dks[p + c + 2] = last.z; dks[p + c + 3] = last.w;
    dqs[p + c] = dq.x; dqs[p + c + 1] = dq.y; dqs[p + c + 2] = dq.z; dqs[p + c + 3] = dq.w;
  }
  __syncthreads();
  for (int c = tid; c < C; c += kThreads) {
    float dy = 0.f;
    for (int j = 0; j < ks; ++j) {
      dy = fmaf(wq[j], dqs[c - j + 2 * p], dy);
      dy = fmaf(wk[j], dks[c - j + 2 * p], dy);
    }
    dyx[(size_t)b * C + c] = dy * inv_hw;
    if (tok_part) {    // token path (see gate_bwd_kernel in gate.hip): complete the LayerNorm parameter partials with dy
      float hsum = 0.f;
      for (int z = 0; z < tok_bands; ++z) hsum += tok_part[(((size_t)z * gridDim.x + b) * C + c) * kTokParts + kTokPartHat];
      float* pr = tok_part + ((size_t)b * C + c) * kTokParts;
      pr[kTokPartLnxW] = fmaf(dy * inv_hw, hsum, pr[kTokPartLnxW]);
      pr[kTokPartLnxB] += dy;
    }
  }
  for (int j = 0; j < ks; ++j) {
    float aq = 0.f, ak = 0.f;
    for (int c = tid; c < C; c += kThreads) {
      aq = fmaf(dqs[p + c], ys[c + j], aq);
      ak = fmaf(dks[p + c], ys[c + j], ak);
    }
    aq = block_sum2(aq, scratch);
    ak = block_sum2(ak, scratch);
    if (tid == 0) {
      dwqk_part[(size_t)b * 2 * ks + j] = aq;
      dwqk_part[(size_t)b * 2 * ks + ks + j] = ak;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// value backward: dV_t = sum_{t'=t..Tc} P_t'[b,g,t-1] * dA_t'; dx = dOut + dwconv^T(dV_t) + dyx; dWv partials
// ------------------------------------------------------------------------------------------------
template <typename T, int VW>
__device__ __forceinline__ void value_bwd_linear(float* __restrict__ dvs, const float* __restrict__ coef /*[np][nl]*/,
                                                 const T* __restrict__ dAring, const SlabGeo& g, int T_, int t, int nl,
                                                 int b, int c0, int n, int tid) {
  for (int e0 = tid * VW; e0 < n; e0 += kThreads * VW) {
    const int p0 = e0 / g.HW, p1 = (e0 + VW - 1) / g.HW;
    float acc[VW], v[VW];
#pragma unroll
    for (int i = 0; i < VW; ++i) acc[i] = 0.f;
    for (int l = 0; l < nl; ++l) {
      ld_vec<T, VW>(dAring + ring_off(g, T_, b, t - 1 + l, c0) + e0, v);
      if (p0 == p1) {
        const float cl = coef[p0 * nl + l];
#pragma unroll
        for (int i = 0; i < VW; ++i) acc[i] = fmaf(cl, v[i], acc[i]);
      } else {
#pragma unroll
        for (int i = 0; i < VW; ++i) acc[i] = fmaf(coef[((e0 + i) / g.HW) * nl + l], v[i], acc[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < VW; ++i) dvs[e0 + i] = acc[i];
  }
}

template <typename T, int TPW>
__global__ __launch_bounds__(kThreads) void base_value_bwd_nchw(
    const T* __restrict__ dout, const T* __restrict__ x, const float* __restrict__ wv, const T* __restrict__ dAring,
    const float* __restrict__ Pall, const float* __restrict__ dyx, T* __restrict__ dx, float* __restrict__ dwv_part,
    SlabGeo g, int d, int T_, int t, int Tc, int res) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* xs = reinterpret_cast<T*>(smem);
  T* gs = xs + g.astride;                                   // dOut in, dx out (same lane, same element)
  float* dvs = reinterpret_cast<float*>(gs + g.astride);    // [astride] dV_t in fp32
  float* coef = dvs + g.astride;                            // [CP][nl]
  float* red = coef + g.CP * (Tc - t + 1);                  // [ntasks][PW][9]
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const LaneMap lmap = make_lane_map(g, lane);
  const int c0 = blockIdx.x * g.CP;
  const int np = min(g.CP, g.C - c0);
  const int n = np * g.HW;
  const int ntasks = g.NG * g.NB;
  const int G = g.C / d;
  const int nl = Tc - t + 1;
  constexpr int VEC = 16 / sizeof(T);
  const int b_end = min(g.B, (int)(blockIdx.y + 1) * g.BG);
  float wg[TPW][9];
#pragma unroll
  for (int s = 0; s < TPW; ++s)
#pragma unroll
    for (int k = 0; k < 9; ++k) wg[s][k] = 0.f;
  for (int b = blockIdx.y * g.BG; b < b_end; ++b) {
    const size_t off = ((size_t)b * g.C + c0) * g.HW;
    slab_load(xs, x + off, n, tid);
    slab_load(gs, dout + off, n, tid);
    for (int i = tid; i < np * nl; i += kThreads) {
      const int p = i / nl, l = i - p * nl;                 // layer t' = t + l reads its P row t'-1, column t-1
      coef[i] = Pall[(((size_t)b * G + (c0 + p) / d) * T_ + (t - 1 + l)) * T_ + (t - 1)];
    }
    __syncthreads();
    const bool vec_ok = (n % VEC == 0) && aligned16(dAring + ring_off(g, T_, b, 0, c0)) &&
                        ((((size_t)g.C * g.HW * sizeof(T)) & 15) == 0);
    if (vec_ok) value_bwd_linear<T, VEC>(dvs, coef, dAring, g, T_, t, nl, b, c0, n, tid);
    else        value_bwd_linear<T, 1>(dvs, coef, dAring, g, T_, t, nl, b, c0, n, tid);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
      const int task = wave + s * kWaves;
      const LaneTask tk = make_task(g, lmap, min(task, ntasks - 1), np);
      if (task < ntasks && tk.live) {
        const int c = c0 + tk.p;
        const T* xp = xs + tk.p * g.HW;
        T* gp = gs + tk.p * g.HW;
        const float* dvp = dvs + tk.p * g.HW;
        float w[9];
        load_w9(w, wv, c);
        const float dy = dyx[(size_t)b * g.C + c];
        // dx[r][w] = sum_{i,j} wv[i][j] * dV[r-i+1][w-j+1];  dWv[i][j] += dV[r][w] * x[r+i-1][w+j-1]
        Row3 ua = load_row3(dvp, tk.r0 - 1, g, tk);          // dV[r-1]
        Row3 ub = load_row3(dvp, tk.r0, g, tk);              // dV[r]
        Row3 xa = load_row3(xp, tk.r0 - 1, g, tk);           // x[r-1]
        Row3 xb = load_row3(xp, tk.r0, g, tk);               // x[r]
        for (int r = tk.r0; r < tk.r1; ++r) {
          const Row3 uc = load_row3(dvp, r + 1, g, tk);
          const Row3 xc = load_row3(xp, r + 1, g, tk);
          float s9 = w[0] * uc.r;
          s9 = fmaf(w[1], uc.c, s9); s9 = fmaf(w[2], uc.l, s9);
          s9 = fmaf(w[3], ub.r, s9); s9 = fmaf(w[4], ub.c, s9); s9 = fmaf(w[5], ub.l, s9);
          s9 = fmaf(w[6], ua.r, s9); s9 = fmaf(w[7], ua.c, s9); s9 = fmaf(w[8], ua.l, s9);
          const float du = tk.valid ? ub.c : 0.f;
          wg[s][0] = fmaf(du, xa.l, wg[s][0]); wg[s][1] = fmaf(du, xa.c, wg[s][1]); wg[s][2] = fmaf(du, xa.r, wg[s][2]);
          wg[s][3] = fmaf(du, xb.l, wg[s][3]); wg[s][4] = fmaf(du, xb.c, wg[s][4]); wg[s][5] = fmaf(du, xb.r, wg[s][5]);
          wg[s][6] = fmaf(du, xc.l, wg[s][6]); wg[s][7] = fmaf(du, xc.c, wg[s][7]); wg[s][8] = fmaf(du, xc.r, wg[s][8]);
          if (tk.valid) {
            float y = ((res & 1) ? to_f(gp[r * g.W + tk.col]) : 0.f) + s9 + dy;
            if ((res & 2) && !(xb.c > 0.f)) y = 0.f;          // x_t = relu(pre + identity): gradient wrt the pre-activation
            gp[r * g.W + tk.col] = from_f<T>(y);
          }
          ua = ub; ub = uc; xa = xb; xb = xc;
        }
      }
    }
    __syncthreads();
    slab_store(dx + off, gs, n, tid);
    __syncthreads();
  }
#pragma unroll
  for (int s = 0; s < TPW; ++s) {
    const int task = wave + s * kWaves;
    const LaneTask tk = make_task(g, lmap, min(task, ntasks - 1), np);
    if (task < ntasks && tk.live) {
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float v = seg_sum(tk.valid ? wg[s][k] : 0.f, lane, g.WS);
        if (tk.last) red[(task * g.PW + tk.pl) * 9 + k] = v;
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < np * 9; idx += kThreads) {
    const int p = idx / 9, k = idx - p * 9;
    const int grp = p / g.PW, pl = p - grp * g.PW;
    float s = 0.f;
    for (int band = 0; band < g.NB; ++band) s += red[((grp * g.NB + band) * g.PW + pl) * 9 + k];
    dwv_part[((size_t)blockIdx.y * g.C + c0 + p) * 9 + k] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
template <typename K>
static hipError_t set_lds2(K kernel, size_t bytes) {
  return lds_opt_in(reinterpret_cast<const void*>(kernel), bytes);
}

#define MRLA_DISPATCH_T(DT, CALL)        \
  switch (DT) {                          \
    case MRLA_F32:  CALL(float); break;  \
    case MRLA_BF16: CALL(bf16_t); break; \
    case MRLA_F16:  CALL(f16_t); break;  \
    default: return MRLA_EINVAL;         \
  }

int launch_base_gate_fwd(const float* mom, const float* wq, const float* wk, int ks, float* Kring, float* Pall,
                         float* q, int B, int C, int HW, int d, int T, int t, hipStream_t st) {
  const size_t lds = (size_t)(3 * C + 2 * ((ks - 1) / 2) + (C / d) * t) * sizeof(float);
  if (lds > 64 * 1024) return MRLA_EUNSUPPORTED;
  hipLaunchKernelGGL(base_gate_fwd_kernel, dim3(B), dim3(kThreads), lds, st, mom, wq, wk, ks, Kring, Pall, q, C, HW, d,
                     T, t);
  return hip_status(hipGetLastError());
}

int launch_base_attend_fwd(const void* x, const float* wv, void* Vring, const float* Pall, void* attn, float* amom,
                           const SlabGeo& g, int d, int T, int t, int dtype, hipStream_t st) {
  const size_t es = dtype_size(dtype);
  const size_t lds = (size_t)g.astride * (2 * es + sizeof(float)) + (size_t)g.CP * t * sizeof(float);
  if (lds > 150 * 1024) return MRLA_EUNSUPPORTED;
  const dim3 grid(g.slabs, (g.B + g.BG - 1) / g.BG);
#define CALL(TT)                                                                                          \
  {                                                                                                       \
    if (set_lds2(base_attend_fwd_nchw<TT>, lds) != hipSuccess) return MRLA_EHIP;                            \
    hipLaunchKernelGGL((base_attend_fwd_nchw<TT>), grid, dim3(kThreads), lds, st, (const TT*)x, wv,        \
                       (TT*)Vring, Pall, (TT*)attn, amom, g, d, T, t);                                    \
  }
  MRLA_DISPATCH_T(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_plain_bn_fwd(const float* amom, const float* gamma, const float* beta, float* run_mean, float* run_var,
                        int training, float momentum, float eps, float* sc, float* sh, float* save_mean,
                        float* save_inv, const float* pivot, int B, int C, int HW, hipStream_t st) {
  hipLaunchKernelGGL(plain_bn_fwd_kernel, dim3((C + kBnCh2 - 1) / kBnCh2), dim3(kThreads), 0, st, amom, gamma, beta,
                     run_mean, run_var, training, momentum, eps, sc, sh, save_mean, save_inv, pivot, B, C, HW);
  return hip_status(hipGetLastError());
}

int launch_plain_bn_fwd_rec(const float* rec, const float* gamma, const float* beta, float* run_mean, float* run_var,
                            int training, float momentum, float eps, float* sc, float* sh, float* save_mean,
                            float* save_inv, int R, int C, hipStream_t st) {
  hipLaunchKernelGGL(plain_bn_fwd_rec_kernel, dim3((C + kBnCh2 - 1) / kBnCh2), dim3(kThreads), 0, st, rec, gamma, beta,
                     run_mean, run_var, training, momentum, eps, sc, sh, save_mean, save_inv, R, C);
  return hip_status(hipGetLastError());
}

int launch_plain_bn_bwd(const float* tmom, const float* gamma, const float* save_mean, const float* save_inv,
                        int training, int centered, float* cb, float* dgamma, float* dbeta, int B, int C, int HW,
                        hipStream_t st) {
  hipLaunchKernelGGL(plain_bn_bwd_kernel, dim3((C + kBnCh2 - 1) / kBnCh2), dim3(kThreads), 0, st, tmom, gamma,
                     save_mean, save_inv, training, centered, cb, dgamma, dbeta, B, C, HW);
  return hip_status(hipGetLastError());
}

int launch_base_tail_fwd(const void* x, const void* attn, const float* sc, const float* sh, const float* dp, void* out,
                         int B, int C, int HW, int dtype, hipStream_t st) {
  const size_t total = (size_t)B * C * HW;
  const size_t vec = 16 / dtype_size(dtype);
  const size_t want = (total / vec + kThreads - 1) / kThreads;
  const int grid = (int)std::max<size_t>(1, std::min<size_t>(want, 256 * 16));
#define CALL(TT)                                                                                               \
  hipLaunchKernelGGL((base_tail_fwd_nchw<TT>), dim3(grid), dim3(kThreads), 0, st, (const TT*)x, (const TT*)attn, \
                     sc, sh, dp, (TT*)out, total, C, HW);
  MRLA_DISPATCH_T(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_base_tail_stats_bwd(const void* dout, const void* attn, const float* sc, const float* sh, const float* dp,
                               float* tmom, const SlabGeo& g, int dtype, hipStream_t st) {
  const size_t lds = (size_t)g.astride * dtype_size(dtype) * 2;
  const dim3 grid(g.slabs, (g.B + g.BG - 1) / g.BG);
#define CALL(TT)                                                                                            \
  {                                                                                                         \
    if (set_lds2(base_tail_stats_bwd_nchw<TT>, lds) != hipSuccess) return MRLA_EHIP;                          \
    hipLaunchKernelGGL((base_tail_stats_bwd_nchw<TT>), grid, dim3(kThreads), lds, st, (const TT*)dout,       \
                       (const TT*)attn, sc, sh, dp, tmom, g);                                               \
  }
  MRLA_DISPATCH_T(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_base_attend_bwd(const void* dout, const void* attn, const float* sc, const float* sh, const float* dp,
                           const float* cb, const void* Vring, void* dAring, float* pmom, const SlabGeo& g, int T,
                           int t, int dtype, hipStream_t st) {
  const size_t es = dtype_size(dtype);
  const size_t lds = (size_t)g.astride * (2 * es + sizeof(float));
  if (lds > 150 * 1024) return MRLA_EUNSUPPORTED;
  const dim3 grid(g.slabs, (g.B + g.BG - 1) / g.BG);
#define CALL(TT)                                                                                            \
  {                                                                                                         \
    if (set_lds2(base_attend_bwd_nchw<TT>, lds) != hipSuccess) return MRLA_EHIP;                              \
    hipLaunchKernelGGL((base_attend_bwd_nchw<TT>), grid, dim3(kThreads), lds, st, (const TT*)dout,           \
                       (const TT*)attn, sc, sh, dp, cb, (const TT*)Vring, (TT*)dAring, pmom, g, T, t);      \
  }
  MRLA_DISPATCH_T(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_base_gate_bwd(const float* mom, const float* pmom, const float* Pall, const float* q, const float* Kring,
                         float* dKring, const float* wq, const float* wk, int ks, float* dyx, float* dwqk_part, int B,
                         int C, int HW, int d, int T, int t, int first_touch, hipStream_t st, float* tok_part,
                         int tok_bands) {
  const int p = (ks - 1) / 2;
  size_t lds = (size_t)(3 * (C + 2 * p) + (C / d) * t + kWaves) * sizeof(float);
  if (lds > 64 * 1024 || (C & 3)) return MRLA_EUNSUPPORTED;
  const size_t staged = lds + (size_t)(C * t + 4) * sizeof(float);        // pmom[b] parked in LDS when it fits
  const int stage_pmom = staged <= 150 * 1024;
  if (stage_pmom) lds = staged;
  if (set_lds2(base_gate_bwd_kernel, lds) != hipSuccess) return MRLA_EHIP;
  hipLaunchKernelGGL(base_gate_bwd_kernel, dim3(B), dim3(kThreads), lds, st, mom, pmom, Pall, q, Kring, dKring, wq, wk,
                     ks, dyx, dwqk_part, C, HW, d, T, t, first_touch, stage_pmom, tok_part, tok_bands);
  return hip_status(hipGetLastError());
}

int launch_base_value_bwd(const void* dout, const void* x, const float* wv, const void* dAring, const float* Pall,
                          const float* dyx, void* dx, float* dwv_part, const SlabGeo& g, int d, int T, int t, int Tc,
                          int res, int dtype, hipStream_t st) {
  const size_t es = dtype_size(dtype);
  const size_t lds = (size_t)g.astride * (2 * es + sizeof(float)) + (size_t)g.CP * (Tc - t + 1) * sizeof(float) +
                     (size_t)g.NG * g.NB * g.PW * 9 * sizeof(float);
  if (lds > 150 * 1024) return MRLA_EUNSUPPORTED;
  const dim3 grid(g.slabs, (g.B + g.BG - 1) / g.BG);
  const int tpw = (g.NG * g.NB + kWaves - 1) / kWaves;
#define CALL_TPW(TT, TPW)                                                                                   \
  {                                                                                                         \
    if (set_lds2(base_value_bwd_nchw<TT, TPW>, lds) != hipSuccess) return MRLA_EHIP;                          \
    hipLaunchKernelGGL((base_value_bwd_nchw<TT, TPW>), grid, dim3(kThreads), lds, st, (const TT*)dout,       \
                       (const TT*)x, wv, (const TT*)dAring, Pall, dyx, (TT*)dx, dwv_part, g, d, T, t, Tc, res);   \
  }
#define CALL(TT)                                                  \
  {                                                               \
    if (tpw <= 1) CALL_TPW(TT, 1)                                 \
    else if (tpw == 2) CALL_TPW(TT, 2)                            \
    else if (tpw <= kMaxTasksPerWave) CALL_TPW(TT, kMaxTasksPerWave) \
    else return MRLA_EUNSUPPORTED;                                \
  }
  MRLA_DISPATCH_T(dtype, CALL)
#undef CALL
#undef CALL_TPW
  return hip_status(hipGetLastError());
}

}  // namespace mrla
