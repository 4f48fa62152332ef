// The small kernels of the MRLA-light path: everything that lives on [b, c] / [b, g] / [c] tensors.
// They turn the per-plane moments produced by the streaming passes into the gate, the BatchNorm
// statistics and all parameter gradients in closed form (derivation: DESIGN.md section 3).
//
// Reference statements covered: mrla_light_module.py:59-60,67,70 (Wq/Wk conv1d, per-head dot product,
// sigmoid); nn.BatchNorm2d statistics / running update / backward of `bn_mrla`
// (resnet_mrla_light.py:85,116); the reductions autograd performs for lambda_t, Wq, Wk.
#include <algorithm>

#include "mrla_device.h"
#include "mrla_kernels.h"

namespace mrla {

// bmom[B][C][D_N]: record 0 of the backward statistics pass's buffer -- where the pass leaves partial records (strip / row
// ranges, mrla_light_bmom_splits) it folds them into record 0 itself (launch_fold_rows).
struct BwdMoments { float d, dv, d_o; };
__device__ __forceinline__ BwdMoments load_bmom(const float* __restrict__ bmom, int b, int c, int C) {
  const float* p = bmom + ((size_t)b * C + c) * D_N;
  return {p[D_D], p[D_DV], p[D_DO]};
}

constexpr int kBnCh = 4;                     // channels per workgroup in the per-channel kernels (more, smaller workgroups:
                                             // C/4 of them, 64 image lanes each -- these kernels are latency-bound)
constexpr int kBnLanes = kThreads / kBnCh;   // images summed in parallel per channel

// ------------------------------------------------------------------------------------------------
// gate forward: one workgroup per image
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void gate_fwd_kernel(const float* __restrict__ mom, const float* __restrict__ wq,
                                                            const float* __restrict__ wk, int ks,
                                                            float* __restrict__ gate, int C, int HW, int d) {
  extern __shared__ float sm[];
  const int p = (ks - 1) / 2;
  float* ys = sm;                 // [C + 2p], zero padded
  float* qk = ys + C + 2 * p;     // [C]
  const int b = blockIdx.x, tid = threadIdx.x;
  const float inv_hw = 1.0f / (float)HW;
  for (int i = tid; i < C + 2 * p; i += kThreads) {
    const int c = i - p;
    ys[i] = (c >= 0 && c < C) ? mom[((size_t)b * C + c) * M_REC + M_SX] * inv_hw : 0.f;
  }
  __syncthreads();
  for (int c = tid; c < C; c += kThreads) {
    float q = 0.f, k = 0.f;
    for (int j = 0; j < ks; ++j) {
      q = fmaf(wq[j], ys[c + j], q);
      k = fmaf(wk[j], ys[c + j], k);
    }
    qk[c] = q * k;
  }
  __syncthreads();
  const int G = C / d;
  const float s = rsqrtf((float)d);
  for (int g = tid; g < G; g += kThreads) {
    float acc = 0.f;
    for (int i = 0; i < d; ++i) acc += qk[g * d + i];
    gate[(size_t)b * G + g] = 1.0f / (1.0f + expf(-acc * s));
  }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm statistics of m = a*V + lam*o from the moments: kBnCh channels x kBnLanes image lanes
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void bn_fwd_kernel(
    const float* __restrict__ mom, const float* __restrict__ gate, const float* __restrict__ lam,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ run_mean,
    float* __restrict__ run_var, int training, float momentum, float eps, float* __restrict__ sc,
    float* __restrict__ sh, float* __restrict__ save_mean, float* __restrict__ save_inv, int B, int C, int HW, int d) {
  __shared__ double r1[kBnLanes][kBnCh], r2[kBnLanes][kBnCh];
  const int cc = threadIdx.x % kBnCh, bl = threadIdx.x / kBnCh;
  const int c = blockIdx.x * kBnCh + cc;
  const bool live = c < C;
  const int G = C / d;
  double mean = 0.0, var = 1.0;
  if (training) {
    double s1 = 0.0, s2 = 0.0;
    if (live) {
      const float l = lam ? lam[c] : 0.f;
#pragma unroll 4
      for (int b = bl; b < B; b += kBnLanes) {
        const float* m = mom + ((size_t)b * C + c) * M_REC;
        const double a = gate[(size_t)b * G + c / d];
        const RawMoments r = raw_moments(m, (double)HW);
        s1 += a * r.sv + (double)l * r.so;
        s2 += a * a * r.svv + 2.0 * a * l * r.svo + (double)l * l * r.soo;
      }
    }
    r1[bl][cc] = s1; r2[bl][cc] = s2;
    __syncthreads();
    if (bl == 0 && live) {
      s1 = 0.0; s2 = 0.0;
      for (int i = 0; i < kBnLanes; ++i) { s1 += r1[i][cc]; s2 += r2[i][cc]; }
      const double M = (double)B * HW;
      mean = s1 / M;
      var = s2 / M - mean * mean;
      if (var < 0.0) var = 0.0;
      const double unbiased = var * (M / (M > 1.0 ? M - 1.0 : 1.0));
      run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mean);
      run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unbiased);
    }
  } else if (bl == 0 && live) {
    mean = run_mean[c];
    var = run_var[c];
  }
  if (bl == 0 && live) {
    const double inv = 1.0 / sqrt(var + (double)eps);
    const double scale = gamma[c] * inv;
    sc[c] = (float)scale;
    sh[c] = (float)(beta[c] - scale * mean);
    save_mean[c] = (float)mean;
    save_inv[c] = (float)inv;
  }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm backward constants, dgamma, dbeta, dlambda
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void bn_bwd_kernel(
    const float* __restrict__ mom, const float* __restrict__ bmom, const float* __restrict__ gate,
    const float* __restrict__ lam, const float* __restrict__ gamma, const float* __restrict__ dp,
    const float* __restrict__ save_mean, const float* __restrict__ save_inv, int training, float* __restrict__ cb,
    float* __restrict__ cb_lo, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dlam, int B,
    int C, int HW, int d) {
  __shared__ double r1[kBnLanes][kBnCh], r2[kBnLanes][kBnCh];
  __shared__ double coef[4][kBnCh];
  const int cc = threadIdx.x % kBnCh, bl = threadIdx.x / kBnCh;
  const int c = blockIdx.x * kBnCh + cc;
  const bool live = c < C;
  const int G = C / d;
  const float l = (live && lam) ? lam[c] : 0.f;
  double s1 = 0.0, s2 = 0.0;
  if (live) {
#pragma unroll 4
    for (int b = bl; b < B; b += kBnLanes) {
      const BwdMoments bm = load_bmom(bmom, b, c, C);
      const float* m = mom + ((size_t)b * C + c) * M_REC;
      const float a = gate[(size_t)b * G + c / d];
      const double dpb = dp ? dp[b] : 1.f;
      // (bmom's sums over dOut*V and dOut*o are about the forward pivots: un-shift in double)
      const double dv = (double)bm.dv + (double)m[M_PV] * bm.d, dO = (double)bm.d_o + (double)m[M_PO] * bm.d;
      s1 += dpb * bm.d;
      s2 += dpb * ((double)a * dv + (double)l * dO);
    }
  }
  r1[bl][cc] = s1; r2[bl][cc] = s2;
  __syncthreads();
  if (bl == 0 && live) {
    s1 = 0.0; s2 = 0.0;
    for (int i = 0; i < kBnLanes; ++i) { s1 += r1[i][cc]; s2 += r2[i][cc]; }
    double e = 1.0, f = 0.0, Gc = 0.0, Hc = 0.0;
    if (gamma) {
      const double mean = save_mean[c], inv = save_inv[c];
      const double dbe = s1;
      const double dga = inv * (s2 - mean * s1);
      e = gamma[c] * inv;
      if (training) {
        const double M = (double)B * HW;
        const double c1 = dbe / M, c2 = dga / M;
        f = -e * inv * c2;
        Gc = f * l;
        Hc = e * (-c1 + inv * mean * c2);
      }
      dgamma[c] = (float)dga;
      dbeta[c] = (float)dbe;
    }
    coef[0][cc] = e; coef[1][cc] = f; coef[2][cc] = Gc; coef[3][cc] = Hc;
    cb[c * 4 + 0] = (float)e; cb[c * 4 + 1] = (float)f; cb[c * 4 + 2] = (float)Gc; cb[c * 4 + 3] = (float)Hc;
    if (cb_lo) {        // the float remainders: the closed-form gate gradient cancels large terms and needs e..H beyond fp32
      cb_lo[c * 4 + 0] = (float)(e - (double)(float)e);   cb_lo[c * 4 + 1] = (float)(f - (double)(float)f);
      cb_lo[c * 4 + 2] = (float)(Gc - (double)(float)Gc); cb_lo[c * 4 + 3] = (float)(Hc - (double)(float)Hc);
    }
  }
  __syncthreads();
  if (!dlam) return;
  // dlambda[c] = sum over images and pixels of dm * o.  o sits at |mean| >> sigma in deep stages, dm sums to zero over
  // the batch (train-mode BatchNorm), so the sum is taken about the planes' pivots: sum dm*(o - pO_b) from the shifted
  // moments, plus (pO_b - P0) * (sum of dm over image b), plus P0 * (sum of dm over the batch) -- which is zero by
  // construction of (f, G, H) in training mode and is left out there instead of being re-derived from cancelling terms.
  double s3 = 0.0, st = 0.0;
  if (live) {
    const double e = coef[0][cc], f = coef[1][cc], Gc = coef[2][cc], Hc = coef[3][cc];
    const double P0 = mom[(size_t)c * M_REC + M_PO];                 // image 0's pivot of o for this channel
    for (int b = bl; b < B; b += kBnLanes) {
      const float* m = mom + ((size_t)b * C + c) * M_REC;
      const BwdMoments bm = load_bmom(bmom, b, c, C);
      const double a = gate[(size_t)b * G + c / d];
      const double dpb = dp ? dp[b] : 1.f;
      const double n = (double)HW, pv = m[M_PV], po = m[M_PO];
      const double so = m[M_SO], sv = m[M_SV];                       // shifted: sum (o - po), sum (V - pv)
      const double dm_o = e * dpb * bm.d_o + f * a * ((double)m[M_SVO] + pv * so) + Gc * ((double)m[M_SOO] + po * so) + Hc * so;
      const double dm_1 = e * dpb * bm.d + f * a * (sv + n * pv) + Gc * (so + n * po) + Hc * n;
      s3 += dm_o + (po - P0) * dm_1;
      st += dm_1;
    }
  }
  r1[bl][cc] = s3; r2[bl][cc] = st;
  __syncthreads();
  if (bl == 0 && live) {
    s3 = 0.0; st = 0.0;
    for (int i = 0; i < kBnLanes; ++i) { s3 += r1[i][cc]; st += r2[i][cc]; }
    if (!(gamma && training)) s3 += (double)mom[(size_t)c * M_REC + M_PO] * st;
    dlam[c] = (float)s3;
  }
}

// ------------------------------------------------------------------------------------------------
// gate backward: one workgroup per image
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* scratch /*[kWaves]*/) {
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

__global__ __launch_bounds__(kThreads) void gate_bwd_kernel(
    const float* __restrict__ mom, const float* __restrict__ bmom, const float* __restrict__ gate,
    const float* __restrict__ cb, const float* __restrict__ cb_lo, const float* __restrict__ dp,
    const float* __restrict__ wq, const float* __restrict__ wk, int ks, float* __restrict__ dyx,
    float* __restrict__ dwqk_part, int C, int HW, int d, float* __restrict__ tok_part, int tok_bands) {
  extern __shared__ float sm[];
  const int p = (ks - 1) / 2;
  const int CPD = C + 2 * p;
  float* ys = sm;            // [CPD] padded y
  float* qs = ys + CPD;      // [C]
  float* kk = qs + C;        // [C]
  float* dqs = kk + C;       // [CPD] padded dq   (first used as da-per-channel scratch)
  float* dks = dqs + CPD;    // [CPD] padded dk
  float* dl = dks + CPD;     // [G]
  float* scratch = dl + C / d;   // [kWaves]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int G = C / d;
  const float inv_hw = 1.0f / (float)HW;
  const float dpb = dp ? dp[b] : 1.f;
  for (int i = tid; i < CPD; i += kThreads) {
    const int c = i - p;
    ys[i] = (c >= 0 && c < C) ? mom[((size_t)b * C + c) * M_REC + M_SX] * inv_hw : 0.f;
    dqs[i] = 0.f;
    dks[i] = 0.f;
  }
  __syncthreads();
  for (int c = tid; c < C; c += kThreads) {
    float q = 0.f, k = 0.f;
    for (int j = 0; j < ks; ++j) {
      q = fmaf(wq[j], ys[c + j], q);
      k = fmaf(wk[j], ys[c + j], k);
    }
    qs[c] = q;
    kk[c] = k;
    const float* m = mom + ((size_t)b * C + c) * M_REC;
    const BwdMoments bm = load_bmom(bmom, b, c, C);
    const float a = gate[(size_t)b * G + c / d];
    double e = 1.0, f = 0.0, Gc = 0.0, Hc = 0.0;
    if (cb) { e = cb[c * 4 + 0]; f = cb[c * 4 + 1]; Gc = cb[c * 4 + 2]; Hc = cb[c * 4 + 3]; }
    if (cb && cb_lo) { e += cb_lo[c * 4 + 0]; f += cb_lo[c * 4 + 1]; Gc += cb_lo[c * 4 + 2]; Hc += cb_lo[c * 4 + 3]; }
    // sum_hw dm * V  for this channel
    const RawMoments r = raw_moments(m, (double)HW);
    dqs[p + c] = (float)(e * dpb * ((double)bm.dv + (double)m[M_PV] * bm.d) + f * a * r.svv + Gc * r.svo + Hc * r.sv);
  }
  __syncthreads();
  const float s = rsqrtf((float)d);
  for (int g = tid; g < G; g += kThreads) {
    float da = 0.f;
    for (int i = 0; i < d; ++i) da += dqs[p + g * d + i];
    const float a = gate[(size_t)b * G + g];
    dl[g] = da * a * (1.f - a) * s;
  }
  __syncthreads();
  for (int c = tid; c < C; c += kThreads) {
    const float t = dl[c / d];
    dqs[p + c] = t * kk[c];
    dks[p + c] = t * qs[c];
  }
  __syncthreads();
  for (int c = tid; c < C; c += kThreads) {
    // dy[c] = sum_j wq[j]*dq[c-j+p] + wk[j]*dk[c-j+p]; padded index of (c-j+p) is c-j+2p
    float dy = 0.f;
    for (int j = 0; j < ks; ++j) {
      dy = fmaf(wq[j], dqs[c - j + 2 * p], dy);
      dy = fmaf(wk[j], dks[c - j + 2 * p], dy);
    }
    dyx[(size_t)b * C + c] = dy * inv_hw;
    if (tok_part) {
      // token path: every map token's LN_x output receives dy/hw on top of what mrla_token_apply_bwd wrote, so the
      // LayerNorm parameter partials of this (image, channel) gain dy/hw * sum xhat and dy/hw * hw
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
    aq = block_sum(aq, scratch);
    ak = block_sum(ak, scratch);
    if (tid == 0) {
      dwqk_part[(size_t)b * 2 * ks + j] = aq;
      dwqk_part[(size_t)b * 2 * ks + ks + j] = ak;
    }
  }
}

// out[i] = sum_r in[r, i].  A workgroup covers `cpb` columns (power of two <= 64) with 256/cpb row lanes each, so
// narrow matrices (the [b, 2k] Wq/Wk partials) still get hundreds of loads in flight; fixed summation order.
__global__ __launch_bounds__(kThreads) void reduce_rows_kernel(const float* in, float* out,       // (out may be row 0 of in)
                                                               int rows, int n, int cpb) {
  __shared__ double part[kThreads];
  const int cx = threadIdx.x % cpb, ry = threadIdx.x / cpb, rl = kThreads / cpb;
  const int i = blockIdx.x * cpb + cx;
  double s = 0.0;
  if (i < n)
    for (int r = ry; r < rows; r += rl) s += in[(size_t)r * n + i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (ry == 0 && i < n) {
    for (int k = 1; k < rl; ++k) s += part[k * cpb + cx];
    out[i] = (float)s;
  }
}

// Two independent column sums in one launch (the dWv and the dWq / dWk partials of one block's backward): blocks
// [0, g1) take the first matrix, the rest the second.
struct ReduceJob { const float* in; float* out; int rows, n, cpb; };
__global__ __launch_bounds__(kThreads) void reduce_rows2_kernel(ReduceJob a, ReduceJob b, int g1) {
  __shared__ double part[kThreads];
  const bool first = (int)blockIdx.x < g1;
  const ReduceJob j = first ? a : b;
  const int blk = first ? (int)blockIdx.x : (int)blockIdx.x - g1;
  const int cx = threadIdx.x % j.cpb, ry = threadIdx.x / j.cpb, rl = kThreads / j.cpb;
  const int i = blk * j.cpb + cx;
  double s = 0.0;
  if (i < j.n)
    for (int r = ry; r < j.rows; r += rl) s += j.in[(size_t)r * j.n + i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (ry == 0 && i < j.n) {
    for (int k = 1; k < rl; ++k) s += part[k * j.cpb + cx];
    j.out[i] = (float)s;
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
int launch_gate_fwd(const float* mom, const float* wq, const float* wk, int ks, float* gate, int B, int C, int HW,
                    int d, hipStream_t st) {
  const size_t lds = (size_t)(2 * C + 2 * ((ks - 1) / 2)) * sizeof(float);
  if (lds > 64 * 1024) return MRLA_EUNSUPPORTED;
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(B), dim3(kThreads), lds, st, mom, wq, wk, ks, gate, C, HW, d);
  return hip_status(hipGetLastError());
}

int launch_bn_fwd(const float* mom, const float* gate, const float* lam, const float* gamma, const float* beta,
                  float* run_mean, float* run_var, int training, float momentum, float eps, float* sc, float* sh,
                  float* save_mean, float* save_inv, int B, int C, int HW, int d, hipStream_t st) {
  hipLaunchKernelGGL(bn_fwd_kernel, dim3((C + kBnCh - 1) / kBnCh), dim3(kThreads), 0, st, mom, gate, lam, gamma, beta,
                     run_mean, run_var, training, momentum, eps, sc, sh, save_mean, save_inv, B, C, HW, d);
  return hip_status(hipGetLastError());
}

int launch_bn_bwd(const float* mom, const float* bmom, const float* gate, const float* lam, const float* gamma,
                  const float* dp, const float* save_mean, const float* save_inv, int training, float* cb, float* cb_lo,
                  float* dgamma, float* dbeta, float* dlam, int B, int C, int HW, int d, hipStream_t st) {
  hipLaunchKernelGGL(bn_bwd_kernel, dim3((C + kBnCh - 1) / kBnCh), dim3(kThreads), 0, st, mom, bmom, gate, lam, gamma,
                     dp, save_mean, save_inv, training, cb, cb_lo, dgamma, dbeta, dlam, B, C, HW, d);
  return hip_status(hipGetLastError());
}

int launch_gate_bwd(const float* mom, const float* bmom, const float* gate, const float* cb, const float* cb_lo,
                    const float* dp, const float* wq, const float* wk, int ks, float* dyx, float* dwqk_part, int B, int C, int HW,
                    int d, hipStream_t st, float* tok_part, int tok_bands) {
  const int p = (ks - 1) / 2;
  const size_t lds = (size_t)(3 * (C + 2 * p) + 2 * C + C / d + kWaves) * sizeof(float);
  if (lds > 64 * 1024) return MRLA_EUNSUPPORTED;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(B), dim3(kThreads), lds, st, mom, bmom, gate, cb, cb_lo, dp, wq, wk, ks, dyx,
                     dwqk_part, C, HW, d, tok_part, tok_bands);
  return hip_status(hipGetLastError());
}

static int reduce_cpb(int rows, int n) {
  int cpb = 64;
  while (cpb > 1 && cpb / 2 >= n) cpb >>= 1;                  // narrow matrices: fewer columns, more row lanes
  if (rows < 16) return 64;
  // too few workgroups to hide the latency of their `rows / row-lanes` dependent loads (45 for the token path's
  // [256, 2880] partials: 18 us for 3 MB): narrower column groups, more row lanes, while a lane still sums >= 4 rows
  while (cpb > 8 && (n + cpb - 1) / cpb < 512 && rows >= 4 * (kThreads / (cpb / 2))) cpb >>= 1;
  return cpb;
}

int launch_reduce_rows(const float* in, float* out, int rows, int n, hipStream_t st) {
  const int cpb = reduce_cpb(rows, n);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((n + cpb - 1) / cpb), dim3(kThreads), 0, st, in, out, rows, n, cpb);
  return hip_status(hipGetLastError());
}

// buf[0, i] = sum_r buf[r, i] in place (every column is read by the threads that hold it before its sum is written)
int launch_fold_rows(float* buf, int rows, int n, hipStream_t st) {
  if (rows <= 1) return MRLA_OK;
  return launch_reduce_rows(buf, buf, rows, n, st);
}

int launch_reduce_rows2(const float* in1, float* out1, int rows1, int n1, const float* in2, float* out2, int rows2, int n2,
                        hipStream_t st) {
  const ReduceJob a{in1, out1, rows1, n1, reduce_cpb(rows1, n1)}, b{in2, out2, rows2, n2, reduce_cpb(rows2, n2)};
  const int g1 = (n1 + a.cpb - 1) / a.cpb, g2 = (n2 + b.cpb - 1) / b.cpb;
  hipLaunchKernelGGL(reduce_rows2_kernel, dim3(g1 + g2), dim3(kThreads), 0, st, a, b, g1);
  return hip_status(hipGetLastError());
}

}  // namespace mrla
