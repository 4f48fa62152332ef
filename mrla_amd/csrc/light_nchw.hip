// MRLA-light streaming kernels for NCHW-contiguous activations (gfx950).
//
// What they compute (reference: resnet/models/modules/
// mrla_light_module.py:52-74, resnet/models/resnet_mrla_light.py:40-43,113-116):
//
//   V   = act(dwconv3x3(x, wv))                     per (image, channel) plane
//   out = res*x + A*V + B*o + C                     A,B,C per (image, channel)   [apply_fwd]
//   dm  = E*dOut + F*V + G*o + H                    E..H  per (image, channel)   [apply_bwd]
//   do  = lam*dm ; dU = a*dm*act'(U) ; dx = res*dOut + dwconv3x3^T(dU) + dyx
//
// and the per-(image, channel) moments that let the tiny "gate" kernels (gate.hip) evaluate the
// sigmoid gate, the train-mode BatchNorm statistics of m = a*V + lam*o and every parameter gradient in
// closed form WITHOUT materialising V or m in HBM:
//   forward : Sx, SV, So, SVV, SVo, Soo      backward: D = sum dOut, DV = sum dOut*V, Do = sum dOut*o
//
// Data movement: a workgroup owns a "slab" = CP consecutive channel planes of one image, which is one
// contiguous run of CP*H*W elements in NCHW.  The slab is copied HBM -> LDS with 16-byte-per-lane
// loads, the 3x3 stencil is evaluated by row-marching waves (lanes = columns of PW side-by-side
// planes, vertical neighbours in a register sliding window, horizontal neighbours by DPP wave
// shifts), results are staged in LDS and written back with 16-byte-per-lane stores.  HBM traffic per
// element of N = b*c*h*w:  stats_fwd 2 reads, apply_fwd 2 reads + 1 write, stats_bwd 3 reads,
// apply_bwd 3 reads + 2 writes.
#include "mrla_device.h"
#include "mrla_kernels.h"
#include "mrla_march.h"

namespace mrla {

// ------------------------------------------------------------------------------------------------
// forward statistics:  mom[b, c, 0..5]
// ------------------------------------------------------------------------------------------------
template <typename T, bool GELU>
__global__ __launch_bounds__(kThreads) void light_stats_fwd_nchw(
    const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv,
    float* __restrict__ mom, SlabGeo g) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* xs = reinterpret_cast<T*>(smem);
  T* os = xs + g.astride;
  float* red = reinterpret_cast<float*>(os + (o ? g.astride : 0));   // [ntasks][PW][M_N]
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int c0 = blockIdx.x * g.CP;
  const int np = min(g.CP, g.C - c0);
  const int n = np * g.HW;
  const int ntasks = g.NG * g.NB;
  const int b_end = min(g.B, (int)(blockIdx.y + 1) * g.BG);
  for (int b = blockIdx.y * g.BG; b < b_end; ++b) {
    const size_t off = ((size_t)b * g.C + c0) * g.HW;
    slab_load(xs, x + off, n, tid);
    if (o) slab_load(os, o + off, n, tid);
    __syncthreads();
    for (int task = wave; task < ntasks; task += kWaves) {
      const LaneTask t = make_task(g, task, np, lane);
      if (!t.live) continue;
      const T* xp = xs + t.p * g.HW;
      const T* op = os + t.p * g.HW;
      float w[9];
      load_w9(w, wv, c0 + t.p);
      float acc[M_N] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      Row3 ra = load_row3(xp, t.r0 - 1, g, t);
      Row3 rb = load_row3(xp, t.r0, g, t);
      for (int r = t.r0; r < t.r1; ++r) {
        const Row3 rc = load_row3(xp, r + 1, g, t);
        float v = conv9(w, ra, rb, rc);
        if (GELU) v = gelu_f(v);
        acc[M_SX] += rb.c;
        acc[M_SV] += v;
        acc[M_SVV] = fmaf(v, v, acc[M_SVV]);
        if (o) {
          const float ov = to_f(op[r * g.W + t.col]);
          acc[M_SO] += ov;
          acc[M_SVO] = fmaf(v, ov, acc[M_SVO]);
          acc[M_SOO] = fmaf(ov, ov, acc[M_SOO]);
        }
        ra = rb; rb = rc;
      }
#pragma unroll
      for (int k = 0; k < M_N; ++k) {
        const float s = seg_sum(t.valid ? acc[k] : 0.f, t.col, g.W);
        if (t.valid && t.col == 0) red[(task * g.PW + t.pl) * M_N + k] = s;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < np * M_N; idx += kThreads) {
      const int p = idx / M_N, k = idx - p * M_N;
      const int grp = p / g.PW, pl = p - grp * g.PW;
      float s = 0.f;
      for (int band = 0; band < g.NB; ++band) s += red[((grp * g.NB + band) * g.PW + pl) * M_N + k];
      mom[((size_t)b * g.C + c0 + p) * M_N + k] = s;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// forward apply:  out = res*x + A*V + B*o + C
// ------------------------------------------------------------------------------------------------
template <typename T, bool GELU>
__global__ __launch_bounds__(kThreads) void light_apply_fwd_nchw(
    const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv,
    const float* __restrict__ gate /*[b,g]*/, const float* __restrict__ sc, const float* __restrict__ sh,
    const float* __restrict__ lam, const float* __restrict__ dp, T* __restrict__ out, SlabGeo g, int d, int res) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* xs = reinterpret_cast<T*>(smem);
  T* os = xs + g.astride;          // o on input, `out` staging on output (same lane reads then writes)
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int c0 = blockIdx.x * g.CP;
  const int np = min(g.CP, g.C - c0);
  const int n = np * g.HW;
  const int ntasks = g.NG * g.NB;
  const int G = g.C / d;
  const int b_end = min(g.B, (int)(blockIdx.y + 1) * g.BG);
  for (int b = blockIdx.y * g.BG; b < b_end; ++b) {
    const size_t off = ((size_t)b * g.C + c0) * g.HW;
    slab_load(xs, x + off, n, tid);
    if (o) slab_load(os, o + off, n, tid);
    __syncthreads();
    const float dpb = dp ? dp[b] : 1.f;
    for (int task = wave; task < ntasks; task += kWaves) {
      const LaneTask t = make_task(g, task, np, lane);
      if (!t.live) continue;
      const int c = c0 + t.p;
      const T* xp = xs + t.p * g.HW;
      T* op = os + t.p * g.HW;
      float w[9];
      load_w9(w, wv, c);
      const float scale = dpb * (sc ? sc[c] : 1.f);
      const float A = scale * gate[(size_t)b * G + c / d];
      const float Bc = (o && lam) ? scale * lam[c] : (o ? scale : 0.f);
      const float Cc = sh ? dpb * sh[c] : 0.f;
      Row3 ra = load_row3(xp, t.r0 - 1, g, t);
      Row3 rb = load_row3(xp, t.r0, g, t);
      for (int r = t.r0; r < t.r1; ++r) {
        const Row3 rc = load_row3(xp, r + 1, g, t);
        float v = conv9(w, ra, rb, rc);
        if (GELU) v = gelu_f(v);
        float y = fmaf(A, v, Cc);
        if (o) y = fmaf(Bc, to_f(op[r * g.W + t.col]), y);
        if (res) y += rb.c;
        if (t.valid) op[r * g.W + t.col] = from_f<T>(y);
        ra = rb; rb = rc;
      }
    }
    __syncthreads();
    slab_store(out + off, os, n, tid);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// backward statistics:  bmom[b, c, 0..2] = (sum dOut, sum dOut*V, sum dOut*o)
// ------------------------------------------------------------------------------------------------
template <typename T, bool GELU>
__global__ __launch_bounds__(kThreads) void light_stats_bwd_nchw(
    const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ o,
    const float* __restrict__ wv, float* __restrict__ bmom, SlabGeo g) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* xs = reinterpret_cast<T*>(smem);
  T* gs = xs + g.astride;
  T* os = gs + g.astride;
  float* red = reinterpret_cast<float*>(os + (o ? g.astride : 0));
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int c0 = blockIdx.x * g.CP;
  const int np = min(g.CP, g.C - c0);
  const int n = np * g.HW;
  const int ntasks = g.NG * g.NB;
  const int b_end = min(g.B, (int)(blockIdx.y + 1) * g.BG);
  for (int b = blockIdx.y * g.BG; b < b_end; ++b) {
    const size_t off = ((size_t)b * g.C + c0) * g.HW;
    slab_load(xs, x + off, n, tid);
    slab_load(gs, dout + off, n, tid);
    if (o) slab_load(os, o + off, n, tid);
    __syncthreads();
    for (int task = wave; task < ntasks; task += kWaves) {
      const LaneTask t = make_task(g, task, np, lane);
      if (!t.live) continue;
      const T* xp = xs + t.p * g.HW;
      const T* gp = gs + t.p * g.HW;
      const T* op = os + t.p * g.HW;
      float w[9];
      load_w9(w, wv, c0 + t.p);
      float acc[D_N] = {0.f, 0.f, 0.f};
      Row3 ra = load_row3(xp, t.r0 - 1, g, t);
      Row3 rb = load_row3(xp, t.r0, g, t);
      for (int r = t.r0; r < t.r1; ++r) {
        const Row3 rc = load_row3(xp, r + 1, g, t);
        float v = conv9(w, ra, rb, rc);
        if (GELU) v = gelu_f(v);
        const float gv = to_f(gp[r * g.W + t.col]);
        acc[D_D] += gv;
        acc[D_DV] = fmaf(gv, v, acc[D_DV]);
        if (o) acc[D_DO] = fmaf(gv, to_f(op[r * g.W + t.col]), acc[D_DO]);
        ra = rb; rb = rc;
      }
#pragma unroll
      for (int k = 0; k < D_N; ++k) {
        const float s = seg_sum(t.valid ? acc[k] : 0.f, t.col, g.W);
        if (t.valid && t.col == 0) red[(task * g.PW + t.pl) * D_N + k] = s;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < np * D_N; idx += kThreads) {
      const int p = idx / D_N, k = idx - p * D_N;
      const int grp = p / g.PW, pl = p - grp * g.PW;
      float s = 0.f;
      for (int band = 0; band < g.NB; ++band) s += red[((grp * g.NB + band) * g.PW + pl) * D_N + k];
      bmom[((size_t)b * g.C + c0 + p) * D_N + k] = s;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// backward apply:  dx, do, and per-(image group, channel) partial sums of dwv
// ------------------------------------------------------------------------------------------------
template <typename T, bool GELU, int TPW>
__global__ __launch_bounds__(kThreads) void light_apply_bwd_nchw(
    const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv,
    const float* __restrict__ gate /*[b,g]*/, const float* __restrict__ cb /*[c,4]: e,f,G,H or null*/,
    const float* __restrict__ lam, const float* __restrict__ dp, const float* __restrict__ dyx /*[b,c]*/,
    T* __restrict__ dx, T* __restrict__ dprev, float* __restrict__ dwv_part /*[gridDim.y, c, 9]*/,
    SlabGeo g, int d, int res) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* xs = reinterpret_cast<T*>(smem);
  T* gs = xs + g.astride;
  T* dxs = gs + g.astride;
  T* os = dxs + g.astride;
  T* dos = os + (o ? g.astride : 0);
  float* red = reinterpret_cast<float*>(dos + (o ? g.astride : 0));   // [ntasks][PW][9]
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int c0 = blockIdx.x * g.CP;
  const int np = min(g.CP, g.C - c0);
  const int n = np * g.HW;
  const int ntasks = g.NG * g.NB;
  const int G = g.C / d;
  const int b_end = min(g.B, (int)(blockIdx.y + 1) * g.BG);
  // wgrad accumulators live across the image loop; a wave revisits the same (task -> plane) mapping for
  // every image, so its TPW = ceil(ntasks / 4) task slots are kept in registers.
  float wg[TPW][9];
#pragma unroll
  for (int s = 0; s < TPW; ++s)
#pragma unroll
    for (int k = 0; k < 9; ++k) wg[s][k] = 0.f;

  for (int b = blockIdx.y * g.BG; b < b_end; ++b) {
    const size_t off = ((size_t)b * g.C + c0) * g.HW;
    slab_load(xs, x + off, n, tid);
    slab_load(gs, dout + off, n, tid);
    if (o) slab_load(os, o + off, n, tid);
    __syncthreads();
    const float dpb = dp ? dp[b] : 1.f;
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
      const int task = wave + s * kWaves;
      const LaneTask t = make_task(g, min(task, ntasks - 1), np, lane);
      if (task < ntasks && t.live) {
        const int c = c0 + t.p;
        const T* xp = xs + t.p * g.HW;
        const T* gp = gs + t.p * g.HW;
        const T* op = os + t.p * g.HW;
        T* dxp = dxs + t.p * g.HW;
        T* dop = dos + t.p * g.HW;
        float w[9];
        load_w9(w, wv, c);
        const float a = gate[(size_t)b * G + c / d];
        const float lm = lam ? lam[c] : 1.f;
        float E = dpb, F = 0.f, Gc = 0.f, Hc = 0.f;
        if (cb) { E = cb[c * 4 + 0] * dpb; F = cb[c * 4 + 1] * a; Gc = cb[c * 4 + 2]; Hc = cb[c * 4 + 3]; }
        const float dy = dyx[(size_t)b * g.C + c];
        // march rr over [r0-1, r1]: produce dU[rr]; emit dx[rr-1] once dU[rr-2..rr] are known
        Row3 xa = load_row3(xp, t.r0 - 2, g, t);   // x[rr-1]
        Row3 xb = load_row3(xp, t.r0 - 1, g, t);   // x[rr]
        Row3 ua = {0.f, 0.f, 0.f};                 // dU[rr-2]
        Row3 ub = {0.f, 0.f, 0.f};                 // dU[rr-1]
        float gprev = 0.f;                         // dOut[rr-1]
        for (int rr = t.r0 - 1; rr <= t.r1; ++rr) {
          const Row3 xc = load_row3(xp, rr + 1, g, t);
          Row3 uc = {0.f, 0.f, 0.f};
          float gcur = 0.f;
          if (rr >= 0 && rr < g.H) {                         // wave-uniform
            const float u = conv9(w, xa, xb, xc);
            const float v = GELU ? gelu_f(u) : u;
            gcur = to_f(gp[rr * g.W + t.col]);
            float dm = fmaf(E, gcur, Hc);
            dm = fmaf(F, v, dm);
            if (o) dm = fmaf(Gc, to_f(op[rr * g.W + t.col]), dm);
            float du = a * dm;
            if (GELU) du *= gelu_grad_f(u);
            if (!t.valid) du = 0.f;
            uc.c = du;
            if (rr >= t.r0 && rr < t.r1) {                   // wave-uniform: this band owns row rr
              if (o && t.valid) dop[rr * g.W + t.col] = from_f<T>(lm * dm);
              wg[s][0] = fmaf(du, xa.l, wg[s][0]); wg[s][1] = fmaf(du, xa.c, wg[s][1]); wg[s][2] = fmaf(du, xa.r, wg[s][2]);
              wg[s][3] = fmaf(du, xb.l, wg[s][3]); wg[s][4] = fmaf(du, xb.c, wg[s][4]); wg[s][5] = fmaf(du, xb.r, wg[s][5]);
              wg[s][6] = fmaf(du, xc.l, wg[s][6]); wg[s][7] = fmaf(du, xc.c, wg[s][7]); wg[s][8] = fmaf(du, xc.r, wg[s][8]);
            }
          }
          uc.l = lane_prev(uc.c) * t.lmask;
          uc.r = lane_next(uc.c) * t.rmask;
          const int ro = rr - 1;                             // row whose dx is now complete
          if (ro >= t.r0 && ro < t.r1) {
            // dx[ro][w] = sum_{i,j} wv[i][j] * dU[ro-i+1][w-j+1]
            float s9 = w[0] * uc.r;
            s9 = fmaf(w[1], uc.c, s9); s9 = fmaf(w[2], uc.l, s9);
            s9 = fmaf(w[3], ub.r, s9); s9 = fmaf(w[4], ub.c, s9); s9 = fmaf(w[5], ub.l, s9);
            s9 = fmaf(w[6], ua.r, s9); s9 = fmaf(w[7], ua.c, s9); s9 = fmaf(w[8], ua.l, s9);
            float y = s9 + dy;
            if (res) y += gprev;
            if (t.valid) dxp[ro * g.W + t.col] = from_f<T>(y);
          }
          xa = xb; xb = xc; ua = ub; ub = uc; gprev = gcur;
        }
      }
    }
    __syncthreads();
    slab_store(dx + off, dxs, n, tid);
    if (o) slab_store(dprev + off, dos, n, tid);
    __syncthreads();
  }
  // reduce the wgrad accumulators: lanes of a plane row -> bands -> one value per (plane, tap)
#pragma unroll
  for (int s = 0; s < TPW; ++s) {
    const int task = wave + s * kWaves;
    const LaneTask t = make_task(g, min(task, ntasks - 1), np, lane);
    if (task < ntasks && t.live) {
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float v = seg_sum(t.valid ? wg[s][k] : 0.f, t.col, g.W);
        if (t.valid && t.col == 0) red[(task * g.PW + t.pl) * 9 + k] = v;
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
// host-side launchers
// ------------------------------------------------------------------------------------------------
template <typename K>
static hipError_t set_lds(K kernel, size_t bytes) {
  if (bytes <= 48 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

#define MRLA_DISPATCH_T_ACT(DT, ACT, CALL)                                             \
  switch (DT) {                                                                        \
    case MRLA_F32:  if (ACT) { CALL(float, true); } else { CALL(float, false); } break;  \
    case MRLA_BF16: if (ACT) { CALL(bf16_t, true); } else { CALL(bf16_t, false); } break; \
    case MRLA_F16:  if (ACT) { CALL(f16_t, true); } else { CALL(f16_t, false); } break;  \
    default: return MRLA_EINVAL;                                                       \
  }

int launch_light_stats_fwd_nchw(const void* x, const void* o, const float* wv, float* mom, const SlabGeo& g,
                                int dtype, int act, hipStream_t st) {
  const size_t es = dtype_size(dtype);
  const size_t lds = (size_t)g.astride * es * (o ? 2 : 1) + (size_t)g.NG * g.NB * g.PW * M_N * sizeof(float);
  const dim3 grid(g.slabs, (g.B + g.BG - 1) / g.BG);
#define CALL(T, A)                                                                                  \
  {                                                                                                 \
    if (set_lds(light_stats_fwd_nchw<T, A>, lds) != hipSuccess) return MRLA_EHIP;                     \
    hipLaunchKernelGGL((light_stats_fwd_nchw<T, A>), grid, dim3(kThreads), lds, st, (const T*)x,      \
                       (const T*)o, wv, mom, g);                                                    \
  }
  MRLA_DISPATCH_T_ACT(dtype, act, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_light_apply_fwd_nchw(const void* x, const void* o, const float* wv, const float* gate, const float* sc,
                                const float* sh, const float* lam, const float* dp, void* out, const SlabGeo& g,
                                int d, int res, int dtype, int act, hipStream_t st) {
  const size_t es = dtype_size(dtype);
  const size_t lds = (size_t)g.astride * es * 2;
  const dim3 grid(g.slabs, (g.B + g.BG - 1) / g.BG);
#define CALL(T, A)                                                                                  \
  {                                                                                                 \
    if (set_lds(light_apply_fwd_nchw<T, A>, lds) != hipSuccess) return MRLA_EHIP;                     \
    hipLaunchKernelGGL((light_apply_fwd_nchw<T, A>), grid, dim3(kThreads), lds, st, (const T*)x,      \
                       (const T*)o, wv, gate, sc, sh, lam, dp, (T*)out, g, d, res);                 \
  }
  MRLA_DISPATCH_T_ACT(dtype, act, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_light_stats_bwd_nchw(const void* dout, const void* x, const void* o, const float* wv, float* bmom,
                                const SlabGeo& g, int dtype, int act, hipStream_t st) {
  const size_t es = dtype_size(dtype);
  const size_t lds = (size_t)g.astride * es * (o ? 3 : 2) + (size_t)g.NG * g.NB * g.PW * D_N * sizeof(float);
  const dim3 grid(g.slabs, (g.B + g.BG - 1) / g.BG);
#define CALL(T, A)                                                                                  \
  {                                                                                                 \
    if (set_lds(light_stats_bwd_nchw<T, A>, lds) != hipSuccess) return MRLA_EHIP;                     \
    hipLaunchKernelGGL((light_stats_bwd_nchw<T, A>), grid, dim3(kThreads), lds, st, (const T*)dout,   \
                       (const T*)x, (const T*)o, wv, bmom, g);                                      \
  }
  MRLA_DISPATCH_T_ACT(dtype, act, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_light_apply_bwd_nchw(const void* dout, const void* x, const void* o, const float* wv, const float* gate,
                                const float* cb, const float* lam, const float* dp, const float* dyx, void* dx,
                                void* dprev, float* dwv_part, const SlabGeo& g, int d, int res, int dtype, int act,
                                hipStream_t st) {
  const size_t es = dtype_size(dtype);
  const size_t lds = (size_t)g.astride * es * (o ? 5 : 3) + (size_t)g.NG * g.NB * g.PW * 9 * sizeof(float);
  const dim3 grid(g.slabs, (g.B + g.BG - 1) / g.BG);
  const int tpw = (g.NG * g.NB + kWaves - 1) / kWaves;
#define CALL_TPW(T, A, TPW)                                                                         \
  {                                                                                                 \
    if (set_lds(light_apply_bwd_nchw<T, A, TPW>, lds) != hipSuccess) return MRLA_EHIP;                \
    hipLaunchKernelGGL((light_apply_bwd_nchw<T, A, TPW>), grid, dim3(kThreads), lds, st,              \
                       (const T*)dout, (const T*)x, (const T*)o, wv, gate, cb, lam, dp, dyx, (T*)dx, \
                       (T*)dprev, dwv_part, g, d, res);                                             \
  }
#define CALL(T, A)                                                                                  \
  {                                                                                                 \
    if (tpw <= 1) CALL_TPW(T, A, 1)                                                                 \
    else if (tpw == 2) CALL_TPW(T, A, 2)                                                            \
    else if (tpw <= kMaxTasksPerWave) CALL_TPW(T, A, kMaxTasksPerWave)                              \
    else return MRLA_EUNSUPPORTED;                                                                  \
  }
  MRLA_DISPATCH_T_ACT(dtype, act, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

}  // namespace mrla
