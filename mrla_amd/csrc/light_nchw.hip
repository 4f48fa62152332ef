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
// Workgroup pipeline shared by the four kernels.  A workgroup walks over up to BG images of its slab;
// the input arrays of image i+1 are fetched by LDS-DMA into the other half of a double buffer while the
// waves march over image i, so HBM latency overlaps the stencil arithmetic:
//     wait(i) ; barrier ; prefetch(i+1) ; march(i) ; barrier ; store(i)
// Every per-plane / per-image parameter the march needs (taps, gate, BN constants, drop-path scale) is
// gathered ONCE per workgroup into small LDS tables, so the loop issues no vector-memory loads besides the
// DMA prefetch: a `s_waitcnt vmcnt` for a parameter would also wait for the whole prefetch (vmcnt is in-order).
// ------------------------------------------------------------------------------------------------
constexpr int kPT = 12;   // floats per plane row of the parameter table (9 taps + 3 kernel-specific)
constexpr int kIT = 4;    // floats per (image, plane) row of the per-image table

__device__ __forceinline__ void load_taps(float (&w)[9], const float* __restrict__ prow) {
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = prow[k];
}

#define MRLA_PIPELINE_PROLOGUE(NA_, SRC0, SRC1, SRC2)                                   \
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;            \
  const int c0 = blockIdx.x * g.CP;                                                     \
  const int np = min(g.CP, g.C - c0);                                                   \
  const int n = np * g.HW;                                                              \
  const int ntasks = g.NG * g.NB;                                                       \
  const int b0 = blockIdx.y * g.BG;                                                     \
  const int b_end = min(g.B, b0 + g.BG);                                                \
  const LaneMap lmap = make_lane_map(g, lane);                                          \
  const int lastrow = (g.H - 1) * g.W;                                                  \
  {                                                                                     \
    const size_t off0 = ((size_t)b0 * g.C + c0) * g.HW;                                 \
    slab_prefetch(buf, (SRC0) + off0, n, tid);                                          \
    if ((NA_) > 1) slab_prefetch(buf + g.astride, (SRC1) + off0, n, tid);               \
    if ((NA_) > 2) slab_prefetch(buf + 2 * g.astride, (SRC2) + off0, n, tid);           \
  }

#define MRLA_PIPELINE_NEXT(NA_, SRC0, SRC1, SRC2)                                       \
  wait_async_copies();                                                                  \
  __syncthreads();                                                                      \
  if (b + 1 < b_end) {                                                                  \
    const size_t off1 = ((size_t)(b + 1) * g.C + c0) * g.HW;                            \
    T* nb = buf + (cur ^ 1) * (NA_) * g.astride;                                        \
    slab_prefetch(nb, (SRC0) + off1, n, tid);                                           \
    if ((NA_) > 1) slab_prefetch(nb + g.astride, (SRC1) + off1, n, tid);                \
    if ((NA_) > 2) slab_prefetch(nb + 2 * g.astride, (SRC2) + off1, n, tid);            \
  }

// ------------------------------------------------------------------------------------------------
// forward statistics:  mom[b, c, 0..7] -- the sums over V and o are taken about per-plane PIVOTS (pV = V at the plane's
// first pixel, pO = o there: samples of the plane, so nothing cancels when |mean| >> sigma), exactly the record the NHWC
// row pipeline writes (mrla_device.h: M_REC); the per-channel kernels un-shift in double.
// ------------------------------------------------------------------------------------------------
// pV / pO of the slab's planes into ptab[p][9] / ptab[p][10] (the caller synchronises afterwards)
template <typename T, bool GELU, bool HAS_O>
__device__ __forceinline__ void plane_pivots(float* __restrict__ ptab, const T* __restrict__ xs, const T* __restrict__ os,
                                             int np, int H, int W, int tid) {
  for (int p = tid; p < np; p += kThreads) {
    const T* xp = xs + (size_t)p * H * W;
    const float* w = ptab + p * kPT;
    float v = w[4] * to_f(xp[0]);                            // V at pixel (0, 0): the taps that fall inside the plane
    if (W > 1) v = fmaf(w[5], to_f(xp[1]), v);
    if (H > 1) {
      v = fmaf(w[7], to_f(xp[W]), v);
      if (W > 1) v = fmaf(w[8], to_f(xp[W + 1]), v);
    }
    ptab[p * kPT + 9] = GELU ? gelu_f(v) : v;
    ptab[p * kPT + 10] = HAS_O ? to_f(os[(size_t)p * H * W]) : 0.f;
  }
}

// elementwise x = relu(pre + o) on a slab held in LDS, in place over `pre` (rounded to T exactly as the eager
// `out += identity; relu(out)` of resnet_mrla_light.py:113-114 would have materialised it)
template <typename T>
__device__ __forceinline__ void relu_add_inplace(T* __restrict__ xs, const T* __restrict__ os, int n, int tid) {
  constexpr int VEC = 16 / sizeof(T);
  typedef T VT __attribute__((ext_vector_type(VEC)));
  const int nv = n / VEC;
  VT* x4 = reinterpret_cast<VT*>(xs);
  const VT* o4 = reinterpret_cast<const VT*>(os);
  for (int i = tid; i < nv; i += kThreads) {
    VT a = x4[i];
    const VT b = o4[i];
#pragma unroll
    for (int k = 0; k < VEC; ++k) a[k] = from_f<T>(fmaxf(to_f(from_f<T>(to_f(a[k]) + to_f(b[k]))), 0.f));
    x4[i] = a;
  }
  for (int i = nv * VEC + tid; i < n; i += kThreads)
    xs[i] = from_f<T>(fmaxf(to_f(from_f<T>(to_f(xs[i]) + to_f(os[i]))), 0.f));
}

// x = relu(round(sc[p]*pre + sh[p]) + o) on an LDS slab of `np` planes (the bn3 affine handed over by the caller)
template <typename T>
__device__ __forceinline__ void relu_affine_add_inplace(T* xs, const T* os, int n, int hw, const float* __restrict__ sc,
                                                        const float* __restrict__ sh, int tid) {
  for (int i = tid; i < n; i += kThreads) {
    const int p = i / hw;
    const float z = to_f(from_f<T>(fmaf(sc[p], to_f(xs[i]), sh[p])));
    xs[i] = from_f<T>(fmaxf(to_f(from_f<T>(z + to_f(os[i]))), 0.f));
  }
}

// FUSE: `x` is the pre-activation (bn3 output); the kernel forms x_t = relu(pre + o) itself, writes it to `xout`
// (it is saved for backward and read by the apply pass) and takes the moments of that.
template <typename T, bool GELU, bool HAS_O, bool FUSE>
__global__ __launch_bounds__(kThreads) void light_stats_fwd_nchw(
    const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv,
    float* __restrict__ mom, T* __restrict__ xout, const float* __restrict__ psc, const float* __restrict__ psh,
    SlabGeo g) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int NA = HAS_O ? 2 : 1;
  T* buf = reinterpret_cast<T*>(smem);                                     // [2][NA][astride]
  float* ptab = reinterpret_cast<float*>(buf + 2 * NA * g.astride);        // [CP][kPT]
  float* red = ptab + g.CP * kPT;                                          // [ntasks][PW][M_N]
  MRLA_PIPELINE_PROLOGUE(NA, x, o, o)
  for (int i = tid; i < np * 9; i += kThreads) ptab[(i / 9) * kPT + i % 9] = wv[(size_t)c0 * 9 + i];
  int cur = 0;
  for (int b = b0; b < b_end; ++b, cur ^= 1) {
    MRLA_PIPELINE_NEXT(NA, x, o, o)
    T* xs = buf + cur * NA * g.astride;
    const T* os = xs + g.astride;
    if (FUSE) {
      if (psc) relu_affine_add_inplace(xs, os, n, g.HW, psc + c0, psh + c0, tid);
      else relu_add_inplace(xs, os, n, tid);
      __syncthreads();
      slab_store(xout + ((size_t)b * g.C + c0) * g.HW, xs, n, tid);
    }
    plane_pivots<T, GELU, HAS_O>(ptab, xs, os, np, g.H, g.W, tid);
    __syncthreads();
    for (int task = wave; task < ntasks; task += kWaves) {
      const LaneTask t = make_task(g, lmap, task, np);
      if (!t.live) continue;
      const T* xp = xs + t.p * g.HW + t.col;
      const T* op = os + t.p * g.HW + t.col;
      float w[9];
      load_taps(w, ptab + t.p * kPT);
      const float pV = ptab[t.p * kPT + 9], pO = ptab[t.p * kPT + 10];
      mask_conv(w, t);
      float acc[M_N] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int idx = t.r0 * g.W;
      Row3 ra = row_of(ld_centre_at(xp, idx - g.W, lastrow));
      Row3 rb = row_of(ld_centre_at(xp, idx, lastrow));
      float cn = ld_centre_at(xp, idx + g.W, lastrow);
      for (int r = t.r0; r < t.r1; ++r, idx += g.W) {
        const float cnn = ld_centre_at(xp, idx + 2 * g.W, lastrow);   // LDS read for the next iteration, issued early
        const Row3 rc = row_of(cn);
        float v = conv9(w, ra, rb, rc);
        if (GELU) v = gelu_f(v);
        v -= pV;
        acc[M_SX] += rb.c;
        acc[M_SV] += v;
        acc[M_SVV] = fmaf(v, v, acc[M_SVV]);
        if (HAS_O) {
          const float ov = to_f(op[idx]) - pO;
          acc[M_SO] += ov;
          acc[M_SVO] = fmaf(v, ov, acc[M_SVO]);
          acc[M_SOO] = fmaf(ov, ov, acc[M_SOO]);
        }
        ra = rb; rb = rc; cn = cnn;
      }
#pragma unroll
      for (int k = 0; k < M_N; ++k) {
        const float s = seg_sum(t.valid ? acc[k] : 0.f, lane, g.WS);
        if (t.last) red[(task * g.PW + t.pl) * M_N + k] = s;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < np * M_N; idx += kThreads) {
      const int p = idx / M_N, k = idx - p * M_N;
      const int grp = p / g.PW, pl = p - grp * g.PW;
      float s = 0.f;
      for (int band = 0; band < g.NB; ++band) s += red[((grp * g.NB + band) * g.PW + pl) * M_N + k];
      mom[((size_t)b * g.C + c0 + p) * M_REC + k] = s;
      if (k == 0) {
        mom[((size_t)b * g.C + c0 + p) * M_REC + M_PV] = ptab[p * kPT + 9];
        mom[((size_t)b * g.C + c0 + p) * M_REC + M_PO] = ptab[p * kPT + 10];
      }
    }
    // (the next image's pivots overwrite ptab[.][9..10] only after the barrier inside MRLA_PIPELINE_NEXT)
  }
}

// ------------------------------------------------------------------------------------------------
// forward apply:  out = res*x + A*V + B*o + C
// ------------------------------------------------------------------------------------------------
template <typename T, bool GELU, bool HAS_O>
__global__ __launch_bounds__(kThreads) void light_apply_fwd_nchw(
    const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv,
    const float* __restrict__ gate /*[b,g]*/, const float* __restrict__ sc, const float* __restrict__ sh,
    const float* __restrict__ lam, const float* __restrict__ dp, T* __restrict__ out, SlabGeo g, int d, int res) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int NA = HAS_O ? 2 : 1;
  T* buf = reinterpret_cast<T*>(smem);                       // [2][NA][astride] inputs
  float* ptab = reinterpret_cast<float*>(buf + 2 * NA * g.astride);  // [CP][kPT]: taps
  float* itab = ptab + g.CP * kPT;                           // [BG][CP][kIT]: A, B, C
  MRLA_PIPELINE_PROLOGUE(NA, x, o, o)
  const int G = g.C / d;
  const float resf = res ? 1.f : 0.f;
  for (int i = tid; i < np * 9; i += kThreads) ptab[(i / 9) * kPT + i % 9] = wv[(size_t)c0 * 9 + i];
  for (int i = tid; i < (b_end - b0) * np; i += kThreads) {
    const int bi = i / np, p = i - bi * np, c = c0 + p, b = b0 + bi;
    const float dpb = dp ? dp[b] : 1.f;
    const float scale = dpb * (sc ? sc[c] : 1.f);
    float* row = itab + ((size_t)bi * g.CP + p) * kIT;
    row[0] = scale * gate[(size_t)b * G + c / d];
    row[1] = HAS_O ? scale * lam[c] : 0.f;
    row[2] = sh ? dpb * sh[c] : 0.f;
  }
  int cur = 0;
  for (int b = b0; b < b_end; ++b, cur ^= 1) {
    MRLA_PIPELINE_NEXT(NA, x, o, o)
    const T* xs = buf + cur * NA * g.astride;
    const T* os = xs + g.astride;
    for (int task = wave; task < ntasks; task += kWaves) {
      const LaneTask t = make_task(g, lmap, task, np);
      if (!t.live) continue;
      const T* xp = xs + t.p * g.HW + t.col;
      const T* op = os + t.p * g.HW + t.col;
      // results leave straight from the marching lanes (one 2..4-byte store per lane and row, a contiguous run per
      // wave): no LDS staging, no second barrier; the L2 merges the row pieces into full lines
      T* yb = out + ((size_t)b * g.C + c0) * g.HW;            // wave-uniform base + 32-bit lane offset
      const int lo = t.p * g.HW + t.col;
      float w[9];
      load_taps(w, ptab + t.p * kPT);
      mask_conv(w, t);
      const float* irow = itab + ((size_t)(b - b0) * g.CP + t.p) * kIT;
      const float A = irow[0], Bc = irow[1], Cc = irow[2];
      if (!GELU) {                                            // fold gate / BN scale into the taps and the
#pragma unroll
        for (int k = 0; k < 9; ++k) w[k] *= A;                // residual into the centre tap (with an activation
        w[4] += resf;                                         // between conv and gate nothing can be folded)
      }
      if (t.valid) {                                          // lanes without a real column sit the march out
        int idx = t.r0 * g.W;
        Row3 ra = row_of(ld_centre_at(xp, idx - g.W, lastrow));
        Row3 rb = row_of(ld_centre_at(xp, idx, lastrow));
        float cn = ld_centre_at(xp, idx + g.W, lastrow);
        for (int r = t.r0; r < t.r1; ++r, idx += g.W) {
          const float cnn = ld_centre_at(xp, idx + 2 * g.W, lastrow);
          const Row3 rc = row_of(cn);
          float y;
          if (GELU) y = fmaf(A, gelu_f(conv9(w, ra, rb, rc)), fmaf(resf, rb.c, Cc));
          else      y = conv9(w, ra, rb, rc) + Cc;
          if (HAS_O) y = fmaf(Bc, to_f(op[idx]), y);
          yb[(unsigned)(lo + idx)] = from_f<T>(y);
          ra = rb; rb = rc; cn = cnn;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward statistics:  bmom[b, c, 0..2] = (sum dOut, sum dOut*V, sum dOut*o)
// ------------------------------------------------------------------------------------------------
template <typename T, bool GELU, bool HAS_O>
__global__ __launch_bounds__(kThreads) void light_stats_bwd_nchw(
    const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ o,
    const float* __restrict__ wv, const float* __restrict__ mom, float* __restrict__ bmom, SlabGeo g) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int NA = HAS_O ? 3 : 2;
  T* buf = reinterpret_cast<T*>(smem);                                     // [2][NA][astride]: x, dOut, o
  float* ptab = reinterpret_cast<float*>(buf + 2 * NA * g.astride);        // [CP][kPT]
  float* red = ptab + g.CP * kPT;
  MRLA_PIPELINE_PROLOGUE(NA, x, dout, o)
  for (int i = tid; i < np * 9; i += kThreads) ptab[(i / 9) * kPT + i % 9] = wv[(size_t)c0 * 9 + i];
  int cur = 0;
  for (int b = b0; b < b_end; ++b, cur ^= 1) {
    MRLA_PIPELINE_NEXT(NA, x, dout, o)
    const T* xs = buf + cur * NA * g.astride;
    const T* gs = xs + g.astride;
    const T* os = gs + g.astride;
    // sums about the pivots the forward statistics pass recorded for the plane (the per-channel kernels un-shift in double)
    for (int p = tid; p < np; p += kThreads) {
      const size_t rec = ((size_t)b * g.C + c0 + p) * M_REC;
      ptab[p * kPT + 9] = mom ? mom[rec + M_PV] : 0.f;
      ptab[p * kPT + 10] = (mom && HAS_O) ? mom[rec + M_PO] : 0.f;
    }
    __syncthreads();
    for (int task = wave; task < ntasks; task += kWaves) {
      const LaneTask t = make_task(g, lmap, task, np);
      if (!t.live) continue;
      const T* xp = xs + t.p * g.HW + t.col;
      const T* gp = gs + t.p * g.HW + t.col;
      const T* op = os + t.p * g.HW + t.col;
      float w[9];
      load_taps(w, ptab + t.p * kPT);
      const float pV = ptab[t.p * kPT + 9], pO = ptab[t.p * kPT + 10];
      mask_conv(w, t);
      float acc[D_N] = {0.f, 0.f, 0.f};
      int idx = t.r0 * g.W;
      Row3 ra = row_of(ld_centre_at(xp, idx - g.W, lastrow));
      Row3 rb = row_of(ld_centre_at(xp, idx, lastrow));
      float cn = ld_centre_at(xp, idx + g.W, lastrow);
      for (int r = t.r0; r < t.r1; ++r, idx += g.W) {
        const float cnn = ld_centre_at(xp, idx + 2 * g.W, lastrow);
        const Row3 rc = row_of(cn);
        float v = conv9(w, ra, rb, rc);
        if (GELU) v = gelu_f(v);
        const float gv = to_f(gp[idx]);
        acc[D_D] += gv;
        acc[D_DV] = fmaf(gv, v - pV, acc[D_DV]);
        if (HAS_O) acc[D_DO] = fmaf(gv, to_f(op[idx]) - pO, acc[D_DO]);
        ra = rb; rb = rc; cn = cnn;
      }
#pragma unroll
      for (int k = 0; k < D_N; ++k) {
        const float s = seg_sum(t.valid ? acc[k] : 0.f, lane, g.WS);
        if (t.last) red[(task * g.PW + t.pl) * D_N + k] = s;
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
  }
}

// ------------------------------------------------------------------------------------------------
// backward apply:  dx, do, and per-(image group, channel) partial sums of dwv
// ------------------------------------------------------------------------------------------------
// Per-task constants of the backward march.
struct BwdConsts {
  float w[9];      // conv taps, edge-masked       (U = conv(x))
  float wt[9];     // conv^T taps, edge-masked     (dx = conv^T(dU))
  float E, F, Gc, Hc;   // dm = E*dOut + F*V + Gc*o + Hc
  float am;        // a[b,g] (0 on lanes that hold no real column): dU = am*dm
  float lm;        // lambda: do = lm*dm
  float dy;        // dyx[b,c]
  float resf;      // residual flag
};

// One row step of the backward march at row `rr` (element offset idx = rr*W):
//   C    <- x[rr+1] row (centre was preloaded in xn; xn <- centre of x[rr+2])
//   UC   <- dU[rr]   (zero when !INSIDE; taken from dOut/o/x rows at rr)
//   OWNED: row rr belongs to this band -> do[rr] (or its deferral), wgrad accumulation
//   EMIT : dx[rr-1] = res*dOut[rr-1] + conv^T(dU[rr-2..rr]) + dyx   (+ ReLU mask / identity gradient when RELU)
// Register roles rotate in the caller (A,B,C / UA,UB,UC), so no window copies are needed.
template <typename T, bool GELU, bool HAS_O, bool RELU, bool INSIDE, bool OWNED, bool EMIT>
__device__ __forceinline__ void bwd_row(const BwdConsts& k, const T* __restrict__ xp, const T* __restrict__ gp,
                                        const T* __restrict__ op, T* __restrict__ dxb, T* __restrict__ dob, int lo,
                                        int idx, int W, int lastrow, const Row3& A, const Row3& B, Row3& C,
                                        float& xn, const Row3& UA, const Row3& UB, Row3& UC, float& gprev,
                                        float& dohold, float (&wg)[9]) {
  C = row_of(xn);
  xn = ld_centre_at(xp, idx + 2 * W, lastrow);
  float gcur = 0.f, du = 0.f, dmo = 0.f;
  if (INSIDE) {
    gcur = to_f(gp[idx]);
    const float u = conv9(k.w, A, B, C);
    const float v = GELU ? gelu_f(u) : u;
    float dm = fmaf(k.E, gcur, k.Hc);
    dm = fmaf(k.F, v, dm);
    if (HAS_O) dm = fmaf(k.Gc, to_f(op[idx]), dm);
    du = k.am * dm;
    if (GELU) du *= gelu_grad_f(u);
    dmo = k.lm * dm;
  }
  if (OWNED) {
    if (HAS_O && !RELU) dob[(unsigned)(lo + idx)] = from_f<T>(dmo);
    wg[0] = fmaf(du, A.l, wg[0]); wg[1] = fmaf(du, A.c, wg[1]); wg[2] = fmaf(du, A.r, wg[2]);
    wg[3] = fmaf(du, B.l, wg[3]); wg[4] = fmaf(du, B.c, wg[4]); wg[5] = fmaf(du, B.r, wg[5]);
    wg[6] = fmaf(du, C.l, wg[6]); wg[7] = fmaf(du, C.c, wg[7]); wg[8] = fmaf(du, C.r, wg[8]);
  }
  UC = row_of(du);
  if (EMIT) {
    // dx[ro][w] = sum_{i,j} wv[i][j] * dU[ro-i+1][w-j+1],  ro = rr-1
    float s9 = k.wt[0] * UC.r;
    s9 = fmaf(k.wt[1], UC.c, s9); s9 = fmaf(k.wt[2], UC.l, s9);
    s9 = fmaf(k.wt[3], UB.r, s9); s9 = fmaf(k.wt[4], UB.c, s9); s9 = fmaf(k.wt[5], UB.l, s9);
    s9 = fmaf(k.wt[6], UA.r, s9); s9 = fmaf(k.wt[7], UA.c, s9); s9 = fmaf(k.wt[8], UA.l, s9);
    float y = fmaf(k.resf, gprev, s9 + k.dy);
    if (RELU) {
      y = (A.c > 0.f) ? y : 0.f;                              // A = x[rr-1] = x[ro]
      if (HAS_O) dob[(unsigned)(lo + idx - W)] = from_f<T>(dohold + y);
    }
    dxb[(unsigned)(lo + idx - W)] = from_f<T>(y);
  }
  gprev = gcur;
  dohold = dmo;
}

// RELU (fused producer): x_t = relu(pre + o_prev) was formed by the forward statistics kernel, so the gradient that
// leaves here is dpre = [x > 0] * dx and the identity receives lam*dm + dpre (resnet_mrla_light.py:113-116 backward).
template <typename T, bool GELU, bool HAS_O, bool RELU>
__global__ __launch_bounds__(kThreads) void light_apply_bwd_nchw(
    const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ o, const float* __restrict__ wv,
    const float* __restrict__ gate /*[b,g]*/, const float* __restrict__ cb /*[c,4]: e,f,G,H or null*/,
    const float* __restrict__ lam, const float* __restrict__ dp, const float* __restrict__ dyx /*[b,c]*/,
    T* __restrict__ dx, T* __restrict__ dprev, float* __restrict__ dwv_part /*[gridDim.y, c, 9]*/,
    SlabGeo g, int d, int res) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int NA = HAS_O ? 3 : 2;
  T* buf = reinterpret_cast<T*>(smem);                        // [2][NA][astride]: x, dOut, o
  float* ptab = reinterpret_cast<float*>(buf + 2 * NA * g.astride);        // [CP][kPT]: taps, G, H, lambda
  float* itab = ptab + g.CP * kPT;                            // [BG][CP][kIT]: E, F, a, dy
  float* red = itab + g.BG * g.CP * kIT;                      // [ntasks][PW][9] wgrad sums over the images
  MRLA_PIPELINE_PROLOGUE(NA, x, dout, o)
  {
    const int G = g.C / d;
    for (int i = tid; i < np * kPT; i += kThreads) {
      const int p = i / kPT, k = i - p * kPT, c = c0 + p;
      float v;
      if (k < 9) v = wv[(size_t)c * 9 + k];
      else if (k == 9) v = cb ? cb[c * 4 + 2] : 0.f;
      else if (k == 10) v = cb ? cb[c * 4 + 3] : 0.f;
      else v = (HAS_O && lam) ? lam[c] : 1.f;
      ptab[i] = v;
    }
    for (int i = tid; i < (b_end - b0) * np; i += kThreads) {
      const int bi = i / np, p = i - bi * np, c = c0 + p, b = b0 + bi;
      const float dpb = dp ? dp[b] : 1.f;
      const float a = gate[(size_t)b * G + c / d];
      float* row = itab + ((size_t)bi * g.CP + p) * kIT;
      row[0] = cb ? cb[c * 4 + 0] * dpb : dpb;
      row[1] = cb ? cb[c * 4 + 1] * a : 0.f;
      row[2] = a;
      row[3] = dyx[(size_t)b * g.C + c];
    }
    for (int i = tid; i < ntasks * g.PW * 9; i += kThreads) red[i] = 0.f;
  }
  const int W = g.W;
  int cur = 0;
  for (int b = b0; b < b_end; ++b, cur ^= 1) {
    MRLA_PIPELINE_NEXT(NA, x, dout, o)
    const T* xs = buf + cur * NA * g.astride;
    const T* gs = xs + g.astride;
    const T* os = gs + g.astride;
    for (int task = wave; task < ntasks; task += kWaves) {
      const LaneTask t = make_task(g, lmap, task, np);
      if (!t.live) continue;
      const T* xp = xs + t.p * g.HW + t.col;
      const T* gp = gs + t.p * g.HW + t.col;
      const T* op = os + t.p * g.HW + t.col;
      // results leave straight from the lanes: wave-uniform base + 32-bit lane offset
      T* dxb = dx + ((size_t)b * g.C + c0) * g.HW;
      T* dob = HAS_O ? dprev + ((size_t)b * g.C + c0) * g.HW : nullptr;
      const int lo = t.p * g.HW + t.col;
      BwdConsts k;
      const float* prow = ptab + t.p * kPT;
      load_taps(k.w, prow);
#pragma unroll
      for (int i = 0; i < 9; ++i) k.wt[i] = k.w[i];
      mask_conv(k.w, t);
      mask_convT(k.wt, t);
      k.Gc = prow[9]; k.Hc = prow[10]; k.lm = prow[11];
      const float* irow = itab + ((size_t)(b - b0) * g.CP + t.p) * kIT;
      k.E = irow[0]; k.F = irow[1]; k.am = irow[2]; k.dy = irow[3];
      k.resf = res ? 1.f : 0.f;
      float wg[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const bool valid = t.valid;
      // Lanes that hold no real column sit out the whole march (EXEC off): DPP reads from them return 0, which is what
      // the edge-masked taps assume anyway, and the stores need no per-row predicate.  Trip counts are wave-uniform.
      if (valid) {
      // window state before the step at row rr:  xa = x[rr-1], xb = x[rr], xn = centre of x[rr+1],
      //                                          ua = dU[rr-2], ub = dU[rr-1]
      int rr = t.r0 - 1;
      int idx = rr * W;
      Row3 xa = row_of(ld_centre_at(xp, idx - W, lastrow));
      Row3 xb = row_of(ld_centre_at(xp, idx, lastrow));
      Row3 xc;
      float xn = ld_centre_at(xp, idx + W, lastrow);
      Row3 ua = {0.f, 0.f, 0.f}, ub = {0.f, 0.f, 0.f}, uc;
      float gprev = 0.f, dohold = 0.f;
#define MRLA_BWD_STEP(INS, OWN, EMI, A, B, C, UA, UB, UC)                                                            \
  bwd_row<T, GELU, HAS_O, RELU, INS, OWN, EMI>(k, xp, gp, op, dxb, dob, lo, idx, W, lastrow, A, B, C, xn, UA, UB, UC, \
                                               gprev, dohold, wg);                                                   \
  ++rr; idx += W;
      // halo row above the band: dU[r0-1] only (zero above the plane)
      if (rr >= 0) { MRLA_BWD_STEP(true, false, false, xa, xb, xc, ua, ub, uc) }
      else         { MRLA_BWD_STEP(false, false, false, xa, xb, xc, ua, ub, uc) }
      // first owned row: nothing to emit yet        (roles rotated once: x = (xb, xc, xa), u = (ub, uc, ua))
      MRLA_BWD_STEP(true, true, false, xb, xc, xa, ub, uc, ua)
      // steady state, three rows per trip so the register roles return to (xc, xa, xb) / (uc, ua, ub)
      while (rr + 2 < t.r1) {
        MRLA_BWD_STEP(true, true, true, xc, xa, xb, uc, ua, ub)
        MRLA_BWD_STEP(true, true, true, xa, xb, xc, ua, ub, uc)
        MRLA_BWD_STEP(true, true, true, xb, xc, xa, ub, uc, ua)
      }
      const bool below = t.r1 < g.H;                          // is the halo row under the band inside the plane?
      if (rr < t.r1) {
        MRLA_BWD_STEP(true, true, true, xc, xa, xb, uc, ua, ub)
        if (rr < t.r1) {
          MRLA_BWD_STEP(true, true, true, xa, xb, xc, ua, ub, uc)
          if (below) { MRLA_BWD_STEP(true, false, true, xb, xc, xa, ub, uc, ua) }
          else       { MRLA_BWD_STEP(false, false, true, xb, xc, xa, ub, uc, ua) }
        } else {
          if (below) { MRLA_BWD_STEP(true, false, true, xa, xb, xc, ua, ub, uc) }
          else       { MRLA_BWD_STEP(false, false, true, xa, xb, xc, ua, ub, uc) }
        }
      } else {
        if (below) { MRLA_BWD_STEP(true, false, true, xc, xa, xb, uc, ua, ub) }
        else       { MRLA_BWD_STEP(false, false, true, xc, xa, xb, uc, ua, ub) }
      }
#undef MRLA_BWD_STEP
      }
      // fold this image's wgrad contribution into the per-(task, plane) sums
      // (edge columns: the left/right x neighbours of an edge lane belong to another plane -> masked here)
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const float m = (i % 3 == 0) ? t.lmask : ((i % 3 == 2) ? t.rmask : 1.f);
        const float v = seg_sum(valid ? wg[i] * m : 0.f, lane, g.WS);
        if (t.last) red[(task * g.PW + t.pl) * 9 + i] += v;
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
constexpr size_t kMaxLds = 150 * 1024;

template <typename K>
static hipError_t set_lds(K kernel, size_t bytes) {
  return lds_opt_in(reinterpret_cast<const void*>(kernel), bytes);
}

#define MRLA_DISPATCH_AO(TT, ACT, HASO, CALL)                          \
  if (ACT) { if (HASO) { CALL(TT, true, true); } else { CALL(TT, true, false); } } \
  else     { if (HASO) { CALL(TT, false, true); } else { CALL(TT, false, false); } }
#define MRLA_DISPATCH_T_ACT(DT, ACT, HASO, CALL)                       \
  switch (DT) {                                                        \
    case MRLA_F32:  MRLA_DISPATCH_AO(float, ACT, HASO, CALL) break;    \
    case MRLA_BF16: MRLA_DISPATCH_AO(bf16_t, ACT, HASO, CALL) break;   \
    case MRLA_F16:  MRLA_DISPATCH_AO(f16_t, ACT, HASO, CALL) break;    \
    default: return MRLA_EINVAL;                                       \
  }

int launch_light_stats_fwd_nchw(const void* x, const void* o, const float* wv, float* mom, void* xout,
                                const float* psc, const float* psh, const SlabGeo& g, int dtype, int act,
                                hipStream_t st) {
  const size_t es = dtype_size(dtype);
  const size_t lds = (size_t)g.astride * es * 2 * (o ? 2 : 1) +
                     ((size_t)g.CP * kPT + (size_t)g.NG * g.NB * g.PW * M_N) * sizeof(float);
  if (lds > kMaxLds) return MRLA_EUNSUPPORTED;
  const dim3 grid(g.slabs, (g.B + g.BG - 1) / g.BG);
#define CALL_F(T, A, O, F)                                                                          \
  {                                                                                                 \
    if (set_lds(light_stats_fwd_nchw<T, A, O, F>, lds) != hipSuccess) return MRLA_EHIP;               \
    hipLaunchKernelGGL((light_stats_fwd_nchw<T, A, O, F>), grid, dim3(kThreads), lds, st,             \
                       (const T*)x, (const T*)o, wv, mom, (T*)xout, psc, psh, g);                   \
  }
#define CALL(T, A, O)                                                                               \
  {                                                                                                 \
    if (xout) { if (O) CALL_F(T, A, true, true) else return MRLA_EINVAL; }                          \
    else CALL_F(T, A, O, false)                                                                     \
  }
  MRLA_DISPATCH_T_ACT(dtype, act, o != nullptr, CALL)
#undef CALL
#undef CALL_F
  return hip_status(hipGetLastError());
}

int launch_light_apply_fwd_nchw(const void* x, const void* o, const float* wv, const float* gate, const float* sc,
                                const float* sh, const float* lam, const float* dp, void* out, const SlabGeo& g,
                                int d, int res, int dtype, int act, hipStream_t st) {
  const size_t es = dtype_size(dtype);
  const size_t lds = (size_t)g.astride * es * 2 * (o ? 2 : 1) + (size_t)g.CP * (kPT + g.BG * kIT) * sizeof(float);
  if (lds > kMaxLds) return MRLA_EUNSUPPORTED;
  const dim3 grid(g.slabs, (g.B + g.BG - 1) / g.BG);
#define CALL(T, A, O)                                                                               \
  {                                                                                                 \
    if (set_lds(light_apply_fwd_nchw<T, A, O>, lds) != hipSuccess) return MRLA_EHIP;                  \
    hipLaunchKernelGGL((light_apply_fwd_nchw<T, A, O>), grid, dim3(kThreads), lds, st, (const T*)x,   \
                       (const T*)o, wv, gate, sc, sh, lam, dp, (T*)out, g, d, res);                 \
  }
  MRLA_DISPATCH_T_ACT(dtype, act, o != nullptr, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_light_stats_bwd_nchw(const void* dout, const void* x, const void* o, const float* wv, const float* mom,
                                float* bmom, const SlabGeo& g, int dtype, int act, hipStream_t st) {
  const size_t es = dtype_size(dtype);
  const size_t lds = (size_t)g.astride * es * 2 * (o ? 3 : 2) +
                     ((size_t)g.CP * kPT + (size_t)g.NG * g.NB * g.PW * D_N) * sizeof(float);
  if (lds > kMaxLds) return MRLA_EUNSUPPORTED;
  const dim3 grid(g.slabs, (g.B + g.BG - 1) / g.BG);
#define CALL(T, A, O)                                                                               \
  {                                                                                                 \
    if (set_lds(light_stats_bwd_nchw<T, A, O>, lds) != hipSuccess) return MRLA_EHIP;                  \
    hipLaunchKernelGGL((light_stats_bwd_nchw<T, A, O>), grid, dim3(kThreads), lds, st, (const T*)dout, \
                       (const T*)x, (const T*)o, wv, mom, bmom, g);                                 \
  }
  MRLA_DISPATCH_T_ACT(dtype, act, o != nullptr, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_light_apply_bwd_nchw(const void* dout, const void* x, const void* o, const float* wv, const float* gate,
                                const float* cb, const float* lam, const float* dp, const float* dyx, void* dx,
                                void* dprev, float* dwv_part, const SlabGeo& g, int d, int res, int relu, int dtype,
                                int act, hipStream_t st) {
  const size_t es = dtype_size(dtype);
  const size_t lds = (size_t)g.astride * es * 2 * (o ? 3 : 2) +
                     ((size_t)g.CP * (kPT + g.BG * kIT) + (size_t)g.NG * g.NB * g.PW * 9) * sizeof(float);
  if (lds > kMaxLds) return MRLA_EUNSUPPORTED;
  const dim3 grid(g.slabs, (g.B + g.BG - 1) / g.BG);
#define CALL_TR(T, A, O, R)                                                                         \
  {                                                                                                 \
    if (set_lds(light_apply_bwd_nchw<T, A, O, R>, lds) != hipSuccess) return MRLA_EHIP;               \
    hipLaunchKernelGGL((light_apply_bwd_nchw<T, A, O, R>), grid, dim3(kThreads), lds, st,             \
                       (const T*)dout, (const T*)x, (const T*)o, wv, gate, cb, lam, dp, dyx, (T*)dx, \
                       (T*)dprev, dwv_part, g, d, res);                                             \
  }
#define CALL(T, A, O)                                                                               \
  {                                                                                                 \
    if (relu) { if (O && !(A)) CALL_TR(T, false, true, true) else return MRLA_EINVAL; }             \
    else CALL_TR(T, A, O, false)                                                                    \
  }
  MRLA_DISPATCH_T_ACT(dtype, act, o != nullptr, CALL)
#undef CALL
#undef CALL_TR
  return hip_status(hipGetLastError());
}

}  // namespace mrla
