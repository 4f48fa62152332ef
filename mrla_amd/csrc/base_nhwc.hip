// MRLA-base (softmax over the depth of a stage) for channels_last (NHWC) activations.
//
// Rings are SLOT-MAJOR here: v_ring / da_ring [T][b, h, w, c] (one slot = one ordinary NHWC activation tensor), so
//   * the value pass (light_stats_fwd_nhwc with `vout`) writes V_t = dwconv3x3(x_t) like any other NHWC output while it
//     pools x_t, and
//   * everything that walks the history is a FLAT streaming kernel: a workgroup owns a tile of R whole pixels x C
//     channels (R*C/VEC <= 2048 16-byte vectors, R divides h*w), thread k owns vectors k, k+256, ... of the tile -- all
//     of the same 8 channels, because 256 is a multiple of C/VEC -- keeps its slice of the tile in registers and streams
//     the history slots past it, 8 independent 16-byte loads in flight per lane.
// Arithmetic, rounding points and summation order over the history follow base_nchw.hip (reference:
// resnet/models/modules/mrla_base_module.py:54-89), so the two layouts agree to the last bit on everything except the
// order in which per-plane sums are reduced.
//
//   forward : pool+value (light_nhwc.hip) -> gate (base_nchw.hip) -> base_combine<0> (attn, moments) -> tail
//   backward: tail statistics (bnact_nhwc.hip moments with dp) -> base_attend_bwd (dA -> ring, <dA, V_j> partials)
//             -> pmom reduce -> gate bwd -> base_combine<1> (dV_t) -> base_value_bwd (transposed 3x3, dWv)
#include <algorithm>

#include "light_nhwc.h"

namespace mrla {

constexpr int kNV = 8;                          // 16-byte vectors per thread and tile
constexpr int kNVc = 4;                         // ... of the kernels that stream the history past an accumulator tile:
                                                // smaller tiles, more workgroups per CU, a shorter last round

template <typename T> struct Vec16 { typedef T type __attribute__((ext_vector_type(16 / sizeof(T)))); };

template <typename T>
__device__ __forceinline__ void ldv(const T* __restrict__ p, float (&v)[16 / sizeof(T)]) {
  typedef typename Vec16<T>::type VT;
  const VT t = *reinterpret_cast<const VT*>(p);
#pragma unroll
  for (int i = 0; i < (int)(16 / sizeof(T)); ++i) v[i] = to_f(static_cast<T>(t[i]));
}
template <typename T>
__device__ __forceinline__ void stv(T* __restrict__ p, const float (&v)[16 / sizeof(T)]) {
  typedef typename Vec16<T>::type VT;
  VT t;
#pragma unroll
  for (int i = 0; i < (int)(16 / sizeof(T)); ++i) t[i] = from_f<T>(v[i]);
  *reinterpret_cast<VT*>(p) = t;
}

// 16 bytes as they come from memory; unpacked to floats only where they are consumed (halves the live registers of a
// tile slice that waits for its use)
template <typename T>
__device__ __forceinline__ u32x4 ldraw(const T* __restrict__ p) { return *reinterpret_cast<const u32x4*>(p); }
// the same with the streaming (nt) cache policy: history slots, far larger than the caches and read once per kernel
template <typename T>
__device__ __forceinline__ u32x4 ldraw_stream(const T* __restrict__ p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}
template <typename T>
__device__ __forceinline__ void unpack(const u32x4& r, float (&v)[16 / sizeof(T)]) {
  typedef typename Vec16<T>::type VT;
  const VT t = __builtin_bit_cast(VT, r);
#pragma unroll
  for (int i = 0; i < (int)(16 / sizeof(T)); ++i) v[i] = to_f(static_cast<T>(t[i]));
}

// Tile geometry shared by the flat kernels.
struct FlatGeo {
  int B, C, HW, R, tiles;       // R pixels per tile, tiles per image
  int nvec;                     // 16-byte vectors per tile (R * C / VEC)
  // The layer's EXTERNAL tensor (attend forward: out; attend backward: dOut) may be the map rows of a token tensor
  // [b, 1 + HW, C] (deit/deit_mrla_base.py:236-241) instead of a dense image: elements to skip per image and up front.
  int ext_gap, ext_off;
};

// per-thread coefficient of history slot `row/col` for its VEC channels (one value when VEC | d and aligned)
template <int VEC>
__device__ __forceinline__ void load_coef(const float* __restrict__ Pall, int b, int G, int d, int T_, int row, int col,
                                          int c0, float (&cf)[VEC]) {
  if (d % VEC == 0) {
    const float v = Pall[(((size_t)b * G + c0 / d) * T_ + row) * T_ + col];
#pragma unroll
    for (int i = 0; i < VEC; ++i) cf[i] = v;
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) cf[i] = Pall[(((size_t)b * G + (c0 + i) / d) * T_ + row) * T_ + col];
  }
}

// Sum per-thread channel partials over the threads of the workgroup that own the same channels (thread k owns channel
// vector k % (C/VEC)); `out[ch]` for ch < C is written by the first C threads' worth of work.  red: [kThreads * VEC].
// The workgroup size is the largest multiple of C/VEC up to 256 (240 threads for the 192 channels of DeiT-tiny in fp32).
template <int VEC, typename F>
__device__ __forceinline__ void reduce_same_channels(const float (&s)[VEC], float* __restrict__ red, int C, F&& emit) {
  const int t = threadIdx.x, nt = blockDim.x;
#pragma unroll
  for (int i = 0; i < VEC; ++i) red[t * VEC + i] = s[i];
  __syncthreads();
  const int reps = (nt * VEC) / C;
  for (int ch = t; ch < C; ch += nt) {
    float sum = 0.f;
    for (int r = 0; r < reps; ++r) sum += red[r * C + ch];
    emit(ch, sum);
  }
}

// ------------------------------------------------------------------------------------------------
// MODE 0 (attend forward):  out = sum_{j<t} P[b,g,t-1,j] * V_j   (slot t-1 first, then j = 0..t-2, as base_nchw.hip),
//                           out rounded to T; amom_part[b*tiles + tile, c, 2] = (sum out, sum out^2) of the tile.
// MODE 1 (value gradient):  out = sum_{l=0..Tc-t} P[b,g,t-1+l,t-1] * dA_{t+l}   stored in T (as autograd stores dV).
// grid: (tiles, B)
// ------------------------------------------------------------------------------------------------
template <typename T, typename OUT, int MODE, int NV>
__global__ __launch_bounds__(kThreads) void base_combine_nhwc(const T* __restrict__ ring, const float* __restrict__ Pall,
                                                              OUT* __restrict__ out, float* __restrict__ amom_part,
                                                              FlatGeo g, int d, int T_, int t, int Tc) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[2][kThreads * VEC];
  const int tid = threadIdx.x, b = blockIdx.y, tile = blockIdx.x, NT = blockDim.x;
  const int G = g.C / d;
  const int c0 = (tid * VEC) % g.C;
  const size_t slot = (size_t)g.B * g.HW * g.C;
  const size_t base = ((size_t)b * g.HW + (size_t)tile * g.R) * g.C + (size_t)tid * VEC;
  const size_t obase = base + (size_t)b * g.ext_gap + g.ext_off;
  float acc[NV][VEC];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[i][k] = 0.f;
  const int n = MODE == 0 ? t : Tc - t + 1;
  // MODE 0 visits slot t-1 first and then 0 .. t-2; MODE 1 visits slots t-1 .. Tc-1
  auto slot_of = [&](int s) { return MODE == 0 ? (s == 0 ? t - 1 : s - 1) : t - 1 + s; };
  u32x4 raw[NV];
  auto issue = [&](int s) {
    const T* src = ring + (size_t)slot_of(s) * slot + base;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      raw[i] = (u32x4){0u, 0u, 0u, 0u};
      if (tid + i * NT < g.nvec) raw[i] = ldraw_stream<T>(src + (size_t)i * NT * VEC);
    }
  };
  issue(0);
  for (int s = 0; s < n; ++s) {
    const int j = slot_of(s);
    float cf[VEC];
    if (MODE == 0) load_coef<VEC>(Pall, b, G, d, T_, t - 1, j, c0, cf);
    else           load_coef<VEC>(Pall, b, G, d, T_, j, t - 1, c0, cf);
    u32x4 cur[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) cur[i] = raw[i];
    if (s + 1 < n) issue(s + 1);                 // the next slot's loads are in flight while this one is folded in
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float v[VEC];
      unpack<T>(cur[i], v);
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[i][k] = fmaf(cf[k], v[k], acc[i][k]);
    }
  }
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (tid + i * NT < g.nvec) {
      if constexpr (MODE == 0) {
        stv<T>(reinterpret_cast<T*>(out) + obase + (size_t)i * NT * VEC, acc[i]);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const float a = to_f(from_f<T>(acc[i][k]));       // statistics of the rounded values the tail reads back
          s1[k] += a;
          s2[k] = fmaf(a, a, s2[k]);
        }
      } else {
        stv<T>(reinterpret_cast<T*>(out) + obase + (size_t)i * NT * VEC, acc[i]);
      }
    }
  }
  if constexpr (MODE == 0) {
    float* dst = amom_part + ((size_t)b * g.tiles + tile) * g.C * 2;
    reduce_same_channels<VEC>(s1, red[0], g.C, [&](int ch, float v) { dst[ch * 2 + 0] = v; });
    reduce_same_channels<VEC>(s2, red[1], g.C, [&](int ch, float v) { dst[ch * 2 + 1] = v; });
  }
}

// ------------------------------------------------------------------------------------------------
// tail forward: out = x + dp[b] * relu(sc[c]*attn + sh[c])     grid: (tiles, B)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void base_tail_fwd_nhwc(const T* __restrict__ x, const T* __restrict__ attn,
                                                               const float* __restrict__ sc, const float* __restrict__ sh,
                                                               const float* __restrict__ dp, T* __restrict__ out,
                                                               FlatGeo g) {
  constexpr int VEC = 16 / sizeof(T);
  const int tid = threadIdx.x, b = blockIdx.y, NT = blockDim.x;
  const int c0 = (tid * VEC) % g.C;
  const size_t base = ((size_t)b * g.HW + (size_t)blockIdx.x * g.R) * g.C + (size_t)tid * VEC;
  const float dpb = dp ? dp[b] : 1.f;
  float s[VEC], h[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) { s[k] = sc[c0 + k]; h[k] = sh[c0 + k]; }
  float xv[kNV][VEC], av[kNV][VEC];
#pragma unroll
  for (int i = 0; i < kNV; ++i)
    if (tid + i * NT < g.nvec) {
      ldv<T>(x + base + (size_t)i * NT * VEC, xv[i]);
      ldv<T>(attn + base + (size_t)i * NT * VEC, av[i]);
    }
#pragma unroll
  for (int i = 0; i < kNV; ++i)
    if (tid + i * NT < g.nvec) {
      float y[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) y[k] = fmaf(dpb, fmaxf(fmaf(s[k], av[i][k], h[k]), 0.f), xv[i][k]);
      stv<T>(out + base + (size_t)i * NT * VEC, y);
    }
}

// ------------------------------------------------------------------------------------------------
// attend backward: dA_t = e*dz + f*attn + h (dz = dp*dOut*[sc*attn+sh > 0]; sc == null: dA_t = dOut) -> da_ring slot
// t-1 (rounded to T);  pmom_part[b*tiles + tile, j, c] = sum over the tile of dA_t * V_j, j < t.     grid: (tiles, B)
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ u32x4 pack16(const float (&v)[16 / sizeof(T)]) {
  typedef typename Vec16<T>::type VT;
  VT t;
#pragma unroll
  for (int i = 0; i < (int)(16 / sizeof(T)); ++i) t[i] = from_f<T>(v[i]);
  return __builtin_bit_cast(u32x4, t);
}

// OCC: waves per SIMD the register allocation aims at.  3 (<= 168 VGPRs) is faster while the history is short (the
// dOut / attn / dA phase dominates); with a long history the unconstrained allocation streams the slots faster.
template <typename T, int OCC>
__global__ __launch_bounds__(kThreads, OCC) void base_attend_bwd_nhwc(
    const T* __restrict__ dout, const T* __restrict__ attn, const float* __restrict__ sc, const float* __restrict__ sh,
    const float* __restrict__ dp, const float* __restrict__ cb /*[c,3]*/, const T* __restrict__ Vring,
    T* __restrict__ dAring, float* __restrict__ pmom_part, FlatGeo g, int T_, int t) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[2][kThreads * VEC];
  const int tid = threadIdx.x, b = blockIdx.y, tile = blockIdx.x, NT = blockDim.x;
  const int c0 = (tid * VEC) % g.C;
  const size_t slot = (size_t)g.B * g.HW * g.C;
  const size_t base = ((size_t)b * g.HW + (size_t)tile * g.R) * g.C + (size_t)tid * VEC;
  const size_t gbase = base + (size_t)b * g.ext_gap + g.ext_off;
  u32x4 da[kNV];                                 // this thread's slice of dA_t, rounded to T, packed
  u32x4 raw[kNV];
  auto issue = [&](int j) {
    const T* src = Vring + (size_t)j * slot + base;
#pragma unroll
    for (int i = 0; i < kNV; ++i) {
      raw[i] = (u32x4){0u, 0u, 0u, 0u};
      if (tid + i * NT < g.nvec) raw[i] = ldraw_stream<T>(src + (size_t)i * NT * VEC);
    }
  };
  {
    const float dpb = dp ? dp[b] : 1.f;
    float s[VEC], h[VEC], e_[VEC], f_[VEC], h_[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      s[k] = sc ? sc[c0 + k] : 0.f; h[k] = sc ? sh[c0 + k] : 1.f;
      e_[k] = sc ? cb[(c0 + k) * 3 + 0] : 1.f; f_[k] = sc ? cb[(c0 + k) * 3 + 1] : 0.f;
      h_[k] = sc ? cb[(c0 + k) * 3 + 2] : 0.f;
    }
    // two halves of kNV/2 vectors: enough loads in flight, half the registers
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      constexpr int HV = kNV / 2;
      u32x4 graw[HV], araw[HV];
#pragma unroll
      for (int k = 0; k < HV; ++k) {
        const int i = half * HV + k;
        graw[k] = (u32x4){0u, 0u, 0u, 0u};
        araw[k] = (u32x4){0u, 0u, 0u, 0u};
        if (tid + i * NT < g.nvec) {
          graw[k] = ldraw<T>(dout + gbase + (size_t)i * NT * VEC);
          if (attn) araw[k] = ldraw<T>(attn + base + (size_t)i * NT * VEC);
        }
      }
#pragma unroll
      for (int k = 0; k < HV; ++k) {
        const int i = half * HV + k;
        float gv[VEC], av[VEC], r[VEC];
        unpack<T>(graw[k], gv);
        unpack<T>(araw[k], av);
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          const float dz = (fmaf(s[q], av[q], h[q]) > 0.f) ? dpb * gv[q] : 0.f;
          r[q] = fmaf(e_[q], dz, fmaf(f_[q], av[q], h_[q]));
        }
        da[i] = pack16<T>(r);
        if (tid + i * NT < g.nvec)
          *reinterpret_cast<u32x4*>(dAring + (size_t)(t - 1) * slot + base + (size_t)i * NT * VEC) = da[i];
        else
          da[i] = (u32x4){0u, 0u, 0u, 0u};
      }
    }
  }
  issue(0);
  for (int j = 0; j < t; ++j) {
    u32x4 cur[kNV];
#pragma unroll
    for (int i = 0; i < kNV; ++i) cur[i] = raw[i];
    if (j + 1 < t) issue(j + 1);
    float s1[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) s1[k] = 0.f;
#pragma unroll
    for (int i = 0; i < kNV; ++i) {
      float a[VEC], v[VEC];
      unpack<T>(da[i], a);
      unpack<T>(cur[i], v);
#pragma unroll
      for (int k = 0; k < VEC; ++k) s1[k] = fmaf(a[k], v[k], s1[k]);
    }
    float* dst = pmom_part + (((size_t)b * g.tiles + tile) * t + j) * g.C;
    // two staging buffers: the reads of round j overlap the writes of round j+1 without a second barrier
    reduce_same_channels<VEC>(s1, red[j & 1], g.C, [&](int ch, float v) { dst[ch] = v; });
  }
}

// pmom[b, c, j] = sum_tiles part[b*tiles + tile, j, c]
__global__ __launch_bounds__(kThreads) void base_pmom_reduce_kernel(const float* __restrict__ part, float* __restrict__ pmom,
                                                                    int C, int t, int tiles) {
  const int b = blockIdx.y;
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < t * C; i += gridDim.x * kThreads) {
    const int j = i / C, ch = i - j * C;
    float s = 0.f;
    for (int k = 0; k < tiles; ++k) s += part[(((size_t)b * tiles + k) * t + j) * C + ch];
    pmom[((size_t)b * C + ch) * t + j] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// value backward: dx = [x > 0 if res&2] * ((res&1) * dOut + dwconv3x3^T(dV) + dyx);  dWv partials.
// Row-marching stencil kernel, lane = channel, as light_nhwc.hip; dV arrives in the storage type from base_combine<1>.
//   dx[r][col]  = sum_{i,k} w[i][k] * dV[r-i+1][col-k+1]
//   dWv[i][k]  += x[r][col] * dV[r-i+1][col-k+1]      (the same window, centred on x)
// PRE (WIDE, with the res&2 mask): the caller deferred the BatchNorm in front of the fused producer (bn3,
// resnet_mrla_base.py:103-104,120-122) and its backward needs sum(dpre) and sum(dpre * (y3 - mean)) per channel, y3 =
// conv3's raw output `pre`.  dpre = dx is formed here and nowhere else, so the two sums are taken here (one more owned-
// columns row fetch per step, two accumulators; of dpre AS STORED, i.e. rounded to T): mrla_bn_plane_dmoments' 2N pass over
// (dpre, y3) disappears -- the MRLA-light tail does the same in mrla_light_apply_bwd.
// ------------------------------------------------------------------------------------------------
template <typename T, bool WIDE, bool PRE = false>
__global__ __launch_bounds__(kMaxStrips * kWave) void base_value_bwd_nhwc(
    const T* __restrict__ dout, const T* __restrict__ x, const float* __restrict__ wv, const T* __restrict__ dv,
    const float* __restrict__ dyx, T* __restrict__ dx, float* __restrict__ dwv_part, const T* __restrict__ pre,
    const float* __restrict__ pre_center, float* __restrict__ pre_tmom, int B, int C, int H, int W, int BG, int res) {
  static_assert(!PRE || WIDE, "the deferred-BatchNorm sums exist on the whole-wave (C % 64 == 0) form only");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;
  float* red = reinterpret_cast<float*>(smem_raw);
  constexpr int FB = scratch_bytes<T>(), TB = scratch_bytes<T>();
  unsigned char* my = smem_raw + (size_t)nwaves * 9 * kWave * sizeof(float) + (size_t)wave * (FB + 2 * TB);
  T* scrF = reinterpret_cast<T*>(my);                      // dV gathers
  T* scrT = reinterpret_cast<T*>(my + FB);                 // x / dOut gathers
  T* scrS = reinterpret_cast<T*>(my + FB + TB);            // dx scatters
  const int cbase = blockIdx.x * kWave, c = cbase + lane;
  const bool cv = c < C;
  const int cc = cv ? c : C - 1;
  const int nstrips = (W + kS - 1) / kS;
  float w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = wv[cc * 9 + k];
  float wg[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float pm[2] = {0.f, 0.f};                          // (PRE) sum dpre, sum dpre * (y3 - center) over this workgroup's images
  const float pcen = (PRE && pre_center) ? pre_center[cc] : 0.f;
  const float resf = (res & 1) ? 1.f : 0.f;
  const bool mask = (res & 2) != 0;
  const int b_end = min(B, (int)(blockIdx.y + 1) * BG);
  for (int b = blockIdx.y * BG; b < b_end; ++b) {
    const size_t ioff = (size_t)b * H * W * C;
    const T* xi = x + ioff;
    const T* gi = dout + ioff;
    const T* ui = dv + ioff;
    const T* pri = PRE ? pre + ioff : nullptr;
    T* dxo = dx + ioff;
    const float dy = dyx[(size_t)b * C + cc];
    for (int s = wave; s < nstrips; s += nwaves) {
      const int s0 = s * kS, nc = min(kS, W - s0);
      float ua[kS + 2], ub[kS + 2], uc[kS + 2];                 // dV rows r-1, r, r+1 on columns s0-1 .. s0+kS
#pragma unroll
      for (int j = 0; j < kS + 2; ++j) ua[j] = 0.f;
      read_row<T, WIDE, kS + 2>(ui, 0, s0 - 1, H, W, C, cbase, cc, lane, scrF, ub);
      RowLoad<T, kS + 2> qu;
      RowLoad<T, kS> qx, qg, qp;
      RowAddr<T, kS + 2> au;
      RowAddr<T, kS> ax;
      if (WIDE) {
        make_row_addr<T, kS + 2>(au, s0 - 1, W, C, cbase, lane);
        make_row_addr<T, kS>(ax, s0, W, C, cbase, lane);
        issue_row<T, kS + 2>(qu, ui, 1, H, W * C, au);
        issue_row<T, kS>(qx, xi, 0, H, W * C, ax);
        issue_row<T, kS>(qg, gi, 0, H, W * C, ax);
        if (PRE) issue_row<T, kS>(qp, pri, 0, H, W * C, ax);
      }
      for (int r = 0; r < H; ++r) {
        float xr[kS], gr[kS], pr[kS];
        if (WIDE) {
          finish_row<T, kS + 2>(qu, lane, scrF, uc);
          finish_row<T, kS>(qx, lane, scrT, xr);
          finish_row<T, kS>(qg, lane, scrT, gr);
          if (PRE) finish_row<T, kS>(qp, lane, scrT, pr);
          issue_row<T, kS + 2>(qu, ui, r + 2, H, W * C, au);
          issue_row<T, kS>(qx, xi, r + 1, H, W * C, ax);
          issue_row<T, kS>(qg, gi, r + 1, H, W * C, ax);
          if (PRE) issue_row<T, kS>(qp, pri, r + 1, H, W * C, ax);
        } else {
          read_row<T, false, kS + 2>(ui, r + 1, s0 - 1, H, W, C, cbase, cc, lane, scrF, uc);
          read_row<T, false, kS>(xi, r, s0, H, W, C, cbase, cc, lane, scrT, xr);
          read_row<T, false, kS>(gi, r, s0, H, W, C, cbase, cc, lane, scrT, gr);
        }
        float yrow[kS];
#pragma unroll
        for (int j = 0; j < kS; ++j) {
          // window index of column (col + 1 - k) in the dV arrays (which start at col - 1): j + 2 - k
          float s9 = w[0] * uc[j + 2];
          s9 = fmaf(w[1], uc[j + 1], s9); s9 = fmaf(w[2], uc[j], s9);
          s9 = fmaf(w[3], ub[j + 2], s9); s9 = fmaf(w[4], ub[j + 1], s9); s9 = fmaf(w[5], ub[j], s9);
          s9 = fmaf(w[6], ua[j + 2], s9); s9 = fmaf(w[7], ua[j + 1], s9); s9 = fmaf(w[8], ua[j], s9);
          float y = fmaf(resf, gr[j], s9 + dy);
          if (mask) y = (xr[j] > 0.f) ? y : 0.f;
          yrow[j] = y;
          if (PRE && j < nc) {       // of dpre AS STORED (rounded to T): the BatchNorm backward must see the stored values
            const float yq = to_f(from_f<T>(y));
            pm[0] += yq;
            pm[1] = fmaf(yq, pr[j] - pcen, pm[1]);
          }
          if (j < nc) {
            const float xv = xr[j];
            wg[0] = fmaf(xv, uc[j + 2], wg[0]); wg[1] = fmaf(xv, uc[j + 1], wg[1]); wg[2] = fmaf(xv, uc[j], wg[2]);
            wg[3] = fmaf(xv, ub[j + 2], wg[3]); wg[4] = fmaf(xv, ub[j + 1], wg[4]); wg[5] = fmaf(xv, ub[j], wg[5]);
            wg[6] = fmaf(xv, ua[j + 2], wg[6]); wg[7] = fmaf(xv, ua[j + 1], wg[7]); wg[8] = fmaf(xv, ua[j], wg[8]);
          }
        }
        write_row<T, WIDE, kS>(dxo, r, s0, nc, W, C, cbase, c, cv, lane, scrS, yrow);
#pragma unroll
        for (int j = 0; j < kS + 2; ++j) { ua[j] = ub[j]; ub[j] = uc[j]; }
      }
    }
  }
  wg_reduce<9>(wg, red, lane, wave, nwaves);
  if (wave == 0 && cv) {
#pragma unroll
    for (int k = 0; k < 9; ++k) dwv_part[((size_t)blockIdx.y * C + c) * 9 + k] = wg[k];
  }
  if (PRE) {
    wg_reduce<2>(pm, red, lane, wave, nwaves);
    if (wave == 0 && cv) {
      pre_tmom[((size_t)blockIdx.y * C + c) * 2 + 0] = pm[0];
      pre_tmom[((size_t)blockIdx.y * C + c) * 2 + 1] = pm[1];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// geometry + launchers
// ------------------------------------------------------------------------------------------------
// The flat kernels need every thread to keep its channels from vector to vector: the workgroup size is the largest
// multiple of C / VEC up to 256 (256 for the power-of-two ResNet widths, 240 / 192 for DeiT's 192 / 384 / 768).
bool base_nhwc_supported(int C, int dtype) {
  const int vec = 16 / (int)dtype_size(dtype);
  return C % 64 == 0 && C / vec <= kThreads;
}
static int flat_threads(int C, int dtype) {
  const int lpp = C / (16 / (int)dtype_size(dtype));            // lanes (16-byte vectors) per pixel
  return (kThreads / lpp) * lpp;
}

// nv: 16-byte vectors per thread the kernel holds (kNVc for the history-combining kernels, kNV for the others)
static FlatGeo flat_geo(int B, int C, int HW, int dtype, int nv) {
  const int vec = 16 / (int)dtype_size(dtype);
  const int limit = std::max(1, nv * flat_threads(C, dtype) * vec / C);      // pixels that fit one tile
  int R = 1;
  for (int r = 1; r <= std::min(limit, HW); ++r)
    if (HW % r == 0) R = r;
  FlatGeo g;
  g.B = B; g.C = C; g.HW = HW; g.R = R; g.tiles = HW / R; g.nvec = R * C / vec;
  g.ext_gap = 0; g.ext_off = 0;
  return g;
}

// tiles per image of the forward statistics (attend) and of the backward <dA, V_j> partials
int base_nhwc_tiles(int B, int C, int HW, int dtype) { return flat_geo(B, C, HW, dtype, kNVc).tiles; }
int base_nhwc_pmom_tiles(int B, int C, int HW, int dtype) { return flat_geo(B, C, HW, dtype, kNV).tiles; }

#define MRLA_DISPATCH_B(DT, CALL)        \
  switch (DT) {                          \
    case MRLA_F32:  CALL(float); break;  \
    case MRLA_BF16: CALL(bf16_t); break; \
    case MRLA_F16:  CALL(f16_t); break;  \
    default: return MRLA_EINVAL;         \
  }

int launch_base_attend_fwd_nhwc(const void* Vring, const float* Pall, void* attn, float* amom_part, int B, int C, int HW,
                                int d, int T, int t, int dtype, hipStream_t st, int ext_gap, int ext_off) {
  if (!base_nhwc_supported(C, dtype)) return MRLA_EUNSUPPORTED;
  FlatGeo g = flat_geo(B, C, HW, dtype, kNVc);
  g.ext_gap = ext_gap; g.ext_off = ext_off;
#define CALL(TT)                                                                                                         \
  hipLaunchKernelGGL((base_combine_nhwc<TT, TT, 0, kNVc>), dim3(g.tiles, B), dim3(flat_threads(C, dtype)), 0, st, (const TT*)Vring, Pall, \
                     (TT*)attn, amom_part, g, d, T, t, t);
  MRLA_DISPATCH_B(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_base_dv_combine_nhwc(const void* dAring, const float* Pall, void* dv, int B, int C, int HW, int d, int T,
                                int t, int Tc, int dtype, hipStream_t st) {
  if (!base_nhwc_supported(C, dtype)) return MRLA_EUNSUPPORTED;
  const FlatGeo g = flat_geo(B, C, HW, dtype, kNVc);
#define CALL(TT)                                                                                                     \
  hipLaunchKernelGGL((base_combine_nhwc<TT, TT, 1, kNVc>), dim3(g.tiles, B), dim3(flat_threads(C, dtype)), 0, st, (const TT*)dAring, \
                     Pall, (TT*)dv, (float*)nullptr, g, d, T, t, Tc);
  MRLA_DISPATCH_B(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_base_tail_fwd_nhwc(const void* x, const void* attn, const float* sc, const float* sh, const float* dp,
                              void* out, int B, int C, int HW, int dtype, hipStream_t st) {
  if (!base_nhwc_supported(C, dtype)) return MRLA_EUNSUPPORTED;
  const FlatGeo g = flat_geo(B, C, HW, dtype, kNV);
#define CALL(TT)                                                                                                    \
  hipLaunchKernelGGL((base_tail_fwd_nhwc<TT>), dim3(g.tiles, B), dim3(flat_threads(C, dtype)), 0, st, (const TT*)x, (const TT*)attn, \
                     sc, sh, dp, (TT*)out, g);
  MRLA_DISPATCH_B(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_base_attend_bwd_nhwc(const void* dout, const void* attn, const float* sc, const float* sh, const float* dp,
                                const float* cb, const void* Vring, void* dAring, float* pmom_part, int B, int C, int HW,
                                int T, int t, int dtype, hipStream_t st, int ext_gap, int ext_off) {
  if (!base_nhwc_supported(C, dtype)) return MRLA_EUNSUPPORTED;
  FlatGeo g = flat_geo(B, C, HW, dtype, kNV);
  g.ext_gap = ext_gap; g.ext_off = ext_off;
#define CALL_O(TT, OCC)                                                                                           \
  hipLaunchKernelGGL((base_attend_bwd_nhwc<TT, OCC>), dim3(g.tiles, B), dim3(flat_threads(C, dtype)), 0, st, (const TT*)dout,    \
                     (const TT*)attn, sc, sh, dp, cb, (const TT*)Vring, (TT*)dAring, pmom_part, g, T, t);
#define CALL(TT) { if (t <= 16) CALL_O(TT, 3) else CALL_O(TT, 1) }
  MRLA_DISPATCH_B(dtype, CALL)
#undef CALL
#undef CALL_O
  return hip_status(hipGetLastError());
}

int launch_base_pmom_reduce(const float* part, float* pmom, int B, int C, int t, int tiles, hipStream_t st) {
  const int gx = std::max(1, std::min(64, (t * C + kThreads - 1) / kThreads));
  hipLaunchKernelGGL(base_pmom_reduce_kernel, dim3(gx, B), dim3(kThreads), 0, st, part, pmom, C, t, tiles);
  return hip_status(hipGetLastError());
}

int launch_base_value_bwd_nhwc(const void* dout, const void* x, const float* wv, const void* dv, const float* dyx,
                               void* dx, float* dwv_part, const void* pre, const float* pre_center, float* pre_tmom, int B,
                               int C, int H, int W, int res, int dtype, hipStream_t st) {
  const int nstrips = (W + kS - 1) / kS;
  const int nwaves = std::min(nstrips, kMaxStrips);
  const int BG = nhwc_images_per_group(B, C, W);
  const dim3 grid((C + kWave - 1) / kWave, (B + BG - 1) / BG), block(nwaves * kWave);
  // C % 64 == 0 runs on the LDS-DMA row pipeline (launch_base_value_bwd_wide, capi.hip); only the other channel counts come
  // here, and the deferred-BatchNorm sums (pre / pre_tmom) exist on that form only
  if (C % kWave == 0 || pre_tmom) return MRLA_EUNSUPPORTED;
  (void)pre;
  const size_t tb = dtype == MRLA_F32 ? scratch_bytes<float>() : scratch_bytes<bf16_t>();
  const size_t lds = (size_t)nwaves * 9 * kWave * sizeof(float) + (size_t)nwaves * 3 * tb;
#define CALL_W(TT, WD, PR)                                                                                          \
  {                                                                                                                 \
    if (set_lds_n(base_value_bwd_nhwc<TT, WD, PR>, lds) != hipSuccess) return MRLA_EHIP;                              \
    hipLaunchKernelGGL((base_value_bwd_nhwc<TT, WD, PR>), grid, block, lds, st, (const TT*)dout, (const TT*)x, wv,     \
                       (const TT*)dv, dyx, (TT*)dx, dwv_part, (const TT*)pre, pre_center, pre_tmom, B, C, H, W, BG, res); \
  }
#define CALL(TT) CALL_W(TT, false, false)
  MRLA_DISPATCH_B(dtype, CALL)
#undef CALL
#undef CALL_W
  return hip_status(hipGetLastError());
}

}  // namespace mrla
