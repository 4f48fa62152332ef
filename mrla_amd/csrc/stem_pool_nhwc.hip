// ResNet stem tail  maxpool3x3/s2/p1(relu(bn1(x)))  on a channels_last tensor, C % 64 == 0, without materialising the
// BatchNorm+ReLU output (resnet/models/resnet_mrla_light.py:220-222: `x = self.bn1(x); x = self.relu(x); x = self.maxpool(x)`
// on the 112x112x64 convolution output -- 411 MB at b = 256 in bf16, the largest activation of the network):
//   forward : out[ho, wo]   = max over the window of a,  a = round_T(relu(sc*x + sh))           (1 read of x, 1/4 write)
//   moments : tmom[row,c,0] = sum dz, [..,1] = sum dz*x,  dz = dP[window] at the window's first maximum if it is > 0
//   backward: dx            = cb0*dz + cb1*x + cb2     (dz gathered from the <= 4 windows a pixel belongs to)
// The stock route is bn_act (r+w 2N), max_pool2d (r N, w N/4 + int64 indices 2N), its backward (0.63 ms at b = 256:
// index-driven scatter) and the two BatchNorm backward passes over the full-size dz (5N): here x is read once in the
// forward and twice in the backward and nothing full-size is written except dx.
// The window maximum follows ATen's rule (first maximum in row-major scan order wins, `val > maxval`), on the values
// ROUNDED to the tensor type, as the stock max_pool2d sees them; a window whose maximum is 0 passes no gradient (ReLU).
// Row pipeline of nhwc_rows.h: lane = channel, a wave owns a strip of PS output columns and walks down the output rows;
// input rows 2ho-1 .. 2ho+1 rotate by name, row 2ho+1 is reused as the next window row's top.
#include <algorithm>

#include "light_nhwc.h"
#include "nhwc_rows.h"

namespace mrla {
namespace {

constexpr int kPoolWaves = 8;

template <typename T>
__device__ __forceinline__ float round_through(float v) { return to_f(from_f<T>(v)); }
template <>
__device__ __forceinline__ float round_through<float>(float v) { return v; }

// a[j] = rounded relu(bn(x[j])) of one row piece; -1 where the pixel does not exist (never a window's maximum)
template <typename T, int NPX>
__device__ __forceinline__ void activate(const RawRow<NPX>& x, float (&a)[NPX], bool rowok, int col0, int W, float sc, float sh) {
#pragma unroll
  for (int j = 0; j < NPX; ++j) {
    const bool ok = rowok && col0 + j >= 0 && col0 + j < W;        // wave-uniform
    a[j] = ok ? round_through<T>(fmaxf(fmaf(x.v[j], sc, sh), 0.f)) : -1.f;
  }
}

#define MRLA_POOL_PROLOGUE(NRED, WAVE_BYTES)                                                             \
  extern __shared__ __align__(16) unsigned char smem_raw[];                                              \
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;   \
  float* red = reinterpret_cast<float*>(smem_raw);                                                       \
  unsigned char* wbuf = smem_raw + (size_t)nwaves * (NRED) * kWave * sizeof(float) + (size_t)wave * (WAVE_BYTES); \
  const int cbase = blockIdx.x * kWave, c = cbase + lane;                                                \
  const int b = blockIdx.y / bands, band = blockIdx.y - b * bands;                                       \
  const int rows_per = (Ho + bands - 1) / bands;                                                         \
  const int ho0 = band * rows_per, ho1 = min(Ho, ho0 + rows_per);                                        \
  const int rowelems = W * C, orowelems = Wo * C;                                                        \
  const float scc = sc[c], shc = sh[c];                                                                  \
  (void)red;

// ------------------------------------------------------------------------------------------------
// forward; grid (C/64, B*bands)
// ------------------------------------------------------------------------------------------------
template <typename T, int PS>
constexpr int pool_fwd_wave_bytes() { return 2 * RowIO<T, 2 * PS + 1>::kBytes + RowIO<T, PS>::kBytes; }

template <typename T, int PS>
__global__ __launch_bounds__(kPoolWaves* kWave) void bn_relu_pool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ sc,
                                                                             const float* __restrict__ sh, T* __restrict__ out,
                                                                             int C, int H, int W, int Ho, int Wo, int bands) {
  constexpr int NX = 2 * PS + 1;
  MRLA_POOL_PROLOGUE(0, (pool_fwd_wave_bytes<T, PS>()))
  T* bufB = reinterpret_cast<T*>(wbuf);
  T* bufC = reinterpret_cast<T*>(wbuf + RowIO<T, NX>::kBytes);
  T* bufS = reinterpret_cast<T*>(wbuf + 2 * RowIO<T, NX>::kBytes);
  const T* xi = x + (size_t)b * H * rowelems;
  T* oi = out + (size_t)b * Ho * orowelems;
  const int nstrips = (Wo + PS - 1) / PS;
  for (int s = wave; s < nstrips; s += nwaves) {
    const int wo0 = s * PS, col0 = 2 * wo0 - 1;
    RowIO<T, NX> ax;
    RowIO<T, PS> ao;
    make_row_io<T, NX>(ax, col0, NX, W, C, cbase, lane);
    make_row_io<T, PS>(ao, wo0, min(PS, Wo - wo0), Wo, C, cbase, lane);
    RawRow<NX> raw;
    raw.clear();
    float aA[NX], aB[NX], aC[NX];
    row_fetch<T, NX>(ax, xi, 2 * ho0 - 1, H, rowelems, bufB);
    rows_landed();
    row_read<T, NX>(bufB, lane, raw);
    activate<T, NX>(raw, aA, 2 * ho0 - 1 >= 0, col0, W, scc, shc);
    row_fetch<T, NX>(ax, xi, 2 * ho0, H, rowelems, bufB);
    row_fetch<T, NX>(ax, xi, 2 * ho0 + 1, H, rowelems, bufC);
    for (int ho = ho0; ho < ho1; ++ho) {
      rows_landed();
      row_read<T, NX>(bufB, lane, raw);
      activate<T, NX>(raw, aB, true, col0, W, scc, shc);
      row_read<T, NX>(bufC, lane, raw);
      activate<T, NX>(raw, aC, 2 * ho + 1 < H, col0, W, scc, shc);
      if (ho + 1 < ho1) {
        row_fetch<T, NX>(ax, xi, 2 * ho + 2, H, rowelems, bufB);
        row_fetch<T, NX>(ax, xi, 2 * ho + 3, H, rowelems, bufC);
      }
      float y[PS];
#pragma unroll
      for (int v = 0; v < PS; ++v) {
        float m = fmaxf(fmaxf(aA[2 * v], aA[2 * v + 1]), aA[2 * v + 2]);
        m = fmaxf(m, fmaxf(fmaxf(aB[2 * v], aB[2 * v + 1]), aB[2 * v + 2]));
        m = fmaxf(m, fmaxf(fmaxf(aC[2 * v], aC[2 * v + 1]), aC[2 * v + 2]));
        y[v] = m;                                   // >= 0: the window centre always exists
      }
      row_store<T, PS>(ao, oi, ho, orowelems, lane, bufS, y);
#pragma unroll
      for (int j = 0; j < NX; ++j) aA[j] = aC[j];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The two backward kernels fetch TWO window rows ahead (two sets of row buffers by parity): with ~100 registers per lane
// only 8-16 waves fit a CU, too few to cover the DMA latency with one row in flight.  The wait in front of a step leaves
// exactly the newer step's fetches (and the previous step's row stores) outstanding: the counts are compile-time
// constants because a step ALWAYS issues the same instructions -- rows that do not exist (or are not this band's) are
// fetched / stored through an empty buffer descriptor, which moves no data.
// ------------------------------------------------------------------------------------------------
template <typename T, int NPX>
__device__ __forceinline__ void row_store_live(const RowIO<T, NPX>& a, T* img, int r, bool live, int rowelems, int lane, T* buf,
                                               const float (&v)[NPX]) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef RowIO<T, NPX> Q;
  typedef __attribute__((address_space(3))) T* lds_T_ptr;
  lds_T_ptr p = (lds_T_ptr)buf + lane;
#pragma unroll
  for (int j = 0; j < NPX; ++j) p[j * kWave] = from_f<T>(v[j]);
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(img + (size_t)(live ? r : 0) * rowelems, 0,
                                                    live ? rowelems * (int)sizeof(T) : 0, kBufFlags);
  typedef __attribute__((address_space(3))) const u32x4* lds_v4_ptr;
  lds_v4_ptr s4 = (lds_v4_ptr)buf + lane;
#pragma unroll
  for (int l = 0; l < Q::NL; ++l) __builtin_amdgcn_raw_buffer_store_b128(s4[l * kWave], rs, a.voff[l], 0, 0);
#endif
}

template <int N>
__device__ __forceinline__ void pool_wait() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// ------------------------------------------------------------------------------------------------
// backward statistics; grid (C/64, B*bands); tmom[blockIdx.y][c][0..1]
// ------------------------------------------------------------------------------------------------
template <typename T, int PS>
constexpr int pool_mom_set_bytes() { return 2 * RowIO<T, 2 * PS + 1>::kBytes + RowIO<T, PS>::kBytes; }
template <typename T, int PS>
constexpr int pool_mom_wave_bytes() { return 2 * pool_mom_set_bytes<T, PS>(); }

template <typename T, int PS>
__global__ __launch_bounds__(kPoolWaves* kWave) void bn_relu_pool_dmoments_kernel(const T* __restrict__ dp, const T* __restrict__ x,
                                                                                  const float* __restrict__ sc, const float* __restrict__ sh,
                                                                                  const float* __restrict__ center,
                                                                                  float* __restrict__ tmom, int C, int H, int W, int Ho,
                                                                                  int Wo, int bands) {
  constexpr int NX = 2 * PS + 1;
  constexpr int F = 2 * RowIO<T, NX>::NL + RowIO<T, PS>::NL;       // fetch instructions of one step
  MRLA_POOL_PROLOGUE(2, (pool_mom_wave_bytes<T, PS>()))
  const float ncen = center ? -center[c] : 0.f;                    // sum dz*(x - center): dz*x - center*dz, term by term
  const T* xi = x + (size_t)b * H * rowelems;
  const T* gi = dp + (size_t)b * Ho * orowelems;
  const int nstrips = (Wo + PS - 1) / PS;
  float acc[2] = {0.f, 0.f};
  for (int s = wave; s < nstrips; s += nwaves) {
    const int wo0 = s * PS, col0 = 2 * wo0 - 1;
    RowIO<T, NX> ax;
    RowIO<T, PS> ag;
    make_row_io<T, NX>(ax, col0, NX, W, C, cbase, lane);
    make_row_io<T, PS>(ag, wo0, PS, Wo, C, cbase, lane);          // columns beyond Wo read as 0: no contribution
    auto fetch = [&](int ho) {                                      // window row ho -> buffer set (ho - ho0) & 1
      unsigned char* set = wbuf + ((ho - ho0) & 1) * pool_mom_set_bytes<T, PS>();
      const bool ok = ho < ho1;
      row_fetch<T, NX>(ax, xi, ok ? 2 * ho : -1, H, rowelems, reinterpret_cast<T*>(set));
      row_fetch<T, NX>(ax, xi, ok ? 2 * ho + 1 : -1, H, rowelems, reinterpret_cast<T*>(set + RowIO<T, NX>::kBytes));
      row_fetch<T, PS>(ag, gi, ok ? ho : -1, Ho, orowelems, reinterpret_cast<T*>(set + 2 * RowIO<T, NX>::kBytes));
    };
    RawRow<NX> xA, xB, xC;
    RawRow<PS> g;
    xA.clear(); xB.clear(); xC.clear(); g.clear();
    float a0[NX], a1[NX], a2[NX];
    row_fetch<T, NX>(ax, xi, 2 * ho0 - 1, H, rowelems, reinterpret_cast<T*>(wbuf));
    rows_landed();
    row_read<T, NX>(reinterpret_cast<T*>(wbuf), lane, xA);
    activate<T, NX>(xA, a0, 2 * ho0 - 1 >= 0, col0, W, scc, shc);
    fetch(ho0);
    fetch(ho0 + 1);
    auto step = [&](int ho, RawRow<NX>& XA, RawRow<NX>& XB, RawRow<NX>& XC, float (&aA)[NX], float (&aB)[NX], float (&aC)[NX]) {
      pool_wait<F>();                                               // this step's rows are in; the next step's may be in flight
      unsigned char* set = wbuf + ((ho - ho0) & 1) * pool_mom_set_bytes<T, PS>();
      row_read<T, NX>(reinterpret_cast<T*>(set), lane, XB);
      row_read<T, NX>(reinterpret_cast<T*>(set + RowIO<T, NX>::kBytes), lane, XC);
      row_read<T, PS>(reinterpret_cast<T*>(set + 2 * RowIO<T, NX>::kBytes), lane, g);
      fetch(ho + 2);
      activate<T, NX>(XB, aB, true, col0, W, scc, shc);
      activate<T, NX>(XC, aC, 2 * ho + 1 < H, col0, W, scc, shc);
#pragma unroll
      for (int v = 0; v < PS; ++v) {
        const float m = max3(max3(aA[2 * v], aA[2 * v + 1], aA[2 * v + 2]), max3(aB[2 * v], aB[2 * v + 1], aB[2 * v + 2]),
                             max3(aC[2 * v], aC[2 * v + 1], aC[2 * v + 2]));
        // dz = dP at the FIRST position (scan order) that holds the maximum, if that maximum is > 0
        // (explicit first-hit masks: a chain of selects over the row arrays is turned into a scratch-memory lookup)
        const float gg = m > 0.f ? g.v[v] : 0.f;
        acc[0] += gg;
        bool open = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const bool hit = open && aA[2 * v + k] == m;
          acc[1] = fmaf(hit ? gg : 0.f, XA.v[2 * v + k], acc[1]);
          open = open && !hit;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const bool hit = open && aB[2 * v + k] == m;
          acc[1] = fmaf(hit ? gg : 0.f, XB.v[2 * v + k], acc[1]);
          open = open && !hit;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const bool hit = open && aC[2 * v + k] == m;
          acc[1] = fmaf(hit ? gg : 0.f, XC.v[2 * v + k], acc[1]);
          open = open && !hit;
        }
        acc[1] = fmaf(ncen, gg, acc[1]);       // (gg != 0 implies exactly one hit above)
      }
    };
    // (the row windows rotate by name: the raw rows' registers are only ever written by the LDS reads)
    int ho = ho0;
    for (; ho + 3 <= ho1; ho += 3) {
      step(ho, xA, xB, xC, a0, a1, a2);
      step(ho + 1, xC, xA, xB, a2, a0, a1);
      step(ho + 2, xB, xC, xA, a1, a2, a0);
    }
    if (ho < ho1) {
      step(ho, xA, xB, xC, a0, a1, a2);
      if (ho + 1 < ho1) step(ho + 1, xC, xA, xB, a2, a0, a1);
    }
    rows_landed();                                                  // (the dead fetches past the band)
  }
  wg_reduce<2>(acc, red, lane, wave, nwaves);
  if (wave == 0) {
    float* t = tmom + ((size_t)blockIdx.y * C + c) * 2;
    t[0] = acc[0];
    t[1] = acc[1];
  }
}

// ------------------------------------------------------------------------------------------------
// backward; grid (C/64, B*bands).  A strip owns the input columns 2*wo0 .. 2*wo0 + 2*PS - 1 and evaluates the PS + 1
// windows that touch them; a band owns the input rows 2*ho0 .. 2*ho1 - 1 (the last band: to the end of the image) and
// evaluates window row ho1 too, for what it hands to its top row.
// ------------------------------------------------------------------------------------------------
template <typename T, int PS>
constexpr int pool_bwd_set_bytes() { return 2 * RowIO<T, 2 * PS + 3>::kBytes + RowIO<T, PS + 1>::kBytes; }
template <typename T, int PS>
constexpr int pool_bwd_wave_bytes() { return 2 * pool_bwd_set_bytes<T, PS>() + RowIO<T, 2 * PS>::kBytes; }

template <typename T, int PS>
__global__ __launch_bounds__(kPoolWaves* kWave) void bn_relu_pool_bwd_kernel(const T* __restrict__ dp, const T* __restrict__ x,
                                                                             const float* __restrict__ sc, const float* __restrict__ sh,
                                                                             const float* __restrict__ cb, T* __restrict__ dx, int C,
                                                                             int H, int W, int Ho, int Wo, int bands) {
  constexpr int NX = 2 * PS + 3, NG = PS + 1, NO = 2 * PS;
  constexpr int F = 2 * RowIO<T, NX>::NL + RowIO<T, NG>::NL;       // fetch instructions of one step
  constexpr int S = 2 * RowIO<T, NO>::NL;                          // store instructions of one step
  MRLA_POOL_PROLOGUE(0, (pool_bwd_wave_bytes<T, PS>()))
  T* bufS = reinterpret_cast<T*>(wbuf + 2 * pool_bwd_set_bytes<T, PS>());
  const T* xi = x + (size_t)b * H * rowelems;
  const T* gi = dp + (size_t)b * Ho * orowelems;
  T* di = dx + (size_t)b * H * rowelems;
  const float cb0 = cb[c * 3 + 0], cb1 = cb[c * 3 + 1], cb2 = cb[c * 3 + 2];
  const int nstrips = (Wo + PS - 1) / PS;
  const int ho_last = min(ho1, Ho - 1);              // window rows ho0 .. ho_last are evaluated
  const bool tail = ho1 == Ho;                       // this band also finishes the row below the last window centre
  for (int s = wave; s < nstrips; s += nwaves) {
    const int wo0 = s * PS, col0 = 2 * wo0 - 1;
    RowIO<T, NX> ax;
    RowIO<T, NG> ag;
    RowIO<T, NO> ad;
    make_row_io<T, NX>(ax, col0, NX, W, C, cbase, lane);
    make_row_io<T, NG>(ag, wo0, NG, Wo, C, cbase, lane);
    make_row_io<T, NO>(ad, 2 * wo0, min(NO, W - 2 * wo0), W, C, cbase, lane);
    auto fetch = [&](int ho) {                                      // window row ho -> buffer set (ho - ho0) & 1
      unsigned char* set = wbuf + ((ho - ho0) & 1) * pool_bwd_set_bytes<T, PS>();
      const bool ok = ho <= ho_last;
      row_fetch<T, NX>(ax, xi, ok ? 2 * ho : -1, H, rowelems, reinterpret_cast<T*>(set));
      row_fetch<T, NX>(ax, xi, ok ? 2 * ho + 1 : -1, H, rowelems, reinterpret_cast<T*>(set + RowIO<T, NX>::kBytes));
      row_fetch<T, NG>(ag, gi, ok ? ho : -1, Ho, orowelems, reinterpret_cast<T*>(set + 2 * RowIO<T, NX>::kBytes));
    };
    auto finish = [&](int row, bool live, const RawRow<NX>& X, const float (&D)[NX]) {
      float v[NO];
#pragma unroll
      for (int j = 0; j < NO; ++j) v[j] = fmaf(cb0, D[j + 1], fmaf(cb1, X.v[j + 1], cb2));
      row_store_live<T, NO>(ad, di, row, live, rowelems, lane, bufS, v);
    };
    RawRow<NX> xA, xB, xC;
    RawRow<NG> g;
    xA.clear(); xB.clear(); xC.clear(); g.clear();
    float a0[NX], a1[NX], a2[NX], d0[NX], d1[NX], d2[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) d0[j] = 0.f;
    row_fetch<T, NX>(ax, xi, 2 * ho0 - 1, H, rowelems, reinterpret_cast<T*>(wbuf));
    rows_landed();
    row_read<T, NX>(reinterpret_cast<T*>(wbuf), lane, xA);
    activate<T, NX>(xA, a0, 2 * ho0 - 1 >= 0, col0, W, scc, shc);
    fetch(ho0);
    fetch(ho0 + 1);
    auto step = [&](int ho, RawRow<NX>& XA, RawRow<NX>& XB, RawRow<NX>& XC, float (&aA)[NX], float (&aB)[NX], float (&aC)[NX],
                    float (&dA)[NX], float (&dB)[NX], float (&dC)[NX]) {
      // this step's rows are in; the next step's fetches and the previous step's stores may still be in flight
      if (ho == ho0) pool_wait<F>(); else pool_wait<F + S>();
      unsigned char* set = wbuf + ((ho - ho0) & 1) * pool_bwd_set_bytes<T, PS>();
      row_read<T, NX>(reinterpret_cast<T*>(set), lane, XB);
      row_read<T, NX>(reinterpret_cast<T*>(set + RowIO<T, NX>::kBytes), lane, XC);
      row_read<T, NG>(reinterpret_cast<T*>(set + 2 * RowIO<T, NX>::kBytes), lane, g);
      activate<T, NX>(XB, aB, true, col0, W, scc, shc);
      activate<T, NX>(XC, aC, 2 * ho + 1 < H, col0, W, scc, shc);
#pragma unroll
      for (int j = 0; j < NX; ++j) { dB[j] = 0.f; dC[j] = 0.f; }
#pragma unroll
      for (int v = 0; v < NG; ++v) {
        const float m = max3(max3(aA[2 * v], aA[2 * v + 1], aA[2 * v + 2]), max3(aB[2 * v], aB[2 * v + 1], aB[2 * v + 2]),
                             max3(aC[2 * v], aC[2 * v + 1], aC[2 * v + 2]));
        // the gradient goes to the FIRST position (scan order) that holds the maximum, if that maximum is > 0
        // (dP is 0 for window columns beyond Wo)
        bool open = m > 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const bool hit = open && aA[2 * v + k] == m;
          dA[2 * v + k] += hit ? g.v[v] : 0.f;
          open = open && !hit;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const bool hit = open && aB[2 * v + k] == m;
          dB[2 * v + k] += hit ? g.v[v] : 0.f;
          open = open && !hit;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const bool hit = open && aC[2 * v + k] == m;
          dC[2 * v + k] += hit ? g.v[v] : 0.f;
          open = open && !hit;
        }
      }
      // rows 2ho-1 and 2ho have all their windows now (always two stores: a row that is not this band's goes nowhere)
      finish(2 * ho - 1, ho > ho0, XA, dA);
      finish(2 * ho, ho < ho1, XB, dB);
      fetch(ho + 2);
    };
    int ho = ho0;
    for (; ho + 3 <= ho_last + 1; ho += 3) {
      step(ho, xA, xB, xC, a0, a1, a2, d0, d1, d2);
      step(ho + 1, xC, xA, xB, a2, a0, a1, d2, d0, d1);
      step(ho + 2, xB, xC, xA, a1, a2, a0, d1, d2, d0);
    }
    const int left = ho_last + 1 - ho;                 // 0, 1 or 2 steps remain
    if (left == 0) {
      if (tail && 2 * Ho - 1 < H) finish(2 * Ho - 1, true, xA, d0);       // the last step's bottom row
    } else if (left == 1) {
      step(ho, xA, xB, xC, a0, a1, a2, d0, d1, d2);
      if (tail && 2 * Ho - 1 < H) finish(2 * Ho - 1, true, xC, d2);
    } else {
      step(ho, xA, xB, xC, a0, a1, a2, d0, d1, d2);
      step(ho + 1, xC, xA, xB, a2, a0, a1, d2, d0, d1);
      if (tail && 2 * Ho - 1 < H) finish(2 * Ho - 1, true, xB, d1);
    }
    rows_landed();                                      // stores and the dead fetches: the row buffers are reused
  }
}

template <typename K>
hipError_t pool_lds(K kernel, size_t bytes) {
  return lds_opt_in(reinterpret_cast<const void*>(kernel), bytes);
}

constexpr int kPS = 4;

struct PoolGeo {
  int Ho, Wo, bands, nwaves;
};
PoolGeo pool_geo(int B, int C, int H, int W, int ps) {
  PoolGeo g;
  g.Ho = (H - 1) / 2 + 1;
  g.Wo = (W - 1) / 2 + 1;
  g.nwaves = std::min(kPoolWaves, (g.Wo + ps - 1) / ps);
  // ~2 workgroups per CU; bands must split the pixel count evenly for the statistics kernel's (rows, count) form.
  // (up to 32 row bands: a detection batch -- 2 x 64 x 400 x 672 -- is two (image, channel-group) pairs; with the 4 bands that
  // were enough at b = 256 its stem ran on 8 workgroups at 0.17 TB/s, round 6)
  const int wgs = (C / kWave) * B;
  g.bands = 1;
  while (wgs * g.bands < 512 && g.bands < 32 && (H * W) % (g.bands * 2) == 0 && g.Ho / (g.bands * 2) >= 4) g.bands *= 2;
  return g;
}

}  // namespace

bool bn_pool_supported(int B, int C, int H, int W) { return C % kWave == 0 && H >= 2 && W >= 2 && B > 0; }

int bn_pool_rows(int B, int C, int H, int W) {
  if (!bn_pool_supported(B, C, H, W)) return MRLA_EUNSUPPORTED;
  return B * pool_geo(B, C, H, W, kPS).bands;
}

#define MRLA_POOL_DISPATCH(DT, CALL)     \
  switch (DT) {                          \
    case MRLA_F32:  CALL(float); break;  \
    case MRLA_BF16: CALL(bf16_t); break; \
    case MRLA_F16:  CALL(f16_t); break;  \
    default: return MRLA_EINVAL;         \
  }

int launch_bn_relu_pool_fwd(const void* x, const float* sc, const float* sh, void* out, int B, int C, int H, int W,
                            int dtype, hipStream_t st) {
  if (!bn_pool_supported(B, C, H, W)) return MRLA_EUNSUPPORTED;
  const PoolGeo g = pool_geo(B, C, H, W, kPS);
  const dim3 grid(C / kWave, B * g.bands), block(g.nwaves * kWave);
#define CALL(TT)                                                                                                     \
  {                                                                                                                  \
    const size_t lds = (size_t)g.nwaves * pool_fwd_wave_bytes<TT, kPS>();                                            \
    if (pool_lds(bn_relu_pool_fwd_kernel<TT, kPS>, lds) != hipSuccess) return MRLA_EHIP;                             \
    hipLaunchKernelGGL((bn_relu_pool_fwd_kernel<TT, kPS>), grid, block, lds, st, (const TT*)x, sc, sh, (TT*)out, C, \
                       H, W, g.Ho, g.Wo, g.bands);                                                                   \
  }
  MRLA_POOL_DISPATCH(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_bn_relu_pool_dmoments(const void* dp, const void* x, const float* sc, const float* sh, const float* center,
                                 float* tmom, int B, int C, int H, int W, int dtype, hipStream_t st) {
  if (!bn_pool_supported(B, C, H, W)) return MRLA_EUNSUPPORTED;
  const PoolGeo g = pool_geo(B, C, H, W, kPS);
  const dim3 grid(C / kWave, B * g.bands), block(g.nwaves * kWave);
#define CALL(TT)                                                                                                    \
  {                                                                                                                 \
    const size_t lds = (size_t)g.nwaves * (2 * kWave * sizeof(float) + pool_mom_wave_bytes<TT, kPS>());             \
    if (pool_lds(bn_relu_pool_dmoments_kernel<TT, kPS>, lds) != hipSuccess) return MRLA_EHIP;                       \
    hipLaunchKernelGGL((bn_relu_pool_dmoments_kernel<TT, kPS>), grid, block, lds, st, (const TT*)dp, (const TT*)x,  \
                       sc, sh, center, tmom, C, H, W, g.Ho, g.Wo, g.bands);                                         \
  }
  MRLA_POOL_DISPATCH(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

int launch_bn_relu_pool_bwd(const void* dp, const void* x, const float* sc, const float* sh, const float* cb, void* dx,
                            int B, int C, int H, int W, int dtype, hipStream_t st) {
  if (!bn_pool_supported(B, C, H, W)) return MRLA_EUNSUPPORTED;
  const PoolGeo g = pool_geo(B, C, H, W, kPS);
  const dim3 grid(C / kWave, B * g.bands), block(g.nwaves * kWave);
#define CALL(TT)                                                                                                   \
  {                                                                                                                \
    const size_t lds = (size_t)g.nwaves * pool_bwd_wave_bytes<TT, kPS>();                                          \
    if (pool_lds(bn_relu_pool_bwd_kernel<TT, kPS>, lds) != hipSuccess) return MRLA_EHIP;                           \
    hipLaunchKernelGGL((bn_relu_pool_bwd_kernel<TT, kPS>), grid, block, lds, st, (const TT*)dp, (const TT*)x, sc,  \
                       sh, cb, (TT*)dx, C, H, W, g.Ho, g.Wo, g.bands);                                             \
  }
  MRLA_POOL_DISPATCH(dtype, CALL)
#undef CALL
  return hip_status(hipGetLastError());
}

}  // namespace mrla
