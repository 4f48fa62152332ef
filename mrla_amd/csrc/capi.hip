// extern "C" surface of libmrla_hip.so (declared in include/mrla_hip.h): argument validation, slab
// geometry, dispatch on dtype / layout.  No global state; nothing here allocates or synchronises.
#include <algorithm>
#include <atomic>
#include <mutex>

#include "mrla_device.h"
#include "mrla_kernels.h"

namespace mrla {

// What hipFuncAttributeMaxDynamicSharedMemorySize was last raised to, per (kernel, device): an open-addressed table of
// atomics read without a lock; raising it (a handful of times per kernel in a process) is serialised by a mutex so
// that the attribute always equals the largest size ever asked for.  A cache of an idempotent driver call, no
// behavioural state: a full table degrades to one attribute call per launch.
hipError_t lds_opt_in(const void* kernel, size_t bytes) {
  if (bytes <= 48 * 1024) return hipSuccess;
  constexpr unsigned kSlots = 1024;
  static std::atomic<uintptr_t> keys[kSlots];
  static std::atomic<int> granted[kSlots];
  static std::mutex raise_mutex;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
  const uintptr_t key = reinterpret_cast<uintptr_t>(kernel) ^ ((uintptr_t)(dev + 1) << 52);
  unsigned i = (unsigned)((key >> 4) * 0x9E3779B1u) % kSlots;
  int slot = -1;
  for (unsigned probe = 0; probe < kSlots; ++probe, i = (i + 1) % kSlots) {
    uintptr_t k = keys[i].load(std::memory_order_acquire);
    if (k == 0 && keys[i].compare_exchange_strong(k, key, std::memory_order_acq_rel)) k = key;
    if (k == key) { slot = (int)i; break; }
  }
  if (slot >= 0 && granted[slot].load(std::memory_order_acquire) >= (int)bytes) return hipSuccess;
  std::lock_guard<std::mutex> lock(raise_mutex);
  if (slot >= 0 && granted[slot].load(std::memory_order_relaxed) >= (int)bytes) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess && slot >= 0) granted[slot].store((int)bytes, std::memory_order_release);
  return e;
}

constexpr int kWavesPerGroup = 4;
static int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

// LDS budget: `arrays` slab-sized LDS arrays per workgroup (double-buffered inputs + output staging; 8 for the
// backward apply kernel).  Keep them near 64 KB so 2-3 workgroups share a CU, never above 150 KB (160 KB per CU).
int make_slab_geo(SlabGeo* g, int B, int C, int H, int W, int dtype, int arrays, int bg_hint) {
  if (W > 64) return MRLA_EUNSUPPORTED;               // a plane row must fit one wave
  const int es = (int)dtype_size(dtype);
  const int HW = H * W;
  const int soft = 64 * 1024, hard = 150 * 1024;
  const int target = std::min(4096, soft / (arrays * es));
  const int max_tasks = kMaxTasksPerWave * kWavesPerGroup;
  const int maxplanes = std::max(1, std::min(C, target / HW));
  int WS = 1;
  while (WS < W) WS <<= 1;
  int PW = std::max(1, std::min(64 / WS, maxplanes));
  int NGc = std::max(1, maxplanes / PW);
  if (NGc >= kWavesPerGroup) NGc -= NGc % kWavesPerGroup;     // whole rounds over the 4 waves
  NGc = std::min(NGc, max_tasks);
  int CP = PW * NGc;
  // make every slab start 16-byte aligned when the tensor allows it
  const int q = (16 / es) / gcd_i(16 / es, HW);
  if (CP >= q) CP -= CP % q;
  if (PW > CP) PW = CP;
  const int NG = (CP + PW - 1) / PW;
  int NB = std::max(1, (kWavesPerGroup + NG - 1) / NG);
  NB = std::min(NB, std::max(1, H / 4));
  NB = std::min(NB, max_tasks / NG > 0 ? max_tasks / NG : 1);
  const int RB = (H + NB - 1) / NB;
  NB = (H + RB - 1) / RB;
  const int vec = 16 / es;
  const int astride = ((CP * HW + vec - 1) / vec) * vec;
  const size_t lds = (size_t)astride * es * arrays + (size_t)NG * NB * PW * 9 * sizeof(float);
  if (lds > (size_t)hard) return MRLA_EUNSUPPORTED;
  const int slabs = (C + CP - 1) / CP;
  int BG = bg_hint;
  if (BG <= 0) {
    const long total = (long)B * slabs;
    BG = (int)std::max(1L, std::min(8L, total / 2048));
  }
  *g = SlabGeo{B, C, H, W, HW, CP, slabs, WS, PW, NG, NB, RB, BG, astride};
  return MRLA_OK;
}

static bool bad_dims(int b, int c, int h, int w) { return b <= 0 || c <= 0 || h <= 0 || w <= 0; }
static bool bad_dtype(int dt) { return dt != MRLA_F32 && dt != MRLA_BF16 && dt != MRLA_F16; }

// One geometry for all four streaming kernels of a problem (sized for the 5-array backward pass), so
// the wgrad partial-row count is a pure function of the shape.
static int light_geo(SlabGeo* g, int b, int c, int h, int w, int dtype) { return make_slab_geo(g, b, c, h, w, dtype, 6, 0); }

}  // namespace mrla

using namespace mrla;

extern "C" {

int mrla_abi_version(void) { return MRLA_ABI_VERSION; }

int mrla_light_wgrad_rows(int b, int c, int h, int w, int dtype, int layout) {
  if (bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout == MRLA_NHWC) {          // image groups x (strip ranges x row ranges) (ranges > 1: few, large images -- detection batches)
    const int bg = nhwc_images_per_group(b, c, w);
    return (b + bg - 1) / bg * nhwc_wgrad_ranges(b, c, h, w);
  }
  if (layout != MRLA_NCHW) return MRLA_EINVAL;
  SlabGeo g;
  const int rc = light_geo(&g, b, c, h, w, dtype);
  if (rc != MRLA_OK) return rc;
  return (b + g.BG - 1) / g.BG;
}

int mrla_bn_moment_rows(int b, int c, int h, int w, int layout) {
  if (bad_dims(b, c, h, w)) return MRLA_EINVAL;
  return layout == MRLA_NHWC ? b * nhwc_bn_splits(b, c, h * w) : b;
}

int mrla_light_stats_fwd(const void* x, const void* o_prev, const float* wv, float* mom, int b, int c, int h, int w,
                         int dtype, int layout, int act, void* stream) {
  if (!x || !wv || !mom || bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout == MRLA_NHWC)
    return launch_light_stats_fwd_nhwc(x, o_prev, wv, mom, nullptr, nullptr, nullptr, nullptr, b, c, h, w, dtype, act,
                                       (hipStream_t)stream, false, nhwc_mom_ranges(b, c, h, w));
  if (layout != MRLA_NCHW) return MRLA_EINVAL;
  SlabGeo g;
  const int rc = light_geo(&g, b, c, h, w, dtype);
  if (rc != MRLA_OK) return rc;
  return launch_light_stats_fwd_nchw(x, o_prev, wv, mom, nullptr, nullptr, nullptr, g, dtype, act, (hipStream_t)stream);
}

int mrla_light_stats_fwd_fused(const void* pre, const float* pre_sc, const float* pre_sh, const void* o_prev,
                               const float* wv, float* mom, void* x_out, int b, int c, int h, int w, int dtype,
                               int layout, void* stream) {
  if (!pre || !o_prev || !wv || !mom || bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if ((pre_sc == nullptr) != (pre_sh == nullptr)) return MRLA_EINVAL;
  if (!x_out && mrla_light_lean_supported(b, c, h, w, dtype, layout) != 1) return MRLA_EUNSUPPORTED;
  if (layout == MRLA_NHWC)
    return launch_light_stats_fwd_nhwc(pre, o_prev, wv, mom, x_out, pre_sc, pre_sh, nullptr, b, c, h, w, dtype,
                                       MRLA_ACT_NONE, (hipStream_t)stream, x_out == nullptr, nhwc_mom_ranges(b, c, h, w));
  if (layout != MRLA_NCHW) return MRLA_EINVAL;
  SlabGeo g;
  const int rc = light_geo(&g, b, c, h, w, dtype);
  if (rc != MRLA_OK) return rc;
  return launch_light_stats_fwd_nchw(pre, o_prev, wv, mom, x_out, pre_sc, pre_sh, g, dtype, MRLA_ACT_NONE,
                                     (hipStream_t)stream);
}

int mrla_light_pool_fused(const void* pre, const float* pre_sc, const float* pre_sh, const void* o_prev, float* part,
                          float* mom, int b, int c, int h, int w, int dtype, int layout, void* stream) {
  if (!pre || !o_prev || !part || !mom || bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if ((pre_sc == nullptr) != (pre_sh == nullptr)) return MRLA_EINVAL;
  if (layout != MRLA_NHWC) return MRLA_EUNSUPPORTED;
  return launch_nhwc_pool_fused(pre, pre_sc, pre_sh, o_prev, part, mom, b, c, h * w, dtype, (hipStream_t)stream);
}

int mrla_light_apply_fwd_fused(const void* pre, const float* pre_sc, const float* pre_sh, const void* o_prev,
                               const float* wv, const float* gate, const float* sc, const float* sh, const float* lam,
                               const float* dp, void* out, int b, int c, int h, int w, int d, int res, int dtype,
                               int layout, void* stream) {
  if (!pre || !o_prev || !wv || !gate || !out || bad_dims(b, c, h, w) || bad_dtype(dtype) || d <= 0 || c % d)
    return MRLA_EINVAL;
  if ((pre_sc == nullptr) != (pre_sh == nullptr) || (sc == nullptr) != (sh == nullptr)) return MRLA_EINVAL;
  if (layout != MRLA_NHWC) return MRLA_EUNSUPPORTED;
  return launch_light_apply_fwd_pre_nhwc(pre, o_prev, pre_sc, pre_sh, wv, gate, sc, sh, lam, dp, out, b, c, h, w, d, res,
                                         dtype, (hipStream_t)stream);
}

int mrla_light_gate_fwd(const float* mom, const float* wq, const float* wk, int ksize, float* gate, int b, int c,
                        int hw, int d, void* stream) {
  if (!mom || !wq || !wk || !gate || b <= 0 || c <= 0 || hw <= 0 || d <= 0 || c % d || ksize <= 0 || !(ksize & 1))
    return MRLA_EINVAL;
  return launch_gate_fwd(mom, wq, wk, ksize, gate, b, c, hw, d, (hipStream_t)stream);
}

int mrla_light_bn_fwd(const float* mom, const float* gate, const float* lam, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, int bn_mode, float momentum, float eps, float* sc,
                      float* sh, float* save_mean, float* save_inv, int b, int c, int hw, int d, void* stream) {
  if (!mom || !gate || !gamma || !beta || !running_mean || !running_var || !sc || !sh || !save_mean || !save_inv ||
      b <= 0 || c <= 0 || hw <= 0 || d <= 0 || c % d || (bn_mode != MRLA_BN_TRAIN && bn_mode != MRLA_BN_EVAL))
    return MRLA_EINVAL;
  return launch_bn_fwd(mom, gate, lam, gamma, beta, running_mean, running_var, bn_mode == MRLA_BN_TRAIN, momentum, eps,
                       sc, sh, save_mean, save_inv, b, c, hw, d, (hipStream_t)stream);
}

int mrla_light_apply_fwd(const void* x, const void* o_prev, const float* wv, const float* gate, const float* sc,
                         const float* sh, const float* lam, const float* dp, void* out, int b, int c, int h, int w,
                         int d, int res, int dtype, int layout, int act, void* stream) {
  if (!x || !wv || !gate || !out || bad_dims(b, c, h, w) || bad_dtype(dtype) || d <= 0 || c % d) return MRLA_EINVAL;
  if (o_prev && !lam) return MRLA_EINVAL;
  if (layout == MRLA_NHWC)
    return launch_light_apply_fwd_nhwc(x, o_prev, wv, gate, sc, sh, lam, dp, out, b, c, h, w, d, res, dtype, act,
                                       (hipStream_t)stream);
  if (layout != MRLA_NCHW) return MRLA_EINVAL;
  SlabGeo g;
  const int rc = light_geo(&g, b, c, h, w, dtype);
  if (rc != MRLA_OK) return rc;
  return launch_light_apply_fwd_nchw(x, o_prev, wv, gate, sc, sh, lam, dp, out, g, d, res, dtype, act,
                                     (hipStream_t)stream);
}

int mrla_light_stats_bwd(const void* dout, const void* x, const void* o_prev, const float* wv, const float* mom,
                         float* bmom, int b, int c, int h, int w, int dtype, int layout, int act, void* stream) {
  if (!dout || !x || !wv || !mom || !bmom || bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout == MRLA_NHWC)
    return launch_light_stats_bwd_nhwc(dout, x, o_prev, wv, mom, bmom, b, c, h, w, dtype, act, (hipStream_t)stream);
  if (layout != MRLA_NCHW) return MRLA_EINVAL;
  SlabGeo g;
  const int rc = light_geo(&g, b, c, h, w, dtype);
  if (rc != MRLA_OK) return rc;
  return launch_light_stats_bwd_nchw(dout, x, o_prev, wv, mom, bmom, g, dtype, act, (hipStream_t)stream);
}

int mrla_tuning_row_ranges(int mode) {
  if (mode < 0 || mode > 2) return MRLA_EINVAL;
  return nhwc_set_row_cut_mode(mode);
}

int mrla_light_mom_splits(int b, int c, int h, int w, int dtype, int layout) {
  if (bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout != MRLA_NHWC && layout != MRLA_NCHW) return MRLA_EINVAL;
  return layout == MRLA_NHWC ? nhwc_mom_ranges(b, c, h, w) : 1;
}

int mrla_light_bmom_splits(int b, int c, int h, int w, int dtype, int layout) {
  if (bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout != MRLA_NHWC && layout != MRLA_NCHW) return MRLA_EINVAL;
  return layout == MRLA_NHWC ? nhwc_bmom_ranges(b, c, h, w) : 1;
}

int mrla_light_bn_bwd(const float* mom, const float* bmom, const float* gate, const float* lam, const float* gamma,
                      const float* dp, const float* save_mean, const float* save_inv, int bn_mode, float* cb,
                      float* cb_lo, float* dgamma, float* dbeta, float* dlam, int b, int c, int hw, int d, void* stream) {
  if (!mom || !bmom || !gate || !cb || b <= 0 || c <= 0 || hw <= 0 || d <= 0 || c % d) return MRLA_EINVAL;
  if (gamma && (!save_mean || !save_inv || !dgamma || !dbeta)) return MRLA_EINVAL;
  if (dlam && !lam) return MRLA_EINVAL;
  return launch_bn_bwd(mom, bmom, gate, lam, gamma, dp, save_mean, save_inv, bn_mode == MRLA_BN_TRAIN, cb, cb_lo, dgamma,
                       dbeta, dlam, b, c, hw, d, (hipStream_t)stream);
}

int mrla_light_gate_bwd(const float* mom, const float* bmom, const float* gate, const float* cb, const float* cb_lo,
                        const float* dp, const float* wq, const float* wk, int ksize, float* dyx, float* dwqk_part, int b, int c,
                        int hw, int d, void* stream) {
  if (!mom || !bmom || !gate || !wq || !wk || !dyx || !dwqk_part || b <= 0 || c <= 0 || hw <= 0 || d <= 0 || c % d ||
      ksize <= 0 || !(ksize & 1))
    return MRLA_EINVAL;
  return launch_gate_bwd(mom, bmom, gate, cb, cb_lo, dp, wq, wk, ksize, dyx, dwqk_part, b, c, hw, d, (hipStream_t)stream);
}

int mrla_light_apply_bwd_pre_sums(int b, int c, int h, int w, int dtype, int layout) {
  if (bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout != MRLA_NHWC && layout != MRLA_NCHW) return MRLA_EINVAL;
  // 16-bit activations only: with fp32 rows the extra row buffer would not fit the 160 KB of LDS beside the others
  return (layout == MRLA_NHWC && c % kWave == 0 && dtype != MRLA_F32) ? 1 : MRLA_EUNSUPPORTED;
}

int mrla_light_apply_bwd(const void* dout, const void* x, const void* o_prev, const float* wv, const float* gate,
                         const float* cb, const float* lam, const float* dp, const float* dyx, void* dx,
                         void* do_prev, float* dwv_part, const void* pre, const float* pre_center, float* pre_tmom, int b,
                         int c, int h, int w, int d, int res, int relu_mask, int dtype, int layout, int act, void* stream) {
  if (!dout || !x || !wv || !gate || !dyx || !dx || !dwv_part || bad_dims(b, c, h, w) || bad_dtype(dtype) || d <= 0 ||
      c % d)
    return MRLA_EINVAL;
  if (o_prev && (!lam || !do_prev)) return MRLA_EINVAL;
  if (relu_mask && (!o_prev || act != MRLA_ACT_NONE)) return MRLA_EINVAL;
  if ((pre == nullptr) != (pre_tmom == nullptr) || (pre_tmom && !relu_mask)) return MRLA_EINVAL;
  if (layout == MRLA_NHWC)
    return launch_light_apply_bwd_nhwc(dout, x, o_prev, wv, gate, cb, lam, dp, dyx, dx, do_prev, dwv_part, pre, pre_center,
                                       pre_tmom, b, c, h, w, d, res, relu_mask, dtype, act, (hipStream_t)stream);
  if (layout != MRLA_NCHW) return MRLA_EINVAL;
  if (pre_tmom) return MRLA_EUNSUPPORTED;
  SlabGeo g;
  const int rc = light_geo(&g, b, c, h, w, dtype);
  if (rc != MRLA_OK) return rc;
  return launch_light_apply_bwd_nchw(dout, x, o_prev, wv, gate, cb, lam, dp, dyx, dx, do_prev, dwv_part, g, d, res,
                                     relu_mask, dtype, act, (hipStream_t)stream);
}

int mrla_light_lean_supported(int b, int c, int h, int w, int dtype, int layout) {
  if (bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout != MRLA_NHWC && layout != MRLA_NCHW) return MRLA_EINVAL;
  return (layout == MRLA_NHWC && light_lean_supported(b, c, h, w, dtype) == 1) ? 1 : 0;
}

int mrla_light_stats_bwd_fused(const void* dout, const void* pre, const float* pre_sc, const float* pre_sh,
                               const void* o_prev, const float* wv, const float* mom, float* bmom, int b, int c, int h, int w,
                               int dtype, int layout, void* stream) {
  if (!dout || !pre || !o_prev || !wv || !mom || !bmom || bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if ((pre_sc == nullptr) != (pre_sh == nullptr)) return MRLA_EINVAL;
  if (mrla_light_lean_supported(b, c, h, w, dtype, layout) != 1) return MRLA_EUNSUPPORTED;
  return launch_light_stats_bwd_lean_wide(dout, pre, o_prev, wv, pre_sc, pre_sh, mom, bmom, b, c, h, w, dtype,
                                          (hipStream_t)stream);
}

int mrla_light_apply_bwd_fused(const void* dout, const void* pre, const float* pre_sc, const float* pre_sh,
                               const void* o_prev, const float* wv, const float* gate, const float* cb, const float* lam,
                               const float* dp, const float* dyx, void* dx, void* do_prev, float* dwv_part,
                               const float* pre_center, float* pre_tmom, int b, int c, int h, int w, int d, int res,
                               int dtype, int layout, void* stream) {
  if (!dout || !pre || !o_prev || !wv || !gate || !lam || !dyx || !dx || !do_prev || !dwv_part || bad_dims(b, c, h, w) ||
      bad_dtype(dtype) || d <= 0 || c % d)
    return MRLA_EINVAL;
  if ((pre_sc == nullptr) != (pre_sh == nullptr)) return MRLA_EINVAL;
  if (mrla_light_lean_supported(b, c, h, w, dtype, layout) != 1) return MRLA_EUNSUPPORTED;
  return launch_light_apply_bwd_lean_wide(dout, pre, o_prev, wv, pre_sc, pre_sh, gate, cb, lam, dp, dyx, dx, do_prev,
                                          dwv_part, pre_center, pre_tmom, b, c, h, w, d, res, dtype, (hipStream_t)stream);
}

static bool bad_ring(int T, int t) { return T <= 0 || t <= 0 || t > T; }

int mrla_base_gate_fwd(const float* mom, const float* wq, const float* wk, int ksize, float* k_ring, float* p_all,
                       float* q, int b, int c, int hw, int d, int T, int t, void* stream) {
  if (!mom || !wq || !wk || !k_ring || !p_all || !q || b <= 0 || c <= 0 || hw <= 0 || d <= 0 || c % d || ksize <= 0 ||
      !(ksize & 1) || bad_ring(T, t))
    return MRLA_EINVAL;
  return launch_base_gate_fwd(mom, wq, wk, ksize, k_ring, p_all, q, b, c, hw, d, T, t, (hipStream_t)stream);
}

int mrla_base_attend_fwd(const void* x, const float* wv, void* v_ring, const float* p_all, void* attn, float* amom,
                         int b, int c, int h, int w, int d, int T, int t, int dtype, int layout, void* stream) {
  if (!v_ring || !p_all || !attn || !amom || bad_dims(b, c, h, w) || bad_dtype(dtype) || d <= 0 || c % d || bad_ring(T, t))
    return MRLA_EINVAL;
  if (layout == MRLA_NHWC)      // slot t-1 was written by mrla_base_pool_value_fwd; x / wv are not read
    return launch_base_attend_fwd_nhwc(v_ring, p_all, attn, amom, b, c, h * w, d, T, t, dtype, (hipStream_t)stream);
  if (layout != MRLA_NCHW || !x || !wv) return MRLA_EINVAL;
  SlabGeo g;
  const int rc = light_geo(&g, b, c, h, w, dtype);
  if (rc != MRLA_OK) return rc;
  return launch_base_attend_fwd(x, wv, v_ring, p_all, attn, amom, g, d, T, t, dtype, (hipStream_t)stream);
}

int mrla_bn_stats_fwd(const float* amom, const float* pivot, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, int bn_mode, float momentum, float eps, float* sc, float* sh,
                      float* save_mean, float* save_inv, int b, int c, int hw, void* stream) {
  if (!amom || !gamma || !beta || !running_mean || !running_var || !sc || !sh || !save_mean || !save_inv || b <= 0 ||
      c <= 0 || hw <= 0 || (bn_mode != MRLA_BN_TRAIN && bn_mode != MRLA_BN_EVAL))
    return MRLA_EINVAL;
  return launch_plain_bn_fwd(amom, gamma, beta, running_mean, running_var, bn_mode == MRLA_BN_TRAIN, momentum, eps, sc,
                             sh, save_mean, save_inv, pivot, b, c, hw, (hipStream_t)stream);
}

int mrla_bn_stats_fwd_rows(const float* rec, const float* gamma, const float* beta, float* running_mean, float* running_var,
                           int bn_mode, float momentum, float eps, float* sc, float* sh, float* save_mean, float* save_inv,
                           int rows, int c, void* stream) {
  if (!rec || !gamma || !beta || !running_mean || !running_var || !sc || !sh || !save_mean || !save_inv || rows <= 0 ||
      c <= 0 || (bn_mode != MRLA_BN_TRAIN && bn_mode != MRLA_BN_EVAL))
    return MRLA_EINVAL;
  return launch_plain_bn_fwd_rec(rec, gamma, beta, running_mean, running_var, bn_mode == MRLA_BN_TRAIN, momentum, eps, sc,
                                 sh, save_mean, save_inv, rows, c, (hipStream_t)stream);
}

int mrla_base_tail_fwd(const void* x, const void* attn, const float* sc, const float* sh, const float* dp, void* out,
                       int b, int c, int h, int w, int dtype, int layout, void* stream) {
  if (!x || !attn || !sc || !sh || !out || bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout == MRLA_NHWC)
    return launch_base_tail_fwd_nhwc(x, attn, sc, sh, dp, out, b, c, h * w, dtype, (hipStream_t)stream);
  if (layout != MRLA_NCHW) return MRLA_EINVAL;
  return launch_base_tail_fwd(x, attn, sc, sh, dp, out, b, c, h * w, dtype, (hipStream_t)stream);
}

int mrla_base_tail_stats_bwd(const void* dout, const void* attn, const float* sc, const float* sh, const float* center,
                             const float* dp, float* tmom, int b, int c, int h, int w, int dtype, int layout, void* stream) {
  if (!dout || !attn || !sc || !sh || !tmom || bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout == MRLA_NHWC)      // tmom then has mrla_bn_moment_rows() rows
    return launch_nhwc_moments(attn, dout, sc, sh, 1, dp, tmom, const_cast<float*>(center), b, c, h * w, dtype, 1,
                               (hipStream_t)stream);
  if (layout != MRLA_NCHW) return MRLA_EINVAL;
  if (center) return MRLA_EUNSUPPORTED;        // (the NCHW slab kernel keeps raw sums)
  SlabGeo g;
  const int rc = light_geo(&g, b, c, h, w, dtype);
  if (rc != MRLA_OK) return rc;
  return launch_base_tail_stats_bwd(dout, attn, sc, sh, dp, tmom, g, dtype, (hipStream_t)stream);
}

int mrla_bn_stats_bwd(const float* tmom, const float* gamma, const float* save_mean, const float* save_inv,
                      int bn_mode, int centered, float* cb, float* dgamma, float* dbeta, int b, int c, int hw, void* stream) {
  if (!tmom || !gamma || !save_mean || !save_inv || !cb || !dgamma || !dbeta || b <= 0 || c <= 0 || hw <= 0 ||
      (bn_mode != MRLA_BN_TRAIN && bn_mode != MRLA_BN_EVAL))
    return MRLA_EINVAL;
  return launch_plain_bn_bwd(tmom, gamma, save_mean, save_inv, bn_mode == MRLA_BN_TRAIN, centered != 0, cb, dgamma, dbeta,
                             b, c, hw, (hipStream_t)stream);
}

int mrla_base_attend_bwd(const void* dout, const void* attn, const float* sc, const float* sh, const float* dp,
                         const float* cb, const void* v_ring, void* da_ring, float* pmom, int b, int c, int h, int w,
                         int T, int t, int dtype, int layout, void* stream) {
  if (!dout || !v_ring || !da_ring || !pmom || bad_dims(b, c, h, w) || bad_dtype(dtype) || bad_ring(T, t))
    return MRLA_EINVAL;
  if (sc && (!attn || !sh || !cb)) return MRLA_EINVAL;
  if (layout == MRLA_NHWC)      // pmom is the [mrla_base_tile_rows(), t, c] partial buffer; see mrla_base_pmom_reduce
    return launch_base_attend_bwd_nhwc(dout, sc ? attn : nullptr, sc, sh, dp, cb, v_ring, da_ring, pmom, b, c, h * w, T, t,
                                       dtype, (hipStream_t)stream);
  if (layout != MRLA_NCHW) return MRLA_EINVAL;
  SlabGeo g;
  const int rc = light_geo(&g, b, c, h, w, dtype);
  if (rc != MRLA_OK) return rc;
  return launch_base_attend_bwd(dout, sc ? attn : nullptr, sc, sh, dp, cb, v_ring, da_ring, pmom, g, T, t, dtype,
                                (hipStream_t)stream);
}

int mrla_base_gate_bwd(const float* mom, const float* pmom, const float* p_all, const float* q, const float* k_ring,
                       float* dk_ring, const float* wq, const float* wk, int ksize, float* dyx, float* dwqk_part,
                       int b, int c, int hw, int d, int T, int t, int first_touch, void* stream) {
  if (!mom || !pmom || !p_all || !q || !k_ring || !dk_ring || !wq || !wk || !dyx || !dwqk_part || b <= 0 || c <= 0 ||
      hw <= 0 || d <= 0 || c % d || ksize <= 0 || !(ksize & 1) || bad_ring(T, t))
    return MRLA_EINVAL;
  return launch_base_gate_bwd(mom, pmom, p_all, q, k_ring, dk_ring, wq, wk, ksize, dyx, dwqk_part, b, c, hw, d, T, t,
                              first_touch, (hipStream_t)stream);
}

int mrla_base_value_bwd(const void* dout, const void* x, const float* wv, const void* da_ring, const float* p_all,
                        const float* dyx, void* dx, float* dwv_part, int b, int c, int h, int w, int d, int T, int t,
                        int Tc, int res, int dtype, int layout, void* stream) {
  if (!dout || !x || !wv || !da_ring || !p_all || !dyx || !dx || !dwv_part || bad_dims(b, c, h, w) ||
      bad_dtype(dtype) || d <= 0 || c % d || bad_ring(T, t) || Tc < t || Tc > T)
    return MRLA_EINVAL;
  if (layout != MRLA_NCHW) return MRLA_EUNSUPPORTED;
  SlabGeo g;
  const int rc = light_geo(&g, b, c, h, w, dtype);
  if (rc != MRLA_OK) return rc;
  return launch_base_value_bwd(dout, x, wv, da_ring, p_all, dyx, dx, dwv_part, g, d, T, t, Tc, res, dtype,
                               (hipStream_t)stream);
}

// ---- channels_last MRLA-base: entry points that exist for the slot-major NHWC rings only ------------------------
int mrla_base_tile_rows(int b, int c, int h, int w, int dtype, int layout) {
  if (bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout == MRLA_NCHW) return b;
  if (layout != MRLA_NHWC) return MRLA_EINVAL;
  if (!base_nhwc_supported(c, dtype)) return MRLA_EUNSUPPORTED;
  return b * base_nhwc_tiles(b, c, h * w, dtype);
}

int mrla_base_pmom_rows(int b, int c, int h, int w, int dtype, int layout) {
  if (bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout == MRLA_NCHW) return b;
  if (layout != MRLA_NHWC) return MRLA_EINVAL;
  if (!base_nhwc_supported(c, dtype)) return MRLA_EUNSUPPORTED;
  return b * base_nhwc_pmom_tiles(b, c, h * w, dtype);
}

int mrla_base_pool_value_fwd(const void* x, const float* pre_sc, const float* pre_sh, const void* identity,
                             const float* wv, float* mom, void* x_out, void* v_slot, int b, int c, int h, int w,
                             int dtype, int layout, void* stream) {
  if (!x || !wv || !mom || !v_slot || bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if ((identity == nullptr) != (x_out == nullptr)) return MRLA_EINVAL;
  if ((pre_sc == nullptr) != (pre_sh == nullptr) || (pre_sc && !identity)) return MRLA_EINVAL;
  if (layout != MRLA_NHWC) return MRLA_EUNSUPPORTED;
  return launch_light_stats_fwd_nhwc(x, identity, wv, mom, x_out, pre_sc, pre_sh, v_slot, b, c, h, w, dtype,
                                     MRLA_ACT_NONE, (hipStream_t)stream);
}

int mrla_base_pmom_reduce(const float* part, float* pmom, int b, int c, int t, int rows, void* stream) {
  if (!part || !pmom || b <= 0 || c <= 0 || t <= 0 || rows <= 0 || rows % b) return MRLA_EINVAL;
  return launch_base_pmom_reduce(part, pmom, b, c, t, rows / b, (hipStream_t)stream);
}

int mrla_base_dv_combine(const void* da_ring, const float* p_all, void* dv, int b, int c, int h, int w, int d, int T,
                         int t, int Tc, int dtype, int layout, void* stream) {
  if (!da_ring || !p_all || !dv || bad_dims(b, c, h, w) || bad_dtype(dtype) || d <= 0 || c % d || bad_ring(T, t) ||
      Tc < t || Tc > T)
    return MRLA_EINVAL;
  if (layout != MRLA_NHWC) return MRLA_EUNSUPPORTED;
  return launch_base_dv_combine_nhwc(da_ring, p_all, dv, b, c, h * w, d, T, t, Tc, dtype, (hipStream_t)stream);
}

int mrla_base_value_bwd_pre_sums(int b, int c, int h, int w, int dtype, int layout) {
  if (bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout != MRLA_NHWC && layout != MRLA_NCHW) return MRLA_EINVAL;
  return (layout == MRLA_NHWC && c % kWave == 0) ? 1 : MRLA_EUNSUPPORTED;
}

int mrla_base_value_bwd_dv(const void* dout, const void* x, const float* wv, const void* dv, const float* dyx, void* dx,
                           float* dwv_part, const void* pre, const float* pre_center, float* pre_tmom, int b, int c, int h,
                           int w, int res, int dtype, int layout, void* stream) {
  if (!dout || !x || !wv || !dv || !dyx || !dx || !dwv_part || bad_dims(b, c, h, w) || bad_dtype(dtype))
    return MRLA_EINVAL;
  if ((pre == nullptr) != (pre_tmom == nullptr) || (pre_tmom && !(res & 2))) return MRLA_EINVAL;
  if (layout != MRLA_NHWC) return MRLA_EUNSUPPORTED;
  if (c % kWave == 0)       // the LDS-DMA row pipeline
    return launch_base_value_bwd_wide(dout, x, wv, dv, dyx, dx, dwv_part, pre, pre_center, pre_tmom, b, c, h, w, res, dtype,
                                      (hipStream_t)stream);
  return launch_base_value_bwd_nhwc(dout, x, wv, dv, dyx, dx, dwv_part, pre, pre_center, pre_tmom, b, c, h, w, res, dtype,
                                    (hipStream_t)stream);
}

static int token_side(int n) {
  if (n < 2) return 0;
  int s = 1;
  while ((s + 1) * (s + 1) <= n - 1) ++s;
  return s * s == n - 1 ? s : 0;
}

int mrla_token_norm_pool(const void* x, const void* o_prev, const float* lnx_w, const float* lnx_b, float eps,
                         float* stats, float* mom, int b, int n, int c, int dtype, void* stream) {
  if (!x || !lnx_w || !lnx_b || !stats || !mom || b <= 0 || c <= 0 || bad_dtype(dtype) || !token_side(n))
    return MRLA_EINVAL;
  return launch_token_norm_pool(x, o_prev, lnx_w, lnx_b, eps, stats, mom, b, n, c, dtype, (hipStream_t)stream);
}

int mrla_token_apply_fwd(const void* x, const void* o_prev, const float* stats, const float* lnx_w,
                         const float* lnx_b, const float* lno_w, const float* lno_b, const float* wv,
                         const float* gate, const float* lam, void* out, int b, int n, int c, int d, int res,
                         int dtype, void* stream) {
  if (!x || !o_prev || !stats || !lnx_w || !lnx_b || !lno_w || !lno_b || !wv || !gate || !lam || !out || b <= 0 ||
      c <= 0 || d <= 0 || c % d || bad_dtype(dtype) || !token_side(n))
    return MRLA_EINVAL;
  return launch_token_apply_fwd(x, o_prev, stats, lnx_w, lnx_b, lno_w, lno_b, wv, gate, lam, out, b, n, c,
                                token_side(n), d, res, dtype, (hipStream_t)stream);
}

int mrla_token_part_rows(int b, int n, int c, int dtype) {
  if (b <= 0 || c <= 0 || bad_dtype(dtype) || !token_side(n)) return MRLA_EINVAL;
  return b * token_bands_bwd(b, c, token_side(n));
}

int mrla_token_apply_bwd(const void* dout, const void* x, const void* o_prev, const float* stats, const float* lnx_w,
                         const float* lnx_b, const float* lno_w, const float* lno_b, const float* wv,
                         const float* gate, const float* lam, float* dxn, float* part, float* bmom, int b, int n, int c,
                         int d, int dtype, void* stream) {
  if (!dout || !x || !o_prev || !stats || !lnx_w || !lnx_b || !lno_w || !lno_b || !wv || !gate || !lam || !dxn ||
      !part || !bmom || b <= 0 || c <= 0 || d <= 0 || c % d || bad_dtype(dtype) || !token_side(n))
    return MRLA_EINVAL;
  return launch_token_apply_bwd(dout, x, o_prev, stats, lnx_w, lnx_b, lno_w, lno_b, wv, gate, lam, dxn, part, bmom, b, n,
                                c, token_side(n), d, dtype, (hipStream_t)stream);
}

int mrla_token_gate_bwd(const float* mom, const float* bmom, const float* gate, const float* wq, const float* wk,
                        int ksize, float* dyx, float* dwqk_part, float* part, int b, int n, int c, int d, int dtype,
                        void* stream) {
  if (!mom || !bmom || !gate || !wq || !wk || !dyx || !dwqk_part || !part || b <= 0 || c <= 0 || d <= 0 || c % d ||
      ksize <= 0 || !(ksize & 1) || bad_dtype(dtype) || !token_side(n))
    return MRLA_EINVAL;
  return launch_gate_bwd(mom, bmom, gate, nullptr, nullptr, nullptr, wq, wk, ksize, dyx, dwqk_part, b, c, n - 1, d,
                         (hipStream_t)stream, part, token_bands_bwd(b, c, token_side(n)));
}

int mrla_token_ln_bwd(const void* dout, const void* x, const void* o_prev, const float* dxn, const float* dyx,
                      const float* stats, const float* lnx_w, const float* lno_w, const float* lam, void* dx,
                      void* do_prev, int b, int n, int c, int res, int dtype, void* stream) {
  if (!dout || !x || !dxn || !stats || !lnx_w || !dx || b <= 0 || c <= 0 || bad_dtype(dtype) || !token_side(n) ||
      (o_prev && (!lno_w || !lam || !do_prev)))
    return MRLA_EINVAL;
  return launch_token_ln_bwd(dout, x, o_prev, dxn, dyx, stats, lnx_w, lno_w, lam, dx, do_prev, b, n, c, res, dtype,
                             (hipStream_t)stream);
}

// ---- MRLA-base on tokens (deit/deit_mrla_base.py:224-243) ------------------------------------------------------------
int mrla_token_base_supported(int b, int n, int c, int dtype) {
  if (b <= 0 || c <= 0 || bad_dtype(dtype) || !token_side(n)) return MRLA_EINVAL;
  return (token_nhwc_applies(c) && base_nhwc_supported(c, dtype)) ? 1 : 0;
}

int mrla_token_base_value_fwd(const void* x, const float* stats, const float* lnx_w, const float* lnx_b, const float* wv,
                              void* v_slot, int b, int n, int c, int dtype, void* stream) {
  if (!x || !stats || !lnx_w || !lnx_b || !wv || !v_slot || b <= 0 || c <= 0 || bad_dtype(dtype) || !token_side(n))
    return MRLA_EINVAL;
  return launch_token_value_fwd_nhwc(x, stats, lnx_w, lnx_b, wv, v_slot, b, n, c, token_side(n), dtype,
                                     (hipStream_t)stream);
}

int mrla_token_base_attend_fwd(const void* v_ring, const float* p_all, const void* x, const float* stats,
                               const float* lnx_w, const float* lnx_b, void* out, float* amom, int b, int n, int c, int d,
                               int T, int t, int dtype, void* stream) {
  if (!v_ring || !p_all || !x || !stats || !lnx_w || !lnx_b || !out || !amom || b <= 0 || c <= 0 || d <= 0 || c % d ||
      bad_dtype(dtype) || !token_side(n) || bad_ring(T, t))
    return MRLA_EINVAL;
  // the map rows of out[b, n, c]: c elements more per image than a dense map, the first one c elements in
  const int rc = launch_base_attend_fwd_nhwc(v_ring, p_all, out, amom, b, c, n - 1, d, T, t, dtype, (hipStream_t)stream, c, c);
  if (rc != MRLA_OK) return rc;
  return launch_token_cls_fwd(x, stats, lnx_w, lnx_b, out, b, n, c, dtype, (hipStream_t)stream);
}

int mrla_token_base_attend_bwd(const void* dout, const void* v_ring, void* da_ring, float* pmom_part, int b, int n, int c,
                               int T, int t, int dtype, void* stream) {
  if (!dout || !v_ring || !da_ring || !pmom_part || b <= 0 || c <= 0 || bad_dtype(dtype) || !token_side(n) || bad_ring(T, t))
    return MRLA_EINVAL;
  return launch_base_attend_bwd_nhwc(dout, nullptr, nullptr, nullptr, nullptr, nullptr, v_ring, da_ring, pmom_part, b, c,
                                     n - 1, T, t, dtype, (hipStream_t)stream, c, c);
}

int mrla_token_base_gate_bwd(const float* mom, const float* pmom, const float* p_all, const float* q, const float* k_ring,
                             float* dk_ring, const float* wq, const float* wk, int ksize, float* dyx, float* dwqk_part,
                             float* part, int b, int n, int c, int d, int T, int t, int first_touch, int dtype,
                             void* stream) {
  if (!mom || !pmom || !p_all || !q || !k_ring || !dk_ring || !wq || !wk || !dyx || !dwqk_part || !part || b <= 0 ||
      c <= 0 || d <= 0 || c % d || ksize <= 0 || !(ksize & 1) || bad_dtype(dtype) || !token_side(n) || bad_ring(T, t))
    return MRLA_EINVAL;
  return launch_base_gate_bwd(mom, pmom, p_all, q, k_ring, dk_ring, wq, wk, ksize, dyx, dwqk_part, b, c, n - 1, d, T, t,
                              first_touch, (hipStream_t)stream, part, token_bands_bwd(b, c, token_side(n)));
}

int mrla_token_base_value_bwd(const void* dout, const void* x, const float* stats, const float* lnx_w, const float* lnx_b,
                              const float* wv, const void* dv, float* dxn, float* part, int b, int n, int c, int dtype,
                              void* stream) {
  if (!dout || !x || !stats || !lnx_w || !lnx_b || !wv || !dv || !dxn || !part || b <= 0 || c <= 0 || bad_dtype(dtype) ||
      !token_side(n))
    return MRLA_EINVAL;
  return launch_token_value_bwd_nhwc(dout, x, stats, lnx_w, lnx_b, wv, dv, dxn, part, b, n, c, token_side(n), dtype,
                                     (hipStream_t)stream);
}

int mrla_bn_plane_moments(const void* x, float* amom, float* pivot, int b, int c, int h, int w, int dtype, int layout,
                          void* stream) {
  if (!x || !amom || bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout == MRLA_NHWC)
    return launch_nhwc_moments(x, nullptr, nullptr, nullptr, 0, nullptr, amom, pivot, b, c, h * w, dtype, 0,
                               (hipStream_t)stream);
  if (layout != MRLA_NCHW) return MRLA_EINVAL;
  return launch_plane_moments(x, amom, pivot, b, c, h * w, dtype, (hipStream_t)stream);
}

int mrla_bn_act_fwd(const void* x, const float* sc, const float* sh, int relu, void* y, int b, int c, int h, int w,
                    int dtype, int layout, void* stream) {
  if (!x || !sc || !sh || !y || bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout == MRLA_NHWC)
    return launch_nhwc_affine(x, nullptr, nullptr, sc, sh, relu, y, b, c, h * w, dtype, 0, (hipStream_t)stream);
  if (layout != MRLA_NCHW) return MRLA_EINVAL;
  return launch_affine_act(x, nullptr, nullptr, sc, sh, relu, y, b, c, h * w, dtype, 0, (hipStream_t)stream);
}

int mrla_bn_plane_dmoments(const void* dy, const void* x, const float* sc, const float* sh, const float* center, int relu,
                           float* tmom, int b, int c, int h, int w, int dtype, int layout, void* stream) {
  if (!dy || !x || !sc || !sh || !tmom || bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout == MRLA_NHWC)
    return launch_nhwc_moments(x, dy, sc, sh, relu, nullptr, tmom, const_cast<float*>(center), b, c, h * w, dtype, 1,
                               (hipStream_t)stream);
  if (layout != MRLA_NCHW) return MRLA_EINVAL;
  return launch_plane_dmoments(dy, x, sc, sh, center, relu, tmom, b, c, h * w, dtype, (hipStream_t)stream);
}

int mrla_bn_act_bwd(const void* dy, const void* x, const float* sc, const float* sh, const float* cb, int relu,
                    void* dx, int b, int c, int h, int w, int dtype, int layout, void* stream) {
  if (!dy || !x || !sc || !sh || !cb || !dx || bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout == MRLA_NHWC)
    return launch_nhwc_affine(x, dy, cb, sc, sh, relu, dx, b, c, h * w, dtype, 1, (hipStream_t)stream);
  if (layout != MRLA_NCHW) return MRLA_EINVAL;
  return launch_affine_act(x, dy, cb, sc, sh, relu, dx, b, c, h * w, dtype, 1, (hipStream_t)stream);
}

int mrla_bn_pool_rows(int b, int c, int h, int w, int dtype, int layout) {
  if (bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout != MRLA_NHWC) return layout == MRLA_NCHW ? MRLA_EUNSUPPORTED : MRLA_EINVAL;
  return bn_pool_rows(b, c, h, w);
}

int mrla_bn_relu_pool_fwd(const void* x, const float* sc, const float* sh, void* out, int b, int c, int h, int w, int dtype,
                          int layout, void* stream) {
  if (!x || !sc || !sh || !out || bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout != MRLA_NHWC) return layout == MRLA_NCHW ? MRLA_EUNSUPPORTED : MRLA_EINVAL;
  return launch_bn_relu_pool_fwd(x, sc, sh, out, b, c, h, w, dtype, (hipStream_t)stream);
}

int mrla_bn_relu_pool_dmoments(const void* dp, const void* x, const float* sc, const float* sh, const float* center,
                               float* tmom, int b, int c, int h, int w, int dtype, int layout, void* stream) {
  if (!dp || !x || !sc || !sh || !tmom || bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout != MRLA_NHWC) return layout == MRLA_NCHW ? MRLA_EUNSUPPORTED : MRLA_EINVAL;
  return launch_bn_relu_pool_dmoments(dp, x, sc, sh, center, tmom, b, c, h, w, dtype, (hipStream_t)stream);
}

int mrla_bn_relu_pool_bwd(const void* dp, const void* x, const float* sc, const float* sh, const float* cb, void* dx, int b,
                          int c, int h, int w, int dtype, int layout, void* stream) {
  if (!dp || !x || !sc || !sh || !cb || !dx || bad_dims(b, c, h, w) || bad_dtype(dtype)) return MRLA_EINVAL;
  if (layout != MRLA_NHWC) return layout == MRLA_NCHW ? MRLA_EUNSUPPORTED : MRLA_EINVAL;
  return launch_bn_relu_pool_bwd(dp, x, sc, sh, cb, dx, b, c, h, w, dtype, (hipStream_t)stream);
}

int mrla_conv1x1_rows(int m, int k, int n, int dtype) {
  if (m <= 0 || k <= 0 || n <= 0 || bad_dtype(dtype)) return MRLA_EINVAL;
  if (dtype != MRLA_BF16) return MRLA_EUNSUPPORTED;
  return conv1x1_rows(m, k, n);
}

int mrla_conv1x1_plan(int m, int k, int n, int dtype, int addend, int* out) {
  if (m <= 0 || k <= 0 || n <= 0 || bad_dtype(dtype) || !out) return MRLA_EINVAL;
  if (dtype != MRLA_BF16) return MRLA_EUNSUPPORTED;
  return conv1x1_plan(m, k, n, addend, out);
}

int mrla_conv1x1_wgrad_plan(int m, int k, int n, int dtype, int* out) {
  if (m <= 0 || k <= 0 || n <= 0 || bad_dtype(dtype) || !out) return MRLA_EINVAL;
  if (dtype != MRLA_BF16) return MRLA_EUNSUPPORTED;
  return conv1x1_wgrad_plan(m, k, n, out);
}

int mrla_conv1x1_fwd(const void* x, const void* w, void* y, float* mom_part, int m, int k, int n, int dtype, void* stream) {
  if (!x || !w || !y || m <= 0 || k <= 0 || n <= 0 || bad_dtype(dtype)) return MRLA_EINVAL;
  if (dtype != MRLA_BF16) return MRLA_EUNSUPPORTED;
  return launch_conv1x1_fwd(x, w, y, mom_part, m, k, n, (hipStream_t)stream);
}

int mrla_conv1x1_add_supported(int m, int k, int n, int dtype) {
  if (m <= 0 || k <= 0 || n <= 0 || bad_dtype(dtype)) return MRLA_EINVAL;
  if (dtype != MRLA_BF16) return MRLA_EUNSUPPORTED;
  return conv1x1_wide_rows(m, k, n) > 0 ? 1 : MRLA_EUNSUPPORTED;
}

int mrla_conv1x1_fwd_add(const void* x, const void* w, const void* addend, void* y, int m, int k, int n, int dtype,
                         void* stream) {
  if (!x || !w || !addend || !y || m <= 0 || k <= 0 || n <= 0 || bad_dtype(dtype)) return MRLA_EINVAL;
  if (dtype != MRLA_BF16) return MRLA_EUNSUPPORTED;
  return launch_conv1x1_wide(x, w, addend, y, nullptr, m, k, n, (hipStream_t)stream);
}

int mrla_conv1x1_wgrad_rows(int m, int k, int n, int dtype) {
  if (m <= 0 || k <= 0 || n <= 0 || bad_dtype(dtype)) return MRLA_EINVAL;
  if (dtype != MRLA_BF16) return MRLA_EUNSUPPORTED;
  return conv1x1_wgrad_rows(m, k, n);
}

int mrla_conv1x1_wgrad(const void* dy, const void* x, float* part, void* dw, int m, int k, int n, int dtype, int dw_dtype,
                       void* stream) {
  if (!dy || !x || !part || !dw || m <= 0 || k <= 0 || n <= 0 || bad_dtype(dtype)) return MRLA_EINVAL;
  if (dw_dtype != MRLA_BF16 && dw_dtype != MRLA_F32) return MRLA_EINVAL;
  if (dtype != MRLA_BF16) return MRLA_EUNSUPPORTED;
  return launch_conv1x1_wgrad(dy, x, part, dw, dw_dtype == MRLA_F32, m, k, n, (hipStream_t)stream);
}

int mrla_weight_bank_refresh(const void* table, int entries, int max_tiles, void* stream) {
  if (!table || entries <= 0 || max_tiles <= 0) return MRLA_EINVAL;
  return launch_weight_bank_refresh((const long long*)table, entries, max_tiles, (hipStream_t)stream);
}

int mrla_reduce_rows(const float* in, float* out, int rows, int n, void* stream) {
  if (!in || !out || rows <= 0 || n <= 0) return MRLA_EINVAL;
  return launch_reduce_rows(in, out, rows, n, (hipStream_t)stream);
}

int mrla_reduce_rows2(const float* in1, float* out1, int rows1, int n1, const float* in2, float* out2, int rows2, int n2,
                      void* stream) {
  if (!in1 || !out1 || !in2 || !out2 || rows1 <= 0 || n1 <= 0 || rows2 <= 0 || n2 <= 0) return MRLA_EINVAL;
  return launch_reduce_rows2(in1, out1, rows1, n1, in2, out2, rows2, n2, (hipStream_t)stream);
}

}  // extern "C"
