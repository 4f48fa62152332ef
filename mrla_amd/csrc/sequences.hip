// Sequence entry points of the C ABI (include/mrla_hip.h, ABI 4): one call issues the static launch sequence of a whole
// tail and direction.  Host code only -- every pass is the per-pass entry point of capi.hip, called in the documented
// order on the caller's stream with the caller's buffers; results are therefore bit-identical to the per-pass calls.
// Reference statements: resnet_mrla_light.py:113-116 (light tail), resnet_mrla_base.py:120-129 (base tail),
// deit_mrla_light.py:194-209,234 (token module), resnet_mrla_light.py:93-102 (BatchNorm call sites).
#include "mrla_kernels.h"

#define MRLA_TRY(call)          \
  do {                          \
    const int rc_ = (call);     \
    if (rc_ != MRLA_OK) return rc_; \
  } while (0)

extern "C" {

int mrla_light_tail_fwd(const void* x, const float* pre_sc, const float* pre_sh, const void* o_prev, const float* wq,
                        const float* wk, int ksize, const float* wv, const float* lam, const float* gamma,
                        const float* beta, float* running_mean, float* running_var, int bn_mode, float momentum, float eps,
                        const float* dp, float* mom, void* x_out, float* gate, float* bnbuf, void* out, int b, int c,
                        int h, int w, int d, int res, int fuse, int dtype, int layout, int act, void* stream) {
  if (bn_mode != MRLA_BN_NONE && !bnbuf) return MRLA_EINVAL;
  if (fuse && ((fuse == 1 && !x_out) || act != MRLA_ACT_NONE)) return MRLA_EINVAL;
  const void* xt = x;
  if (fuse == 2) {                 // x_t is never written: the apply pass re-forms it from (pre = x, o_prev)
    if (mrla_light_lean_supported(b, c, h, w, dtype, layout) != 1) return MRLA_EUNSUPPORTED;
    MRLA_TRY(mrla_light_stats_fwd_fused(x, pre_sc, pre_sh, o_prev, wv, mom, nullptr, b, c, h, w, dtype, layout, stream));
    xt = nullptr;
  } else if (fuse) {
    MRLA_TRY(mrla_light_stats_fwd_fused(x, pre_sc, pre_sh, o_prev, wv, mom, x_out, b, c, h, w, dtype, layout, stream));
    xt = x_out;
  } else {
    MRLA_TRY(mrla_light_stats_fwd(x, o_prev, wv, mom, b, c, h, w, dtype, layout, act, stream));
  }
  MRLA_TRY(mrla_light_gate_fwd(mom, wq, wk, ksize, gate, b, c, h * w, d, stream));
  const float *sc = nullptr, *sh = nullptr;
  if (bn_mode != MRLA_BN_NONE) {
    MRLA_TRY(mrla_light_bn_fwd(mom, gate, lam, gamma, beta, running_mean, running_var, bn_mode, momentum, eps, bnbuf,
                               bnbuf + c, bnbuf + 2 * (size_t)c, bnbuf + 3 * (size_t)c, b, c, h * w, d, stream));
    sc = bnbuf;
    sh = bnbuf + c;
  }
  if (fuse == 2)
    return mrla_light_apply_fwd_fused(x, pre_sc, pre_sh, o_prev, wv, gate, sc, sh, lam, dp, out, b, c, h, w, d, res, dtype,
                                      layout, stream);
  return mrla_light_apply_fwd(xt, o_prev, wv, gate, sc, sh, lam, dp, out, b, c, h, w, d, res, dtype, layout, act, stream);
}

int mrla_light_tail_bwd(const void* dout, const void* x, const void* o_prev, const float* wq, const float* wk, int ksize,
                        const float* wv, const float* lam, const float* gamma, const float* dp, const float* mom,
                        const float* gate, const float* bnbuf, int bn_mode, float* bmom, float* small, float* dyx,
                        float* dwqk_part, float* dwv_part, int rows, void* dx, void* do_prev, const void* pre,
                        const float* pre_sc, const float* pre_sh, const float* pre_center, float* pre_tmom, float* wsum,
                        int b, int c, int h, int w, int d, int res, int relu_mask, int dtype, int layout, int act,
                        void* stream) {
  if (!small || !wsum || rows <= 0 || ksize <= 0) return MRLA_EINVAL;
  const bool has_bn = bn_mode != MRLA_BN_NONE;
  if (has_bn && (!bnbuf || !gamma)) return MRLA_EINVAL;
  const size_t C = (size_t)c;
  float *cb = small, *dgamma = small + 4 * C, *dbeta = small + 5 * C, *dlam = small + 6 * C, *cb_lo = small + 7 * C;
  const bool lean = x == nullptr;      // the forward never wrote x_t: re-formed from (pre, pre_sc, pre_sh, o_prev)
  if (lean && (!pre || !relu_mask || act != MRLA_ACT_NONE)) return MRLA_EINVAL;
  if (lean) MRLA_TRY(mrla_light_stats_bwd_fused(dout, pre, pre_sc, pre_sh, o_prev, wv, mom, bmom, b, c, h, w, dtype, layout, stream));
  else MRLA_TRY(mrla_light_stats_bwd(dout, x, o_prev, wv, mom, bmom, b, c, h, w, dtype, layout, act, stream));
  MRLA_TRY(mrla_light_bn_bwd(mom, bmom, gate, lam, has_bn ? gamma : nullptr, dp, has_bn ? bnbuf + 2 * C : nullptr,
                             has_bn ? bnbuf + 3 * C : nullptr, bn_mode, cb, cb_lo, has_bn ? dgamma : nullptr,
                             has_bn ? dbeta : nullptr, lam ? dlam : nullptr, b, c, h * w, d, stream));
  MRLA_TRY(mrla_light_gate_bwd(mom, bmom, gate, cb, cb_lo, dp, wq, wk, ksize, dyx, dwqk_part, b, c, h * w, d, stream));
  if (lean)
    MRLA_TRY(mrla_light_apply_bwd_fused(dout, pre, pre_sc, pre_sh, o_prev, wv, gate, cb, lam, dp, dyx, dx, do_prev, dwv_part,
                                        pre_center, pre_tmom, b, c, h, w, d, res, dtype, layout, stream));
  else
    MRLA_TRY(mrla_light_apply_bwd(dout, x, o_prev, wv, gate, cb, lam, dp, dyx, dx, do_prev, dwv_part, pre, pre_center,
                                  pre_tmom, b, c, h, w, d, res, relu_mask, dtype, layout, act, stream));
  return mrla_reduce_rows2(dwv_part, wsum, rows, c * 9, dwqk_part, wsum + 9 * C, b, 2 * ksize, stream);
}

int mrla_bn_fwd(const void* x, const float* records, int rec_rows, float* amom, float* pivot, int rows, const float* gamma,
                const float* beta, float* running_mean, float* running_var, int bn_mode, float momentum, float eps,
                float* bnbuf, int relu, void* y, int b, int c, int h, int w, int dtype, int layout, void* stream) {
  if (!bnbuf || (bn_mode != MRLA_BN_TRAIN && bn_mode != MRLA_BN_EVAL)) return MRLA_EINVAL;
  const size_t C = (size_t)c;
  if (records) {
    if (bn_mode != MRLA_BN_TRAIN) return MRLA_EINVAL;
    MRLA_TRY(mrla_bn_stats_fwd_rows(records, gamma, beta, running_mean, running_var, bn_mode, momentum, eps, bnbuf,
                                    bnbuf + C, bnbuf + 2 * C, bnbuf + 3 * C, rec_rows, c, stream));
  } else {
    if (!amom || rows <= 0 || ((long)b * h * w) % rows) return MRLA_EINVAL;
    const bool train = bn_mode == MRLA_BN_TRAIN;
    if (train) MRLA_TRY(mrla_bn_plane_moments(x, amom, pivot, b, c, h, w, dtype, layout, stream));
    MRLA_TRY(mrla_bn_stats_fwd(amom, train ? pivot : nullptr, gamma, beta, running_mean, running_var, bn_mode, momentum,
                               eps, bnbuf, bnbuf + C, bnbuf + 2 * C, bnbuf + 3 * C, rows, c, (int)((long)b * h * w / rows),
                               stream));
  }
  if (!y) return MRLA_OK;
  return mrla_bn_act_fwd(x, bnbuf, bnbuf + C, relu, y, b, c, h, w, dtype, layout, stream);
}

int mrla_bn_bwd(const void* dy, const void* x, const float* gamma, const float* bnbuf, float* tmom, int rows, int have_tmom,
                int bn_mode, int relu, float* small, void* dx, int b, int c, int h, int w, int dtype, int layout,
                void* stream) {
  if (!bnbuf || !tmom || !small || rows <= 0 || ((long)b * h * w) % rows) return MRLA_EINVAL;
  const size_t C = (size_t)c;
  if (!have_tmom)
    MRLA_TRY(mrla_bn_plane_dmoments(dy, x, bnbuf, bnbuf + C, bnbuf + 2 * C, relu, tmom, b, c, h, w, dtype, layout, stream));
  MRLA_TRY(mrla_bn_stats_bwd(tmom, gamma, bnbuf + 2 * C, bnbuf + 3 * C, bn_mode, 1, small, small + 3 * C, small + 4 * C, rows,
                             c, (int)((long)b * h * w / rows), stream));
  return mrla_bn_act_bwd(dy, x, bnbuf, bnbuf + C, small, relu, dx, b, c, h, w, dtype, layout, stream);
}

int mrla_stem_fwd(const void* x, float* amom, float* pivot, int rows, const float* gamma, const float* beta,
                  float* running_mean, float* running_var, int bn_mode, float momentum, float eps, float* bnbuf, void* out,
                  int b, int c, int h, int w, int dtype, int layout, void* stream) {
  if (!bnbuf || !amom || rows <= 0 || ((long)b * h * w) % rows || (bn_mode != MRLA_BN_TRAIN && bn_mode != MRLA_BN_EVAL))
    return MRLA_EINVAL;
  const size_t C = (size_t)c;
  const bool train = bn_mode == MRLA_BN_TRAIN;
  if (train) MRLA_TRY(mrla_bn_plane_moments(x, amom, pivot, b, c, h, w, dtype, layout, stream));
  MRLA_TRY(mrla_bn_stats_fwd(amom, train ? pivot : nullptr, gamma, beta, running_mean, running_var, bn_mode, momentum, eps,
                             bnbuf, bnbuf + C, bnbuf + 2 * C, bnbuf + 3 * C, rows, c, (int)((long)b * h * w / rows), stream));
  return mrla_bn_relu_pool_fwd(x, bnbuf, bnbuf + C, out, b, c, h, w, dtype, layout, stream);
}

int mrla_stem_bwd(const void* dp, const void* x, const float* gamma, const float* bnbuf, float* tmom, int rows, int bn_mode,
                  float* small, void* dx, int b, int c, int h, int w, int dtype, int layout, void* stream) {
  if (!bnbuf || !tmom || !small || rows <= 0 || ((long)b * h * w) % rows) return MRLA_EINVAL;
  const size_t C = (size_t)c;
  MRLA_TRY(mrla_bn_relu_pool_dmoments(dp, x, bnbuf, bnbuf + C, bnbuf + 2 * C, tmom, b, c, h, w, dtype, layout, stream));
  MRLA_TRY(mrla_bn_stats_bwd(tmom, gamma, bnbuf + 2 * C, bnbuf + 3 * C, bn_mode, 1, small, small + 3 * C, small + 4 * C, rows, c,
                             (int)((long)b * h * w / rows), stream));
  if (!dx) return MRLA_OK;
  return mrla_bn_relu_pool_bwd(dp, x, bnbuf, bnbuf + C, small, dx, b, c, h, w, dtype, layout, stream);
}

int mrla_base_layer_fwd(const void* x, const float* pre_sc, const float* pre_sh, const void* identity, const float* wq,
                        const float* wk, int ksize, const float* wv, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, int bn_mode, float momentum, float eps, const float* dp,
                        float* mom, void* x_out, void* v_ring, float* k_ring, float* p_all, float* q, void* attn,
                        float* amom, int arows, float* bnbuf, void* out, int tail, int b, int c, int h, int w, int d, int T,
                        int t, int dtype, void* stream) {
  if (!v_ring || t <= 0 || t > T || arows <= 0 || (tail && (!bnbuf || !out))) return MRLA_EINVAL;
  const size_t C = (size_t)c, slot = (size_t)b * h * w * C * mrla::dtype_size(dtype);
  void* v_slot = (char*)v_ring + (size_t)(t - 1) * slot;
  MRLA_TRY(mrla_base_pool_value_fwd(x, pre_sc, pre_sh, identity, wv, mom, x_out, v_slot, b, c, h, w, dtype, MRLA_NHWC, stream));
  MRLA_TRY(mrla_base_gate_fwd(mom, wq, wk, ksize, k_ring, p_all, q, b, c, h * w, d, T, t, stream));
  MRLA_TRY(mrla_base_attend_fwd(nullptr, wv, v_ring, p_all, attn, amom, b, c, h, w, d, T, t, dtype, MRLA_NHWC, stream));
  if (!tail) return MRLA_OK;
  if (((long)b * h * w) % arows) return MRLA_EINVAL;
  MRLA_TRY(mrla_bn_stats_fwd(amom, nullptr, gamma, beta, running_mean, running_var, bn_mode, momentum, eps, bnbuf, bnbuf + C,
                             bnbuf + 2 * C, bnbuf + 3 * C, arows, c, (int)((long)b * h * w / arows), stream));
  return mrla_base_tail_fwd(identity ? x_out : x, attn, bnbuf, bnbuf + C, dp, out, b, c, h, w, dtype, MRLA_NHWC, stream);
}

int mrla_base_layer_bwd(const void* dout, const void* x, const void* attn, const float* wq, const float* wk, int ksize,
                        const float* wv, const float* gamma, const float* dp, const float* mom, const float* q,
                        const float* bnbuf, int bn_mode, const void* v_ring, void* da_ring, const float* k_ring,
                        float* dk_ring, const float* p_all, float* tmom, int trows, float* small, float* ppart, int prows,
                        float* pmom, float* dyx, float* dwqk_part, void* dv, float* dwv_part, int rows, void* dx,
                        const void* pre, const float* pre_center, float* pre_tmom, float* wsum, int tail, int first_touch,
                        int res, int b, int c, int h, int w, int d, int T, int t, int Tc, int dtype, void* stream) {
  if (!wsum || rows <= 0 || prows <= 0 || ksize <= 0) return MRLA_EINVAL;
  const size_t C = (size_t)c;
  const float *sc = nullptr, *sh = nullptr, *cb = nullptr;
  if (tail) {
    if (!bnbuf || !small || !tmom || trows <= 0 || ((long)b * h * w) % trows) return MRLA_EINVAL;
    sc = bnbuf;
    sh = bnbuf + C;
    cb = small;
    MRLA_TRY(mrla_base_tail_stats_bwd(dout, attn, sc, sh, bnbuf + 2 * C, dp, tmom, b, c, h, w, dtype, MRLA_NHWC, stream));
    MRLA_TRY(mrla_bn_stats_bwd(tmom, gamma, bnbuf + 2 * C, bnbuf + 3 * C, bn_mode, 1, small, small + 3 * C, small + 4 * C,
                               trows, c, (int)((long)b * h * w / trows), stream));
  }
  MRLA_TRY(mrla_base_attend_bwd(dout, attn, sc, sh, dp, cb, v_ring, da_ring, ppart, b, c, h, w, T, t, dtype, MRLA_NHWC, stream));
  MRLA_TRY(mrla_base_pmom_reduce(ppart, pmom, b, c, t, prows, stream));
  MRLA_TRY(mrla_base_gate_bwd(mom, pmom, p_all, q, k_ring, dk_ring, wq, wk, ksize, dyx, dwqk_part, b, c, h * w, d, T, t,
                              first_touch, stream));
  MRLA_TRY(mrla_base_dv_combine(da_ring, p_all, dv, b, c, h, w, d, T, t, Tc, dtype, MRLA_NHWC, stream));
  MRLA_TRY(mrla_base_value_bwd_dv(dout, x, wv, dv, dyx, dx, dwv_part, pre, pre_center, pre_tmom, b, c, h, w, res, dtype,
                                  MRLA_NHWC, stream));
  return mrla_reduce_rows2(dwv_part, wsum, rows, c * 9, dwqk_part, wsum + 9 * C, b, 2 * ksize, stream);
}

int mrla_token_light_fwd(const void* x, const void* o_prev, const float* lnx_w, const float* lnx_b, const float* lno_w,
                         const float* lno_b, const float* wq, const float* wk, int ksize, const float* wv,
                         const float* lam, float eps, float* stats, float* mom, float* gate, void* out, int b, int n, int c,
                         int d, int res, int dtype, void* stream) {
  MRLA_TRY(mrla_token_norm_pool(x, o_prev, lnx_w, lnx_b, eps, stats, mom, b, n, c, dtype, stream));
  MRLA_TRY(mrla_light_gate_fwd(mom, wq, wk, ksize, gate, b, c, n - 1, d, stream));
  return mrla_token_apply_fwd(x, o_prev, stats, lnx_w, lnx_b, lno_w, lno_b, wv, gate, lam, out, b, n, c, d, res, dtype,
                              stream);
}

int mrla_token_light_bwd(const void* dout, const void* x, const void* o_prev, const float* stats, const float* lnx_w,
                         const float* lnx_b, const float* lno_w, const float* lno_b, const float* wq, const float* wk,
                         int ksize, const float* wv, const float* gate, const float* lam, const float* mom, float* dxn,
                         float* part, int prow, float* bmom, float* dyx, float* dwqk_part, void* dx, void* do_prev,
                         float* sums, int b, int n, int c, int d, int res, int dtype, void* stream) {
  if (!sums || prow <= 0 || ksize <= 0) return MRLA_EINVAL;
  MRLA_TRY(mrla_token_apply_bwd(dout, x, o_prev, stats, lnx_w, lnx_b, lno_w, lno_b, wv, gate, lam, dxn, part, bmom, b, n, c,
                                d, dtype, stream));
  MRLA_TRY(mrla_token_gate_bwd(mom, bmom, gate, wq, wk, ksize, dyx, dwqk_part, part, b, n, c, d, dtype, stream));
  MRLA_TRY(mrla_token_ln_bwd(dout, x, o_prev, dxn, dyx, stats, lnx_w, lno_w, lam, dx, do_prev, b, n, c, res, dtype, stream));
  const size_t np = (size_t)c * MRLA_TOKEN_PARTIALS;
  return mrla_reduce_rows2(part, sums, prow, (int)np, dwqk_part, sums + np, b, 2 * ksize, stream);
}

}  // extern "C"
