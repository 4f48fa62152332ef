/* libmrla_hip -- C ABI of the MI355X (gfx950) MRLA hot path.
 *
 * The reference (joyfang1106/MRLA) has no FFI layer: the boundary its hot path sits behind is the
 * PyTorch nn.Module API.  These entry points are what a binding for that path attaches to; each one
 * names the reference statements it replaces (paths relative to the reference repo root).  The host
 * side that mirrors the reference's modules on top of them is `mrla_amd/` (ctypes).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; the library never allocates, frees,
 *     copies to the host or synchronises; every launch goes to the `stream` argument (a hipStream_t);
 *   - re-entrant and thread-safe (no global state), graph-capturable;
 *   - return value: MRLA_OK (0) or a negative MRLA_E* code; nothing throws or aborts;
 *   - activations are `dtype` (MRLA_F32 / MRLA_BF16 / MRLA_F16) in `layout`; parameters, gates, moments
 *     and every reduction are float32;
 *   - nullable arguments are marked [opt].
 *
 * Math (b = image, c = channel, g = c / d = head, V = act(dwconv3x3(x, wv)), s = 1/sqrt(d)):
 *   y = mean_hw x ; q = corr1d(y, wq) ; k = corr1d(y, wk)                      (zero padded, along c)
 *   light:  a[b,g] = sigmoid(s * sum_{c in g} q*k) ;  m = a*V + lam*o_prev
 *           out = res*x + dp[b] * BN(m)      (BN: batch statistics / running statistics / identity)
 */
#ifndef MRLA_HIP_H_
#define MRLA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRLA_OK 0
#define MRLA_EINVAL (-1)        /* bad argument (null pointer, non-positive dim, c % d != 0, ...) */
#define MRLA_EUNSUPPORTED (-2)  /* shape outside what the kernels handle (e.g. plane too large for LDS) */
#define MRLA_EHIP (-3)          /* the HIP runtime reported an error at launch */

enum { MRLA_F32 = 0, MRLA_BF16 = 1, MRLA_F16 = 2 };
enum { MRLA_NCHW = 0, MRLA_NHWC = 1 };   /* NHWC = torch.channels_last: x[b,h,w,c], dims are still passed as b,c,h,w */
enum { MRLA_ACT_NONE = 0, MRLA_ACT_GELU = 1 };
enum { MRLA_BN_NONE = 0, MRLA_BN_TRAIN = 1, MRLA_BN_EVAL = 2 };

/* Moment record sizes (floats per (image, channel)). */
#define MRLA_FWD_MOMENTS 8 /* sum x, sum V', sum o', sum V'^2, sum V'*o', sum o'^2, pV, po  with V' = V - pV, o' = o - po:
                              the second moments are taken about per-plane pivots (samples of V / o; 0 where a producer
                              does not shift) so that they stay well conditioned when |mean| >> sigma */
#define MRLA_BWD_MOMENTS 3 /* sum dOut, sum dOut*V', sum dOut*o'  (V' = V - pV, o' = o - pO: about the forward record's pivots) */
#define MRLA_GEMM_MOMENTS 4 /* the 1x1-convolution GEMM's epilogue, per (workgroup row, out-channel): sum (y - p),
                               sum (y - p)^2, the pivot p (the row's first rounded output of the channel), pixel count n */

/* ABI version of this header.  Bumped whenever an existing entry point changes its arguments or a record changes size:
 *   1 -> 2: mrla_light_bn_bwd / mrla_light_gate_bwd gained cb_lo, mrla_bn_stats_fwd / mrla_bn_plane_moments gained pivot,
 *           MRLA_FWD_MOMENTS grew from 6 to 8 floats, mrla_light_apply_bwd gained pre / pre_tmom,
 *           mrla_conv1x1_wgrad gained dw_dtype, mrla_conv1x1_fwd's mom_part became
 *           [rows, n, MRLA_GEMM_MOMENTS] records (was [rows, n, 2] raw sums) read by the new mrla_bn_stats_fwd_rows,
 *           mrla_bn_plane_dmoments / mrla_bn_relu_pool_dmoments / mrla_base_tail_stats_bwd gained center and
 *           mrla_bn_stats_bwd gained centered (BatchNorm-backward sums about the saved mean), mrla_light_stats_bwd gained
 *           mom (MRLA_BWD_MOMENTS are about the forward record's pivots);
 *           mrla_conv1x1_plan, mrla_conv1x1_wgrad_plan, mrla_light_apply_bwd_pre_sums, mrla_reduce_rows2 and
 *           mrla_weight_bank_refresh were added;  the token backward became one pass: mrla_token_stats_bwd was removed,
 *           mrla_token_apply_bwd writes bmom and no longer takes dyx, MRLA_TOKEN_PARTIALS grew from 14 to 15,
 *           mrla_token_gate_bwd was added and mrla_token_ln_bwd gained dyx;  the mrla_token_base_* entry points were added.
 *   2 -> 3: mrla_base_value_bwd_dv gained pre / pre_center / pre_tmom (bn3's backward sums folded into the MRLA-base value
 *           backward, as mrla_light_apply_bwd has them); mrla_base_value_bwd_pre_sums was added.
 *   3 -> 4: the "sequence" entry points were added (mrla_light_tail_fwd / _bwd, mrla_bn_fwd / _bwd, mrla_base_layer_fwd /
 *           _bwd, mrla_token_light_fwd / _bwd, mrla_stem_fwd / _bwd: one call issues the static launch sequence of a whole tail and direction;
 *           the per-pass entry points are unchanged).
 *   4 -> 5: the training tail without a stored x_t (13N instead of 15N elements per block and step): mrla_light_lean_supported,
 *           mrla_light_stats_bwd_fused and mrla_light_apply_bwd_fused were added, mrla_light_stats_fwd_fused takes x_out = NULL,
 *           mrla_light_tail_fwd takes fuse = 2 and mrla_light_tail_bwd gained pre_sc / pre_sh (x = NULL selects that form);
 *           few, large images (detection batches): the backward passes spread an image's column strips over more workgroups --
 *           mrla_light_wgrad_rows counts the extra partial rows, mrla_light_bmom_splits was added (bmom of
 *           mrla_light_stats_bwd* / mrla_light_tail_bwd is [bmom_splits, b, c, MRLA_BWD_MOMENTS], folded into bmom[0] by the
 *           pass) and so was mrla_light_mom_splits (mom of mrla_light_stats_fwd / _fused / mrla_light_tail_fwd is
 *           [mom_splits, b, c, MRLA_FWD_MOMENTS], folded into mom[0]).
 *           Where the strips do not fill the chip either, the ROWS of an image are cut into ranges as well (a range re-fetches
 *           its two halo rows; the backward apply pass re-computes one row of dU): the same three queries count those
 *           ranges too, nothing else changes for a caller.  mrla_tuning_row_ranges() was added.
 * A consumer compares mrla_abi_version() (what the loaded library was built from) against this constant before its
 * first call. */
#define MRLA_ABI_VERSION 5
int mrla_abi_version(void);

/* Number of rows of the `dwv_part` scratch that mrla_light_apply_bwd writes for this problem
 * (each row is [c, 9] floats); negative on error. */
int mrla_light_wgrad_rows(int b, int c, int h, int w, int dtype, int layout);

/* ---- MRLA-light, pass 1 of 2 (forward) ----------------------------------------------------------
 * mom[b, c, 6] <- per-plane moments of x, V = act(dwconv(x)), o_prev.
 * Replaces: avg_pool + Wv conv of mrla_light_module.py:56,61 (deit_mrla_light.py:161,166-167) as the
 * producer of everything the gate and the train-mode bn_mrla statistics (resnet_mrla_light.py:116)
 * need.  o_prev [opt]: block input `identity` (resnet_mrla_light.py:110-111). */
/* mom is [mrla_light_mom_splits(), b, c, MRLA_FWD_MOMENTS] floats (ABI 5): 1 at the classification batches; for few, large
 * images the statistics passes spread an image's column strips over that many workgroup ranges, each leaving its record in
 * mom[z], and fold them into mom[0] -- the [b, c, MRLA_FWD_MOMENTS] block every other entry point reads. */
int mrla_light_mom_splits(int b, int c, int h, int w, int dtype, int layout);
/* Row ranges of the channels_last row pipeline (see ABI 5 above): mode 0 = cut an image's rows where a launch would leave most
 * CUs without a workgroup and the tensor has >= 8 Mi elements (default), 1 = never, 2 = wherever an image has >= 16 rows
 * (the parity tests drive small shapes through the cut kernels with it).  Process-wide; returns the previous mode (or
 * MRLA_EINVAL).  Set it BEFORE sizing buffers with mrla_light_mom_splits / mrla_light_bmom_splits / mrla_light_wgrad_rows:
 * the passes read the mode again when they are launched. */
int mrla_tuning_row_ranges(int mode);
int mrla_light_stats_fwd(const void* x, const void* o_prev, const float* wv /*[c,3,3]*/, float* mom,
                         int b, int c, int h, int w, int dtype, int layout, int act, void* stream);

/* Fused producer form: `pre` is the block's pre-activation (bn3 output).  x_t = relu(pre + o_prev) is formed on
 * the fly, written to x_out (it is what the apply pass and the backward read) and the moments are taken of it.
 * Replaces `out += identity; out = self.relu(out)` (resnet_mrla_light.py:113-114) on top of the statistics pass.
 * pre_sc / pre_sh [opt, c floats each]: the affine of the BatchNorm in front (bn3, resnet_mrla_light.py:102) when the
 * caller defers its elementwise pass: x_t = relu((pre_sc*pre + pre_sh) + o_prev), the affine result rounded to the
 * storage type as the stand-alone pass would have stored it; `pre` is then conv3's raw output.
 * x_out == NULL (only where mrla_light_lean_supported() == 1): x_t is formed for the statistics and NOT written -- the caller
 * re-forms it in every later pass (mrla_light_apply_fwd_fused, mrla_light_stats_bwd_fused, mrla_light_apply_bwd_fused). */
int mrla_light_stats_fwd_fused(const void* pre, const float* pre_sc, const float* pre_sh, const void* o_prev,
                               const float* wv, float* mom, void* x_out, int b, int c, int h, int w, int dtype,
                               int layout, void* stream);

/* ---- inference form of the fused block tail (MRLA_NHWC, no gradients asked for): 5N instead of 6N elements ----
 * x_t = relu((pre_sc*pre + pre_sh) + o_prev) is never written: mrla_light_pool_fused takes its pooled sums (slot 0 of
 * mom[b,c,6]; part = workspace of mrla_bn_moment_rows() x c x 2 floats), mrla_light_gate_fwd / mrla_light_bn_fwd
 * (MRLA_BN_EVAL) follow as usual, and mrla_light_apply_fwd_fused re-forms x_t from pre and o_prev while it applies
 * out = res*x_t + dp*(sc*(a*dwconv3x3(x_t) + lam*o_prev) + sh).  Same reference lines as the two passes above. */
int mrla_light_pool_fused(const void* pre, const float* pre_sc, const float* pre_sh, const void* o_prev, float* part,
                          float* mom, int b, int c, int h, int w, int dtype, int layout, void* stream);
int mrla_light_apply_fwd_fused(const void* pre, const float* pre_sc, const float* pre_sh, const void* o_prev,
                               const float* wv, const float* gate, const float* sc, const float* sh, const float* lam,
                               const float* dp, void* out, int b, int c, int h, int w, int d, int res, int dtype,
                               int layout, void* stream);

/* ---- gate: a[b, g] ------------------------------------------------------------------------------
 * Replaces Wq/Wk Conv1d, the per-head einsum and the sigmoid of mrla_light_module.py:59-60,67,70. */
int mrla_light_gate_fwd(const float* mom, const float* wq, const float* wk, int ksize, float* gate /*[b, c/d]*/,
                        int b, int c, int hw, int d, void* stream);

/* ---- BatchNorm statistics of m = a*V + lam*o_prev in closed form from the moments ---------------
 * Replaces the statistics half of nn.BatchNorm2d `bn_mrla` (resnet_mrla_light.py:85,116):
 *   bn_mode TRAIN: batch mean / biased var; running_mean/var updated in place (unbiased var, momentum);
 *   bn_mode EVAL : running statistics.
 * Outputs sc[c] = gamma*inv_std, sh[c] = beta - sc*mean, and save_mean / save_inv for backward.
 * lam [opt] (null = no o_prev term). */
int mrla_light_bn_fwd(const float* mom, const float* gate, const float* lam, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, int bn_mode, float momentum, float eps, float* sc,
                      float* sh, float* save_mean, float* save_inv, int b, int c, int hw, int d, void* stream);

/* ---- MRLA-light, pass 2 of 2 (forward) ----------------------------------------------------------
 * out = res*x + dp[b]*( sc[c]*(a[b,g]*V + lam[c]*o_prev) + sh[c] )
 * Replaces: Wv conv + broadcast mul (mrla_light_module.py:61,71-72), lambda_t*o_{t-1} + add
 * (resnet_mrla_light.py:42), the normalisation half of bn_mrla, DropPath (utils/drop.py:17-24) and
 * the residual add (resnet_mrla_light.py:116).  sc, sh, lam, dp, o_prev are [opt] (null = 1, 0, -, 1, -);
 * with all of them null and res = 0 this is exactly mrla_light_layer.forward. */
int mrla_light_apply_fwd(const void* x, const void* o_prev, const float* wv, const float* gate, const float* sc,
                         const float* sh, const float* lam, const float* dp, void* out, int b, int c, int h, int w,
                         int d, int res, int dtype, int layout, int act, void* stream);

/* ---- backward pass 1 of 2: bmom[splits, b, c, 3] ---------------------------------------------------
 * The reductions autograd performs in MulBackward / ExpandBackward / NativeBatchNormBackward.
 * splits = mrla_light_bmom_splits(): 1 at the classification batches; for few, large images (a detection batch: 2 x 256 x
 * 200 x 336) the pass spreads an image's column strips (and rows) over `splits` workgroup ranges, each leaving its own partial
 * record, and folds them -- in range order, in double -- into bmom[0]: the [b, c, 3] block mrla_light_bn_bwd /
 * mrla_light_gate_bwd read (the same convention as mom).
 * mom: the forward record of the same block (mrla_light_stats_fwd*): the sums over dOut*V and dOut*o are taken about its
 * pivots (pV, pO), so that what survives the cancellations of the BatchNorm backward keeps fp32 accuracy when
 * |mean| >> sigma; mrla_light_bn_bwd / mrla_light_gate_bwd (given the same mom) undo the shift in double. */
int mrla_light_bmom_splits(int b, int c, int h, int w, int dtype, int layout);
int mrla_light_stats_bwd(const void* dout, const void* x, const void* o_prev, const float* wv, const float* mom,
                         float* bmom, int b, int c, int h, int w, int dtype, int layout, int act, void* stream);

/* ---- BatchNorm backward constants + dgamma, dbeta, dlambda ---------------------------------------
 * cb[c, 4] = (e, f, G, H) such that dm = e*dp[b]*dOut + f*a[b,g]*V + G*o_prev + H.
 * cb_lo[c, 4] [opt, OUTPUT]: the float remainders of the four constants (they are computed in double; the closed-form
 * gate gradient of mrla_light_gate_bwd cancels large terms and wants them beyond fp32 when |mean| >> sigma).
 * gamma [opt]: null = no BatchNorm (e = 1, f = G = H = 0; dgamma/dbeta untouched).
 * lam, dp, dlam [opt]. */
int mrla_light_bn_bwd(const float* mom, const float* bmom, const float* gate, const float* lam, const float* gamma,
                      const float* dp, const float* save_mean, const float* save_inv, int bn_mode, float* cb,
                      float* cb_lo, float* dgamma, float* dbeta, float* dlam, int b, int c, int hw, int d,
                      void* stream);

/* ---- gate backward -------------------------------------------------------------------------------
 * dyx[b, c] = (gradient wrt the pooled descriptor y) / hw ; dwqk_part[b, 2*ksize] = per-image partial
 * sums of dWq (first ksize) and dWk.  cb, cb_lo, dp [opt]. */
int mrla_light_gate_bwd(const float* mom, const float* bmom, const float* gate, const float* cb, const float* cb_lo,
                        const float* dp,
                        const float* wq, const float* wk, int ksize, float* dyx, float* dwqk_part, int b, int c,
                        int hw, int d, void* stream);

/* ---- backward pass 2 of 2 ------------------------------------------------------------------------
 * dx = res*dOut + dwconv^T(a*dm*act'(U)) + dyx ;  do_prev = lam*dm ;  dwv_part[rows, c, 9] partial sums
 * of dWv over groups of images (rows = mrla_light_wgrad_rows()).  cb, lam, dp, o_prev, do_prev [opt].
 * relu_mask != 0 (fused producer, x_t = relu(pre + o_prev)): dx <- [x_t > 0]*dx is the gradient wrt `pre` and
 * do_prev <- lam*dm + [x_t > 0]*dx the total gradient wrt o_prev (ReLU + shortcut-add backward folded in).
 * pre, pre_tmom [opt, both or neither; relu_mask only]: the caller deferred the BatchNorm in front of the fused producer
 * (bn3, resnet_mrla_light.py:101-102: x_t = relu(bn3(pre) + o_prev)) and its backward needs, per channel, sum(dpre) and
 * sum(dpre * pre) with dpre = the dx written here.  Given `pre` (conv3's raw output), the kernel takes both sums on the
 * way: pre_tmom[rows, c, 2] = (sum dpre, sum dpre * (pre - pre_center[c])) with rows = mrla_light_wgrad_rows(), in the
 * layout mrla_bn_stats_bwd reads (pre_center [opt]: that BatchNorm's saved batch mean -> pass centered = 1 there)
 * (mrla_bn_plane_dmoments's separate pass over (dpre, pre) is then not needed).  mrla_light_apply_bwd_pre_sums says
 * whether the kernels of this shape / layout can (1) or not (MRLA_EUNSUPPORTED). */
int mrla_light_apply_bwd_pre_sums(int b, int c, int h, int w, int dtype, int layout);
int mrla_light_apply_bwd(const void* dout, const void* x, const void* o_prev, const float* wv, const float* gate,
                         const float* cb, const float* lam, const float* dp, const float* dyx, void* dx,
                         void* do_prev, float* dwv_part, const void* pre, const float* pre_center, float* pre_tmom, int b,
                         int c, int h, int w, int d, int res, int relu_mask, int dtype, int layout, int act, void* stream);

/* ---- the training tail WITHOUT a stored x_t (ABI 5; MRLA_NHWC, c % 64 == 0, 16-bit activations) -------------------------
 * x_t = relu((pre_sc*pre + pre_sh) + o_prev) (resnet_mrla_light.py:101-114) is what the MRLA branch convolves
 * (mrla_light_module.py:61).  Every pass of the tail reads o_prev anyway, and the backward apply pass reads `pre` for bn3's
 * sums, so x_t can be re-formed from rows that are in LDS already -- by the formula of mrla_light_stats_fwd_fused, hence
 * bit-identical -- instead of being written once and read three times: 13N elements per block and training step, not 15N
 *   forward : mrla_light_stats_fwd_fused(x_out = NULL) [2N]  ->  gate, bn  ->  mrla_light_apply_fwd_fused [3N]
 *   backward: mrla_light_stats_bwd_fused [3N]  ->  bn_bwd, gate_bwd  ->  mrla_light_apply_bwd_fused [5N = section 8(d)'s count].
 * mrla_light_lean_supported: 1 where these kernels exist for the shape, else 0.  The two backward entry points take the
 * arguments of mrla_light_stats_bwd / mrla_light_apply_bwd (relu_mask = 1, act = none) with (pre, pre_sc, pre_sh) in the
 * place of x; pre_tmom [opt] as there (pre itself is always given here). */
int mrla_light_lean_supported(int b, int c, int h, int w, int dtype, int layout);
int mrla_light_stats_bwd_fused(const void* dout, const void* pre, const float* pre_sc, const float* pre_sh,
                               const void* o_prev, const float* wv, const float* mom, float* bmom, int b, int c, int h, int w,
                               int dtype, int layout, void* stream);
int mrla_light_apply_bwd_fused(const void* dout, const void* pre, const float* pre_sc, const float* pre_sh,
                               const void* o_prev, const float* wv, const float* gate, const float* cb, const float* lam,
                               const float* dp, const float* dyx, void* dx, void* do_prev, float* dwv_part,
                               const float* pre_center, float* pre_tmom, int b, int c, int h, int w, int d, int res,
                               int dtype, int layout, void* stream);

/* =====================================================================================================
 * MRLA-base: softmax over the depth of a stage (resnet/models/modules/mrla_base_module.py:54-89 and the
 * block tail resnet/models/resnet_mrla_base.py:120-129).  The K/V history of a stage lives in caller-owned
 * ring buffers that replace the reference's torch.cat growth (mrla_base_module.py:69-70) and the einops
 * rearrange copies (:76-77,83):
 *   v_ring  [b, T, c, h, w] activation dtype    k_ring [b, T, c] float32      p_all [b, c/d, T, T] float32
 *   da_ring [b, T, c, h, w] activation dtype    dk_ring [b, T, c] float32
 * `t` (1-based) is the history length at this layer = the slot (t-1) it appends; T is the ring capacity.
 * The pooled descriptor comes from mrla_light_stats_fwd(x, o_prev = NULL) (slot 0 of `mom`).
 * ===================================================================================================== */

/* q_t, k_t (-> k_ring slot t-1), p_all row t-1 = softmax_j(<q_t, k_j>/sqrt(d)), j < t.
 * Replaces mrla_base_module.py:61-62 (Wq, Wk), :69 (cat K), :76,:79,:82 (rearrange, einsum, softmax). */
int mrla_base_gate_fwd(const float* mom, const float* wq, const float* wk, int ksize, float* k_ring, float* p_all,
                       float* q /*[b,c]*/, int b, int c, int hw, int d, int T, int t, void* stream);

/* v_t = dwconv3x3(x) -> v_ring slot t-1;  attn = sum_{j<t} p[b,g,j] * v_j;  amom[b,c,2] = (sum attn, sum attn^2).
 * Replaces mrla_base_module.py:63 (Wv), :70 (cat V), :77,:83 (rearrange copies), :86-87 (einsum, reshape). */
int mrla_base_attend_fwd(const void* x, const float* wv, void* v_ring, const float* p_all, void* attn, float* amom,
                         int b, int c, int h, int w, int d, int T, int t, int dtype, int layout, void* stream);

/* BatchNorm statistics from per-(image, channel) (sum, sum of squares): sc, sh, save_mean, save_inv; running
 * statistics updated in place in TRAIN mode.  Replaces the statistics half of bn_mrla (resnet_mrla_base.py:125).
 * pivot [opt, c]: the sums are of (x - pivot[c]) (as mrla_bn_plane_moments produces them when given a pivot buffer): the
 * one-pass variance E[d^2] - E[d]^2 then stays well conditioned for channels with |mean| >> sigma. */
int mrla_bn_stats_fwd(const float* amom, const float* pivot, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, int bn_mode, float momentum, float eps, float* sc, float* sh,
                      float* save_mean, float* save_inv, int b, int c, int hw, void* stream);

/* The same from moment records rec[rows, c, MRLA_GEMM_MOMENTS] (pivot and pixel count per row: mrla_conv1x1_fwd's
 * mom_part): rows are merged by re-basing them onto one pivot, in double. */
int mrla_bn_stats_fwd_rows(const float* rec, const float* gamma, const float* beta, float* running_mean, float* running_var,
                           int bn_mode, float momentum, float eps, float* sc, float* sh, float* save_mean, float* save_inv,
                           int rows, int c, void* stream);

/* out = x + dp[b] * relu(sc[c]*attn + sh[c]).  Replaces the normalisation half of bn_mrla, the ReLU, DropPath and
 * the residual add of resnet_mrla_base.py:125-127.  dp [opt]. */
int mrla_base_tail_fwd(const void* x, const void* attn, const float* sc, const float* sh, const float* dp, void* out,
                       int b, int c, int h, int w, int dtype, int layout, void* stream);

/* tmom[b,c,2] = (sum dz, sum dz*(attn - center[c])), dz = dp[b]*dOut*[sc*attn + sh > 0].
 * center [opt, c floats; MRLA_NHWC only]: the batch mean bn_mrla saved -- the second sum then needs no cancelling
 * subtraction in mrla_bn_stats_bwd (centered = 1). */
int mrla_base_tail_stats_bwd(const void* dout, const void* attn, const float* sc, const float* sh, const float* center,
                             const float* dp, float* tmom, int b, int c, int h, int w, int dtype, int layout, void* stream);

/* cb[c,3] = (e, f, h) with d attn = e*dz + f*attn + h; dgamma, dbeta.
 * centered != 0: tmom's second sum is already sum dz*(x - save_mean[c]) (the producer was given `center` = save_mean):
 * dgamma = save_inv * that sum, with nothing to cancel when |mean| >> sigma. */
int mrla_bn_stats_bwd(const float* tmom, const float* gamma, const float* save_mean, const float* save_inv,
                      int bn_mode, int centered, float* cb, float* dgamma, float* dbeta, int b, int c, int hw, void* stream);

/* dA_t -> da_ring slot t-1; pmom[b,c,t] = sum_hw dA_t * v_j (j < t).  sc == NULL: no tail, dA_t = dOut
 * (attn, sh, dp, cb then unused) -- the backward of a bare mrla_base_layer. */
int mrla_base_attend_bwd(const void* dout, const void* attn, const float* sc, const float* sh, const float* dp,
                         const float* cb, const void* v_ring, void* da_ring, float* pmom, int b, int c, int h, int w,
                         int T, int t, int dtype, int layout, void* stream);

/* softmax backward; dq_t; dk_ring[:, j] += dlogit_j * q_t (j < t; first_touch != 0: overwrite instead of add, for the
 * first backward call of a stage); dyx[b,c] = (gradient wrt pooled y)/hw; dwqk_part[b, 2*ksize].
 * c % 4 == 0 (16-byte accesses of the fp32 rings), MRLA_EUNSUPPORTED otherwise. */
int mrla_base_gate_bwd(const float* mom, const float* pmom, const float* p_all, const float* q, const float* k_ring,
                       float* dk_ring, const float* wq, const float* wk, int ksize, float* dyx, float* dwqk_part,
                       int b, int c, int hw, int d, int T, int t, int first_touch, void* stream);

/* dV_t = sum_{t'=t..Tc} p_all[b,g,t'-1,t-1] * dA_t';  dx = res*dOut + dwconv3x3^T(dV_t) + dyx;  dwv_part[rows, c, 9]
 * (rows = mrla_light_wgrad_rows()).  Tc = number of layers of the stage that ran forward.
 * res bit 0: add dOut (block residual); bit 1: multiply by [x > 0] (x_t = relu(pre + identity) was formed by
 * mrla_light_stats_fwd_fused, so dx is the gradient wrt the pre-activation and wrt the identity alike). */
int mrla_base_value_bwd(const void* dout, const void* x, const float* wv, const void* da_ring, const float* p_all,
                        const float* dyx, void* dx, float* dwv_part, int b, int c, int h, int w, int d, int T, int t,
                        int Tc, int res, int dtype, int layout, void* stream);

/* ---- channels_last (MRLA_NHWC) MRLA-base --------------------------------------------------------------------
 * The rings are SLOT-MAJOR for this layout: v_ring / da_ring [T, b, h, w, c] (a slot is an ordinary NHWC tensor).
 * Call sequence per layer (same reference lines as above):
 *   forward : mrla_base_pool_value_fwd -> mrla_base_gate_fwd -> mrla_base_attend_fwd (x, wv ignored; amom has
 *             mrla_base_tile_rows() rows) -> mrla_bn_stats_fwd(rows, hw*b/rows) -> mrla_base_tail_fwd
 *   backward: mrla_base_tail_stats_bwd (tmom: mrla_bn_moment_rows() rows) -> mrla_bn_stats_bwd -> mrla_base_attend_bwd
 *             (pmom argument = partial buffer [mrla_base_pmom_rows(), t, c]) -> mrla_base_pmom_reduce ->
 *             mrla_base_gate_bwd -> mrla_base_dv_combine -> mrla_base_value_bwd_dv
 * Supported when c % 64 == 0 and 256 % (c / (16 / sizeof(dtype))) == 0 (c a power of two up to 2048 for 16-bit types);
 * mrla_base_tile_rows returns MRLA_EUNSUPPORTED otherwise and the caller keeps the stage in MRLA_NCHW. */
int mrla_base_tile_rows(int b, int c, int h, int w, int dtype, int layout);
int mrla_base_pmom_rows(int b, int c, int h, int w, int dtype, int layout);

/* mom[b,c,6] <- pooling moments of x_t (slot 0 = sum x_t);  v_slot <- V_t = dwconv3x3(x_t) (ring slot t-1).
 * identity [opt]: x is the pre-activation, x_t = relu(x + identity) is formed here and also written to x_out;
 * pre_sc / pre_sh [opt, with identity]: deferred bn3 affine as in mrla_light_stats_fwd_fused.
 * Replaces avg_pool + Wv of mrla_base_module.py:57-58,63 and resnet_mrla_base.py:120-121. */
int mrla_base_pool_value_fwd(const void* x, const float* pre_sc, const float* pre_sh, const void* identity,
                             const float* wv, float* mom, void* x_out, void* v_slot, int b, int c, int h, int w,
                             int dtype, int layout, void* stream);

/* pmom[b, c, t] <- sum over the rows/b tiles of an image of part[rows, t, c]. */
int mrla_base_pmom_reduce(const float* part, float* pmom, int b, int c, int t, int rows, void* stream);

/* dv[b,h,w,c] (activation dtype, as autograd stores the gradient of a 16-bit V) <- dV_t =
 * sum_{t'=t..Tc} p_all[b,g,t'-1,t-1] * dA_t', accumulated in fp32. */
int mrla_base_dv_combine(const void* da_ring, const float* p_all, void* dv, int b, int c, int h, int w, int d, int T,
                         int t, int Tc, int dtype, int layout, void* stream);

/* dx = [x > 0 if res bit 1] * ((res bit 0) * dOut + dwconv3x3^T(dv) + dyx);  dwv_part[rows, c, 9]
 * (rows = mrla_light_wgrad_rows()).
 * pre, pre_tmom [opt, both or neither; res bit 1 only]: the caller deferred the BatchNorm in front of the fused producer
 * (bn3 of resnet_mrla_base.py:103-104; x_t = relu(bn3(conv3) + identity), :120-122) -- `pre` is conv3's raw output and dx
 * is dpre, the gradient of that BatchNorm's output, which exists only here; the pass takes the BatchNorm backward's two
 * sums on the way: pre_tmom[rows, c, 2] = (sum dpre, sum dpre * (pre - pre_center[c])) of the STORED (rounded) dpre, in
 * the layout mrla_bn_stats_bwd reads (pre_center [opt]: the saved batch mean -> centered = 1 there), so
 * mrla_bn_plane_dmoments' separate 2N pass is not needed.  mrla_base_value_bwd_pre_sums says whether a shape has this
 * form (1) or not (MRLA_EUNSUPPORTED: c % 64 != 0, NCHW). */
int mrla_base_value_bwd_pre_sums(int b, int c, int h, int w, int dtype, int layout);
int mrla_base_value_bwd_dv(const void* dout, const void* x, const float* wv, const void* dv, const float* dyx, void* dx,
                           float* dwv_part, const void* pre, const float* pre_center, float* pre_tmom, int b, int c, int h,
                           int w, int res, int dtype, int layout, void* stream);

/* =====================================================================================================
 * MRLA-light on token sequences (DeiT): x[b, n, c], n = 1 + side*side, channels contiguous.
 * Reference: deit/deit_mrla_light.py:157-180 (mrlal_layer, GELU on V), :194-209 (mrlal_module: two LayerNorms,
 * cls split, token<->map permutes, lambda_t, cat), :234 (block residual).  The gate itself is computed with
 * mrla_light_gate_fwd / mrla_token_gate_bwd on the `mom` / `bmom` records written here (hw = n - 1).
 *   stats[b, n, 4] = (mean_x, rstd_x, mean_o, rstd_o) per token;   params: lnx_* / lno_* = LayerNorm weight, bias.
 * ===================================================================================================== */
#define MRLA_TOKEN_PARTIALS 15 /* per (image, channel): dWv[9], dlambda, dlnx_w, dlnx_b, dlno_w, dlno_b, sum of xhat */

/* LayerNorm statistics of x and o_prev; mom[b,c,0] = (n-1) * mean_{i>=1} LN_x(x)[b,i,c], other slots 0.
 * Replaces normx / normo statistics (deit_mrla_light.py:195-196) and avg_pool (:161).  o_prev may be NULL (MRLA-base on
 * tokens has no o_{t-1}): its two statistics slots then repeat x's. */
int mrla_token_norm_pool(const void* x, const void* o_prev, const float* lnx_w, const float* lnx_b, float eps,
                         float* stats, float* mom, int b, int n, int c, int dtype, void* stream);

/* out[b,0] = res*x + LN_x(x);  out[b,i>=1] = res*x + a*gelu(dwconv3x3(LN_x(x) map)) + lam*LN_o(o_prev).
 * Replaces deit_mrla_light.py:195-207 (normalisation halves, split, reshape/permute, Wv, GELU, gate multiply,
 * flatten/permute, lambda_t term, cat) and the residual add of :234. */
int mrla_token_apply_fwd(const void* x, const void* o_prev, const float* stats, const float* lnx_w,
                         const float* lnx_b, const float* lno_w, const float* lno_b, const float* wv,
                         const float* gate, const float* lam, void* out, int b, int n, int c, int d, int res,
                         int dtype, void* stream);

/* Backward (autograd of the lines above), ONE pass over the map + the per-token LayerNorm backward:
 *   mrla_token_apply_bwd -> mrla_token_gate_bwd -> mrla_token_ln_bwd -> (sum `part` over its rows)
 * The gradient dy of the pooled descriptor is a per-(image, channel) constant on the map rows of dxn and is known only
 * once the gate backward has seen bmom; it is folded in afterwards instead of costing a second pass over (dOut, x):
 *   mrla_token_apply_bwd: dxn[b,n,c] (float32) = gradient wrt LN_x(x) WITHOUT dy (cls row included);
 *       bmom[b,c,1] = sum_i dOut * gelu(U) (slots 0, 2 zeroed);  part[rows,c,15] parameter-gradient partials (without
 *       dy's share; slot 14 = sum over the map of xhat), rows = mrla_token_part_rows(b, n, c, dtype) (>= b).
 *   mrla_token_gate_bwd: mrla_light_gate_bwd on (mom, bmom) -> dyx[b,c] = dy/hw, dwqk_part[b, 2*ksize]; completes
 *       part[b,c,10] += dyx * part[.,c,14], part[b,c,11] += dy (the LN_x weight / bias partials).
 *   mrla_token_ln_bwd: dx = LN_x^T(dxn + dyx on map tokens) + res*dOut;  do_prev = LN_o^T(lam*dOut) on map tokens, 0 on
 *       the cls row.  dyx may be NULL (dxn already complete); o_prev may be NULL (then lno_w, lam, do_prev are not used and
 *       only dx is written). */
int mrla_token_part_rows(int b, int n, int c, int dtype);
int mrla_token_apply_bwd(const void* dout, const void* x, const void* o_prev, const float* stats, const float* lnx_w,
                         const float* lnx_b, const float* lno_w, const float* lno_b, const float* wv,
                         const float* gate, const float* lam, float* dxn, float* part, float* bmom, int b, int n, int c,
                         int d, int dtype, void* stream);
int mrla_token_gate_bwd(const float* mom, const float* bmom, const float* gate, const float* wq, const float* wk,
                        int ksize, float* dyx, float* dwqk_part, float* part, int b, int n, int c, int d, int dtype,
                        void* stream);
int mrla_token_ln_bwd(const void* dout, const void* x, const void* o_prev, const float* dxn, const float* dyx,
                      const float* stats, const float* lnx_w, const float* lno_w, const float* lam, void* dx,
                      void* do_prev, int b, int n, int c, int res, int dtype, void* stream);

/* ---- MRLA-base on token sequences (DeiT): deit/deit_mrla_base.py:204-243 (mrlab_module) -------------------------
 * xt = normx(x); the 14x14 map tokens of xt go through mrla_base_layer (:120-201, the same math as the ResNet one) and
 * come back behind the cls token of xt (:236-241).  LN_x(x) is never materialised and there is no token <-> map copy and
 * no cat: the value map goes straight into the stage's slot-major NHWC ring ([T, b, side, side, c], a dense image per
 * slot), the attend kernel writes the map rows of out[b, n, c] in place, the cls kernel the first row.
 *   forward : mrla_token_norm_pool (o_prev = NULL) -> mrla_token_base_value_fwd (V_t -> v_ring slot t-1) ->
 *             mrla_base_gate_fwd (hw = n - 1) -> mrla_token_base_attend_fwd
 *   backward: mrla_token_base_attend_bwd (dA_t = the map rows of dOut -> da_ring slot t-1; pmom partials, rows =
 *             mrla_base_pmom_rows(b, c, side, side, dtype, MRLA_NHWC)) -> mrla_base_pmom_reduce ->
 *             mrla_token_base_gate_bwd (= mrla_base_gate_bwd + the completion of `part` with dy, as mrla_token_gate_bwd) ->
 *             mrla_base_dv_combine (dense dV_t) -> mrla_token_base_value_bwd (dxn without dy, `part`) ->
 *             mrla_token_ln_bwd (o_prev = NULL: only dx) -> (sum `part` over its rows)
 * `part` is the [mrla_token_part_rows(), c, MRLA_TOKEN_PARTIALS] record of the light path (slots 0-8 dWv, 10 / 11
 * dlnx_w / dlnx_b, 14 sum of xhat; the others zero).  amom: scratch [mrla_base_tile_rows(b, c, side, side, dtype,
 * MRLA_NHWC), c, 2].  mrla_token_base_supported: 1 when these entry points take (n, c, dtype) (c % 64 == 0), else 0. */
int mrla_token_base_supported(int b, int n, int c, int dtype);
int mrla_token_base_value_fwd(const void* x, const float* stats, const float* lnx_w, const float* lnx_b, const float* wv,
                              void* v_slot, int b, int n, int c, int dtype, void* stream);
int mrla_token_base_attend_fwd(const void* v_ring, const float* p_all, const void* x, const float* stats,
                               const float* lnx_w, const float* lnx_b, void* out, float* amom, int b, int n, int c, int d,
                               int T, int t, int dtype, void* stream);
int mrla_token_base_attend_bwd(const void* dout, const void* v_ring, void* da_ring, float* pmom_part, int b, int n, int c,
                               int T, int t, int dtype, void* stream);
int mrla_token_base_gate_bwd(const float* mom, const float* pmom, const float* p_all, const float* q, const float* k_ring,
                             float* dk_ring, const float* wq, const float* wk, int ksize, float* dyx, float* dwqk_part,
                             float* part, int b, int n, int c, int d, int T, int t, int first_touch, int dtype,
                             void* stream);
int mrla_token_base_value_bwd(const void* dout, const void* x, const float* stats, const float* lnx_w, const float* lnx_b,
                              const float* wv, const void* dv, float* dxn, float* part, int b, int n, int c, int dtype,
                              void* stream);

/* =====================================================================================================
 * Fused BatchNorm2d (+ReLU) in front of the MRLA tail (reference call sites
 * resnet/models/resnet_mrla_light.py:93-102: `bn1/relu`, `bn2/relu`, `bn3`, downsample BN, stem `bn1/relu`).
 *   forward : mrla_bn_plane_moments -> mrla_bn_stats_fwd -> mrla_bn_act_fwd      y  = relu?(sc*x + sh)
 *   backward: mrla_bn_plane_dmoments -> mrla_bn_stats_bwd -> mrla_bn_act_bwd     dx = e*dz + f*x + h, dz = dy*[y>0]
 * ===================================================================================================== */
/* Rows of the partial-moment buffers written by mrla_bn_plane_moments / _dmoments: b for NCHW, b*nsplit for NHWC
 * (pixels of an image are split over nsplit workgroups).  Pass (rows, b*h*w/rows) as (b, hw) to mrla_bn_stats_*. */
int mrla_bn_moment_rows(int b, int c, int h, int w, int layout);
/* pivot [opt, c floats, OUTPUT]: when given, the kernel records p[c] = x[0, c, 0, 0] there and accumulates sums of
 * (x - p[c]) and (x - p[c])^2 instead of raw moments (hand the same buffer to mrla_bn_stats_fwd). */
int mrla_bn_plane_moments(const void* x, float* amom /*[rows,c,2]: sum x, sum x^2*/, float* pivot, int b, int c, int h, int w,
                          int dtype, int layout, void* stream);
int mrla_bn_act_fwd(const void* x, const float* sc, const float* sh, int relu, void* y, int b, int c, int h, int w,
                    int dtype, int layout, void* stream);
/* center [opt, c floats]: the saved batch mean -> tmom's second sum is sum dz*(x - center) (see mrla_bn_stats_bwd). */
int mrla_bn_plane_dmoments(const void* dy, const void* x, const float* sc, const float* sh, const float* center, int relu,
                           float* tmom /*[b,c,2]: sum dz, sum dz*(x - center)*/, int b, int c, int h, int w, int dtype,
                           int layout, void* stream);
int mrla_bn_act_bwd(const void* dy, const void* x, const float* sc, const float* sh, const float* cb /*[c,3]*/, int relu,
                    void* dx, int b, int c, int h, int w, int dtype, int layout, void* stream);

/* The stem tail maxpool3x3/s2/p1(relu(bn1(x))) (resnet/models/resnet_mrla_light.py:220-222; nn.MaxPool2d(kernel_size=3,
 * stride=2, padding=1) after bn1/relu) without the full-size BatchNorm+ReLU tensor: MRLA_NHWC, c % 64 == 0.
 *   forward : mrla_bn_plane_moments -> mrla_bn_stats_fwd -> mrla_bn_relu_pool_fwd   out[b,c,ho,wo], ho = (h-1)/2+1
 *   backward: mrla_bn_relu_pool_dmoments -> mrla_bn_stats_bwd -> mrla_bn_relu_pool_bwd
 *             dz = dP[window] at the window's first maximum (ATen's rule, on the values rounded to dtype) when that
 *             maximum is > 0;  tmom[rows,c,2] = (sum dz, sum dz*x) with rows = mrla_bn_pool_rows();  dx = e*dz + f*x + h.
 * mrla_bn_pool_rows returns MRLA_EUNSUPPORTED for MRLA_NCHW or c % 64 != 0: the caller keeps bn_act + its own pooling. */
int mrla_bn_pool_rows(int b, int c, int h, int w, int dtype, int layout);
int mrla_bn_relu_pool_fwd(const void* x, const float* sc, const float* sh, void* out, int b, int c, int h, int w, int dtype,
                          int layout, void* stream);
int mrla_bn_relu_pool_dmoments(const void* dp, const void* x, const float* sc, const float* sh, const float* center /*[opt]*/,
                               float* tmom, int b, int c, int h, int w, int dtype, int layout, void* stream);
int mrla_bn_relu_pool_bwd(const void* dp, const void* x, const float* sc, const float* sh, const float* cb /*[c,3]*/,
                          void* dx, int b, int c, int h, int w, int dtype, int layout, void* stream);

/* =====================================================================================================
 * The 1x1 stride-1 convolutions in front of those BatchNorms as an MFMA GEMM whose epilogue takes the BatchNorm
 * statistics (first half; reference call sites resnet/models/resnet_mrla_light.py:93-94
 * `conv1 -> bn1` and :100-101 `conv3 -> bn3`, both nn.Conv2d(kernel_size=1, stride=1, bias=False)).
 *   y[m, n] = sum_k x[m, k] * w[n, k]        x: channels_last activation viewed as [m = b*h*w, k = c_in] (bf16),
 *                                            w: the conv weight [n = c_out, k] (bf16), y: [m, n] (bf16, fp32 accumulate)
 *   mom_part[row, n, 0..3] [opt] = moment record (MRLA_GEMM_MOMENTS) over the row's pixels of the ROUNDED outputs --
 *   the statistics mrla_bn_plane_moments would read back from y, taken about a per-row pivot so that the one-pass
 *   variance stays well conditioned when |mean| >> sigma; hand it to mrla_bn_stats_fwd_rows.
 * Three kernel families behind it: k in {64, 128, 256} with n % 64 == 0 (weights resident on chip; one record row per
 * workgroup row), and k >= 512 with k % 32 == 0, n % 128 == 0 (both operands streamed; one record row per pixel tile).
 * mom_part may be NULL (no statistics wanted: the input-gradient use, inference).
 * MRLA_EUNSUPPORTED for other shapes and for dtypes other than MRLA_BF16: the caller keeps its stock convolution there.
 * (The input gradient dX = dY * W is the same entry point with w^T.) */
int mrla_conv1x1_rows(int m, int k, int n, int dtype);      /* rows of mom_part (> 0: every supported shape writes
                                                               records), or a negative code (unsupported shape) */
/* How the launch of this problem is laid out (host-side query, `out` is a HOST array of 4 ints): out[0] = 32-pixel blocks
 * one workgroup (n % 256 == 0) / one pixel-wave (narrow outputs) walks, out[1] = depth in blocks of its LDS ring / register
 * prefetch, out[2] = workgroups, out[3] = mrla_conv1x1_rows().  addend != 0: the mrla_conv1x1_fwd_add form.  Lets a
 * test prove that a case runs the software pipeline in steady state (out[0] >> out[1]). */
int mrla_conv1x1_plan(int m, int k, int n, int dtype, int addend, int* out);
int mrla_conv1x1_fwd(const void* x, const void* w, void* y, float* mom_part, int m, int k, int n, int dtype, void* stream);

/* y = x * w^T + addend in one pass (fp32 sum, one rounding).  Used for the input gradient of the bottleneck's conv1
 * (resnet_mrla_light.py:93) with x = dY, w = W^T and addend = the gradient arriving at the block input through the
 * shortcut (:110-114), which autograd would otherwise add in a separate elementwise pass.  addend: [m, n] like y; it may
 * alias y.  Only the wide form (n % 256 == 0, k in {64, 128, 256}): mrla_conv1x1_add_supported returns 1 or
 * MRLA_EUNSUPPORTED. */
int mrla_conv1x1_add_supported(int m, int k, int n, int dtype);
int mrla_conv1x1_fwd_add(const void* x, const void* w, const void* addend, void* y, int m, int k, int n, int dtype,
                         void* stream);

/* Weight gradient of the same convolution (the backward of the reference's nn.Conv2d(kernel_size=1) call sites above,
 * which the reference leaves to cuDNN):   dw[n, k] = sum_m dy[m, n] * x[m, k]
 *   dy: [m, n] and x: [m, k] channels_last activations (bf16), dw: [n, k] (bf16, fp32 accumulation),
 *   part: fp32 workspace [rows, n, k] with rows = mrla_conv1x1_wgrad_rows(m, k, n, dtype): the per-workgroup partial
 *   tiles of the split over m, summed in a fixed order by a second kernel (no atomics, no memset of dw).
 * MRLA_EUNSUPPORTED unless n % 64 == 0, k % 64 == 0, dtype MRLA_BF16 and m * max(n, k) * 2 < 2^31. */
int mrla_conv1x1_wgrad_rows(int m, int k, int n, int dtype);      /* rows of part (> 0), or a negative code */
/* Host-side query, `out` is a HOST array of 6 ints: 32-pixel chunks per workgroup, LDS stages, tile n, tile k,
 * splits (= mrla_conv1x1_wgrad_rows()), output tiles. */
int mrla_conv1x1_wgrad_plan(int m, int k, int n, int dtype, int* out);
/* dw_dtype: MRLA_BF16 (the autocast copy's gradient, as the stock backward produces it) or MRLA_F32 (the fp32 master
 * weight's gradient directly: the sum over the partial tiles is fp32 anyway, and the cast kernel autograd would append
 * disappears). */
int mrla_conv1x1_wgrad(const void* dy, const void* x, float* part, void* dw, int m, int k, int n, int dtype, int dw_dtype,
                       void* stream);

/* The bf16 working copies of every fp32 convolution weight the GEMMs above multiply with, refreshed in ONE launch per
 * training step instead of one autocast cast kernel per convolution and forward (and one transposing copy per input
 * gradient): replaces what torch.autocast does in front of nn.Conv2d (resnet/train.py runs fp32; the bf16 configuration of
 * BASELINE.json casts each weight per forward).
 *   table: DEVICE array of `entries` x 4 int64: { src fp32 [n, k] pointer, dst bf16 [n, k] pointer,
 *          dst_t bf16 [k, n] pointer or 0, (n << 32) | k };  n % 64 == 0, k % 64 == 0.
 *   max_tiles: the largest (n / 64) * (k / 64) over the entries. */
int mrla_weight_bank_refresh(const void* table, int entries, int max_tiles, void* stream);

/* out[n] = sum over rows of in[rows, n] (fixed order, double accumulation). */
int mrla_reduce_rows(const float* in, float* out, int rows, int n, void* stream);
/* Two such sums in one launch (a block's dWv partials and its dWq / dWk partials). */
int mrla_reduce_rows2(const float* in1, float* out1, int rows1, int n1, const float* in2, float* out2, int rows2, int n2,
                      void* stream);


/* =====================================================================================================
 * Sequence entry points (ABI 4): ONE call issues the static launch sequence of a whole tail and direction on `stream`,
 * in the order the per-pass entry points above document, with the same arguments and the same caller-owned buffers
 * (nothing is allocated; the intermediate records are arguments because the backward reads them).  For a host whose
 * per-call cost matters (the reference's host is Python: ~4 us per call through ctypes, ~500 calls per resnet50_mrlal
 * step).  Results are bit-identical to the per-pass calls: they ARE those calls.  The first failing pass's code is
 * returned and nothing after it is launched.
 * ===================================================================================================== */

/* Light block tail, forward (resnet_mrla_light.py:113-116 / mrla_light_module.py:52-74):
 *   fuse == 0: mrla_light_stats_fwd(x, ...)                      fuse == 1: mrla_light_stats_fwd_fused(x = pre, ..., x_out)
 *   -> mrla_light_gate_fwd -> [bn_mode != MRLA_BN_NONE: mrla_light_bn_fwd] -> mrla_light_apply_fwd (on x_out when fused).
 *   fuse == 2 (x_out = NULL; mrla_light_lean_supported): mrla_light_stats_fwd_fused(x = pre, ..., NULL) -> gate -> bn ->
 *   mrla_light_apply_fwd_fused(pre, ...): x_t is never written.
 * bnbuf [opt unless BN]: [4, c] floats = sc | sh | save_mean | save_inv (rows 0, 1 feed the apply pass). */
int mrla_light_tail_fwd(const void* x, const float* pre_sc, const float* pre_sh, const void* o_prev, const float* wq,
                        const float* wk, int ksize, const float* wv, const float* lam, const float* gamma,
                        const float* beta, float* running_mean, float* running_var, int bn_mode, float momentum, float eps,
                        const float* dp, float* mom, void* x_out, float* gate, float* bnbuf, void* out, int b, int c,
                        int h, int w, int d, int res, int fuse, int dtype, int layout, int act, void* stream);

/* Light block tail, backward:
 *   mrla_light_stats_bwd -> mrla_light_bn_bwd -> mrla_light_gate_bwd -> mrla_light_apply_bwd -> mrla_reduce_rows2.
 * x == NULL (the forward ran with fuse = 2): mrla_light_stats_bwd_fused / mrla_light_apply_bwd_fused on (pre, pre_sc, pre_sh)
 * instead; pre_sc / pre_sh are read in that form only.
 * small: [11, c] floats = cb[c,4] | dgamma | dbeta | dlam | cb_lo[c,4] (the layout mrla_amd/functional.py uses);
 * wsum: [c*9 + 2*ksize] floats = dWv | dWq | dWk; rows = mrla_light_wgrad_rows(). */
int mrla_light_tail_bwd(const void* dout, const void* x, const void* o_prev, const float* wq, const float* wk, int ksize,
                        const float* wv, const float* lam, const float* gamma, const float* dp, const float* mom,
                        const float* gate, const float* bnbuf, int bn_mode, float* bmom, float* small, float* dyx,
                        float* dwqk_part, float* dwv_part, int rows, void* dx, void* do_prev, const void* pre,
                        const float* pre_sc, const float* pre_sh, const float* pre_center, float* pre_tmom, float* wsum,
                        int b, int c, int h, int w, int d, int res, int relu_mask, int dtype, int layout, int act,
                        void* stream);

/* BatchNorm2d(+ReLU), forward:  [records == NULL and TRAIN: mrla_bn_plane_moments(x, amom, pivot)] ->
 *   records != NULL: mrla_bn_stats_fwd_rows(records, ..., rec_rows)   else: mrla_bn_stats_fwd(amom, pivot, ..., rows)
 *   -> [y != NULL: mrla_bn_act_fwd].   y == NULL: the deferred form (statistics only; the consumer applies bnbuf[0:2]).
 * amom [rows, c, 2] and pivot [c]: scratch (unused with records; pivot unused in EVAL); rows = mrla_bn_moment_rows(). */
int mrla_bn_fwd(const void* x, const float* records, int rec_rows, float* amom, float* pivot, int rows, const float* gamma,
                const float* beta, float* running_mean, float* running_var, int bn_mode, float momentum, float eps,
                float* bnbuf, int relu, void* y, int b, int c, int h, int w, int dtype, int layout, void* stream);

/* BatchNorm2d(+ReLU), backward:  [have_tmom == 0: mrla_bn_plane_dmoments(dy, x, ..., center = save_mean, tmom)] ->
 *   mrla_bn_stats_bwd(tmom, ..., centered = 1, rows) -> mrla_bn_act_bwd.
 * have_tmom != 0: tmom[rows, c, 2] was taken by the producer of dy (mrla_light_apply_bwd's pre_tmom).
 * small: [5, c] floats = cb[c,3] | dgamma | dbeta. */
int mrla_bn_bwd(const void* dy, const void* x, const float* gamma, const float* bnbuf, float* tmom, int rows, int have_tmom,
                int bn_mode, int relu, float* small, void* dx, int b, int c, int h, int w, int dtype, int layout,
                void* stream);

/* The stem tail maxpool3x3/s2/p1(relu(bn1(x))) (resnet_mrla_light.py:220-222), forward:
 *   [TRAIN: mrla_bn_plane_moments(x, amom, pivot)] -> mrla_bn_stats_fwd(amom, pivot, ..., rows) -> mrla_bn_relu_pool_fwd;
 * backward: mrla_bn_relu_pool_dmoments(center = save_mean) -> mrla_bn_stats_bwd(centered = 1, rows) ->
 *   [dx != NULL: mrla_bn_relu_pool_bwd].   rows: mrla_bn_moment_rows() forward, mrla_bn_pool_rows() backward; small: [5, c]. */
int mrla_stem_fwd(const void* x, float* amom, float* pivot, int rows, const float* gamma, const float* beta,
                  float* running_mean, float* running_var, int bn_mode, float momentum, float eps, float* bnbuf, void* out,
                  int b, int c, int h, int w, int dtype, int layout, void* stream);
int mrla_stem_bwd(const void* dp, const void* x, const float* gamma, const float* bnbuf, float* tmom, int rows, int bn_mode,
                  float* small, void* dx, int b, int c, int h, int w, int dtype, int layout, void* stream);

/* MRLA-base layer + tail on a channels_last stage (MRLA_NHWC rings), forward -- the sequence documented at
 * mrla_base_tile_rows:  mrla_base_pool_value_fwd -> mrla_base_gate_fwd -> mrla_base_attend_fwd ->
 *   [tail != 0: mrla_bn_stats_fwd(amom, arows) -> mrla_base_tail_fwd].
 * v_ring: the whole ring (slot t-1 is written); out [opt unless tail]. */
int mrla_base_layer_fwd(const void* x, const float* pre_sc, const float* pre_sh, const void* identity, const float* wq,
                        const float* wk, int ksize, const float* wv, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, int bn_mode, float momentum, float eps, const float* dp,
                        float* mom, void* x_out, void* v_ring, float* k_ring, float* p_all, float* q, void* attn,
                        float* amom, int arows, float* bnbuf, void* out, int tail, int b, int c, int h, int w, int d, int T,
                        int t, int dtype, void* stream);

/* ... backward:  [tail: mrla_base_tail_stats_bwd(center = save_mean) -> mrla_bn_stats_bwd(centered = 1)] ->
 *   mrla_base_attend_bwd -> mrla_base_pmom_reduce -> mrla_base_gate_bwd -> mrla_base_dv_combine ->
 *   mrla_base_value_bwd_dv -> mrla_reduce_rows2.
 * small: [5, c] floats = cb[c,3] | dgamma | dbeta (tail only); tmom [trows, c, 2]; ppart [prows, t, c]; pmom [b, c, t];
 * dv [b, h, w, c] activation dtype; wsum as in mrla_light_tail_bwd; res as in mrla_base_value_bwd_dv. */
int mrla_base_layer_bwd(const void* dout, const void* x, const void* attn, const float* wq, const float* wk, int ksize,
                        const float* wv, const float* gamma, const float* dp, const float* mom, const float* q,
                        const float* bnbuf, int bn_mode, const void* v_ring, void* da_ring, const float* k_ring,
                        float* dk_ring, const float* p_all, float* tmom, int trows, float* small, float* ppart, int prows,
                        float* pmom, float* dyx, float* dwqk_part, void* dv, float* dwv_part, int rows, void* dx,
                        const void* pre, const float* pre_center, float* pre_tmom, float* wsum, int tail, int first_touch,
                        int res, int b, int c, int h, int w, int d, int T, int t, int Tc, int dtype, void* stream);

/* Token MRLA-light module (deit_mrla_light.py:194-209,234), forward:
 *   mrla_token_norm_pool -> mrla_light_gate_fwd (hw = n - 1) -> mrla_token_apply_fwd. */
int mrla_token_light_fwd(const void* x, const void* o_prev, const float* lnx_w, const float* lnx_b, const float* lno_w,
                         const float* lno_b, const float* wq, const float* wk, int ksize, const float* wv,
                         const float* lam, float eps, float* stats, float* mom, float* gate, void* out, int b, int n, int c,
                         int d, int res, int dtype, void* stream);

/* ... backward:  mrla_token_apply_bwd -> mrla_token_gate_bwd -> mrla_token_ln_bwd -> mrla_reduce_rows2.
 * part [prow, c, MRLA_TOKEN_PARTIALS] with prow = mrla_token_part_rows(); sums [c*MRLA_TOKEN_PARTIALS + 2*ksize]. */
int mrla_token_light_bwd(const void* dout, const void* x, const void* o_prev, const float* stats, const float* lnx_w,
                         const float* lnx_b, const float* lno_w, const float* lno_b, const float* wq, const float* wk,
                         int ksize, const float* wv, const float* gate, const float* lam, const float* mom, float* dxn,
                         float* part, int prow, float* bmom, float* dyx, float* dwqk_part, void* dx, void* do_prev,
                         float* sums, int b, int n, int c, int d, int res, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MRLA_HIP_H_ */
