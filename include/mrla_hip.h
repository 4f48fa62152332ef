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
enum { MRLA_NCHW = 0, MRLA_NHWC = 1 };
enum { MRLA_ACT_NONE = 0, MRLA_ACT_GELU = 1 };
enum { MRLA_BN_NONE = 0, MRLA_BN_TRAIN = 1, MRLA_BN_EVAL = 2 };

/* Moment record sizes (floats per (image, channel)). */
#define MRLA_FWD_MOMENTS 6 /* sum x, sum V, sum o, sum V^2, sum V*o, sum o^2 */
#define MRLA_BWD_MOMENTS 3 /* sum dOut, sum dOut*V, sum dOut*o */

int mrla_abi_version(void);

/* Number of rows of the `dwv_part` scratch that mrla_light_apply_bwd writes for this problem
 * (each row is [c, 9] floats); negative on error. */
int mrla_light_wgrad_rows(int b, int c, int h, int w, int dtype, int layout);

/* ---- MRLA-light, pass 1 of 2 (forward) ----------------------------------------------------------
 * mom[b, c, 6] <- per-plane moments of x, V = act(dwconv(x)), o_prev.
 * Replaces: avg_pool + Wv conv of mrla_light_module.py:56,61 (deit_mrla_light.py:161,166-167) as the
 * producer of everything the gate and the train-mode bn_mrla statistics (resnet_mrla_light.py:116)
 * need.  o_prev [opt]: block input `identity` (resnet_mrla_light.py:110-111). */
int mrla_light_stats_fwd(const void* x, const void* o_prev, const float* wv /*[c,3,3]*/, float* mom,
                         int b, int c, int h, int w, int dtype, int layout, int act, void* stream);

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

/* ---- backward pass 1 of 2: bmom[b, c, 3] ---------------------------------------------------------
 * The reductions autograd performs in MulBackward / ExpandBackward / NativeBatchNormBackward. */
int mrla_light_stats_bwd(const void* dout, const void* x, const void* o_prev, const float* wv, float* bmom, int b,
                         int c, int h, int w, int dtype, int layout, int act, void* stream);

/* ---- BatchNorm backward constants + dgamma, dbeta, dlambda ---------------------------------------
 * cb[c, 4] = (e, f, G, H) such that dm = e*dp[b]*dOut + f*a[b,g]*V + G*o_prev + H.
 * gamma [opt]: null = no BatchNorm (e = 1, f = G = H = 0; dgamma/dbeta untouched).
 * lam, dp, dlam [opt]. */
int mrla_light_bn_bwd(const float* mom, const float* bmom, const float* gate, const float* lam, const float* gamma,
                      const float* dp, const float* save_mean, const float* save_inv, int bn_mode, float* cb,
                      float* dgamma, float* dbeta, float* dlam, int b, int c, int hw, int d, void* stream);

/* ---- gate backward -------------------------------------------------------------------------------
 * dyx[b, c] = (gradient wrt the pooled descriptor y) / hw ; dwqk_part[b, 2*ksize] = per-image partial
 * sums of dWq (first ksize) and dWk.  cb, dp [opt]. */
int mrla_light_gate_bwd(const float* mom, const float* bmom, const float* gate, const float* cb, const float* dp,
                        const float* wq, const float* wk, int ksize, float* dyx, float* dwqk_part, int b, int c,
                        int hw, int d, void* stream);

/* ---- backward pass 2 of 2 ------------------------------------------------------------------------
 * dx = res*dOut + dwconv^T(a*dm*act'(U)) + dyx ;  do_prev = lam*dm ;  dwv_part[rows, c, 9] partial sums
 * of dWv over groups of images (rows = mrla_light_wgrad_rows()).  cb, lam, dp, o_prev, do_prev [opt]. */
int mrla_light_apply_bwd(const void* dout, const void* x, const void* o_prev, const float* wv, const float* gate,
                         const float* cb, const float* lam, const float* dp, const float* dyx, void* dx,
                         void* do_prev, float* dwv_part, int b, int c, int h, int w, int d, int res, int dtype,
                         int layout, int act, void* stream);

/* out[n] = sum over rows of in[rows, n] (fixed order, double accumulation). */
int mrla_reduce_rows(const float* in, float* out, int rows, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MRLA_HIP_H_ */
