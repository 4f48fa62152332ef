// Host-visible declarations shared by the kernel translation units and the C-ABI glue (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/mrla_hip.h"

namespace mrla {

constexpr int kMaxTasksPerWave = 4;   // apply_bwd keeps this many (plane-group, row-band) wgrad slots per wave

// Geometry of the NCHW slab decomposition; built on the host by make_slab_geo().
struct SlabGeo {
  int B, C, H, W, HW;
  int CP;        // channel planes per slab (one slab = CP*HW contiguous elements of one image)
  int slabs;     // slabs per image = ceil(C / CP)
  int WS;        // lanes per plane row = smallest power of two >= W
  int PW;        // planes marched side by side by one wave = 64 / WS
  int NG;        // wave groups per slab = ceil(CP / PW)
  int NB;        // row bands per plane group
  int RB;        // rows per band
  int BG;        // images looped over by one workgroup
  int astride;   // LDS elements reserved per staged array (16-byte multiple)
};

inline size_t dtype_size(int dtype) { return dtype == MRLA_F32 ? 4 : 2; }

inline int hip_status(hipError_t e) { return e == hipSuccess ? MRLA_OK : MRLA_EHIP; }

// Opt a kernel in to more than the default 48 KB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize) -- once per
// (kernel, device) and size, not on every launch: the attribute call costs host time on a path that issues ~600 launches
// per training step.  A lock-free lookup of what was granted before; the first call per kernel takes a mutex.  (capi.hip)
hipError_t lds_opt_in(const void* kernel, size_t bytes);

// Returns MRLA_OK or MRLA_EUNSUPPORTED (plane wider than a wave / slab does not fit LDS).
int make_slab_geo(SlabGeo* g, int B, int C, int H, int W, int dtype, int arrays, int bg_hint);

int launch_light_stats_fwd_nchw(const void* x, const void* o, const float* wv, float* mom, void* xout,
                                const float* psc, const float* psh, const SlabGeo& g, int dtype, int act,
                                hipStream_t st);
int launch_light_apply_fwd_nchw(const void* x, const void* o, const float* wv, const float* gate, const float* sc,
                                const float* sh, const float* lam, const float* dp, void* out, const SlabGeo& g,
                                int d, int res, int dtype, int act, hipStream_t st);
int launch_light_stats_bwd_nchw(const void* dout, const void* x, const void* o, const float* wv, const float* mom,
                                float* bmom, const SlabGeo& g, int dtype, int act, hipStream_t st);
int launch_light_apply_bwd_nchw(const void* dout, const void* x, const void* o, const float* wv, const float* gate,
                                const float* cb, const float* lam, const float* dp, const float* dyx, void* dx,
                                void* dprev, float* dwv_part, const SlabGeo& g, int d, int res, int relu, int dtype,
                                int act, hipStream_t st);

// gate.hip -- the small per-(image, channel) kernels
int launch_gate_fwd(const float* mom, const float* wq, const float* wk, int ks, float* gate, int B, int C, int HW,
                    int d, hipStream_t st);
int launch_bn_fwd(const float* mom, const float* gate, const float* lam, const float* gamma, const float* beta,
                  float* run_mean, float* run_var, int training, float momentum, float eps, float* sc, float* sh,
                  float* save_mean, float* save_inv, int B, int C, int HW, int d, hipStream_t st);
int launch_bn_bwd(const float* mom, const float* bmom, const float* gate, const float* lam, const float* gamma,
                  const float* dp, const float* save_mean, const float* save_inv, int training, float* cb, float* cb_lo,
                  float* dgamma, float* dbeta, float* dlam, int B, int C, int HW, int d, hipStream_t st);
// tok_part != null (token path): [tok_bands * B][C][kTokParts] partials of mrla_token_apply_bwd, completed in place with dy
int launch_gate_bwd(const float* mom, const float* bmom, const float* gate, const float* cb, const float* cb_lo,
                    const float* dp, const float* wq, const float* wk, int ks, float* dyx, float* dwqk_part, int B, int C, int HW,
                    int d, hipStream_t st, float* tok_part = nullptr, int tok_bands = 0);
// (the ranges gridDim.z cuts an image's strips AND rows into where few, large images would leave the chip idle: light_nhwc_wide.h)
int nhwc_set_row_cut_mode(int mode);              // 0 auto, 1 never, 2 wherever possible; returns the previous mode
int nhwc_wgrad_ranges(int B, int C, int H, int W);
int nhwc_bmom_ranges(int B, int C, int H, int W);
int nhwc_mom_ranges(int B, int C, int H, int W);  // ... of the forward statistics passes (records mom[z], merged into mom[0])
int launch_reduce_rows(const float* in, float* out, int rows, int n, hipStream_t st);
int launch_fold_rows(float* buf, int rows, int n, hipStream_t st);      // buf[0, :] = sum of the rows, in place
int launch_reduce_rows2(const float* in1, float* out1, int rows1, int n1, const float* in2, float* out2, int rows2, int n2,
                        hipStream_t st);


// base_nchw.hip -- MRLA-base (softmax over depth) kernels
int launch_base_gate_fwd(const float* mom, const float* wq, const float* wk, int ks, float* Kring, float* Pall,
                         float* q, int B, int C, int HW, int d, int T, int t, hipStream_t st);
int launch_base_attend_fwd(const void* x, const float* wv, void* Vring, const float* Pall, void* attn, float* amom,
                           const SlabGeo& g, int d, int T, int t, int dtype, hipStream_t st);
int launch_plain_bn_fwd(const float* amom, const float* gamma, const float* beta, float* run_mean, float* run_var,
                        int training, float momentum, float eps, float* sc, float* sh, float* save_mean,
                        float* save_inv, const float* pivot, int B, int C, int HW, hipStream_t st);
int launch_plain_bn_fwd_rec(const float* rec, const float* gamma, const float* beta, float* run_mean, float* run_var,
                            int training, float momentum, float eps, float* sc, float* sh, float* save_mean,
                            float* save_inv, int R, int C, hipStream_t st);
int launch_plain_bn_bwd(const float* tmom, const float* gamma, const float* save_mean, const float* save_inv,
                        int training, int centered, float* cb, float* dgamma, float* dbeta, int B, int C, int HW,
                        hipStream_t st);
int launch_base_tail_fwd(const void* x, const void* attn, const float* sc, const float* sh, const float* dp, void* out,
                         int B, int C, int HW, int dtype, hipStream_t st);
int launch_base_tail_stats_bwd(const void* dout, const void* attn, const float* sc, const float* sh, const float* dp,
                               float* tmom, const SlabGeo& g, int dtype, hipStream_t st);
int launch_base_attend_bwd(const void* dout, const void* attn, const float* sc, const float* sh, const float* dp,
                           const float* cb, const void* Vring, void* dAring, float* pmom, const SlabGeo& g, int T,
                           int t, int dtype, hipStream_t st);
int launch_base_gate_bwd(const float* mom, const float* pmom, const float* Pall, const float* q, const float* Kring,
                         float* dKring, const float* wq, const float* wk, int ks, float* dyx, float* dwqk_part, int B,
                         int C, int HW, int d, int T, int t, int first_touch, hipStream_t st, float* tok_part = nullptr,
                         int tok_bands = 0);
int launch_base_value_bwd(const void* dout, const void* x, const float* wv, const void* dAring, const float* Pall,
                          const float* dyx, void* dx, float* dwv_part, const SlabGeo& g, int d, int T, int t, int Tc,
                          int res, int dtype, hipStream_t st);


// tokens_nhwc.hip -- row-marching token map kernels (C % 64 == 0)
bool token_nhwc_applies(int C);
int token_bands(int B, int C, int side);       // row bands of the NHWC forward apply kernel
int token_bands_bwd(int B, int C, int side);   // ... of the backward apply kernel (= rows of `part` per image)
int launch_token_apply_fwd_nhwc(const void* x, const void* o, const float* stats, const float* wx, const float* bx,
                                const float* wo, const float* bo, const float* wv, const float* gate, const float* lam,
                                void* out, int B, int n, int C, int side, int d, int res, int dtype, hipStream_t st);
int launch_token_apply_bwd_nhwc(const void* dout, const void* x, const void* o, const float* stats, const float* wx,
                                const float* bx, const float* wo, const float* bo, const float* wv, const float* gate,
                                const float* lam, float* dxn, float* part, float* bmom, int B, int n, int C, int side,
                                int d, int dtype, hipStream_t st);

// base_nhwc.hip -- MRLA-base for channels_last activations; rings are slot-major [T][b,h,w,c]
bool base_nhwc_supported(int C, int dtype);
int base_nhwc_tiles(int B, int C, int HW, int dtype);
int base_nhwc_pmom_tiles(int B, int C, int HW, int dtype);
int launch_base_attend_fwd_nhwc(const void* Vring, const float* Pall, void* attn, float* amom_part, int B, int C, int HW,
                                int d, int T, int t, int dtype, hipStream_t st, int ext_gap = 0, int ext_off = 0);
int launch_base_dv_combine_nhwc(const void* dAring, const float* Pall, void* dv, int B, int C, int HW, int d, int T,
                                int t, int Tc, int dtype, hipStream_t st);
int launch_base_tail_fwd_nhwc(const void* x, const void* attn, const float* sc, const float* sh, const float* dp,
                              void* out, int B, int C, int HW, int dtype, hipStream_t st);
int launch_base_attend_bwd_nhwc(const void* dout, const void* attn, const float* sc, const float* sh, const float* dp,
                                const float* cb, const void* Vring, void* dAring, float* pmom_part, int B, int C, int HW,
                                int T, int t, int dtype, hipStream_t st, int ext_gap = 0, int ext_off = 0);
int launch_base_pmom_reduce(const float* part, float* pmom, int B, int C, int t, int tiles, hipStream_t st);
int launch_base_value_bwd_wide(const void* dout, const void* x, const float* wv, const void* dv, const float* dyx, void* dx,
                               float* dwv_part, const void* pre, const float* pre_center, float* pre_tmom, int B, int C,
                               int H, int W, int res, int dtype, hipStream_t st);      // light_nhwc_wide.hip (C % 64 == 0)
int launch_base_value_bwd_nhwc(const void* dout, const void* x, const float* wv, const void* dv, const float* dyx,
                               void* dx, float* dwv_part, const void* pre, const float* pre_center, float* pre_tmom, int B,
                               int C, int H, int W, int res, int dtype, hipStream_t st);


// tokens.hip -- MRLA-light on token sequences (DeiT)
int launch_token_value_fwd_nhwc(const void* x, const float* stats, const float* wx, const float* bx, const float* wv,
                                void* vslot, int B, int n, int C, int side, int dtype, hipStream_t st);
int launch_token_cls_fwd(const void* x, const float* stats, const float* wx, const float* bx, void* out, int B, int n,
                         int C, int dtype, hipStream_t st);
int launch_token_value_bwd_nhwc(const void* dout, const void* x, const float* stats, const float* wx, const float* bx,
                                const float* wv, const void* dv, float* dxn, float* part, int B, int n, int C, int side,
                                int dtype, hipStream_t st);
int launch_token_norm_pool(const void* x, const void* o, const float* wx, const float* bx, float eps, float* stats,
                           float* mom, int B, int n, int C, int dtype, hipStream_t st);
int launch_token_apply_fwd(const void* x, const void* o, const float* stats, const float* wx, const float* bx,
                           const float* wo, const float* bo, const float* wv, const float* gate, const float* lam,
                           void* out, int B, int n, int C, int side, int d, int res, int dtype, hipStream_t st);
int launch_token_apply_bwd(const void* dout, const void* x, const void* o, const float* stats, const float* wx,
                           const float* bx, const float* wo, const float* bo, const float* wv, const float* gate,
                           const float* lam, float* dxn, float* part, float* bmom, int B, int n, int C, int side, int d,
                           int dtype, hipStream_t st);
int launch_token_ln_bwd(const void* dout, const void* x, const void* o, const float* dxn, const float* dyx,
                        const float* stats, const float* wx, const float* wo, const float* lam, void* dx, void* dprev, int B,
                        int n, int C, int res, int dtype, hipStream_t st);


// bnact_nchw.hip -- fused BatchNorm2d (+ReLU) passes
int launch_plane_moments(const void* x, float* amom, float* pivot, int B, int C, int HW, int dtype, hipStream_t st);
int launch_plane_dmoments(const void* dy, const void* x, const float* sc, const float* sh, const float* center, int relu,
                          float* tmom, int B, int C, int HW, int dtype, hipStream_t st);
int launch_affine_act(const void* x, const void* dy, const float* a, const float* sc, const float* sh, int relu,
                      void* out, int B, int C, int HW, int dtype, int bwd, hipStream_t st);


// light_nhwc.hip / bnact_nhwc.hip -- channels_last variants
int nhwc_images_per_group(int B, int C, int W);
// conv1x1.hip -- 1x1 convolution as an MFMA GEMM with a BatchNorm-moments epilogue (bf16)
int conv1x1_rows(int M, int K, int N);
int conv1x1_plan(int M, int K, int N, int add, int* out);
int launch_conv1x1_fwd(const void* x, const void* w, void* y, float* part, int M, int K, int N, hipStream_t st);
// stem_pool_nhwc.hip -- maxpool3x3/s2/p1(relu(bn(x))) without the intermediate tensor (channels_last, C % 64 == 0)
int bn_pool_rows(int B, int C, int H, int W);
int launch_bn_relu_pool_fwd(const void* x, const float* sc, const float* sh, void* out, int B, int C, int H, int W,
                            int dtype, hipStream_t st);
int launch_bn_relu_pool_dmoments(const void* dp, const void* x, const float* sc, const float* sh, const float* center,
                                 float* tmom, int B, int C, int H, int W, int dtype, hipStream_t st);
int launch_bn_relu_pool_bwd(const void* dp, const void* x, const float* sc, const float* sh, const float* cb, void* dx,
                            int B, int C, int H, int W, int dtype, hipStream_t st);
// conv1x1_wide.hip -- the same product for wide outputs (N % 256 == 0): X streamed through LDS, optional addend
int conv1x1_wide_rows(int M, int K, int N);
int conv1x1_wide_plan(int M, int K, int N, int add, int* out);
int launch_conv1x1_wide(const void* x, const void* w, const void* addend, void* y, float* part, int M, int K, int N,
                        hipStream_t st);
// conv1x1_kstream.hip -- the same product for wide reductions (K >= 512): both operands streamed through LDS; the tile copy-out takes the BatchNorm moment records as well
int conv1x1_kstream_supported(int M, int K, int N);
int conv1x1_kstream_stages(int M, int K, int N);      // LDS stages of the kernel the planner picks (3, or 4: the 256 x 256 tile)
int conv1x1_kstream_rows(int M, int K, int N);        // rows of the moment records (one per pixel tile)
int launch_conv1x1_kstream(const void* x, const void* w, void* y, float* part, int M, int K, int N, hipStream_t st);
// conv1x1_wgrad.hip -- its weight gradient dW[n,k] = sum_m dY[m,n] X[m,k] as a split-M MFMA GEMM (bf16)
int conv1x1_wgrad_rows(int M, int K, int N);
int conv1x1_wgrad_plan(int M, int K, int N, int* out);
int launch_conv1x1_wgrad(const void* dy, const void* x, float* part, void* dw, int dw_f32, int M, int K, int N,
                         hipStream_t st);
// weight_bank.hip -- all eligible fp32 conv weights -> bf16 copies (+ transposes) in one launch
int launch_weight_bank_refresh(const long long* table, int entries, int max_tiles, hipStream_t st);
// light_nhwc_wide.hip -- the C % 64 == 0 forms on the LDS-DMA row pipeline (nhwc_rows.h)
int launch_light_stats_fwd_wide(const void* x, const void* o, const float* wv, float* mom, void* xout, const float* psc,
                                const float* psh, void* vout, int B, int C, int H, int W, int dtype, int act,
                                hipStream_t st, bool fused_no_x = false, int mom_ranges = 1);
int launch_light_apply_fwd_wide(const void* x, const void* o, const float* wv, const float* gate, const float* sc,
                                const float* sh, const float* lam, const float* dp, void* out, int B, int C, int H,
                                int W, int d, int res, int dtype, int act, hipStream_t st);
int launch_light_apply_fwd_pre_wide(const void* pre, const void* o, const float* psc, const float* psh, const float* wv,
                                    const float* gate, const float* sc, const float* sh, const float* lam,
                                    const float* dp, void* out, int B, int C, int H, int W, int d, int res, int dtype,
                                    hipStream_t st);
int launch_light_apply_bwd_wide(const void* dout, const void* x, const void* o, const float* wv, const float* gate,
                                const float* cb, const float* lam, const float* dp, const float* dyx, void* dx,
                                void* dprev, float* dwv_part, const void* pre, const float* pre_center, float* pre_tmom,
                                int B, int C, int H, int W, int d, int res, int relu, int dtype, int act, hipStream_t st);
int launch_light_stats_bwd_wide(const void* dout, const void* x, const void* o, const float* wv, const float* mom,
                                float* bmom, int B, int C, int H, int W, int dtype, int act, hipStream_t st);
// light_nhwc_lean.hip -- the backward passes that re-form x_t from conv3's output and the shortcut (no stored x_t)
int light_lean_supported(int B, int C, int H, int W, int dtype);
int launch_light_stats_bwd_lean_wide(const void* dout, const void* pre, const void* o, const float* wv, const float* psc,
                                     const float* psh, const float* mom, float* bmom, int B, int C, int H, int W, int dtype,
                                     hipStream_t st);
int launch_light_apply_bwd_lean_wide(const void* dout, const void* pre, const void* o, const float* wv, const float* psc,
                                     const float* psh, const float* gate, const float* cb, const float* lam,
                                     const float* dp, const float* dyx, void* dx, void* dprev, float* dwv_part,
                                     const float* pre_center, float* pre_tmom, int B, int C, int H, int W, int d, int res,
                                     int dtype, hipStream_t st);
int launch_light_stats_fwd_nhwc(const void* x, const void* o, const float* wv, float* mom, void* xout,
                                const float* psc, const float* psh, void* vout, int B, int C, int H, int W, int dtype,
                                int act, hipStream_t st, bool fused_no_x = false, int mom_ranges = 1);
int launch_light_apply_fwd_nhwc(const void* x, const void* o, const float* wv, const float* gate, const float* sc,
                                const float* sh, const float* lam, const float* dp, void* out, int B, int C, int H,
                                int W, int d, int res, int dtype, int act, hipStream_t st);
int launch_light_stats_bwd_nhwc(const void* dout, const void* x, const void* o, const float* wv, const float* mom,
                                float* bmom, int B, int C, int H, int W, int dtype, int act, hipStream_t st);
int launch_light_apply_bwd_nhwc(const void* dout, const void* x, const void* o, const float* wv, const float* gate,
                                const float* cb, const float* lam, const float* dp, const float* dyx, void* dx,
                                void* dprev, float* dwv_part, const void* pre, const float* pre_center, float* pre_tmom,
                                int B, int C, int H, int W, int d, int res, int relu, int dtype, int act, hipStream_t st);
int nhwc_bn_splits(int B, int C, int HW);
int launch_nhwc_pool_fused(const void* pre, const float* sc, const float* sh, const void* o, float* part, float* mom,
                           int B, int C, int HW, int dtype, hipStream_t st);
int launch_light_apply_fwd_pre_nhwc(const void* pre, const void* o, const float* psc, const float* psh, const float* wv,
                                    const float* gate, const float* sc, const float* sh, const float* lam,
                                    const float* dp, void* out, int B, int C, int H, int W, int d, int res, int dtype,
                                    hipStream_t st);
int launch_nhwc_moments(const void* x, const void* dy, const float* sc, const float* sh, int relu, const float* dp,
                        float* out, float* pivot, int B, int C, int HW, int dtype, int mode, hipStream_t st);
int launch_nhwc_affine(const void* x, const void* dy, const float* a, const float* sc, const float* sh, int relu,
                       void* out, int B, int C, int HW, int dtype, int bwd, hipStream_t st);

}  // namespace mrla
