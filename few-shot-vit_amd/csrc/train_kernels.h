// Launchers of the training-path kernels (train_kernels.hip, attention_bwd.hip, head_bwd in head.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace fsvit {

// one job of the batched weight pack (same meaning as launch_pack_weight's arguments); `out` is the packed destination
struct PackJob { const float* w; void* out; int O, Ig, KH, KW, groups, mode, rows_pad, Kw, hd_rows, hdp_rows, hd_cols, hdp_cols; };
struct PackJobs { static constexpr int MAX = 40; PackJob job[MAX]; };
int launch_pack_weight_multi(const PackJob* jobs, int n, int dtype, hipStream_t s);
int launch_pack_weight(const float* w, void* out, int O, int Ig, int KH, int KW, int groups, int mode, int rows_pad, int Kw, int hd_rows, int hdp_rows,
                       int hd_cols, int hdp_cols, int dtype, hipStream_t s);
// Deferred split-slab finalizes of a whole backward pass in one or two launches (the 38 per-layer finalize launches of a step were 10 .. 25 us
// of latency each): kind 0 = wgrad_finalize_kernel, 1 = its 1x1 fast path, 2 = dense grouped, 3 = wgrad3x3_finalize_kernel (a = grouped flag,
// b = njobs); the job table travels as a kernel argument, blockIdx.y = job
struct FinJob { const float* y; float* dw; int kind, Ng, Ig, KH, KW, g, splits, Kc_pad, hd_rows, hdp_rows, hd_cols, hdp_cols; };
struct FinJobs { static constexpr int MAX = 48; FinJob job[MAX]; };
int launch_wgrad_finalize_multi(const FinJob* jobs, int n, hipStream_t s);
// DropPath scales of all calls of a step in one launch: scales[k][b] = masks[k][b] / keep[k]
struct DropKeep { static constexpr int MAX = 64; float inv[MAX]; };
int launch_droppath_scales(const float* masks, float* scales, int ncalls, int n_img, const float* keep, hipStream_t s);
int launch_wgrad_finalize(const float* y, float* dw, int Ng, int Ig, int KH, int KW, int g, int splits, int Kc_pad, int hd_rows, int hdp_rows, int hd_cols,
                          int hdp_cols, hipStream_t s);
// grouped conv via one dense GEMM: keeps the diagonal (same-group) blocks
int launch_wgrad_finalize_dense(const float* y, float* dw, int Ng, int Ig, int KH, int KW, int groups, int splits, int Kc_pad, hipStream_t s);
int launch_transpose_cols(const void* in, void* out, int M, int ld, int c0, int ncols, int Mpad, int dtype, hipStream_t s);
int launch_im2col_t(const void* x, void* out, int B, int H, int W, int ld, int c0, int C, int KH, int KW, int stride, int pad, int OH, int OW, int Mpad,
                    int dtype, hipStream_t s);
int launch_unpatch2(const void* g, void* dx, int B, int OH, int OW, int C, int dtype, hipStream_t s);
int bn_reduce_blocks(int M);
int launch_bn_reduce(const void* a, const void* z, const float* mean, const float* invstd, float* partial, int M, int C, int bwd, int dtype, hipStream_t s,
                     const void* add_a = nullptr, const void* add_b = nullptr, const float* add_scale = nullptr, int rows_per_img = 0,
                     const float* act_sa = nullptr, const float* act_sb = nullptr);
int launch_bn_fwd_finalize(const float* partial, int M, int C, float eps, float momentum, const float* gamma, const float* beta, float* rmean, float* rvar,
                           float* mean, float* invstd, float* sa, float* sb, hipStream_t s);
int launch_bn_bwd_finalize(const float* partial, int M, int C, const float* gamma, const float* invstd, float* dgamma, float* dbeta, float* ca, float* cb,
                           float* cc, int frozen, hipStream_t s);
int launch_bn_frozen_coeffs(int C, float eps, const float* gamma, const float* beta, const float* rmean, const float* rvar, float* mean, float* invstd, float* sa,
                            float* sb, hipStream_t s);
int launch_bn_apply(const void* z, const float* sa, const float* sb, const void* res, void* y, size_t M, int C, int act, int dtype, hipStream_t s);
int launch_bn_act_bwd(const void* dout, const void* z, const float* sa, const float* sb, const void* res, void* g, size_t M, int C, int dtype, hipStream_t s);
int launch_bn_bwd_apply(const void* dy, const void* z, const float* mean, const float* invstd, const float* ca, const float* cb, const float* cc, void* dz,
                        size_t M, int C, int dtype, hipStream_t s, const void* acc = nullptr, const float* scale2 = nullptr, void* out2 = nullptr,
                        size_t rows_per_img = 0, const float* act_sa = nullptr, const float* act_sb = nullptr);
// stem tail: LeakyReLU(sa * z + sb + res) -> MaxPool2d(2) -> + pos in one pass (arg = window position | 4 if the maximum is positive), and its backward
// (rsa / rsb: `res` is the identity path's PRE-normalisation map, its BatchNorm scale / shift are applied in the same pass)
int launch_bn_pool_fwd(const void* z, const float* sa, const float* sb, const void* res, const float* pos, void* out, unsigned char* arg, int B, int OH, int OW,
                       int C, int dtype, hipStream_t s, const float* rsa = nullptr, const float* rsb = nullptr);
// both BatchNorm backwards behind that tail (bn3 on z3, the identity path's on zd) straight from the pooled gradient + arg: the routed 2 x 2 gradient
// map is never stored.  partial3 / partiald: pool_bn_bwd_blocks() * 2 * C floats each; coef3 / coefd: [3][C] = ca | cb | cc of launch_bn_bwd_finalize
bool pool_bn_bwd_supported(int C, int dtype);
int pool_bn_bwd_blocks(int B, int OH, int OW, int C, int dtype);
int launch_pool_bn_bwd_reduce(const void* dout, const unsigned char* arg, const void* z3, const void* zd, const float* mean3, const float* is3, const float* meand,
                              const float* isd, float* partial3, float* partiald, int B, int OH, int OW, int C, int dtype, hipStream_t s);
int launch_pool_bn_bwd_apply(const void* dout, const unsigned char* arg, const void* z3, const void* zd, const float* mean3, const float* is3, const float* meand,
                             const float* isd, const float* coef3, const float* coefd, void* dz3, void* dzd, int B, int OH, int OW, int C, int dtype, hipStream_t s);
// 1x1 conv weights with a pre-norm BatchNorm's scale / shift folded in: wf [N][Kw] (storage type), bf [N] fp32 (fold_prenorm_kernel)
int launch_fold_prenorm(const float* W, const float* sa, const float* sb, void* wf, float* bf, int N, int C, int Kw, int dtype, hipStream_t s);
int launch_bn_fwd_finalize_nblk(const float* partial, int nblk, int M, int C, float eps, float momentum, const float* gamma, const float* beta, float* rmean, float* rvar,
                                float* mean, float* invstd, float* sa, float* sb, hipStream_t s);
int launch_bn_bwd_finalize_nblk(const float* partial, int nblk, int M, int C, const float* gamma, const float* invstd, float* dgamma, float* dbeta, float* ca, float* cb,
                                float* cc, int frozen, hipStream_t s);
int launch_pool_act_bwd(const void* dout, const unsigned char* arg, void* g, int B, int OH, int OW, int C, int dtype, hipStream_t s);
int launch_gelu_fwd(const void* z, void* h, size_t n, int dtype, hipStream_t s);
int launch_gelu_bwd(const void* dh, const void* z, void* dz, size_t n, int dtype, hipStream_t s);
int launch_add_scaled(const void* a, const void* br, const float* scale, void* out, size_t n, size_t per_img, int dtype, hipStream_t s);
int launch_maxpool2_idx(const void* in, const float* pos, void* out, unsigned char* arg, int B, int OH, int OW, int C, int dtype, hipStream_t s);
int launch_maxpool2_bwd(const void* dout, const unsigned char* arg, void* din, int B, int OH, int OW, int C, int dtype, hipStream_t s);
int launch_avgpool_bwd(const float* dfeat, void* dx, int B, int HW, int C, int dtype, hipStream_t s);
int launch_batch_sum(const void* g, float* out, int B, size_t per_img, int dtype, hipStream_t s);
int launch_bcast_add(const void* x, const float* p, void* y, int B, size_t per_img, int dtype, hipStream_t s);
int launch_fill_f32(float* p, float v, size_t n, hipStream_t s);
int launch_scale_copy(const float* in, float* out, size_t n, float sc, hipStream_t s);
// out[c] = sum_m a[m][c]; partial: bn_reduce_blocks(M) * 2 * C floats of scratch
int launch_colsum(const void* a, float* partial, float* out, int M, int C, int dtype, hipStream_t s);
int launch_sgd_multi(const void* items_dev, int n_items, size_t max_numel, float lr, float momentum, float wd, int first, hipStream_t s);
int launch_sgd(float* p, const float* g, float* buf, size_t n, float lr, float momentum, float wd, int first, hipStream_t s);
// attention backward (attention_bwd.hip): qkv [B*S][3*heads*hdp], dctx [B*S][heads*hdp] -> dqkv [B*S][3*heads*hdp]
int launch_attention_bwd(const void* qkv, const void* dctx, void* dqkv, int B, int S, int heads, int hd, int hdp, float scale, int dtype, hipStream_t s);
// prototype head backward (head.hip): dlogits [E,Q,way] -> dfeat_shot [E,way,shot,D], dfeat_query [E,Q,D], dtemp[E] (cos method).
// up: device scalar every gradient is multiplied with (the upstream gradient of a scalar loss); ticket: zeroed device word -> dtemp[E] = sum_e dtemp[e]
int launch_proto_head_bwd(const float* feat_shot, const float* feat_query, const float* dlogits, int E, int way, int shot, int Q, int D, float temp,
                          float* dfeat_shot, float* dfeat_query, float* dtemp, hipStream_t s, const float* temp_dev = nullptr, const float* up = nullptr,
                          unsigned* ticket = nullptr);
int launch_proto_head_sqr_bwd(const float* feat_shot, const float* feat_query, const float* dlogits, int E, int way, int shot, int Q, int D, float temp,
                          float* dfeat_shot, float* dfeat_query, float* dtemp, hipStream_t s, const float* temp_dev = nullptr, const float* up = nullptr,
                          unsigned* ticket = nullptr);

// ---- ViT / DeiT training (deit.py): LayerNorm with kept row statistics, token assembly, the final norm on the cls row
int launch_ln_train_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M, int D, float eps, int dtype, hipStream_t s);
int ln_bwd_blocks(int M);
int launch_ln_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const void* add, void* dx, float* partial,
                  float* dgamma, float* dbeta, int M, int D, int dtype, hipStream_t s);
int launch_vit_assemble(const void* zpe, const float* cls, const float* pos, void* tokens, int B, int S, int D, int dtype, hipStream_t s);
int launch_vit_patch_rows(const void* dtok, void* dzpe, int B, int S, int D, int dtype, hipStream_t s);
int launch_vit_cls_ln_fwd(const void* tokens, const float* gamma, const float* beta, float* feat, float* mean, float* rstd, int B, int S, int D, float eps, int dtype,
                          hipStream_t s);
int launch_vit_cls_ln_bwd(const float* dfeat, const void* tokens, const float* mean, const float* rstd, const float* gamma, void* dtok, float* partial, float* dgamma,
                          float* dbeta, int B, int S, int D, int dtype, hipStream_t s);
}  // namespace fsvit
