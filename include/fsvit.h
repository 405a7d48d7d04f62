/* fsvit — C-ABI of the MI355X-native few-shot ViT hot path (libfsvit.so).
 *
 * The reference (DongSky/few-shot-vit) is 100 % Python: its boundary for this path is the
 * `models.make / nn.Module.forward` contract, not an FFI.  These entry points are
 * what a ctypes binding of that contract calls; each one names the reference
 * interface it replaces.  Plain pointers and sizes only — no torch types.
 *
 * Conventions
 *   - every `*_dev` / activation / output pointer is a DEVICE pointer owned by the caller;
 *     `fsvit_tensor.data` pointers are HOST pointers (state-dict tensors, fp32, contiguous);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work;
 *   - return 0 on success; > 0 is a hipError_t, < 0 an fsvit error; `fsvit_last_error()` (thread
 *     local) describes the last failure;
 *   - `dtype` selects storage + MFMA arithmetic of activations/weights: FSVIT_F32 = exact fp32
 *     MFMA (v_mfma_f32_16x16x4_f32, the parity mode), FSVIT_BF16 = bf16 MFMA with fp32 accumulate, FSVIT_F16 = fp16 MFMA with
 *     fp32 accumulate (eval engines only: same kernels, same rate, 3 more mantissa bits than bf16; |activations| must stay
 *     below 65504, which BatchNorm / LayerNorm networks do with a wide margin).  FSVIT_BF16X2 / FSVIT_F16X2 (eval engines,
 *     fsvit_conv_gemm and - FSVIT_BF16X2 only - the two trainers, which pack the limb words of the weights and of the weight-gradient
 *     GEMM's activation operand on the device every step): fp32 storage as FSVIT_F32, but every GEMM runs on the 16-bit MFMA with both operands split into two 16-bit
 *     limbs (hi + lo; two MFMAs per K chunk = all four limb products, fp32 accumulate): 16 (bf16 limbs) / 22 (fp16 limbs) significand
 *     bits per operand at 1/4 of the 16-bit MFMA rate = 4 x the fp32-MFMA rate.  Weights given to fsvit_conv_gemm in these modes are
 *     4-byte limb pairs (upper half hi, lower half lo).
 */
#ifndef FSVIT_H
#define FSVIT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { FSVIT_F32 = 0, FSVIT_BF16 = 1, FSVIT_F16 = 2, FSVIT_BF16X2 = 3, FSVIT_F16X2 = 4 };
enum { FSVIT_ACT_NONE = 0, FSVIT_ACT_GELU = 1, FSVIT_ACT_LRELU = 2 };
enum { FSVIT_HEAD_COS = 0, FSVIT_HEAD_SQR = 1, FSVIT_HEAD_DOT = 2 };
enum {
  FSVIT_ERR_ARG = -1,        /* bad argument / unsupported configuration        */
  FSVIT_ERR_KEY = -2,        /* state-dict key missing or wrong shape (KeyError) */
  FSVIT_ERR_IMG_SIZE = -3,   /* input image size does not match the model (AssertionError) */
  FSVIT_ERR_WORKSPACE = -4   /* workspace too small                              */
};

const char* fsvit_last_error(void);
/* Build identification: "fsvit <version> gfx950". */
const char* fsvit_version(void);

/* ---------------------------------------------------------------- Visformer encoder
 * Replaces `Visformer.__init__` + `load_state_dict` + `Visformer.forward` in eval mode
 * (test_phase/models/visformer.py:291-462; factory visformer_small_80 :482-487). */
typedef struct fsvit_visformer fsvit_visformer;

typedef struct fsvit_tensor {
  const char* name;       /* state-dict key relative to the encoder, e.g. "stem.conv1.weight" */
  const float* data;      /* HOST pointer, fp32, contiguous, PyTorch layout ([O,I,kh,kw], [C], [1,C,H,W]) */
  int ndim;
  int64_t shape[4];
} fsvit_tensor;

typedef struct fsvit_visformer_cfg {   /* visformer.py:292-295 arguments that shape the eval path */
  int img_size;
  int init_channels;
  int embed_dim;
  int depth[3];
  int num_heads;
  float mlp_ratio;
  int group;
  float bn_eps;
} fsvit_visformer_cfg;

/* Folds every eval-mode BatchNorm into the adjacent conv (visformer.py:118-124), reorders conv
 * weights to K-major (ky,kx,c), pads head dims for MFMA, converts to `dtype` and uploads.
 * Fails with FSVIT_ERR_KEY when a key of SURVEY.md Appendix A is absent or mis-shaped
 * (load_state_dict(strict=True) behaviour, models/models.py:21-26). */
int fsvit_visformer_create(const fsvit_visformer_cfg* cfg, const fsvit_tensor* state_dict, int n_tensors,
                           int dtype, fsvit_visformer** out);
void fsvit_visformer_destroy(fsvit_visformer* h);
int fsvit_visformer_out_dim(const fsvit_visformer* h);            /* `.out_dim` (visformer.py:298) */
int fsvit_visformer_dtype(const fsvit_visformer* h);
/* Bytes of scratch needed to push `chunk_images` images through the encoder at once. */
size_t fsvit_visformer_workspace_bytes(const fsvit_visformer* h, int chunk_images);
/* x_nchw_dev: [n_img,3,img,img] fp32 (what `data.cuda()` hands the model, test_few_shot.py:81).
 * feat_dev:   [n_img,out_dim] fp32 pooled features (visformer.py:462).
 * Images are processed in chunks sized to `ws_bytes` (at least one image must fit). */
int fsvit_visformer_forward(fsvit_visformer* h, const float* x_nchw_dev, int n_img, int img_h, int img_w,
                            float* feat_dev, void* ws_dev, size_t ws_bytes, void* stream);
/* Test hook: copy the named residual-stream activation of the FIRST chunk (NHWC, storage dtype) to
 * dst_dev during the next forwards.  Names: "stem" (after max-pool + pos_embed1), "stage1.N",
 * "patch_embed2"/"patch_embed3" (incl. pos_embed), "stage2.N", "stage3.N".  dst_dev NULL clears. */
int fsvit_encoder_set_tap(void* encoder, const char* name, void* dst_dev, size_t bytes);   /* ViT names: "embed", "blocks.N" */

/* Live per-launch timing (bench.py roofline leg): between begin and end every kernel launch of
 * fsvit_visformer_forward / fsvit_vit_forward is bracketed by HIP events ON THE STREAM IT IS LAUNCHED ON.  `end`
 * synchronises and returns one record per (layer, kernel) with summed algorithmic FLOPs (2*MAC,
 * unpadded dims), summed milliseconds and the launch count. */
typedef struct fsvit_prof_rec {
  char layer[48];     /* e.g. "stem.conv3", "stage2.attn.qkv" */
  int kernel_id;      /* see fsvit_kernel_name */
  int launches;
  double flops;
  double ms;
} fsvit_prof_rec;
int fsvit_encoder_profile_begin(void* encoder);
int fsvit_encoder_profile_end(void* encoder, fsvit_prof_rec* out, int max_recs, int* n_out);
/* Device kernel (template instantiation) behind a kernel_id, as rocprofv3 --kernel-trace names it. */
const char* fsvit_kernel_name(int kernel_id, int dtype);

/* ---------------------------------------------------------------- ViT / DeiT encoder
 * Replaces `VisionTransformer.__init__` + `load_state_dict` + `forward` in eval mode (test_phase/models/deit.py:139-218;
 * factories deit_{tiny,small,base,nano}_patch16_224, deit_{nano,micro}_patch6_84, :220-357).  LayerNorm eps 1e-6,
 * Linear layers with bias, qkv_bias=True, cls token at index 0, features = norm(x)[:, 0].  Same conventions as the
 * Visformer handle; state-dict keys of SURVEY.md Appendix A (DeiT part). */
typedef struct fsvit_vit fsvit_vit;
typedef struct fsvit_vit_cfg {          /* deit.py:142-144 */
  int img_size;
  int patch_size;
  int embed_dim;
  int depth;
  int num_heads;
  float mlp_ratio;
  float ln_eps;
} fsvit_vit_cfg;
int fsvit_vit_create(const fsvit_vit_cfg* cfg, const fsvit_tensor* state_dict, int n_tensors, int dtype, fsvit_vit** out);
void fsvit_vit_destroy(fsvit_vit* h);
int fsvit_vit_out_dim(const fsvit_vit* h);                           /* `.out_dim` (deit.py:147) */
size_t fsvit_vit_workspace_bytes(const fsvit_vit* h, int chunk_images);
int fsvit_vit_forward(fsvit_vit* h, const float* x_nchw_dev, int n_img, int img_h, int img_w, float* feat_dev,
                      void* ws_dev, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- episode head
 * Replaces MetaBaseline.forward after the encoder call (test_phase/models/meta_baseline.py:33-47),
 * utils.compute_logits (utils/__init__.py:78-101) and, per episode, F.cross_entropy +
 * utils.compute_acc with fs.make_nk_label labels (test_few_shot.py:87-90).
 * feat_shot_dev [E,way,shot,D], feat_query_dev [E,Q,D] fp32 -> logits_dev [E,Q,way];
 * acc_dev / loss_dev [E] may be NULL. */
int fsvit_proto_head(const float* feat_shot_dev, const float* feat_query_dev, int E, int way, int shot, int Q, int D,
                     float temp, int method, float* logits_dev, float* acc_dev, float* loss_dev, void* stream);

/* The same with the temperature read from DEVICE memory (`self.temp` is an nn.Parameter the optimizer updates on the GPU, meta_baseline.py:20-21):
 * the meta-tuning step never brings it to the host, so the step issues no stream synchronisation. */
int fsvit_proto_head_devtemp(const float* feat_shot_dev, const float* feat_query_dev, int E, int way, int shot, int Q, int D,
                             const float* temp_dev, int method, float* logits_dev, float* acc_dev, float* loss_dev, void* stream);

/* Whole `MetaBaseline.forward(x_shot, x_query)` (meta_baseline.py:24-47) in eval mode:
 * x_shot_dev [E,way,shot,3,H,W], x_query_dev [E,Q,3,H,W] fp32 -> logits_dev [E,Q,way].
 * feat_dev: scratch [(E*way*shot + E*Q), out_dim] fp32. */
int fsvit_meta_baseline_forward(void* encoder /* fsvit_visformer* or fsvit_vit* */, const float* x_shot_dev, const float* x_query_dev,
                                int E, int way, int shot, int Q, int img_h, int img_w, float temp, int method,
                                float* logits_dev, float* acc_dev, float* loss_dev, float* feat_dev,
                                void* ws_dev, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- operator level (tests, reuse)
 * Implicit-GEMM conv with fused epilogue; see few-shot-vit_amd/csrc/conv_gemm.h for the layout.
 * x NHWC [B,H,W,x_cstride]; w [groups][N][Kw] K-major (ky,kx,c); y NHWC [B,OH,OW,y_cstride]. */
int fsvit_conv_gemm(const void* x_dev, const void* w_dev, const float* bias_dev, const void* res_dev,
                    const float* pos_dev, void* y_dev, int B, int H, int W, int Cin, int x_cstride,
                    int KH, int KW, int stride, int pad, int N, int y_cstride, int Kw, int groups,
                    int act, int res_first, int dtype, void* stream);
/* Fused tail of the stem ConvBlock (visformer.py:213 conv3, :216-217 downsample, :232-237 add + LeakyReLU + MaxPool2d(2)) + pos_embed1
 * (:431): y [B, H/2, W/2, N] = maxpool2(lrelu(conv3x3_pad1(x [B,H,W,Cin], w[:, 0:9*Cin]) + x2 [B*H*W][x2_cstride](:, 0:K2) . w[:, Kw-bke : Kw-bke+K2]
 * + bias)) + pos [(H/2)*(W/2)][N]; BatchNorms folded into w / bias by the caller; bke = 128 bytes of K (64 bf16 / 32 fp32). */
int fsvit_conv_stem_tail(const void* x_dev, const void* w_dev, const float* bias_dev, const float* pos_dev, const void* x2_dev, int x2_cstride,
                         int K2, void* y_dev, int B, int H, int W, int Cin, int N, int Kw, int dtype, void* stream);
/* qkv [B*S][3*heads*hdp] -> ctx [B*S][heads*hdp] (visformer.py:183-190) */
int fsvit_attention(const void* qkv_dev, void* ctx_dev, int B, int S, int heads, int hdp, float scale,
                    int dtype, void* stream);
/* Fused qkv conv + attention core of a Visformer stage-2 Attention block (visformer.py:172-190; the eval BatchNorm folded into the
 * conv as column scale + bias by the caller), bf16, Visformer-S geometry only (C = 256, 6 heads, head dim padded to 48, S <= 112):
 * x rows [B*S][C] -> ctx rows [B*S][heads*hdp].  wqkv [3*heads*hdp][kw] K-major bf16, rows ordered (q|k|v, head, z); bias fp32
 * [3*heads*hdp] or NULL.  Equals fsvit_conv_gemm (1x1, N = 3*heads*hdp) followed by fsvit_attention without the qkv tensor.
 * Also the stage-3 geometry on the row-wise kernel: C = 512, head dim padded to 96, S <= 32 (one image per wave). */
int fsvit_qkv_attention(const void* x_dev, const void* wqkv_dev, int kw, const float* bias_dev, void* ctx_dev, int B, int S, int C,
                        int heads, int hdp, float scale, void* stream);
/* norm1 + qkv Linear + attention core of a ViT / DeiT block in one launch (deit.py:40-58 Attention.forward, :69 `x + attn(norm1(x))` without the
 * proj): x rows [B*S][C] bf16 -> ctx rows [B*S][heads*hdp]; LayerNorm without affine (gamma / beta folded into wqkv / bias by the caller),
 * wqkv [3*heads*hdp][kw] K-major bf16 rows ordered (q|k|v, head, z), bias fp32 or NULL.  Built for C = 384, head dim 64, S <= 256
 * (DeiT-S/16: 197 tokens).  Equals fsvit_ln_linear_rows followed by fsvit_attention without the qkv tensor in HBM. */
int fsvit_vit_ln_qkv_attention(const void* x_dev, const void* wqkv_dev, int kw, const float* bias_dev, void* ctx_dev, int B, int S, int C,
                               int heads, int hdp, float eps, float scale, void* stream);
/* One fused Visformer stage-1 block (visformer.py:259-263 with attn_disabled + spatial_conv Mlp :152-163), bf16,
 * 20x20 tokens, 128 channels, 256 hidden, 8 groups: y = x + conv3(GELU(conv2_g(GELU(conv1(x)+b1)))).
 * x, y NHWC [B,20,20,128] bf16 (distinct buffers); w1 [256][128], w2 [8][32][320], w3 [128][256] packed K-major bf16.
 * Always the 16-wave ring kernel (stage1_ring.hip): the in-process cross-check of fsvit_stage1_block_hw. */
int fsvit_stage1_block(const void* x_dev, void* y_dev, const void* w1_dev, const float* b1_dev, const void* w2_dev,
                       const void* w3_dev, int B, void* stream);
/* The same block for any square token map of 4 .. 20 a side: x, y NHWC [B,H,W,128], dtype FSVIT_BF16 / FSVIT_F16; weights as above.  This is the launch the
 * engines run: stage1_w4.hip (one wave per SIMD, weights in registers / AGPRs, x and the first hidden map in pixel rings; bf16: GELU by table look-up on
 * the bf16-rounded pre-activation). */
int fsvit_stage1_block_hw(const void* x_dev, void* y_dev, const void* w1_dev, const float* b1_dev, const void* w2_dev, const void* w3_dev, int B, int H, int W,
                          int dtype, void* stream);
/* Fused Mlp of a Visformer attention block (visformer.py:146-150 with spatial_conv=False, + the residual of :262):
 * y = x + W2 GELU(W1 x + b1) (+ b2), rows = tokens.  bf16; C = 256 / hidden = 1024 and C = 512 / hidden = 2048 (stages 2, 3 of
 * Visformer-S).  x, y [M][C]
 * (y may alias x); w1 [hid][k1w], w2 [C][k2w] K-contiguous bf16 rows (BatchNorm already folded into w1 / b1); b1 [hid], b2 [C]
 * fp32 or NULL.  The operator form packs the weights on every call (the engine packs once per checkpoint). */
int fsvit_mlp_rows(const void* x_dev, void* y_dev, const void* w1_dev, int k1w, const float* b1_dev, const void* w2_dev, int k2w,
                   const float* b2_dev, int M, int C, int hid, void* stream);
/* The same with the attention block's proj conv + residual (visformer.py:176,:261) as a prologue on the same rows:
 * x1 = x + wp ctx; y = x1 + W2 GELU(W1 x1 + b1) (+ b2).  ctx [M][KC] = attention output with zero-padded head dims, wp [C][kpw];
 * (C, KC) = (256, 288) or (512, 576).  ctx == NULL: plain fsvit_mlp_rows. */
int fsvit_proj_mlp_rows(const void* x_dev, void* y_dev, const void* ctx_dev, const void* wp_dev, int kpw, int KC, const void* w1_dev, int k1w,
                        const float* b1_dev, const void* w2_dev, int k2w, const float* b2_dev, int M, int C, int hid, void* stream);
/* The ViT / DeiT block tail (test_phase/models/deit.py:69-72, `x = x + attn.proj(ctx); x = x + mlp(norm2(x))`) on token-major rows, bf16,
 * (C, hidden, KC) = (384, 1536, 384):  x1 = x + bp + wp ctx;  y = x1 + b2 + W2 GELU(W1 LN(x1) + b1), LN = (x1 - mean) / sqrt(var + eps) without
 * affine - the caller folds norm2's gamma / beta into w1 / b1 (W1 diag(gamma), b1 + W1 beta).  In place (y == x) allowed. */
int fsvit_vit_block_tail(const void* x_dev, void* y_dev, const void* ctx_dev, const void* wp_dev, int kpw, int KC, const float* bp_dev,
                         const void* w1_dev, int k1w, const float* b1_dev, const void* w2_dev, int k2w, const float* b2_dev, int M, int C, int hid,
                         float eps, void* stream);
/* The ViT / DeiT block head up to the qkv Linear (deit.py:40-47,:69 `attn(norm1(x))`): y [M][N] = b + W LN(x [M][C]) on token-major bf16 rows,
 * C = 384, N a multiple of 32, LN without affine (the caller folds norm1's gamma / beta into w [N][kw] / b).  C = 512: the same row-wise kernel
 * WITHOUT the LayerNorm, y = b + W x (the Visformer stage-3 qkv conv with its eval BatchNorm folded, visformer.py:175; eps ignored, b may be NULL). */
int fsvit_ln_linear_rows(const void* x_dev, void* y_dev, const void* w_dev, int kw, const float* b_dev, int M, int C, int N, float eps, void* stream);
/* PatchEmbed of a Visformer stage (test_phase/models/visformer.py:266-288: Conv2d k2 s2 -> BatchNorm, then `x + pos_embed`, :437-447) on the
 * row-wise kernel: x NHWC bf16 [B][H][H][Ci], 4 Ci = 512; w [N][kw] K-major in (ky, kx, c) order with the eval BatchNorm folded in, bias fp32 [N]
 * or NULL, pos fp32 [(H/2)^2][N]; y [B (H/2)^2][N] bf16. */
int fsvit_patch_embed2x2(const void* x_dev, void* y_dev, const void* w_dev, int kw, const float* bias_dev, const float* pos_dev, int B, int H, int Ci,
                         int N, void* stream);
/* ---------------------------------------------------------------- distillation head
 * Replaces, for sun_meta_training/offline.py: `LinearClassifier.forward` / its autograd (models/classifier.py:27-34) as used by
 * `TokenLabelOffline` (models/token_label.py:36-60) on the 25 tokens and on the pooled feature, `generate_softlabel`
 * (offline.py:57-76), `SoftTargetCrossEntropy` (offline.py:34-45) and the AdamW update (offline.py:233).  fp32, token-major rows. */
/* The distillation phase's encoder returns `(x, pooled)` (sun_meta_training/models/visformer.py:464): the post-norm token map
 * [n_img][25][512] fp32 (token-major; the reference's [B, 512, 5, 5] permuted by (0, 2, 3, 1)) of the last eval forward / train forward,
 * and the gradient hand-over for the train backward (added to the pooled feature's gradient path; consumed by the next backward). */
int fsvit_visformer_last_tokens(fsvit_visformer* h, const void* ws_dev, size_t ws_bytes, int n_img, float* tokens_dev, void* stream);
/* (train-mode counterparts: fsvit_visformer_train_tokens / fsvit_visformer_train_set_token_grad, declared with the trainer below) */
/* y [M][N] = x [M][K] w[N][K]^T + b[N] (b may be NULL); K % 4 == 0. */
int fsvit_linear_forward(const float* x_dev, const float* w_dev, const float* b_dev, float* y_dev, int M, int N, int K, void* stream);
/* dx [M][K] (+)= dy w (NULL: skipped), dw [N][K] = dy^T x and db [N] = column sums of dy (NULL: skipped); any N (dy is staged 256 classes at a time). */
int fsvit_linear_backward(const float* dy_dev, const float* x_dev, const float* w_dev, float* dx_dev, int accumulate_dx, float* dw_dev,
                          float* db_dev, int M, int N, int K, void* stream);
/* teacher token logits [B][T][C] (T = 25 tokens in (h, w) order = the reference's [B, C, 5, 5].permute(0, 2, 3, 1)) -> soft labels
 * [B*T][C+1]: top-k scatter for the T - bp tokens with the largest max logit, the reference's background row (on_value at column 1,
 * offline.py:61,71) for the other bp. */
int fsvit_token_softlabel(const float* teacher_logits_dev, float* soft_dev, int B, int T, int C, int k, int bp, double smoothing, void* stream);
/* row_loss [R] = -sum_c target log_softmax(logits) (the caller's mean is over R), dlogits [R][C] = grad_scale (softmax * sum(target) -
 * target) or NULL; any C. */
int fsvit_soft_target_ce(const float* logits_dev, const float* target_dev, float* row_loss_dev, float* dlogits_dev, int R, int C,
                         float grad_scale, void* stream);
/* F.normalize(x, dim=-1) of R rows of length D (utils.compute_logits metric 'cos', test_phase/utils/__init__.py:82-84; the
 * `nn-classifier` head, test_phase/models/classifier.py:38-55): y = x / max(|x|, 1e-12), inv_norm [R] for the backward
 * dx = (dy - y <y, dy>) * inv_norm. */
int fsvit_row_normalize(const float* x_dev, float* y_dev, float* inv_norm_dev, int R, int D, void* stream);
int fsvit_row_normalize_backward(const float* y_dev, const float* inv_norm_dev, const float* dy_dev, float* dx_dev, int R, int D, void* stream);
/* AdamW (decoupled weight decay), update number `step` >= 1: p, exp_avg m, exp_avg_sq v updated in place. */
int fsvit_adamw_step(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, size_t n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, void* stream);
/* The same update for a table of tensors in ONE launch (offline.py:233 `optimizer.step()` over every parameter of the student): items_dev = n_items rows of
 * five 8-byte words {p, g, m, v, numel} in device memory; every tensor at update number `step`; max_numel = the largest numel (sizes the grid). */
int fsvit_adamw_step_multi(const void* items_dev, int n_items, size_t max_numel, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                           void* stream);

int fsvit_im2col27(const float* x_nchw_dev, void* out_dev, int B, int H, int W, int dtype, void* stream);
/* im2col27 + stem conv1 + bn1 (folded) + LeakyReLU(0.1) in one pass (visformer.py:209-210,218), bf16, 80x80 images, 64 channels:
 * patches [B*1600][32] (the rows fsvit_im2col27 writes) and c1 [B*1600][64]; w [64][kw] K-major bf16 with K columns (ky,kx,c), kw >= 32. */
int fsvit_stem_conv1(const float* x_nchw_dev, const void* w_dev, int kw, const float* bias_dev, void* patches_dev, void* c1_dev,
                     int B, int H, int W, void* stream);
int fsvit_maxpool2_pos(const void* in_dev, const float* pos_dev, void* out_dev, int B, int OH, int OW, int C,
                       int dtype, void* stream);
int fsvit_pool_affine(const void* x_dev, const float* scale_dev, const float* shift_dev, float* feat_dev,
                      int B, int HW, int C, int dtype, void* stream);

/* ---------------------------------------------------------------- meta-tuning step
 * Replaces the encoder part of `model.train(); logits = model(x_shot, x_query); loss.backward(); optimizer.step()`
 * (meta_tuning_sun_m/train_meta.py:161-177): train-mode Visformer forward (batch-statistics BatchNorm that also
 * updates running_mean / running_var in place, momentum 0.1, visformer.py:53-64; DropPath visformer.py:89-96)
 * with saved activations, and the backward to every parameter gradient.
 * Parameters stay in the framework's own fp32 tensors ON THE DEVICE (they change every step):
 *   name  reference state-dict key of the encoder ("stem.conv1.weight", "stage2.0.attn.qkv.weight", "norm.bn.running_var", ...)
 *   data  device pointer, fp32, PyTorch layout;  grad  device pointer of the same shape that train_backward OVERWRITES
 *         (NULL: not wanted; buffers such as running stats never get one). */
typedef struct fsvit_param {
  const char* name;
  float* data;
  float* grad;
  int64_t numel;
} fsvit_param;
typedef struct fsvit_visformer_trainer fsvit_visformer_trainer;
/* post-norm token map of the last train_forward / gradient hand-over for the next train_backward (distillation head, see above) */
int fsvit_visformer_train_tokens(fsvit_visformer_trainer* t, float* tokens_dev, void* stream);
int fsvit_visformer_train_set_token_grad(fsvit_visformer_trainer* t, const float* dtokens_dev);
int fsvit_visformer_trainer_create(const fsvit_visformer_cfg* cfg, int dtype, fsvit_visformer_trainer** out);
void fsvit_visformer_trainer_destroy(fsvit_visformer_trainer* t);
/* bytes of workspace one forward+backward over n_img images needs (saved activations + temporaries) */
size_t fsvit_visformer_trainer_workspace_bytes(fsvit_visformer_trainer* t, const fsvit_param* params, int n_params, int n_img,
                                               float drop_path_rate);
/* BatchNorm layers frozen inside the training step (utils.freeze_bn, meta_tuning_sun_m/train_meta.py:156-157, utils/__init__.py:150-153):
 * train_forward normalises with the running statistics and leaves them untouched; train_backward returns dz = gamma * invstd * dy and the
 * gamma / beta gradients against the running statistics.  Default off. */
int fsvit_visformer_trainer_set_freeze_bn(fsvit_visformer_trainer* t, int on);

/* x_nchw_dev [n_img,3,H,W] fp32 -> feat_dev [n_img,out_dim] fp32.  masks_dev: the DropPath Bernoulli draws, one row of
 * n_img 0/1 floats per DropPath call with a non-zero rate, in call order (stage-1 blocks: one call; stage-2/3 blocks: attn
 * then mlp) - the caller draws them (floor(keep + rand), visformer.py:93-95) so the random stream stays the framework's.
 * The workspace must stay untouched until train_backward returns. */
int fsvit_visformer_train_forward(fsvit_visformer_trainer* t, const fsvit_param* params, int n_params, const float* x_nchw_dev,
                                  int n_img, int img_h, int img_w, float drop_path_rate, const float* masks_dev, float* feat_dev,
                                  void* ws_dev, size_t ws_bytes, void* stream);
/* dfeat_dev [n_img,out_dim] fp32 -> every params[i].grad */
int fsvit_visformer_train_backward(fsvit_visformer_trainer* t, const fsvit_param* params, int n_params, const float* dfeat_dev,
                                   void* stream);
/* ---- ViT / DeiT training step (test_phase/models/deit.py:61-78 Block, :139-218 VisionTransformer in train mode; what loss.backward() does for
 * the DeiT encoders in meta_tuning_sun_m/train_meta.py:161-177).  Same contract as the Visformer trainer above: parameters stay in the caller's
 * fp32 device tensors (fsvit_param table with the reference's state-dict names: cls_token, pos_embed, patch_embed.proj.*, blocks.N.{norm1,norm2}.*,
 * blocks.N.attn.{qkv,proj}.*, blocks.N.mlp.{fc1,fc2}.*, norm.*), train_forward saves the activations in ws_dev, train_backward overwrites every
 * params[i].grad.  DropPath: 2 calls per block whose rate linspace(0, drop_path_rate, depth)[i] is non-zero, masks_dev [calls][n_img] of 0 / 1 in
 * forward order (fsvit_vit_trainer_droppath_calls); Dropout rates are 0 in every shipped factory and are not built.  dtype FSVIT_F32 or FSVIT_BF16
 * (both cover the 197-token factories; fp32 runs plain FMA loops). */
typedef struct fsvit_vit_trainer fsvit_vit_trainer;
int fsvit_vit_trainer_create(const fsvit_vit_cfg* cfg, int dtype, fsvit_vit_trainer** out);
void fsvit_vit_trainer_destroy(fsvit_vit_trainer* t);
int fsvit_vit_trainer_droppath_calls(const fsvit_vit_trainer* t, float drop_path_rate);
size_t fsvit_vit_trainer_workspace_bytes(fsvit_vit_trainer* t, const fsvit_param* params, int n_params, int n_img, float drop_path_rate);
int fsvit_vit_train_forward(fsvit_vit_trainer* t, const fsvit_param* params, int n_params, const float* x_nchw_dev, int n_img, int img_h, int img_w,
                            float drop_path_rate, const float* masks_dev, float* feat_dev, void* ws_dev, size_t ws_bytes, void* stream);
int fsvit_vit_train_backward(fsvit_vit_trainer* t, const fsvit_param* params, int n_params, const float* dfeat_dev, void* stream);

/* Backward of fsvit_proto_head, method 'cos' (meta_baseline.py:33-47): dlogits [E,Q,way] -> dfeat_shot [E,way,shot,D],
 * dfeat_query [E,Q,D], dtemp_per_episode [E] (sum it for the learnable temperature, meta_baseline.py:20-21). */
int fsvit_proto_head_backward(const float* feat_shot_dev, const float* feat_query_dev, const float* dlogits_dev, int E, int way,
                              int shot, int Q, int D, float temp, float* dfeat_shot_dev, float* dfeat_query_dev,
                              float* dtemp_per_episode_dev, void* stream);
/* The same for method 'sqr' (meta_baseline.py:38-41: logits = -temp * |q - mean_shot|^2). */
int fsvit_proto_head_backward_sqr(const float* feat_shot_dev, const float* feat_query_dev, const float* dlogits_dev, int E, int way,
                                  int shot, int Q, int D, float temp, float* dfeat_shot_dev, float* dfeat_query_dev,
                                  float* dtemp_per_episode_dev, void* stream);
/* Both backward forms with the temperature read from device memory (method FSVIT_HEAD_COS or FSVIT_HEAD_SQR). */
int fsvit_proto_head_backward_devtemp(const float* feat_shot_dev, const float* feat_query_dev, const float* dlogits_dev, int E, int way,
                                      int shot, int Q, int D, const float* temp_dev, int method, float* dfeat_shot_dev,
                                      float* dfeat_query_dev, float* dtemp_per_episode_dev, void* stream);
/* Head + loss of the meta-tuning step in one launch (train_meta.py:167-169: `logits = model(x_shot, x_query).view(-1, n_way); loss = F.cross_entropy(logits,
 * label); acc = utils.compute_acc(logits, label)`): logits [E,Q,way], per-episode accuracy / mean cross entropy [E], dlogits [E,Q,way] = d(mean CE over all
 * E * Q rows) / dlogits (may be NULL), and loss_acc_mean_dev[2] = {F.cross_entropy value, compute_acc value} (may be NULL).  labels_dev: int64 [E*Q] class
 * indices, or NULL for fs.make_nk_label's (few_shot.py:13-16: q / (Q / way)).  temp_dev (may be NULL): the temperature read from device memory instead of
 * `temp`.  ticket_dev: a zero-initialised 4-byte device word (required with loss_acc_mean_dev): the last workgroup to finish sums the episodes in index order
 * and leaves the word at zero - no second launch, no atomics on floating-point data, results bit-reproducible.
 * A label outside [0, way) - including F.cross_entropy's ignore_index, which this head does not implement - makes the episode's loss (and the batch mean) NaN. */
int fsvit_proto_head_ce(const float* feat_shot_dev, const float* feat_query_dev, const long long* labels_dev, int E, int way, int shot, int Q, int D, float temp,
                        const float* temp_dev, int method, float* logits_dev, float* dlogits_dev, float* acc_per_episode_dev, float* loss_per_episode_dev,
                        float* loss_acc_mean_dev, unsigned* ticket_dev, void* stream);
/* Its backward (methods FSVIT_HEAD_COS / FSVIT_HEAD_SQR): dlogits of fsvit_proto_head_ce times the upstream gradient *dloss_dev (NULL = 1) -> dfeat_shot,
 * dfeat_query, dtemp_dev [E] per episode (may be NULL) and, with ticket_dev, dtemp_dev[E] = their sum (dtemp_dev then holds E + 1 floats). */
int fsvit_proto_head_ce_backward(const float* feat_shot_dev, const float* feat_query_dev, const float* dlogits_dev, const float* dloss_dev, int E, int way,
                                 int shot, int Q, int D, float temp, const float* temp_dev, int method, float* dfeat_shot_dev, float* dfeat_query_dev,
                                 float* dtemp_dev, unsigned* ticket_dev, void* stream);
/* The same update for a table of tensors in one launch.  items_dev: DEVICE array of n_items records {float* param; const float* grad;
 * float* momentum_buf; size_t numel} (4 x 8 bytes each); max_numel = the largest numel.  All tensors share lr / momentum / weight_decay /
 * first_step (one param_group of torch.optim.SGD, meta_tuning_sun_m/utils/__init__.py:128-139). */
int fsvit_sgd_step_multi(const void* items_dev, int n_items, size_t max_numel, float lr, float momentum, float weight_decay, int first_step,
                         void* stream);

/* torch.optim.SGD(momentum, weight_decay) update of one fp32 tensor (utils/__init__.py:127-139) */
int fsvit_sgd_step(float* param_dev, const float* grad_dev, float* momentum_buf_dev, size_t n, float lr, float momentum,
                   float weight_decay, int first_step, void* stream);
/* ---------------------------------------------------------------- device-resident dataset transform
 * Replaces, for a gathered batch of dataset indices, the per-image CPU pipeline of the reference's datasets
 * (test_phase/datasets/mini_imagenet.py:47-56: Resize((88,88)) -> CenterCrop(80) -> ToTensor -> Normalize;
 * tiered_imagenet.py:53-57: Resize(80) -> ToTensor -> Normalize): images_dev uint8 [N,H,W,3] resident in HBM,
 * index_dev int64 [B] -> out_dev fp32 [B,3,OH,OW].  The resize is Pillow's BILINEAR (two 8-bit passes, 22-bit fixed point),
 * bit-exact: the caller passes Pillow's coefficient tables (xmin / count / coef[ksize] per resized column and row, device
 * int32; few-shot-vit_amd/datasets/transforms.py computes them as Resample.c does).  crop_*0 = offset of the OHxOW crop in
 * the resized image; mean3 / std3 are HOST pointers to 3 floats. */
int fsvit_image_transform_gather(const uint8_t* images_dev, int H, int W, const int64_t* index_dev, int B, const int32_t* xmin_h_dev,
                                 const int32_t* cnt_h_dev, const int32_t* coef_h_dev, int ksize_h, const int32_t* xmin_v_dev,
                                 const int32_t* cnt_v_dev, const int32_t* coef_v_dev, int ksize_v, int crop_y0, int crop_x0, int OH, int OW,
                                 const float* mean3_host, const float* std3_host, float* out_dev, void* stream);

/* Operator level of the training path: attention backward (qkv, dctx -> dqkv; hd real / hdp padded head dim). */
/* Weight gradient of a 3x3 / stride 1 / pad 1 convolution straight from the NHWC activations (no im2col, no transposed copies):
 * dw[O][Ig][3][3] (fp32, overwritten) = sum_m dz[m][o] * x[pix(m) + tap][i].  x [B,H,W,groups*Ig], dz [B*H*W][O], dtype FSVIT_BF16 / FSVIT_F16, or
 * FSVIT_BF16X2 / FSVIT_F16X2 = fp32 activations split into two 16-bit limbs on the way into LDS (the two-limb trainers' kernel).
 * Replaces what autograd computes for stem.conv2 / conv3 and stage1.*.mlp.conv2 in train_meta.py:228-232 (visformer.py:152-163, :209-237).
 * Built shapes: groups = 8 with 32 -> 32 channels per group (W <= 20); dense O = 128, Ig = 64 / 128 (W <= 40). */
int fsvit_conv3x3_wgrad(const void* x_dev, const void* dz_dev, float* dw_dev, int B, int H, int W, int O, int Ig, int groups, int dtype, void* stream);

/* The grouped 3x3 / stride 1 / pad 1 convolution of the stage-1 Mlp (visformer.py:148: conv2, 8 groups of 32 -> 32 channels) as the training step
 * runs it, forward and - on the transposed, tap-flipped pack - as its data gradient: one wave per group, weights resident in registers.
 * x, y [B,H,W,256] NHWC; w_packed [256][Kw], row o = output channel, columns (ky, kx, c) of its group, Kw >= 288; dtype FSVIT_BF16 / FSVIT_F16; W <= 20. */
int fsvit_gconv3x3(const void* x_dev, const void* w_packed_dev, int Kw, void* y_dev, int B, int H, int W, int dtype, void* stream);

/* Weight gradient of a 1x1 convolution / Linear from the row-major activations: dw[N][C] (fp32, overwritten) = sum_m dz[m][n] * x[m][c].
 * x [M][C], dz [M][N], dtype FSVIT_BF16 / FSVIT_F16 (16-bit rows) or FSVIT_BF16X2 / FSVIT_F16X2 (fp32 rows, split into two 16-bit limbs on the way into
 * LDS: the two-limb trainers' kernel); N and C multiples of 8.  (conv1 / conv3 of the Mlps, qkv, proj in train_meta.py:228-232.) */
int fsvit_conv1x1_wgrad(const void* x_dev, const void* dz_dev, float* dw_dev, int M, int N, int C, int dtype, void* stream);

int fsvit_attention_backward(const void* qkv_dev, const void* dctx_dev, void* dqkv_dev, int B, int S, int heads, int hd, int hdp,
                             float scale, int dtype, void* stream);

/* ---------------------------------------------------------------- episode sampler (host only, no GPU)
 * The draws of `CategoriesSampler.__iter__` (test_phase/datasets/samplers.py:19-35) replayed natively on the legacy numpy generator state: per episode
 * `np.random.choice(n_cat, n_cls, replace=False)`, then per chosen class `np.random.choice(catlocs[c], n_per, replace=False)`.  mt_key [624] / mt_pos =
 * the MT19937 state of `np.random.get_state()` (updated in place: hand it back with `np.random.set_state`); cat_items = the concatenated `catlocs`,
 * cat_offsets [n_cat + 1] their boundaries; out [n_batch][ep_per_batch][n_cls][n_per] dataset indices.  Bit-identical to numpy's stream. */
int fsvit_sampler_draw(unsigned int* mt_key, int* mt_pos, const long long* cat_offsets, const long long* cat_items, int n_cat, int n_batch, int ep_per_batch,
                       int n_cls, int n_per, long long* out_indices);

#ifdef __cplusplus
}
#endif
#endif /* FSVIT_H */
