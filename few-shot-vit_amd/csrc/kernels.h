// Internal launchers of the fsvit gfx950 kernels (dtype: 0 = f32, 1 = bf16 storage).
#pragma once
#include <hip/hip_runtime.h>
#include "conv_gemm.h"

namespace fsvit {

int launch_im2col27(const float* x_nchw, void* out, int B, int H, int W, int OH, int OW, int dtype, hipStream_t s);
// im2col + stem conv1 + bn1 + LeakyReLU in one pass (bf16, 80x80 images, 64 channels): patches [B*1600][32], c1 [B*1600][64]; w = the packed conv1 layer [64][kw]
bool stem_conv1_supported(int dtype, int img, int C0);
int launch_stem_conv1(const float* x_nchw, void* patches, void* c1, const void* w, int kw, const float* bias, int B, hipStream_t s);
int launch_maxpool2_pos(const void* in, const float* pos, void* out, int B, int OH, int OW, int C, int dtype, hipStream_t s);

// qkv [B*S][3*heads*hdp] (channel = x*heads*hdp + y*hdp + z) -> ctx [B*S][heads*hdp]
int launch_attention(const void* qkv, void* ctx, int B, int S, int heads, int hdp, float scale, int dtype, hipStream_t s);
size_t attention_lds_bytes(int S, int hdp, int dtype);
int attention_padded_head_dim(int hd, int S, int dtype);      // what the weight packer pads a head to

// x [B][HW][C] -> feat [B][C] fp32 = scale[c] * mean_hw(x) + shift[c]
int launch_pool_affine(const void* x, const float* scale, const float* shift, float* feat, int B, int HW, int C, int dtype, hipStream_t s);

// cosine / squared-distance prototype head (meta_baseline.py:33-47, utils/__init__.py:78-109)
int launch_proto_head(const float* feat_shot, const float* feat_query, int E, int way, int shot, int Q, int D,
                      float temp, int method, float* logits, float* acc, float* loss, hipStream_t s);

// fused stage-1 block (bf16, Visformer-S geometry only); x and y must be different buffers
bool stage1_fused_supported(int dtype, int C1, int hid, int group, int H1);
int launch_stage1_block(const void* x, void* y, const void* w1, const float* b1, const void* w2, const void* w3, int B, hipStream_t s);

// stage-1 block, band-per-wave design (stage1_rows.hip; same contract as launch_stage1_block); wimg is built once by launch_stage1_pack
// from the standard packed layers w1 [256][128], w2 [8][32][320], w3 [128][256]
bool stage1_rows_supported(int dtype, int C1, int hid, int group, int H1);
size_t stage1_rows_image_bytes();
int launch_stage1_pack(const void* w1, const void* w2, const void* w3, void* wimg, hipStream_t s);
int launch_stage1_rows(const void* x, void* y, const void* wimg, const float* b1, int B, hipStream_t s);

// fused row-wise Mlp of the attention blocks (mlp_rows.hip; bf16, C = 256, hidden = 1024): y = x + W2 GELU(W1 x + b1) (+ b2), in place allowed.
// wimg / b1img are built once by launch_mlp_pack from the standard packed layers (w1 [hid][k1w], w2 [C][k2w]).
bool mlp_rows_supported(int dtype, int C, int hid);
bool mlp_rows_proj_supported(int C, int hid, int KC);        // the attention block's proj conv + residual as a prologue on the same rows
size_t mlp_rows_image_bytes(int C, int hid, int KC);
int launch_mlp_pack(const void* w1, int k1w, const float* b1, const void* w2, int k2w, const void* wp, int kpw, int KC, void* wimg, float* b1img, int C,
                    int hid, hipStream_t s);
int launch_mlp_rows(const void* x, void* y, const void* wimg, const float* b1img, const float* b2, const void* ctx, int KC, int M, int C, int hid,
                    hipStream_t s);

// fused qkv conv + attention core of the Visformer stage-2 block (qkv_attn.hip; bf16, C = 256, 6 heads x 48, S <= 112): ctx rows
// [B*S][heads*hdp] from x rows [B*S][C]; wimg is built once by launch_qkv_attn_pack from the packed qkv layer (w [3*heads*hdp][kw])
bool qkv_attn_supported(int dtype, int C, int heads, int hdp, int S);
size_t qkv_attn_image_bytes();
int launch_qkv_attn_pack(const void* w, int kw, void* img, hipStream_t s);
int launch_qkv_attn(const void* x, void* ctx, const void* wimg, const float* bias, int B, int S, float scale, hipStream_t s);

// distillation head (token_label.hip; fp32): LinearClassifier forward / backward, generate_softlabel, SoftTargetCrossEntropy, AdamW
int launch_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, int K, hipStream_t s);
int launch_linear_bwd(const float* dy, const float* x, const float* w, float* dx, int accumulate_dx, float* dw, float* db, int M, int N, int K, hipStream_t s);
int launch_token_softlabel(const float* lt, float* soft, int B, int T, int C, int k, int bp, double smoothing, hipStream_t s);
int launch_soft_target_ce(const float* z, const float* tgt, float* rowloss, float* dz, int R, int C, float gscale, hipStream_t s);
int launch_tokens_to_f32(const void* in, const float* scale, const float* shift, float* out, size_t n, int C, int dtype, hipStream_t s);
int launch_add_f32_into(void* inout, const float* add, size_t n, int dtype, hipStream_t s);
int launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, float wd, int step, hipStream_t s);

// ViT / DeiT helpers (vit.hip)
int launch_patchify(const float* x_nchw, void* out, int B, int img, int p, int Kp, int dtype, hipStream_t s);
int launch_cls_pos(const float* cls_plus_pos0, void* tokens, int B, int S, int D, int dtype, hipStream_t s);
int launch_layernorm(const void* x, void* y, int M, int D, float eps, int dtype, hipStream_t s);
int launch_final_ln_cls(const void* tokens, const float* gamma, const float* beta, float* feat, int B, int S, int D, float eps, int dtype, hipStream_t s);

}  // namespace fsvit
