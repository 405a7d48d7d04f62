// Implicit-GEMM convolution / 1x1-conv GEMM on MFMA with a fused epilogue.
//
//   y[m, g*N + n] = epi( sum_k x_patch[m, g, k] * w[g, n, k] )      m = (b, oy, ox), k = (ky, kx, c)
//   epi(v) = v + bias[n];  (+res if res_first);  act;  (+res if !res_first);  + pos[m % (OH*OW), n]
//
// Covers every conv of the Visformer eval path with folded BatchNorm (test_phase/models/visformer.py:
// stem conv2/conv3 :211-213, Mlp conv1/conv2/conv3 :146-150, Attention qkv/proj :175-177,
// PatchEmbed proj :276) and, after a 27->32 im2col, stem conv1/downsample (:209,:216).
#pragma once
#include <stddef.h>
#include <hip/hip_runtime.h>

// Types shared by every kernel namespace.  They live in a namespace WITHOUT functions so that argument-dependent lookup on a
// ConvGemmParams argument never drags in the other 16-bit build's launchers (see fsvit_common.h: kernel sources are compiled twice).
namespace fsvit_types {


// ACT_MUL (training, data gradients): y = (acc + bias) * res - `res` is a MULTIPLIER, not an addend (the saved GELU derivative of the layer in front)
enum Act { ACT_NONE = 0, ACT_GELU = 1, ACT_LRELU = 2, ACT_MUL = 3 };

struct ConvGemmParams {
  const void* x;       // activations NHWC [B, H, W, x_cstride]
  const void* w;       // packed weights [groups][N][Kw], K order (ky, kx, c), rows zero-padded to Kw
  const float* bias;   // [groups*N] or nullptr
  const void* res;     // residual [M, y_cstride] or nullptr (same dtype as y)
  const float* pos;    // [OH*OW, y_cstride] fp32 or nullptr
  void* y;             // [M, y_cstride]
  int B, H, W;
  int Cin;             // input channels per group
  int x_cstride;       // channels per input pixel (all groups)
  int OH, OW;
  int KH, KW, stride, pad;
  int N;               // output channels per group
  int y_cstride;       // channels per output pixel (all groups)
  int K;               // KH*KW*Cin
  int Kw;              // packed weight row length (K rounded up to the 128-byte K slice)
  int M;               // B*OH*OW
  int groups;
  int act;             // enum Act
  int res_first;       // 1: residual is added before the activation (stem), 0: after (blocks)
  int log2Cin;         // valid when KH*KW > 1
  // v2 only -------------------------------------------------------------------------------------
  // Tail operand: ONE extra 128-byte K slice whose activation rows come from x2[m][0..K2) (rows in
  // natural (b, oy, ox) order, aligned with the output pixels) and whose weights are columns
  // [Kw - BKE, Kw) of the packed rows.  Used to fold the stem's downsample/identity conv (K = 27 im2col
  // taps) into conv3 (visformer.py:232-235: out = bn3(conv3(.)) + bn_d(conv_d(x))).
  const void* x2;
  int x2_cstride;
  int K2;              // 0 = no tail operand
  // pool2: rows are enumerated 2x2-window-major and the epilogue max-pools each window after the
  // activation (MaxPool2d(2), visformer.py:237); y / pos rows are then the pooled pixels.
  int pool2;
  // Row remap of the OUTPUT (and residual): row m of image b = m / (OH*OW) goes to row b * y_rpi + y_row0 + m % (OH*OW).
  // y_rpi == 0: dense rows.  Used to write the ViT patch tokens behind each image's cls token (deit.py:200-201).
  int y_rpi, y_row0;
  // Split-K wgrad (training): weight row n of group g starts at w + g*w_gstride + n*w_rstride elements (both 0:
  // the packed default g*N*Kw + n*Kw), and the output can be stored as fp32 whatever the storage dtype.
  long w_gstride, w_rstride;
  int out_f32;
  // Training forward: with act == ACT_GELU and y2 != nullptr the epilogue also stores the GELU's DERIVATIVE at the pre-activation, y2[m][n]
  // (same layout and dtype as y) - the backward multiplies by it inside the data-gradient GEMM's epilogue (ACT_MUL) instead of keeping the
  // pre-activation map for a separate elementwise pass in each direction.
  void* y2;
  // Training forward, BatchNorm statistics from the producer (round 4): stats != nullptr -> the kernel also writes per-workgroup partial sums of the
  // values it stores, stats[(wg * 2 + 0) * y_cstride + n] = sum over the workgroup's output rows of y[m][n], [(wg * 2 + 1) ...] = sum of y^2
  // (fp32, of the accumulators before the rounding to the storage type; fixed summation order: bit-reproducible).  The number of partial
  // rows is conv_stats_rows(p, dtype) (0: this layer's kernel does not produce them); bn_fwd_finalize sums them - the separate reduce pass
  // over the stored map is gone.  conv3x3_halo only so far.
  float* stats;
  // Round 6, conv3x3_halo only (the eval engine's stem): x_planar / y_planar != 0 -> that operand is ROW-CHUNK-PLANAR instead of NHWC: element (b, row, col, c)
  // at ((b * H + row) * (C / 8) + c / 8) * W * 8 + col * 8 + c % 8 - a 16-byte chunk of 8 channels is contiguous along a ROW.  The halo tile's LDS image is
  // chunk-planar, and an LDS-DMA piece writes 64 consecutive slots of ONE chunk plane: from NHWC that is 64 lanes on 64 different 128-byte lines (16 bytes
  // of each: 20 % of the kernel's time went into the halo fetch, tools/probes/variants/conv3x3_halo.diag.patch -DH_NO_HDMA); from this layout the lanes of
  // a piece read row segments of 640 contiguous bytes.  Same size as NHWC; only stem_conv1 / conv3x3_halo produce and consume it.
  int x_planar = 0, y_planar = 0;
};

}  // namespace fsvit_types

// The launchers exist twice: namespace fsvit (16-bit type = __bf16) and namespace fsvit_f16 (16-bit type = _Float16); `dtype` is
// 0 = f32 (exact fp32 MFMA), 1 = the namespace's 16-bit type.  Returns hipError_t as int.
#define FSVIT_DECLARE_CONV_GEMM(NS)                                                                                        \
  namespace NS {                                                                                                           \
  using namespace fsvit_types;                                                                                             \
  int launch_conv_gemm(const ConvGemmParams& p, int dtype, hipStream_t stream);                                            \
  int conv_gemm_route(const ConvGemmParams& p, int dtype); /* 0 conv3x3_halo, 1 gemm256 (whole or row slices), 2 conv_gemm_v2 */ \
  int launch_conv_gemm_v2(const ConvGemmParams& p, int dtype, hipStream_t stream);                                         \
  /* gemm256.hip: 256x256-tile, 8-wave, 4-phase dense 16-bit GEMM for plain [M][K] x [N][K] layers (N >= 192, K % 64 == 0) */ \
  bool gemm256_eligible(const ConvGemmParams& p, int dtype);                                                               \
  int launch_gemm256(const ConvGemmParams& p, hipStream_t stream);                                                         \
  int launch_gemm256_x2(const ConvGemmParams& p, hipStream_t stream); /* dtype 2: two-limb arithmetic on fp32 storage */     \
  /* conv3x3_halo.hip: 3x3 / stride 1 conv with the input tile + halo resident in LDS (stem conv2 / conv3 geometry) */      \
  bool conv3x3_halo_eligible(const ConvGemmParams& p, int dtype);                                                          \
  int launch_conv3x3_halo(const ConvGemmParams& p, hipStream_t stream);                                                    \
  int conv_stats_rows(const ConvGemmParams& p, int dtype); /* partial rows `stats` receives from this layer's kernel, 0 = none */ \
  int conv3x3_halo_stats_rows(const ConvGemmParams& p, int dtype);                                                           \
  int conv_gemm_v2_config(const ConvGemmParams& p); /* which tile configuration launch_conv_gemm_v2 picks */               \
  /* wgrad3x3.hip: the grouped 3x3 conv of the stage-1 Mlp in the two-limb modes (wave = group, weights in registers) */     \
  bool gconv3x3_x2_eligible(const ConvGemmParams& p, int dtype);                                                           \
  int launch_gconv3x3_x2(const ConvGemmParams& p, hipStream_t stream);                                                     \
  }
FSVIT_DECLARE_CONV_GEMM(fsvit)
FSVIT_DECLARE_CONV_GEMM(fsvit_f16)
