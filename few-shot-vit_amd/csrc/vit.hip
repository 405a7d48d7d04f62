// ViT / DeiT helpers (test_phase/models/deit.py):
//  * patchify: NCHW fp32 image -> [B*np][Kp] rows, k = (c, py, px) (the natural order of the conv weight
//    [D][3][p][p], deit.py:93), zero-padded to Kp, so PatchEmbed.proj (:99) is one GEMM.
//  * cls_pos: token 0 of every image = cls_token + pos_embed[0] (deit.py:200-202).
//  * layernorm: x_hat = (x - mean) / sqrt(var + eps) per token (nn.LayerNorm eps 1e-6, deit.py:66,71); gamma / beta
//    are folded into the following Linear by the weight packer.  One wave per token.
//  * final_ln_cls: norm(x)[:, 0] (deit.py:212-213) with explicit gamma / beta -> fp32 features.
#include "fsvit_common.h"
#include "kernels.h"

namespace FSVIT_NS {

template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, T* __restrict__ out, int B, int img, int p, int Kp) {
  const int npw = img / p, np = npw * npw, K = 3 * p * p;
  const size_t total = (size_t)B * np * (Kp / 4);
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int k4 = (int)(idx % (Kp / 4)) * 4;
    const size_t row = idx / (Kp / 4);
    const int pi = (int)(row % np);
    const size_t b = row / np;
    const int py0 = (pi / npw) * p, px0 = (pi % npw) * p;
    f32x4 v;
    if ((p & 3) == 0 && (img & 3) == 0 && k4 + 3 < K) {      // 4 consecutive pixels of one patch row: one 16-byte load
      const int c = k4 / (p * p), r = k4 - c * p * p, py = r / p, px = r - py * p;
      store4<T>(out + row * Kp + k4, *reinterpret_cast<const f32x4*>(x + ((b * 3 + c) * img + py0 + py) * img + px0 + px));
      continue;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k4 + e;
      float val = 0.0f;
      if (k < K) {
        const int c = k / (p * p), r = k - c * p * p, py = r / p, px = r - py * p;
        val = x[((b * 3 + c) * img + py0 + py) * img + px0 + px];
      }
      v[e] = val;
    }
    store4<T>(out + row * Kp + k4, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cls_pos_kernel(const float* __restrict__ cls_plus_pos0, T* __restrict__ tokens, int B, int S, int D) {
  const size_t total = (size_t)B * (D / 4);
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int d4 = (int)(idx % (D / 4)) * 4;
    const size_t b = idx / (D / 4);
    store4<T>(tokens + b * S * D + d4, *reinterpret_cast<const f32x4*>(cls_plus_pos0 + d4));
  }
}

// LayerNorm without affine (gamma / beta are folded into the following Linear): y = (x - mean) * rstd, biased variance as nn.LayerNorm.
// HBM-bound (read + write of the token stream, 24 calls per DeiT forward = 13 % of the step in round 1): a row of D elements is handled
// by G = 16 / 32 / 64 lanes with 16-byte accesses (8 bf16 / 4 fp32 per lane and pass), so a wave normalises 64 / G rows at once and
// every lane is busy - the first version gave a 768-byte DeiT-S row to a whole wave of 8-byte accesses (1.5 of 8 passes populated).
template <typename T, int G>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, T* __restrict__ y, int M, int D, float eps) {
  constexpr int EPL = 16 / sizeof(T);                 // elements per lane and pass
  constexpr int MAXP = 4;                             // passes: D <= G * EPL * MAXP
  const int lane = threadIdx.x & 63, sub = lane % G;
  const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / G) + lane / G;
  const bool rok = row < M;
  const T* xr = x + (size_t)(rok ? row : 0) * D;
  float v[MAXP][EPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int c = (sub + G * i) * EPL;
    if (c < D) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(xr + c);
      const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
      for (int j = 0; j < EPL; ++j) { v[i][j] = to_f32<T>(e[j]); s += v[i][j]; }
    } else {
#pragma unroll
      for (int j = 0; j < EPL; ++j) v[i][j] = 0.f;
    }
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP; ++i)
    if ((sub + G * i) * EPL < D) {
#pragma unroll
      for (int j = 0; j < EPL; ++j) { const float d = v[i][j] - mean; q += d * d; }
    }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.0f / sqrtf(q / (float)D + eps);
  if (!rok) return;
  T* yr = y + (size_t)row * D;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int c = (sub + G * i) * EPL;
    if (c < D) {
      u32x4 raw;
      T* e = reinterpret_cast<T*>(&raw);
#pragma unroll
      for (int j = 0; j < EPL; ++j) e[j] = from_f32<T>((v[i][j] - mean) * rstd);
      *reinterpret_cast<u32x4*>(yr + c) = raw;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(64) void final_ln_cls_kernel(const T* __restrict__ tokens, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ feat, int S, int D, float eps) {
  const int lane = threadIdx.x, b = blockIdx.x;
  const T* xr = tokens + (size_t)b * S * D;                        // token 0 of image b
  f32x4 v[8];
  float s = 0.f;
  const int n4 = D / 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < n4 ? load4<T>(xr + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + 64 * i;
    if (c < n4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + 64 * i;
    if (c < n4) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c * 4), bt = *reinterpret_cast<const f32x4*>(beta + c * 4);
      *reinterpret_cast<f32x4*>(feat + (size_t)b * D + c * 4) = (v[i] - mean) * rstd * g + bt;
    }
  }
}

static unsigned grid_for(size_t total) {
  size_t nb = (total + 255) / 256;
  return (unsigned)(nb > 16384 ? 16384 : (nb < 1 ? 1 : nb));
}

int launch_patchify(const float* x, void* out, int B, int img, int p, int Kp, int dtype, hipStream_t s) {
  if (B <= 0) return 0;
  const size_t total = (size_t)B * (img / p) * (img / p) * (Kp / 4);
  if (dtype == 0) hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, x, (float*)out, B, img, p, Kp);
  else hipLaunchKernelGGL(patchify_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, s, x, (bf16*)out, B, img, p, Kp);
  return (int)hipGetLastError();
}

int launch_cls_pos(const float* cls_plus_pos0, void* tokens, int B, int S, int D, int dtype, hipStream_t s) {
  if (B <= 0) return 0;
  const size_t total = (size_t)B * (D / 4);
  if (dtype == 0) hipLaunchKernelGGL(cls_pos_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, cls_plus_pos0, (float*)tokens, B, S, D);
  else hipLaunchKernelGGL(cls_pos_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, s, cls_plus_pos0, (bf16*)tokens, B, S, D);
  return (int)hipGetLastError();
}

int launch_layernorm(const void* x, void* y, int M, int D, float eps, int dtype, hipStream_t s) {
  if (M <= 0) return 0;
  const int epl = dtype == 0 ? 4 : 8;
  if (D % epl || D > 64 * epl * 4) return (int)hipErrorInvalidValue;
  const int lanes = (D / epl + 3) / 4;                           // lanes a row needs with 4 passes
  const int G = lanes <= 16 ? 16 : (lanes <= 32 ? 32 : 64);
  const int rows_per_wg = 4 * (64 / G);
  dim3 grid((M + rows_per_wg - 1) / rows_per_wg), block(256);
#define FSVIT_LN(TT, GG) hipLaunchKernelGGL((layernorm_kernel<TT, GG>), grid, block, 0, s, (const TT*)x, (TT*)y, M, D, eps)
  if (dtype == 0) { if (G == 16) FSVIT_LN(float, 16); else if (G == 32) FSVIT_LN(float, 32); else FSVIT_LN(float, 64); }
  else { if (G == 16) FSVIT_LN(bf16, 16); else if (G == 32) FSVIT_LN(bf16, 32); else FSVIT_LN(bf16, 64); }
#undef FSVIT_LN
  return (int)hipGetLastError();
}

int launch_final_ln_cls(const void* tokens, const float* gamma, const float* beta, float* feat, int B, int S, int D, float eps, int dtype, hipStream_t s) {
  if (B <= 0) return 0;
  if (D % 4 || D > 2048) return (int)hipErrorInvalidValue;
  if (dtype == 0) hipLaunchKernelGGL(final_ln_cls_kernel<float>, dim3(B), dim3(64), 0, s, (const float*)tokens, gamma, beta, feat, S, D, eps);
  else hipLaunchKernelGGL(final_ln_cls_kernel<bf16>, dim3(B), dim3(64), 0, s, (const bf16*)tokens, gamma, beta, feat, S, D, eps);
  return (int)hipGetLastError();
}

}  // namespace FSVIT_NS
