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

// one wave per row; D % 4 == 0, D <= 64 * 4 * 8
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, T* __restrict__ y, int M, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const T* xr = x + (size_t)row * D;
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
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);     // biased variance, as nn.LayerNorm
  T* yr = y + (size_t)row * D;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + 64 * i;
    if (c < n4) store4<T>(yr + c * 4, (v[i] - mean) * rstd);
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
  if (D % 4 || D > 2048) return (int)hipErrorInvalidValue;
  dim3 grid((M + 3) / 4), block(256);
  if (dtype == 0) hipLaunchKernelGGL(layernorm_kernel<float>, grid, block, 0, s, (const float*)x, (float*)y, M, D, eps);
  else hipLaunchKernelGGL(layernorm_kernel<bf16>, grid, block, 0, s, (const bf16*)x, (bf16*)y, M, D, eps);
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
