// Attention backward for the Visformer training step (backward of test_phase/models/visformer.py:183-190):
// given q, k, v (the qkv GEMM output, head dim padded to hdp) and d(ctx), recompute P = softmax(scale q k^T) and form
//   dV = P^T dO,  dP = dO V^T,  dS = scale * P o (dP - rowsum(dP o P)),  dQ = dS K,  dK = dS^T Q.
// Attention is 1.2 % of the step's FLOPs: v1 keeps everything of one (image, head) in LDS as fp32 and
// uses plain FMA loops (fp32 in both numerics modes); an MFMA version is queued behind the GEMM-shaped backward work.
#include "fsvit_common.h"
#include "train_kernels.h"

namespace fsvit {

template <typename T>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dctx, T* __restrict__ dqkv,
                                                            int S, int heads, int hd, int hdp, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ld = hd + 1, ls = S + 1;
  float* Q = reinterpret_cast<float*>(smem);
  float* K = Q + S * ld;
  float* V = K + S * ld;
  float* dO = V + S * ld;
  float* P = dO + S * ld;          // [S][S+1]
  float* dP = P + S * ls;          // [S][S+1]  (dP, then dS)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int rowlen = 3 * heads * hdp;
  const T* base = qkv + (size_t)b * S * rowlen + h * hdp;
  const T* dob = dctx + (size_t)b * S * heads * hdp + h * hdp;
  for (int idx = t; idx < S * hd; idx += 256) {
    const int i = idx / hd, d = idx - i * hd;
    const T* r = base + (size_t)i * rowlen + d;
    Q[i * ld + d] = to_f32<T>(r[0]);
    K[i * ld + d] = to_f32<T>(r[heads * hdp]);
    V[i * ld + d] = to_f32<T>(r[2 * heads * hdp]);
    dO[i * ld + d] = to_f32<T>(dob[(size_t)i * heads * hdp + d]);
  }
  __syncthreads();
  // scores and dP
  for (int idx = t; idx < S * S; idx += 256) {
    const int i = idx / S, j = idx - i * S;
    float s = 0.f, dp = 0.f;
    for (int d = 0; d < hd; ++d) { s += Q[i * ld + d] * K[j * ld + d]; dp += dO[i * ld + d] * V[j * ld + d]; }
    P[i * ls + j] = s * scale;
    dP[i * ls + j] = dp;
  }
  __syncthreads();
  // row softmax + dS (one wave per row)
  for (int i = wave; i < S; i += 4) {
    float m = -INFINITY;
    for (int j = lane; j < S; j += 64) m = fmaxf(m, P[i * ls + j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < S; j += 64) { const float e = expf(P[i * ls + j] - m); P[i * ls + j] = e; sum += e; }
    const float inv = 1.0f / wave_sum(sum);
    float dot = 0.f;
    for (int j = lane; j < S; j += 64) { const float p = P[i * ls + j] * inv; P[i * ls + j] = p; dot += p * dP[i * ls + j]; }
    dot = wave_sum(dot);
    for (int j = lane; j < S; j += 64) dP[i * ls + j] = scale * P[i * ls + j] * (dP[i * ls + j] - dot);
  }
  __syncthreads();
  T* dq = dqkv + (size_t)b * S * rowlen + h * hdp;
  for (int idx = t; idx < S * hdp; idx += 256) {
    const int i = idx / hdp, d = idx - i * hdp;
    float gq = 0.f, gk = 0.f, gv = 0.f;
    if (d < hd) {
      for (int j = 0; j < S; ++j) {
        gq += dP[i * ls + j] * K[j * ld + d];          // dQ[i] = sum_j dS[i][j] K[j]
        gk += dP[j * ls + i] * Q[j * ld + d];          // dK[i] = sum_j dS[j][i] Q[j]
        gv += P[j * ls + i] * dO[j * ld + d];          // dV[i] = sum_j P[j][i] dO[j]
      }
    }
    T* r = dq + (size_t)i * rowlen + d;                // padded head dims get exact zeros
    r[0] = from_f32<T>(gq);
    r[heads * hdp] = from_f32<T>(gk);
    r[2 * heads * hdp] = from_f32<T>(gv);
  }
}

int launch_attention_bwd(const void* qkv, const void* dctx, void* dqkv, int B, int S, int heads, int hd, int hdp, float scale, int dtype, hipStream_t s) {
  if (B <= 0) return 0;
  const size_t lds = ((size_t)4 * S * (hd + 1) + (size_t)2 * S * (S + 1)) * sizeof(float);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  hipError_t e;
  if (dtype == 0) {
    e = hipFuncSetAttribute((const void*)attention_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(attention_bwd_kernel<float>, dim3(B * heads), dim3(256), lds, s, (const float*)qkv, (const float*)dctx, (float*)dqkv, S, heads, hd, hdp, scale);
  } else {
    e = hipFuncSetAttribute((const void*)attention_bwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(attention_bwd_kernel<bf16>, dim3(B * heads), dim3(256), lds, s, (const bf16*)qkv, (const bf16*)dctx, (bf16*)dqkv, S, heads, hd, hdp, scale);
  }
  return (int)hipGetLastError();
}

}  // namespace fsvit
