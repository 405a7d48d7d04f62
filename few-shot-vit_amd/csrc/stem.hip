// Stem helpers of the Visformer ConvBlock (test_phase/models/visformer.py:202-239).
//  * im2col27: NCHW fp32 image -> [B*OH*OW][32] rows of the 3x3/stride-2/pad-1 patch, K order
//    (ky, kx, c), 27 real taps + 5 zeros, so stem.conv1 (:209) and stem.downsample.0 (:216)
//    become K=32 GEMMs on MFMA (conv_gemm).
//  * maxpool2_pos: MaxPool2d(2) (:237) fused with `x + pos_embed1` (:431), NHWC.
#include "fsvit_common.h"
#include "kernels.h"

namespace fsvit {

template <typename T>
__global__ __launch_bounds__(256) void im2col27_kernel(const float* __restrict__ x, T* __restrict__ out,
                                                       int B, int H, int W, int OH, int OW) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  const int M = B * OH * OW;
  if (m >= M) return;
  const int b = m / (OH * OW);
  const int rem = m - b * OH * OW;
  const int oy = rem / OW, ox = rem - oy * OW;
  const float* xb = x + (size_t)b * 3 * H * W;
  float v[32];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * 2 - 1 + ky;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * 2 - 1 + kx;
      const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        v[(ky * 3 + kx) * 3 + c] = ok ? xb[(size_t)c * H * W + (size_t)iy * W + ix] : 0.0f;
    }
  }
#pragma unroll
  for (int k = 27; k < 32; ++k) v[k] = 0.0f;
  T* o = out + (size_t)m * 32;
#pragma unroll
  for (int k = 0; k < 32; k += 4) store4<T>(o + k, f32x4{v[k], v[k + 1], v[k + 2], v[k + 3]});
}

// in [B, 2*OH, 2*OW, C] -> out [B, OH, OW, C] = max over the 2x2 window (+ pos[oy*OW+ox][c])
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_pos_kernel(const T* __restrict__ in, const float* __restrict__ pos,
                                                           T* __restrict__ out, int B, int OH, int OW, int C) {
  const int c4 = C / 4;
  const size_t total = (size_t)B * OH * OW * c4;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int cq = (int)(idx % c4);
    const size_t pix = idx / c4;
    const int ox = (int)(pix % OW);
    const size_t t2 = pix / OW;
    const int oy = (int)(t2 % OH);
    const size_t b = t2 / OH;
    const int W = OW * 2;
    const T* p00 = in + ((b * OH * 2 + oy * 2) * W + ox * 2) * C + cq * 4;
    f32x4 a = load4<T>(p00), b1 = load4<T>(p00 + C), c1 = load4<T>(p00 + (size_t)W * C), d = load4<T>(p00 + (size_t)W * C + C);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaxf(a[e], b1[e]), fmaxf(c1[e], d[e]));
    if (pos) v += *reinterpret_cast<const f32x4*>(pos + (size_t)(oy * OW + ox) * C + cq * 4);
    store4<T>(out + pix * C + cq * 4, v);
  }
}

int launch_im2col27(const float* x, void* out, int B, int H, int W, int OH, int OW, int dtype, hipStream_t s) {
  const int M = B * OH * OW;
  if (M <= 0) return 0;
  dim3 grid((M + 255) / 256), block(256);
  if (dtype == 0) hipLaunchKernelGGL(im2col27_kernel<float>, grid, block, 0, s, x, (float*)out, B, H, W, OH, OW);
  else hipLaunchKernelGGL(im2col27_kernel<bf16>, grid, block, 0, s, x, (bf16*)out, B, H, W, OH, OW);
  return (int)hipGetLastError();
}

int launch_maxpool2_pos(const void* in, const float* pos, void* out, int B, int OH, int OW, int C, int dtype, hipStream_t s) {
  const size_t total = (size_t)B * OH * OW * (C / 4);
  if (total == 0) return 0;
  size_t nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  dim3 grid((unsigned)nb), block(256);
  if (dtype == 0) hipLaunchKernelGGL(maxpool2_pos_kernel<float>, grid, block, 0, s, (const float*)in, pos, (float*)out, B, OH, OW, C);
  else hipLaunchKernelGGL(maxpool2_pos_kernel<bf16>, grid, block, 0, s, (const bf16*)in, pos, (bf16*)out, B, OH, OW, C);
  return (int)hipGetLastError();
}

}  // namespace fsvit
