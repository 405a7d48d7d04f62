// Device-resident dataset transform: the reference's per-image CPU pipeline
//   Image.fromarray(uint8 HxWx3) -> Resize (Pillow BILINEAR, two 8-bit passes) -> CenterCrop -> ToTensor -> Normalize
// (test_phase/datasets/mini_imagenet.py:47-56, tiered_imagenet.py:53-57), executed for a gathered batch of dataset
// indices on the GPU: the uint8 dataset stays in HBM (60 000 x 84x84x3 = 1.27 GB), one workgroup per output image.
// Byte/integer work, bit-exact with Pillow: the 22-bit fixed-point coefficient tables are computed on the host exactly as
// Resample.c does (datasets/transforms.py) and both passes round to uint8 like ImagingResampleHorizontal/Vertical_8bpc.
// HBM-bound: 21 KB in, 77 KB out per image; the image is staged in LDS once, the horizontal pass result lives in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fsvit {

struct TransformParams {
  const uint8_t* images;       // [N][H][W][3]
  const int64_t* index;        // [B]
  const int32_t *xmin_h, *cnt_h, *coef_h;      // [RW], [RW], [RW][ksize_h]
  const int32_t *xmin_v, *cnt_v, *coef_v;      // [RH], [RH], [RH][ksize_v]
  float* out;                  // [B][3][OH][OW]
  int H, W, ksize_h, ksize_v, crop_y0, crop_x0, OH, OW;
  float mean[3], inv255, stdv[3];
};

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__global__ __launch_bounds__(256) void transform_gather_kernel(TransformParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PB = 22;                                 // Pillow PRECISION_BITS = 32 - 8 - 2
  const int t = threadIdx.x;
  const size_t img_bytes = (size_t)p.H * p.W * 3;
  unsigned char* raw = smem;                             // [H][W][3]
  unsigned char* hp = smem + ((img_bytes + 15) & ~(size_t)15);   // [H][OW][3]: horizontal pass, cropped columns only
  const uint8_t* src = p.images + (size_t)p.index[blockIdx.x] * img_bytes;
  if ((img_bytes & 3) == 0 && (((uintptr_t)src) & 3) == 0) {
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d4 = reinterpret_cast<uint32_t*>(raw);
    for (int i = t; i < (int)(img_bytes >> 2); i += 256) d4[i] = s4[i];
  } else {
    for (int i = t; i < (int)img_bytes; i += 256) raw[i] = src[i];
  }
  __syncthreads();
  // horizontal pass: every input row, the OW cropped output columns
  const int nh = p.H * p.OW * 3;
  for (int i = t; i < nh; i += 256) {
    const int c = i % 3, x = (i / 3) % p.OW, r = i / (3 * p.OW);
    const int xo = x + p.crop_x0;
    const int x0 = p.xmin_h[xo], n = p.cnt_h[xo];
    const int32_t* k = p.coef_h + (size_t)xo * p.ksize_h;
    int acc = 1 << (PB - 1);
    for (int j = 0; j < n; ++j) acc += (int)raw[(r * p.W + x0 + j) * 3 + c] * k[j];
    hp[i] = (unsigned char)clip8(acc >> PB);
  }
  __syncthreads();
  // vertical pass + ToTensor + Normalize, NCHW fp32 (x fastest: coalesced stores)
  const int no = 3 * p.OH * p.OW;
  float* out = p.out + (size_t)blockIdx.x * no;
  for (int i = t; i < no; i += 256) {
    const int x = i % p.OW, y = (i / p.OW) % p.OH, c = i / (p.OW * p.OH);
    const int yo = y + p.crop_y0;
    const int y0 = p.xmin_v[yo], n = p.cnt_v[yo];
    const int32_t* k = p.coef_v + (size_t)yo * p.ksize_v;
    int acc = 1 << (PB - 1);
    for (int j = 0; j < n; ++j) acc += (int)hp[((y0 + j) * p.OW + x) * 3 + c] * k[j];
    const float v = (float)clip8(acc >> PB) / 255.0f;                 // ToTensor
    out[i] = (v - p.mean[c]) / p.stdv[c];                              // Normalize (IEEE division, as torch does)
  }
}

int launch_transform_gather(const TransformParams& p, int B, hipStream_t s) {
  if (B <= 0) return 0;
  const size_t img_bytes = (size_t)p.H * p.W * 3;
  const size_t lds = ((img_bytes + 15) & ~(size_t)15) + (size_t)p.H * p.OW * 3;
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)transform_gather_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(transform_gather_kernel, dim3(B), dim3(256), lds, s, p);
  return (int)hipGetLastError();
}

}  // namespace fsvit

int fsvit_set_error(int code, const char* fmt, ...);      // engine.hip (library-internal, C++ linkage)

extern "C" int fsvit_image_transform_gather(const uint8_t* images_dev, int H, int W, const int64_t* index_dev, int B, const int32_t* xmin_h,
                                            const int32_t* cnt_h, const int32_t* coef_h, int ksize_h, const int32_t* xmin_v,
                                            const int32_t* cnt_v, const int32_t* coef_v, int ksize_v, int crop_y0, int crop_x0, int OH, int OW,
                                            const float* mean3_host, const float* std3_host, float* out_dev, void* stream) {
  if (!images_dev || !index_dev || !xmin_h || !cnt_h || !coef_h || !xmin_v || !cnt_v || !coef_v || !mean3_host || !std3_host || !out_dev)
    return fsvit_set_error(-1, "%s", "fsvit_image_transform_gather: null argument");
  if (H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || ksize_h <= 0 || ksize_v <= 0 || crop_y0 < 0 || crop_x0 < 0)
    return fsvit_set_error(-1, "%s", "fsvit_image_transform_gather: bad geometry");
  fsvit::TransformParams p;
  p.images = images_dev; p.index = index_dev;
  p.xmin_h = xmin_h; p.cnt_h = cnt_h; p.coef_h = coef_h;
  p.xmin_v = xmin_v; p.cnt_v = cnt_v; p.coef_v = coef_v;
  p.out = out_dev;
  p.H = H; p.W = W; p.ksize_h = ksize_h; p.ksize_v = ksize_v; p.crop_y0 = crop_y0; p.crop_x0 = crop_x0; p.OH = OH; p.OW = OW;
  for (int c = 0; c < 3; ++c) { p.mean[c] = mean3_host[c]; p.stdv[c] = std3_host[c]; }
  p.inv255 = 1.0f / 255.0f;
  int rc = fsvit::launch_transform_gather(p, B, (hipStream_t)stream);
  if (rc) return fsvit_set_error(rc, "%s", "fsvit_image_transform_gather: launch failed (image too large for LDS?)");
  return 0;
}
