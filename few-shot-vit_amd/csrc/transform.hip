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

// Round 5: a thread owns one (column, channel) pair of a pass and walks the rows - its coefficients stay in registers, no division in the loops (the
// first version decomposed a flat output index with two div / mod pairs per output and re-read its coefficients from global memory per output:
// 1.5 ms per 12 800-image gather, 4.4 % of the test_few_shot loop, profiles/r05_e2e_gaps.txt) - and ToTensor + Normalize come from a 3 x 256-entry table
// built per workgroup with the SAME IEEE operations (u / 255, then (v - mean) / std), so the bytes -> floats map is unchanged bit for bit.
__global__ __launch_bounds__(256) void transform_gather_kernel(TransformParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PB = 22;                                 // Pillow PRECISION_BITS = 32 - 8 - 2
  constexpr int KMAX = 6;                                // coefficients kept in registers (Pillow bilinear: support 1 / scale, 3 for an upscale)
  const int t = threadIdx.x;
  const size_t img_bytes = (size_t)p.H * p.W * 3;
  const size_t raw_pad = (img_bytes + 15) & ~(size_t)15, hp_pad = ((size_t)p.H * p.OW * 3 + 15) & ~(size_t)15;
  unsigned char* raw = smem;                             // [H][W][3]
  unsigned char* hp = smem + raw_pad;                    // [H][OW][3]: horizontal pass, cropped columns only
  float* lut = reinterpret_cast<float*>(smem + raw_pad + hp_pad);      // [3][256]
  const uint8_t* src = p.images + (size_t)p.index[blockIdx.x] * img_bytes;
  if ((img_bytes & 15) == 0 && (((uintptr_t)src) & 15) == 0) {
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(raw);
    for (int i = t; i < (int)(img_bytes >> 4); i += 256) d4[i] = s4[i];
  } else if ((img_bytes & 3) == 0 && (((uintptr_t)src) & 3) == 0) {
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d4 = reinterpret_cast<uint32_t*>(raw);
    for (int i = t; i < (int)(img_bytes >> 2); i += 256) d4[i] = s4[i];
  } else {
    for (int i = t; i < (int)img_bytes; i += 256) raw[i] = src[i];
  }
  for (int i = t; i < 768; i += 256) {
    const int c = i >> 8;
    const float v = (float)(i & 255) / 255.0f;                        // ToTensor
    lut[i] = (v - p.mean[c]) / p.stdv[c];                              // Normalize (IEEE division, as torch does)
  }
  __syncthreads();
  // horizontal pass: thread = (output column x, channel c), all input rows
  const int rowb = p.W * 3;
  for (int pc = t; pc < p.OW * 3; pc += 256) {
    const int x = pc / 3, c = pc - 3 * x, xo = x + p.crop_x0;
    const int x0 = p.xmin_h[xo], n = p.cnt_h[xo];
    const int32_t* k = p.coef_h + (size_t)xo * p.ksize_h;
    const unsigned char* rp = raw + x0 * 3 + c;
    unsigned char* op = hp + pc;
    if (n <= KMAX) {
      int kr[KMAX];
#pragma unroll
      for (int j = 0; j < KMAX; ++j) kr[j] = j < n ? k[j] : 0;
      for (int r = 0; r < p.H; ++r, rp += rowb, op += p.OW * 3) {
        int acc = 1 << (PB - 1);
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
          if (j < n) acc += (int)rp[3 * j] * kr[j];
        *op = (unsigned char)clip8(acc >> PB);
      }
    } else {
      for (int r = 0; r < p.H; ++r, rp += rowb, op += p.OW * 3) {
        int acc = 1 << (PB - 1);
        for (int j = 0; j < n; ++j) acc += (int)rp[3 * j] * k[j];
        *op = (unsigned char)clip8(acc >> PB);
      }
    }
  }
  __syncthreads();
  // vertical pass + ToTensor + Normalize, NCHW fp32: thread = (channel c, output column x) with x fastest (coalesced stores), all output rows; the row's
  // coefficients are the same for every thread (scalar loads)
  float* out = p.out + (size_t)blockIdx.x * 3 * p.OH * p.OW;
  const int hrow = p.OW * 3;
  for (int pc = t; pc < 3 * p.OW; pc += 256) {
    const int c = pc / p.OW, x = pc - c * p.OW;
    const unsigned char* hb = hp + x * 3 + c;
    const float* lc = lut + 256 * c;
    float* oc = out + (size_t)c * p.OH * p.OW + x;
    for (int y = 0; y < p.OH; ++y) {
      const int yo = y + p.crop_y0;
      const int y0 = p.xmin_v[yo], n = p.cnt_v[yo];
      const int32_t* k = p.coef_v + (size_t)yo * p.ksize_v;
      const unsigned char* hr = hb + y0 * hrow;
      int acc = 1 << (PB - 1);
      for (int j = 0; j < n; ++j) acc += (int)hr[j * hrow] * k[j];
      oc[(size_t)y * p.OW] = lc[clip8(acc >> PB)];
    }
  }
}

int launch_transform_gather(const TransformParams& p, int B, hipStream_t s) {
  if (B <= 0) return 0;
  const size_t img_bytes = (size_t)p.H * p.W * 3;
  const size_t lds = ((img_bytes + 15) & ~(size_t)15) + (((size_t)p.H * p.OW * 3 + 15) & ~(size_t)15) + 768 * sizeof(float);
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
