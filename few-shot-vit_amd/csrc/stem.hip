// Stem helpers of the Visformer ConvBlock (test_phase/models/visformer.py:202-239).
//  * im2col27: NCHW fp32 image -> [B*OH*OW][32] rows of the 3x3/stride-2/pad-1 patch, K order
//    (ky, kx, c), 27 real taps + 5 zeros, so stem.conv1 (:209) and stem.downsample.0 (:216)
//    become K=32 GEMMs on MFMA (conv_gemm).
//  * maxpool2_pos: MaxPool2d(2) (:237) fused with `x + pos_embed1` (:431), NHWC.
#include "fsvit_common.h"
#include "kernels.h"

namespace FSVIT_NS {

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

// Fused im2col + stem conv1 (bf16): NCHW fp32 image -> the [B*OH*OW][32] patch rows (still needed by the downsample tail of the
// conv3 kernel) AND c1 = LeakyReLU(bn1(conv1(x))) [B*OH*OW][64] in one pass (visformer.py:209-210,:218).
// As two launches the pair moves 3.1 GB for 1.03 ms (im2col: scalar gathers, 2.7 TB/s; conv1 as a K = 32 GEMM re-reads the rows it just
// wrote).  Here a 4-wave workgroup owns 8 output rows of one image: the 17 x 80 x 3 input window goes to LDS as bf16 (coalesced float4
// reads, zero border for the pad = 1 taps), and each wave builds the patch rows of a 16-pixel tile DIRECTLY in MFMA operand layout
// (lane (pixel m, lq) gathers taps 8 lq .. 8 lq + 7 with eight 2-byte LDS reads): that one register quartet is both the 16-byte
// chunk of the patch row it stores (64 lanes x 16 B = 1 KB contiguous) and the B operand of the four conv1 MFMAs of the tile
// (D^T = W X^T: a lane ends up with 4 consecutive channels of its pixel -> 8-byte stores).
namespace stem1 {
constexpr int IMG = 80, OH = 40, ROWS = 8;                  // output rows per workgroup
constexpr int IR = 2 * ROWS + 1, LPAD = 4, ICP = IMG + 2 * LPAD;   // input rows held; row = 4 pad elements (column -1 is the last of them), 80 columns, 4 pad: 8-byte aligned groups of 4
constexpr int PLANE = IR * ICP;                             // elements per channel plane
constexpr int NW = 4;
}  // namespace stem1

// c1_planar: c1 is stored row-chunk-planar ([b][row][8 chunks][40 columns][8 channels], conv_gemm.h x_planar) for conv3x3_halo's halo fetch
__global__ __launch_bounds__(stem1::NW * 64) void stem_conv1_kernel(const float* __restrict__ x, bf16* __restrict__ patches, bf16* __restrict__ c1,
                                                                    const bf16* __restrict__ w, const int kw, const float* __restrict__ bias, const int c1_planar) {
  using namespace stem1;
  __shared__ __attribute__((aligned(16))) bf16 tile[3 * PLANE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m = lane & 15, lq = lane >> 4;
  const int b = blockIdx.x / (OH / ROWS), strip = blockIdx.x % (OH / ROWS);
  const int oy0 = strip * ROWS, iy0 = 2 * oy0 - 1;          // first output row, first input row held
  const float* xb = x + (size_t)b * 3 * IMG * IMG;

  // ---- input window -> LDS (bf16, zero border): 3 channels x 17 rows x 20 float4
  for (int i = t; i < 3 * IR * (IMG / 4); i += NW * 64) {
    const int c = i / (IR * (IMG / 4)), r = (i / (IMG / 4)) % IR, q4 = i % (IMG / 4);
    const int iy = iy0 + r;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (iy >= 0 && iy < IMG) v = *reinterpret_cast<const f32x4*>(xb + ((size_t)c * IMG + iy) * IMG + q4 * 4);
    *reinterpret_cast<bf16x4*>(tile + c * PLANE + r * ICP + LPAD + q4 * 4) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  }
  for (int i = t; i < 3 * IR; i += NW * 64)                 // column -1 (with its three pad neighbours)
    *reinterpret_cast<bf16x4*>(tile + (i / IR) * PLANE + (i % IR) * ICP) = bf16x4{(bf16)0.0f, (bf16)0.0f, (bf16)0.0f, (bf16)0.0f};
  // conv1 weights: A fragments of the four 16-channel tiles, k = 8 lq .. +7.  Which channel sits in which MFMA row is free: row
  // 4 g + e of tile 2 p + h carries channel 32 p + 8 g + 4 h + e, so a lane (rows 4 lq + e of both tiles of a pair) ends up with 8
  // CONSECUTIVE channels of its pixel - one 16-byte store per tile pair instead of two 8-byte ones.
  u32x4 wf[4];
  f32x4 bv[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int p = ct >> 1, hh = ct & 1;
    wf[ct] = *reinterpret_cast<const u32x4*>(w + (size_t)(32 * p + 8 * (m >> 2) + 4 * hh + (m & 3)) * kw + 8 * lq);
    bv[ct] = bias ? *reinterpret_cast<const f32x4*>(bias + 32 * p + 8 * lq + 4 * hh) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // this lane's taps: k = 8 lq + j = (ky * 3 + kx) * 3 + c  ->  LDS element offset of the tap relative to the pixel's window origin
  int koff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * lq + j, tap = k / 3, c = k - 3 * tap;
    koff[j] = k < 27 ? c * PLANE + (tap / 3) * ICP + (tap % 3) : -1;
  }
  __syncthreads();

  const size_t prow0 = ((size_t)b * OH + oy0) * OH;         // first output pixel (row-major) of the strip
  for (int tl = wave; tl < ROWS * OH / 16; tl += NW) {      // 20 tiles of 16 pixels
    const int pix = tl * 16 + m, oy = pix / OH, ox = pix - oy * OH;
    const int org = (2 * oy) * ICP + 2 * ox + LPAD - 1;     // window origin: input row 2 oy - 1 = held row 2 oy, column 2 ox - 1 = element LPAD - 1 + 2 ox
    unsigned short e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = koff[j] >= 0 ? __builtin_bit_cast(unsigned short, tile[org + koff[j]]) : (unsigned short)0;
    const u32x4 pf = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16), (unsigned)e[4] | ((unsigned)e[5] << 16),
                      (unsigned)e[6] | ((unsigned)e[7] << 16)};
    *reinterpret_cast<u32x4*>(patches + (prow0 + pix) * 32 + 8 * lq) = pf;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f32x4 a0 = mma_chunk<bf16>(wf[2 * p], pf, bv[2 * p]), a1 = mma_chunk<bf16>(wf[2 * p + 1], pf, bv[2 * p + 1]);
      bf16x8 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {                          // LeakyReLU(0.1) = max(v, 0.1 v)
        o[q] = (bf16)fmaxf(a0[q], 0.1f * a0[q]);
        o[4 + q] = (bf16)fmaxf(a1[q], 0.1f * a1[q]);
      }
      if (c1_planar) *reinterpret_cast<bf16x8*>(c1 + ((((size_t)b * OH + oy0 + oy) * 8 + 4 * p + lq) * OH + ox) * 8) = o;      // 16 lanes = 256 contiguous bytes
      else *reinterpret_cast<bf16x8*>(c1 + (prow0 + pix) * 64 + 32 * p + 8 * lq) = o;
    }
  }
}

bool stem_conv1_supported(int dtype, int img, int C0) { return dtype == 1 && img == stem1::IMG && C0 == 64; }

int launch_stem_conv1(const float* x, void* patches, void* c1, const void* w, int kw, const float* bias, int B, hipStream_t s, int c1_planar) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(stem_conv1_kernel, dim3(B * (stem1::OH / stem1::ROWS)), dim3(stem1::NW * 64), 0, s, x, (bf16*)patches, (bf16*)c1, (const bf16*)w, kw, bias, c1_planar);
  return (int)hipGetLastError();
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

}  // namespace FSVIT_NS
