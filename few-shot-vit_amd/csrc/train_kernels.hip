// Training-path kernels of the Visformer meta-tuning step (meta_tuning_sun_m/train_meta.py:161-177):
// everything the eval path folds away or never needs.  All are HBM-bound elementwise / reduction /
// layout kernels except attention backward; the GEMM-shaped work (forward convs, dgrad, split-K wgrad)
// reuses conv_gemm_v2.  T = storage dtype of activations and activation gradients (fp32 or bf16);
// parameters, parameter gradients, BN statistics and reductions are fp32.
#include <stdlib.h>

#include "fsvit_common.h"
#include "train_kernels.h"

namespace fsvit {

static inline unsigned gs_grid(size_t total, int per_block = 256) {
  size_t nb = (total + per_block - 1) / per_block;
  return (unsigned)(nb > 32768 ? 32768 : (nb < 1 ? 1 : nb));
}
// a two-limb weight word of the `bf16x2` GEMMs (conv_gemm_v2.hip x2_split / engine.py ops.x2_limbs): upper half hi = the value rounded to bf16,
// lower half lo = (value - hi) rounded to bf16.  As an output type of the pack / transpose kernels below (training in the two-limb mode packs
// its weights - and the activation operand of the split-K weight-gradient GEMM - on the device every step).
struct limbw { unsigned w; };
template <> __device__ __forceinline__ limbw from_f32<limbw>(float v) {
  const bf16 h = (bf16)v;
  const bf16 l = (bf16)(v - (float)h);
  return limbw{((unsigned)__builtin_bit_cast(unsigned short, h) << 16) | (unsigned)__builtin_bit_cast(unsigned short, l)};
}
template <> __device__ __forceinline__ void store4<limbw>(limbw* p, f32x4 v) {
  const u32x4 o = {from_f32<limbw>(v[0]).w, from_f32<limbw>(v[1]).w, from_f32<limbw>(v[2]).w, from_f32<limbw>(v[3]).w};
  *reinterpret_cast<u32x4*>(p) = o;
}

#define GS_LOOP(idx, total) for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (total); idx += (size_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------------------------------------ weights
// PyTorch conv weight W[O][Ig][KH][KW] (O = groups * Ng) -> packed [groups][Npad][Kw] (see conv_gemm.h).
//  mode 0 (forward): row n = output channel, k = (ky*KW + kx) * Ig + i
//  mode 1 (dgrad):   row c = input channel of the group, k = (ky*KW + kx) * Ng + n with the taps FLIPPED, so a
//                    stride-1 conv of dY with it is the transposed convolution
//  mode 2 (k2s2 dgrad): row r = (ky*KW + kx) * Ig + i, k = n (no flip): dY[m] . W[:, i, ky, kx] for the non-overlapping patch conv
// Head-dim padding (qkv rows / proj columns, visformer.py:172-177): logical index j < real maps to
// (j / hd) * hdp + j % hd in the padded dimension (hd == hdp: identity).
// modes 3 / 4: fragment images of the row-wise training Mlp (mlp_train.hip).  The logical matrix A[R = rows_pad][K = Kw] has element (row, k) =
// w[row * O + k * Ig] (O / Ig carry the two strides, so a transposed view is a stride swap); the image is the sequence of 1 KiB MFMA A-operand
// fragments (lane (r, kh): 8 consecutive k of one row) in the order the kernel consumes them:
//   mode 3 (first GEMM: rows = hidden units, k = channels):   fragment (chunk j, step s): row 32 j + hperm(r), k = 32 (s >> 1) + 16 kh + 8 (s & 1) + q
//   mode 4 (second GEMM: rows = output channels, k = hidden): fragment (chunk j, tile ct, half s2): row 32 ct + hperm(r), k = 32 j + 16 kh + 8 s2 + q
// hperm(rho) = 16 ((rho >> 2) & 1) + 4 (rho >> 3) + (rho & 3): the 32 x 32 result's lane then holds 16 consecutive units of its token.
__device__ __forceinline__ float frag_image_elem(const float* __restrict__ w, int mode, size_t idx, int R, int K, int rs, int ks) {
  const int q = (int)(idx & 7), lane = (int)((idx >> 3) & 63), r = lane & 31, kh = lane >> 5;
  const int f = (int)(idx >> 9);
  const int hp = 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
  int row, k;
  if (mode == 3) {
    const int nks = K / 16, j = f / nks, s = f % nks;
    row = 32 * j + hp;
    k = 32 * (s >> 1) + 16 * kh + 8 * (s & 1) + q;
  } else {
    const int npc = R / 16, j = f / npc, rem = f % npc, ct = rem >> 1, s2 = rem & 1;
    row = 32 * ct + hp;
    k = 32 * j + 16 * kh + 8 * s2 + q;
  }
  return w[(size_t)row * rs + (size_t)k * ks];
}
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int O, int Ig, int KH, int KW,
                                                          int groups, int mode, int rows_pad, int Kw, int hd_rows, int hdp_rows,
                                                          int hd_cols, int hdp_cols) {
  const int Ng = O / groups;
  const size_t total = (size_t)groups * rows_pad * Kw;
  GS_LOOP(idx, total) {
    const int k = (int)(idx % Kw);
    const size_t t2 = idx / Kw;
    const int r = (int)(t2 % rows_pad), g = (int)(t2 / rows_pad);
    float v = 0.0f;
    if (mode >= 3) { out[idx] = from_f32<T>(frag_image_elem(w, mode, idx, rows_pad, Kw, O, Ig)); continue; }
    if (mode == 2) {
      if (r < KH * KW * Ig && k < Ng) {
        const int tap = r / Ig, ch = r % Ig;
        v = w[(((size_t)(g * Ng + k) * Ig + ch) * KH + tap / KW) * KW + tap % KW];
      }
      out[idx] = from_f32<T>(v);
      continue;
    }
    const int nin = mode == 0 ? Ig : Ng;            // channels per tap in the K dimension
    // undo the head padding of the K (column) dimension: only for 1x1 layers
    int kk = k;
    bool ok = true;
    if (hdp_cols != hd_cols) { const int y = k / hdp_cols, zz = k % hdp_cols; ok = zz < hd_cols; kk = y * hd_cols + zz; }
    int rr = r;
    if (hdp_rows != hd_rows) { const int y = r / hdp_rows, zz = r % hdp_rows; ok = ok && zz < hd_rows; rr = y * hd_rows + zz; }
    const int nrows = mode == 0 ? Ng : Ig;
    if (ok && rr < nrows && kk < KH * KW * nin) {
      const int tap = kk / nin, ch = kk % nin;
      const int ky = tap / KW, kx = tap % KW;
      if (mode == 0) v = w[(((size_t)(g * Ng + rr) * Ig + ch) * KH + ky) * KW + kx];
      else v = w[(((size_t)(g * Ng + ch) * Ig + rr) * KH + (KH - 1 - ky)) * KW + (KW - 1 - kx)];
    }
    out[idx] = from_f32<T>(v);
  }
}

// All weight packs of a training step in one launch (74 layers x (forward | transposed for the data gradient)): the job table travels as a
// kernel argument (<= 40 jobs of 64 bytes per launch), blockIdx.y = job.
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const PackJobs jobs) {
  const PackJob j = jobs.job[blockIdx.y];
  const int Ng = j.O / j.groups;
  const size_t total = (size_t)j.groups * j.rows_pad * j.Kw;
  T* out = reinterpret_cast<T*>(j.out);
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int k = (int)(idx % j.Kw);
    const size_t t2 = idx / j.Kw;
    const int r = (int)(t2 % j.rows_pad), g = (int)(t2 / j.rows_pad);
    float v = 0.0f;
    if (j.mode >= 3) { out[idx] = from_f32<T>(frag_image_elem(j.w, j.mode, idx, j.rows_pad, j.Kw, j.O, j.Ig)); continue; }
    if (j.mode == 2) {
      if (r < j.KH * j.KW * j.Ig && k < Ng) {
        const int tap = r / j.Ig, ch = r % j.Ig;
        v = j.w[(((size_t)(g * Ng + k) * j.Ig + ch) * j.KH + tap / j.KW) * j.KW + tap % j.KW];
      }
      out[idx] = from_f32<T>(v);
      continue;
    }
    const int nin = j.mode == 0 ? j.Ig : Ng;
    int kk = k;
    bool ok = true;
    if (j.hdp_cols != j.hd_cols) { const int y = k / j.hdp_cols, zz = k % j.hdp_cols; ok = zz < j.hd_cols; kk = y * j.hd_cols + zz; }
    int rr = r;
    if (j.hdp_rows != j.hd_rows) { const int y = r / j.hdp_rows, zz = r % j.hdp_rows; ok = ok && zz < j.hd_rows; rr = y * j.hd_rows + zz; }
    const int nrows = j.mode == 0 ? Ng : j.Ig;
    if (ok && rr < nrows && kk < j.KH * j.KW * nin) {
      const int tap = kk / nin, ch = kk % nin;
      const int ky = tap / j.KW, kx = tap % j.KW;
      if (j.mode == 0) v = j.w[(((size_t)(g * Ng + rr) * j.Ig + ch) * j.KH + ky) * j.KW + kx];
      else v = j.w[(((size_t)(g * Ng + ch) * j.Ig + rr) * j.KH + (j.KH - 1 - ky)) * j.KW + (j.KW - 1 - kx)];
    }
    out[idx] = from_f32<T>(v);
  }
}

// split-K wgrad result Y[Ng][splits * Kc] (fp32, Kc = KH*KW*Ig padded to Kc_pad) of ONE group -> dW[O][Ig][KH][KW] (overwrite)
__global__ __launch_bounds__(256) void wgrad_finalize_kernel(const float* __restrict__ y, float* __restrict__ dw, int Ng, int Ig, int KH, int KW,
                                                             int g, int splits, int Kc_pad, int hd_rows, int hdp_rows, int hd_cols, int hdp_cols) {
  const size_t total = (size_t)Ng * Ig * KH * KW;
  GS_LOOP(idx, total) {
    const int kx = (int)(idx % KW);
    size_t t2 = idx / KW;
    const int ky = (int)(t2 % KH); t2 /= KH;
    const int i = (int)(t2 % Ig);
    const int n = (int)(t2 / Ig);
    int k = (ky * KW + kx) * Ig + i;
    if (hdp_cols != hd_cols) k = (k / hd_cols) * hdp_cols + k % hd_cols;
    int r = n;
    if (hdp_rows != hd_rows) r = (n / hd_rows) * hdp_rows + n % hd_rows;
    const float* src = y + (size_t)r * splits * Kc_pad + k;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // four slabs in flight (a single dependent chain of 4-byte loads ran at 0.5 TB/s)
    int sp = 0;
    for (; sp + 4 <= splits; sp += 4) {
      s0 += src[(size_t)sp * Kc_pad];
      s1 += src[(size_t)(sp + 1) * Kc_pad];
      s2 += src[(size_t)(sp + 2) * Kc_pad];
      s3 += src[(size_t)(sp + 3) * Kc_pad];
    }
    for (; sp < splits; ++sp) s0 += src[(size_t)sp * Kc_pad];
    dw[(((size_t)(g * Ng + n) * Ig + i) * KH + ky) * KW + kx] = (s0 + s1) + (s2 + s3);
  }
}

// The same for a plain 1x1 layer without head-dim padding (dW[n][i] = sum_sp y[n][sp * Kc_pad + i], Ig % 4 == 0): 16-byte accesses and four
// independent partial sums per thread - the generic kernel's serial scalar loop ran at ~0.5 TB/s over the 64 MB of partials a layer has.
__global__ __launch_bounds__(256) void wgrad_finalize_1x1_kernel(const float* __restrict__ y, float* __restrict__ dw, int Ng, int Ig, int splits, int Kc_pad) {
  const int i4n = Ig / 4;
  const size_t total = (size_t)Ng * i4n;
  GS_LOOP(idx, total) {
    const int i = (int)(idx % i4n) * 4, n = (int)(idx / i4n);
    const float* src = y + (size_t)n * splits * Kc_pad + i;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int sp = 0;
    for (; sp + 4 <= splits; sp += 4) {
      s0 += *reinterpret_cast<const f32x4*>(src + (size_t)sp * Kc_pad);
      s1 += *reinterpret_cast<const f32x4*>(src + (size_t)(sp + 1) * Kc_pad);
      s2 += *reinterpret_cast<const f32x4*>(src + (size_t)(sp + 2) * Kc_pad);
      s3 += *reinterpret_cast<const f32x4*>(src + (size_t)(sp + 3) * Kc_pad);
    }
    for (; sp < splits; ++sp) s0 += *reinterpret_cast<const f32x4*>(src + (size_t)sp * Kc_pad);
    *reinterpret_cast<f32x4*>(dw + (size_t)n * Ig + i) = (s0 + s1) + (s2 + s3);
  }
}

// Grouped conv computed as ONE dense split-K GEMM (all cross-group products included, only the diagonal blocks are kept):
// y[groups*Ng][splits * Kc_pad] with k = (ky*KW + kx) * (groups*Ig) + g*Ig + i  ->  dW[groups*Ng][Ig][KH][KW]
__global__ __launch_bounds__(256) void wgrad_finalize_dense_kernel(const float* __restrict__ y, float* __restrict__ dw, int Ng, int Ig, int KH, int KW,
                                                                   int groups, int splits, int Kc_pad) {
  const size_t total = (size_t)groups * Ng * Ig * KH * KW;
  GS_LOOP(idx, total) {
    const int kx = (int)(idx % KW);
    size_t t2 = idx / KW;
    const int ky = (int)(t2 % KH); t2 /= KH;
    const int i = (int)(t2 % Ig);
    const int o = (int)(t2 / Ig), g = o / Ng;
    const int k = (ky * KW + kx) * (groups * Ig) + g * Ig + i;
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += y[(size_t)o * splits * Kc_pad + (size_t)sp * Kc_pad + k];
    dw[idx] = s;
  }
}

// All deferred finalizes of a backward pass: blockIdx.y = job, grid-stride over x inside the job (same arithmetic, same summation order as the
// per-layer kernels above and wgrad3x3_finalize_kernel)
__global__ __launch_bounds__(256) void wgrad_finalize_multi_kernel(const FinJobs jobs) {
  const FinJob j = jobs.job[blockIdx.y];
  const float* __restrict__ y = j.y;
  float* __restrict__ dw = j.dw;
  const int splits = j.splits, Kc_pad = j.Kc_pad;
  const size_t gstep = (size_t)gridDim.x * 256, g0 = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j.kind == 1) {
    const int i4n = j.Ig / 4;
    const size_t total = (size_t)j.Ng * i4n;
    for (size_t idx = g0; idx < total; idx += gstep) {
      const int i = (int)(idx % i4n) * 4, n = (int)(idx / i4n);
      const float* src = y + (size_t)n * splits * Kc_pad + i;
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
      int sp = 0;
      for (; sp + 4 <= splits; sp += 4) {
        s0 += *reinterpret_cast<const f32x4*>(src + (size_t)sp * Kc_pad);
        s1 += *reinterpret_cast<const f32x4*>(src + (size_t)(sp + 1) * Kc_pad);
        s2 += *reinterpret_cast<const f32x4*>(src + (size_t)(sp + 2) * Kc_pad);
        s3 += *reinterpret_cast<const f32x4*>(src + (size_t)(sp + 3) * Kc_pad);
      }
      for (; sp < splits; ++sp) s0 += *reinterpret_cast<const f32x4*>(src + (size_t)sp * Kc_pad);
      *reinterpret_cast<f32x4*>(dw + (size_t)n * j.Ig + i) = (s0 + s1) + (s2 + s3);
    }
  } else if (j.kind == 0) {
    const size_t total = (size_t)j.Ng * j.Ig * j.KH * j.KW;
    for (size_t idx = g0; idx < total; idx += gstep) {
      const int kx = (int)(idx % j.KW);
      size_t t2 = idx / j.KW;
      const int ky = (int)(t2 % j.KH); t2 /= j.KH;
      const int i = (int)(t2 % j.Ig);
      const int n = (int)(t2 / j.Ig);
      int k = (ky * j.KW + kx) * j.Ig + i;
      if (j.hdp_cols != j.hd_cols) k = (k / j.hd_cols) * j.hdp_cols + k % j.hd_cols;
      int r = n;
      if (j.hdp_rows != j.hd_rows) r = (n / j.hd_rows) * j.hdp_rows + n % j.hd_rows;
      const float* src = y + (size_t)r * splits * Kc_pad + k;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int sp = 0;
      for (; sp + 4 <= splits; sp += 4) {
        s0 += src[(size_t)sp * Kc_pad];
        s1 += src[(size_t)(sp + 1) * Kc_pad];
        s2 += src[(size_t)(sp + 2) * Kc_pad];
        s3 += src[(size_t)(sp + 3) * Kc_pad];
      }
      for (; sp < splits; ++sp) s0 += src[(size_t)sp * Kc_pad];
      dw[(((size_t)(j.g * j.Ng + n) * j.Ig + i) * j.KH + ky) * j.KW + kx] = (s0 + s1) + (s2 + s3);
    }
  } else if (j.kind == 2) {
    const int groups = j.g;
    const size_t total = (size_t)groups * j.Ng * j.Ig * j.KH * j.KW;
    for (size_t idx = g0; idx < total; idx += gstep) {
      const int kx = (int)(idx % j.KW);
      size_t t2 = idx / j.KW;
      const int ky = (int)(t2 % j.KH); t2 /= j.KH;
      const int i = (int)(t2 % j.Ig);
      const int o = (int)(t2 / j.Ig), g = o / j.Ng;
      const int k = (ky * j.KW + kx) * (groups * j.Ig) + g * j.Ig + i;
      float s = 0.f;
      for (int sp = 0; sp < splits; ++sp) s += y[(size_t)o * splits * Kc_pad + (size_t)sp * Kc_pad + k];
      dw[idx] = s;
    }
  } else {                                                   // wgrad3x3 partials [split][job][tap][32][32]
    constexpr int JOB = 9 * 32 * 32;
    const int grouped = j.g, njobs = j.Kc_pad, Ig = j.Ig, total = njobs * JOB;
    for (size_t i4 = g0; i4 < (size_t)(total / 4); i4 += gstep) {
      const int idx = (int)i4 * 4;
      const int c = idx & 31, n = (idx >> 5) & 31, tp = (idx >> 10) % 9, job = idx / JOB;
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
      int sp = 0;
      for (; sp + 4 <= splits; sp += 4) {
        s0 += *reinterpret_cast<const f32x4*>(y + (size_t)sp * total + idx);
        s1 += *reinterpret_cast<const f32x4*>(y + (size_t)(sp + 1) * total + idx);
        s2 += *reinterpret_cast<const f32x4*>(y + (size_t)(sp + 2) * total + idx);
        s3 += *reinterpret_cast<const f32x4*>(y + (size_t)(sp + 3) * total + idx);
      }
      for (; sp < splits; ++sp) s0 += *reinterpret_cast<const f32x4*>(y + (size_t)sp * total + idx);
      const f32x4 sv = (s0 + s1) + (s2 + s3);
      const int o = grouped ? job * 32 + n : (job & 3) * 32 + n;
      const int ig = grouped ? c : (job >> 3) * 64 + ((job >> 2) & 1) * 32 + c;
#pragma unroll
      for (int e = 0; e < 4; ++e) dw[((size_t)o * Ig + ig + e) * 9 + tp] = sv[e];
    }
  }
}

__global__ __launch_bounds__(256) void droppath_scales_kernel(const float* __restrict__ masks, float* __restrict__ scales, int n_img, const DropKeep keep) {
  const float inv = keep.inv[blockIdx.y];
  for (int b = blockIdx.x * 256 + threadIdx.x; b < n_img; b += gridDim.x * 256) scales[(size_t)blockIdx.y * n_img + b] = masks[(size_t)blockIdx.y * n_img + b] * inv;
}

// ------------------------------------------------------------------------------------------------ transposes for wgrad
// in [M][ld] (columns c0 .. c0+ncols) -> out [ncols][Mpad], zero for m >= M.  32x32 LDS tile transpose.
template <typename T, typename TO = T>
__global__ __launch_bounds__(256) void transpose_cols_kernel(const T* __restrict__ in, TO* __restrict__ out, int M, int ld, int c0, int ncols, int Mpad) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  const int mt = blockIdx.x * 32, ct = blockIdx.y * 32;
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int m = mt + ty + r, c = ct + tx;
    tile[ty + r][tx] = (m < M && c < ncols) ? to_f32<T>(in[(size_t)m * ld + c0 + c]) : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int c = ct + ty + r, m = mt + tx;
    if (c < ncols && m < Mpad) out[(size_t)c * Mpad + m] = from_f32<TO>(tile[tx][ty + r]);
  }
}

// x NHWC [B,H,W,ld] (channels c0 .. c0+C) -> out [(ky*KW+kx)*C + c][Mpad], m = (b, oy, ox); zero outside the image / m >= M
template <typename T, typename TO = T>
__global__ __launch_bounds__(256) void im2col_t_kernel(const T* __restrict__ x, TO* __restrict__ out, int B, int H, int W, int ld, int c0, int C,
                                                       int KH, int KW, int stride, int pad, int OH, int OW, int Mpad) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int mt = blockIdx.x * 32, ct = blockIdx.y * 32, tap = blockIdx.z;
  const int ky = tap / KW, kx = tap % KW;
  const int M = B * OH * OW;
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int m = mt + ty + r, c = ct + tx;
    float v = 0.0f;
    if (m < M && c < C) {
      const int b = m / (OH * OW), rem = m % (OH * OW), oy = rem / OW, ox = rem % OW;
      const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = to_f32<T>(x[((size_t)(b * H + iy) * W + ix) * ld + c0 + c]);
    }
    tile[ty + r][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int c = ct + ty + r, m = mt + tx;
    if (c < C && m < Mpad) out[(size_t)(tap * C + c) * Mpad + m] = from_f32<TO>(tile[tx][ty + r]);
  }
}

// Vectorised 64 x 64 tile transposes (8/16-byte accesses on both sides): require ld, c0, ncols, Mpad % 4 == 0.
// `row_src(m)` gives the source row of output column m (or -1 for zero fill).
template <typename T, typename TO, typename RowFn>
__device__ __forceinline__ void transpose_tile64(const T* __restrict__ in, TO* __restrict__ out, int ld, int c0, int ncols, int Mpad, size_t out_row0,
                                                 int mt, int ct, RowFn row_src) {
  __shared__ float tile[64][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // 16 channel quads x 16 rows
#pragma unroll
  for (int r = 0; r < 64; r += 16) {
    const int m = mt + ty + r, c = ct + tx * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < ncols) {
      const long src = row_src(m);
      if (src >= 0) v = load4<T>(in + (size_t)src * ld + c0 + c);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[tx * 4 + e][ty + r] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 64; r += 16) {
    const int c = ct + ty + r, m = mt + tx * 4;
    if (c < ncols && m < Mpad) {
      const f32x4 v = {tile[ty + r][tx * 4], tile[ty + r][tx * 4 + 1], tile[ty + r][tx * 4 + 2], tile[ty + r][tx * 4 + 3]};
      store4<TO>(out + (out_row0 + c) * (size_t)Mpad + m, v);
    }
  }
}

template <typename T, typename TO = T>
__global__ __launch_bounds__(256) void transpose_cols_v4_kernel(const T* __restrict__ in, TO* __restrict__ out, int M, int ld, int c0, int ncols, int Mpad) {
  transpose_tile64<T, TO>(in, out, ld, c0, ncols, Mpad, 0, blockIdx.x * 64, blockIdx.y * 64, [&](int m) -> long { return m < M ? m : -1; });
}

// x NHWC [B,H,W,ld] (channels c0 .. c0+C) -> out [(ky*KW+kx)*C + c][Mpad], m = (b, oy, ox); zero outside the image / m >= M
template <typename T, typename TO = T>
__global__ __launch_bounds__(256) void im2col_t_v4_kernel(const T* __restrict__ x, TO* __restrict__ out, int B, int H, int W, int ld, int c0, int C,
                                                          int KH, int KW, int stride, int pad, int OH, int OW, int Mpad) {
  const int tap = blockIdx.z, ky = tap / KW, kx = tap % KW, M = B * OH * OW;
  transpose_tile64<T, TO>(x, out, ld, c0, C, Mpad, (size_t)tap * C, blockIdx.x * 64, blockIdx.y * 64, [&](int m) -> long {
    if (m >= M) return -1;
    const int b = m / (OH * OW), rem = m % (OH * OW), oy = rem / OW, ox = rem % OW;
    const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
    return ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? ((long)(b * H + iy) * W + ix) : -1;
  });
}

// k2s2 dgrad scatter: g [B*OH*OW][4*C] with k = (ky*2+kx)*C + c  ->  dx [B, 2*OH, 2*OW, C]
template <typename T>
__global__ __launch_bounds__(256) void unpatch2_kernel(const T* __restrict__ g, T* __restrict__ dx, int B, int OH, int OW, int C) {
  const size_t total = (size_t)B * OH * OW * 4 * C;
  GS_LOOP(idx, total) {
    const int c = (int)(idx % C);
    size_t t2 = idx / C;
    const int tap = (int)(t2 % 4); t2 /= 4;
    const int ox = (int)(t2 % OW); t2 /= OW;
    const int oy = (int)(t2 % OH);
    const size_t b = t2 / OH;
    dx[((b * 2 * OH + 2 * oy + (tap >> 1)) * 2 * OW + 2 * ox + (tap & 1)) * C + c] = g[idx];
  }
}

// ------------------------------------------------------------------------------------------------ BatchNorm (train)
// partial[blk][2][C]: per-block sums of z and z^2 (or of dy and dy*xhat in the backward form) over an interleaved row set.
// 256 threads = LC channel lanes (4 channels each, 8/16-byte loads) x R row lanes; rows r = blk*R + rl, += gridDim*R.
// A lane owns V = 16 / sizeof(T) channels (16-byte loads) and walks its rows four at a time with the loads issued back to back: the first
// version (4 channels per lane, one dependent 8-byte load per iteration, 78 iterations per thread) reached 0.5 .. 0.9 TB/s on the 82 / 164 MB
// maps of stage 1.
// Forward form with add_b != nullptr: the residual add in front of the BatchNorm rides along - v = add_a + scale[row / rows_per_img] * add_b
// (scale == nullptr: 1) is stored to `a` (rounded to T, as add_scaled_kernel would) and the sums are taken over the stored values.
template <typename T, bool BWD, int V>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const T* a, const T* __restrict__ z, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, float* __restrict__ partial, int M, int C,
                                                        const T* __restrict__ add_a = nullptr, const T* __restrict__ add_b = nullptr,
                                                        const float* __restrict__ add_scale = nullptr, int rows_per_img = 1,
                                                        const float* __restrict__ act_sa = nullptr, const float* __restrict__ act_sb = nullptr) {
  constexpr int Q = V / 4;                               // V channels per lane (16 / sizeof(T), or 4 when C is not a multiple of that) = Q f32x4 groups
  __shared__ f32x4 red[2][Q][256];
  const int lanesC = C / V;
  const int LC = lanesC < 256 ? lanesC : 256, R = 256 / LC;
  const int cl = threadIdx.x % LC, rl = threadIdx.x / LC;
  const bool live = rl < R;
  auto ld = [&](const T* p, f32x4 (&o)[Q]) {
    if constexpr (V == 4) o[0] = load4<T>(p);
    else {
      const bf16x8 h = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e >> 2][e & 3] = (float)h[e];
    }
  };
  for (int cv = cl; cv < lanesC; cv += LC) {            // (one pass for C <= 2048)
    const int c = cv * V;
    // Backward form with act_sa != nullptr: `a` is the gradient BEHIND a LeakyReLU(0.1) that followed this BatchNorm (y = act_sa * z + act_sb);
    // the gradient at the BatchNorm output, dy = a * lrelu'(y), is formed on the fly (the stem's bn_act_bwd pass and its 328 MB map are gone)
    f32x4 s0[Q], s1[Q], mu[Q], is[Q], asa[Q], asb[Q];
    const bool act = BWD && act_sa != nullptr;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      s0[q] = s1[q] = mu[q] = is[q] = asa[q] = asb[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (BWD) { mu[q] = *reinterpret_cast<const f32x4*>(mean + c + 4 * q); is[q] = *reinterpret_cast<const f32x4*>(invstd + c + 4 * q); }
      if (act) { asa[q] = *reinterpret_cast<const f32x4*>(act_sa + c + 4 * q); asb[q] = *reinterpret_cast<const f32x4*>(act_sb + c + 4 * q); }
    }
    auto lrelu_grad = [&](f32x4& d, const f32x4& zz, int q) {
      const f32x4 yv = zz * asa[q] + asb[q];
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = yv[e] > 0.f ? d[e] : 0.1f * d[e];
    };
    if (live) {
      const size_t step = (size_t)gridDim.x * R;
      size_t m = (size_t)blockIdx.x * R + rl;
      auto fin_add = [&](size_t row, f32x4 (&v)[Q], const f32x4 (&vb)[Q]) {     // v = add_a + s * add_b -> stored to a[row], v := the stored (rounded) values
        const float sc = add_scale ? add_scale[row / rows_per_img] : 1.0f;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          v[q] += vb[q] * sc;
          store4<T>(const_cast<T*>(a) + row * C + c + 4 * q, v[q]);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[q][e] = to_f32<T>(from_f32<T>(v[q][e]));
          s0[q] += v[q];
          s1[q] += v[q] * v[q];
        }
      };
      if (!BWD && add_b) {
        for (; m + step < (size_t)M; m += 2 * step) {
          f32x4 va[2][Q], vb[2][Q];
#pragma unroll
          for (int u = 0; u < 2; ++u) { ld(add_a + (m + u * step) * C + c, va[u]); ld(add_b + (m + u * step) * C + c, vb[u]); }
#pragma unroll
          for (int u = 0; u < 2; ++u) fin_add(m + u * step, va[u], vb[u]);
        }
        for (; m < (size_t)M; m += step) {
          f32x4 va[Q], vb[Q];
          ld(add_a + m * C + c, va);
          ld(add_b + m * C + c, vb);
          fin_add(m, va, vb);
        }
      }
      for (; m + 3 * step < (size_t)M; m += 4 * step) {
        f32x4 va[4][Q], vz[4][Q];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          ld(a + (m + u * step) * C + c, va[u]);
          if (BWD) ld(z + (m + u * step) * C + c, vz[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            if (act) lrelu_grad(va[u][q], vz[u][q], q);
            s0[q] += va[u][q];
            if (BWD) s1[q] += va[u][q] * (vz[u][q] - mu[q]) * is[q];
            else s1[q] += va[u][q] * va[u][q];
          }
      }
      for (; m < (size_t)M; m += step) {
        f32x4 va[Q], vz[Q];
        ld(a + m * C + c, va);
        if (BWD) ld(z + m * C + c, vz);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          if (act) lrelu_grad(va[q], vz[q], q);
          s0[q] += va[q];
          if (BWD) s1[q] += va[q] * (vz[q] - mu[q]) * is[q];
          else s1[q] += va[q] * va[q];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) { red[0][q][threadIdx.x] = s0[q]; red[1][q][threadIdx.x] = s1[q]; }
    __syncthreads();
    if (rl == 0) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        for (int r = 1; r < R; ++r) { s0[q] += red[0][q][r * LC + cl]; s1[q] += red[1][q][r * LC + cl]; }
        *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * 2 + 0) * C + c + 4 * q) = s0[q];
        *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * 2 + 1) * C + c + 4 * q) = s1[q];
      }
    }
    __syncthreads();
  }
}

// sum of the per-block partials of one channel: one wave per channel (4 channels per block), fp64 accumulation.  (8 threads per channel made the two
// finalize kernels 20 us each - 64 dependent loads per thread - times 294 launches per step.)
__device__ __forceinline__ void bn_partial_sums(const float* __restrict__ partial, int nblk, int C, int c, int sub, double& s0, double& s1) {
  s0 = 0.0; s1 = 0.0;
  if (c < C)
    for (int b = sub; b < nblk; b += 64) { s0 += partial[((size_t)b * 2 + 0) * C + c]; s1 += partial[((size_t)b * 2 + 1) * C + c]; }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
}

// forward finalize: mean / invstd (biased variance), scale/shift for the apply pass, running-stat update (momentum, unbiased var)
__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(const float* __restrict__ partial, int nblk, int M, int C, float eps, float momentum,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ sa, float* __restrict__ sb) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), sub = threadIdx.x & 63;
  double s0, s1;
  bn_partial_sums(partial, nblk, C, c, sub, s0, s1);
  if (c >= C || sub) return;
  const double mu = s0 / M;
  double var = s1 / M - mu * mu;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = (float)mu;
  invstd[c] = is;
  sa[c] = gamma[c] * is;
  sb[c] = beta[c] - (float)mu * gamma[c] * is;
  if (running_mean) {
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mu;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(var * ((double)M / (double)(M - 1)));
  }
}

// frozen BatchNorm inside a training step (utils.freeze_bn, meta_tuning_sun_m/train_meta.py:156-157): the running statistics normalise, nothing is updated
__global__ __launch_bounds__(256) void bn_frozen_coeffs_kernel(int C, float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                                               float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ sa, float* __restrict__ sb) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float is = (float)(1.0 / sqrt((double)running_var[c] + (double)eps));
  mean[c] = running_mean[c];
  invstd[c] = is;
  sa[c] = gamma[c] * is;
  sb[c] = beta[c] - running_mean[c] * gamma[c] * is;
}

// backward finalize: dgamma = sum(dy * xhat), dbeta = sum(dy); coefficients of dz = ca * dy + cb + cc * xhat  (frozen statistics: dz = ca * dy)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int M, int C, const float* __restrict__ gamma,
                                                              const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ ca, float* __restrict__ cb, float* __restrict__ cc, int frozen) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), sub = threadIdx.x & 63;
  double s0, s1;
  bn_partial_sums(partial, nblk, C, c, sub, s0, s1);
  if (c >= C || sub) return;
  dbeta[c] = (float)s0;
  dgamma[c] = (float)s1;
  const float gi = gamma[c] * invstd[c];
  ca[c] = gi;
  cb[c] = frozen ? 0.f : (float)(-gi * s0 / M);
  cc[c] = frozen ? 0.f : (float)(-gi * s1 / M);
}

// y = act(sa[c] * z + sb[c] (+ res))      act: 0 none, 2 LeakyReLU(0.1)
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ z, const float* __restrict__ sa, const float* __restrict__ sb,
                                                       const T* __restrict__ res, T* __restrict__ y, size_t M, int C, int act) {
  const int c4n = C / 4;
  const size_t total = M * c4n;
  GS_LOOP(idx, total) {
    const int c = (int)(idx % c4n) * 4;
    const size_t off = (idx / c4n) * C + c;
    f32x4 v = load4<T>(z + off) * *reinterpret_cast<const f32x4*>(sa + c) + *reinterpret_cast<const f32x4*>(sb + c);
    if (res) v += load4<T>(res + off);
    if (act == ACT_LRELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.1f * v[e];
    }
    store4<T>(y + off, v);
  }
}

// dyin = dout * act'(y)  with y = sa*z + sb (+res) recomputed;  writes g (gradient w.r.t. the BN output, also the residual's gradient)
template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ z, const float* __restrict__ sa,
                                                         const float* __restrict__ sb, const T* __restrict__ res, T* __restrict__ g, size_t M, int C) {
  const int c4n = C / 4;
  const size_t total = M * c4n;
  GS_LOOP(idx, total) {
    const int c = (int)(idx % c4n) * 4;
    const size_t off = (idx / c4n) * C + c;
    f32x4 y = load4<T>(z + off) * *reinterpret_cast<const f32x4*>(sa + c) + *reinterpret_cast<const f32x4*>(sb + c);
    if (res) y += load4<T>(res + off);
    f32x4 d = load4<T>(dout + off);
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = y[e] > 0.f ? d[e] : 0.1f * d[e];
    store4<T>(g + off, d);
  }
}

// dz = ca[c] * dy + cb[c] + cc[c] * xhat,  xhat = (z - mean) * invstd
// Optional fused tail of a residual block's backward (train_engine.hip): acc != nullptr adds the residual-stream gradient (dz = acc + ...,
// dz may be acc itself), out2 != nullptr also stores scale2[image] * dz - the drop-path-scaled copy the next branch's weight / data
// gradients start from (out2 may be dy itself: each thread reads its elements before it writes them).
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* dy, const T* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ ca, const float* __restrict__ cb,
                                                           const float* __restrict__ cc, T* dz, size_t M, int C, const T* acc, const float* __restrict__ scale2,
                                                           T* out2, size_t rows_per_img, const float* __restrict__ act_sa, const float* __restrict__ act_sb) {
  const int c4n = C / 4;
  const size_t total = M * c4n;
  GS_LOOP(idx, total) {
    const int c = (int)(idx % c4n) * 4;
    const size_t row = idx / c4n, off = row * C + c;
    const f32x4 zv = load4<T>(z + off);
    const f32x4 xh = (zv - *reinterpret_cast<const f32x4*>(mean + c)) * *reinterpret_cast<const f32x4*>(invstd + c);
    f32x4 d = load4<T>(dy + off);
    if (act_sa) {                                     // dy = gradient behind the LeakyReLU(0.1) that followed this BatchNorm (see bn_reduce_kernel)
      const f32x4 yv = zv * *reinterpret_cast<const f32x4*>(act_sa + c) + *reinterpret_cast<const f32x4*>(act_sb + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = yv[e] > 0.f ? d[e] : 0.1f * d[e];
    }
    f32x4 v = *reinterpret_cast<const f32x4*>(ca + c) * d + *reinterpret_cast<const f32x4*>(cb + c) +
              *reinterpret_cast<const f32x4*>(cc + c) * xh;
    if (acc) v += load4<T>(acc + off);
    store4<T>(dz + off, v);
    if (out2) {
      // the copy scales what the NEXT kernels will read back from dz, i.e. the value rounded to the storage type
      f32x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = to_f32<T>(from_f32<T>(v[e]));
      if (scale2) w = w * scale2[row / rows_per_img];
      store4<T>(out2 + off, w);
    }
  }
}

// ---- row-walking forms of bn_apply / bn_bwd_apply (round 3): a thread owns V = 16 / sizeof(T) channels (16-byte accesses) of the rows blk * R + rl,
// += gridDim * R; its per-channel coefficients are loaded ONCE, there is no index division in the loop, and U rows' loads are issued back to back.
// (The element-indexed kernels above did a 64-bit div / mod per 8-byte access and reached 3 TB/s on the 164 ... 328 MB maps of the stem and stage 1.)
template <typename T, int V>
__device__ __forceinline__ void ldv(const T* p, float (&o)[V]) {
  if constexpr (sizeof(T) == 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = v[e];
  } else {
    const bf16x8 h = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)h[e];
  }
}
template <typename T, int V>
__device__ __forceinline__ void stv(T* p, const float (&v)[V]) {
  if constexpr (sizeof(T) == 4) *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  else {
    const bf16x8 h = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3], (bf16)v[4], (bf16)v[5], (bf16)v[6], (bf16)v[7]};
    *reinterpret_cast<u32x4*>(p) = __builtin_bit_cast(u32x4, h);
  }
}
template <int V>
__device__ __forceinline__ void ldc(const float* p, float (&o)[V]) {
#pragma unroll
  for (int e = 0; e < V; e += 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(p + e); o[e] = v[0]; o[e + 1] = v[1]; o[e + 2] = v[2]; o[e + 3] = v[3]; }
}

template <typename T, int V, int U>
__global__ __launch_bounds__(256) void bn_apply_rows_kernel(const T* __restrict__ z, const float* __restrict__ sa, const float* __restrict__ sb,
                                                            const T* __restrict__ res, T* __restrict__ y, size_t M, int C, int act) {
  const int lanesC = C / V, R = 256 / lanesC;
  const int cl = threadIdx.x % lanesC, rl = threadIdx.x / lanesC, c = cl * V;
  float a[V], bsh[V];
  ldc<V>(sa + c, a);
  ldc<V>(sb + c, bsh);
  const size_t step = (size_t)gridDim.x * R;
  for (size_t m = (size_t)blockIdx.x * R + rl; m < M; m += U * step) {
    float zv[U][V], rv[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t mm = m + u * step;
      if (mm < M) { ldv<T, V>(z + mm * C + c, zv[u]); if (res) ldv<T, V>(res + mm * C + c, rv[u]); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t mm = m + u * step;
      if (mm < M) {
        float o[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
          float v = zv[u][e] * a[e] + bsh[e];
          if (res) v += rv[u][e];
          if (act == ACT_LRELU) v = v > 0.f ? v : 0.1f * v;
          o[e] = v;
        }
        stv<T, V>(y + mm * C + c, o);
      }
    }
  }
}

template <typename T, int V, int U>
__global__ __launch_bounds__(256) void bn_bwd_apply_rows_kernel(const T* dy, const T* __restrict__ z, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ ca, const float* __restrict__ cb,
                                                                const float* __restrict__ cc, T* dz, size_t M, int C, const T* acc, const float* __restrict__ scale2,
                                                                T* out2, size_t rows_per_img, const float* __restrict__ act_sa, const float* __restrict__ act_sb) {
  const int lanesC = C / V, R = 256 / lanesC;
  const int cl = threadIdx.x % lanesC, rl = threadIdx.x / lanesC, c = cl * V;
  float A[V], MU[V], IS[V], B0[V], C0[V], asa[V], asb[V];      // the same expression, in the same order, as bn_bwd_apply_kernel
  ldc<V>(ca + c, A); ldc<V>(mean + c, MU); ldc<V>(invstd + c, IS); ldc<V>(cb + c, B0); ldc<V>(cc + c, C0);
#pragma unroll
  for (int e = 0; e < V; ++e) asa[e] = asb[e] = 0.f;
  if (act_sa) { ldc<V>(act_sa + c, asa); ldc<V>(act_sb + c, asb); }
  const size_t step = (size_t)gridDim.x * R;
  for (size_t m = (size_t)blockIdx.x * R + rl; m < M; m += U * step) {
    float zv[U][V], dv[U][V], av[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t mm = m + u * step;
      if (mm < M) { ldv<T, V>(z + mm * C + c, zv[u]); ldv<T, V>(dy + mm * C + c, dv[u]); if (acc) ldv<T, V>(acc + mm * C + c, av[u]); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t mm = m + u * step;
      if (mm < M) {
        float o[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
          float d = dv[u][e];
          if (act_sa) d = zv[u][e] * asa[e] + asb[e] > 0.f ? d : 0.1f * d;
          float v = A[e] * d + B0[e] + C0[e] * ((zv[u][e] - MU[e]) * IS[e]);
          if (acc) v += av[u][e];
          o[e] = v;
        }
        stv<T, V>(dz + mm * C + c, o);
        if (out2) {                       // the copy scales what the NEXT kernels will read back from dz, i.e. the value rounded to the storage type
          const float sc = scale2 ? scale2[mm / rows_per_img] : 1.0f;
#pragma unroll
          for (int e = 0; e < V; ++e) o[e] = to_f32<T>(from_f32<T>(o[e])) * sc;
          stv<T, V>(out2 + mm * C + c, o);
        }
      }
    }
  }
}
static inline bool rows_form_ok(int C, int V) { const int lanesC = C / V; return C % V == 0 && lanesC >= 1 && lanesC <= 256 && 256 % lanesC == 0; }
static inline unsigned rows_grid(size_t M, int C, int V, int U) {
  const size_t R = 256 / (C / V);
  size_t nb = (M + R * U - 1) / (R * U);
  return (unsigned)(nb > 4096 ? 4096 : (nb < 1 ? 1 : nb));
}

// ------------------------------------------------------------------------------------------------ elementwise
template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ z, T* __restrict__ h, size_t n4) {
  GS_LOOP(idx, n4) {
    f32x4 v = load4<T>(z + idx * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = sizeof(T) == 2 ? gelu_sig(v[e]) : gelu_erf(v[e]);
    store4<T>(h + idx * 4, v);
  }
}
// dz = dh * (Phi(z) + z * phi(z))
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ z, T* __restrict__ dz, size_t n4) {
  GS_LOOP(idx, n4) {
    const f32x4 x = load4<T>(z + idx * 4), d = load4<T>(dh + idx * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float cdf = 0.5f * (1.0f + erff(x[e] * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * expf(-0.5f * x[e] * x[e]);
      o[e] = d[e] * (cdf + x[e] * pdf);
    }
    store4<T>(dz + idx * 4, o);
  }
}
// out = a + scale[b] * br   (scale == nullptr: 1);  rows_per_img rows of C per image
template <typename T>
__global__ __launch_bounds__(256) void add_scaled_kernel(const T* __restrict__ a, const T* __restrict__ br, const float* __restrict__ scale,
                                                         T* __restrict__ out, size_t n4, size_t per_img4) {
  GS_LOOP(idx, n4) {
    const float s = scale ? scale[idx / per_img4] : 1.0f;
    f32x4 v = load4<T>(br + idx * 4) * s;
    if (a) v += load4<T>(a + idx * 4);
    store4<T>(out + idx * 4, v);
  }
}

// MaxPool2d(2) with argmax (0..3) + pos add (forward), and its scatter backward
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_idx_kernel(const T* __restrict__ in, const float* __restrict__ pos, T* __restrict__ out,
                                                           unsigned char* __restrict__ arg, int B, int OH, int OW, int C) {
  const size_t total = (size_t)B * OH * OW * C;
  GS_LOOP(idx, total) {
    const int c = (int)(idx % C);
    const size_t pix = idx / C;
    const int ox = (int)(pix % OW);
    const size_t t2 = pix / OW;
    const int oy = (int)(t2 % OH);
    const size_t b = t2 / OH;
    const int W = OW * 2;
    const T* p = in + ((b * OH * 2 + oy * 2) * W + ox * 2) * C + c;
    float best = to_f32<T>(p[0]);
    int bi = 0;
    const float v1 = to_f32<T>(p[C]), v2 = to_f32<T>(p[(size_t)W * C]), v3 = to_f32<T>(p[(size_t)W * C + C]);
    if (v1 > best) { best = v1; bi = 1; }
    if (v2 > best) { best = v2; bi = 2; }
    if (v3 > best) { best = v3; bi = 3; }
    arg[idx] = (unsigned char)bi;
    out[idx] = from_f32<T>(best + (pos ? pos[(size_t)(oy * OW + ox) * C + c] : 0.f));
  }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* __restrict__ dout, const unsigned char* __restrict__ arg, T* __restrict__ din,
                                                           int B, int OH, int OW, int C) {
  const size_t total = (size_t)B * OH * OW * C;
  GS_LOOP(idx, total) {
    const int c = (int)(idx % C);
    const size_t pix = idx / C;
    const int ox = (int)(pix % OW);
    const size_t t2 = pix / OW;
    const int oy = (int)(t2 % OH);
    const size_t b = t2 / OH;
    const int W = OW * 2;
    T* p = din + ((b * OH * 2 + oy * 2) * W + ox * 2) * C + c;
    const int a = arg[idx];
    const T d = dout[idx], z = from_f32<T>(0.f);
    p[0] = a == 0 ? d : z;
    p[C] = a == 1 ? d : z;
    p[(size_t)W * C] = a == 2 ? d : z;
    p[(size_t)W * C + C] = a == 3 ? d : z;
  }
}

// Stem tail in one pass (visformer.py:224-237): y = LeakyReLU(sa * z + sb + res), MaxPool2d(2), + pos_embed1.  The activated map is never stored
// (the separate bn_apply + maxpool pair wrote and re-read its 328 MB at 800 images).  arg = window position of the maximum (first of equals,
// compared in fp32) | 4 when the maximum is positive - all the backward needs (LeakyReLU slope of the routed gradient).  V channels per thread.
// rsa != nullptr: `res` is the PRE-normalisation map of the identity path and its BatchNorm is applied here too (res * rsa + rsb): the normalised
// identity map (328 MB written and read back at 800 images) is never stored either.
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_pool_fwd_kernel(const T* __restrict__ z, const float* __restrict__ sa, const float* __restrict__ sb,
                                                          const T* __restrict__ res, const float* __restrict__ pos, T* __restrict__ out,
                                                          unsigned char* __restrict__ arg, int B, int OH, int OW, int C,
                                                          const float* __restrict__ rsa, const float* __restrict__ rsb) {
  const int cvn = C / V, W = OW * 2;
  const size_t total = (size_t)B * OH * OW * cvn;
  GS_LOOP(idx, total) {
    const int c = (int)(idx % cvn) * V;
    const size_t pix = idx / cvn;
    const int ox = (int)(pix % OW);
    const size_t t2 = pix / OW;
    const int oy = (int)(t2 % OH);
    const size_t b = t2 / OH;
    const size_t base = ((b * OH * 2 + oy * 2) * W + ox * 2) * C + c;
    float best[V];
    int bi[V];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t off = base + ((size_t)(k >> 1) * W + (k & 1)) * C;
#pragma unroll
      for (int q = 0; q < V / 4; ++q) {
        f32x4 v = load4<T>(z + off + 4 * q) * *reinterpret_cast<const f32x4*>(sa + c + 4 * q) + *reinterpret_cast<const f32x4*>(sb + c + 4 * q);
        if (res) {
          f32x4 r = load4<T>(res + off + 4 * q);
          if (rsa) {      // the normalised identity value as the separate apply pass stored it: rounded to the storage type
            r = r * *reinterpret_cast<const f32x4*>(rsa + c + 4 * q) + *reinterpret_cast<const f32x4*>(rsb + c + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = to_f32<T>(from_f32<T>(r[e]));
          }
          v += r;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float y = v[e] > 0.f ? v[e] : 0.1f * v[e];
          if (k == 0 || y > best[4 * q + e]) { best[4 * q + e] = y; bi[4 * q + e] = k; }
        }
      }
    }
    const size_t o = pix * C + c;
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
      f32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r[e] = best[4 * q + e] + (pos ? pos[(size_t)(oy * OW + ox) * C + c + 4 * q + e] : 0.f);
        arg[o + 4 * q + e] = (unsigned char)(bi[4 * q + e] | (best[4 * q + e] > 0.f ? 4 : 0));
      }
      store4<T>(out + o + 4 * q, r);
    }
  }
}
// Its backward: the pooled gradient is routed to the arg-max position with the LeakyReLU slope of that value, zeros elsewhere
// (= maxpool2_bwd + bn_act_bwd of the unfused path, from 123 MB of inputs instead of 1.1 GB)
template <typename T, int V>
__global__ __launch_bounds__(256) void pool_act_bwd_kernel(const T* __restrict__ dout, const unsigned char* __restrict__ arg, T* __restrict__ g,
                                                           int B, int OH, int OW, int C) {
  const int cvn = C / V, W = OW * 2;
  const size_t total = (size_t)B * OH * OW * cvn;
  GS_LOOP(idx, total) {
    const int c = (int)(idx % cvn) * V;
    const size_t pix = idx / cvn;
    const int ox = (int)(pix % OW);
    const size_t t2 = pix / OW;
    const int oy = (int)(t2 % OH);
    const size_t b = t2 / OH;
    const size_t base = ((b * OH * 2 + oy * 2) * W + ox * 2) * C + c, o = pix * C + c;
    float d[V];
    int a[V];
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
      const f32x4 v = load4<T>(dout + o + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[4 * q + e] = arg[o + 4 * q + e]; d[4 * q + e] = (a[4 * q + e] & 4) ? v[e] : 0.1f * v[e]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t off = base + ((size_t)(k >> 1) * W + (k & 1)) * C;
#pragma unroll
      for (int q = 0; q < V / 4; ++q) {
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (a[4 * q + e] & 3) == k ? d[4 * q + e] : 0.f;
        store4<T>(g + off + 4 * q, r);
      }
    }
  }
}

// ---- Stem tail backward without the routed gradient map (round 4).  g = the pooled gradient routed to its window's arg-max position with the
// LeakyReLU slope (pool_act_bwd_kernel's output) has ONE non-zero per 2 x 2 window and feeds TWO BatchNorm backwards (bn3 on z3, the identity
// path's BatchNorm on zd).  Their reductions  sum g, sum g xhat3, sum g xhatd  and their apply passes  dz = ca g + cb + cc xhat  are formed straight
// from the pooled gradient + arg: pool_act_bwd's 328 MB write and the four reads of that map are gone (2.7 GB -> 1.6 GB at 800 images).
// Row-walking form: a thread owns V channels and walks pooled pixels; partial3 / partiald are [blocks][2][C] like bn_reduce_kernel's.
template <typename T, int V>
__global__ __launch_bounds__(256) void pool_bn_bwd_reduce_kernel(const T* __restrict__ dout, const unsigned char* __restrict__ arg, const T* __restrict__ z3,
                                                                 const T* __restrict__ zd, const float* __restrict__ mean3, const float* __restrict__ is3,
                                                                 const float* __restrict__ meand, const float* __restrict__ isd, float* __restrict__ partial3,
                                                                 float* __restrict__ partiald, int B, int OH, int OW, int C) {
  __shared__ float red[3][V][256];
  const int lanesC = C / V, R = 256 / lanesC, W = OW * 2;
  const int cl = threadIdx.x % lanesC, rl = threadIdx.x / lanesC, c = cl * V;
  float M3[V], I3[V], MD[V], ID[V], s0[V], s1[V], s2[V];
  ldc<V>(mean3 + c, M3); ldc<V>(is3 + c, I3); ldc<V>(meand + c, MD); ldc<V>(isd + c, ID);
#pragma unroll
  for (int e = 0; e < V; ++e) s0[e] = s1[e] = s2[e] = 0.f;
  const size_t npix = (size_t)B * OH * OW, step = (size_t)gridDim.x * R;
  for (size_t pix = (size_t)blockIdx.x * R + rl; pix < npix; pix += step) {
    const int ox = (int)(pix % OW);
    const size_t t2 = pix / OW;
    const int oy = (int)(t2 % OH);
    const size_t b = t2 / OH;
    const size_t base = ((b * OH * 2 + oy * 2) * W + ox * 2) * C + c, o = pix * C + c;
    float d[V], zv[4][V], dv[4][V];
    ldv<T, V>(dout + o, d);
    unsigned char a[V];
    if constexpr (V == 8) *reinterpret_cast<uint2*>(a) = *reinterpret_cast<const uint2*>(arg + o);
    else *reinterpret_cast<unsigned*>(a) = *reinterpret_cast<const unsigned*>(arg + o);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t off = base + ((size_t)(k >> 1) * W + (k & 1)) * C;
      ldv<T, V>(z3 + off, zv[k]);
      ldv<T, V>(zd + off, dv[k]);
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float g = to_f32<T>(from_f32<T>((a[e] & 4) ? d[e] : 0.1f * d[e]));      // the routed gradient as pool_act_bwd_kernel stored it
      const int k = a[e] & 3;
      const float x3 = k == 0 ? zv[0][e] : k == 1 ? zv[1][e] : k == 2 ? zv[2][e] : zv[3][e];
      const float xd = k == 0 ? dv[0][e] : k == 1 ? dv[1][e] : k == 2 ? dv[2][e] : dv[3][e];
      s0[e] += g;
      s1[e] += g * ((x3 - M3[e]) * I3[e]);
      s2[e] += g * ((xd - MD[e]) * ID[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) { red[0][e][threadIdx.x] = s0[e]; red[1][e][threadIdx.x] = s1[e]; red[2][e][threadIdx.x] = s2[e]; }
  __syncthreads();
  if (rl == 0) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      for (int r = 1; r < R; ++r) { s0[e] += red[0][e][r * lanesC + cl]; s1[e] += red[1][e][r * lanesC + cl]; s2[e] += red[2][e][r * lanesC + cl]; }
      partial3[((size_t)blockIdx.x * 2 + 0) * C + c + e] = s0[e];
      partial3[((size_t)blockIdx.x * 2 + 1) * C + c + e] = s1[e];
      partiald[((size_t)blockIdx.x * 2 + 0) * C + c + e] = s0[e];
      partiald[((size_t)blockIdx.x * 2 + 1) * C + c + e] = s2[e];
    }
  }
}
// coef3 / coefd: [3][C] = ca | cb | cc of bn_bwd_finalize_kernel
template <typename T, int V>
__global__ __launch_bounds__(256) void pool_bn_bwd_apply_kernel(const T* __restrict__ dout, const unsigned char* __restrict__ arg, const T* __restrict__ z3,
                                                                const T* __restrict__ zd, const float* __restrict__ mean3, const float* __restrict__ is3,
                                                                const float* __restrict__ meand, const float* __restrict__ isd, const float* __restrict__ coef3,
                                                                const float* __restrict__ coefd, T* __restrict__ dz3, T* __restrict__ dzd, int B, int OH, int OW, int C) {
  const int lanesC = C / V, R = 256 / lanesC, W = OW * 2;
  const int cl = threadIdx.x % lanesC, rl = threadIdx.x / lanesC, c = cl * V;
  float M3[V], I3[V], MD[V], ID[V], A3[V], B3[V], C3[V], AD[V], BD[V], CD[V];
  ldc<V>(mean3 + c, M3); ldc<V>(is3 + c, I3); ldc<V>(meand + c, MD); ldc<V>(isd + c, ID);
  ldc<V>(coef3 + c, A3); ldc<V>(coef3 + C + c, B3); ldc<V>(coef3 + 2 * C + c, C3);
  ldc<V>(coefd + c, AD); ldc<V>(coefd + C + c, BD); ldc<V>(coefd + 2 * C + c, CD);
  const size_t npix = (size_t)B * OH * OW, step = (size_t)gridDim.x * R;
  for (size_t pix = (size_t)blockIdx.x * R + rl; pix < npix; pix += step) {
    const int ox = (int)(pix % OW);
    const size_t t2 = pix / OW;
    const int oy = (int)(t2 % OH);
    const size_t b = t2 / OH;
    const size_t base = ((b * OH * 2 + oy * 2) * W + ox * 2) * C + c, o = pix * C + c;
    float d[V], zv[4][V], dv[4][V];
    ldv<T, V>(dout + o, d);
    unsigned char a[V];
    if constexpr (V == 8) *reinterpret_cast<uint2*>(a) = *reinterpret_cast<const uint2*>(arg + o);
    else *reinterpret_cast<unsigned*>(a) = *reinterpret_cast<const unsigned*>(arg + o);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t off = base + ((size_t)(k >> 1) * W + (k & 1)) * C;
      ldv<T, V>(z3 + off, zv[k]);
      ldv<T, V>(zd + off, dv[k]);
    }
#pragma unroll
    for (int e = 0; e < V; ++e) d[e] = to_f32<T>(from_f32<T>((a[e] & 4) ? d[e] : 0.1f * d[e]));      // the routed gradient as pool_act_bwd_kernel stored it
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t off = base + ((size_t)(k >> 1) * W + (k & 1)) * C;
      float o3[V], od[V];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float g = (a[e] & 3) == k ? d[e] : 0.f;
        o3[e] = A3[e] * g + B3[e] + C3[e] * ((zv[k][e] - M3[e]) * I3[e]);
        od[e] = AD[e] * g + BD[e] + CD[e] * ((dv[k][e] - MD[e]) * ID[e]);
      }
      stv<T, V>(dz3 + off, o3);
      stv<T, V>(dzd + off, od);
    }
  }
}

// dx[b][hw][c] = dfeat[b][c] / HW   (AdaptiveAvgPool2d(1) backward)
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dfeat, T* __restrict__ dx, int B, int HW, int C) {
  const size_t total = (size_t)B * HW * C;
  GS_LOOP(idx, total) {
    const int c = (int)(idx % C);
    const size_t b = idx / ((size_t)HW * C);
    dx[idx] = from_f32<T>(dfeat[b * C + c] / (float)HW);
  }
}

// out[r][c] = sum_b g[b][r][c]  (pos_embed gradient: r over HW) ; rows = HW, reduces over B images
template <typename T>
__global__ __launch_bounds__(1024) void batch_sum_kernel(const T* __restrict__ g, float* __restrict__ out, int B, size_t per_img) {
  // block = 256 consecutive elements (4 per lane) x 16 waves over the batch (wave w sums images w, w + 16, ..., two at a time), LDS reduce in
  // wave order.  (4 waves with one dependent load per image: 200 serial loads per thread at 800 images, 0.7 TB/s)
  __shared__ f32x4 red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t idx = ((size_t)blockIdx.x * 64 + lane) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, s2 = s;
  if (idx < per_img) {
    int b = wave;
    for (; b + 16 < B; b += 32) {
      s += load4<T>(g + (size_t)b * per_img + idx);
      s2 += load4<T>(g + (size_t)(b + 16) * per_img + idx);
    }
    if (b < B) s += load4<T>(g + (size_t)b * per_img + idx);
  }
  red[wave][lane] = s + s2;
  __syncthreads();
  if (wave == 0 && idx < per_img) {
    f32x4 t = red[0][lane];
#pragma unroll
    for (int w = 1; w < 16; ++w) t += red[w][lane];
    *reinterpret_cast<f32x4*>(out + idx) = t;
  }
}

// y[b][i] = x[b][i] + p[i]   (pos_embed add after the PatchEmbed norm, visformer.py:437-438)
template <typename T>
__global__ __launch_bounds__(256) void bcast_add_kernel(const T* __restrict__ x, const float* __restrict__ p, T* __restrict__ y, size_t n4, size_t per_img4) {
  GS_LOOP(idx, n4) {
    const f32x4 v = load4<T>(x + idx * 4) + *reinterpret_cast<const f32x4*>(p + (idx % per_img4) * 4);
    store4<T>(y + idx * 4, v);
  }
}
__global__ __launch_bounds__(256) void fill_f32_kernel(float* __restrict__ p, float v, size_t n) { GS_LOOP(idx, n) p[idx] = v; }
__global__ __launch_bounds__(256) void scale_copy_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n, float s) { GS_LOOP(idx, n) out[idx] = in[idx] * s; }
// column sums from bn_reduce partials (conv bias gradient)
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ out) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), sub = threadIdx.x & 63;
  double s0, s1;
  bn_partial_sums(partial, nblk, C, c, sub, s0, s1);
  if (c >= C || sub) return;
  out[c] = (float)s0;
}

// SGD with momentum and weight decay, torch.optim.SGD semantics (utils/__init__.py:131-132): d = g + wd*p; buf = mom*buf + d (buf = d on the
// first step); p -= lr * buf
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, size_t n, float lr,
                                                  float momentum, float wd, int first) {
  GS_LOOP(idx, n) {
    const float d = g[idx] + wd * p[idx];
    const float b = first ? d : momentum * buf[idx] + d;
    buf[idx] = b;
    p[idx] -= lr * b;
  }
}

// the same update for a table of tensors in ONE launch (blockIdx.y = tensor): 86 parameter tensors per meta-tuning step
struct SgdItem { float* p; const float* g; float* buf; size_t n; };
__global__ __launch_bounds__(256) void sgd_multi_kernel(const SgdItem* __restrict__ items, float lr, float momentum, float wd, int first) {
  const SgdItem it = items[blockIdx.y];
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < it.n; idx += (size_t)gridDim.x * 256) {
    const float d = it.g[idx] + wd * it.p[idx];
    const float b = first ? d : momentum * it.buf[idx] + d;
    it.buf[idx] = b;
    it.p[idx] -= lr * b;
  }
}

// ------------------------------------------------------------------------------------------------ ViT / DeiT training (deit.py:61-78, 139-218)
// nn.LayerNorm in train mode: y = (x - mean) * rstd * gamma + beta (biased variance, deit.py:68,73,170), row statistics kept for the backward.
// One wave per row (D <= 2048), 4 rows per workgroup.
template <typename T>
__global__ __launch_bounds__(256) void ln_train_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int M, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const T* xr = x + (size_t)row * D;
  float s = 0.f, ss = 0.f;
  for (int d = lane * 4; d < D; d += 256) { const f32x4 v = load4<T>(xr + d); s += v[0] + v[1] + v[2] + v[3]; ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
  const float mu = s / D;
  float var = ss / D - mu * mu;
  var = var < 0.f ? 0.f : var;
  const float rs = 1.0f / sqrtf(var + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  T* yr = y + (size_t)row * D;
  for (int d = lane * 4; d < D; d += 256) {
    const f32x4 v = (load4<T>(xr + d) - mu) * rs * *reinterpret_cast<const f32x4*>(gamma + d) + *reinterpret_cast<const f32x4*>(beta + d);
    store4<T>(yr + d, v);
  }
}

// LayerNorm backward: dx = rstd * (g - mean_d(g) - xhat * mean_d(g * xhat)), g = dy * gamma (+ add, the residual's gradient);
// partial[blk][0][d] = sum over the block's rows of dy, [1][d] of dy * xhat  (-> dbeta, dgamma by ln_param_grad_kernel).
// gridDim.x blocks of 4 waves, block b owns rows b*R .. (R = rows per block), each wave every 4th of them.
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma, const T* add,
                                                     T* dx, float* __restrict__ partial, int M, int D, int rows_per_blk) {      // (add may be dx: no restrict)
  extern __shared__ float red[];                       // [4 waves][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* mine = red + (size_t)wave * 2 * D;
  for (int d = lane; d < 2 * D; d += 64) mine[d] = 0.f;
  const long r0 = (long)blockIdx.x * rows_per_blk;
  for (long row = r0 + wave; row < r0 + rows_per_blk && row < M; row += 4) {
    const T* dyr = dy + (size_t)row * D;
    const T* xr = x + (size_t)row * D;
    const float mu = mean[row], rs = rstd[row];
    float c1 = 0.f, c2 = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
      const f32x4 g = load4<T>(dyr + d) * *reinterpret_cast<const f32x4*>(gamma + d);
      const f32x4 xh = (load4<T>(xr + d) - mu) * rs;
      c1 += g[0] + g[1] + g[2] + g[3];
      c2 += g[0] * xh[0] + g[1] * xh[1] + g[2] * xh[2] + g[3] * xh[3];
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { c1 += __shfl_xor(c1, o); c2 += __shfl_xor(c2, o); }
    c1 /= D; c2 /= D;
    for (int d = lane * 4; d < D; d += 256) {
      const f32x4 dyv = load4<T>(dyr + d);
      const f32x4 g = dyv * *reinterpret_cast<const f32x4*>(gamma + d);
      const f32x4 xh = (load4<T>(xr + d) - mu) * rs;
      f32x4 v = (g - c1 - xh * c2) * rs;
      if (add) v += load4<T>(add + (size_t)row * D + d);
      store4<T>(dx + (size_t)row * D + d, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) { mine[d + e] += dyv[e]; mine[D + d + e] += dyv[e] * xh[e]; }
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < 2 * D; d += 256)
    partial[(size_t)blockIdx.x * 2 * D + d] = red[d] + red[2 * D + d] + red[4 * D + d] + red[6 * D + d];
}

// ln_bwd_kernel with the row in registers (round 3): a lane owns the channels 4 lane + 256 g .. + 3 (g < NG = ceil(D / 256)); dy / x are read ONCE per
// row (the kernel above read them for the row statistics and again for dx), the dgamma / dbeta sums live in registers until the end (it did a
// read-modify-write of LDS per element and row), two rows per wave are in flight.  76 -> ~35 us on the [39 400 x 384] maps of a 200-image DeiT-S step.
template <typename T, int NG>
__global__ __launch_bounds__(256) void ln_bwd_rows_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma, const T* add,
                                                          T* dx, float* __restrict__ partial, int M, int D, int rows_per_blk) {      // (add may be dx: no restrict)
  extern __shared__ float red[];                       // [4 waves][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 gam[NG], accb[NG], accg[NG];
  bool act[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int d = lane * 4 + 256 * g;
    act[g] = d < D;
    gam[g] = act[g] ? *reinterpret_cast<const f32x4*>(gamma + d) : f32x4{0.f, 0.f, 0.f, 0.f};
    accb[g] = accg[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const long r0 = (long)blockIdx.x * rows_per_blk;
  long rend = r0 + rows_per_blk; rend = rend < M ? rend : M;
  const float invD = 1.0f / D;
  for (long row = r0 + wave; row < rend; row += 8) {
    f32x4 dyv[2][NG], xv[2][NG], av[2][NG];
    float mu[2], rs[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long rr = row + 4 * u;
      ok[u] = rr < rend;
      const long rc = ok[u] ? rr : row;
      mu[u] = mean[rc]; rs[u] = rstd[rc];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int d = act[g] ? lane * 4 + 256 * g : 0;
        dyv[u][g] = load4<T>(dy + (size_t)rc * D + d);
        xv[u][g] = load4<T>(x + (size_t)rc * D + d);
        if (add) av[u][g] = load4<T>(add + (size_t)rc * D + d);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float c1 = 0.f, c2 = 0.f;
      f32x4 gg[NG], xh[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        xh[g] = (xv[u][g] - mu[u]) * rs[u];
        gg[g] = dyv[u][g] * gam[g];
        if (act[g]) {
          c1 += gg[g][0] + gg[g][1] + gg[g][2] + gg[g][3];
          c2 += gg[g][0] * xh[g][0] + gg[g][1] * xh[g][1] + gg[g][2] * xh[g][2] + gg[g][3] * xh[g][3];
        }
      }
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { c1 += __shfl_xor(c1, o); c2 += __shfl_xor(c2, o); }
      c1 *= invD; c2 *= invD;
      if (ok[u]) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
          if (act[g]) {
            f32x4 v = (gg[g] - c1 - xh[g] * c2) * rs[u];
            if (add) v += av[u][g];
            store4<T>(dx + (size_t)(row + 4 * u) * D + lane * 4 + 256 * g, v);
            accb[g] += dyv[u][g];
            accg[g] += dyv[u][g] * xh[g];
          }
      }
    }
  }
  float* mine = red + (size_t)wave * 2 * D;
#pragma unroll
  for (int g = 0; g < NG; ++g)
    if (act[g]) {
      const int d = lane * 4 + 256 * g;
      *reinterpret_cast<f32x4*>(mine + d) = accb[g];
      *reinterpret_cast<f32x4*>(mine + D + d) = accg[g];
    }
  __syncthreads();
  for (int d = threadIdx.x; d < 2 * D; d += 256)
    partial[(size_t)blockIdx.x * 2 * D + d] = red[d] + red[2 * D + d] + red[4 * D + d] + red[6 * D + d];
}

// dbeta[d] = sum_blk partial[blk][0][d], dgamma[d] = sum_blk partial[blk][1][d]  (fp64 accumulation, one wave per channel)
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const float* __restrict__ partial, int nblk, int D, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), sub = threadIdx.x & 63;
  double s0, s1;
  bn_partial_sums(partial, nblk, D, c, sub, s0, s1);
  if (c >= D || sub) return;
  if (dbeta) dbeta[c] = (float)s0;
  if (dgamma) dgamma[c] = (float)s1;
}

// tokens[b][0] = cls + pos[0];  tokens[b][1 + i] = zpe[b * np + i] + pos[1 + i]   (deit.py:196-202)
template <typename T>
__global__ __launch_bounds__(256) void vit_assemble_kernel(const T* __restrict__ zpe, const float* __restrict__ cls, const float* __restrict__ pos,
                                                           T* __restrict__ tokens, int B, int S, int D) {
  const size_t total = (size_t)B * S * (D / 4);
  GS_LOOP(idx, total) {
    const int d = (int)(idx % (D / 4)) * 4;
    const size_t r = idx / (D / 4);
    const int s = (int)(r % S);
    const size_t b = r / S;
    f32x4 v = *reinterpret_cast<const f32x4*>(pos + (size_t)s * D + d);
    if (s == 0) v += *reinterpret_cast<const f32x4*>(cls + d);
    else v += load4<T>(zpe + (b * (S - 1) + s - 1) * D + d);
    store4<T>(tokens + r * D + d, v);
  }
}
// dzpe[b * np + i] = dtok[b][1 + i]
template <typename T>
__global__ __launch_bounds__(256) void vit_patch_rows_kernel(const T* __restrict__ dtok, T* __restrict__ dzpe, int B, int S, int D) {
  const size_t total = (size_t)B * (S - 1) * (D / 4);
  GS_LOOP(idx, total) {
    const int d = (int)(idx % (D / 4)) * 4;
    const size_t r = idx / (D / 4);
    const size_t b = r / (S - 1), i = r % (S - 1);
    store4<T>(dzpe + r * D + d, load4<T>(dtok + (b * S + 1 + i) * D + d));
  }
}
// final norm on the cls row (deit.py:204-205: x = norm(x); return x[:, 0]): feat[b] = LN(tokens[b][0]) * gamma + beta, fp32; statistics kept
template <typename T>
__global__ __launch_bounds__(64) void vit_cls_ln_fwd_kernel(const T* __restrict__ tokens, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ feat, float* __restrict__ mean, float* __restrict__ rstd, int S, int D, float eps) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const T* xr = tokens + (size_t)b * S * D;
  float s = 0.f, ss = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = to_f32<T>(xr[d]); s += v; ss += v * v; }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
  const float mu = s / D;
  float var = ss / D - mu * mu;
  var = var < 0.f ? 0.f : var;
  const float rs = 1.0f / sqrtf(var + eps);
  if (lane == 0) { mean[b] = mu; rstd[b] = rs; }
  for (int d = lane; d < D; d += 64) feat[(size_t)b * D + d] = (to_f32<T>(xr[d]) - mu) * rs * gamma[d] + beta[d];
}
// its backward: dtok[b][0] = LN backward of dfeat[b]; every other row of dtok is zero.  partial[b][2][D] for dgamma / dbeta (one block per image).
template <typename T>
__global__ __launch_bounds__(64) void vit_cls_ln_bwd_kernel(const float* __restrict__ dfeat, const T* __restrict__ tokens, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma, T* __restrict__ dtok,
                                                            float* __restrict__ partial, int S, int D) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const T* xr = tokens + (size_t)b * S * D;
  const float* dyr = dfeat + (size_t)b * D;
  const float mu = mean[b], rs = rstd[b];
  float c1 = 0.f, c2 = 0.f;
  for (int d = lane; d < D; d += 64) { const float g = dyr[d] * gamma[d], xh = (to_f32<T>(xr[d]) - mu) * rs; c1 += g; c2 += g * xh; }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { c1 += __shfl_xor(c1, o); c2 += __shfl_xor(c2, o); }
  c1 /= D; c2 /= D;
  for (int d = lane; d < D; d += 64) {
    const float xh = (to_f32<T>(xr[d]) - mu) * rs;
    dtok[(size_t)b * S * D + d] = from_f32<T>((dyr[d] * gamma[d] - c1 - xh * c2) * rs);
    partial[((size_t)b * 2 + 0) * D + d] = dyr[d];
    partial[((size_t)b * 2 + 1) * D + d] = dyr[d] * xh;
  }
}

// A 1x1 conv behind a pre-norm BatchNorm with the batch statistics folded in (what the eval engine's packer does with the running statistics):
// wf[n][c] = W[n][c] * sa[c] rounded ONCE to the storage type ([N][Kw] rows, zero padded), bf[n] = sum_c W[n][c] * sb[c].  One wave per row.
template <typename T>
__global__ __launch_bounds__(256) void fold_prenorm_kernel(const float* __restrict__ W, const float* __restrict__ sa, const float* __restrict__ sb,
                                                           T* __restrict__ wf, float* __restrict__ bf, int N, int C, int Kw) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float acc = 0.f;
  for (int c = lane; c < Kw; c += 64) {
    float v = 0.f;
    if (c < C) { const float w = W[(size_t)n * C + c]; v = w * sa[c]; acc += w * sb[c]; }
    wf[(size_t)n * Kw + c] = from_f32<T>(v);
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) bf[n] = acc;
}

// ------------------------------------------------------------------------------------------------ launchers
#define DISPATCH_T(dtype, CALL_F32, CALL_BF16) do { if ((dtype) == 0) { CALL_F32; } else { CALL_BF16; } } while (0)

int launch_pack_weight(const float* w, void* out, int O, int Ig, int KH, int KW, int groups, int mode, int rows_pad, int Kw, int hd_rows, int hdp_rows,
                       int hd_cols, int hdp_cols, int dtype, hipStream_t s) {
  const size_t total = (size_t)groups * rows_pad * Kw;
  if (dtype == 2) {      // two-limb words (fp32 storage, 16-bit MFMA arithmetic)
    hipLaunchKernelGGL(pack_weight_kernel<limbw>, dim3(gs_grid(total)), dim3(256), 0, s, w, (limbw*)out, O, Ig, KH, KW, groups, mode, rows_pad, Kw, hd_rows, hdp_rows, hd_cols, hdp_cols);
    return (int)hipGetLastError();
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(gs_grid(total)), dim3(256), 0, s, w, (float*)out, O, Ig, KH, KW, groups, mode, rows_pad, Kw, hd_rows, hdp_rows, hd_cols, hdp_cols),
             hipLaunchKernelGGL(pack_weight_kernel<bf16>, dim3(gs_grid(total)), dim3(256), 0, s, w, (bf16*)out, O, Ig, KH, KW, groups, mode, rows_pad, Kw, hd_rows, hdp_rows, hd_cols, hdp_cols));
  return (int)hipGetLastError();
}
int launch_pack_weight_multi(const PackJob* jobs, int n, int dtype, hipStream_t s) {
  for (int i0 = 0; i0 < n; i0 += PackJobs::MAX) {
    PackJobs pj;
    const int m = n - i0 < PackJobs::MAX ? n - i0 : PackJobs::MAX;
    size_t biggest = 0;
    for (int i = 0; i < m; ++i) {
      pj.job[i] = jobs[i0 + i];
      const size_t tot = (size_t)pj.job[i].groups * pj.job[i].rows_pad * pj.job[i].Kw;
      if (tot > biggest) biggest = tot;
    }
    unsigned gx = (unsigned)((biggest + 1023) / 1024);
    if (gx > 128) gx = 128;
    if (gx < 1) gx = 1;
    if (dtype == 2) hipLaunchKernelGGL(pack_weight_multi_kernel<limbw>, dim3(gx, (unsigned)m), dim3(256), 0, s, pj);
    else
      DISPATCH_T(dtype, hipLaunchKernelGGL(pack_weight_multi_kernel<float>, dim3(gx, (unsigned)m), dim3(256), 0, s, pj),
                 hipLaunchKernelGGL(pack_weight_multi_kernel<bf16>, dim3(gx, (unsigned)m), dim3(256), 0, s, pj));
    const int rc = (int)hipGetLastError();
    if (rc) return rc;
  }
  return 0;
}
int launch_wgrad_finalize_multi(const FinJob* jobs, int n, hipStream_t s) {
  for (int i0 = 0; i0 < n; i0 += FinJobs::MAX) {
    FinJobs fj;
    const int m = n - i0 < FinJobs::MAX ? n - i0 : FinJobs::MAX;
    size_t biggest = 0;
    for (int i = 0; i < m; ++i) {
      const FinJob& j = fj.job[i] = jobs[i0 + i];
      size_t tot = j.kind == 1 ? (size_t)j.Ng * (j.Ig / 4) : j.kind == 0 ? (size_t)j.Ng * j.Ig * j.KH * j.KW
                   : j.kind == 2 ? (size_t)j.g * j.Ng * j.Ig * j.KH * j.KW : (size_t)j.Kc_pad * 9 * 32 * 32 / 4;
      if (tot > biggest) biggest = tot;
    }
    unsigned gx = (unsigned)((biggest + 255) / 256);          // blocks per job: the largest layer gets one pass, smaller jobs' surplus blocks exit at once
    if (gx > 1024) gx = 1024;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(wgrad_finalize_multi_kernel, dim3(gx, (unsigned)m), dim3(256), 0, s, fj);
    const int rc = (int)hipGetLastError();
    if (rc) return rc;
  }
  return 0;
}
int launch_droppath_scales(const float* masks, float* scales, int ncalls, int n_img, const float* keep, hipStream_t s) {
  // the keep table travels as a kernel argument, DropKeep::MAX calls per launch: deeper encoders (> 64 DropPath calls a step) take further launches
  for (int c0 = 0; c0 < ncalls; c0 += DropKeep::MAX) {
    const int n = ncalls - c0 < DropKeep::MAX ? ncalls - c0 : DropKeep::MAX;
    DropKeep k;
    for (int i = 0; i < n; ++i) k.inv[i] = 1.0f / keep[c0 + i];
    unsigned gx = (unsigned)((n_img + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(droppath_scales_kernel, dim3(gx, (unsigned)n), dim3(256), 0, s, masks + (size_t)c0 * n_img, scales + (size_t)c0 * n_img, n_img, k);
    const int rc = (int)hipGetLastError();
    if (rc) return rc;
  }
  return 0;
}
int launch_wgrad_finalize(const float* y, float* dw, int Ng, int Ig, int KH, int KW, int g, int splits, int Kc_pad, int hd_rows, int hdp_rows, int hd_cols,
                          int hdp_cols, hipStream_t s) {
  const size_t total = (size_t)Ng * Ig * KH * KW;
  if (KH == 1 && KW == 1 && g == 0 && hd_rows == hdp_rows && hd_cols == hdp_cols && (Ig & 3) == 0 && (Kc_pad & 3) == 0) {
    hipLaunchKernelGGL(wgrad_finalize_1x1_kernel, dim3(gs_grid((size_t)Ng * (Ig / 4))), dim3(256), 0, s, y, dw, Ng, Ig, splits, Kc_pad);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(gs_grid(total)), dim3(256), 0, s, y, dw, Ng, Ig, KH, KW, g, splits, Kc_pad, hd_rows, hdp_rows, hd_cols, hdp_cols);
  return (int)hipGetLastError();
}
int launch_wgrad_finalize_dense(const float* y, float* dw, int Ng, int Ig, int KH, int KW, int groups, int splits, int Kc_pad, hipStream_t s) {
  const size_t total = (size_t)groups * Ng * Ig * KH * KW;
  hipLaunchKernelGGL(wgrad_finalize_dense_kernel, dim3(gs_grid(total)), dim3(256), 0, s, y, dw, Ng, Ig, KH, KW, groups, splits, Kc_pad);
  return (int)hipGetLastError();
}
// dtype 2: fp32 in, two-limb words out (the weight-side operand of a `bf16x2` GEMM)
int launch_transpose_cols(const void* in, void* out, int M, int ld, int c0, int ncols, int Mpad, int dtype, hipStream_t s) {
  if (dtype == 2) {
    if (((ld | c0 | ncols | Mpad) & 3) == 0) {
      hipLaunchKernelGGL((transpose_cols_v4_kernel<float, limbw>), dim3((Mpad + 63) / 64, (ncols + 63) / 64), dim3(256), 0, s, (const float*)in, (limbw*)out, M, ld, c0, ncols, Mpad);
    } else {
      hipLaunchKernelGGL((transpose_cols_kernel<float, limbw>), dim3((Mpad + 31) / 32, (ncols + 31) / 32), dim3(256), 0, s, (const float*)in, (limbw*)out, M, ld, c0, ncols, Mpad);
    }
    return (int)hipGetLastError();
  }
  if (((ld | c0 | ncols | Mpad) & 3) == 0) {
    dim3 g4((Mpad + 63) / 64, (ncols + 63) / 64);
    DISPATCH_T(dtype, hipLaunchKernelGGL(transpose_cols_v4_kernel<float>, g4, dim3(256), 0, s, (const float*)in, (float*)out, M, ld, c0, ncols, Mpad),
               hipLaunchKernelGGL(transpose_cols_v4_kernel<bf16>, g4, dim3(256), 0, s, (const bf16*)in, (bf16*)out, M, ld, c0, ncols, Mpad));
    return (int)hipGetLastError();
  }
  dim3 grid((Mpad + 31) / 32, (ncols + 31) / 32);
  DISPATCH_T(dtype, hipLaunchKernelGGL(transpose_cols_kernel<float>, grid, dim3(256), 0, s, (const float*)in, (float*)out, M, ld, c0, ncols, Mpad),
             hipLaunchKernelGGL(transpose_cols_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)in, (bf16*)out, M, ld, c0, ncols, Mpad));
  return (int)hipGetLastError();
}
int launch_im2col_t(const void* x, void* out, int B, int H, int W, int ld, int c0, int C, int KH, int KW, int stride, int pad, int OH, int OW, int Mpad,
                    int dtype, hipStream_t s) {
  if (dtype == 2) {
    if (((ld | c0 | C | Mpad) & 3) == 0) {
      hipLaunchKernelGGL((im2col_t_v4_kernel<float, limbw>), dim3((Mpad + 63) / 64, (C + 63) / 64, KH * KW), dim3(256), 0, s, (const float*)x, (limbw*)out, B, H, W, ld, c0, C, KH, KW,
                         stride, pad, OH, OW, Mpad);
    } else {
      hipLaunchKernelGGL((im2col_t_kernel<float, limbw>), dim3((Mpad + 31) / 32, (C + 31) / 32, KH * KW), dim3(256), 0, s, (const float*)x, (limbw*)out, B, H, W, ld, c0, C, KH, KW, stride,
                         pad, OH, OW, Mpad);
    }
    return (int)hipGetLastError();
  }
  if (((ld | c0 | C | Mpad) & 3) == 0) {
    dim3 g4((Mpad + 63) / 64, (C + 63) / 64, KH * KW);
    DISPATCH_T(dtype, hipLaunchKernelGGL(im2col_t_v4_kernel<float>, g4, dim3(256), 0, s, (const float*)x, (float*)out, B, H, W, ld, c0, C, KH, KW, stride, pad, OH, OW, Mpad),
               hipLaunchKernelGGL(im2col_t_v4_kernel<bf16>, g4, dim3(256), 0, s, (const bf16*)x, (bf16*)out, B, H, W, ld, c0, C, KH, KW, stride, pad, OH, OW, Mpad));
    return (int)hipGetLastError();
  }
  dim3 grid((Mpad + 31) / 32, (C + 31) / 32, KH * KW);
  DISPATCH_T(dtype, hipLaunchKernelGGL(im2col_t_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (float*)out, B, H, W, ld, c0, C, KH, KW, stride, pad, OH, OW, Mpad),
             hipLaunchKernelGGL(im2col_t_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)x, (bf16*)out, B, H, W, ld, c0, C, KH, KW, stride, pad, OH, OW, Mpad));
  return (int)hipGetLastError();
}
int launch_unpatch2(const void* g, void* dx, int B, int OH, int OW, int C, int dtype, hipStream_t s) {
  const size_t total = (size_t)B * OH * OW * 4 * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL(unpatch2_kernel<float>, dim3(gs_grid(total)), dim3(256), 0, s, (const float*)g, (float*)dx, B, OH, OW, C),
             hipLaunchKernelGGL(unpatch2_kernel<bf16>, dim3(gs_grid(total)), dim3(256), 0, s, (const bf16*)g, (bf16*)dx, B, OH, OW, C));
  return (int)hipGetLastError();
}
int bn_reduce_blocks(int M) { int nb = (M + 63) / 64; return nb > 512 ? 512 : nb; }
int launch_bn_reduce(const void* a, const void* z, const float* mean, const float* invstd, float* partial, int M, int C, int bwd, int dtype, hipStream_t s,
                     const void* add_a, const void* add_b, const float* add_scale, int rows_per_img, const float* act_sa, const float* act_sb) {
  const int nb = bn_reduce_blocks(M);
  if (!rows_per_img) rows_per_img = 1;
#define FSVIT_BNR(T, V) do { if (bwd) hipLaunchKernelGGL((bn_reduce_kernel<T, true, V>), dim3(nb), dim3(256), 0, s, (const T*)a, (const T*)z, mean, invstd, partial, M, C, \
                                                         (const T*)nullptr, (const T*)nullptr, (const float*)nullptr, 1, act_sa, act_sb); \
                             else hipLaunchKernelGGL((bn_reduce_kernel<T, false, V>), dim3(nb), dim3(256), 0, s, (const T*)a, (const T*)z, mean, invstd, partial, M, C, \
                                                     (const T*)add_a, (const T*)add_b, add_scale, rows_per_img, (const float*)nullptr, (const float*)nullptr); } while (0)
  if (dtype == 0) FSVIT_BNR(float, 4);
  else if (C % 8 == 0) FSVIT_BNR(bf16, 8);
  else FSVIT_BNR(bf16, 4);
#undef FSVIT_BNR
  return (int)hipGetLastError();
}
int launch_fold_prenorm(const float* W, const float* sa, const float* sb, void* wf, float* bf, int N, int C, int Kw, int dtype, hipStream_t s) {
  DISPATCH_T(dtype, hipLaunchKernelGGL(fold_prenorm_kernel<float>, dim3((N + 3) / 4), dim3(256), 0, s, W, sa, sb, (float*)wf, bf, N, C, Kw),
             hipLaunchKernelGGL(fold_prenorm_kernel<bf16>, dim3((N + 3) / 4), dim3(256), 0, s, W, sa, sb, (bf16*)wf, bf, N, C, Kw));
  return (int)hipGetLastError();
}
int launch_bn_fwd_finalize_nblk(const float* partial, int nblk, int M, int C, float eps, float momentum, const float* gamma, const float* beta, float* rmean, float* rvar,
                                float* mean, float* invstd, float* sa, float* sb, hipStream_t s) {
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, partial, nblk, M, C, eps, momentum, gamma, beta, rmean, rvar, mean, invstd, sa, sb);
  return (int)hipGetLastError();
}
int launch_bn_fwd_finalize(const float* partial, int M, int C, float eps, float momentum, const float* gamma, const float* beta, float* rmean, float* rvar,
                           float* mean, float* invstd, float* sa, float* sb, hipStream_t s) {
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, partial, bn_reduce_blocks(M), M, C, eps, momentum, gamma, beta, rmean, rvar, mean, invstd, sa, sb);
  return (int)hipGetLastError();
}
int launch_bn_bwd_finalize_nblk(const float* partial, int nblk, int M, int C, const float* gamma, const float* invstd, float* dgamma, float* dbeta, float* ca, float* cb,
                                float* cc, int frozen, hipStream_t s) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, partial, nblk, M, C, gamma, invstd, dgamma, dbeta, ca, cb, cc, frozen);
  return (int)hipGetLastError();
}
int launch_bn_bwd_finalize(const float* partial, int M, int C, const float* gamma, const float* invstd, float* dgamma, float* dbeta, float* ca, float* cb,
                           float* cc, int frozen, hipStream_t s) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, partial, bn_reduce_blocks(M), M, C, gamma, invstd, dgamma, dbeta, ca, cb, cc, frozen);
  return (int)hipGetLastError();
}
int launch_bn_frozen_coeffs(int C, float eps, const float* gamma, const float* beta, const float* rmean, const float* rvar, float* mean, float* invstd, float* sa,
                            float* sb, hipStream_t s) {
  hipLaunchKernelGGL(bn_frozen_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, s, C, eps, gamma, beta, rmean, rvar, mean, invstd, sa, sb);
  return (int)hipGetLastError();
}
int launch_bn_apply(const void* z, const float* sa, const float* sb, const void* res, void* y, size_t M, int C, int act, int dtype, hipStream_t s) {
  constexpr bool rows_off = false;
  if (!rows_off && rows_form_ok(C, dtype == 0 ? 4 : 8)) {
    if (dtype == 0) hipLaunchKernelGGL((bn_apply_rows_kernel<float, 4, 4>), dim3(rows_grid(M, C, 4, 4)), dim3(256), 0, s, (const float*)z, sa, sb, (const float*)res, (float*)y, M, C, act);
    else hipLaunchKernelGGL((bn_apply_rows_kernel<bf16, 8, 4>), dim3(rows_grid(M, C, 8, 4)), dim3(256), 0, s, (const bf16*)z, sa, sb, (const bf16*)res, (bf16*)y, M, C, act);
    return (int)hipGetLastError();
  }
  const size_t total = M * (C / 4);
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(gs_grid(total)), dim3(256), 0, s, (const float*)z, sa, sb, (const float*)res, (float*)y, M, C, act),
             hipLaunchKernelGGL(bn_apply_kernel<bf16>, dim3(gs_grid(total)), dim3(256), 0, s, (const bf16*)z, sa, sb, (const bf16*)res, (bf16*)y, M, C, act));
  return (int)hipGetLastError();
}
int launch_bn_act_bwd(const void* dout, const void* z, const float* sa, const float* sb, const void* res, void* g, size_t M, int C, int dtype, hipStream_t s) {
  const size_t total = M * (C / 4);
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_act_bwd_kernel<float>, dim3(gs_grid(total)), dim3(256), 0, s, (const float*)dout, (const float*)z, sa, sb, (const float*)res, (float*)g, M, C),
             hipLaunchKernelGGL(bn_act_bwd_kernel<bf16>, dim3(gs_grid(total)), dim3(256), 0, s, (const bf16*)dout, (const bf16*)z, sa, sb, (const bf16*)res, (bf16*)g, M, C));
  return (int)hipGetLastError();
}
int launch_bn_bwd_apply(const void* dy, const void* z, const float* mean, const float* invstd, const float* ca, const float* cb, const float* cc, void* dz,
                        size_t M, int C, int dtype, hipStream_t s, const void* acc, const float* scale2, void* out2, size_t rows_per_img, const float* act_sa,
                        const float* act_sb) {
  const size_t total = M * (C / 4);
  if (!rows_per_img) rows_per_img = 1;
  constexpr bool rows_off = false;
  if (!rows_off && rows_form_ok(C, dtype == 0 ? 4 : 8)) {
    if (dtype == 0)
      hipLaunchKernelGGL((bn_bwd_apply_rows_kernel<float, 4, 2>), dim3(rows_grid(M, C, 4, 2)), dim3(256), 0, s, (const float*)dy, (const float*)z, mean, invstd, ca, cb, cc, (float*)dz, M, C,
                         (const float*)acc, scale2, (float*)out2, rows_per_img, act_sa, act_sb);
    else
      hipLaunchKernelGGL((bn_bwd_apply_rows_kernel<bf16, 8, 2>), dim3(rows_grid(M, C, 8, 2)), dim3(256), 0, s, (const bf16*)dy, (const bf16*)z, mean, invstd, ca, cb, cc, (bf16*)dz, M, C,
                         (const bf16*)acc, scale2, (bf16*)out2, rows_per_img, act_sa, act_sb);
    return (int)hipGetLastError();
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(gs_grid(total)), dim3(256), 0, s, (const float*)dy, (const float*)z, mean, invstd, ca, cb, cc, (float*)dz, M, C,
                                       (const float*)acc, scale2, (float*)out2, rows_per_img, act_sa, act_sb),
             hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16>, dim3(gs_grid(total)), dim3(256), 0, s, (const bf16*)dy, (const bf16*)z, mean, invstd, ca, cb, cc, (bf16*)dz, M, C,
                                (const bf16*)acc, scale2, (bf16*)out2, rows_per_img, act_sa, act_sb));
  return (int)hipGetLastError();
}
int launch_bn_pool_fwd(const void* z, const float* sa, const float* sb, const void* res, const float* pos, void* out, unsigned char* arg, int B, int OH, int OW,
                       int C, int dtype, hipStream_t s, const float* rsa, const float* rsb) {
  if (C % 8) return (int)hipErrorInvalidValue;
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_pool_fwd_kernel<float, 4>), dim3(gs_grid((size_t)B * OH * OW * (C / 4))), dim3(256), 0, s, (const float*)z, sa, sb, (const float*)res, pos,
                                       (float*)out, arg, B, OH, OW, C, rsa, rsb),
             hipLaunchKernelGGL((bn_pool_fwd_kernel<bf16, 8>), dim3(gs_grid((size_t)B * OH * OW * (C / 8))), dim3(256), 0, s, (const bf16*)z, sa, sb, (const bf16*)res, pos,
                                (bf16*)out, arg, B, OH, OW, C, rsa, rsb));
  return (int)hipGetLastError();
}
// the two BatchNorm backwards behind the stem's pooled tail, from the pooled gradient (pool_bn_bwd_*_kernel)
bool pool_bn_bwd_supported(int C, int dtype) { return rows_form_ok(C, dtype == 0 ? 4 : 8); }
static inline unsigned pool_rows_grid(size_t npix, int C, int V) {
  const size_t R = 256 / (C / V);
  size_t nb = (npix + R - 1) / R;
  return (unsigned)(nb > 512 ? 512 : (nb < 1 ? 1 : nb));
}
int pool_bn_bwd_blocks(int B, int OH, int OW, int C, int dtype) { return (int)pool_rows_grid((size_t)B * OH * OW, C, dtype == 0 ? 4 : 8); }
int launch_pool_bn_bwd_reduce(const void* dout, const unsigned char* arg, const void* z3, const void* zd, const float* mean3, const float* is3, const float* meand,
                              const float* isd, float* partial3, float* partiald, int B, int OH, int OW, int C, int dtype, hipStream_t s) {
  const unsigned nb = pool_rows_grid((size_t)B * OH * OW, C, dtype == 0 ? 4 : 8);
  DISPATCH_T(dtype, hipLaunchKernelGGL((pool_bn_bwd_reduce_kernel<float, 4>), dim3(nb), dim3(256), 0, s, (const float*)dout, arg, (const float*)z3, (const float*)zd, mean3, is3,
                                       meand, isd, partial3, partiald, B, OH, OW, C),
             hipLaunchKernelGGL((pool_bn_bwd_reduce_kernel<bf16, 8>), dim3(nb), dim3(256), 0, s, (const bf16*)dout, arg, (const bf16*)z3, (const bf16*)zd, mean3, is3, meand,
                                isd, partial3, partiald, B, OH, OW, C));
  return (int)hipGetLastError();
}
int launch_pool_bn_bwd_apply(const void* dout, const unsigned char* arg, const void* z3, const void* zd, const float* mean3, const float* is3, const float* meand,
                             const float* isd, const float* coef3, const float* coefd, void* dz3, void* dzd, int B, int OH, int OW, int C, int dtype, hipStream_t s) {
  const size_t npix = (size_t)B * OH * OW;
  const int V = dtype == 0 ? 4 : 8;
  const size_t R = 256 / (C / V);
  size_t nb = (npix + R - 1) / R;
  if (nb > 8192) nb = 8192;
  DISPATCH_T(dtype, hipLaunchKernelGGL((pool_bn_bwd_apply_kernel<float, 4>), dim3((unsigned)nb), dim3(256), 0, s, (const float*)dout, arg, (const float*)z3, (const float*)zd, mean3,
                                       is3, meand, isd, coef3, coefd, (float*)dz3, (float*)dzd, B, OH, OW, C),
             hipLaunchKernelGGL((pool_bn_bwd_apply_kernel<bf16, 8>), dim3((unsigned)nb), dim3(256), 0, s, (const bf16*)dout, arg, (const bf16*)z3, (const bf16*)zd, mean3, is3,
                                meand, isd, coef3, coefd, (bf16*)dz3, (bf16*)dzd, B, OH, OW, C));
  return (int)hipGetLastError();
}
int launch_pool_act_bwd(const void* dout, const unsigned char* arg, void* g, int B, int OH, int OW, int C, int dtype, hipStream_t s) {
  if (C % 8) return (int)hipErrorInvalidValue;
  DISPATCH_T(dtype, hipLaunchKernelGGL((pool_act_bwd_kernel<float, 4>), dim3(gs_grid((size_t)B * OH * OW * (C / 4))), dim3(256), 0, s, (const float*)dout, arg, (float*)g, B, OH, OW, C),
             hipLaunchKernelGGL((pool_act_bwd_kernel<bf16, 8>), dim3(gs_grid((size_t)B * OH * OW * (C / 8))), dim3(256), 0, s, (const bf16*)dout, arg, (bf16*)g, B, OH, OW, C));
  return (int)hipGetLastError();
}
int launch_gelu_fwd(const void* z, void* h, size_t n, int dtype, hipStream_t s) {
  DISPATCH_T(dtype, hipLaunchKernelGGL(gelu_fwd_kernel<float>, dim3(gs_grid(n / 4)), dim3(256), 0, s, (const float*)z, (float*)h, n / 4),
             hipLaunchKernelGGL(gelu_fwd_kernel<bf16>, dim3(gs_grid(n / 4)), dim3(256), 0, s, (const bf16*)z, (bf16*)h, n / 4));
  return (int)hipGetLastError();
}
int launch_gelu_bwd(const void* dh, const void* z, void* dz, size_t n, int dtype, hipStream_t s) {
  DISPATCH_T(dtype, hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(gs_grid(n / 4)), dim3(256), 0, s, (const float*)dh, (const float*)z, (float*)dz, n / 4),
             hipLaunchKernelGGL(gelu_bwd_kernel<bf16>, dim3(gs_grid(n / 4)), dim3(256), 0, s, (const bf16*)dh, (const bf16*)z, (bf16*)dz, n / 4));
  return (int)hipGetLastError();
}
int launch_add_scaled(const void* a, const void* br, const float* scale, void* out, size_t n, size_t per_img, int dtype, hipStream_t s) {
  DISPATCH_T(dtype, hipLaunchKernelGGL(add_scaled_kernel<float>, dim3(gs_grid(n / 4)), dim3(256), 0, s, (const float*)a, (const float*)br, scale, (float*)out, n / 4, per_img / 4),
             hipLaunchKernelGGL(add_scaled_kernel<bf16>, dim3(gs_grid(n / 4)), dim3(256), 0, s, (const bf16*)a, (const bf16*)br, scale, (bf16*)out, n / 4, per_img / 4));
  return (int)hipGetLastError();
}
int launch_maxpool2_idx(const void* in, const float* pos, void* out, unsigned char* arg, int B, int OH, int OW, int C, int dtype, hipStream_t s) {
  const size_t total = (size_t)B * OH * OW * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool2_idx_kernel<float>, dim3(gs_grid(total)), dim3(256), 0, s, (const float*)in, pos, (float*)out, arg, B, OH, OW, C),
             hipLaunchKernelGGL(maxpool2_idx_kernel<bf16>, dim3(gs_grid(total)), dim3(256), 0, s, (const bf16*)in, pos, (bf16*)out, arg, B, OH, OW, C));
  return (int)hipGetLastError();
}
int launch_maxpool2_bwd(const void* dout, const unsigned char* arg, void* din, int B, int OH, int OW, int C, int dtype, hipStream_t s) {
  const size_t total = (size_t)B * OH * OW * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(gs_grid(total)), dim3(256), 0, s, (const float*)dout, arg, (float*)din, B, OH, OW, C),
             hipLaunchKernelGGL(maxpool2_bwd_kernel<bf16>, dim3(gs_grid(total)), dim3(256), 0, s, (const bf16*)dout, arg, (bf16*)din, B, OH, OW, C));
  return (int)hipGetLastError();
}
int launch_avgpool_bwd(const float* dfeat, void* dx, int B, int HW, int C, int dtype, hipStream_t s) {
  const size_t total = (size_t)B * HW * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(gs_grid(total)), dim3(256), 0, s, dfeat, (float*)dx, B, HW, C),
             hipLaunchKernelGGL(avgpool_bwd_kernel<bf16>, dim3(gs_grid(total)), dim3(256), 0, s, dfeat, (bf16*)dx, B, HW, C));
  return (int)hipGetLastError();
}
int launch_batch_sum(const void* g, float* out, int B, size_t per_img, int dtype, hipStream_t s) {
  if (per_img % 4) return (int)hipErrorInvalidValue;
  const unsigned nb = (unsigned)((per_img / 4 + 63) / 64);
  DISPATCH_T(dtype, hipLaunchKernelGGL(batch_sum_kernel<float>, dim3(nb), dim3(1024), 0, s, (const float*)g, out, B, per_img),
             hipLaunchKernelGGL(batch_sum_kernel<bf16>, dim3(nb), dim3(1024), 0, s, (const bf16*)g, out, B, per_img));
  return (int)hipGetLastError();
}
int launch_bcast_add(const void* x, const float* p, void* y, int B, size_t per_img, int dtype, hipStream_t s) {
  const size_t n4 = (size_t)B * per_img / 4;
  DISPATCH_T(dtype, hipLaunchKernelGGL(bcast_add_kernel<float>, dim3(gs_grid(n4)), dim3(256), 0, s, (const float*)x, p, (float*)y, n4, per_img / 4),
             hipLaunchKernelGGL(bcast_add_kernel<bf16>, dim3(gs_grid(n4)), dim3(256), 0, s, (const bf16*)x, p, (bf16*)y, n4, per_img / 4));
  return (int)hipGetLastError();
}
int launch_fill_f32(float* p, float v, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(fill_f32_kernel, dim3(gs_grid(n)), dim3(256), 0, s, p, v, n);
  return (int)hipGetLastError();
}
int launch_scale_copy(const float* in, float* out, size_t n, float sc, hipStream_t s) {
  hipLaunchKernelGGL(scale_copy_kernel, dim3(gs_grid(n)), dim3(256), 0, s, in, out, n, sc);
  return (int)hipGetLastError();
}
int launch_colsum(const void* a, float* partial, float* out, int M, int C, int dtype, hipStream_t s) {
  int rc = launch_bn_reduce(a, nullptr, nullptr, nullptr, partial, M, C, 0, dtype, s);
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, partial, bn_reduce_blocks(M), C, out);
  return (int)hipGetLastError();
}
int launch_sgd_multi(const void* items_dev, int n_items, size_t max_numel, float lr, float momentum, float wd, int first, hipStream_t s) {
  if (n_items <= 0 || max_numel == 0) return 0;
  size_t gx = (max_numel + 1023) / 1024;                 // 4 elements per thread at the largest tensor; smaller ones leave blocks idle
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(sgd_multi_kernel, dim3((unsigned)gx, (unsigned)n_items), dim3(256), 0, s, (const SgdItem*)items_dev, lr, momentum, wd, first);
  return (int)hipGetLastError();
}
int launch_sgd(float* p, const float* g, float* buf, size_t n, float lr, float momentum, float wd, int first, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(sgd_kernel, dim3(gs_grid(n)), dim3(256), 0, s, p, g, buf, n, lr, momentum, wd, first);
  return (int)hipGetLastError();
}

int launch_ln_train_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M, int D, float eps, int dtype, hipStream_t s) {
  if (M <= 0) return 0;
  if (D % 4) return (int)hipErrorInvalidValue;
  DISPATCH_T(dtype, hipLaunchKernelGGL(ln_train_fwd_kernel<float>, dim3((M + 3) / 4), dim3(256), 0, s, (const float*)x, gamma, beta, (float*)y, mean, rstd, M, D, eps),
             hipLaunchKernelGGL(ln_train_fwd_kernel<bf16>, dim3((M + 3) / 4), dim3(256), 0, s, (const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, M, D, eps));
  return (int)hipGetLastError();
}
int ln_bwd_blocks(int M) { int nb = (M + 63) / 64; return nb > 1024 ? 1024 : (nb < 1 ? 1 : nb); }
// partial: ln_bwd_blocks(M) * 2 * D floats; dgamma / dbeta may be null
int launch_ln_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const void* add, void* dx, float* partial,
                  float* dgamma, float* dbeta, int M, int D, int dtype, hipStream_t s) {
  if (M <= 0) return 0;
  if (D % 4 || (size_t)8 * D * 4 > 64 * 1024) return (int)hipErrorInvalidValue;
  const int nb = ln_bwd_blocks(M), rpb = (M + nb - 1) / nb;
  const size_t lds = (size_t)8 * D * sizeof(float);
  constexpr bool rows_off = false;
  const int ng = (D + 255) / 256;
  if (!rows_off && ng <= 4) {
#define FSVIT_LNB(T, NG) hipLaunchKernelGGL((ln_bwd_rows_kernel<T, NG>), dim3(nb), dim3(256), lds, s, (const T*)dy, (const T*)x, mean, rstd, gamma, (const T*)add, (T*)dx, partial, M, D, rpb)
    if (dtype == 0) { if (ng == 1) FSVIT_LNB(float, 1); else if (ng == 2) FSVIT_LNB(float, 2); else if (ng == 3) FSVIT_LNB(float, 3); else FSVIT_LNB(float, 4); }
    else { if (ng == 1) FSVIT_LNB(bf16, 1); else if (ng == 2) FSVIT_LNB(bf16, 2); else if (ng == 3) FSVIT_LNB(bf16, 3); else FSVIT_LNB(bf16, 4); }
#undef FSVIT_LNB
  } else
  DISPATCH_T(dtype, hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(nb), dim3(256), lds, s, (const float*)dy, (const float*)x, mean, rstd, gamma, (const float*)add, (float*)dx, partial, M, D, rpb),
             hipLaunchKernelGGL(ln_bwd_kernel<bf16>, dim3(nb), dim3(256), lds, s, (const bf16*)dy, (const bf16*)x, mean, rstd, gamma, (const bf16*)add, (bf16*)dx, partial, M, D, rpb));
  hipLaunchKernelGGL(ln_param_grad_kernel, dim3((D + 3) / 4), dim3(256), 0, s, partial, nb, D, dgamma, dbeta);
  return (int)hipGetLastError();
}
int launch_vit_assemble(const void* zpe, const float* cls, const float* pos, void* tokens, int B, int S, int D, int dtype, hipStream_t s) {
  const size_t total = (size_t)B * S * (D / 4);
  DISPATCH_T(dtype, hipLaunchKernelGGL(vit_assemble_kernel<float>, dim3(gs_grid(total)), dim3(256), 0, s, (const float*)zpe, cls, pos, (float*)tokens, B, S, D),
             hipLaunchKernelGGL(vit_assemble_kernel<bf16>, dim3(gs_grid(total)), dim3(256), 0, s, (const bf16*)zpe, cls, pos, (bf16*)tokens, B, S, D));
  return (int)hipGetLastError();
}
int launch_vit_patch_rows(const void* dtok, void* dzpe, int B, int S, int D, int dtype, hipStream_t s) {
  const size_t total = (size_t)B * (S - 1) * (D / 4);
  DISPATCH_T(dtype, hipLaunchKernelGGL(vit_patch_rows_kernel<float>, dim3(gs_grid(total)), dim3(256), 0, s, (const float*)dtok, (float*)dzpe, B, S, D),
             hipLaunchKernelGGL(vit_patch_rows_kernel<bf16>, dim3(gs_grid(total)), dim3(256), 0, s, (const bf16*)dtok, (bf16*)dzpe, B, S, D));
  return (int)hipGetLastError();
}
int launch_vit_cls_ln_fwd(const void* tokens, const float* gamma, const float* beta, float* feat, float* mean, float* rstd, int B, int S, int D, float eps, int dtype,
                          hipStream_t s) {
  DISPATCH_T(dtype, hipLaunchKernelGGL(vit_cls_ln_fwd_kernel<float>, dim3(B), dim3(64), 0, s, (const float*)tokens, gamma, beta, feat, mean, rstd, S, D, eps),
             hipLaunchKernelGGL(vit_cls_ln_fwd_kernel<bf16>, dim3(B), dim3(64), 0, s, (const bf16*)tokens, gamma, beta, feat, mean, rstd, S, D, eps));
  return (int)hipGetLastError();
}
// dtok must be zeroed by the caller (only the cls rows are written); partial: B * 2 * D floats
int launch_vit_cls_ln_bwd(const float* dfeat, const void* tokens, const float* mean, const float* rstd, const float* gamma, void* dtok, float* partial, float* dgamma,
                          float* dbeta, int B, int S, int D, int dtype, hipStream_t s) {
  DISPATCH_T(dtype, hipLaunchKernelGGL(vit_cls_ln_bwd_kernel<float>, dim3(B), dim3(64), 0, s, dfeat, (const float*)tokens, mean, rstd, gamma, (float*)dtok, partial, S, D),
             hipLaunchKernelGGL(vit_cls_ln_bwd_kernel<bf16>, dim3(B), dim3(64), 0, s, dfeat, (const bf16*)tokens, mean, rstd, gamma, (bf16*)dtok, partial, S, D));
  hipLaunchKernelGGL(ln_param_grad_kernel, dim3((D + 3) / 4), dim3(256), 0, s, partial, B, D, dgamma, dbeta);
  return (int)hipGetLastError();
}

}  // namespace fsvit
