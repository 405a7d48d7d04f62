// Distillation head of the SUN meta-training phase (sun_meta_training/offline.py, models/token_label.py), fp32:
//   * LinearClassifier forward / backward (classifier.py:27-34) for the per-token head (512 -> n_classes + 1 on every one of the
//     25 tokens) and the global head (512 -> n_classes on the pooled feature) of TokenLabelOffline (token_label.py:36-60);
//   * generate_softlabel (offline.py:57-76): per-token top-k scatter + background-token mask of the teacher's token logits;
//   * SoftTargetCrossEntropy (offline.py:34-45) forward + gradient in one pass;
//   * the AdamW update the phase trains with (offline.py:233).
// Sizes are tiny next to the encoder (B*25 x 512 x 65 per step): these kernels are written for exactness and determinism
// (fixed reduction orders, no atomics), not for the MFMA roofline - one wave per dot product / row.
#include "fsvit_common.h"
#include "kernels.h"

namespace FSVIT_NS {

// y[m][n] = x[m][:] . w[n][:] + b[n];  one workgroup per row m, wave w takes n = w, w + 4, ...; K % 4 == 0
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                         float* __restrict__ y, int M, int N, int K) {
  const int m = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* xr = x + (size_t)m * K;
  for (int n = wave; n < N; n += 4) {
    const float* wr = w + (size_t)n * K;
    float s = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xr + k), c = *reinterpret_cast<const f32x4*>(wr + k);
      s = fmaf(a[0], c[0], s); s = fmaf(a[1], c[1], s); s = fmaf(a[2], c[2], s); s = fmaf(a[3], c[3], s);
    }
    s = wave_sum(s);
    if (lane == 0) y[(size_t)m * N + n] = s + (b ? b[n] : 0.f);
  }
}

// dx[m][k] = sum_n dy[m][n] w[n][k]   (one workgroup per row, thread per k; dy staged through LDS 256 classes at a time, so any N
// works - tieredImageNet pre-training has 351 / 352 classes - and the sum over n keeps its ascending order whatever N is)
__global__ __launch_bounds__(256) void linear_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                           int M, int N, int K, int accumulate) {
  __shared__ float dys[256];
  const int m = blockIdx.x;
  constexpr int KPT = 4;                       // k columns per thread and pass: K <= 1024 in one pass, partial sums stay in registers
  for (int k0 = 0; k0 < K; k0 += 256 * KPT) {
    float s[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) s[j] = 0.f;
    for (int n0 = 0; n0 < N; n0 += 256) {
      const int nn = N - n0 < 256 ? N - n0 : 256;
      __syncthreads();
      if ((int)threadIdx.x < nn) dys[threadIdx.x] = dy[(size_t)m * N + n0 + threadIdx.x];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const int k = k0 + j * 256 + threadIdx.x;
        if (k < K)
          for (int n = 0; n < nn; ++n) s[j] = fmaf(dys[n], w[(size_t)(n0 + n) * K + k], s[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int k = k0 + j * 256 + threadIdx.x;
      if (k < K) {
        float* o = dx + (size_t)m * K + k;
        *o = accumulate ? *o + s[j] : s[j];
      }
    }
  }
}

// dw[n][k] = sum_m dy[m][n] x[m][k],  db[n] = sum_m dy[m][n]   (one workgroup per n, thread per k, fixed order over m)
__global__ __launch_bounds__(256) void linear_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw,
                                                           float* __restrict__ db, int M, int N, int K) {
  const int n = blockIdx.x;
  for (int k = threadIdx.x; k < K; k += 256) {
    float s = 0.f;
    for (int m = 0; m < M; ++m) s = fmaf(dy[(size_t)m * N + n], x[(size_t)m * K + k], s);
    dw[(size_t)n * K + k] = s;
  }
  if (db && threadIdx.x == 0) {
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += dy[(size_t)m * N + n];
    db[n] = s;
  }
}
// The same for many rows (the distillation step's token classifier at batch 512: M = 12 800 rows - the serial loop above took 12.5 ms of a 25 ms
// step): the rows are split into S slabs, a workgroup owns 4 outputs n of one slab (x is re-read N / 4 times, from L2), partial [S][N][K + 1]
// (column K = the bias partial); linear_bwd_w_sum_kernel adds the slabs in fixed order: deterministic, fp32.
__global__ __launch_bounds__(256) void linear_bwd_w_split_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ partial, int M, int N, int K,
                                                                 int rows_per_slab) {
  const int n0 = blockIdx.x * 4, sl = blockIdx.y;
  const int m0 = sl * rows_per_slab, m1 = m0 + rows_per_slab < M ? m0 + rows_per_slab : M;
  float* out = partial + (size_t)sl * N * (K + 1);
  for (int k = threadIdx.x; k < K + 1; k += 256) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int m = m0; m < m1; ++m) {
      const float xv = k < K ? x[(size_t)m * K + k] : 1.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] = fmaf(n0 + j < N ? dy[(size_t)m * N + n0 + j] : 0.f, xv, s[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n0 + j < N) out[(size_t)(n0 + j) * (K + 1) + k] = s[j];
  }
}
__global__ __launch_bounds__(256) void linear_bwd_w_sum_kernel(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ db, int S, int N, int K) {
  const size_t total = (size_t)N * (K + 1);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    float s = 0.f;
    for (int sl = 0; sl < S; ++sl) s += partial[(size_t)sl * total + i];
    const int n = (int)(i / (K + 1)), k = (int)(i % (K + 1));
    if (k < K) dw[(size_t)n * K + k] = s;
    else if (db) db[n] = s;
  }
}

// generate_softlabel: teacher token logits lt [B][T][C] -> soft [B*T][C+1].  One workgroup (64 threads) per image.
//   positive tokens = the T - bp tokens with the largest per-token max (ties: lower token index first);
//   positive row: on_value at its top-k classes (ties: lower class index first), off_value elsewhere;
//   background row: on_value at column BG (= 1: offline.py:61 rebinds `c` to logits_max.size(1) before :71 uses it), off elsewhere.
__global__ __launch_bounds__(64) void token_softlabel_kernel(const float* __restrict__ lt, float* __restrict__ soft, int T, int C, int k, int bp,
                                                             float on_value, float off_value) {
  __shared__ float tmax[64];
  __shared__ int ispos[64];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* base = lt + (size_t)b * T * C;
  if (t < T) {
    float mx = base[(size_t)t * C];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, base[(size_t)t * C + c]);
    tmax[t] = mx;
  }
  __syncthreads();
  if (t < T) {
    int rank = 0;
    for (int u = 0; u < T; ++u) rank += (tmax[u] > tmax[t] || (tmax[u] == tmax[t] && u < t)) ? 1 : 0;
    ispos[t] = rank < T - bp;
  }
  __syncthreads();
  if (t < T) {
    float* row = soft + ((size_t)b * T + t) * (C + 1);
    for (int c = 0; c <= C; ++c) row[c] = off_value;
    if (ispos[t]) {
      const float* z = base + (size_t)t * C;
      float prev = 3.4e38f;
      int prev_idx = -1;
      for (int j = 0; j < k; ++j) {            // j-th largest: the largest value strictly below (prev, prev_idx) in (value desc, index asc) order
        float best = -3.4e38f;
        int bi = -1;
        for (int c = 0; c < C; ++c) {
          const float v = z[c];
          const bool below = v < prev || (v == prev && c > prev_idx);
          if (below && (bi < 0 || v > best)) { best = v; bi = c; }
        }
        if (bi < 0) break;
        row[bi] = on_value;
        prev = best;
        prev_idx = bi;
      }
    } else {
      row[1] = on_value;
    }
  }
}

// SoftTargetCrossEntropy: rowloss[r] = -sum_c t[r][c] log_softmax(z[r])[c];  dz[r][c] = gscale (softmax(z[r])[c] sum_c t[r][c] - t[r][c]).
// One wave per row; lane l owns columns l, l + 64, ... (any C: the tieredImageNet distillation head has 352 columns).
__global__ __launch_bounds__(256) void soft_target_ce_kernel(const float* __restrict__ z, const float* __restrict__ tgt, float* __restrict__ rowloss,
                                                             float* __restrict__ dz, int R, int C, float gscale) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const float* zr = z + (size_t)r * C;
  const float* tr = tgt + (size_t)r * C;
  float mx = -3.4e38f;
  for (int c = lane; c < C; c += 64) mx = fmaxf(mx, zr[c]);
  mx = wave_max(mx);
  float se = 0.f, st = 0.f, tz = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float zc = zr[c], tc = tr[c];
    se += expf(zc - mx);
    st += tc;
    tz += tc * zc;
  }
  se = wave_sum(se);
  st = wave_sum(st);
  tz = wave_sum(tz);
  const float lse = mx + logf(se);
  if (lane == 0) rowloss[r] = lse * st - tz;
  if (dz)
    for (int c = lane; c < C; c += 64) dz[(size_t)r * C + c] = gscale * (expf(zr[c] - mx) / se * st - tr[c]);
}

// F.normalize(x, dim=-1) rows (utils.compute_logits metric 'cos', test_phase/utils/__init__.py:82-84; the nn-classifier head,
// test_phase/models/classifier.py:38-55): y = x / max(|x|, 1e-12), inv[r] = 1 / max(|x_r|, 1e-12) kept for the backward.  One wave per row.
__global__ __launch_bounds__(256) void row_normalize_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv, int R, int D) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const float* xr = x + (size_t)r * D;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) ss = fmaf(xr[d], xr[d], ss);
  const float iv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
  for (int d = lane; d < D; d += 64) y[(size_t)r * D + d] = xr[d] * iv;
  if (lane == 0) inv[r] = iv;
}
// dx = (dy - y <y, dy>) * inv   (the Jacobian of x / |x|; for |x| < 1e-12 torch's clamp makes it dy * 1e12 - not distinguished here)
__global__ __launch_bounds__(256) void row_normalize_bwd_kernel(const float* __restrict__ y, const float* __restrict__ inv, const float* __restrict__ dy,
                                                                float* __restrict__ dx, int R, int D) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const float* yr = y + (size_t)r * D;
  const float* gr = dy + (size_t)r * D;
  float dot = 0.f;
  for (int d = lane; d < D; d += 64) dot = fmaf(yr[d], gr[d], dot);
  dot = wave_sum(dot);
  const float iv = inv[r];
  for (int d = lane; d < D; d += 64) dx[(size_t)r * D + d] = (gr[d] - yr[d] * dot) * iv;
}
int launch_row_normalize(const float* x, float* y, float* inv, int R, int D, hipStream_t s) {
  if (R <= 0) return 0;
  hipLaunchKernelGGL(row_normalize_kernel, dim3((R + 3) / 4), dim3(256), 0, s, x, y, inv, R, D);
  return (int)hipGetLastError();
}
int launch_row_normalize_bwd(const float* y, const float* inv, const float* dy, float* dx, int R, int D, hipStream_t s) {
  if (R <= 0) return 0;
  hipLaunchKernelGGL(row_normalize_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, s, y, inv, dy, dx, R, D);
  return (int)hipGetLastError();
}

// torch.optim.AdamW / timm AdamW (decoupled weight decay), update number `step` (1-based)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                             float beta1, float beta2, float eps, float wd, float bc1, float rsqrt_bc2) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  float pi = p[i] * (1.0f - lr * wd);
  const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
  const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
  p[i] = pi - (lr / bc1) * (mi / denom);
}

// the same update for a table of tensors in ONE launch (blockIdx.y = tensor; rows of 5 x 8 bytes: p, g, m, v, n): the distillation phase steps 89
// parameter tensors, i.e. 89 launches of ~4 us and as many host calls per step (profiles/r05_distill_kernel_stats.csv)
struct AdamwItem { float* p; const float* g; float* m; float* v; size_t n; };
__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamwItem* __restrict__ items, float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                          float rsqrt_bc2) {
  const AdamwItem it = items[blockIdx.y];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < it.n; i += (size_t)gridDim.x * 256) {
    const float gi = it.g[i];
    const float pi = it.p[i] * (1.0f - lr * wd);
    const float mi = beta1 * it.m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * it.v[i] + (1.0f - beta2) * gi * gi;
    it.m[i] = mi;
    it.v[i] = vi;
    const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
    it.p[i] = pi - (lr / bc1) * (mi / denom);
  }
}

// token map plumbing: storage dtype <-> fp32
template <typename T>
__global__ void tokens_to_f32_kernel(const T* __restrict__ in, const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ out, size_t n, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const float v = to_f32<T>(in[i]);
  out[i] = scale ? v * scale[c] + shift[c] : v;
}
template <typename T>
__global__ void add_f32_into_kernel(T* __restrict__ inout, const float* __restrict__ add, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inout[i] = from_f32<T>(to_f32<T>(inout[i]) + add[i]);
}
// out[i] = in[i] (* scale[c] + shift[c] when scale != NULL): the post-norm token map in fp32
int launch_tokens_to_f32(const void* in, const float* scale, const float* shift, float* out, size_t n, int C, int dtype, hipStream_t s) {
  if (n == 0) return 0;
  const unsigned g = (unsigned)((n + 255) / 256);
  if (dtype == 0) hipLaunchKernelGGL(tokens_to_f32_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)in, scale, shift, out, n, C);
  else hipLaunchKernelGGL(tokens_to_f32_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)in, scale, shift, out, n, C);
  return (int)hipGetLastError();
}
int launch_add_f32_into(void* inout, const float* add, size_t n, int dtype, hipStream_t s) {
  if (n == 0) return 0;
  const unsigned g = (unsigned)((n + 255) / 256);
  if (dtype == 0) hipLaunchKernelGGL(add_f32_into_kernel<float>, dim3(g), dim3(256), 0, s, (float*)inout, add, n);
  else hipLaunchKernelGGL(add_f32_into_kernel<bf16>, dim3(g), dim3(256), 0, s, (bf16*)inout, add, n);
  return (int)hipGetLastError();
}

int launch_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, int K, hipStream_t s) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(M), dim3(256), 0, s, x, w, b, y, M, N, K);
  return (int)hipGetLastError();
}
int launch_linear_bwd(const float* dy, const float* x, const float* w, float* dx, int accumulate_dx, float* dw, float* db, int M, int N, int K, hipStream_t s) {
  if (M <= 0) return 0;
  if (dx) hipLaunchKernelGGL(linear_bwd_x_kernel, dim3(M), dim3(256), 0, s, dy, w, dx, M, N, K, accumulate_dx);
  if (dw && M >= 256) {
    // many rows: slabs of >= 32 rows, at most 64 of them; the partials live in a stream-ordered allocation (no workspace in this operator's contract)
    int S = M / 32;
    S = S > 64 ? 64 : S;
    const int rps = (M + S - 1) / S;
    S = (M + rps - 1) / rps;
    float* partial = nullptr;
    hipError_t e = hipMallocAsync((void**)&partial, (size_t)S * N * (K + 1) * 4, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(linear_bwd_w_split_kernel, dim3((N + 3) / 4, S), dim3(256), 0, s, dy, x, partial, M, N, K, rps);
    hipLaunchKernelGGL(linear_bwd_w_sum_kernel, dim3((unsigned)(((size_t)N * (K + 1) + 255) / 256)), dim3(256), 0, s, partial, dw, db, S, N, K);
    e = hipFreeAsync(partial, s);
    if (e != hipSuccess) return (int)e;
  } else if (dw) {
    hipLaunchKernelGGL(linear_bwd_w_kernel, dim3(N), dim3(256), 0, s, dy, x, dw, db, M, N, K);
  }
  return (int)hipGetLastError();
}
int launch_token_softlabel(const float* lt, float* soft, int B, int T, int C, int k, int bp, double smoothing, hipStream_t s) {
  if (B <= 0) return 0;
  const double offd = smoothing / (double)C;                 // python-float arithmetic of offline.py:58-59, rounded once to fp32 by torch.full
  const float off = (float)offd, on = (float)(1.0 - smoothing + offd);
  hipLaunchKernelGGL(token_softlabel_kernel, dim3(B), dim3(64), 0, s, lt, soft, T, C, k, bp, on, off);
  return (int)hipGetLastError();
}
int launch_soft_target_ce(const float* z, const float* tgt, float* rowloss, float* dz, int R, int C, float gscale, hipStream_t s) {
  if (R <= 0) return 0;
  hipLaunchKernelGGL(soft_target_ce_kernel, dim3((R + 3) / 4), dim3(256), 0, s, z, tgt, rowloss, dz, R, C, gscale);
  return (int)hipGetLastError();
}
int launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, float wd, int step, hipStream_t s) {
  if (n == 0) return 0;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, wd, (float)bc1,
                     (float)(1.0 / sqrt(bc2)));
  return (int)hipGetLastError();
}

int launch_adamw_multi(const void* items_dev, int n_items, size_t max_numel, float lr, float beta1, float beta2, float eps, float wd, int step, hipStream_t s) {
  if (n_items <= 0 || max_numel == 0) return 0;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  size_t gx = (max_numel + 1023) / 1024;
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)gx, (unsigned)n_items), dim3(256), 0, s, (const AdamwItem*)items_dev, lr, beta1, beta2, eps, wd, (float)bc1,
                     (float)(1.0 / sqrt(bc2)));
  return (int)hipGetLastError();
}

}  // namespace FSVIT_NS
