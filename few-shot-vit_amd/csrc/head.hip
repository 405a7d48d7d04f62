// Episode head kernels.
//  * pool_affine: final BatchNorm (eval, folded to scale/shift) + AdaptiveAvgPool2d(1) + flatten
//    (test_phase/models/visformer.py:455-462): feat[b][c] = scale[c] * mean_hw x[b][hw][c] + shift[c].
//  * proto_head: MetaBaseline head (test_phase/models/meta_baseline.py:33-47) with
//    utils.compute_logits 'dot' after F.normalize / 'sqr' (utils/__init__.py:78-101), plus the
//    per-episode accuracy and mean cross-entropy of the eval loop (test_few_shot.py:89-90,
//    utils/__init__.py:104-109) so the host needs no per-episode device sync.
//    One workgroup per episode; wavefront reductions (DPP/shuffle) for norms and dots.
#include "fsvit_common.h"
#include "kernels.h"
#include "train_kernels.h"

namespace FSVIT_NS {

template <typename T>
__global__ __launch_bounds__(256) void pool_affine_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float* __restrict__ feat,
                                                          int HW, int C) {
  const int b = blockIdx.x;
  const T* xb = x + (size_t)b * HW * C;
  const float inv = 1.0f / (float)HW;
  for (int c4 = threadIdx.x; c4 < C / 4; c4 += 256) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < HW; ++p) acc += load4<T>(xb + (size_t)p * C + c4 * 4);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
    *reinterpret_cast<f32x4*>(feat + (size_t)b * C + c4 * 4) = acc * inv * sc + sh;
  }
}

int launch_pool_affine(const void* x, const float* scale, const float* shift, float* feat, int B, int HW, int C, int dtype, hipStream_t s) {
  if (B <= 0) return 0;
  dim3 grid(B), block(256);
  if (dtype == 0) hipLaunchKernelGGL(pool_affine_kernel<float>, grid, block, 0, s, (const float*)x, scale, shift, feat, HW, C);
  else hipLaunchKernelGGL(pool_affine_kernel<bf16>, grid, block, 0, s, (const bf16*)x, scale, shift, feat, HW, C);
  return (int)hipGetLastError();
}

// method: 0 = 'cos' (normalise both, dot), 1 = 'sqr' (negative squared distance to the mean prototype),
//         2 = 'dot' (plain dot product with the mean prototype)
template <int NT>
__global__ __launch_bounds__(NT) void proto_head_kernel(const float* __restrict__ feat_shot, const float* __restrict__ feat_query,
                                                         int way, int shot, int Q, int D, float temp, int method,
                                                         float* __restrict__ logits, float* __restrict__ acc, float* __restrict__ loss,
                                                         const float* __restrict__ temp_dev, const long long* __restrict__ labels = nullptr,
                                                         float* __restrict__ dlogits = nullptr, float* __restrict__ mean_out = nullptr,
                                                         unsigned* __restrict__ ticket = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (temp_dev) temp = *temp_dev;                             // the learnable temperature read where it lives (no host round trip per step)
  float* proto = reinterpret_cast<float*>(smem);              // [way][D]
  float* qstat = proto + (size_t)way * D;                     // [Q][2]: correct flag, nll
  const int e = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* fs = feat_shot + (size_t)e * way * shot * D;
  const float* fq = feat_query + (size_t)e * Q * D;

  // prototypes: mean over shots (meta_baseline.py:37 / :42)
  constexpr int NWV = NT / 64;
  for (int i = t; i < way * D; i += NT) {
    const int c = i / D, d = i - c * D;
    float s = 0.f;
    for (int k = 0; k < shot; ++k) s += fs[((size_t)c * shot + k) * D + d];
    proto[i] = s / (float)shot;
  }
  __syncthreads();
  if (method == 0) {                                          // F.normalize(proto), eps 1e-12 (:38)
    for (int c = wave; c < way; c += NWV) {
      float ss = 0.f;
      for (int d = lane; d < D; d += 64) ss += proto[c * D + d] * proto[c * D + d];
      const float inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
      for (int d = lane; d < D; d += 64) proto[c * D + d] *= inv;
    }
    __syncthreads();
  }
  const int per = Q / way;                                    // queries per class (make_nk_label, few_shot.py:13-16)
  for (int q = wave; q < Q; q += NWV) {
    const float* x = fq + (size_t)q * D;
    float inv = 1.0f;
    if (method == 0) {
      float ss = 0.f;
      for (int d = lane; d < D; d += 64) ss += x[d] * x[d];
      inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);        // F.normalize(query) (:39)
    }
    float best = -INFINITY, mx = -INFINITY;
    int arg = 0;
    float* lrow = logits + ((size_t)e * Q + q) * way;
    float lab_logit = 0.f;
    const int label = labels ? (int)labels[(size_t)e * Q + q] : (per > 0 ? q / per : 0);
    for (int c = 0; c < way; ++c) {
      float s = 0.f;
      if (method != 1) {
        for (int d = lane; d < D; d += 64) s += (x[d] * inv) * proto[c * D + d];
      } else {
        for (int d = lane; d < D; d += 64) { const float df = x[d] - proto[c * D + d]; s -= df * df; }
      }
      const float l = wave_sum(s) * temp;
      if (lane == 0) lrow[c] = l;
      if (l > best) { best = l; arg = c; }                    // first maximum, as torch.argmax
      mx = fmaxf(mx, l);
      if (c == label) lab_logit = l;
    }
    // cross entropy of this query: logsumexp - logit[label]; logits recomputed from global (L1-hot)
    if (lane == 0) {
      float se = 0.f;
      for (int c = 0; c < way; ++c) se += expf(lrow[c] - mx);
      qstat[q * 2 + 0] = (arg == label) ? 1.0f : 0.0f;
      // a label outside [0, way) - including F.cross_entropy's ignore_index = -100, which this head does not implement - poisons the episode's loss
      // (and with it the batch mean) instead of silently scoring the query against no class (the ATen path this replaces raises)
      qstat[q * 2 + 1] = (label >= 0 && label < way) ? mx + logf(se) - lab_logit : __builtin_nanf("");
      if (dlogits) {      // d(mean cross entropy over all E * Q rows) / dlogits = (softmax - onehot) / (E * Q)   (F.cross_entropy, train_meta.py:168)
        const float inv_rows = 1.0f / ((float)gridDim.x * (float)Q), inv_se = 1.0f / se;
        float* drow = dlogits + ((size_t)e * Q + q) * way;
        for (int c = 0; c < way; ++c) drow[c] = (expf(lrow[c] - mx) * inv_se - (c == label ? 1.0f : 0.0f)) * inv_rows;
      }
    }
  }
  __syncthreads();
  if (t == 0) {
    float a = 0.f, l = 0.f;
    for (int q = 0; q < Q; ++q) { a += qstat[q * 2]; l += qstat[q * 2 + 1]; }
    if (acc) acc[e] = a / (float)Q;
    if (loss) loss[e] = l / (float)Q;
    if (mean_out && ticket && acc && loss) {      // batch means by the LAST workgroup to finish, summed in episode order (deterministic); the ticket is left at 0
      __threadfence();
      if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
        __threadfence();
        float sl = 0.f, sa = 0.f;
        for (unsigned i = 0; i < gridDim.x; ++i) { sl += __builtin_nontemporal_load(loss + i); sa += __builtin_nontemporal_load(acc + i); }
        mean_out[0] = sl / (float)gridDim.x;
        mean_out[1] = sa / (float)gridDim.x;
        *ticket = 0u;
      }
    }
  }
}

int launch_proto_head(const float* feat_shot, const float* feat_query, int E, int way, int shot, int Q, int D,
                      float temp, int method, float* logits, float* acc, float* loss, hipStream_t s, const float* temp_dev) {
  return launch_proto_head_ce(feat_shot, feat_query, nullptr, E, way, shot, Q, D, temp, method, logits, acc, loss, nullptr, nullptr, nullptr, s, temp_dev);
}

// head + cross entropy of the meta-tuning step (train_meta.py:167-169) in one launch: logits, per-episode loss / accuracy, dlogits of the mean CE, and the
// two batch means (mean_out[0] = loss, [1] = accuracy) behind a device ticket (a zeroed 4-byte word the kernel leaves zeroed)
int launch_proto_head_ce(const float* feat_shot, const float* feat_query, const long long* labels, int E, int way, int shot, int Q, int D, float temp, int method,
                         float* logits, float* acc, float* loss, float* dlogits, float* mean_out, unsigned* ticket, hipStream_t s, const float* temp_dev) {
  if (E <= 0) return 0;
  const size_t lds = ((size_t)way * D + (size_t)Q * 2) * sizeof(float);
  if (lds > 160 * 1024 || way < 1 || shot < 1 || Q < 1) return (int)hipErrorInvalidValue;
  // few episodes per launch -> latency bound: 16 waves per episode (one wave per query at a time)
  hipError_t e = hipFuncSetAttribute((const void*)proto_head_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(proto_head_kernel<1024>, dim3(E), dim3(1024), lds, s, feat_shot, feat_query, way, shot, Q, D, temp, method,
                     logits, acc, loss, temp_dev, labels, dlogits, mean_out, ticket);
  return (int)hipGetLastError();
}

// Backward of the cosine head (meta_baseline.py:33-47 with method 'cos'): logits = temp * <q^, p^_c>, p_c = mean_s f_shot[c][s].
// One workgroup per episode.  x^ = x / max(|x|, 1e-12):  dx = (dx^ - x^ <x^, dx^>) / |x|.
// (16 waves per episode: the step has only ep_per_batch workgroups, each a chain of wave reductions - with 4 waves the kernel took 275 us)
constexpr int HB_NT = 1024;
// dtemp[gridDim.x] = sum over the episodes' dtemp[e], by the last workgroup to arrive, in episode order (ticket: zeroed word, left zeroed)
__device__ __forceinline__ void head_dtemp_sum(float* dtemp, unsigned* ticket) {
  if (!ticket) return;
  __threadfence();
  if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
    __threadfence();
    float s = 0.f;
    for (unsigned i = 0; i < gridDim.x; ++i) s += __builtin_nontemporal_load(dtemp + i);
    dtemp[gridDim.x] = s;
    *ticket = 0u;
  }
}
__global__ __launch_bounds__(HB_NT) void proto_head_bwd_kernel(const float* __restrict__ feat_shot, const float* __restrict__ feat_query,
                                                             const float* __restrict__ dlogits, int way, int shot, int Q, int D, float temp,
                                                             float* __restrict__ dfeat_shot, float* __restrict__ dfeat_query, float* __restrict__ dtemp,
                                                             const float* __restrict__ temp_dev, const float* __restrict__ up = nullptr,
                                                             unsigned* __restrict__ ticket = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (temp_dev) temp = *temp_dev;
  const float upv = up ? *up : 1.0f;                        // the upstream gradient of a scalar loss (the backward is linear in dlogits)
  temp *= upv;
  float* proto = reinterpret_cast<float*>(smem);            // [way][D] normalised prototypes
  float* dproto = proto + (size_t)way * D;                    // [way][D] gradient w.r.t. the normalised prototypes
  float* pinv = dproto + (size_t)way * D;                     // [way] 1 / |p_c|
  float* qinv = pinv + way;                                   // [Q]   1 / |q|
  float* red = qinv + Q;                                      // [HB_NT / 64] per-wave partial of dtemp
  const int e = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* fs = feat_shot + (size_t)e * way * shot * D;
  const float* fq = feat_query + (size_t)e * Q * D;
  const float* dl = dlogits + (size_t)e * Q * way;
  for (int i = t; i < way * D; i += HB_NT) {
    const int c = i / D, d = i - c * D;
    float s = 0.f;
    for (int k = 0; k < shot; ++k) s += fs[((size_t)c * shot + k) * D + d];
    proto[i] = s / (float)shot;
  }
  __syncthreads();
  for (int c = wave; c < way; c += HB_NT / 64) {
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) ss += proto[c * D + d] * proto[c * D + d];
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
    for (int d = lane; d < D; d += 64) proto[c * D + d] *= inv;
    if (lane == 0) pinv[c] = inv;
  }
  __syncthreads();
  // queries: dq^ = temp * sum_c dl[q][c] p^_c ; dq = (dq^ - q^ <q^, dq^>) / |q| ; dtemp += sum_c dl[q][c] <q^, p^_c>
  float dt_acc = 0.f;
  for (int q = wave; q < Q; q += HB_NT / 64) {
    const float* x = fq + (size_t)q * D;
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) ss += x[d] * x[d];
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
    if (lane == 0) qinv[q] = inv;
    float dotqg = 0.f;
    for (int c = 0; c < way; ++c) {
      float s = 0.f;
      for (int d = lane; d < D; d += 64) s += x[d] * inv * proto[c * D + d];
      s = wave_sum(s);
      dt_acc += dl[q * way + c] * s;
      dotqg += temp * dl[q * way + c] * s;                    // <q^, dq^>
    }
    for (int d = lane; d < D; d += 64) {
      float g = 0.f;
      for (int c = 0; c < way; ++c) g += temp * dl[q * way + c] * proto[c * D + d];
      dfeat_query[((size_t)e * Q + q) * D + d] = (g - x[d] * inv * dotqg) * inv;
    }
  }
  if (lane == 0) red[wave] = dt_acc;
  __syncthreads();
  if (t == 0 && dtemp) {
    float s = 0.f;
    for (int w = 0; w < HB_NT / 64; ++w) s += red[w];
    dtemp[e] = s * upv;
    head_dtemp_sum(dtemp, ticket);
  }
  // prototypes: dp^_c[d] = temp * sum_q dl[q][c] q^[d]
  for (int i = t; i < way * D; i += HB_NT) {
    const int c = i / D, d = i - c * D;
    float g = 0.f;
    for (int q = 0; q < Q; ++q) g += dl[q * way + c] * fq[(size_t)q * D + d] * qinv[q];
    dproto[i] = temp * g;
  }
  __syncthreads();
  for (int c = wave; c < way; c += HB_NT / 64) {
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) dot += proto[c * D + d] * dproto[c * D + d];
    dot = wave_sum(dot);
    for (int d = lane; d < D; d += 64) {
      const float g = (dproto[c * D + d] - proto[c * D + d] * dot) * pinv[c] / (float)shot;
      for (int k = 0; k < shot; ++k) dfeat_shot[(((size_t)e * way + c) * shot + k) * D + d] = g;
    }
  }
}

// Backward of the squared-distance head (meta_baseline.py:38-41 method 'sqr', utils compute_logits 'sqr'): logits[q][c] = -temp * |q - p_c|^2,
// p_c = mean_s f_shot[c][s].  dq = -2 temp sum_c dl (q - p_c);  dp_c = 2 temp sum_q dl (q - p_c);  dtemp = -sum dl |q - p_c|^2.
__global__ __launch_bounds__(256) void proto_head_sqr_bwd_kernel(const float* __restrict__ feat_shot, const float* __restrict__ feat_query,
                                                                 const float* __restrict__ dlogits, int way, int shot, int Q, int D, float temp,
                                                                 float* __restrict__ dfeat_shot, float* __restrict__ dfeat_query, float* __restrict__ dtemp,
                                                                 const float* __restrict__ temp_dev, const float* __restrict__ up = nullptr,
                                                                 unsigned* __restrict__ ticket = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (temp_dev) temp = *temp_dev;
  const float upv = up ? *up : 1.0f;
  temp *= upv;
  float* proto = reinterpret_cast<float*>(smem);            // [way][D]
  float* red = proto + (size_t)way * D;                       // [4]
  const int e = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* fs = feat_shot + (size_t)e * way * shot * D;
  const float* fq = feat_query + (size_t)e * Q * D;
  const float* dl = dlogits + (size_t)e * Q * way;
  for (int i = t; i < way * D; i += 256) {
    const int c = i / D, d = i - c * D;
    float s = 0.f;
    for (int k = 0; k < shot; ++k) s += fs[((size_t)c * shot + k) * D + d];
    proto[i] = s / (float)shot;
  }
  __syncthreads();
  float dt_acc = 0.f;
  for (int q = wave; q < Q; q += 4) {
    const float* x = fq + (size_t)q * D;
    for (int c = 0; c < way; ++c) {
      float s = 0.f;
      for (int d = lane; d < D; d += 64) { const float df = x[d] - proto[c * D + d]; s += df * df; }
      dt_acc -= dl[q * way + c] * wave_sum(s);
    }
    for (int d = lane; d < D; d += 64) {
      float g = 0.f;
      for (int c = 0; c < way; ++c) g += dl[q * way + c] * (x[d] - proto[c * D + d]);
      dfeat_query[((size_t)e * Q + q) * D + d] = -2.0f * temp * g;
    }
  }
  if (lane == 0) red[wave] = dt_acc;
  __syncthreads();
  if (t == 0 && dtemp) {
    dtemp[e] = (red[0] + red[1] + red[2] + red[3]) * upv;
    head_dtemp_sum(dtemp, ticket);
  }
  for (int i = t; i < way * D; i += 256) {
    const int c = i / D, d = i - c * D;
    float g = 0.f;
    for (int q = 0; q < Q; ++q) g += dl[q * way + c] * (fq[(size_t)q * D + d] - proto[i]);
    g *= 2.0f * temp / (float)shot;
    for (int k = 0; k < shot; ++k) dfeat_shot[(((size_t)e * way + c) * shot + k) * D + d] = g;
  }
}

int launch_proto_head_sqr_bwd(const float* feat_shot, const float* feat_query, const float* dlogits, int E, int way, int shot, int Q, int D, float temp,
                              float* dfeat_shot, float* dfeat_query, float* dtemp, hipStream_t s, const float* temp_dev, const float* up, unsigned* ticket) {
  if (E <= 0) return 0;
  const size_t lds = ((size_t)way * D + 4) * sizeof(float);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute((const void*)proto_head_sqr_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(proto_head_sqr_bwd_kernel, dim3(E), dim3(256), lds, s, feat_shot, feat_query, dlogits, way, shot, Q, D, temp, dfeat_shot, dfeat_query, dtemp, temp_dev, up, ticket);
  return (int)hipGetLastError();
}

int launch_proto_head_bwd(const float* feat_shot, const float* feat_query, const float* dlogits, int E, int way, int shot, int Q, int D, float temp,
                          float* dfeat_shot, float* dfeat_query, float* dtemp, hipStream_t s, const float* temp_dev, const float* up, unsigned* ticket) {
  if (E <= 0) return 0;
  const size_t lds = ((size_t)2 * way * D + way + Q + HB_NT / 64) * sizeof(float);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute((const void*)proto_head_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(proto_head_bwd_kernel, dim3(E), dim3(HB_NT), lds, s, feat_shot, feat_query, dlogits, way, shot, Q, D, temp, dfeat_shot, dfeat_query, dtemp, temp_dev, up, ticket);
  return (int)hipGetLastError();
}

}  // namespace FSVIT_NS
