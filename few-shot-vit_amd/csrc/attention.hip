// Fused multi-head attention core of the Visformer Attention block
// (test_phase/models/visformer.py:183-190):  softmax((q k^T) * hd^-0.5) v  per (image, head).
//
// Input is the qkv GEMM output in the reference's channel order 'b (x y z) h w' with the head dim
// zero-padded to `hdp` by the weight packer (channel = x*heads*hdp + y*hdp + z), NHWC rows
// [B*S][3*heads*hdp]; output rows [B*S][heads*hdp] feed the proj GEMM ('b (y z) h w').
//
// v1 structure: one 256-thread workgroup per (image, head).  Q, K (row-major) and V^T are staged
// in LDS, q k^T runs on MFMA into an fp32 score matrix in LDS, a wave-parallel softmax rewrites it
// in place as P (storage dtype), and V^T P^T runs on MFMA so each lane stores 4 consecutive
// head-dim values of one token.  S <= 128 tokens (Visformer: 100 and 25).
#include "fsvit_common.h"
#include "kernels.h"

namespace fsvit {

struct AttnGeom {
  int SP;    // tokens rounded up to 16 (MFMA tile rows)
  int SKP;   // keys rounded up to the 64-byte K chunk (PV reduction length)
  int qs;    // Q/K LDS row stride, bytes
  int vs;    // V^T LDS row stride, bytes
  int ss;    // score / P LDS row stride, bytes
  int offK, offV, offS, total;
};

static __host__ __device__ inline AttnGeom attn_geom(int S, int hdp, int es) {
  AttnGeom g;
  const int kchunk = 64 / es;                       // elements per 64-byte chunk
  g.SP = (S + 15) / 16 * 16;
  g.SKP = (S + kchunk - 1) / kchunk * kchunk;
  g.qs = hdp * es + 16;
  g.vs = g.SKP * es + 16;
  const int srow = g.SP * 4 > g.SKP * es ? g.SP * 4 : g.SKP * es;
  g.ss = (srow + 31) / 32 * 32 + 16;
  g.offK = g.SP * g.qs;
  g.offV = g.offK + g.SP * g.qs;
  g.offS = g.offV + hdp * g.vs;
  g.total = g.offS + g.SP * g.ss;
  return g;
}

size_t attention_lds_bytes(int S, int hdp, int dtype) { return (size_t)attn_geom(S, hdp, dtype == 0 ? 4 : 2).total; }

template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ qkv, T* __restrict__ ctx,
                                                        int S, int heads, int hdp, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ES = sizeof(T);
  constexpr int EPC = Elem<T>::kPerChunk;
  const AttnGeom g = attn_geom(S, hdp, ES);
  unsigned char* const Qs = smem;
  unsigned char* const Ks = smem + g.offK;
  unsigned char* const Vt = smem + g.offV;
  unsigned char* const Ss = smem + g.offS;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lrow = lane & 15, lq = lane >> 4;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int rowlen = 3 * heads * hdp;
  const T* base = qkv + (size_t)b * S * rowlen + h * hdp;
  const int cpr = hdp / EPC;                        // 16-byte chunks per head row
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // ---- phase 1: stage Q, K (row-major) and V^T; zero the padding so 0 * pad stays finite
  for (int idx = t; idx < g.SP * cpr; idx += 256) {
    const int row = idx / cpr, ch = idx - row * cpr;
    u32x4 q = zero4, k = zero4;
    if (row < S) {
      const T* src = base + (size_t)row * rowlen + ch * EPC;
      q = *reinterpret_cast<const u32x4*>(src);
      k = *reinterpret_cast<const u32x4*>(src + heads * hdp);
      u32x4 v = *reinterpret_cast<const u32x4*>(src + 2 * heads * hdp);
      const T* ve = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int e = 0; e < EPC; ++e) *reinterpret_cast<T*>(Vt + (ch * EPC + e) * g.vs + row * ES) = ve[e];
    }
    *reinterpret_cast<u32x4*>(Qs + row * g.qs + ch * 16) = q;
    *reinterpret_cast<u32x4*>(Ks + row * g.qs + ch * 16) = k;
  }
  for (int idx = t; idx < hdp * (g.SKP - S); idx += 256) {
    const int d = idx / (g.SKP - S), j = S + idx % (g.SKP - S);
    *reinterpret_cast<T*>(Vt + d * g.vs + j * ES) = from_f32<T>(0.0f);
  }
  __syncthreads();

  // ---- phase 2: scores[q][k] = scale * <Q[q], K[k]>
  const int nt = g.SP / 16;
  const int nkc = hdp * ES / 64;
  for (int id = wave; id < nt * nt; id += 4) {
    const int qt = id / nt, kt = id - qt * nt;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned char* qa = Qs + (qt * 16 + lrow) * g.qs + lq * 16;
    const unsigned char* ka = Ks + (kt * 16 + lrow) * g.qs + lq * 16;
    for (int kc = 0; kc < nkc; ++kc)
      acc = mma_chunk<T>(*reinterpret_cast<const u32x4*>(qa + kc * 64), *reinterpret_cast<const u32x4*>(ka + kc * 64), acc);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<float*>(Ss + (qt * 16 + lq * 4 + r) * g.ss + (kt * 16 + lrow) * 4) = acc[r] * scale;
  }
  __syncthreads();

  // ---- phase 3: row softmax over the S real keys, P written in place (storage dtype), pad keys = 0
  for (int row = wave; row < S; row += 4) {
    unsigned char* srow = Ss + row * g.ss;
    const int c0 = lane, c1 = lane + 64;
    const float v0 = c0 < S ? *reinterpret_cast<const float*>(srow + c0 * 4) : -INFINITY;
    const float v1 = c1 < S ? *reinterpret_cast<const float*>(srow + c1 * 4) : -INFINITY;
    const float m = wave_max(fmaxf(v0, v1));
    const float e0 = c0 < S ? expf(v0 - m) : 0.0f;
    const float e1 = c1 < S ? expf(v1 - m) : 0.0f;
    const float inv = 1.0f / wave_sum(e0 + e1);
    if (c0 < g.SKP) *reinterpret_cast<T*>(srow + c0 * ES) = from_f32<T>(e0 * inv);
    if (c1 < g.SKP) *reinterpret_cast<T*>(srow + c1 * ES) = from_f32<T>(e1 * inv);
  }
  __syncthreads();

  // ---- phase 4: ctx^T[d][q] = sum_k V^T[d][k] P[q][k]
  const int ndt = hdp / 16;
  const int npc = g.SKP * ES / 64;
  T* obase = ctx + (size_t)b * S * heads * hdp + h * hdp;
  for (int id = wave; id < ndt * nt; id += 4) {
    const int qt = id / ndt, dt = id - qt * ndt;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned char* va = Vt + (dt * 16 + lrow) * g.vs + lq * 16;
    const unsigned char* pa = Ss + (qt * 16 + lrow) * g.ss + lq * 16;
    for (int kc = 0; kc < npc; ++kc)
      acc = mma_chunk<T>(*reinterpret_cast<const u32x4*>(va + kc * 64), *reinterpret_cast<const u32x4*>(pa + kc * 64), acc);
    const int q = qt * 16 + lrow;
    if (q < S) store4<T>(obase + (size_t)q * heads * hdp + dt * 16 + lq * 4, acc);
  }
}

int launch_attention(const void* qkv, void* ctx, int B, int S, int heads, int hdp, float scale, int dtype, hipStream_t s) {
  if (B <= 0) return 0;
  const int es = dtype == 0 ? 4 : 2;
  if (S < 1 || S > 128 || hdp % (64 / es) != 0) return (int)hipErrorInvalidValue;
  const size_t lds = attention_lds_bytes(S, hdp, dtype);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  dim3 grid(B * heads), block(256);
  hipError_t e;
  if (dtype == 0) {
    e = hipFuncSetAttribute((const void*)attention_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(attention_kernel<float>, grid, block, lds, s, (const float*)qkv, (float*)ctx, S, heads, hdp, scale);
  } else {
    e = hipFuncSetAttribute((const void*)attention_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(attention_kernel<bf16>, grid, block, lds, s, (const bf16*)qkv, (bf16*)ctx, S, heads, hdp, scale);
  }
  return (int)hipGetLastError();
}

}  // namespace fsvit
