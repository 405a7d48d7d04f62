// Attention backward for the Visformer training step (backward of test_phase/models/visformer.py:183-190):
// given q, k, v (the qkv GEMM output, head dim padded to hdp) and d(ctx), recompute P = softmax(scale q k^T) and form
//   dV = P^T dO,  dP = dO V^T,  dS = scale * P o (dP - rowsum(dP o P)),  dQ = dS K,  dK = dS^T Q.
// Attention is 1.2 % of the step's FLOPs: v1 keeps everything of one (image, head) in LDS as fp32 and
// uses plain FMA loops (fp32 in both numerics modes); an MFMA version is queued behind the GEMM-shaped backward work.
#include "fsvit_common.h"
#include "train_kernels.h"

#include <cstdlib>

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

// fp32, longer sequences (ViT: 197 tokens x head dim 64 do not fit the all-in-LDS kernel above): K and V stay resident, the queries go through in
// blocks of QB rows (Q, dO, P and dS tiles of the block in LDS); dQ of the block is final, dK / dV accumulate in the fp32 output rows themselves -
// element (key, d) is always updated by the same thread of the one workgroup that owns this (image, head), so plain read-modify-write is ordered.
template <int QB>
__global__ __launch_bounds__(256) void attention_bwd_tiled_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx, float* __restrict__ dqkv,
                                                                      int S, int heads, int hd, int hdp, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ld = hd + 1, ls = S + 1;
  float* K = reinterpret_cast<float*>(smem);
  float* V = K + S * ld;
  float* Q = V + S * ld;           // [QB][ld]
  float* dO = Q + QB * ld;
  float* P = dO + QB * ld;         // [QB][ls]
  float* dS = P + QB * ls;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int rowlen = 3 * heads * hdp;
  const float* base = qkv + (size_t)b * S * rowlen + h * hdp;
  const float* dob = dctx + (size_t)b * S * heads * hdp + h * hdp;
  float* dq = dqkv + (size_t)b * S * rowlen + h * hdp;
  for (int idx = t; idx < S * hd; idx += 256) {
    const int i = idx / hd, d = idx - i * hd;
    K[i * ld + d] = base[(size_t)i * rowlen + heads * hdp + d];
    V[i * ld + d] = base[(size_t)i * rowlen + 2 * heads * hdp + d];
  }
  for (int idx = t; idx < S * hdp; idx += 256) {             // dK, dV start at 0 (padded head dims stay exact zeros)
    const int i = idx / hdp, d = idx - i * hdp;
    dq[(size_t)i * rowlen + heads * hdp + d] = 0.f;
    dq[(size_t)i * rowlen + 2 * heads * hdp + d] = 0.f;
  }
  for (int q0 = 0; q0 < S; q0 += QB) {
    const int nq = S - q0 < QB ? S - q0 : QB;
    __syncthreads();
    for (int idx = t; idx < QB * hd; idx += 256) {
      const int i = idx / hd, d = idx - i * hd;
      Q[i * ld + d] = i < nq ? base[(size_t)(q0 + i) * rowlen + d] : 0.f;
      dO[i * ld + d] = i < nq ? dob[(size_t)(q0 + i) * heads * hdp + d] : 0.f;
    }
    __syncthreads();
    for (int idx = t; idx < QB * S; idx += 256) {
      const int i = idx / S, j = idx - i * S;
      float sc = 0.f, dp = 0.f;
      for (int d = 0; d < hd; ++d) { sc += Q[i * ld + d] * K[j * ld + d]; dp += dO[i * ld + d] * V[j * ld + d]; }
      P[i * ls + j] = sc * scale;
      dS[i * ls + j] = dp;
    }
    __syncthreads();
    for (int i = wave; i < QB; i += 4) {
      float m = -INFINITY;
      for (int j = lane; j < S; j += 64) m = fmaxf(m, P[i * ls + j]);
      m = wave_max(m);
      float sum = 0.f;
      for (int j = lane; j < S; j += 64) { const float e = expf(P[i * ls + j] - m); P[i * ls + j] = e; sum += e; }
      const float inv = i < nq ? 1.0f / wave_sum(sum) : 0.f;                 // rows past the end: P = dS = 0
      float dot = 0.f;
      for (int j = lane; j < S; j += 64) { const float p = P[i * ls + j] * inv; P[i * ls + j] = p; dot += p * dS[i * ls + j]; }
      dot = wave_sum(dot);
      for (int j = lane; j < S; j += 64) dS[i * ls + j] = scale * P[i * ls + j] * (dS[i * ls + j] - dot);
    }
    __syncthreads();
    for (int idx = t; idx < nq * hdp; idx += 256) {            // dQ of this block
      const int i = idx / hdp, d = idx - i * hdp;
      float g = 0.f;
      if (d < hd) for (int j = 0; j < S; ++j) g += dS[i * ls + j] * K[j * ld + d];
      dq[(size_t)(q0 + i) * rowlen + d] = g;
    }
    for (int idx = t; idx < S * hd; idx += 256) {              // dK, dV contributions of this block (thread <-> element mapping fixed)
      const int j = idx / hd, d = idx - j * hd;
      float gk = 0.f, gv = 0.f;
      for (int i = 0; i < QB; ++i) { gk += dS[i * ls + j] * Q[i * ld + d]; gv += P[i * ls + j] * dO[i * ld + d]; }
      dq[(size_t)j * rowlen + heads * hdp + d] += gk;
      dq[(size_t)j * rowlen + 2 * heads * hdp + d] += gv;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 MFMA version (round 3; the `parity` and `bf16x2` trainers, Visformer shapes): exact fp32 products on v_mfma_f32_16x16x4_f32 instead of the FMA
// loops above (7.9 of the 66 ms of a two-limb training step).  One workgroup per (image, head), one wave per block of 16 tokens; Q, K, V, dO resident
// in LDS as [KT * 16 rows][DT * 16 + 1] fp32 (rows >= S and columns >= hd zero) plus ONE [query][key] matrix that holds P, then dS.
// A 16x16x4 MFMA takes one float per lane and operand - A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15] - so either orientation of a
// row-major LDS array is a plain ds_read_b32 (no transposed copies), and the result D[4 (lane >> 4) + r][lane & 15] puts 4 consecutive A-rows of one
// B-row in a lane.  Four phases, three barriers, no cross-wave reduction (every output element has ONE owner: bit-reproducible):
//   A  wave = query block: S^T and dP^T over all keys (A = K / V rows, B = Q / dO rows: a lane holds 4 keys of ITS query), softmax statistics inside the
//      4 lanes of a query (two shuffles), P -> LDS;            B  wave = key block: dV^T = dO^T P over all queries -> global;
//   C  wave = query block: dS = scale P (dP - sum_j P dP) from the registers of phase A -> LDS over P;
//   D  wave = key block: dK^T = Q^T dS -> global; wave = query block: dQ^T = K^T dS^T -> global.
template <int KT, int DT>
__global__ __launch_bounds__(KT * 64) void attention_bwd_f32mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx, float* __restrict__ dqkv,
                                                                        int S, int heads, int hd, int hdp, float scale) {
  constexpr int SP = KT * 16, HD = DT * 16, LD = HD + 1, PP = SP + 4, NT = KT * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Q = reinterpret_cast<float*>(smem);
  float* K = Q + SP * LD;
  float* V = K + SP * LD;
  float* dO = V + SP * LD;
  float* PS = dO + SP * LD;                             // [SP queries][PP]: P, then dS
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lrow = lane & 15, lq = lane >> 4;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int rowlen = 3 * heads * hdp;
  const float* base = qkv + (size_t)b * S * rowlen + h * hdp;
  const float* dob = dctx + (size_t)b * S * heads * hdp + h * hdp;
  float* dq = dqkv + (size_t)b * S * rowlen + h * hdp;
  {   // 16-byte global loads (hdp % 4 == 0: the launcher checks), ALL in flight before the first LDS store: unconditional on clamped coordinates
    constexpr int NIT = (SP * (HD / 4) + NT - 1) / NT;
    f32x4 vq[NIT], vk[NIT], vv[NIT], vo[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = t + it * NT, i = idx / (HD / 4), d = (idx - i * (HD / 4)) * 4;
      const bool in = i < S && d < hdp;
      const float* r = base + (size_t)(in ? i : 0) * rowlen + (in ? d : 0);
      vq[it] = *reinterpret_cast<const f32x4*>(r);
      vk[it] = *reinterpret_cast<const f32x4*>(r + heads * hdp);
      vv[it] = *reinterpret_cast<const f32x4*>(r + 2 * heads * hdp);
      vo[it] = *reinterpret_cast<const f32x4*>(dob + (size_t)(in ? i : 0) * heads * hdp + (in ? d : 0));
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = t + it * NT, i = idx / (HD / 4), d = (idx - i * (HD / 4)) * 4;
      if (idx < SP * (HD / 4)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = i < S && d + e < hd;
          Q[i * LD + d + e] = ok ? vq[it][e] : 0.f;
          K[i * LD + d + e] = ok ? vk[it][e] : 0.f;
          V[i * LD + d + e] = ok ? vv[it][e] : 0.f;
          dO[i * LD + d + e] = ok ? vo[it][e] : 0.f;
        }
      }
    }
  }
  __syncthreads();
  const int qb = wave, kb = wave;                       // this wave's query block and key block
  const float* Qb = Q + (qb * 16) * LD;
  const float* Ob = dO + (qb * 16) * LD;
  // ---- A: S^T / dP^T (lane = query qb*16 + lrow, keys kt*16 + 4 lq + r); k outermost: 2 KT independent accumulators per step
  f32x4 sc[KT], dp[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) sc[kt] = dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k0 = 0; k0 < HD; k0 += 4) {
    const float bq = Qb[lrow * LD + k0 + lq], bo = Ob[lrow * LD + k0 + lq];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(K[(kt * 16 + lrow) * LD + k0 + lq], bq, sc[kt], 0, 0, 0);
      dp[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[(kt * 16 + lrow) * LD + k0 + lq], bo, dp[kt], 0, 0, 0);
    }
  }
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = kt * 16 + 4 * lq + r < S ? sc[kt][r] * scale : -INFINITY;
      sc[kt][r] = v;
      m = fmaxf(m, v);
    }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float e = expf(sc[kt][r] - m); sc[kt][r] = e; sum += e; }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  float dot = 0.f;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float pr = sc[kt][r] * inv; sc[kt][r] = pr; dot += pr * dp[kt][r]; }
  dot += __shfl_xor(dot, 16, 64);
  dot += __shfl_xor(dot, 32, 64);
  float* PSq = PS + (qb * 16 + lrow) * PP;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) *reinterpret_cast<f32x4*>(PSq + kt * 16 + 4 * lq) = sc[kt];
  __syncthreads();
  // ---- B: dV^T[d][key] = sum_q dO[q][d] P[q][key] for this wave's key block -> global (lane: key kb*16 + lrow, d = dt*16 + 4 lq + r)
  auto key_block_product = [&](const float* X, int which) {      // X = dO (-> dV, which = 2) or Q (-> dK, which = 1)
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k0 = 0; k0 < SP; k0 += 4) {
      const float bp = PS[(k0 + lq) * PP + kb * 16 + lrow];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(X[(k0 + lq) * LD + dt * 16 + lrow], bp, acc[dt], 0, 0, 0);
    }
    const int key = kb * 16 + lrow;
    if (key < S) {
      float* r0 = dq + (size_t)key * rowlen + which * heads * hdp;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = dt * 16 + 4 * lq;
        if (d + 3 < hdp) *reinterpret_cast<f32x4*>(r0 + d) = acc[dt];         // (columns hd .. hdp: the operand is zero there -> exact zeros)
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (d + r < hdp) r0[d + r] = acc[dt][r];
      }
    }
  };
  key_block_product(dO, 2);
  __syncthreads();
  // ---- C: dS over P
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    f32x4 ds;
#pragma unroll
    for (int r = 0; r < 4; ++r) ds[r] = scale * sc[kt][r] * (dp[kt][r] - dot);
    *reinterpret_cast<f32x4*>(PSq + kt * 16 + 4 * lq) = ds;
  }
  __syncthreads();
  // ---- D: dK^T for the key block; dQ^T[d][q] = sum_key K[key][d] dS[q][key] for the query block (lane: query lrow, d = dt*16 + 4 lq + r)
  key_block_product(Q, 1);
  {
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k0 = 0; k0 < SP; k0 += 4) {
      const float bs = PSq[k0 + lq];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(K[(k0 + lq) * LD + dt * 16 + lrow], bs, acc[dt], 0, 0, 0);
    }
    const int qrow = qb * 16 + lrow;
    if (qrow < S) {
      float* r0 = dq + (size_t)qrow * rowlen;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = dt * 16 + 4 * lq;
        if (d + 3 < hdp) *reinterpret_cast<f32x4*>(r0 + d) = acc[dt];
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (d + r < hdp) r0[d + r] = acc[dt][r];
      }
    }
  }
}

template <int KT, int DT>
static int launch_bwd_f32mfma(const void* qkv, const void* dctx, void* dqkv, int B, int S, int heads, int hd, int hdp, float scale, hipStream_t s) {
  constexpr int SP = KT * 16, LD = DT * 16 + 1, PP = SP + 4;
  const size_t lds = ((size_t)4 * SP * LD + (size_t)SP * PP) * sizeof(float);
  if ((hdp & 3) || hdp > DT * 16) return (int)hipErrorInvalidValue;
  auto kern = attention_bwd_f32mfma_kernel<KT, DT>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(B * heads), dim3(KT * 64), lds, s, (const float*)qkv, (const float*)dctx, (float*)dqkv, S, heads, hd, hdp, scale);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 MFMA, long sequences (ViT / DeiT: 197 tokens x 64; the `parity` and `bf16x2` ViT trainers): Q, K, V, dO do not fit the LDS together (216 KB),
// so the head is walked twice, flash-style, with TWO of the four matrices resident at a time (111 KB) and the other two entering as per-lane B operands
// straight from global memory (16 floats per lane and matrix).  attention_bwd_tiled_f32_kernel (FMA loops) took 9.6 ms per DeiT-S launch = 73 % of a
// two-limb training step.
//   pass 1 (K, V resident; a wave owns query blocks): S^T / dP^T over all keys (lane = query lrow, keys 4 lq + r), the softmax statistics
//          (max, 1 / sum, sum_j P dP) -> LDS, dS in registers, dQ^T = K^T dS^T -> global;
//   pass 2 (Q, dO resident; a wave owns key blocks): per query block S / dP recomputed in the OTHER orientation (lane = key lrow, queries 4 lq + r),
//          P and dS from the stored statistics, dV^T += dO^T P, dK^T += Q^T dS -> global.
// No matrix is ever transposed or staged for the second product of a chain: a 16x16x4 step may sum ANY four k indices as long as both operands agree,
// so step (tile, r) takes k = lane >> 4 -> index 16 tile + 4 (lane >> 4) + r: the B operand is the lane's own result register [r] of the previous product.
template <int KT, int DT, int NW>
__global__ __launch_bounds__(NW * 64) void attention_bwd_f32mfma_long_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx, float* __restrict__ dqkv,
                                                                             int S, int heads, int hd, int hdp, float scale) {
  constexpr int SP = KT * 16, HD = DT * 16, LD = HD + 1, NT = NW * 64, KS = HD / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* A0 = reinterpret_cast<float*>(smem);          // pass 1: K, pass 2: Q        [SP][LD]
  float* A1 = A0 + SP * LD;                            // pass 1: V, pass 2: dO
  float* st_m = A1 + SP * LD;                          // per query: row maximum (scaled scores), 1 / sum, sum_j P dP
  float* st_i = st_m + SP;
  float* st_d = st_i + SP;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lrow = lane & 15, lq = lane >> 4;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int rowlen = 3 * heads * hdp;
  const float* base = qkv + (size_t)b * S * rowlen + h * hdp;
  const float* dob = dctx + (size_t)b * S * heads * hdp + h * hdp;
  const size_t dorow = (size_t)heads * hdp;
  float* dq = dqkv + (size_t)b * S * rowlen + h * hdp;
  // two matrices ([S][hd], row strides sa / sb floats) -> A0 / A1, zero padded; all loads in flight before the first LDS store
  auto stage2 = [&](const float* pa, size_t sa, const float* pb, size_t sb) {
    constexpr int NIT = (SP * (HD / 4) + NT - 1) / NT;
    f32x4 va[NIT], vb[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = t + it * NT, i = idx / (HD / 4), d = (idx - i * (HD / 4)) * 4;
      const bool in = i < S && d < hdp;
      va[it] = *reinterpret_cast<const f32x4*>(pa + (size_t)(in ? i : 0) * sa + (in ? d : 0));
      vb[it] = *reinterpret_cast<const f32x4*>(pb + (size_t)(in ? i : 0) * sb + (in ? d : 0));
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = t + it * NT, i = idx / (HD / 4), d = (idx - i * (HD / 4)) * 4;
      if (idx < SP * (HD / 4)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = i < S && d + e < hd;
          A0[i * LD + d + e] = ok ? va[it][e] : 0.f;
          A1[i * LD + d + e] = ok ? vb[it][e] : 0.f;
        }
      }
    }
  };
  // the rows blk * 16 + lrow of a matrix as B operands: element [k = 4 s + lq] for the KS steps (zero past the row / the head dim)
  auto rows_b = [&](const float* pm, size_t stride, int blk, float (&o)[KS]) {
    const int row = blk * 16 + lrow;
#pragma unroll
    for (int s0 = 0; s0 < KS; ++s0) {
      const int d = 4 * s0 + lq;
      const bool in = row < S && d < hd;
      const float v = pm[(size_t)(in ? row : 0) * stride + (in ? d : 0)];
      o[s0] = in ? v : 0.f;
    }
  };

  // ---------------- pass 1
  stage2(base + heads * hdp, rowlen, base + 2 * heads * hdp, rowlen);          // K, V
  __syncthreads();
  for (int qb = wave; qb < KT; qb += NW) {
    float bq[KS], bo[KS];
    rows_b(base, rowlen, qb, bq);
    rows_b(dob, dorow, qb, bo);
    f32x4 sc[KT], dp[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) sc[kt] = dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s0 = 0; s0 < KS; ++s0) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[(kt * 16 + lrow) * LD + 4 * s0 + lq], bq[s0], sc[kt], 0, 0, 0);
        dp[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[(kt * 16 + lrow) * LD + 4 * s0 + lq], bo[s0], dp[kt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);               // (keeps hipcc from hoisting every fragment read of the 416-MFMA block to its top: 98 spilled VGPRs)
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = kt * 16 + 4 * lq + r < S ? sc[kt][r] * scale : -INFINITY;
        sc[kt][r] = v;
        m = fmaxf(m, v);
      }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float e = expf(sc[kt][r] - m); sc[kt][r] = e; sum += e; }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    float dot = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float pr = sc[kt][r] * inv; sc[kt][r] = pr; dot += pr * dp[kt][r]; }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
    const int qrow = qb * 16 + lrow;
    if (lq == 0) { st_m[qrow] = m; st_i[qrow] = qrow < S ? inv : 0.f; st_d[qrow] = dot; }      // (queries past the end: P = 0 in pass 2)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[kt][r] = scale * sc[kt][r] * (dp[kt][r] - dot);           // dS
    // dQ^T[d][q] = sum_key K[key][d] dS[q][key]: step (kt, r) sums the keys 16 kt + 4 lq + r
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[(kt * 16 + 4 * lq + r) * LD + dt * 16 + lrow], sc[kt][r], acc[dt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (qrow < S) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = dt * 16 + 4 * lq;
        if (d + 3 < hdp) *reinterpret_cast<f32x4*>(dq + (size_t)qrow * rowlen + d) = acc[dt];   // (columns hd .. hdp: K is zero there -> exact zeros)
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (d + r < hdp) dq[(size_t)qrow * rowlen + d + r] = acc[dt][r];
      }
    }
  }
  __syncthreads();
  // ---------------- pass 2
  stage2(base, rowlen, dob, dorow);                                            // Q, dO
  __syncthreads();
  for (int kb = wave; kb < KT; kb += NW) {
    float bk[KS], bv[KS];
    rows_b(base + heads * hdp, rowlen, kb, bk);
    rows_b(base + 2 * heads * hdp, rowlen, kb, bv);
    const bool key_ok = kb * 16 + lrow < S;
    f32x4 accV[DT], accK[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) accV[dt] = accK[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int qb = 0; qb < KT; ++qb) {
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, p4 = s4;          // lane = key lrow, queries qb*16 + 4 lq + r
#pragma unroll
      for (int s0 = 0; s0 < KS; ++s0) {
        s4 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[(qb * 16 + lrow) * LD + 4 * s0 + lq], bk[s0], s4, 0, 0, 0);
        p4 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[(qb * 16 + lrow) * LD + 4 * s0 + lq], bv[s0], p4, 0, 0, 0);
      }
      const f32x4 m4 = *reinterpret_cast<const f32x4*>(st_m + qb * 16 + 4 * lq), i4 = *reinterpret_cast<const f32x4*>(st_i + qb * 16 + 4 * lq),
                  d4 = *reinterpret_cast<const f32x4*>(st_d + qb * 16 + 4 * lq);
      f32x4 P, dS;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = key_ok ? expf(s4[r] * scale - m4[r]) * i4[r] : 0.f;
        P[r] = pr;
        dS[r] = scale * pr * (p4[r] - d4[r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          accV[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[(qb * 16 + 4 * lq + r) * LD + dt * 16 + lrow], P[r], accV[dt], 0, 0, 0);
          accK[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[(qb * 16 + 4 * lq + r) * LD + dt * 16 + lrow], dS[r], accK[dt], 0, 0, 0);
        }
    }
    const int key = kb * 16 + lrow;
    if (key < S) {
      float* rk = dq + (size_t)key * rowlen + heads * hdp;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = dt * 16 + 4 * lq;
        if (d + 3 < hdp) { *reinterpret_cast<f32x4*>(rk + d) = accK[dt]; *reinterpret_cast<f32x4*>(rk + heads * hdp + d) = accV[dt]; }
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (d + r < hdp) { rk[d + r] = accK[dt][r]; rk[heads * hdp + d + r] = accV[dt][r]; }
      }
    }
  }
}

template <int KT, int DT, int NW>
static int launch_bwd_f32mfma_long(const void* qkv, const void* dctx, void* dqkv, int B, int S, int heads, int hd, int hdp, float scale, hipStream_t s) {
  constexpr int SP = KT * 16, LD = DT * 16 + 1;
  const size_t lds = ((size_t)2 * SP * LD + (size_t)3 * SP) * sizeof(float);
  if ((hdp & 3) || hdp > DT * 16) return (int)hipErrorInvalidValue;
  auto kern = attention_bwd_f32mfma_long_kernel<KT, DT, NW>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(B * heads), dim3(NW * 64), lds, s, (const float*)qkv, (const float*)dctx, (float*)dqkv, S, heads, hd, hdp, scale);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 MFMA version.  One workgroup per (image, head); S <= 224 keys, everything of the head lives in LDS, ROW-major only:
//   Q, K, V, dO  [SKP][HDP]   score-type products read 16-byte k-chunks of a row; the three products that contract over keys / queries take
//   their A operand (8 consecutive rows of one column) from the same images with ds_read_b64_tr_b16 (tools/probes/tr_read_probe.hip) -
//   round 1 kept transposed copies K^T, Q^T, dO^T written with 2-byte stores (126 KB at 100 tokens, S <= 128)
// pass 1 (per 16-query tile, one wave): S^T = K Q^T and dP^T = V dO^T in the [key x query] orientation, in-register
//   softmax (2 shuffles), D = rowsum(P o dP), dS = scale P o (dP - D); dQ^T = K^T dS^T with dS fed back as the MFMA B
//   operand straight from registers (keys permuted consistently in both operands, as attention_v2 does for P V).
//   The per-query statistics (max, 1/sum, D) go to LDS.
// pass 2 (per 16-key tile, one wave): S and dP recomputed in the [query x key] orientation from the statistics, so
//   P and dS are again B operands from registers: dV^T = dO^T P, dK^T = Q^T dS.
template <int NKT, int NDT, int NW>
__global__ __launch_bounds__(NW * 64) void attention_bwd_mfma_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dctx,
                                                                     bf16* __restrict__ dqkv, int S, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int HDP = NDT * 16, SKP = NKT * 16;
  constexpr int RS = HDP * 2 + 32;             // row-major row stride (bytes): 32 mod 64 - the conflict-free pitch for b128 reads whose 16-lane groups mix rows of two lq (round 6: 244 -> 233 us at the stage-2 shape; an odd multiple of 16 - 144 before - reads 2-way)
  constexpr int NKC = HDP / 32;                // 32-element MFMA chunks along the head dim
  constexpr int CPR = HDP / 8;                 // 16-byte chunks per row
  unsigned char* const Qr = smem;
  unsigned char* const Kr = Qr + SKP * RS;
  unsigned char* const Vr = Kr + SKP * RS;
  unsigned char* const Or = Vr + SKP * RS;
  float* const st_m = reinterpret_cast<float*>(Or + SKP * RS);
  float* const st_inv = st_m + SKP;
  float* const st_D = st_inv + SKP;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lrow = lane & 15, lq = lane >> 4;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int rowlen = 3 * heads * HDP;
  const bf16* base = qkv + (size_t)b * S * rowlen + h * HDP;
  const bf16* dob = dctx + (size_t)b * S * heads * HDP + h * HDP;
  bf16* dbase = dqkv + (size_t)b * S * rowlen + h * HDP;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  {   // all staging loads of the head in flight at once: UNCONDITIONAL on clamped rows (rows >= S are zeroed at the LDS store).  The loop form - `if (row
      // < S) { 4 loads }`, 4 stores, next iteration - waited for each iteration's loads before issuing the next ones: 4 HBM latencies per head.
    constexpr int NIT = (SKP * CPR + NW * 64 - 1) / (NW * 64);
    u32x4 q[NIT], k[NIT], v[NIT], o[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = t + it * NW * 64, row = idx / CPR, ch = idx - row * CPR;
      const int rc = row < S ? row : S - 1;
      const bf16* src = base + (size_t)rc * rowlen + ch * 8;
      q[it] = *reinterpret_cast<const u32x4*>(src);
      k[it] = *reinterpret_cast<const u32x4*>(src + heads * HDP);
      v[it] = *reinterpret_cast<const u32x4*>(src + 2 * heads * HDP);
      o[it] = *reinterpret_cast<const u32x4*>(dob + (size_t)rc * heads * HDP + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = t + it * NW * 64, row = idx / CPR, ch = idx - row * CPR;
      if (idx < SKP * CPR) {
        const bool ok = row < S;
        *reinterpret_cast<u32x4*>(Qr + row * RS + ch * 16) = ok ? q[it] : zero4;
        *reinterpret_cast<u32x4*>(Kr + row * RS + ch * 16) = ok ? k[it] : zero4;
        *reinterpret_cast<u32x4*>(Vr + row * RS + ch * 16) = ok ? v[it] : zero4;
        *reinterpret_cast<u32x4*>(Or + row * RS + ch * 16) = ok ? o[it] : zero4;
      }
    }
  }
  __syncthreads();

  const float sscale = scale * 1.44269504088896340736f;        // exp2 domain, exactly as the forward kernel
  // A fragment (16 columns c0 .. c0 + 15 as MFMA rows, K slots = rows r0 + 4 lq + j and r0 + 16 + 4 lq + j) of a row-major image: this lane
  // supplies the address of columns 4 (lrow & 3) .. of row r0 + 4 lq + (lrow >> 2) and receives rows r0 + 4 lq + 0..3 of column c0 + lrow
  typedef short s4t __attribute__((ext_vector_type(4)));
  auto trA = [&](const unsigned char* img, int r0, int c0) -> u32x4 {
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)img + (r0 + lq * 4 + (lrow >> 2)) * RS + (c0 + (lrow & 3) * 4) * 2;
    const u32x2 v0 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4t*)(size_t)a));
    const u32x2 v1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4t*)(size_t)(a + 16 * RS)));
    return u32x4{v0[0], v0[1], v1[0], v1[1]};
  };
  const int nt = (S + 15) / 16;

  // ------------------------------------------------------------------ pass 1: dQ + per-query statistics
  for (int qt = wave; qt < NKT; qt += NW) {
    const int q = qt * 16 + lrow;
    if (qt >= nt) {                                               // padded query tile: P = 0 in pass 2
      if (lq == 0) { st_m[q] = 0.f; st_inv[q] = 0.f; st_D[q] = 0.f; }
      continue;
    }
    u32x4 qf[NKC], of[NKC];
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      qf[kc] = *reinterpret_cast<const u32x4*>(Qr + q * RS + (kc * 4 + lq) * 16);
      of[kc] = *reinterpret_cast<const u32x4*>(Or + q * RS + (kc * 4 + lq) * 16);
    }
    f32x4 sc[NKT], dp[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
      const int krow = (kt * 16 + lrow) * RS + lq * 16;
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc) {
        a0 = mma_chunk<bf16>(*reinterpret_cast<const u32x4*>(Kr + krow + kc * 64), qf[kc], a0);
        a1 = mma_chunk<bf16>(*reinterpret_cast<const u32x4*>(Vr + krow + kc * 64), of[kc], a1);
      }
      sc[kt] = a0;                       // keys kt*16 + lq*4 + r  x  query lrow
      dp[kt] = a1;
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = kt * 16 + lq * 4 + r < S;
        sc[kt][r] = ok ? sc[kt][r] * sscale : -INFINITY;
        m = fmaxf(m, sc[kt][r]);
      }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(sc[kt][r] - m);
        sc[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    float D = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { sc[kt][r] *= inv; D += sc[kt][r] * dp[kt][r]; }
    D += __shfl_xor(D, 16, 64);
    D += __shfl_xor(D, 32, 64);
    const bool qok = q < S;
    if (lq == 0) { st_m[q] = m; st_inv[q] = qok ? inv : 0.f; st_D[q] = D; }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[kt][r] = scale * sc[kt][r] * (dp[kt][r] - D);       // dS
    // dQ^T[d][q] = sum_key K^T[d][key] * dS[q][key]
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < NKT / 2; ++kc) {
        const u32x4 kf = trA(Kr, 32 * kc, dt * 16);
        const bf16x8 sb = {(bf16)sc[2 * kc][0], (bf16)sc[2 * kc][1], (bf16)sc[2 * kc][2], (bf16)sc[2 * kc][3],
                           (bf16)sc[2 * kc + 1][0], (bf16)sc[2 * kc + 1][1], (bf16)sc[2 * kc + 1][2], (bf16)sc[2 * kc + 1][3]};
        acc = mma_chunk<bf16>(kf, __builtin_bit_cast(u32x4, sb), acc);
      }
      if (qok) store4<bf16>(dbase + (size_t)q * rowlen + dt * 16 + lq * 4, acc);
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ pass 2: dK, dV
  for (int jt = wave; jt < nt; jt += NW) {
    const int key = jt * 16 + lrow;
    const bool kok = key < S;
    u32x4 kf[NKC], vf[NKC];
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      kf[kc] = *reinterpret_cast<const u32x4*>(Kr + key * RS + (kc * 4 + lq) * 16);
      vf[kc] = *reinterpret_cast<const u32x4*>(Vr + key * RS + (kc * 4 + lq) * 16);
    }
    f32x4 accK[NDT], accV[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) { accK[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; accV[dt] = accK[dt]; }
#pragma unroll
    for (int qc = 0; qc < NKT / 2; ++qc) {
      bf16x8 pb, sb;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int it = 2 * qc + half;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        const int qrow = (it * 16 + lrow) * RS + lq * 16;
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
          a0 = mma_chunk<bf16>(*reinterpret_cast<const u32x4*>(Qr + qrow + kc * 64), kf[kc], a0);    // queries it*16 + lq*4 + r  x  key lrow
          a1 = mma_chunk<bf16>(*reinterpret_cast<const u32x4*>(Or + qrow + kc * 64), vf[kc], a1);
        }
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(st_m + it * 16 + lq * 4);
        const f32x4 i4 = *reinterpret_cast<const f32x4*>(st_inv + it * 16 + lq * 4);
        const f32x4 D4 = *reinterpret_cast<const f32x4*>(st_D + it * 16 + lq * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = kok ? __builtin_amdgcn_exp2f(a0[r] * sscale - m4[r]) * i4[r] : 0.f;
          pb[half * 4 + r] = (bf16)pr;
          sb[half * 4 + r] = (bf16)(scale * pr * (a1[r] - D4[r]));
        }
      }
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        accV[dt] = mma_chunk<bf16>(trA(Or, 32 * qc, dt * 16), __builtin_bit_cast(u32x4, pb), accV[dt]);
        accK[dt] = mma_chunk<bf16>(trA(Qr, 32 * qc, dt * 16), __builtin_bit_cast(u32x4, sb), accK[dt]);
      }
    }
    if (kok) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        bf16* r = dbase + (size_t)key * rowlen + heads * HDP + dt * 16 + lq * 4;
        store4<bf16>(r, accK[dt]);
        store4<bf16>(r + heads * HDP, accV[dt]);
      }
    }
  }
}

template <int NKT, int NDT, int NW>
static int launch_bwd_mfma(const void* qkv, const void* dctx, void* dqkv, int B, int S, int heads, float scale, hipStream_t s, bool* ran) {
  constexpr int HDP = NDT * 16, SKP = NKT * 16;
  const size_t lds = (size_t)4 * SKP * (HDP * 2 + 32) + (size_t)3 * SKP * sizeof(float);
  *ran = false;
  if (lds > 160 * 1024) return 0;
  auto kern = attention_bwd_mfma_kernel<NKT, NDT, NW>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(B * heads), dim3(NW * 64), lds, s, (const bf16*)qkv, (const bf16*)dctx, (bf16*)dqkv, S, heads, scale);
  *ran = true;
  return (int)hipGetLastError();
}

int launch_attention_bwd(const void* qkv, const void* dctx, void* dqkv, int B, int S, int heads, int hd, int hdp, float scale, int dtype, hipStream_t s) {
  if (B <= 0) return 0;
  if (dtype == 1 && hdp % 32 == 0 && S <= 224) {      // bf16: MFMA kernel
    bool ran = false;
    int rc = 0;
    const int ndt = hdp / 16;
    if (S <= 32) {
      if (ndt == 2) rc = launch_bwd_mfma<2, 2, 2>(qkv, dctx, dqkv, B, S, heads, scale, s, &ran);
      else if (ndt == 4) rc = launch_bwd_mfma<2, 4, 2>(qkv, dctx, dqkv, B, S, heads, scale, s, &ran);
      else if (ndt == 6) rc = launch_bwd_mfma<2, 6, 2>(qkv, dctx, dqkv, B, S, heads, scale, s, &ran);
      else if (ndt == 8) rc = launch_bwd_mfma<2, 8, 2>(qkv, dctx, dqkv, B, S, heads, scale, s, &ran);
    } else if (S <= 128) {                              // (4 waves: with 8 the 512-thread bound left 256 registers per wave and the kernel spilled 228 ..
                                                        //  928 bytes per lane; one wave per SIMD takes accumulators in AGPRs: 311 -> 249 us at S = 100)
      if (ndt == 2) rc = launch_bwd_mfma<8, 2, 4>(qkv, dctx, dqkv, B, S, heads, scale, s, &ran);
      else if (ndt == 4) rc = launch_bwd_mfma<8, 4, 4>(qkv, dctx, dqkv, B, S, heads, scale, s, &ran);
      else if (ndt == 6) rc = launch_bwd_mfma<8, 6, 4>(qkv, dctx, dqkv, B, S, heads, scale, s, &ran);
    } else {                                            // ViT: 196 patches + cls
      if (ndt == 2) rc = launch_bwd_mfma<14, 2, 4>(qkv, dctx, dqkv, B, S, heads, scale, s, &ran);
      else if (ndt == 4) rc = launch_bwd_mfma<14, 4, 4>(qkv, dctx, dqkv, B, S, heads, scale, s, &ran);
    }
    if (rc || ran) return rc;
  }
  if (dtype == 0) {                                     // exact-fp32 MFMA kernel where the head fits the LDS (Visformer stages: 100 x 42, 25 x 85)
    constexpr bool off = false;
    if (!off && (hdp & 3) == 0) {
      if (S <= 32 && hdp <= 48) return launch_bwd_f32mfma<2, 3>(qkv, dctx, dqkv, B, S, heads, hd, hdp, scale, s);
      if (S <= 32 && hdp <= 96) return launch_bwd_f32mfma<2, 6>(qkv, dctx, dqkv, B, S, heads, hd, hdp, scale, s);
      if (S <= 48 && hdp <= 64) return launch_bwd_f32mfma<3, 4>(qkv, dctx, dqkv, B, S, heads, hd, hdp, scale, s);
      if (S <= 112 && hdp <= 48) return launch_bwd_f32mfma<7, 3>(qkv, dctx, dqkv, B, S, heads, hd, hdp, scale, s);
      if (S <= 208 && hdp <= 64) return launch_bwd_f32mfma_long<13, 4, 7>(qkv, dctx, dqkv, B, S, heads, hd, hdp, scale, s);      // ViT / DeiT: 196 patches + cls
    }
  }
  const size_t lds = ((size_t)4 * S * (hd + 1) + (size_t)2 * S * (S + 1)) * sizeof(float);
  hipError_t e;
  if (lds > 160 * 1024) {
    constexpr int QB = 16;
    const size_t lds_t = ((size_t)2 * S * (hd + 1) + (size_t)2 * QB * (hd + 1) + (size_t)2 * QB * (S + 1)) * sizeof(float);
    if (dtype != 0 || lds_t > 160 * 1024) return (int)hipErrorInvalidValue;
    e = hipFuncSetAttribute((const void*)attention_bwd_tiled_f32_kernel<QB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(attention_bwd_tiled_f32_kernel<QB>, dim3(B * heads), dim3(256), lds_t, s, (const float*)qkv, (const float*)dctx, (float*)dqkv, S, heads, hd, hdp, scale);
    return (int)hipGetLastError();
  }
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
