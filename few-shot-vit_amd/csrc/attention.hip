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

namespace FSVIT_NS {

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


// ------------------------------------------------------------------------------------------------
// v2: register-resident softmax.  One workgroup per (image, head), NW waves; K (row-major) and V^T
// are staged once in LDS, each wave then owns whole 16-query tiles with no further barrier:
//   S^T tile = K . Q^T on MFMA (operands swapped) puts all keys of query (lane&15) into the 4 lanes
//   {lane&15 + 16*j}: the row max / sum are an in-lane reduction plus two cross-lane steps;
//   P stays in registers: the C-layout of S^T (4 consecutive keys per lane) is fed straight back as
//   the MFMA B operand of ctx^T = V^T . P^T -- the k-permutation it implies is applied to the V^T
//   fragment read instead (two 8-byte LDS reads per 32-key chunk for bf16; the natural 16-byte read
//   for fp32), so no LDS round trip or lane shuffle is needed for P.
// NKT = key tiles of 16 (keys padded to the 64-byte chunk, masked), NDT = head-dim tiles of 16.
template <typename T, int NKT, int NDT, int NW>
__global__ __launch_bounds__(NW * 64) void attention_v2_kernel(const T* __restrict__ qkv, T* __restrict__ ctx,
                                                               int S, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ES = sizeof(T);
  constexpr bool LIMBS = sizeof(T) == 4 && !__is_same(T, float);      // T = f32x2l (the two-limb modes): fp32 rows in HBM, K / V^T held in LDS as [hi | lo]
                                                                      // limb words (split once at staging), Q and P split once in registers, every product
                                                                      // as two 16-bit MFMAs (all four limb products) instead of four exact-fp32 ones
  constexpr int EPC = Elem<T>::kPerChunk;
  constexpr int HDP = NDT * 16;
  constexpr int SKP = NKT * 16;
  constexpr int QS = HDP * ES + ((32 - (HDP * ES) % 64 + 64) % 64);      // K row stride (bytes), 32 mod 64: the conflict-free strides of ds_read_b128's four
                                                                         // mixed 16-lane groups (an odd multiple of 16, as before, reads 2-way; qkv_attn.hip)
  constexpr int VS = SKP * ES + 16;            // V^T row stride
  constexpr int NKC = (HDP * ES + 63) / 64;    // 64-byte chunks along the head dim; a head dim of 48 bf16 (1.5 chunks) zero-fills the half chunk in registers
  constexpr bool HALF_TAIL = (HDP * ES) % 64 != 0;
  unsigned char* const Ks = smem;
  unsigned char* const Vt = smem + SKP * QS;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lrow = lane & 15, lq = lane >> 4;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int rowlen = 3 * heads * HDP;
  const T* base = qkv + (size_t)b * S * rowlen + h * HDP;
  constexpr int CPR = HDP / EPC;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // Every global load of the workgroup is issued up front and unconditionally (clamped rows, zeroed afterwards): the K / V chunks of all staging
  // passes and the Q fragments of this wave's query tiles.  The first version loaded, waited and stored pass by pass and fetched Q at the top of
  // each query tile - four to six dependent trips to memory per workgroup, ~8 us for 2 us of MFMAs at 197 tokens.
  constexpr int NIT = (SKP * CPR + NW * 64 - 1) / (NW * 64);      // staging passes
  constexpr int NQW = (NKT + NW - 1) / NW;                        // query tiles per wave (S <= 16 NKT)
  const int nqt = (S + 15) / 16;
  u32x4 kreg[NIT], vreg[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = t + it * NW * 64;
    const int row = idx / CPR, ch = idx - row * CPR;
    const T* src = base + (size_t)(row < S ? row : 0) * rowlen + heads * HDP + ch * EPC;
    kreg[it] = *reinterpret_cast<const u32x4*>(src);
    vreg[it] = *reinterpret_cast<const u32x4*>(src + heads * HDP);
  }
  u32x4 qpre[NQW][NKC];
#pragma unroll
  for (int i = 0; i < NQW; ++i) {
    const int q = (wave + i * NW) * 16 + lrow;
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      const bool inrow = !HALF_TAIL || kc + 1 < NKC || lq < 2;                   // 16-byte chunk kc * 4 + lq lies inside the head row
      qpre[i][kc] = *reinterpret_cast<const u32x4*>(base + (size_t)(q < S ? q : 0) * rowlen + (inrow ? (kc * 4 + lq) * EPC : 0));
      if (!(q < S && inrow)) qpre[i][kc] = zero4;
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = t + it * NW * 64;
    const int row = idx / CPR, ch = idx - row * CPR;
    if (idx >= SKP * CPR) break;
    u32x4 k = row < S ? kreg[it] : zero4, v = row < S ? vreg[it] : zero4;
    if constexpr (LIMBS) {
      u32x4 r_;
      x2_split(k, k, r_);
      x2_split(v, v, r_);
    }
    *reinterpret_cast<u32x4*>(Ks + row * QS + ch * 16) = k;
    if constexpr (ES == 2) {
      // 16-bit storage: V stays ROW-major, as [16-column subtile][key][32 B]; the PV step reads its A fragments (8 keys of one head-dim column
      // per lane) with the transposing ds_read_b64_tr_b16.  (Round 1 wrote V^T with eight 2-byte stores per chunk: 79 % of the kernel's LDS
      // cycles were bank conflicts.)
      *reinterpret_cast<u32x4*>(Vt + (ch >> 1) * (SKP * 32) + row * 32 + (ch & 1) * 16) = v;
    } else {
      const T* ve = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int e = 0; e < EPC; ++e) *reinterpret_cast<T*>(Vt + (ch * EPC + e) * VS + row * ES) = ve[e];
    }
  }
  __syncthreads();

  T* obase = ctx + (size_t)b * S * heads * HDP + h * HDP;
#pragma unroll
  for (int qi = 0; qi < NQW; ++qi) {
    const int qt = wave + qi * NW;
    if (qt >= nqt) break;
    const int q = qt * 16 + lrow;
    // Q fragments (row q, 16-byte chunk kc*4 + lq): fetched with the K / V loads above
    u32x4 qf[NKC], qr[LIMBS ? NKC : 1];
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      qf[kc] = qpre[qi][kc];
      if constexpr (LIMBS) x2_split(qf[kc], qf[kc], qr[kc]);
    }
    f32x4 sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const unsigned char* ka = Ks + (kt * 16 + lrow) * QS + lq * 16;
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc) {
        u32x4 kf = *reinterpret_cast<const u32x4*>(ka + kc * 64);
        if (HALF_TAIL && kc + 1 == NKC && lq >= 2) kf = zero4;                   // beyond the row: the LDS bytes there belong to the pad / the next row
        if constexpr (LIMBS) {
          acc = mma_chunk<bf16>(kf, qr[kc], acc);          // lo x hi + hi x lo
          acc = mma_chunk<bf16>(kf, qf[kc], acc);          // hi x hi + lo x lo
        } else
        acc = mma_chunk<T>(kf, qf[kc], acc);
      }
      sc[kt] = acc;                      // keys kt*16 + lq*4 + r  x  query lrow
    }
    // bf16 storage: P is rounded to bf16 anyway, so exp runs on the hardware exp2 (v_exp_f32) with log2(e) folded
    // into the score scale; the fp32 parity path keeps expf.
    constexpr bool FAST_EXP = ES == 2;
    const float sscale = FAST_EXP ? scale * 1.44269504088896340736f : scale;
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
        const float d = sc[kt][r] - m;                 // -inf for masked keys -> exp = 0
        const float e = FAST_EXP ? __builtin_amdgcn_exp2f(d) : expf(d);
        sc[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    // ctx^T[d][q] = sum_key V^T[d][key] * P[q][key]
    u32x4 ps[LIMBS ? NKT : 1], pr[LIMBS ? NKT : 1];          // two-limb modes: the (unnormalised) probabilities as limb words, split once per query tile
    if constexpr (LIMBS) {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) x2_split(__builtin_bit_cast(u32x4, sc[kt]), ps[kt], pr[kt]);
    }
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const unsigned char* va = Vt + (dt * 16 + lrow) * VS;
      if constexpr (LIMBS) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          const u32x4 vf = *reinterpret_cast<const u32x4*>(va + kt * 64 + lq * 16);
          acc = mma_chunk<bf16>(vf, pr[kt], acc);
          acc = mma_chunk<bf16>(vf, ps[kt], acc);
        }
      } else if constexpr (ES == 4) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          const u32x4 vf = *reinterpret_cast<const u32x4*>(va + kt * 64 + lq * 16);
          acc = mma_chunk<T>(vf, __builtin_bit_cast(u32x4, sc[kt]), acc);
        }
      } else {
        // lane (i = lrow, lq) supplies the address of columns 4 (i & 3) .. of key row 32 kc + 4 lq + (i >> 2) (+ 16) and receives keys
        // 32 kc + 4 lq + 0..3 (+ 16) of column i: the K-slot order the packed P fragment below has (tools/probes/tr_read_probe.hip)
        const unsigned vaddr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Vt + dt * (SKP * 32) + (lq * 4 + (lrow >> 2)) * 32 + (lrow & 3) * 8;
        typedef short s4v __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int kc = 0; kc < NKT / 2; ++kc) {
          const u32x2 v0 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(size_t)(vaddr + kc * 1024)));
          const u32x2 v1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(size_t)(vaddr + kc * 1024 + 512)));
          const u32x4 vf = {v0[0], v0[1], v1[0], v1[1]};
          const bf16x8 pb = {(bf16)sc[2 * kc][0], (bf16)sc[2 * kc][1], (bf16)sc[2 * kc][2], (bf16)sc[2 * kc][3],
                             (bf16)sc[2 * kc + 1][0], (bf16)sc[2 * kc + 1][1], (bf16)sc[2 * kc + 1][2], (bf16)sc[2 * kc + 1][3]};
          acc = mma_chunk<T>(vf, __builtin_bit_cast(u32x4, pb), acc);
        }
      }
      if (q < S) store4<T>(obase + (size_t)q * heads * HDP + dt * 16 + lq * 4, acc * inv);
    }
  }
}

template <typename T, int NKT, int NDT, int NW>
static int launch_v2(const void* qkv, void* ctx, int B, int S, int heads, float scale, hipStream_t s) {
  constexpr int ES = sizeof(T);
  constexpr int QSB = NDT * 16 * ES + ((32 - (NDT * 16 * ES) % 64 + 64) % 64);      // the kernel's K row stride
  const size_t lds = (size_t)NKT * 16 * QSB + (size_t)NDT * 16 * (NKT * 16 * ES + 16);
  auto kern = attention_v2_kernel<T, NKT, NDT, NW>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3(B * heads), dim3(NW * 64), lds, s, (const T*)qkv, (T*)ctx, S, heads, scale);
  return (int)hipGetLastError();
}

// returns -1 when no v2 instantiation covers the shape (caller falls back to the generic v1 kernel)
template <typename T, int NKT, int NW>
static int dispatch_ndt(int ndt, const void* qkv, void* ctx, int B, int S, int heads, float scale, hipStream_t s) {
  switch (ndt) {
    case 1: if constexpr (sizeof(T) == 4) return launch_v2<T, NKT, 1, NW>(qkv, ctx, B, S, heads, scale, s); else return -1;
    case 2: return launch_v2<T, NKT, 2, NW>(qkv, ctx, B, S, heads, scale, s);
    case 3: return launch_v2<T, NKT, 3, NW>(qkv, ctx, B, S, heads, scale, s);          // bf16: head dim 48 = one and a half 64-byte chunks
    case 4: return launch_v2<T, NKT, 4, NW>(qkv, ctx, B, S, heads, scale, s);
    case 5: if constexpr (sizeof(T) == 4) return launch_v2<T, NKT, 5, NW>(qkv, ctx, B, S, heads, scale, s); else return -1;
    case 6: return launch_v2<T, NKT, 6, NW>(qkv, ctx, B, S, heads, scale, s);
    case 8: return launch_v2<T, NKT, 8, NW>(qkv, ctx, B, S, heads, scale, s);
    default: return -1;
  }
}

static int launch_attention_v2(const void* qkv, void* ctx, int B, int S, int heads, int hdp, float scale, int dtype, hipStream_t s) {
  const int ndt = hdp / 16;
  if (dtype == 2) {                        // two-limb modes (fp32 storage): the float kernel's shapes on the 16-bit MFMA
    if (S <= 32) return dispatch_ndt<f32x2l, 2, 2>(ndt, qkv, ctx, B, S, heads, scale, s);
    if (S <= 112) return dispatch_ndt<f32x2l, 7, 4>(ndt, qkv, ctx, B, S, heads, scale, s);
    if (S <= 208) return dispatch_ndt<f32x2l, 13, 7>(ndt, qkv, ctx, B, S, heads, scale, s);
    return -1;
  }
  if (dtype == 0) {
    if (S <= 32) return dispatch_ndt<float, 2, 2>(ndt, qkv, ctx, B, S, heads, scale, s);
#ifndef ATT_NWF112
#define ATT_NWF112 4
#endif
#ifndef ATT_NWF208
#define ATT_NWF208 7      // 13 query tiles: 4 waves 153 ms per DeiT-S step (two-limb mode), 7: 138, 8: 141; at 7 tiles (S <= 112) the wave count does not matter
#endif
    if (S <= 112) return dispatch_ndt<float, 7, ATT_NWF112>(ndt, qkv, ctx, B, S, heads, scale, s);
    if (S <= 208) return dispatch_ndt<float, 13, ATT_NWF208>(ndt, qkv, ctx, B, S, heads, scale, s);   // ViT: 196 patches + cls
  } else {
    if (S <= 32) return dispatch_ndt<bf16, 2, 2>(ndt, qkv, ctx, B, S, heads, scale, s);
    if (S <= 128) return dispatch_ndt<bf16, 8, 4>(ndt, qkv, ctx, B, S, heads, scale, s);
#ifndef ATT_NW224
#define ATT_NW224 8      // 13 query tiles of a 197-token map over 8 waves (4: 27.9 ms per DeiT-S step, 7: 26.1, 8: 24.8)
#endif
    if (S <= 224) return dispatch_ndt<bf16, 14, ATT_NW224>(ndt, qkv, ctx, B, S, heads, scale, s);
  }
  return -1;
}

// Padded head dim the packer should use: the smallest multiple of 16 the attention kernels take for (hd, S, dtype).  fp32: any
// multiple of 16.  bf16: 32 / 48 / 64 / 96 / 128 on the register-softmax kernel (48 = one and a half 64-byte MFMA chunks, the half
// chunk zero-filled in registers: Visformer-S stage 2 pads 42 -> 48 instead of 64, 25 % fewer qkv / ctx bytes and qkv / proj flops);
// otherwise multiples of 32.
int attention_padded_head_dim(int hd, int S, int dtype) {
  const int h16 = (hd + 15) / 16 * 16;
  if (dtype == 0) return h16;
  const int ndt = h16 / 16;
  if (S <= 224 && (ndt == 2 || ndt == 3 || ndt == 4 || ndt == 6 || ndt == 8)) return h16;
  return (hd + 31) / 32 * 32;
}

int launch_attention(const void* qkv, void* ctx, int B, int S, int heads, int hdp, float scale, int dtype, hipStream_t s) {
  if (B <= 0) return 0;
  if (dtype == 2) {                        // no two-limb instantiation for this shape: the exact-fp32 kernels take it
    const int rc = launch_attention_v2(qkv, ctx, B, S, heads, hdp, scale, 2, s);
    if (rc != -1) return rc;
    dtype = 0;
  }

  if (S < 1 || hdp % 16 != 0) return (int)hipErrorInvalidValue;
  {
    const int rc = launch_attention_v2(qkv, ctx, B, S, heads, hdp, scale, dtype, s);
    if (rc != -1) return rc;
  }
  if (S > 128) return (int)hipErrorInvalidValue;      // the generic v1 kernel holds at most 128 tokens
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

}  // namespace FSVIT_NS
