// Fused Mlp of the Visformer attention blocks (stage 2: C = 256, hidden = 1024), bf16:
//   y = x + conv3( GELU( conv1( BN(x) ) ) )     test_phase/models/visformer.py:146-150 (spatial_conv=False) + :262 (residual)
// with the eval BatchNorm folded into conv1 (column scale + bias, engine.hip build()).  Both convs are 1x1 = row-wise
// GEMMs, so the hidden activation (4C per token - 655 MB per 3200-image step and block) never has to leave the chip.
//
// Why the two gemm256 launches were slow: at K = 256 / N = 256 a 256x256 output tile has only four K tiles between
// epilogues, and the hidden tensor is written and re-read through HBM.  Here:
//   * a workgroup owns 256 token rows, a WAVE owns 32 of them for the whole Mlp; the wave's x rows live in registers as
//     MFMA B operands (64 VGPRs), the output accumulator (32 rows x 256 channels fp32) in 128 more;
//   * only weights move: the hidden dimension is walked in chunks of 32 units; chunk j needs W1[32j..32j+32][0..C) and
//     W2[0..C)[32j..32j+32) = 32 KB, which every wave reads from LDS as MFMA A operands (v_mfma_f32_32x32x16_bf16:
//     D^T[n][m] = sum_k W[n][k] X[m][k], 32 flop per LDS byte) - 256 flop per staged byte, twice gemm256's;
//   * GEMM1's accumulator IS GEMM2's B operand: a lane of the 32x32 result holds 16 hidden units of one token; after
//     bias + GELU they are packed to bf16 and fed straight back (the k order this implies is baked into the packed W2);
//   * every permutation is paid at PACK time (mlp_pack_kernel): the weight image in HBM is already the sequence of
//     1 KB MFMA fragments (lane-major, 16 bytes per lane) in the order the loop consumes them, so staging is a linear
//     LDS-DMA copy (no swizzle, no address arithmetic) and every ds_read_b128 is lane-linear = conflict-free; the channel
//     order of the x registers / output accumulators is chosen so that a lane holds 16 CONSECUTIVE channels of its token
//     per 32-channel tile: 16-byte loads and stores, and the residual comes from the very registers that fed GEMM1;
//   * the weight stream is periodic (1 MB per 256 rows, L2-resident), so one 4-stage LDS ring with counted vmcnt runs
//     across tiles of the persistent workgroup; the only workgroup-wide synchronisation is the ring's barrier (one per
//     32 MFMAs per wave); waves never exchange data.
// Layout contract with the pack kernel (lane = 32 * kh + r):
//   x regs   xr[s], s < C/16     : token r, channels 32 (s/2) + 16 kh + 8 (s&1) + 0..7
//   W1 frag  (chunk j, step s)   : hidden unit 32 j + r, the same 8 channels
//   hacc[i], i = 4 g + e         : hidden unit 32 j + 8 g + 4 kh + e of token r          (32x32 MFMA C/D layout)
//   hp[s2]                       : hacc[8 s2 .. 8 s2 + 8) packed to bf16
//   W2 frag  (j, ct, s2)         : output channel 32 ct + 16 (r>>2 & 1) + 4 (r>>3) + (r&3), hidden units as hp[s2] of lane kh
//   yacc[ct][i]                  : channel 32 ct + 16 kh + i of token r
#include <stdlib.h>

#include "fsvit_common.h"
#include "kernels.h"

namespace fsvit {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lptrm_t;

namespace {

constexpr int MR_NST = 4;                       // ring stages
#ifndef MR_FD1
#define MR_FD1 4
#endif
#ifndef MR_FD2
#define MR_FD2 4
#endif

// NP consecutive 1 KiB LDS-DMAs: source = sbase + voff + i * 1024, destination = lds + i * 1024.  The immediate offset of
// global_load_lds moves the LDS destination together with the global source (tools/probes/ldsdma_offset.hip, measured on
// gfx950), so a linear copy needs one M0 write and no address arithmetic.
template <int NP>
__device__ __forceinline__ void mr_dma(unsigned voff, const void* sbase, unsigned lds) {
  unsigned keep;
  static_assert(NP == 2 || NP == 4, "pieces per wave");
  if constexpr (NP == 4) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds)
        : "memory");
  } else {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds)
        : "memory");
  }
}
__device__ __forceinline__ void mr_bar() {
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int N> __device__ __forceinline__ void mr_wait_vm() {
  static_assert(N == 0 || N == 2 || N == 4 || N == 8, "add the literal");
  if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// x rows are loaded through inline asm: a compiler-visible global_load inside the tile loop makes hipcc's waitcnt pass carry
// "x load pending" around the back edge and plant s_waitcnt vmcnt(15..0) between the MFMAs of the chunk loop - which drains the
// LDS-DMA ring (issued from asm, invisible to that pass) at every step.
__device__ __forceinline__ u32x4 mr_gload16(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned mr_pk2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const bf16x2_t v = {(bf16)a, (bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

}  // namespace

// Measured and dropped: waves 4-7 - the SIMD partners of waves 0-3 - running GEMM1 one barrier interval early so that
// one partner is in its GELU while the other issues MFMAs (-3 %); read-ahead depths 6 / 8 instead of 4 (+-0 %: LDS latency is not
// the limiter); (s2 outer, ct inner) MFMA order in GEMM2 (+-0 %).
template <int C, int HID>
__device__ __forceinline__ void mlp_rows_body(const bf16* __restrict__ X, bf16* __restrict__ Y, const unsigned char* __restrict__ wimg,
                                              const float* __restrict__ b2, const int M, const int n_tiles, unsigned char* smem, const int wave, const int lane) {
  constexpr int NCT = C / 32, NKS = C / 16, NCH = HID / 32;
  constexpr int STAGE = (NKS + 2 * NCT) * 1024;          // W1 fragments then W2 fragments of one hidden chunk
  constexpr int NP = STAGE / 1024 / 8;                   // 1 KiB DMA pieces per wave and stage
  constexpr int FD = 4;                                  // weight fragments read ahead of the MFMA that consumes them (GELU span)
  constexpr int FD1 = MR_FD1, FD2 = MR_FD2;              // ... inside GEMM1 / GEMM2, where the other phase's registers are free
  const float* const b1tab = reinterpret_cast<const float*>(smem + MR_NST * STAGE);
  const int r = lane & 31, kh = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)(lptrm_t)smem;
  const unsigned voff = (unsigned)(wave * NP * 1024 + lane * 16);      // this lane's 16 bytes inside a stage image

  // ring: stage n (hidden chunk n % NCH; the image repeats every NCH stages) is issued right after barrier n-3, waited for
  // (counted vmcnt) before barrier n, which publishes it; its slot is refilled after barrier n+1, when every wave has
  // finished GEMM2(n).
  int issue_chunk = 0, issue_slot = 0;
  auto issue = [&]() {
    mr_dma<NP>(voff, wimg + (size_t)issue_chunk * STAGE, lds0 + issue_slot * STAGE + wave * NP * 1024);
    issue_chunk = issue_chunk == NCH - 1 ? 0 : issue_chunk + 1;
    issue_slot = issue_slot == MR_NST - 1 ? 0 : issue_slot + 1;
  };
#pragma unroll
  for (int i = 0; i < MR_NST - 1; ++i) issue();
  int slot = 0;

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    // ---- this wave's 32 token rows -> registers (tail rows re-read the last valid row; their results are never stored)
    const int m = tile * 256 + wave * 32 + r;
    const bool mok = m < M;
    const size_t rowoff = (size_t)(mok ? m : M - 1) * C + 16 * kh;
    u32x4 xr[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) xr[s] = mr_gload16(X + rowoff + 32 * (s >> 1) + 8 * (s & 1));
    // The x loads and the previous tile's stores share the vmcnt queue with the ring's DMAs: drain once per tile (every DMA older
    // than these loads landed long ago).  The registers are threaded through the wait so no use can be scheduled above it.
    static_assert(NKS == 16, "operand list of the wait below");
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]),
                   "+v"(xr[8]), "+v"(xr[9]), "+v"(xr[10]), "+v"(xr[11]), "+v"(xr[12]), "+v"(xr[13]), "+v"(xr[14]), "+v"(xr[15])
                 :: "memory");
    f32x16 yacc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int i = 0; i < 16; ++i) yacc[ct][i] = 0.0f;

    f32x16 hacc;
    u32x4 fr[FD2 > FD1 ? FD2 : FD1];
    // GEMM1 of hidden chunk j from ring slot sl: the 16 W1 fragments are read FD ahead of their MFMA through a rotating register set
    // (left alone hipcc reads every fragment into the same VGPRs right before its MFMA and eats the LDS latency 32 times per step)
    auto gemm1 = [&](int j, int sl) {
      const unsigned char* sp = smem + sl * STAGE + lane * 16;
#pragma unroll
      for (int i = 0; i < FD1; ++i) fr[i] = *reinterpret_cast<const u32x4*>(sp + i * 1024);
      const float* bp = b1tab + j * 32 + kh * 16;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bp + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) hacc[4 * g + e] = b[e];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        hacc = mfma32(fr[s % FD1], xr[s], hacc);
        if (s + FD1 < NKS) fr[s % FD1] = *reinterpret_cast<const u32x4*>(sp + (s + FD1) * 1024);
        __builtin_amdgcn_sched_barrier(0);               // pin: MFMA s, then the read FD ahead (the waitcnt pass then emits lgkmcnt(FD-1))
      }
    };
    // GELU + GEMM2 of the chunk in slot sl: GEMM2's first fragments are in flight during the GELU; the GELU'd accumulator, packed to
    // bf16, is GEMM2's B operand
    auto gelu_gemm2 = [&](int sl) {
      const unsigned char* sp = smem + sl * STAGE + NKS * 1024 + lane * 16;
#pragma unroll
      for (int i = 0; i < FD; ++i) fr[i] = *reinterpret_cast<const u32x4*>(sp + (2 * (i % NCT) + i / NCT) * 1024);
      __builtin_amdgcn_sched_barrier(0);
      u32x4 hp[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 4; ++e) hp[s2][e] = mr_pk2(gelu_sig(hacc[8 * s2 + 2 * e]), gelu_sig(hacc[8 * s2 + 2 * e + 1]));
      __builtin_amdgcn_sched_barrier(0);
      // fragment order (s2 outer, ct inner): consecutive MFMAs hit different accumulators (image order is [ct][s2])
      auto foff = [](int f) { return (2 * (f % NCT) + f / NCT) * 1024; };
#pragma unroll
      for (int f = 0; f < 2 * NCT; ++f) {
        // the first FD fragments were read before the GELU; top the read-ahead up to FD2 once the GELU's registers are free
        if (f == 0) {
#pragma unroll
          for (int i = FD; i < FD2; ++i) fr[i] = *reinterpret_cast<const u32x4*>(sp + foff(i));
          __builtin_amdgcn_sched_barrier(0);
        }
        yacc[f % NCT] = mfma32(fr[f % FD2], hp[f / NCT], yacc[f % NCT]);
        if (f + FD2 < 2 * NCT) fr[f % FD2] = *reinterpret_cast<const u32x4*>(sp + foff(f + FD2));
        __builtin_amdgcn_sched_barrier(0);
      }
    };

#pragma unroll 1
    for (int j = 0; j < NCH; ++j) {
      mr_wait_vm<(MR_NST - 2) * NP>();  // this wave's pieces of stage j have landed (the NST-2 newest stages may be in flight)
      mr_bar();
      issue();
      gemm1(j, slot);
      gelu_gemm2(slot);
      slot = slot == MR_NST - 1 ? 0 : slot + 1;
    }

    // ---- epilogue: + residual (the x registers), + optional bias of conv3, 2 x 16-byte stores per 32-channel tile
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const bf16x8 xv = __builtin_bit_cast(bf16x8, xr[2 * ct + q]);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = yacc[ct][8 * q + e] + (float)xv[e];
          if (b2) v += b2[32 * ct + 16 * kh + 8 * q + e];
          o[e] = (bf16)v;
        }
        if (mok) *reinterpret_cast<bf16x8*>(Y + rowoff + 32 * ct + 8 * q) = o;
      }
    }
  }
  mr_wait_vm<0>();     // no DMA may be in flight into the LDS of a finished workgroup
}

template <int C, int HID>
__global__ __launch_bounds__(512, 2) void mlp_rows_kernel(const bf16* __restrict__ X, bf16* __restrict__ Y, const unsigned char* __restrict__ wimg,
                                                          const float* __restrict__ b1img, const float* __restrict__ b2, const int M, const int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = (C / 16 + 2 * (C / 32)) * 1024;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  if ((int)blockIdx.x >= n_tiles) return;
  // bias table of conv1 (already in accumulator order) -> LDS
  float* const b1tab = reinterpret_cast<float*>(smem + MR_NST * STAGE);
  for (int i = t; i < HID; i += 512) b1tab[i] = b1img[i];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        // table written before the first barrier publishes it
  mlp_rows_body<C, HID>(X, Y, wimg, b2, M, n_tiles, smem, wave, lane);
}

// Builds the fragment-major weight image + bias table from the engine's standard packed layers (w1 [HID][k1w], w2 [C][k2w],
// K-contiguous bf16 rows).  One thread per bf16 element of the image.
__global__ void mlp_pack_kernel(const bf16* __restrict__ w1, int k1w, const float* __restrict__ b1, const bf16* __restrict__ w2, int k2w,
                                bf16* __restrict__ wimg, float* __restrict__ b1img, int C, int HID) {
  const int NCT = C / 32, NKS = C / 16, NCH = HID / 32;
  const int per_chunk = (NKS + 2 * NCT) * 512;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < HID) {
    const int j = (int)idx >> 5, w = (int)idx & 31, kh = w >> 4, i = w & 15;
    b1img[idx] = b1 ? b1[j * 32 + 8 * (i >> 2) + 4 * kh + (i & 3)] : 0.0f;
  }
  if (idx >= (long)NCH * per_chunk) return;
  const int j = (int)(idx / per_chunk), e = (int)(idx % per_chunk);
  const int piece = e >> 9, lane = (e >> 3) & 63, e8 = e & 7;
  const int kh = lane >> 5, r = lane & 31;
  bf16 v;
  if (piece < NKS) {
    const int s = piece;
    v = w1[(size_t)(j * 32 + r) * k1w + 32 * (s >> 1) + 16 * kh + 8 * (s & 1) + e8];
  } else {
    const int q = piece - NKS, ct = q >> 1, s2 = q & 1;
    const int c = 32 * ct + 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
    const int n = j * 32 + 8 * (2 * s2 + (e8 >> 2)) + 4 * kh + (e8 & 3);
    v = w2[(size_t)c * k2w + n];
  }
  wimg[idx] = v;
}

bool mlp_rows_supported(int dtype, int C, int hid) {
  static const bool off = [] { const char* e = getenv("FSVIT_MLP_ROWS"); return e && e[0] == '0'; }();
  return !off && dtype == 1 && C == 256 && hid == 1024;
}
size_t mlp_rows_image_bytes(int C, int hid) { return (size_t)(hid / 32) * (C / 16 + 2 * (C / 32)) * 1024; }

int launch_mlp_pack(const void* w1, int k1w, const float* b1, const void* w2, int k2w, void* wimg, float* b1img, int C, int hid, hipStream_t s) {
  const long n = (long)mlp_rows_image_bytes(C, hid) / 2;
  hipLaunchKernelGGL(mlp_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16*)w1, k1w, b1, (const bf16*)w2, k2w,
                     (bf16*)wimg, b1img, C, hid);
  return (int)hipGetLastError();
}

int launch_mlp_rows(const void* x, void* y, const void* wimg, const float* b1img, const float* b2, int M, int C, int hid, hipStream_t s) {
  if (C != 256 || hid != 1024) return (int)hipErrorInvalidValue;
  if (M <= 0) return 0;
  auto kern = mlp_rows_kernel<256, 1024>;
  const int lds = MR_NST * (256 / 16 + 2 * (256 / 32)) * 1024 + 1024 * 4;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const int n_tiles = (M + 255) / 256;
  const int grid = n_tiles < 256 ? n_tiles : 256;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, (const bf16*)x, (bf16*)y, (const unsigned char*)wimg, b1img, b2, M, n_tiles);
  return (int)hipGetLastError();
}

}  // namespace fsvit
