// Fused Mlp of the Visformer attention blocks (stage 2: C = 256, hidden = 1024; stage 3: C = 512, hidden = 2048), bf16:
//   y = x + conv3( GELU( conv1( BN(x) ) ) )     test_phase/models/visformer.py:146-150 (spatial_conv=False) + :262 (residual)
// with the eval BatchNorm folded into conv1 (column scale + bias, engine.hip build()).  Both convs are 1x1 = row-wise
// GEMMs, so the hidden activation (4C per token - 655 MB per 3200-image step and block) never has to leave the chip.
//
// Why the two gemm256 launches were slow: at K = 256 / 512 a 256x256 output tile has only 4 / 8 K tiles between epilogues whose
// GELU + store tail costs as much as the main loop, and the hidden tensor is written and re-read through HBM.  Here:
//   * a WAVE owns 32 RB token rows for the whole Mlp (RB = 2 at C = 256, 1 at C = 512); its x rows live in registers as MFMA
//     B operands (128 VGPRs), the output accumulator (32 RB rows x C channels fp32) in 256 more: one wave per SIMD with the
//     whole 512-entry register file (4 waves per workgroup, one workgroup per CU);
//   * only weights move: the hidden dimension is walked in chunks of 32 units; chunk j needs W1[32j..32j+32][0..C) and
//     W2[0..C)[32j..32j+32) = C/8 KB, which every wave reads from LDS as MFMA A operands (v_mfma_f32_32x32x16_bf16:
//     D^T[n][m] = sum_k W[n][k] X[m][k]); at RB = 2 a fragment feeds two MFMAs.  128 RB flop per staged byte, 32 RB per LDS byte;
//   * GEMM1's accumulator IS GEMM2's B operand: a lane of the 32x32 result holds 16 hidden units of one token; after
//     bias + GELU they are packed to bf16 and fed straight back (the k order this implies is baked into the packed W2);
//   * every permutation is paid at PACK time (mlp_pack_kernel): the weight image in HBM is already the sequence of
//     1 KB MFMA fragments (lane-major, 16 bytes per lane) in the order the loop consumes them, so staging is a linear
//     LDS-DMA copy (no swizzle, no address arithmetic) and every ds_read_b128 is lane-linear = conflict-free; the channel
//     order of the x registers / output accumulators is chosen so that a lane holds 16 CONSECUTIVE channels of its token
//     per 32-channel tile: 16-byte loads and stores, and the residual comes from the very registers that fed GEMM1;
//   * the weight stream is periodic (1 / 4 MB per tile), so one ring of four 32 KB LDS slots (32 fragments each: a whole chunk
//     at C = 256, its W1 / W2 half at C = 512) with counted vmcnt runs across the tiles of the persistent workgroup; the
//     only workgroup-wide synchronisation is the ring's barrier (one per slot); waves never exchange data;
//   * the chunk loop is software-pipelined INSIDE the wave (one wave per SIMD issues in order): body j = GEMM2(j-1)'s MFMAs, then
//     GEMM1(j+1)'s, with the scalar GELU of chunk j cut into micro-stages that issue behind those MFMAs, the ring refill going out
//     one LDS-DMA piece at a time between them, and the weight image packed in that consumption order (see the loop's comment).
//     The image carries W1 / 8, b1 / 8 and 8 W2 (exact in bf16): min(h^2, 64) / 64 becomes one multiply with the clamp modifier.
// Layout contract with the pack kernel (lane = 32 * kh + r):
//   x regs   xr[s], s < C/16     : token r, channels 32 (s/2) + 16 kh + 8 (s&1) + 0..7
//   W1 frag  (chunk j, step s)   : hidden unit 32 j + r, the same 8 channels
//   hacc[i], i = 4 g + e         : hidden unit 32 j + 8 g + 4 kh + e of token r, at 1/8 scale   (32x32 MFMA C/D layout)
//   hp[s2]                       : GELU(hacc[8 s2 .. 8 s2 + 8)) / 8 packed to bf16
//   W2 frag  (j, ct, s2)         : output channel 32 ct + 16 (r>>2 & 1) + 4 (r>>3) + (r&3), hidden units as hp[s2] of lane kh
//   yacc[ct][i]                  : channel 32 ct + 16 kh + i of token r
// History: the first version ran 8 waves x 32 rows (two waves per SIMD, 256 registers each) at C = 256 only:
// 757 TFLOP/s; PMC showed MFMA 34 % and VALU 32 % busy with no overlap, and neither a phase shift between SIMD partners, deeper
// fragment read-ahead nor MFMA reordering moved it.  The second ran GEMM1 -> GELU -> GEMM2 back to back per chunk (846 / 665 us per
// launch at the bench shapes of stage 2 / 3); the pipelined loop measures 793 / 621 us.
// Diagnostics (never in the shipped build; tools/build_variant.sh): -DMR_CLK prints per-workgroup time and achieved shader clock,
// -DMR_DIAG=<bits> drops one cost at a time for timing (1 GELU, 2 ring barrier, 4 refill, 8 in-loop fragment reads) - wrong results.
#include <stdlib.h>

#include <type_traits>

#include "fsvit_common.h"
#include "kernels.h"

namespace FSVIT_NS {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lptrm_t;

namespace {

constexpr int MR_NST = 4;                       // ring slots
// fragments of 1 KiB per ring slot: 32 (a whole chunk at C = 256, its W1 / W2 half at C = 512) or 24 (C = 384: W1 / W2 half of a chunk)
constexpr int mr_slot_frags(int C) { return C == 384 ? 24 : 32; }
constexpr int MR_NW = 4;                        // waves per workgroup (one per SIMD)
constexpr int MR_FD = 6;                        // weight fragments read ahead of the MFMAs that consume them

// This wave's share of one ring slot: PW = 8 (6) consecutive 1 KiB LDS-DMAs, source = sbase + voff + i * 1024, destination = lds + i * 1024.
// The immediate offset of global_load_lds moves the LDS destination together with the global source (tools/probes/
// ldsdma_offset.hip, measured on gfx950), so a linear copy needs no address arithmetic; the 13-bit offset field covers 4 pieces.
template <int PW> __device__ __forceinline__ void mr_dma(unsigned voff, const void* sbase, unsigned lds) {
  static_assert(PW == 8 || PW == 6 || PW == 4 || PW == 3 || PW == 2, "pieces per wave and slot");
  unsigned keep;
  if constexpr (PW == 2)
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
  else if constexpr (PW == 4)
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
  else if constexpr (PW == 3)
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds)
        : "memory");
  else if constexpr (PW == 8)
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:2048\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:3072\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %3\n\t"
        "global_load_lds_dwordx4 %2, %3 offset:1024\n\t"
        "global_load_lds_dwordx4 %2, %3 offset:2048\n\t"
        "global_load_lds_dwordx4 %2, %3 offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "v"(voff + 4096u), "s"(sbase), "s"(lds), "s"(lds + 4096u)
        : "memory");
  else
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:2048\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:3072\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %3\n\t"
        "global_load_lds_dwordx4 %2, %3 offset:1024\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "v"(voff + 4096u), "s"(sbase), "s"(lds), "s"(lds + 4096u)
        : "memory");
}
// One 1 KiB piece.  A VMEM instruction of 64 x 16 bytes keeps the wave's issue stage for ~64 cycles; back to back they queue behind
// each other and hold up the MFMAs that follow, so inside the chunk loop the refill goes out one piece at a time, pieces >= 4 MFMAs apart.
__device__ __forceinline__ void mr_dma1(unsigned voff, const void* sbase, unsigned lds) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds)
      : "memory");
}
__device__ __forceinline__ void mr_bar() {
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// x rows are loaded through inline asm: a compiler-visible global_load inside the tile loop makes hipcc's waitcnt pass carry
// "x load pending" around the back edge and plant s_waitcnt vmcnt(15..0) between the MFMAs of the chunk loop - which drains the
// LDS-DMA ring (issued from asm, invisible to that pass) at every step.
__device__ __forceinline__ u32x4 mr_gload16(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// s_waitcnt vmcnt(0) with N (a multiple of 4) asm-loaded registers threaded through, so that no use can be scheduled above the wait
template <int N> __device__ __forceinline__ void mr_wait_loads(u32x4* a) {
  static_assert(N % 4 == 0 && N >= 4, "groups of 4");
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) :: "memory");
#pragma unroll
  for (int o = 4; o < N; o += 4) asm volatile("" : "+v"(a[o]), "+v"(a[o + 1]), "+v"(a[o + 2]), "+v"(a[o + 3]) :: "memory");
}
__device__ __forceinline__ unsigned mr_pk2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) bf16 bf16x2_t;
  const bf16x2_t v = {(bf16)a, (bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

// MFMAs are issued from inline asm with the register FILE of the accumulator spelled out: the 256 output accumulators must
// live in AGPRs and everything else (x rows, GEMM1 accumulators, fragments) in arch VGPRs.  Left to hipcc's allocator the
// kernel kept the outputs in VGPRs, spilled the x rows to scratch and reloaded one before every MFMA behind s_waitcnt vmcnt(0).
// What the compiler no longer does for these instructions and the kernel does by hand: wait states between the last MFMA that
// writes an accumulator and the first VALU / v_accvgpr_read that reads it (s_nop blocks below), and between the VALU that packs hp
// and the first MFMA that reads it.  Dependent MFMAs on one accumulator issue back to back (same opcode, same vDst).
__device__ __forceinline__ void mfma32_v(u32x4 a, u32x4 b, f32x16& c) {
  asm volatile(FSVIT_MFMA_32x32x16 " %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma32_a(u32x4 a, u32x4 b, f32x16& c) {
  asm volatile(FSVIT_MFMA_32x32x16 " %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// first MFMA of an accumulation chain that starts from zero: D = A B + 0
__device__ __forceinline__ void mfma32_v_z(u32x4 a, u32x4 b, f32x16& d) {
  asm volatile(FSVIT_MFMA_32x32x16 " %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma32_a_z(u32x4 a, u32x4 b, f32x16& d) {
  asm volatile(FSVIT_MFMA_32x32x16 " %0, %1, %2, 0" : "=a"(d) : "v"(a), "v"(b));
}
// accumulator := 0 without ever being a VGPR value (a C++ `= 0` makes the loop-carried accumulators VGPR-class and every asm use
// a 16-register round trip through v_accvgpr_write / read)
__device__ __forceinline__ void mfma32_a_zero(f32x16& c) {
  const u32x4 z = {0u, 0u, 0u, 0u};
  asm volatile("s_nop 7\n\t" FSVIT_MFMA_32x32x16 " %0, %1, %1, 0" : "=a"(c) : "v"(z));      // s_nop: VALU-written z -> MFMA SrcA/B
}

// cache policy of the rows GEMM's streaming accesses (tools/build_variant.sh sweeps): 0 default, 1 nt, 2 sc0 sc1 nt, 3 sc1
#define LGR_ST_MOD ""
#define LGR_LD_MOD ""
__device__ __forceinline__ void mr_gstore16(void* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off" LGR_ST_MOD :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 mr_gload16s(const void* p) {      // streaming row load of the rows GEMM
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" LGR_LD_MOD : "=v"(v) : "v"(p) : "memory");
  return v;
}

// max / sum of a value with its partner in the other wave half (lane ^ 32): one v_permlane32_swap_b32 instead of a trip through the LDS crossbar.
// Issued from inline asm on two copies: hipcc 7.2 treats both results of __builtin_amdgcn_permlane32_swap as interchangeable and folded
// max(r0, r1) -> r0 and r0 + r1 -> 2 r0 (v_fmac 2.0 in the ISA), although the instruction returns [lo | lo] and [hi | hi]
// (tools/probes/permlane32_swap_probe.hip).  s_nop 1: VALU write -> permlane read.
__device__ __forceinline__ void xhalf_swap(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xhalf_max(float v) {
  float a = v, b = v;
  xhalf_swap(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float xhalf_sum(float v) {
  float a = v, b = v;
  xhalf_swap(a, b);
  return a + b;
}

// LayerNorm (no affine) of token rows held as MFMA B operands: lane (r, kh) has half of token r's C channels in NKS 16-byte registers, the
// other half sits in lane r of the other wave half.  Two passes (mean, then centred sum of squares; biased variance as nn.LayerNorm), one lane
// exchange each; the rows are replaced by bf16((x - mean) rstd).  Ends with the wait states VALU write -> MFMA SrcB.
template <int NKS, int C> __device__ __forceinline__ void mr_layernorm_rows(u32x4 (&xr)[NKS], const float eps) {
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    const bf16x8 v = __builtin_bit_cast(bf16x8, xr[s]);
#pragma unroll
    for (int e = 0; e < 8; ++e) s4[e & 3] += (float)v[e];
  }
  float sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  sum += __shfl_xor(sum, 32);
  const float mean = sum * (1.0f / C);
  // opaque between the passes: hipcc otherwise keeps all C / 2 converted values of a pass alive for the next one (247 VGPRs, AGPR copies)
#pragma unroll
  for (int o = 0; o < NKS; o += 4) asm volatile("" : "+v"(xr[o]), "+v"(xr[o + 1]), "+v"(xr[o + 2]), "+v"(xr[o + 3]));
  float q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    const bf16x8 v = __builtin_bit_cast(bf16x8, xr[s]);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = (float)v[e] - mean; q4[e & 3] = fmaf(d, d, q4[e & 3]); }
  }
  float sq = (q4[0] + q4[1]) + (q4[2] + q4[3]);
  sq += __shfl_xor(sq, 32);
  const float rstd = rsqrtf(sq * (1.0f / C) + eps);
  const float nmr = -mean * rstd;
#pragma unroll
  for (int o = 0; o < NKS; o += 4) asm volatile("" : "+v"(xr[o]), "+v"(xr[o + 1]), "+v"(xr[o + 2]), "+v"(xr[o + 3]));
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    const bf16x8 v = __builtin_bit_cast(bf16x8, xr[s]);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = mr_pk2(fmaf((float)v[2 * e], rstd, nmr), fmaf((float)v[2 * e + 1], rstd, nmr));
    xr[s] = o;
  }
  static_assert(NKS % 4 == 0, "groups of 4");
#pragma unroll
  for (int o = 0; o + 4 < NKS; o += 4) asm volatile("" : "+v"(xr[o]), "+v"(xr[o + 1]), "+v"(xr[o + 2]), "+v"(xr[o + 3]) :: "memory");
  asm volatile("s_nop 7" : "+v"(xr[NKS - 4]), "+v"(xr[NKS - 3]), "+v"(xr[NKS - 2]), "+v"(xr[NKS - 1]) :: "memory");
}

}  // namespace

// KC > 0: the attention block's proj conv + residual (visformer.py:176,:261) runs as a prologue on the same rows:
//   x1 = x + Wp ctx  (ctx = attention output [M][KC], head dims zero-padded; Wp's fragments = the first KC/16 * C/32 fragments of the image)
// accumulated in the output AGPRs, rounded to bf16 into the x registers (exactly what the separate proj launch stored), and the Mlp
// continues from there: proj is HBM-bound as a GEMM of its own (K = 384 / 576, N = C, + residual read and write).
//
// LN (the ViT / DeiT block, deit.py:69-72: x = x + proj(attn); x = x + mlp(norm2(x)), C = 384 / hidden = 1536 / KC = 384): the proj has a
// bias (bp), and a LayerNorm sits between the residual stream and fc1.  It is row-local, and a token's row lives in the two lanes
// (r, kh = 0 / 1) of the x registers: x1 = bf16(x + Wp ctx + bp) is moved into the output accumulators as the residual by two MFMAs per
// channel tile against an identity fragment (exact: 1.0 x bf16 into fp32), then normalised in place - two in-lane passes and one lane
// exchange each for mean and variance - so GEMM1 reads (x1 - mean) rstd while gamma / beta are already folded into W1 / b1 by the
// engine's packer.  Replaces the proj GEMM, the LayerNorm launch and both Mlp GEMMs (with the 4C hidden tensor's round trip).
template <int C, int HID, int RB, int KC, bool LN>
__global__ __launch_bounds__(256, 1) void mlp_rows_kernel(const bf16* __restrict__ X, bf16* __restrict__ Y, const unsigned char* __restrict__ wimg,
                                                          const float* __restrict__ b1img, const float* __restrict__ b2, const bf16* __restrict__ CTX,
                                                          const float* __restrict__ bproj, const float ln_eps, const int M, const int n_tiles) {
  constexpr int NCT = C / 32, NKS = C / 16, NCH = HID / 32;
  constexpr int SLF = mr_slot_frags(C), MR_SLOT = SLF * 1024;      // ring slot: SLF fragments
  constexpr int PW = SLF / MR_NW, WSH = PW * 1024;                 // LDS-DMA pieces per wave and slot; a wave's share of a slot image
  constexpr int SL = RB * NKS;                                     // MFMA slots per group (GEMM1 or GEMM2 of one chunk): 32, or 24 at C = 384
  constexpr int PKS = KC / 16, PFR = PKS * NCT;          // proj: k-steps of 16 ctx channels, fragments in (k-step outer, c-tile inner) order
  constexpr int PSLOTS = (PFR + SLF - 1) / SLF;          // ring slots of SLF fragments, the last one padded (its tail is never read)
  static_assert(KC % 16 == 0, "whole k-steps");
  constexpr int PPC = 2 * NKS / SLF;                     // ring slots per hidden chunk: 1 (W1 | W2) or 2 (W1, W2)
  constexpr int BM = MR_NW * 32 * RB;                    // token rows per workgroup tile
  constexpr int FD = MR_FD;
  static_assert((SL == 32 || (SL == 24 && RB == 1)) && 2 * NKS == PPC * SLF && (PPC == 1 || PPC == 2), "register budget: <= 128 x + 256 y VGPRs");
  static_assert(!LN || (KC > 0 && RB == 1), "the LayerNorm variant continues from the proj prologue");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const b1tab = reinterpret_cast<float*>(smem + MR_NST * MR_SLOT);
  float* const bptab = b1tab + HID;                      // LN: proj bias, fc2 bias (channel order)
  float* const b2tab = bptab + C;
  // GELU table of the bf16 build (fsvit_common.h gelu_tab, round 6) behind the bias tables; TABC = LDS address of its centre (wave-uniform)
  unsigned char* const gtab = smem + MR_NST * MR_SLOT + HID * 4 + (LN ? 2 * C * 4 : 0);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, kh = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)(lptrm_t)smem;
  const unsigned voff = (unsigned)(wave * WSH + lane * 16);            // this lane's 16 bytes inside a slot image
  if ((int)blockIdx.x >= n_tiles) return;

  // bias table of conv1 (already in accumulator order) -> LDS
  for (int i = t; i < HID; i += MR_NW * 64) b1tab[i] = b1img[i];
  if constexpr (LN)
    for (int i = t; i < C; i += MR_NW * 64) { bptab[i] = bproj[i]; b2tab[i] = b2[i]; }
  if constexpr (gelu_tab::ON) gelu_tab::fill(gtab, t, MR_NW * 64);
  const unsigned tabc = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptrm_t)gtab + 2u * gelu_tab::TN);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        // table written before the first ring barrier publishes it

  // ring: slot image n (the weight image is a sequence of NCH * PPC slot images, repeated for every tile) is issued right after
  // barrier n-3, waited for (counted vmcnt: 8 DMAs per wave and slot) before barrier n-1 and first read after barrier n: LDS-DMA
  // data is ordered for a ds_read only by the issuing wave's counted vmcnt followed by a barrier the reader passes, and the read
  // must sit one barrier interval AFTER that wait (cdna_hip_programming.md, 8-phase template).  A build that waited for image n
  // right before barrier n and read it right after produced rare inf tiles in the first 2 KB of a slot, coming and going with
  // code placement.  Barrier n also certifies that every wave has finished reading the slot refilled next.
  int issue_img = 0, issue_slot = 0;
  auto issue = [&]() {
    mr_dma<PW>(voff, wimg + (size_t)issue_img * MR_SLOT, lds0 + issue_slot * MR_SLOT + wave * WSH);
    issue_img = issue_img == PSLOTS + NCH * PPC - 1 ? 0 : issue_img + 1;
    issue_slot = issue_slot == MR_NST - 1 ? 0 : issue_slot + 1;
  };
#pragma unroll
  for (int i = 0; i < MR_NST - 1; ++i) issue();
  int slot = 0;
  bool first = true;
  auto ring_wait = [&]() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PW) : "memory");     // all but the newest slot image of this wave have landed: the image read after the NEXT barrier
    mr_bar();
  };
  auto ring_sync = [&]() {
    ring_wait();
    issue();
  };
  // the same refill piece by piece (0..7 in order), each placed between MFMAs of the interval that follows ring_wait()
  auto issue1 = [&](int piece) {
    mr_dma1(voff + piece * 1024, wimg + (size_t)issue_img * MR_SLOT, lds0 + issue_slot * MR_SLOT + wave * WSH + piece * 1024);
    if (piece == PW - 1) {
      issue_img = issue_img == PSLOTS + NCH * PPC - 1 ? 0 : issue_img + 1;
      issue_slot = issue_slot == MR_NST - 1 ? 0 : issue_slot + 1;
    }
  };
  auto next_slot = [&]() { slot = slot == MR_NST - 1 ? 0 : slot + 1; };

  u32x4 idf[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};      // LN: identity fragments of the two k-steps of a channel tile
  if constexpr (LN) {
    const int ch = 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        idf[q][e] = mr_pk2(ch == 16 * kh + 8 * q + 2 * e ? 1.0f : 0.0f, ch == 16 * kh + 8 * q + 2 * e + 1 ? 1.0f : 0.0f);
  }
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    // ---- this wave's 32 RB token rows -> registers (tail rows re-read the last valid row; their results are never stored)
    bool mok[RB];
    size_t rowoff[RB];
    int mrow[RB];
    u32x4 xr[RB][NKS];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int m = tile * BM + (wave * RB + rb) * 32 + r;
      mok[rb] = m < M;
      mrow[rb] = mok[rb] ? m : M - 1;
      rowoff[rb] = (size_t)mrow[rb] * C + 16 * kh;
    }
    // The x (ctx) loads and the previous tile's stores share the vmcnt queue with the ring's DMAs: drain once per tile (every DMA older
    // than these loads landed long ago).  The registers are threaded through the wait so no use can be scheduled above it.
    auto load_x = [&]() {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int s = 0; s < NKS; ++s) xr[rb][s] = mr_gload16(X + rowoff[rb] + 32 * (s >> 1) + 8 * (s & 1));
      mr_wait_loads<RB * NKS>(&xr[0][0]);
    };
    f32x16 yacc[RB][NCT];
    // wait states MFMA -> v_accvgpr_read, accumulators threaded through
    auto yacc_settle = [&]() {
      f32x16* yf = &yacc[0][0];
      static_assert(RB * NCT == 16 || RB * NCT == 12, "operand list");
      if constexpr (RB * NCT == 16)
        asm volatile("s_nop 15\n\ts_nop 3"
                     : "+a"(yf[0]), "+a"(yf[1]), "+a"(yf[2]), "+a"(yf[3]), "+a"(yf[4]), "+a"(yf[5]), "+a"(yf[6]), "+a"(yf[7]),
                       "+a"(yf[8]), "+a"(yf[9]), "+a"(yf[10]), "+a"(yf[11]), "+a"(yf[12]), "+a"(yf[13]), "+a"(yf[14]), "+a"(yf[15]));
      else
        asm volatile("s_nop 15\n\ts_nop 3"
                     : "+a"(yf[0]), "+a"(yf[1]), "+a"(yf[2]), "+a"(yf[3]), "+a"(yf[4]), "+a"(yf[5]), "+a"(yf[6]), "+a"(yf[7]),
                       "+a"(yf[8]), "+a"(yf[9]), "+a"(yf[10]), "+a"(yf[11]));
    };
    if constexpr (KC > 0) {
      u32x4 cr[RB][PKS];                                   // ctx rows as B operands: token r, channels 16 ks + 8 kh .. +7
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int ks = 0; ks < PKS; ++ks) cr[rb][ks] = mr_gload16(CTX + (size_t)mrow[rb] * KC + 16 * ks + 8 * kh);
      mr_wait_loads<RB * PKS>(&cr[0][0]);
      if (first) { mr_bar(); first = false; }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) mfma32_a_zero(yacc[rb][ct]);
#pragma unroll
      for (int ps = 0; ps < PSLOTS; ++ps) {
        ring_sync();
        const unsigned char* sp = smem + slot * MR_SLOT + lane * 16;
        u32x4 fr[FD];
#pragma unroll
        for (int i = 0; i < FD; ++i) fr[i] = *reinterpret_cast<const u32x4*>(sp + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int fi = 0; fi < SLF; ++fi) {
          const int g = ps * SLF + fi, ks = g / NCT, ct = g % NCT;
          if (g < PFR) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) mfma32_a(fr[fi % FD], cr[rb][ks < PKS ? ks : 0], yacc[rb][ct]);
          }
          if (fi + FD < SLF && g + FD < PFR) fr[fi % FD] = *reinterpret_cast<const u32x4*>(sp + (fi + FD) * 1024);
          __builtin_amdgcn_sched_barrier(0);
        }
        next_slot();
      }
      // x1 = proj + x, rounded to bf16 into the x registers (accumulator channel order == x register order)
      load_x();
      yacc_settle();
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const bf16x8 xv = __builtin_bit_cast(bf16x8, xr[rb][2 * ct + q]);
            float pb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if constexpr (LN) {
              const f32x4 b0 = *reinterpret_cast<const f32x4*>(bptab + 32 * ct + 16 * kh + 8 * q);
              const f32x4 b1 = *reinterpret_cast<const f32x4*>(bptab + 32 * ct + 16 * kh + 8 * q + 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) { pb[e] = b0[e]; pb[4 + e] = b1[e]; }
            }
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float a0 = yacc[rb][ct][8 * q + 2 * e] + (float)xv[2 * e], a1 = yacc[rb][ct][8 * q + 2 * e + 1] + (float)xv[2 * e + 1];
              if constexpr (LN) { a0 += pb[2 * e]; a1 += pb[2 * e + 1]; }
              o[e] = mr_pk2(a0, a1);
            }
            xr[rb][2 * ct + q] = o;
          }
    } else {
      load_x();
      if (first) { mr_bar(); first = false; }             // one barrier between the drain above and the first reads of image 0
    }
    if constexpr (LN) {
      // residual: yacc := x1 through the identity fragments (accumulator row n of tile ct <-> channel 16 (n>>2 & 1) + 4 (n>>3) + (n & 3), the k slot
      // (kh, e8) of step 2 ct + q <-> channel 16 kh + 8 q + e8).  Every x register is threaded through before the nops: VALU-written -> MFMA SrcB.
#pragma unroll
      for (int o = 0; o + 4 < NKS; o += 4) asm volatile("" : "+v"(xr[0][o]), "+v"(xr[0][o + 1]), "+v"(xr[0][o + 2]), "+v"(xr[0][o + 3]) :: "memory");
      // (the identity fragments too: the compiler may have parked them in AGPRs - a v_accvgpr_read right in front of the first MFMA is the same hazard)
      asm volatile("s_nop 7" : "+v"(xr[0][NKS - 4]), "+v"(xr[0][NKS - 3]), "+v"(xr[0][NKS - 2]), "+v"(xr[0][NKS - 1]), "+v"(idf[0]), "+v"(idf[1]) :: "memory");
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        mfma32_a_z(idf[0], xr[0][2 * ct], yacc[0][ct]);
        mfma32_a(idf[1], xr[0][2 * ct + 1], yacc[0][ct]);
      }
      mr_layernorm_rows<NKS, C>(xr[0], ln_eps);
    } else {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) mfma32_a_zero(yacc[rb][ct]);
    }

    // ---- hidden chunks, software-pipelined inside the wave.  One wave per SIMD issues in order and MFMA shares its issue port with
    // the VALU, so GEMM1 -> GELU -> GEMM2 run back to back left the matrix pipe idle through every GELU, every exposed LDS latency
    // and every block of LDS-DMA issues.  Measured on gfx950 (tools/probes/mfma_valu_overlap.hip, mfma_chain.hip): behind one
    // 32x32x16 MFMA (32 cycles) ~7 plain VALU issue slots are free (a transcendental takes two); PACKED fp32 (v_pk_*) does not
    // overlap at all; a chain of dependent MFMAs on one accumulator issues at the full rate.  So body j of the chunk loop is 64 MFMA
    // slots with the (scalar) GELU of chunk j sliced between them:
    //     slots  0..31   GEMM2(j-1)   (B operands hp_prev; fragment-major, row blocks inner)
    //     slots 32..63   GEMM1(j+1)   (one chain per row block, row block 0 first: its accumulator is free after slot 31, row
    //                                  block 1's only after slot 47 - the GELU window is 48 slots at RB = 2)
    // hacc(j) is read by the GELU micro-stages only before its chain is restarted; hp is double-buffered across bodies.  The weight
    // stream is packed in exactly this order of 16 / 32-fragment groups:  W1(0) W1(1) | W2(0) W1(2) | W2(1) W1(3) | ... |
    // W2(NCH-3) W1(NCH-1) | W2(NCH-2) W2(NCH-1)   (one ring slot per pair at C = 256, per group at C = 512).
    f32x16 hacc[RB];
    u32x4 hpA[RB][2], hpB[RB][2];
    bool half = false;                                     // C = 256: which half of the current slot the next group sits in
    auto begin_group = [&]() -> const unsigned char* {
      unsigned a;
      if (PPC == 2) {
        ring_wait();
        a = slot * MR_SLOT + lane * 16;
        next_slot();
      } else {
        if (!half) ring_wait();
        a = slot * MR_SLOT + (half ? NKS * 1024 : 0) + lane * 16;
        if (half) next_slot();
        half = !half;
      }
      // opaque: the unrolled loop walks the ring with a fixed period, and hipcc otherwise hoists one address VGPR per fragment
      // beyond the 64 KB ds_read offset field (40 registers, spills); one base per group + immediate offsets is what is wanted
      asm volatile("" : "+v"(a));
      return smem + a;
    };
    // this group's share of the ring refill, called after every MFMA slot m (0..31) of a group: 4 (C = 256: half a slot image,
    // `half` was already flipped by begin_group) or 8 (C = 512) single pieces, 8 / 4 MFMAs apart
    auto refill = [&](int m) {
      constexpr int EVERY = SL / (PW / (PPC == 1 ? 2 : 1));
      if (m % EVERY == 1) issue1((PPC == 1 && !half ? 4 : 0) + m / EVERY);
    };
    // conv1's bias = initial value of a chain, read from LDS straight into the accumulator registers a few slots ahead of the chain
    auto bias_init = [&](int rb, int j) {
      const float* bp = b1tab + j * 32 + kh * 16;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bp + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) hacc[rb][4 * g + e] = b[e];
      }
    };

    // GELU of the chunk in hacc as micro-stages per PAIR of hidden units (pair q = 8 rb + 4 s2 + e -> hp[rb][s2][e]); the
    // schedule below keeps every dependent instruction at least one MFMA behind its producer.  The image carries W1 / 8, b1 / 8 and
    // 8 W2 (exact in bf16), so hacc = h / 8 and min(h^2, 64) / 64 is ONE multiply with the clamp modifier; the polynomial's
    // coefficients absorb the powers of two (every intermediate is the unscaled one times a power of two: results are bit-identical to
    // gelu_sig) and the product h/8 * sigmoid is what 8 W2 expects.
    constexpr int NP = RB * 8;
    float gx[NP][2], gu[NP][2], ge[NP][2];
    auto gA1 = [&](int q) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        gx[q][h] = hacc[q >> 3][2 * (q & 7) + h];
        asm("v_mul_f32_e64 %0, %1, %1 clamp" : "=v"(gu[q][h]) : "v"(gx[q][h]));
      }
    };
    auto gA2 = [&](int q) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float pp = fmaf(1.0153755e-3f * 32768.0f, gu[q][h], -1.0678257e-1f * 512.0f);
        pp = fmaf(pp, gu[q][h], -2.3011138f * 8.0f);
        gu[q][h] = gx[q][h] * pp;
      }
    };
    auto gE = [&](int q) {
#pragma unroll
      for (int h = 0; h < 2; ++h) ge[q][h] = __builtin_amdgcn_exp2f(gu[q][h]);
    };
    auto gBa = [&](int q) {
#pragma unroll
      for (int h = 0; h < 2; ++h) ge[q][h] = 1.0f + ge[q][h];
    };
    auto gBr = [&](int q) {
#pragma unroll
      for (int h = 0; h < 2; ++h) ge[q][h] = __builtin_amdgcn_rcpf(ge[q][h]);
    };
    auto gC = [&](int q, u32x4 (&hp)[RB][2]) { hp[q >> 3][(q >> 2) & 1][q & 3] = mr_pk2(gx[q][0] * ge[q][0], gx[q][1] * ge[q][1]); };
    // Pair q starts at slot st(q) = q CAD2 / 2:  RB = 2: a pair every 3.5 slots (6 issue slots per MFMA), RB = 1: every 4 (6 / 6 / 4 / 5).
    // Seven per MFMA already stretch the MFMA slot by ~20 % (mfma_valu_overlap probe); six cost ~10 %.  The stage that reads the
    // accumulator (A1) may not slip behind the restart of its chain (slot 28 for row block 0, 46 for row block 1): the last pairs of
    // row block 1 run A1 early and carry (x, u) in registers.
#ifndef MR_CAD384
#define MR_CAD384 8
#endif
    constexpr int CAD2 = RB == 2 ? 7 : (SL == 24 ? MR_CAD384 : 8);
    constexpr int O_A2 = 1, O_E = 2, O_BA = RB == 2 ? 4 : 3, O_BR = RB == 2 ? 5 : 4, O_C = RB == 2 ? 6 : 7;
    auto st = [](int q) { return q * CAD2 / 2; };
    auto st_a1 = [&](int q) { const int lim = (RB == 2 ? 45 : SL - 4) - (NP - 1 - q); return (RB == 1 || q >= 8) && st(q) > lim ? lim : st(q); };
    constexpr int G_LAST = (NP - 1) * CAD2 / 2 + O_C;     // slot of the last micro-stage
    // Round 6 (bf16 build): the same passes as TABLE LOOK-UPS on the bf16-rounded pre-activation (fsvit_common.h gelu_tab; stage1_w4.hip's t_slot):
    //   T1 (where A1 reads the accumulator) code pair | +1 magnitudes re-based | +2 top clamp, sign mask | +3 index | +4 addresses, gathers | +6 / +7 join
    // 9 VALU + 2 ds_read_u16 per pair instead of 17 VALU incl. 4 transcendentals; a packed 16-bit result and its reader never share a slot.
    unsigned tc[NP], ta[NP], tm[NP], tl[NP], th[NP];
    auto gelu_slot = [&](int m, u32x4 (&hp)[RB][2]) {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        if constexpr (gelu_tab::ON) {
          if (m == st_a1(q)) {
            tc[q] = mr_pk2(hacc[q >> 3][2 * (q & 7)], hacc[q >> 3][2 * (q & 7) + 1]);
            asm("" : "+v"(tc[q]));                           // (opaque: otherwise the sign shift converts the two floats again)
          }
          if (m == st(q) + 1) ta[q] = gelu_tab::rebase(tc[q]);
          if (m == st(q) + 2) { tm[q] = gelu_tab::sign_mask(tc[q]); ta[q] = gelu_tab::clamp_top(ta[q]); }
          if (m == st(q) + 3) ta[q] ^= tm[q];
          if (m == st(q) + 4) {
            unsigned a0, a1;
            gelu_tab::addresses(ta[q], tabc, a0, a1);
            tl[q] = gelu_tab::gather(a0);
            th[q] = gelu_tab::gather(a1);
          }
          if (m == st(q) + O_C) hp[q >> 3][(q >> 2) & 1][q & 3] = tl[q] | (th[q] << 16);
        } else {
          if (m == st_a1(q)) gA1(q);
          if (m == st(q) + O_A2) gA2(q);
          if (m == st(q) + O_E) gE(q);
          if (m == st(q) + O_BA) gBa(q);
          if (m == st(q) + O_BR) gBr(q);
          if (m == st(q) + O_C) gC(q, hp);
        }
      }
    };
    static_assert(G_LAST < 2 * SL, "the GELU fits in one body");

    // One body.  G2: slots 0..31 carry GEMM2 of the previous chunk (B operands hp_prev);  G1: slots 32..63 carry GEMM1 of chunk jn;
    // GE: the GELU of the chunk in hacc runs through the slots (-> hp_cur).
    auto body = [&](auto g2_, auto ge_, auto g1_, const int jn, u32x4 (&hp_prev)[RB][2], u32x4 (&hp_cur)[RB][2]) {
      constexpr bool G2 = decltype(g2_)::value, GE = decltype(ge_)::value, G1 = decltype(g1_)::value;
      auto foff2 = [](int f) { return (2 * (f % NCT) + f / NCT) * 1024; };       // image order [ct][s2], consumption (s2 outer, ct inner)
      auto foff1 = [](int i) { return (RB == 2 ? i % NKS : i) * 1024; };         // RB = 2: the NKS fragments once per row block
      u32x4 fr[FD];
      if (G2) {
        const unsigned char* sp = begin_group();
#pragma unroll
        for (int i = 0; i < FD; ++i) fr[i] = *reinterpret_cast<const u32x4*>(sp + foff2(i));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < SL; ++m) {
          const int f = m / RB, rb = m % RB;
          mfma32_a(fr[f % FD], hp_prev[rb][f / NCT], yacc[rb][f % NCT]);
          if (rb == RB - 1 && f + FD < NKS) fr[f % FD] = *reinterpret_cast<const u32x4*>(sp + foff2(f + FD));
          refill(m);
          // wait states MFMA (last of GEMM1) -> VALU read of hacc, spent behind the MFMA just issued; the accumulators are threaded
          // through so that no GELU instruction can be scheduled above
          if (m == 0 && GE) {
            if constexpr (RB == 2) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(hacc[0]), "+v"(hacc[RB - 1]));
            else asm volatile("s_nop 15\n\ts_nop 3" : "+v"(hacc[0]));
          }
          if (GE) gelu_slot(m, hp_cur);
          if (G1 && m == SL - 4) bias_init(0, jn);              // after the GELU's last direct read of row block 0's accumulator              // after the GELU's last direct read of row block 0's accumulator
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if (GE) {                                     // body 0: the GELU of chunk 0 on its own
        if constexpr (RB == 2) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(hacc[0]), "+v"(hacc[RB - 1]));
        else asm volatile("s_nop 15\n\ts_nop 3" : "+v"(hacc[0]));
#pragma unroll
        for (int m = 0; m < SL; ++m) gelu_slot(m, hp_cur);
        if (G1) bias_init(0, jn);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (G1) {
        const unsigned char* sp = begin_group();
#pragma unroll
        for (int i = 0; i < FD; ++i) fr[i] = *reinterpret_cast<const u32x4*>(sp + foff1(i));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < SL; ++i) {
          const int rb = RB == 2 ? i / NKS : 0, s = RB == 2 ? i % NKS : i;
          mfma32_v(fr[i % FD], xr[rb][s], hacc[rb]);
          if (i + FD < SL) fr[i % FD] = *reinterpret_cast<const u32x4*>(sp + foff1(i + FD));
          refill(i);
          if (GE) gelu_slot(SL + i, hp_cur);
          if (RB == 2 && i == NKS - 2) bias_init(1, jn);        // after the last GELU read of row block 1's accumulator (slot 45)
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if (GE) {                                     // last body: the tail of the GELU on its own
#pragma unroll
        for (int m = SL; m <= G_LAST; ++m) gelu_slot(m, hp_cur);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    using yes_t = std::integral_constant<bool, true>;
    using no_t = std::integral_constant<bool, false>;

    bias_init(0, 0);
    body(no_t{}, no_t{}, yes_t{}, 0, hpA, hpA);            // GEMM1(0)   (the chain of row block 1 starts from bias_init inside)
    body(no_t{}, yes_t{}, yes_t{}, 1, hpB, hpA);           // body 0:    GELU(0) | GEMM1(1)
#pragma unroll 1
    for (int j = 1; j < NCH - 1; j += 2) {
      body(yes_t{}, yes_t{}, yes_t{}, j + 1, hpA, hpB);    // body j (odd):  GEMM2(j-1) | GELU(j) | GEMM1(j+1)
      body(yes_t{}, yes_t{}, yes_t{}, j + 2, hpB, hpA);    // body j + 1
    }
    body(yes_t{}, yes_t{}, no_t{}, 0, hpA, hpB);           // body NCH-1: GEMM2(NCH-2) | GELU(NCH-1)
    if constexpr (RB == 2) asm volatile("s_nop 7" : "+v"(hpB[0][0]), "+v"(hpB[0][1]), "+v"(hpB[1][0]), "+v"(hpB[1][1]));   // VALU-written hp -> MFMA SrcB
    else asm volatile("s_nop 7" : "+v"(hpB[0][0]), "+v"(hpB[0][1]));
    body(yes_t{}, no_t{}, no_t{}, 0, hpB, hpB);            // GEMM2(NCH-1)

    yacc_settle();
    // ---- epilogue: + residual (the x registers), + optional bias of conv3, 2 x 16-byte stores per 32-channel tile
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const bf16x8 xv = __builtin_bit_cast(bf16x8, xr[rb][2 * ct + q]);
          bf16x8 o;
          if constexpr (LN) {         // the residual is already in the accumulator; fc2's bias from the LDS table
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(b2tab + 32 * ct + 16 * kh + 8 * q);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(b2tab + 32 * ct + 16 * kh + 8 * q + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)(yacc[rb][ct][8 * q + e] + (e < 4 ? b0[e & 3] : b1[e & 3]));
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float v = yacc[rb][ct][8 * q + e] + (float)xv[e];
              if (b2) v += b2[32 * ct + 16 * kh + 8 * q + e];
              o[e] = (bf16)v;
            }
          }
          if (mok[rb]) *reinterpret_cast<bf16x8*>(Y + rowoff[rb] + 32 * ct + 8 * q) = o;
        }
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no DMA may be in flight into the LDS of a finished workgroup
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// LayerNorm + Linear on token rows (the DeiT block head, deit.py:69 `attn(norm1(x))` up to the qkv Linear, deit.py:40-47):
//   Y[m][n] = bias[n] + sum_k W'[n][k] LN(x[m])[k],   C = 384, N a multiple of 32 (qkv: 1152), gamma / beta folded into W' / bias.
// As a 256x256-tile GEMM this layer has 6 K tiles per output tile (720 TF/s) and needs the normalised rows written and re-read by a
// LayerNorm launch.  Here a wave keeps its 32 token rows in registers (96 VGPRs), normalises them in place (mr_layernorm_rows) and walks the
// N / 32 output chunks: 24 MFMAs per chunk against W' fragments streamed through the LDS ring (image = the chunks' fragments in
// consumption order, 12 per 12 KB slot), then bias (the accumulator's initial value) -> bf16 -> two 16-byte stores per lane: the
// fragment rows are permuted at pack time so that a lane's 16 accumulators are 16 consecutive output channels.  Nothing is software-
// pipelined by hand: the kernel needs < 256 registers and 53 KB of LDS, so TWO workgroups share a CU and one's row load / LayerNorm /
// stores run under the other's MFMAs.  The stores share the vmcnt queue with the ring's DMAs: loads complete in order among loads, so
// "at most PW operations outstanding" still implies that every older slot image has landed (stores can only make the wait longer).
// LN = false, C = 512: the plain row-wise Linear (the Visformer stage-3 qkv conv, visformer.py:175, eval BatchNorm folded: 7 x 7 tokens are past
// what qkv_attn holds per image but the layer has the same 256-tile problem: K = 512 is 8 K tiles per epilogue, 753 TF/s).
// Measured (tools/bench_rows_gemm.py under rocprofv3, -DLGR_DIAG variants; C = 384 / 512): 818 / 850 TFLOP/s.  With stores, ring refill,
// barriers AND fragment reads removed the MFMA loop runs at 1280 TF/s (the practical ceiling of every long MFMA stream measured in this
// repository); dropping only the stores gains 25 %, only the ring refill 15 %, the barriers nothing, and - against the expectation that a
// fresh 1 KB fragment per MFMA (= the CU's 128 B/clk of LDS at the full MFMA rate) is the limit - dropping the fragment READS nothing
// either.  Spreading the VMEM instructions between the MFMAs (LGR_SPREAD: refill one piece at a time, stores deferred into the next chunk's
// first slot; the mlp_rows recipe) is worth 1 ... 2 % with two workgroups per CU.  What did NOT help the
// store cost: a deeper ring (slots x depth 2 x 4 ... 3 x 9, so that a wait leaves up to three images and the store acknowledgements in flight),
// whole 64-byte segments per store instruction (v_permlane16_swap of the two pieces between rows r and r + 16), even whole 128-byte lines
// (wrong layout, same bytes: -4 %); `nt` / `sc1` stores are 1.5 ... 2 x slower (the partial lines are no longer merged in L2).
#ifndef LGR_OCC
#define LGR_OCC 2
#endif
// GATHER (C = 4 Ci): the 2 x 2 / stride-2 patch-embedding conv (visformer.py:266-288, eval BatchNorm folded) - an output token's row is the
// concatenation of its four input pixels (k = (ky, kx, c): each 16-byte register load stays inside one pixel), and pos_embed [OH*OW][N]
// fp32 is added before the rounding: its 16 values per lane and chunk are loaded at the top of the chunk through asm (invisible to hipcc's
// waitcnt pass) and are complete after the chunk's second ring wait - they are older than the DMAs that wait leaves in flight.
template <int C, bool LN, int SPC, int NST, bool GATHER = false>
__global__ __launch_bounds__(256, LGR_OCC) void ln_gemm_rows_kernel(const bf16* __restrict__ X, bf16* __restrict__ Y, const unsigned char* __restrict__ wimg,
                                                              const float* __restrict__ bias, const float ln_eps, const int M, const int N,
                                                              const int n_tiles, const int gH = 0, const float* __restrict__ pos = nullptr) {
  constexpr int NKS = C / 16, SLF = NKS / SPC;                // k-steps; fragments per ring slot; SPC slots per chunk
  constexpr int SLOT = SLF * 1024, PW = SLF / MR_NW, WSH = PW * 1024, FD = 4, LAG = (NST - 3) * PW;
  static_assert(NKS % SLF == 0 && SLF % MR_NW == 0, "whole slots");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const btab = reinterpret_cast<float*>(smem + NST * SLOT);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, kh = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)(lptrm_t)smem;
  const unsigned voff = (unsigned)(wave * WSH + lane * 16);
  const int nch = N / 32, n_img = nch * SPC;
  if ((int)blockIdx.x >= n_tiles) return;
  for (int i = t; i < N; i += MR_NW * 64) btab[i] = bias ? bias[i] : 0.0f;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // the ring of mlp_rows_kernel: image n issued after barrier n-3, waited for before barrier n-1, first read after barrier n
  int issue_img = 0, issue_slot = 0, slot = 0;
  auto issue = [&]() {
    mr_dma<PW>(voff, wimg + (size_t)issue_img * SLOT, lds0 + issue_slot * SLOT + wave * WSH);
    issue_img = issue_img == n_img - 1 ? 0 : issue_img + 1;
    issue_slot = issue_slot == NST - 1 ? 0 : issue_slot + 1;
  };
#pragma unroll
  for (int i = 0; i < NST - 1; ++i) issue();
  bool first = true;

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int m = tile * (MR_NW * 32) + wave * 32 + r;
    const bool mok = m < M;
    const size_t row = (size_t)(mok ? m : M - 1);
    u32x4 xr[NKS];
    const float* posrow = nullptr;
    if constexpr (GATHER) {
      constexpr int CI = C / 4, SPT = NKS / 4;                 // channels per input pixel; k-steps per tap
      const int OH = gH >> 1, hw = OH * OH;
      const int b = (int)row / hw, pp = (int)row - b * hw, oy = pp / OH, ox = pp - oy * OH;
      const bf16* const px = X + ((size_t)(b * gH + 2 * oy) * gH + 2 * ox) * CI + 16 * kh;
      posrow = pos + (size_t)pp * N + 16 * kh;
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const int tap = s / SPT, ky = tap >> 1, kx = tap & 1;
        xr[s] = mr_gload16s(px + (size_t)(ky * gH + kx) * CI + 32 * ((s % SPT) >> 1) + 8 * (s & 1));
      }
    } else {
#pragma unroll
      for (int s = 0; s < NKS; ++s) xr[s] = mr_gload16s(X + row * C + 32 * (s >> 1) + 16 * kh + 8 * (s & 1));
    }
    mr_wait_loads<NKS>(&xr[0]);
    if (first) { mr_bar(); first = false; }
    if constexpr (LN) mr_layernorm_rows<NKS, C>(xr, ln_eps);
    bf16* const yrow = Y + row * N + 16 * kh;
    u32x4 po0 = {0u, 0u, 0u, 0u}, po1 = {0u, 0u, 0u, 0u};      // LGR_SPREAD: the previous chunk's output, stored between the MFMAs of this one
#pragma unroll 1
    for (int j = 0; j < nch; ++j) {
      f32x16 hacc;
      u32x4 posv[4];
      if constexpr (GATHER) {
#pragma unroll
        for (int g = 0; g < 4; ++g) posv[g] = mr_gload16(posrow + j * 32 + 4 * g);
      }
      {
        const float* bp = btab + j * 32 + kh * 16;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(bp + 4 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) hacc[4 * g + e] = b[e];
        }
      }
#pragma unroll
      for (int h = 0; h < SPC; ++h) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LAG) : "memory");
        mr_bar();
        // A 1 KB VMEM instruction holds the wave's issue stage for ~64 cycles (mlp_rows): the refill of the slot freed by this barrier goes out one
        // piece at a time, 4 MFMAs apart, and the previous chunk's two stores in the gaps of the chunk's first slot.
        const unsigned char* const dsrc = wimg + (size_t)issue_img * SLOT;
        const unsigned ddst = lds0 + issue_slot * SLOT + wave * WSH;
        issue_img = issue_img == n_img - 1 ? 0 : issue_img + 1;
        issue_slot = issue_slot == NST - 1 ? 0 : issue_slot + 1;
        unsigned a = slot * SLOT + lane * 16;
        asm volatile("" : "+v"(a));
        const unsigned char* sp = smem + a;
        slot = slot == NST - 1 ? 0 : slot + 1;
        u32x4 fr[FD];
#pragma unroll
        for (int i = 0; i < FD; ++i) fr[i] = *reinterpret_cast<const u32x4*>(sp + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < SLF; ++i) {
          mfma32_v(fr[i % FD], xr[h * SLF + i], hacc);
          if (i + FD < SLF) fr[i % FD] = *reinterpret_cast<const u32x4*>(sp + (i + FD) * 1024);
          constexpr int EVERY = SLF / PW;
          if (i % EVERY == 1 && i / EVERY < PW) mr_dma1(voff + (i / EVERY) * 1024, dsrc, ddst + (i / EVERY) * 1024);
          if (h == 0 && j > 0 && mok) {
            if (i == 3) mr_gstore16(yrow + (j - 1) * 32, po0);
            if (i == EVERY + 3) mr_gstore16(yrow + (j - 1) * 32 + 8, po1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(hacc));          // wait states MFMA -> VALU read of the accumulator
      if constexpr (GATHER) {
        static_assert(!GATHER || SPC == 2, "the pos loads are covered by the second ring wait of the chunk");
        asm volatile("" : "+v"(posv[0]), "+v"(posv[1]), "+v"(posv[2]), "+v"(posv[3]));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 pf = __builtin_bit_cast(f32x4, posv[g]);
#pragma unroll
          for (int e = 0; e < 4; ++e) hacc[4 * g + e] += pf[e];
        }
      }
      u32x4 o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o0[e] = mr_pk2(hacc[2 * e], hacc[2 * e + 1]);
        o1[e] = mr_pk2(hacc[8 + 2 * e], hacc[8 + 2 * e + 1]);
      }
      po0 = o0;
      po1 = o1;
    }
    if (mok) {
      mr_gstore16(yrow + (nch - 1) * 32, po0);
      mr_gstore16(yrow + (nch - 1) * 32 + 8, po1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no DMA may be in flight into the LDS of a finished workgroup
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// qkv conv + attention core for token maps of at most 32 tokens (Visformer stage 3 at 80 x 80: 25 tokens, C = 512, 6 heads x 96 padded;
// visformer.py:172-190) - the row-wise GEMM above with one image per wave, so that q, k and v of an image never leave its registers:
//   * per head, the weight stream delivers the chunks K0..2, Q0..2, V0..2 (32 channels each; packed in that order).  A K / Q chunk is the
//     usual D[channel][token] (A = W fragment, B = x rows): a lane (token, kh) holds 16 channels, which packed to bf16 are the two k-steps of
//     an MFMA operand with the token as its row - K as A, Q as B of  S^T[key][query] += K Q^T  (the same trick as GEMM1 -> GEMM2 of mlp_rows);
//   * S^T leaves a query's 32 keys in the 16 accumulators of its two lanes: mask (keys >= S), max / sum with one lane exchange each, exp2;
//     the probabilities packed to bf16 are the B operand P[query][key] of the next product;
//   * a V chunk is computed with the operands SWAPPED (A = x rows, B = W fragment - both are "32 rows x 8 k per lane", so the same registers
//     serve either side): D[token][channel] = V^T with the keys in the accumulators of lane (channel, kh) in exactly the key order of S^T -
//     packed, it is the A operand of  O^T[d][query] = V^T P^T  without any transpose through LDS; the fragment-row permutation of the image
//     makes lane (query, kh) end up with 16 consecutive head channels = two 16-byte stores.
// 7 of a wave's 32 rows are padding at S = 25 (masked as keys, never stored as queries): 28 % more MFMAs than the plain GEMM, in exchange
// for the qkv tensor (2.2 GB per launch written and read back) and the attention launch.
template <int C, int HDC>
__global__ __launch_bounds__(256, 2) void qkv_attn_rows_kernel(const bf16* __restrict__ X, bf16* __restrict__ CTX, const unsigned char* __restrict__ wimg,
                                                               const float* __restrict__ bias, const float scale_log2e, const int B, const int S,
                                                               const int heads, const int n_tiles) {
  constexpr int NKS = C / 16, SPC = 2, SLF = NKS / SPC, NST = 4, FD = 4;
  constexpr int SLOT = SLF * 1024, PW = SLF / MR_NW, WSH = PW * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const btab = reinterpret_cast<float*>(smem + NST * SLOT);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, kh = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)(lptrm_t)smem;
  const unsigned voff = (unsigned)(wave * WSH + lane * 16);
  const int HD = HDC * 32, N = 3 * heads * HD, n_img = (N / 32) * SPC;
  if ((int)blockIdx.x >= n_tiles) return;
  for (int i = t; i < N; i += MR_NW * 64) btab[i] = bias ? bias[i] : 0.0f;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  int issue_img = 0, issue_slot = 0, slot = 0;
  auto issue = [&]() {
    mr_dma<PW>(voff, wimg + (size_t)issue_img * SLOT, lds0 + issue_slot * SLOT + wave * WSH);
    issue_img = issue_img == n_img - 1 ? 0 : issue_img + 1;
    issue_slot = issue_slot == NST - 1 ? 0 : issue_slot + 1;
  };
#pragma unroll
  for (int i = 0; i < NST - 1; ++i) issue();
  bool first = true;
  const int perm = 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);      // fragment row r <-> channel perm of its chunk (pack kernel)

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int img = tile * MR_NW + wave;
    const bool ok = img < B && r < S;
    const size_t row = (size_t)(img < B ? img : B - 1) * S + (r < S ? r : S - 1);
    u32x4 xr[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) xr[s] = mr_gload16s(X + row * C + 32 * (s >> 1) + 16 * kh + 8 * (s & 1));
    mr_wait_loads<NKS>(&xr[0]);
    if (first) { mr_bar(); first = false; }
    bf16* const crow = CTX + row * (size_t)(heads * HD) + 16 * kh;

    // one chunk of 32 output channels: NKS MFMAs over the two ring slots of its fragments.  swapped: A = x rows, B = fragment -> D[token][channel]
    auto chunk = [&](auto swapped_, f32x16& acc) {
      constexpr bool SW = decltype(swapped_)::value;
#pragma unroll
      for (int h2 = 0; h2 < SPC; ++h2) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PW) : "memory");
        mr_bar();
        issue();
        unsigned a = slot * SLOT + lane * 16;
        asm volatile("" : "+v"(a));
        const unsigned char* sp = smem + a;
        slot = slot == NST - 1 ? 0 : slot + 1;
        u32x4 fr[FD];
#pragma unroll
        for (int i = 0; i < FD; ++i) fr[i] = *reinterpret_cast<const u32x4*>(sp + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < SLF; ++i) {
          if constexpr (SW) mfma32_v(xr[h2 * SLF + i], fr[i % FD], acc);
          else mfma32_v(fr[i % FD], xr[h2 * SLF + i], acc);
          if (i + FD < SLF) fr[i % FD] = *reinterpret_cast<const u32x4*>(sp + (i + FD) * 1024);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc));          // wait states MFMA -> VALU read of the accumulator
    };
    auto bias_rows = [&](f32x16& acc, int src) {                 // D[channel][token]: lane (token, kh) holds channels 16 kh + i
      const float* bp = btab + src * 32 + kh * 16;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bp + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * g + e] = b[e];
      }
    };
    auto pack2 = [&](const f32x16& acc, u32x4 (&o)[2]) {          // accumulators 8 t .. 8 t + 7 -> k-step t of an MFMA operand
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[tt][e] = mr_pk2(acc[8 * tt + 2 * e], acc[8 * tt + 2 * e + 1]);
    };
    using yes_t = std::integral_constant<bool, true>;
    using no_t = std::integral_constant<bool, false>;

#pragma unroll 1
    for (int h = 0; h < heads; ++h) {
      u32x4 kp[HDC][2];
#pragma unroll
      for (int c = 0; c < HDC; ++c) {
        f32x16 acc;
        bias_rows(acc, (heads + h) * HDC + c);
        chunk(no_t{}, acc);
        pack2(acc, kp[c]);
      }
      f32x16 sacc;
#pragma unroll
      for (int i = 0; i < 16; ++i) sacc[i] = 0.0f;
#pragma unroll
      for (int c = 0; c < HDC; ++c) {
        f32x16 acc;
        bias_rows(acc, h * HDC + c);
        chunk(no_t{}, acc);
        u32x4 qp[2];
        pack2(acc, qp);
        asm volatile("s_nop 7" : "+v"(qp[0]), "+v"(qp[1]), "+v"(kp[c][0]), "+v"(kp[c][1]), "+v"(sacc));     // VALU-written operands -> MFMA
        mfma32_v(kp[c][0], qp[0], sacc);
        mfma32_v(kp[c][1], qp[1], sacc);
      }
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sacc));
      // softmax over the 32 keys of this lane pair: accumulator i <-> key 8 (i >> 2) + 4 kh + (i & 3)
      float pv[16], mx = -3.0e38f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        pv[i] = (8 * (i >> 2) + 4 * kh + (i & 3)) < S ? sacc[i] : -3.0e38f;
        mx = fmaxf(mx, pv[i]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        pv[i] = __builtin_amdgcn_exp2f((pv[i] - mx) * scale_log2e);
        sum += pv[i];
      }
      sum += __shfl_xor(sum, 32);
      const float inv = __builtin_amdgcn_rcpf(sum);
      u32x4 pp[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int e = 0; e < 4; ++e) pp[tt][e] = mr_pk2(pv[8 * tt + 2 * e], pv[8 * tt + 2 * e + 1]);
#pragma unroll
      for (int c = 0; c < HDC; ++c) {
        f32x16 acc;
        const float bv = btab[((2 * heads + h) * HDC + c) * 32 + perm];       // D[token][channel]: lane = channel, every accumulator a token
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = bv;
        asm volatile("s_nop 1" : "+v"(acc));                        // VALU-written accumulator -> MFMA SrcC (tools/check_mfma_hazard.py)
        chunk(yes_t{}, acc);
        u32x4 vp[2];
        pack2(acc, vp);
        asm volatile("s_nop 7" : "+v"(vp[0]), "+v"(vp[1]), "+v"(pp[0]), "+v"(pp[1]));
        f32x16 oacc;
        mfma32_v_z(vp[0], pp[0], oacc);
        mfma32_v(vp[1], pp[1], oacc);
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(oacc));
        u32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0[e] = mr_pk2(oacc[2 * e] * inv, oacc[2 * e + 1] * inv);
          o1[e] = mr_pk2(oacc[8 + 2 * e] * inv, oacc[8 + 2 * e + 1] * inv);
        }
        if (ok) {
          mr_gstore16(crow + h * HD + c * 32, o0);
          mr_gstore16(crow + h * HD + c * 32 + 8, o1);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// norm1 + qkv Linear + attention core of a ViT / DeiT block in ONE launch (deit.py:40-58,:69; DeiT-S: C = 384, 6 heads x 64, S = 197 tokens):
// the qkv tensor - 5.8 GB per block written by ln_gemm_rows and read back by attention_v2 at 12 800 images - never exists.
// qkv_attn_rows_kernel above keeps an image in ONE wave (<= 32 tokens).  Here an image is NB = ceil(S / 32) <= 8 row blocks and a workgroup of
// 8 waves owns it: wave w = row block w (its 32 token rows normalised in registers, as in ln_gemm_rows).  Per head the weight stream delivers
// K0 K1 | Q0 Q1 | V0 V1 (the image of launch_qkv_attn_rows_pack); every wave computes ITS rows' K / Q / V chunks exactly as above, but
//   * the packed K fragments (A operands of S^T = K Q^T, row = key) and the packed V^T fragments (A operands of O^T = V^T P^T, computed with
//     swapped operands) are written to LDS - 2 x 2 KB per row block and chunk, lane-linear: the reader's register image IS the writer's - and
//     every wave reads all NB blocks back: 64 KB of LDS for both;
//   * the attention runs flash-style over the key blocks: S^T_kb (4 MFMAs) -> masked online softmax in the lane pair of a query (running max
//     and sum, O rescaled by exp2 of the max's change) -> O^T += V^T_kb P_kb (4 MFMAs); registers hold one key block's scores at a time.
// Synchronisation rides on the weight ring: every chunk passes the ring barrier, so K (written before the Q chunks) is visible when the first
// score is formed, V needs one barrier of its own, and the next head's first ring barrier guarantees that all waves have left this head's K / V.
// Rows beyond S (27 of 224 at S = 197) and the waves beyond NB compute on clamped rows and store nothing (12 % of the MFMAs).
template <int C, int HDC>
__global__ __launch_bounds__(512, 1) void vit_attn_rows_kernel(const bf16* __restrict__ X, bf16* __restrict__ CTX, const unsigned char* __restrict__ wimg,
                                                               const float* __restrict__ bias, const float eps, const float scale_log2e, const int B,
                                                               const int S, const int heads) {
  constexpr int NWV = 8, NKS = C / 16, SLF = NKS, NST = 3, FD = 4;
  constexpr int SLOT = SLF * 1024, PW = SLF / NWV, WSH = PW * 1024;
  constexpr int KV1 = HDC * 2 * 1024;                               // one row block's packed K (or V^T) fragments of a head
  static_assert(SLF % NWV == 0 && (PW == 3 || PW == 4 || PW == 2), "pieces per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const kx = smem + NST * SLOT;
  unsigned char* const vx = kx + NWV * KV1;
  float* const btab = reinterpret_cast<float*>(vx + NWV * KV1);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, kh = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)(lptrm_t)smem;
  const unsigned voff = (unsigned)(wave * WSH + lane * 16);
  const int HD = HDC * 32, N = 3 * heads * HD, n_img = N / 32;
  const int NB = (S + 31) >> 5;
  if ((int)blockIdx.x >= B) return;
  for (int i = t; i < N; i += NWV * 64) btab[i] = bias ? bias[i] : 0.0f;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  int issue_img = 0, issue_slot = 0, slot = 0;
  auto issue = [&]() {
    mr_dma<PW>(voff, wimg + (size_t)issue_img * SLOT, lds0 + issue_slot * SLOT + wave * WSH);
    issue_img = issue_img == n_img - 1 ? 0 : issue_img + 1;
    issue_slot = issue_slot == NST - 1 ? 0 : issue_slot + 1;
  };
#pragma unroll
  for (int i = 0; i < NST - 1; ++i) issue();
  bool first = true;
  const int perm = 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);      // fragment row r <-> channel perm of its chunk (pack kernel)
  unsigned char* const kmine = kx + wave * KV1 + lane * 16;
  unsigned char* const vmine = vx + wave * KV1 + lane * 16;

  const int tok = wave * 32 + r;
  const bool ok = tok < S;
  // a wave beyond the image's row blocks (wave 7 at 197 tokens) keeps every barrier and its share of the weight ring's DMA but skips the arithmetic (the
  // chip runs power-limited, 2.15 GHz at 1.3 kW; measured 51.0 -> 50.7 ms per DeiT-S step, i.e. within the noise: its SIMD partner and the other
  // SIMDs still set the pace)
  const bool active = wave < NB;
  u32x4 xr[NKS];
  auto load_rows = [&](int img) {                                    // this wave's 32 token rows of image `img` (clamped), in flight
    const size_t row = (size_t)img * S + (ok ? tok : S - 1);
#pragma unroll
    for (int s2 = 0; s2 < NKS; ++s2) xr[s2] = mr_gload16s(X + row * C + 32 * (s2 >> 1) + 16 * kh + 8 * (s2 & 1));
  };
  load_rows(blockIdx.x);
  for (int img = blockIdx.x; img < B; img += gridDim.x) {
    const size_t row = (size_t)img * S + (ok ? tok : S - 1);
    mr_wait_loads<NKS>(&xr[0]);
    mr_layernorm_rows<NKS, C>(xr, eps);
    if (first) { mr_bar(); first = false; }
    bf16* const crow = CTX + row * (size_t)(heads * HD) + 16 * kh;

    auto chunk = [&](auto swapped_, f32x16& acc) {
      constexpr bool SW = decltype(swapped_)::value;
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(PW) : "memory");      // (lgkmcnt: this wave's K / V fragment stores are in LDS before the barrier)
      mr_bar();
      issue();
      unsigned a = slot * SLOT + lane * 16;
      asm volatile("" : "+v"(a));
      const unsigned char* sp = smem + a;
      slot = slot == NST - 1 ? 0 : slot + 1;
      if (active) {
      u32x4 fr[FD];
#pragma unroll
      for (int i = 0; i < FD; ++i) fr[i] = *reinterpret_cast<const u32x4*>(sp + i * 1024);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < SLF; ++i) {
        if constexpr (SW) mfma32_v(xr[i], fr[i % FD], acc);
        else mfma32_v(fr[i % FD], xr[i], acc);
        if (i + FD < SLF) fr[i % FD] = *reinterpret_cast<const u32x4*>(sp + (i + FD) * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc));          // wait states MFMA -> VALU read of the accumulator
      }
    };
    auto bias_rows = [&](f32x16& acc, int src) {                 // D[channel][token]: lane (token, kh) holds channels 16 kh + i
      const float* bp = btab + src * 32 + kh * 16;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * g + e] = b4[e];
      }
    };
    auto pack2 = [&](const f32x16& acc, u32x4 (&o)[2]) {          // accumulators 8 t .. 8 t + 7 -> k-step t of an MFMA operand
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[tt][e] = mr_pk2(acc[8 * tt + 2 * e], acc[8 * tt + 2 * e + 1]);
    };
    using yes_t = std::integral_constant<bool, true>;
    using no_t = std::integral_constant<bool, false>;

#pragma unroll 1
    for (int h = 0; h < heads; ++h) {
      // ---- K chunks of this wave's rows -> LDS
#pragma unroll
      for (int c = 0; c < HDC; ++c) {
        f32x16 acc;
        bias_rows(acc, (heads + h) * HDC + c);
        chunk(no_t{}, acc);
        u32x4 kp[2];
        pack2(acc, kp);
        *reinterpret_cast<u32x4*>(kmine + (c * 2 + 0) * 1024) = kp[0];
        *reinterpret_cast<u32x4*>(kmine + (c * 2 + 1) * 1024) = kp[1];
      }
      // ---- Q chunks stay in registers (B operands of the scores)
      u32x4 qp[HDC][2];
#pragma unroll
      for (int c = 0; c < HDC; ++c) {
        f32x16 acc;
        bias_rows(acc, h * HDC + c);
        chunk(no_t{}, acc);
        pack2(acc, qp[c]);
      }
      // ---- V chunks, transposed by swapping the MFMA operands -> LDS
#pragma unroll
      for (int c = 0; c < HDC; ++c) {
        f32x16 acc;
        const float bv = btab[((2 * heads + h) * HDC + c) * 32 + perm];       // D[token][channel]: lane = channel, every accumulator a token
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = bv;
        asm volatile("s_nop 1" : "+v"(acc));                        // VALU-written accumulator -> MFMA SrcC (tools/check_mfma_hazard.py)
        chunk(yes_t{}, acc);
        u32x4 vp[2];
        pack2(acc, vp);
        *reinterpret_cast<u32x4*>(vmine + (c * 2 + 0) * 1024) = vp[0];
        *reinterpret_cast<u32x4*>(vmine + (c * 2 + 1) * 1024) = vp[1];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      mr_bar();                                                   // every row block's V^T (and, since the Q chunks, K) is in LDS
      if (h == heads - 1 && img + (int)gridDim.x < B) load_rows(img + gridDim.x);     // the rows are dead from here on: the next image's arrive under this head's attention

      // ---- attention over the key blocks (flash-style, one block's scores in registers at a time)
      f32x16 oacc[HDC];
#pragma unroll
      for (int c = 0; c < HDC; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[c][i] = 0.0f;
      float mrun = -3.0e38f, lrun = 0.0f;
      auto kv_load = [&](int kb, u32x4 (&kf)[HDC][2], u32x4 (&vf)[HDC][2]) {
        const unsigned char* const kb_k = kx + kb * KV1 + lane * 16;
        const unsigned char* const kb_v = vx + kb * KV1 + lane * 16;
#pragma unroll
        for (int c = 0; c < HDC; ++c)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            kf[c][tt] = *reinterpret_cast<const u32x4*>(kb_k + (c * 2 + tt) * 1024);
            vf[c][tt] = *reinterpret_cast<const u32x4*>(kb_v + (c * 2 + tt) * 1024);
          }
      };
      auto kv_step = [&](int kb, u32x4 (&kf)[HDC][2], u32x4 (&vf)[HDC][2]) {
        f32x16 sacc;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7" : "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(qp[0][0]), "+v"(qp[0][1]) :: "memory");
        mfma32_v_z(kf[0][0], qp[0][0], sacc);
        mfma32_v(kf[0][1], qp[0][1], sacc);
#pragma unroll
        for (int c = 1; c < HDC; ++c) {
          mfma32_v(kf[c][0], qp[c][0], sacc);
          mfma32_v(kf[c][1], qp[c][1], sacc);
        }
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sacc));
        // softmax over this block's 32 keys of the lane pair's query: accumulator i <-> key 32 kb + 8 (i >> 2) + 4 kh + (i & 3)
        float pv[16], mx = -3.0e38f;
        if (kb == NB - 1) {                                           // (wave-uniform: only the last block has keys beyond S)
          const int key0 = 32 * kb + 4 * kh;
#pragma unroll
          for (int i = 0; i < 16; ++i) pv[i] = (key0 + 8 * (i >> 2) + (i & 3)) < S ? sacc[i] : -3.0e38f;
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) pv[i] = sacc[i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, pv[i]);
        mx = xhalf_max(mx);                                           // the query's other 16 keys sit in the lane 32 away (v_permlane32_swap, no LDS trip)
        // Deferred rescale: the running maximum only moves when a block's maximum exceeds it by more than 2^DEFER in the exponent (wave-uniform
        // decision) - in between, the probabilities are taken against the stale maximum (at most 2^DEFER too large, fp32 / bf16 range is ample)
        // and the 32 accumulator multiplies of the rescale are skipped; exact in the end because O and the sum carry the same factor.
        constexpr float DEFER = 6.0f;
        if (__builtin_amdgcn_ballot_w64((mx - mrun) * scale_log2e > DEFER) != 0) {
          const float mnew = fmaxf(mrun, mx);
          const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * scale_log2e);
          lrun *= alpha;
          mrun = mnew;
#pragma unroll
          for (int c = 0; c < HDC; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[c][i] *= alpha;
        }
        const float nm = -mrun * scale_log2e;
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          pv[i] = __builtin_amdgcn_exp2f(fmaf(pv[i], scale_log2e, nm));
          sum += pv[i];
        }
        lrun += xhalf_sum(sum);
        u32x4 pp[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int e = 0; e < 4; ++e) pp[tt][e] = mr_pk2(pv[8 * tt + 2 * e], pv[8 * tt + 2 * e + 1]);
        asm volatile("s_nop 7" : "+v"(pp[0]), "+v"(pp[1]), "+v"(oacc[0]), "+v"(oacc[HDC - 1]), "+v"(vf[0][0]), "+v"(vf[0][1]));     // VALU-written operands -> MFMA
#pragma unroll
        for (int c = 0; c < HDC; ++c) {
          mfma32_v(vf[c][0], pp[0], oacc[c]);
          mfma32_v(vf[c][1], pp[1], oacc[c]);
        }
      };
      if (active) {                                                // two fragment sets: block kb + 1 is read from LDS under block kb's work
        u32x4 kfa[HDC][2], vfa[HDC][2], kfb[HDC][2], vfb[HDC][2];
        kv_load(0, kfa, vfa);
#pragma unroll 1
        for (int kb = 0; kb < NB; kb += 2) {
          if (kb + 1 < NB) kv_load(kb + 1, kfb, vfb);
          kv_step(kb, kfa, vfa);
          if (kb + 2 < NB) kv_load(kb + 2, kfa, vfa);
          if (kb + 1 < NB) kv_step(kb + 1, kfb, vfb);
        }
      }
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(oacc[0]), "+v"(oacc[HDC - 1]));
      const float inv = __builtin_amdgcn_rcpf(lrun);
#pragma unroll
      for (int c = 0; c < HDC; ++c) {
        u32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0[e] = mr_pk2(oacc[c][2 * e] * inv, oacc[c][2 * e + 1] * inv);
          o1[e] = mr_pk2(oacc[c][8 + 2 * e] * inv, oacc[c][8 + 2 * e + 1] * inv);
        }
        if (ok) {
          mr_gstore16(crow + h * HD + c * 32, o0);
          mr_gstore16(crow + h * HD + c * 32 + 8, o1);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// image for ln_gemm_rows_kernel from a standard packed layer w [N][kw] (K-contiguous rows): fragment (chunk j, k-step s), lane = 32 kh + r, e8:
//   W[32 j + 16 (r>>2 & 1) + 4 (r>>3) + (r & 3)][32 (s>>1) + 16 kh + 8 (s&1) + e8]   (accumulator i of lane (token, kh) = channel 32 j + 16 kh + i)
// heads > 0 (qkv_attn_rows_kernel; rows of w = (q | k | v, head, hdc chunks of 32)): image chunk (head h, part K | Q | V, c) = w's chunk
// ((k, q, v)[part] * heads + h) * hdc + c
__global__ void ln_gemm_pack_kernel(const bf16* __restrict__ w, int kw, bf16* __restrict__ wimg, int C, int N, int heads, int hdc) {
  const int NKS = C / 16;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)(N / 32) * NKS * 512) return;
  const int g = (int)(idx >> 9), lane = (int)(idx >> 3) & 63, e8 = (int)idx & 7;
  int j = g / NKS;
  if (heads > 0) {
    const int h = j / (3 * hdc), part = (j / hdc) % 3, c = j % hdc;
    j = ((part == 0 ? 1 : part == 1 ? 0 : 2) * heads + h) * hdc + c;
  }
  const int s = g % NKS, kh = lane >> 5, r = lane & 31;
  const int n = 32 * j + 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
  wimg[idx] = w[(size_t)n * kw + 32 * (s >> 1) + 16 * kh + 8 * (s & 1) + e8];
}

bool ln_gemm_rows_supported(int dtype, int C, int N) {
  constexpr bool on = true;
  return on && dtype == 1 && C == 384 && N >= 32 && N % 32 == 0;
}
// the same kernel without the LayerNorm at C = 512 (Visformer stage-3 qkv) (dispatch switch retired in round 6: tools/probes/variants/dispatch_switches.r06.patch)
bool gemm_rows_supported(int dtype, int C, int N) {
  constexpr bool on = true;
  return on && dtype == 1 && C == 512 && N >= 32 && N % 32 == 0;
}
size_t ln_gemm_rows_image_bytes(int C, int N) { return (size_t)(N / 32) * (C / 16) * 1024; }
int launch_ln_gemm_pack(const void* w, int kw, void* wimg, int C, int N, hipStream_t s) {
  const long n = (long)ln_gemm_rows_image_bytes(C, N) / 2;
  hipLaunchKernelGGL(ln_gemm_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16*)w, kw, (bf16*)wimg, C, N, 0, 0);
  return (int)hipGetLastError();
}
// qkv conv + attention on the rows kernel: C = 512, head dim padded to 96, maps of at most 32 tokens (dispatch switch retired in round 6)
bool qkv_attn_rows_supported(int dtype, int C, int heads, int hdp, int S) {
  constexpr bool on = true;
  return on && dtype == 1 && C == 512 && hdp == 96 && heads >= 1 && S >= 1 && S <= 32;
}
int launch_qkv_attn_rows_pack(const void* w, int kw, void* wimg, int C, int heads, int hdp, hipStream_t s) {
  const int N = 3 * heads * hdp;
  const long n = (long)ln_gemm_rows_image_bytes(C, N) / 2;
  hipLaunchKernelGGL(ln_gemm_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16*)w, kw, (bf16*)wimg, C, N, heads, hdp / 32);
  return (int)hipGetLastError();
}
// ctx [B*S][heads*hdp] = softmax(scale q k^T) v per image and head, q | k | v = bias + W x   (image from launch_qkv_attn_rows_pack)
int launch_qkv_attn_rows(const void* x, void* ctx, const void* wimg, const float* bias, int B, int S, int C, int heads, int hdp, float scale, hipStream_t s) {
  if (B <= 0) return 0;
  if (C != 512 || hdp != 96 || S < 1 || S > 32) return (int)hipErrorInvalidValue;
  auto kern = qkv_attn_rows_kernel<512, 3>;
  const int N = 3 * heads * hdp, lds = 4 * (512 / 32) * 1024 + N * 4;
  {    // per launch: the attribute is per device
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
  }
  const int n_tiles = (B + MR_NW - 1) / MR_NW;
  const int grid = n_tiles < 512 ? n_tiles : 512;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(MR_NW * 64), lds, s, (const bf16*)x, (bf16*)ctx, (const unsigned char*)wimg, bias, scale * 1.4426950408889634f, B,
                     S, heads, n_tiles);
  return (int)hipGetLastError();
}
// norm1 + qkv + attention of a ViT block in one launch (vit_attn_rows_kernel): C = 384, head dim 64, up to 256 tokens (dispatch switch retired in round 6)
bool vit_attn_rows_supported(int dtype, int C, int heads, int hdp, int S) {
  constexpr bool on = true;
  return on && dtype == 1 && C == 384 && hdp == 64 && heads >= 1 && S >= 1 && S <= 256;
}
// ctx [B*S][heads*64] = softmax(scale q k^T) v per image and head, q | k | v = bias + W' LN(x)   (image from launch_qkv_attn_rows_pack with C = 384)
int launch_vit_attn_rows(const void* x, void* ctx, const void* wimg, const float* bias, int B, int S, int C, int heads, int hdp, float eps, float scale,
                         hipStream_t s) {
  if (B <= 0) return 0;
  if (C != 384 || hdp != 64 || S < 1 || S > 256) return (int)hipErrorInvalidValue;
  auto kern = vit_attn_rows_kernel<384, 2>;
  const int N = 3 * heads * hdp, lds = 3 * (384 / 16) * 1024 + 2 * 8 * (2 * 2 * 1024) + N * 4;
  {    // per launch: the attribute is per device
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
  }
  const int grid = B < 256 ? B : 256;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, (const bf16*)x, (bf16*)ctx, (const unsigned char*)wimg, bias, eps, scale * 1.4426950408889634f, B, S, heads);
  return (int)hipGetLastError();
}
#ifndef LGR_SPC384     // ring geometry (tools/build_variant.sh sweeps)
#define LGR_SPC384 2
#define LGR_NST384 4
#endif
#ifndef LGR_SPC512
#define LGR_SPC512 2
#define LGR_NST512 4
#endif
template <int C, bool LN, int SPC, int NST, bool GATHER = false>
static int launch_gemm_rows_t(const void* x, void* y, const void* wimg, const float* bias, int M, int N, float eps, hipStream_t s, int gH = 0,
                              const float* pos = nullptr) {
  auto kern = ln_gemm_rows_kernel<C, LN, SPC, NST, GATHER>;
  const int lds = NST * (C / 16 / SPC) * 1024 + N * 4;
  {    // per launch: the attribute is per device
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
  }
  const int n_tiles = (M + MR_NW * 32 - 1) / (MR_NW * 32);
  const int grid = n_tiles < 256 * LGR_OCC ? n_tiles : 256 * LGR_OCC;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(MR_NW * 64), lds, s, (const bf16*)x, (bf16*)y, (const unsigned char*)wimg, bias, eps, M, N, n_tiles, gH, pos);
  return (int)hipGetLastError();
}
// 2 x 2 / stride-2 patch embedding on the rows kernel: x NHWC [B][H][H][Ci] with 4 Ci = 512, y [B (H/2)^2][N] = bias + W patches + pos
bool patch_embed_rows_supported(int dtype, int Ci, int H, int N) {
  constexpr bool on = true;
  return on && dtype == 1 && 4 * Ci == 512 && H >= 2 && H % 2 == 0 && N >= 32 && N % 32 == 0;
}
int launch_patch_embed_rows(const void* x, void* y, const void* wimg, const float* bias, const float* pos, int B, int H, int Ci, int N, hipStream_t s) {
  if (B <= 0) return 0;
  if (4 * Ci != 512 || (H & 1) || N % 32 || !pos) return (int)hipErrorInvalidValue;
  return launch_gemm_rows_t<512, false, 2, 4, true>(x, y, wimg, bias, B * (H / 2) * (H / 2), N, 0.0f, s, H, pos);
}
// y [M][N] = bias + W' LN(x [M][C])   (gamma / beta folded into W' / bias by the caller; image from launch_ln_gemm_pack)
int launch_ln_gemm_rows(const void* x, void* y, const void* wimg, const float* bias, int M, int C, int N, float eps, hipStream_t s) {
  if (M <= 0) return 0;
  if (C != 384 || N % 32) return (int)hipErrorInvalidValue;
  return launch_gemm_rows_t<384, true, LGR_SPC384, LGR_NST384>(x, y, wimg, bias, M, N, eps, s);
}
// y [M][N] = bias + W x [M][C]   (C = 512; bias may be null)
int launch_gemm_rows(const void* x, void* y, const void* wimg, const float* bias, int M, int C, int N, hipStream_t s) {
  if (M <= 0) return 0;
  if (C != 512 || N % 32) return (int)hipErrorInvalidValue;
  return launch_gemm_rows_t<512, false, LGR_SPC512, LGR_NST512>(x, y, wimg, bias, M, N, 0.0f, s);
}

// Builds the fragment-major weight image + bias table from the engine's standard packed layers (w1 [HID][k1w], w2 [C][k2w], optional
// wp [C][kpw] = the proj conv, K-contiguous bf16 rows).  Image = [proj: KC/16 k-steps x C/32 c-tiles of 1 KB fragments][per hidden
// chunk: C/16 W1 fragments, C/16 W2 fragments].  One thread per bf16 element of the image.
__global__ void mlp_pack_kernel(const bf16* __restrict__ w1, int k1w, const float* __restrict__ b1, const bf16* __restrict__ w2, int k2w,
                                const bf16* __restrict__ wp, int kpw, int KC, bf16* __restrict__ wimg, float* __restrict__ b1img, int C, int HID) {
  const int NCT = C / 32, NKS = C / 16, NCH = HID / 32;
  const int per_chunk = (NKS + 2 * NCT) * 512;
  const int slf = mr_slot_frags(C);
  const long proj_elems = (long)(((KC / 16) * NCT + slf - 1) / slf) * slf * 512;      // whole slots; fragments past KC read wp's zero padding (kpw >= 16 * ceil)
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < HID) {
    const int j = (int)idx >> 5, w = (int)idx & 31, kh = w >> 4, i = w & 15;
    b1img[idx] = b1 ? 0.125f * b1[j * 32 + 8 * (i >> 2) + 4 * kh + (i & 3)] : 0.0f;      // conv1 is carried at 1/8 scale (GELU clamp trick)
  }
  if (idx >= proj_elems + (long)NCH * per_chunk) return;
  bf16 v;
  if (idx < proj_elems) {       // fragment g = ks * NCT + ct: row = output channel (accumulator order), k = ctx channel 16 ks + 8 kh + e8
    const int g = (int)(idx >> 9), lane = (int)(idx >> 3) & 63, e8 = (int)idx & 7;
    const int ks = g / NCT, ct = g % NCT, kh = lane >> 5, r = lane & 31;
    const int c = 32 * ct + 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
    const int k = 16 * ks + 8 * kh + e8;
    v = k < kpw ? wp[(size_t)c * kpw + k] : (bf16)0.0f;
    wimg[idx] = v;
    return;
  }
  const long ii = idx - proj_elems;
  // the stream is a sequence of 2 NCH groups of NKS fragments in the order the pipelined chunk loop consumes them:
  //   W1(0) W1(1) | W2(0) W1(2) | W2(1) W1(3) | ... | W2(NCH-3) W1(NCH-1) | W2(NCH-2) W2(NCH-1)
  const int e = (int)(ii % per_chunk);
  const int grp = 2 * (int)(ii / per_chunk) + ((e >> 9) >= NKS ? 1 : 0);
  const bool is_w2 = grp >= 2 && (grp == 2 * NCH - 1 || (grp & 1) == 0);
  const int j = grp < 2 ? grp : grp == 2 * NCH - 1 ? NCH - 1 : is_w2 ? grp / 2 - 1 : (grp + 1) / 2;
  const int piece = (e >> 9) % NKS, lane = (e >> 3) & 63, e8 = e & 7;
  const int kh = lane >> 5, r = lane & 31;
  if (!is_w2) {
    const int s = piece;
    v = (bf16)(0.125f * (float)w1[(size_t)(j * 32 + r) * k1w + 32 * (s >> 1) + 16 * kh + 8 * (s & 1) + e8]);      // exact: a power of two
  } else {
    const int q = piece, ct = q >> 1, s2 = q & 1;
    const int c = 32 * ct + 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
    const int n = j * 32 + 8 * (2 * s2 + (e8 >> 2)) + 4 * kh + (e8 & 3);
    v = (bf16)(8.0f * (float)w2[(size_t)c * k2w + n]);
  }
  wimg[idx] = v;
}

// the ViT / DeiT block (proj + bias + residual, LayerNorm, Mlp with biases): DeiT-S geometry (was bit 3 of the retired FSVIT_MLP_ROWS switch)
bool mlp_rows_ln_supported(int dtype, int C, int hid, int KC) {
  constexpr int mode = 15;
  return dtype == 1 && (mode & 8) && C == 384 && hid == 1536 && KC == 384;
}
bool mlp_rows_supported(int dtype, int C, int hid) {
  constexpr int mode = 7;      // bit 0: C = 256, bit 1: C = 512, bit 2: proj fusion
  if (dtype != 1) return false;
  return (C == 256 && hid == 1024 && (mode & 1)) || (C == 512 && hid == 2048 && (mode & 2));
}
// proj fusion is built for the Visformer-S geometries: (C, KC) = (256, 288) and (512, 576) (6 heads x head dim padded to 48 / 96)
bool mlp_rows_proj_supported(int C, int hid, int KC) {
  constexpr int mode = 7;
  return (mode & 4) && ((C == 256 && hid == 1024 && KC == 288) || (C == 512 && hid == 2048 && KC == 576));
}
size_t mlp_rows_image_bytes(int C, int hid, int KC) {
  const size_t slf = (size_t)mr_slot_frags(C);
  const size_t proj_frags = ((size_t)(KC / 16) * (C / 32) + slf - 1) / slf * slf;
  return (proj_frags + (size_t)(hid / 32) * (C / 16 + 2 * (C / 32))) * 1024;
}

int launch_mlp_pack(const void* w1, int k1w, const float* b1, const void* w2, int k2w, const void* wp, int kpw, int KC, void* wimg, float* b1img, int C,
                    int hid, hipStream_t s) {
  const long n = (long)mlp_rows_image_bytes(C, hid, KC) / 2;
  hipLaunchKernelGGL(mlp_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16*)w1, k1w, b1, (const bf16*)w2, k2w,
                     (const bf16*)wp, kpw, KC, (bf16*)wimg, b1img, C, hid);
  return (int)hipGetLastError();
}

template <int C, int HID, int RB, int KC, bool LN = false>
static int launch_mlp_rows_t(const void* x, void* y, const void* wimg, const float* b1img, const float* b2, const void* ctx, const float* bproj, float eps,
                             int M, hipStream_t s) {
  auto kern = mlp_rows_kernel<C, HID, RB, KC, LN>;
  const int lds = MR_NST * mr_slot_frags(C) * 1024 + HID * 4 + (LN ? 2 * C * 4 : 0) + gelu_tab::BYTES;
  {    // per launch: the attribute is per DEVICE (a process-wide "done" flag skipped it on a second GPU), and the call is cheap
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
  }
  constexpr int BM = MR_NW * 32 * RB;
  const int n_tiles = (M + BM - 1) / BM;
  const int grid = n_tiles < 256 ? n_tiles : 256;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(MR_NW * 64), lds, s, (const bf16*)x, (bf16*)y, (const unsigned char*)wimg, b1img, b2, (const bf16*)ctx, bproj,
                     eps, M, n_tiles);
  return (int)hipGetLastError();
}

// ctx != nullptr: x1 = x + Wp ctx first (the image must have been packed with the same KC)
int launch_mlp_rows(const void* x, void* y, const void* wimg, const float* b1img, const float* b2, const void* ctx, int KC, int M, int C, int hid,
                    hipStream_t s) {
  if (M <= 0) return 0;
  if (!ctx) KC = 0;
  if (C == 256 && hid == 1024 && KC == 0) return launch_mlp_rows_t<256, 1024, 2, 0>(x, y, wimg, b1img, b2, nullptr, nullptr, 0.f, M, s);
  if (C == 512 && hid == 2048 && KC == 0) return launch_mlp_rows_t<512, 2048, 1, 0>(x, y, wimg, b1img, b2, nullptr, nullptr, 0.f, M, s);
  if (C == 256 && hid == 1024 && KC == 288) return launch_mlp_rows_t<256, 1024, 2, 288>(x, y, wimg, b1img, b2, ctx, nullptr, 0.f, M, s);
  if (C == 512 && hid == 2048 && KC == 576) return launch_mlp_rows_t<512, 2048, 1, 576>(x, y, wimg, b1img, b2, ctx, nullptr, 0.f, M, s);
  return (int)hipErrorInvalidValue;
}

// y = x1 + b2 + W2 GELU(W1' LN(x1) + b1'),  x1 = x + bp + Wp ctx   (gamma / beta of the LayerNorm folded into W1' / b1' by the caller)
int launch_mlp_rows_ln(const void* x, void* y, const void* wimg, const float* b1img, const float* bproj, const float* b2, const void* ctx, int KC, int M,
                       int C, int hid, float eps, hipStream_t s) {
  if (M <= 0) return 0;
  if (!ctx || !bproj || !b2) return (int)hipErrorInvalidValue;
  if (C == 384 && hid == 1536 && KC == 384) return launch_mlp_rows_t<384, 1536, 1, 384, true>(x, y, wimg, b1img, b2, ctx, bproj, eps, M, s);
  return (int)hipErrorInvalidValue;
}

}  // namespace FSVIT_NS
