// Visformer stage-1 block, second design (bf16):   y = x + conv3( GELU( conv2_g8_3x3( GELU( conv1( BN(x) ) ) ) ) )
// (test_phase/models/visformer.py:259-263 Block.forward with attn_disabled, Mlp :152-163; eval BN folded into conv1).
//
// stage1_fused.hip gives a half image to a 16-wave workgroup and moves every operand of every MFMA through LDS with two
// workgroup barriers per channel group: 47 ds_read_b128 per 33 MFMAs per wave, 16-token x 16-channel register tiles, waves
// parked at barriers 43 % of the time (rocprof r01_v8: 7800 cycles per group for 2100 cycles of MFMA work per SIMD, 19 % of
// the MFMA peak, 26 % of the whole step).  Here the unit of work is a BAND of 4 image rows owned by ONE WAVE for the whole
// block (one wave per SIMD, the full 512-entry register file, as mlp_rows.hip):
//   * the band's conv1 input - its 4 rows + one halo row above and below, 120 tokens x 128 channels - lives in registers as
//     MFMA B operands (128 VGPRs); the block's output accumulator (80 tokens x 128 channels fp32) in 160 AGPRs;
//   * per channel group g (the grouped conv keeps the 8 groups of 32 hidden channels independent until conv3):
//       conv1  H1g = GELU(W1g x + b1g)        2 n-tiles x 8 m-tiles x 4 k-chunks = 64 MFMAs (v_mfma_f32_16x16x32_bf16)
//              -> the wave's PRIVATE LDS patch (6 x 22 zero-bordered pixel grid, [k-chunk plane][pixel][16 B]): the only
//                 activation that touches LDS, because the 3x3 taps need its neighbours - written and read by the same wave,
//                 so no barrier orders it (the LDS pipeline executes one wave's accesses in issue order);
//       conv2  H2g = GELU(conv3x3(H1g, W2g))  9 taps x 2 n-tiles x 5 m-tiles = 90 MFMAs, B operand = tap-shifted b128 reads;
//       conv3  Y  += W3[:, g] H2g             8 n-tiles x 5 m-tiles = 40 MFMAs; H2g never leaves registers: the GELU'd conv2
//              accumulators, packed to bf16, ARE conv3's B operand (the k order this implies is baked into the packed W3);
//   * only weights are shared: the 34 fragments of a group (36 KB slot, fragment-major image built by stage1_pack_kernel)
//     stream through a 3-slot LDS ring by linear LDS-DMA; every wave reads each fragment once per group and feeds 5-8 MFMAs
//     with it.  The ring barrier (one per group) is the only workgroup-wide synchronisation;
//   * halo rows are recomputed by both neighbours (conv1 x 1.5): 12 % more MFMAs than the minimum, no inter-wave traffic.
// Per group and wave: 194 MFMAs, 79 ds_read_b128 (0.4 per MFMA instead of 1.4), 16 ds_write_b64.
#include <stdlib.h>

#include "fsvit_common.h"
#include "kernels.h"

// GELU of a register pair: packed (gelu_sig2) or two scalar gelu_sig (-DS1_SCALAR_GELU: packed fp32 does not issue behind MFMAs)
#ifdef S1_SCALAR_GELU
#define S1_GELU2(v) (f32x2{gelu_sig((v)[0]), gelu_sig((v)[1])})
#else
#define S1_GELU2(v) gelu_sig2(v)
#endif

namespace fsvit {

typedef __attribute__((address_space(3))) void* lptrs_t;

namespace s1r {
constexpr int C1 = 128, HID = 256, G = 8, W = 20, HIMG = 20;
#ifndef S1R_BAND
#define S1R_BAND 4
#endif
constexpr int BAND = S1R_BAND, NBAND = HIMG / BAND;   // bands of BAND rows per image
constexpr int RT = (BAND + 2) * W;                    // 120 conv1 tokens (band + halo rows)
constexpr int OTOK = BAND * W;                        // 80 output tokens
constexpr int MT1 = (RT + 15) / 16, MT2 = (OTOK + 15) / 16;   // m-tiles of 16 tokens: conv1, conv2 / conv3
constexpr int PW = W + 2;                             // zero-bordered H1 grid is 6 x 22 pixels
constexpr int PLANE = ((BAND + 2) * PW * 16 + 255) / 256 * 256;   // pixels x 16 B, rounded so that the plane stride is 0 mod 256 B
constexpr int SLOT = 36 * 1024;                      // 8 (W1) + 18 (W2) + 8 (W3) fragments, padded to 9 KiB per wave
constexpr int NW = BAND == 4 ? 4 : 8;                  // band 4: one wave per SIMD (512 registers); band 2: two per SIMD (256)
constexpr int NST = BAND == 4 ? 3 : 2;                // ring slots
constexpr int OFF_H1 = NST * SLOT;                    // 110592
constexpr int OFF_B1 = OFF_H1 + NW * 4 * PLANE;       // 147456
constexpr int LDS_BYTES = OFF_B1 + HID * 4;           // 148480
constexpr int FD = 4;                                 // fragment read-ahead
}  // namespace s1r

namespace {

// this wave's share of a 36 KiB slot: NP consecutive 1 KiB LDS-DMAs starting at piece p0 (immediate offsets move the global source
// and the LDS destination together; the 13-bit field covers four pieces per M0 setting)
__device__ __forceinline__ void s1r_dma4(unsigned voff, const void* sbase, unsigned lds) {
  unsigned keep;
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
}
__device__ __forceinline__ void s1r_dma1(unsigned voff, const void* sbase, unsigned lds) {
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
__device__ __forceinline__ void s1r_bar() {
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ u32x4 s1r_gload16(const void* p) {       // asm: invisible to hipcc's vmcnt bookkeeping (see mlp_rows.hip)
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// MFMAs from inline asm with the accumulator's register file spelled out (outputs in AGPRs, everything else in VGPRs); the
// wait states hipcc would insert around them are placed by hand below (mlp_rows.hip has the full account)
// S1R_ASM (band 4, one wave per SIMD, outputs in AGPRs) / compiler builtins (band 2, two waves per SIMD, everything in VGPRs: hipcc
// splits a 256-register budget 128 / 128 as soon as an "a" constraint appears)
constexpr bool S1R_ASM = s1r::BAND == 4;
__device__ __forceinline__ void mma16_v(u32x4 a, u32x4 b, f32x4& c) {
  if constexpr (S1R_ASM) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else c = mma_chunk<bf16>(a, b, c);
}
__device__ __forceinline__ void mma16_a(u32x4 a, u32x4 b, f32x4& c) {
  if constexpr (S1R_ASM) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else c = mma_chunk<bf16>(a, b, c);
}
__device__ __forceinline__ void mma16_a_zero(f32x4& c) {
  if constexpr (S1R_ASM) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %1, 0" : "=a"(c) : "v"(z));
  } else {
    c = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
// `INSN` with N registers threaded through as in/out operands (constraint "+v" or "+a"): everything that produces them is
// scheduled before, everything that consumes them after - how hand-placed wait states are pinned between compiler-scheduled code
#define S1R_T1(c, a, i) c(a[i])
#define S1R_TIE3(insn, c, a) asm volatile(insn : S1R_T1(c, a, 0), S1R_T1(c, a, 1), S1R_T1(c, a, 2))
#define S1R_TIE5(insn, c, a) asm volatile(insn : S1R_T1(c, a, 0), S1R_T1(c, a, 1), S1R_T1(c, a, 2), S1R_T1(c, a, 3), S1R_T1(c, a, 4))
#define S1R_TIE8(insn, c, a) asm volatile(insn : S1R_T1(c, a, 0), S1R_T1(c, a, 1), S1R_T1(c, a, 2), S1R_T1(c, a, 3), S1R_T1(c, a, 4), S1R_T1(c, a, 5), S1R_T1(c, a, 6), S1R_T1(c, a, 7))
#define S1R_TIE4_AT(insn, c, a, o) asm volatile(insn : S1R_T1(c, a, o), S1R_T1(c, a, o + 1), S1R_T1(c, a, o + 2), S1R_T1(c, a, o + 3) :: "memory")
template <int N> __device__ __forceinline__ void s1r_nop_v(f32x4* a, const int kind) {       // kind 0: s_nop 7, 1: s_nop 15 + s_nop 3
  if constexpr (s1r::BAND != 4) return;     // builtin MFMAs: hipcc's hazard recognizer places the wait states
  if (kind == 0) {
    if constexpr (N == 3) S1R_TIE3("s_nop 7", "+v", a);
    else if constexpr (N == 5) S1R_TIE5("s_nop 7", "+v", a);
    else S1R_TIE8("s_nop 7", "+v", a);
  } else {
    if constexpr (N == 3) S1R_TIE3("s_nop 15\n\ts_nop 3", "+v", a);
    else if constexpr (N == 5) S1R_TIE5("s_nop 15\n\ts_nop 3", "+v", a);
    else S1R_TIE8("s_nop 15\n\ts_nop 3", "+v", a);
  }
}
template <int N> __device__ __forceinline__ void s1r_nop_u(u32x4* a) {
  if constexpr (s1r::BAND != 4) return;
  if constexpr (N == 3) S1R_TIE3("s_nop 7", "+v", a);
  else S1R_TIE5("s_nop 7", "+v", a);
}
// s_waitcnt vmcnt(0) with N (a multiple of 4) loaded registers threaded through
template <int N> __device__ __forceinline__ void s1r_wait_loads(u32x4* a) {
  static_assert(N % 4 == 0 && N >= 4, "groups of 4");
  S1R_TIE4_AT("s_waitcnt vmcnt(0)", "+v", a, 0);
#pragma unroll
  for (int o = 4; o < N; o += 4) {
    // (constant offsets after unrolling; the empty asm only anchors the registers behind the wait above)
    asm volatile("" : "+v"(a[o]), "+v"(a[o + 1]), "+v"(a[o + 2]), "+v"(a[o + 3]) :: "memory");
  }
}

__device__ __forceinline__ unsigned s1r_pk2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const bf16x2_t v = {(bf16)a, (bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

}  // namespace

__global__ __launch_bounds__(s1r::NW * 64, s1r::NW / 4) void stage1_rows_kernel(const bf16* __restrict__ X, bf16* __restrict__ Y, const unsigned char* __restrict__ wimg,
                                                             const float* __restrict__ b1, const int n_tasks, const int n_tiles) {
  using namespace s1r;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int m = lane & 15, q = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)(lptrs_t)smem;
  const unsigned voff = (unsigned)(wave * (NW == 4 ? 9216 : 4096) + lane * 16);
  unsigned char* const H1 = smem + OFF_H1 + wave * (4 * PLANE);       // this wave's private patch
  float* const b1tab = reinterpret_cast<float*>(smem + OFF_B1);
  if ((int)blockIdx.x >= n_tiles) return;

  {   // zero the H1 patches once (the border columns / rows are never written afterwards), bias table -> LDS
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = t; i < NW * 4 * PLANE / 16; i += NW * 64) *reinterpret_cast<u32x4*>(smem + OFF_H1 + i * 16) = z;
    if (t < HID) b1tab[t] = b1[t];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }

  // weight ring: slot image n = group n % 8; issued after barrier n-2 (into the slot read during interval n-1 ... n-3), drained
  // (vmcnt(0): nothing newer is in flight at that point) before barrier n-1, first read after barrier n
  int issue_g = 0, issue_slot = 0;
  auto issue = [&]() {
    const unsigned char* src = wimg + (size_t)issue_g * SLOT;
    const unsigned dst = lds0 + issue_slot * SLOT;
    if constexpr (NW == 4) {            // 9 pieces per wave
      s1r_dma4(voff, src, dst + wave * 9216);
      s1r_dma4(voff + 4096u, src, dst + wave * 9216 + 4096u);
      s1r_dma1(voff + 8192u, src, dst + wave * 9216 + 8192u);
    } else {                            // 8 waves: 4 pieces each + pieces 32..35 by waves 0..3 (all waits are vmcnt(0): counts may differ)
      s1r_dma4(voff, src, dst + wave * 4096);
      if (wave < 4) s1r_dma1((unsigned)((32 + wave) * 1024 + lane * 16), src, dst + (32 + wave) * 1024);
    }
    issue_g = issue_g == G - 1 ? 0 : issue_g + 1;
    issue_slot = issue_slot == NST - 1 ? 0 : issue_slot + 1;
  };
  issue();
  if (NST == 3) issue();
  int slot = 0;
  bool first = true;

  // per-lane geometry that does not depend on the task: conv1 token rho = 16 t8 + m of the 6 x 20 region, conv2 token o = 16 t5 + m
  int pixw[MT1];          // H1 pixel of conv1 token (row rr, col c): rr * 22 + c + 1, or -1 for the 8 padding slots
  int rrw[MT1];
#pragma unroll
  for (int t8 = 0; t8 < MT1; ++t8) {
    const int rho = 16 * t8 + m, rr = rho / W, c = rho - rr * W;
    rrw[t8] = rr;
    pixw[t8] = rho < RT ? rr * PW + c + 1 : -1;
  }
  int pix0[MT2];          // top-left tap pixel of conv2 output token o (region row 1 + o / 20): (o / 20) * 22 + o % 20
#pragma unroll
  for (int t5 = 0; t5 < MT2; ++t5) {
    const int o = 16 * t5 + m, ro = o / W;
    pix0[t5] = ro * PW + (o - ro * W);
  }

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    int task = tile * NW + wave;
    const bool tvalid = task < n_tasks;
    task = tvalid ? task : n_tasks - 1;
    const int b = task / NBAND, band = task - b * NBAND;
    const int r0 = band * BAND - 1;                           // image row of region row 0
    const bf16* const xim = X + (size_t)b * (HIMG * W) * C1;

    // ---- conv1 input of the band (+ halo rows) -> registers, natural k order: xr[t8][kc] = channels 32 kc + 8 q .. +7 of token rho
    u32x4 xr[MT1][4];
#pragma unroll
    for (int t8 = 0; t8 < MT1; ++t8) {
      int row = r0 + rrw[t8];
      row = row < 0 ? 0 : (row > HIMG - 1 ? HIMG - 1 : row);             // out-of-image rows: any finite data, their H1 is forced to 0
      const int rho = 16 * t8 + m, c = rho - rrw[t8] * W;
      const bf16* src = xim + (size_t)(row * W + (rho < RT ? c : 0)) * C1 + q * 8;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) xr[t8][kc] = s1r_gload16(src + kc * 32);
    }
    // drain (x loads, the previous task's stores, the ring's DMAs), registers threaded through so no use moves above
    s1r_wait_loads<MT1 * 4>(&xr[0][0]);
    if (first) { s1r_bar(); first = false; }                  // H1 zeros / bias table visible, one barrier between the drain and image 0's reads
    unsigned rowok = 0;                                       // bit t8: conv1 token's image row exists (else its H1 is conv2's zero padding)
#pragma unroll
    for (int t8 = 0; t8 < MT1; ++t8) {
      const int row = r0 + rrw[t8];
      rowok |= (row >= 0 && row < HIMG) ? (1u << t8) : 0u;
    }

    f32x4 yacc[MT2][8];
#pragma unroll
    for (int t5 = 0; t5 < MT2; ++t5)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) mma16_a_zero(yacc[t5][nt]);

#pragma unroll 1
    for (int g = 0; g < G; ++g) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // image g + 1 of this wave has landed (read after the NEXT barrier)
      s1r_bar();
      issue();                                                // image g + 2 -> the slot read during the previous interval
      const unsigned char* sp = smem + slot * SLOT + lane * 16;
      u32x4 fr[FD];

      // One wave per SIMD overlaps nothing by itself: the VALU work of the block (2 x 256 GELUs per token, ~6 VALU issues per MFMA)
      // is interleaved BY HAND with the MFMAs that do not depend on it - GELU + store of conv1's n-tile 0 under the MFMAs of
      // n-tile 1, GELU of conv2's n-tile 0 under the MFMAs of n-tile 1 - and every LDS operand is requested one use ahead of the
      // MFMA that consumes it (in-place rotation: the read for the next tap / fragment is issued right behind the MFMA that
      // read the register).  sched_barrier(0) pins each (MFMAs, VALU slice, reads) group in source order.
      auto h1_store = [&](int nt, int t8, const f32x4& a) {
        // lane holds channels 16 nt + 4 q .. +3 of token (t8, m): plane 2 nt + q / 2, bytes (q & 1) * 8 of the pixel's 16-byte slot
        const f32x2 g0 = S1_GELU2((f32x2{a[0], a[1]})), g1 = S1_GELU2((f32x2{a[2], a[3]}));
        u32x2 o;
        o[0] = s1r_pk2(g0[0], g0[1]);
        o[1] = s1r_pk2(g1[0], g1[1]);
        if (!((rowok >> t8) & 1u)) o = u32x2{0u, 0u};
        if (pixw[t8] >= 0) *reinterpret_cast<u32x2*>(H1 + (2 * nt + (q >> 1)) * PLANE + (q & 1) * 8 + pixw[t8] * 16) = o;
      };
      // ---- conv1 (fragments 4 nt + kc) + GELU -> H1 patch
      {
        f32x4 acc0[MT1], acc1[MT1];
        const f32x4 bias0 = *reinterpret_cast<const f32x4*>(b1tab + g * 32 + q * 4);
        const f32x4 bias1 = *reinterpret_cast<const f32x4*>(b1tab + g * 32 + 16 + q * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) fr[i] = *reinterpret_cast<const u32x4*>(sp + i * 1024);
#pragma unroll
        for (int t8 = 0; t8 < MT1; ++t8) { acc0[t8] = bias0; acc1[t8] = bias1; }
        s1r_nop_v<MT1>(acc0, 0);
        s1r_nop_v<MT1>(acc1, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
          for (int t8 = 0; t8 < MT1; ++t8) mma16_v(fr[kc], xr[t8][kc], acc0[t8]);
          fr[kc] = *reinterpret_cast<const u32x4*>(sp + (4 + kc) * 1024);          // n-tile 1's fragment, one k-chunk ... four uses ahead
          __builtin_amdgcn_sched_barrier(0);
        }
        s1r_nop_v<MT1>(acc0, 1);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int TPK = (MT1 + 3) / 4;                                         // n-tile 0 tiles GELU'd per k-chunk of n-tile 1
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
          for (int t8 = 0; t8 < MT1; ++t8) mma16_v(fr[kc], xr[t8][kc], acc1[t8]);
#pragma unroll
          for (int j = 0; j < TPK; ++j)
            if (kc * TPK + j < MT1) h1_store(0, kc * TPK + j, acc0[kc * TPK + j]);
          __builtin_amdgcn_sched_barrier(0);
        }
        s1r_nop_v<MT1>(acc1, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t8 = 0; t8 < MT1; ++t8) h1_store(1, t8, acc1[t8]);
      }

      // ---- conv2 (9 taps = 9 k-chunks of the 32 group channels) + GELU; fragments 8 + 2 tap + nt.  The packed results of the two
      // n-tiles form conv3's B operand: lane (m, q) holds hidden channels {4 q .. 4 q + 3} (n-tile 0) and {16 + 4 q ..} (n-tile 1)
      u32x4 hp[MT2];
      {
        f32x4 acc0[MT2], acc1[MT2];
        u32x4 hb[MT2];
        const unsigned char* const hrd = H1 + q * PLANE;
        auto tapoff = [](int tap) { return ((tap / 3) * PW + tap % 3) * 16; };
#pragma unroll
        for (int i = 0; i < FD; ++i) fr[i] = *reinterpret_cast<const u32x4*>(sp + (8 + 2 * i) * 1024);
#pragma unroll
        for (int t5 = 0; t5 < MT2; ++t5) hb[t5] = *reinterpret_cast<const u32x4*>(hrd + pix0[t5] * 16 + tapoff(0));
#pragma unroll
        for (int t5 = 0; t5 < MT2; ++t5) { acc0[t5] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[t5] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        s1r_nop_v<MT2>(acc0, 0);
        s1r_nop_v<MT2>(acc1, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {                    // n-tile 0
#pragma unroll
          for (int t5 = 0; t5 < MT2; ++t5) {
            mma16_v(fr[tap % FD], hb[t5], acc0[t5]);
            hb[t5] = *reinterpret_cast<const u32x4*>(hrd + pix0[t5] * 16 + tapoff(tap < 8 ? tap + 1 : 0));     // next tap (after tap 8: tap 0 again, for n-tile 1)
            __builtin_amdgcn_sched_barrier(0);
          }
          fr[tap % FD] = *reinterpret_cast<const u32x4*>(sp + (tap + FD < 9 ? 8 + 2 * (tap + FD) : 8 + 2 * (tap + FD - 9) + 1) * 1024);   // ... then n-tile 1's
          __builtin_amdgcn_sched_barrier(0);
        }
        s1r_nop_v<MT2>(acc0, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {                    // n-tile 1, GELU of n-tile 0 underneath (tile tap / 2 at even taps)
          const int fi = (tap + 9) % FD;
#pragma unroll
          for (int t5 = 0; t5 < MT2; ++t5) {
            mma16_v(fr[fi], hb[t5], acc1[t5]);
            if (tap < 8) hb[t5] = *reinterpret_cast<const u32x4*>(hrd + pix0[t5] * 16 + tapoff(tap + 1));
            __builtin_amdgcn_sched_barrier(0);
          }
          if (tap + FD < 9) fr[fi] = *reinterpret_cast<const u32x4*>(sp + (8 + 2 * (tap + FD) + 1) * 1024);
          if ((tap & 1) == 0 && tap / 2 < MT2) {
            const int t5 = tap / 2;
            const f32x2 g0 = S1_GELU2((f32x2{acc0[t5][0], acc0[t5][1]})), g1 = S1_GELU2((f32x2{acc0[t5][2], acc0[t5][3]}));
            hp[t5][0] = s1r_pk2(g0[0], g0[1]);
            hp[t5][1] = s1r_pk2(g1[0], g1[1]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        s1r_nop_v<MT2>(acc1, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t5 = 0; t5 < MT2; ++t5) {
          const f32x2 g0 = S1_GELU2((f32x2{acc1[t5][0], acc1[t5][1]})), g1 = S1_GELU2((f32x2{acc1[t5][2], acc1[t5][3]}));
          hp[t5][2] = s1r_pk2(g0[0], g0[1]);
          hp[t5][3] = s1r_pk2(g1[0], g1[1]);
        }
      }

      // ---- conv3: all 128 output channels (8 n-tiles), K = this group's 32 hidden channels; fragments 26 + n-tile
#pragma unroll
      for (int i = 0; i < FD; ++i) fr[i] = *reinterpret_cast<const u32x4*>(sp + (26 + i) * 1024);
      s1r_nop_u<MT2>(hp);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
#pragma unroll
        for (int t5 = 0; t5 < MT2; ++t5) mma16_a(fr[nt % FD], hp[t5], yacc[t5][nt]);
        if (nt + FD < 8) fr[nt % FD] = *reinterpret_cast<const u32x4*>(sp + (26 + nt + FD) * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
      slot = slot == NST - 1 ? 0 : slot + 1;
    }

    // ---- epilogue: y = acc + x.  W3's rows are permuted at pack time so that the n-tile pair (2 p, 2 p + 1) gives a lane the 8
    // consecutive channels 32 p + 8 q .. +7 of its token: 16-byte residual loads and stores
    u32x4 res[MT2][4];
    size_t toko[MT2];
#pragma unroll
    for (int t5 = 0; t5 < MT2; ++t5) {
      const int o = 16 * t5 + m, ro = o / W;
      toko[t5] = ((size_t)b * (HIMG * W) + (size_t)(band * BAND + ro) * W + (o - ro * W)) * C1 + q * 8;
#pragma unroll
      for (int p = 0; p < 4; ++p) res[t5][p] = s1r_gload16(X + toko[t5] + 32 * p);
    }
    s1r_wait_loads<MT2 * 4>(&res[0][0]);
#pragma unroll
    for (int t5 = 0; t5 < MT2; ++t5) {
      // wait states MFMA -> v_accvgpr_read for this m-tile's accumulators
      if constexpr (BAND == 4) S1R_TIE8("s_nop 15\n\ts_nop 3", "+a", yacc[t5]);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const bf16x8 r8 = __builtin_bit_cast(bf16x8, res[t5][p]);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (bf16)(yacc[t5][2 * p][e] + (float)r8[e]);
          o[4 + e] = (bf16)(yacc[t5][2 * p + 1][e] + (float)r8[4 + e]);
        }
        if (tvalid) *reinterpret_cast<bf16x8*>(Y + toko[t5] + 32 * p) = o;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Fragment-major weight image from the engine's standard packed layers: w1 [256][128], w2 [8][32][320] (k = tap * 32 + c),
// w3 [128][256], K-contiguous bf16 rows.  One thread per bf16 element; per group 36 fragments of 512 elements (34 used).
__global__ void stage1_pack_kernel(const bf16* __restrict__ w1, const bf16* __restrict__ w2, const bf16* __restrict__ w3, bf16* __restrict__ wimg) {
  using namespace s1r;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G * (SLOT / 2)) return;
  const int g = idx / (SLOT / 2), e = idx % (SLOT / 2);
  const int f = e >> 9, lane = (e >> 3) & 63, e8 = e & 7;
  const int n = lane & 15, kq = lane >> 4;
  bf16 v = (bf16)0.0f;
  if (f < 8) {                    // W1: fragment 4 nt + kc, row = hidden channel 32 g + 16 nt + n, k = input channel 32 kc + 8 kq + e8
    const int nt = f >> 2, kc = f & 3;
    v = w1[(size_t)(g * 32 + nt * 16 + n) * C1 + kc * 32 + kq * 8 + e8];
  } else if (f < 26) {            // W2: fragment 8 + 2 tap + nt, k = the group's input channel 8 kq + e8 of that tap
    const int tap = (f - 8) >> 1, nt = (f - 8) & 1;
    v = w2[(size_t)(g * 32 + nt * 16 + n) * 320 + tap * 32 + kq * 8 + e8];
  } else if (f < 34) {            // W3: fragment 26 + nt; row n of the even / odd tile of pair p = output channel 32 p + 8 (n / 4) + 4 (nt & 1) + n % 4;
    const int nt = f - 26, p = nt >> 1;                         // k slot (kq, e8) = hidden channel 4 kq + e8 (e8 < 4) or 16 + 4 kq + e8 - 4
    const int co = 32 * p + 8 * (n >> 2) + 4 * (nt & 1) + (n & 3);
    const int hc = e8 < 4 ? 4 * kq + e8 : 16 + 4 * kq + (e8 - 4);
    v = w3[(size_t)co * HID + g * 32 + hc];
  }
  wimg[idx] = v;
}

// Opt-in (FSVIT_STAGE1_ROWS=1).  Measured on MI355X (profiles/r01_v9_stage1_rows_pmc.txt, 3200 images): correct (same operator
// test as stage1_fused), 5.66 ms per 64-episode step (5.8 before the hand interleave below) against 5.39 ms for stage1_fused - the design removed the LDS / barrier
// bottleneck (0.4 instead of 1.4 ds_read_b128 per MFMA, no activation barriers) but exposed the next one: the block needs
// 2 x 256 GELUs per token, ~1150 VALU issues per group and wave against 194 MFMAs; with ONE wave per SIMD nothing overlaps them
// (PMC: VALU 44 %, MFMA 23 %, waits 33 % of the wave's cycles), whereas stage1_fused's four waves per SIMD do.  The 2-row-band /
// two-waves-per-SIMD variant (S1R_BAND=2) needs 170 VGPRs + 96 AGPRs of its 256 registers, and hipcc splits the file 128 / 128
// as soon as an "a" constraint appears (83 spills).  Interleaving GELU(conv1 n-tile 0) under conv1 n-tile 1 and GELU(conv2 n-tile 0)
// under conv2 n-tile 1 and requesting every LDS operand one use ahead bought 2.5 %: with a single wave per SIMD every remaining
// latency (first fragments of a phase, x / residual loads per task, barrier skew) is exposed.  Next: a 2-row band with the outputs
// in VGPRs (two waves per SIMD), which trades 25 % more MFMA / GELU work for hardware overlap.
bool stage1_rows_supported(int dtype, int C1, int hid, int group, int H1) {
  static const bool on = [] { const char* e = getenv("FSVIT_STAGE1_ROWS"); return e && e[0] == '1'; }();
  return on && dtype == 1 && C1 == s1r::C1 && hid == s1r::HID && group == s1r::G && H1 == s1r::W;
}
size_t stage1_rows_image_bytes() { return (size_t)s1r::G * s1r::SLOT; }

int launch_stage1_pack(const void* w1, const void* w2, const void* w3, void* wimg, hipStream_t s) {
  const int n = s1r::G * (s1r::SLOT / 2);
  hipLaunchKernelGGL(stage1_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const bf16*)w1, (const bf16*)w2, (const bf16*)w3, (bf16*)wimg);
  return (int)hipGetLastError();
}

int launch_stage1_rows(const void* x, void* y, const void* wimg, const float* b1, int B, hipStream_t s) {
  if (B <= 0) return 0;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)stage1_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, s1r::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const int n_tasks = B * s1r::NBAND, n_tiles = (n_tasks + s1r::NW - 1) / s1r::NW;
  const int grid = n_tiles < 256 ? n_tiles : 256;
  hipLaunchKernelGGL(stage1_rows_kernel, dim3(grid), dim3(s1r::NW * 64), s1r::LDS_BYTES, s, (const bf16*)x, (bf16*)y, (const unsigned char*)wimg, b1,
                     n_tasks, n_tiles);
  return (int)hipGetLastError();
}

}  // namespace fsvit
