// Row-wise Mlp of the Visformer stage-2 / stage-3 blocks for the META-TUNING step (meta_tuning_sun_m/train_meta.py:161-177), 16-bit storage:
//   MODE 0 (forward, visformer.py:146-150 + :262, train mode):   out = xa + s[image] * W2 GELU(W1 BN(xa))
//          BN = the block's norm2 with BATCH statistics, folded into W1 after its finalize (w1f = W1 diag(sa) rounded once, b1f = W1 sb;
//          mlp_fold_pack_kernel), s = the DropPath scale of the call.  By-products for the backward pass: h = GELU(z1) and g' = GELU'(z1)
//          ([M][HID] each) and the normalised input xn = sa xa + sb (the weight gradient's operand).
//   MODE 1 (its data gradient):   dh = (dz W2) * g',   dxn = dh W1      (dz = the gradient of the branch output, dh goes to HBM for the two weight gradients)
// Both are  mid = f(X Wa^T);  out = mid Wb^T  with the [M][HID] middle written once and never re-read by this kernel: the two gemm256 launches per
// direction spent most of their time on that round trip (fc1's epilogue writes 2 x 164 MB per stage-2 block at 800 images, fc2 reads it back).
//
// Structure (the row design of mlp_rows.hip without its hand-slotted pipeline - training launches are 30 x smaller than the eval launches it was
// tuned for): a WAVE owns 32 token rows for the whole Mlp; its rows of X sit in registers as MFMA B operands, its 32 x C output accumulators too;
// only weights move.  The hidden dimension is walked in chunks of 32 units; chunk j needs C/16 fragments (1 KiB each: 64 lanes x 16 B, A operands of
// v_mfma_f32_32x32x16: D^T[n][m] = sum_k W[n][k] X[m][k]) of Wa and as many of Wb.  GEMM1's accumulator is GEMM2's B operand: the row order of the Wa
// fragments is chosen so that a lane of the 32 x 32 result holds 16 CONSECUTIVE hidden units of its token (mt_hperm) - packed to 16 bits they are two
// B operands and one 32-byte run of the h / g' / dh row.  The same permutation on Wb's rows gives a lane 16 consecutive output channels per 32-channel
// tile - the layout of the x registers, so the residual comes from the registers that fed GEMM1.
// Weight images (fragment sequences in consumption order, written by pack_weight_multi_kernel modes 3 / 4 and mlp_fold_pack_kernel) stream through
// LDS by LDS-DMA on counted vmcnt waits; every VMEM instruction inside the chunk loop is issued from inline asm, so the counts are exact - the phase
// structure, the issue schedule and the counts are described at the kernel.  The chunk stream is periodic, so the prefetch runs across the tiles of the
// persistent workgroup.  Rows beyond M are computed on a clamped row and STORED (counts stay uniform): every output has room for n_tiles * 128 rows
// (mlp_train_rows_pad).
// Where it stands (DESIGN.md 4b "Round 5", profiles/r05_mlp_train_*): correct, 1.1 GB and 6 launches less per 800-image step at stage 2, and a draw in
// time against the launches it replaces - one wave per SIMD issues its 12 VMEM instructions, ~200 VALU and 32 MFMAs per chunk in order (~4000 cycles
// against 1024 of MFMA time, none of it waiting); stage 3 keeps its gemm256 launches (mlp_train_preferred).
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "fsvit_common.h"
#include "train_kernels.h"

namespace fsvit {

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* mt_lptr_t;

constexpr int MT_NW = 4;          // waves per workgroup, one per SIMD
constexpr int MT_BM = MT_NW * 32; // rows per tile

// MFMA row rho of a fragment carries unit mt_hperm(rho) of its 32-unit block: the D rows a lane holds (8 g + 4 kh + e) are then units 16 kh + 0..15
__host__ __device__ constexpr int mt_hperm(int rho) { return 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); }

// PW consecutive 1 KiB LDS-DMAs of this wave: source = sbase + voff + i * 1024, destination = lds + i * 1024 (+ lane * 16, implicit).  The
// immediate offset moves source and destination together (tools/probes/ldsdma_offset.hip).
template <int PW> __device__ __forceinline__ void mt_dma(unsigned voff, const void* sbase, unsigned lds) {
  static_assert(PW == 4 || PW == 8, "pieces per wave and part");
  unsigned keep;
  if constexpr (PW == 4)
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
        "global_load_lds_dwordx4 %2, %3 offset:2048\n\t"
        "global_load_lds_dwordx4 %2, %3 offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "v"(voff + 4096u), "s"(sbase), "s"(lds), "s"(lds + 4096u)
        : "memory");
}
__device__ __forceinline__ void mt_bar() {
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// (the byte offset is an instruction immediate: one base pointer per row instead of one 64-bit address per access - hipcc hoisted those out of the
// tile loop and spilled them)
template <int OFF = 0> __device__ __forceinline__ u32x4 mt_gload16(const void* p) {
  static_assert(OFF >= 0 && OFF < 4096, "13-bit signed immediate");
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(p), "n"(OFF) : "memory");
  return v;
}
// s_nop 1 behind the store: a VMEM store of more than 8 bytes reads its upper data dwords AFTER issue, and a VALU write to them in the next cycle is
// picked up by the store (the hazard hipcc pads for its own stores and cannot see through inline asm: the data-gradient epilogue's second
// v_cvt_pk_bf16_f32 group landed in the first store's registers - elements 6 .. 9 of every 16 wrong, tests/test_gpu_mlp_train.py)
template <int OFF = 0> __device__ __forceinline__ void mt_gstore16(void* p, u32x4 v) {
  static_assert(OFF >= 0 && OFF < 4096, "13-bit signed immediate");
  asm volatile("global_store_dwordx4 %0, %1, off offset:%2\n\ts_nop 1" :: "v"(p), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ u32x4 mt_pack8(const float* v) {
  const bf16x8 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3], (bf16)v[4], (bf16)v[5], (bf16)v[6], (bf16)v[7]};
  return __builtin_bit_cast(u32x4, o);
}
// Every MFMA is issued from inline asm.  The output accumulators (32 rows x C channels: 256 registers at C = 512) are pinned to AGPRs that way: left to hipcc's
// allocator the C = 512 data-gradient kernel spilled its 128 x registers and reloaded them inside the chunk loop (158 spills).  What the compiler
// no longer does for these instructions is done by hand, as in mlp_rows.hip: wait states VALU-written operand -> MFMA (mt_settle_b) and
// MFMA -> accumulator read (mt_settle_acc); dependent MFMAs on one accumulator issue back to back (same opcode, same vDst).
__device__ __forceinline__ void mt_mfma_a(u32x4 a, u32x4 b, f32x16& c) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// GEMM1's accumulator is pinned to arch VGPRs the same way (the builtin prefers an AGPR destination - with all 256 taken by the outputs at C = 512 it spilled
// an output tile to scratch around every MFMA)
__device__ __forceinline__ void mt_mfma_v(u32x4 a, u32x4 b, f32x16& c) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mt_mfma_v_z(u32x4 a, u32x4 b, f32x16& d) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mt_settle_v(f32x16& c) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(c)); }      // wait states MFMA -> VALU read of its result
__device__ __forceinline__ void mt_zero_a(f32x16& c) {
  const u32x4 z = {0u, 0u, 0u, 0u};
  asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, 0" : "=a"(c) : "v"(z));
}
__device__ __forceinline__ void mt_settle_b(u32x4& b0, u32x4& b1) { asm volatile("s_nop 7" : "+v"(b0), "+v"(b1)); }

}  // namespace

struct MlpPairArgs {
  const bf16* X;                  // [M][C]: MODE 0 the block's raw residual stream xa, MODE 1 dz (gradient of the branch output)
  const unsigned char *imgA, *imgB;
  const float *b1, *sa, *sb;      // MODE 0: folded bias [HID], BatchNorm scale / shift [C] (xn by-product; unused when XN == nullptr)
  const float* scale;             // MODE 0: DropPath scale per image or nullptr
  bf16 *OUT, *XN, *H, *G;         // OUT [Mp][C]; H [Mp][HID] = h (MODE 0) / dh (MODE 1), written; G [Mp][HID] = g' written (MODE 0) / read (MODE 1, [M] rows)
  int M, n_tiles, rows_per_img;
};

// one 1 KiB LDS-DMA piece (64 lanes x 16 B): a VMEM instruction of this size holds the wave's issue stage for ~60 cycles, so inside the GEMM loops the
// refill goes out one piece at a time, two MFMAs apart (mlp_rows.hip)
__device__ __forceinline__ void mt_dma1(unsigned voff, const void* sbase, unsigned lds) {
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

// GELU + derivative of a PAIR of pre-activations (gelu_sig2_d, fsvit_common.h) cut into four micro-steps of 4 .. 7 VALU instructions: one wave per SIMD
// hides about five single-issue instructions behind a 32 x 32 x 16 MFMA (MI355X_MICROARCH.md), so the 8 pairs of a chunk go out one micro-step per
// MFMA gap instead of as a 1100-cycle block between the two GEMMs (first version) or four 270-cycle blocks behind four MFMAs (second: the matrix pipe
// idles through all but 32 cycles of each).
struct MtGelu { f32x2 x, u, p, s, q; };
template <int K> __device__ __forceinline__ void mt_gelu_step(MtGelu& g, f32x2& h, f32x2& d) {
  if constexpr (K == 0) {
    g.u = g.x * g.x;
    g.u[0] = fminf(g.u[0], 64.0f);
    g.u[1] = fminf(g.u[1], 64.0f);
    g.p = g.u * f32x2{1.0153755e-3f, 1.0153755e-3f} + f32x2{-1.0678257e-1f, -1.0678257e-1f};
  } else if constexpr (K == 1) {
    g.p = g.p * g.u + f32x2{-2.3011138f, -2.3011138f};
    const f32x2 z = g.x * g.p;
    g.s[0] = __builtin_amdgcn_exp2f(z[0]);
    g.s[1] = __builtin_amdgcn_exp2f(z[1]);
  } else if constexpr (K == 2) {
    const f32x2 e = g.s + f32x2{1.0f, 1.0f};
    g.s[0] = __builtin_amdgcn_rcpf(e[0]);
    g.s[1] = __builtin_amdgcn_rcpf(e[1]);
    g.q = g.u * f32x2{5.0f * 1.0153755e-3f, 5.0f * 1.0153755e-3f} + f32x2{3.0f * -1.0678257e-1f, 3.0f * -1.0678257e-1f};
  } else {
    g.q = (g.q * g.u + f32x2{-2.3011138f, -2.3011138f}) * f32x2{-0.69314718055994530942f, -0.69314718055994530942f};
    d = g.s * ((g.x * (f32x2{1.0f, 1.0f} - g.s)) * g.q + f32x2{1.0f, 1.0f});
    h = g.x * g.s;
  }
}

// Phase structure (round 5, third version).  Per tile:   P0 | Q(0) R(0) | Q(1) R(1) | ... | Q(N-1) R(N-1),   one barrier in front of each phase:
//   P0   : GEMM1(0)                                                  reads A(0)
//   Q(j) : GEMM1(j+1) on two alternating accumulators (a filler between two MFMAs of ONE dependent chain costs ~43 cycles, between independent ones
//          ~6), with mid(j) - MODE 0: the 32 GELU micro-steps of chunk j, MODE 1: the multiply by g' - , the by-product stores and the B pieces
//          in its MFMA gaps                                           reads A(j+1)
//   R(j) : GEMM2(j), all channel tiles against the first half of the lane's hidden units, then all against the second (consecutive MFMAs never share an
//          accumulator), the A pieces in its gaps                     reads B(j)
// Fragments are read D MFMAs ahead of their use (rolling, D x 32 cycles of cover for the LDS latency).
// Weight streams (periodic - N parts per tile each -, so they run across the tiles of the persistent workgroup), NBUF buffers per stream (4 at C = 256,
// 2 at C = 512: 128 KB).  A part is re-issued in the phase right behind the one that read its buffer: Q(j) issues B(j - 1 + NBUF) behind R(j-1), R(j)
// issues A(j + 1 + NBUF) behind Q(j) - and Q(0) also A(NBUF) behind P0 (first thing, in one burst), so P0 and R(N-1) issue nothing.  VMEM operations per
// phase, in program order:   Q: [Q(0): PW A pieces] PW B pieces, then NS stores + NL g' pieces      R: PW A pieces
// and the counted waits sit right in front of the barrier whose far side reads the part ("nothing orders a ds_read behind a pending LDS-DMA except the
// issuing wave's covering vmcnt, plus a barrier for other waves' reads", MI355X_MICROARCH.md):
//   end of Q(j): B for R(j)     vmcnt(NS + NL + (NBUF-1) (q + r))       end of R(j): A for Q(j+1)   vmcnt((NBUF-1) (q + r))         q = PW + NS + NL, r = PW
// (tools/probes/mlp_train_schedule_sim.py replays the queue: every read sees its part landed, no buffer is re-issued before its read, and a count larger
// by one fails).  Q(0)'s extra pieces only make a wait stricter; R(N-1)'s missing ones sit in front of the full drain (the x loads' vmcnt(0)) at the
// next tile start.
// MODE 1's multipliers g' never pass through registers on their way in: each wave LDS-DMAs its own 32 rows x 32 units of g'(j+2) into one of two 2 KiB
// slots right behind the multiply that read g'(j) from it (an asynchronous VGPR load carried around the loop back edge can be copied by the register
// allocator before it lands - the second version's wrong rows at more than one tile per workgroup), and reads it back lane for lane after its own
// counted vmcnt (its own data: no barrier needed).
template <int C, int HID, int MODE>
__global__ __launch_bounds__(256, 1) void mlp_pair_kernel(const MlpPairArgs a) {
  constexpr int NKS = C / 16, NCT = C / 32, NCH = HID / 32;
  constexpr int PART = NKS * 1024;                 // one chunk of one image: NKS fragments
  constexpr int PW = NKS / MT_NW;                  // LDS-DMA pieces per wave and part
  constexpr int NBUF = C == 256 ? 4 : 2;           // buffers per stream
  constexpr int D = C == 256 ? 6 : 4;              // fragments read ahead
  constexpr int NL = MODE == 1 ? 2 : 0;            // g' pieces per chunk
  constexpr int NS = MODE == 1 ? 2 : 4;            // by-product stores per chunk
  constexpr int QOPS = PW + NS + NL, ROPS = PW;
  constexpr int WAIT_Q = NS + NL + (NBUF - 1) * (QOPS + ROPS);
  constexpr int WAIT_R = (NBUF - 1) * (QOPS + ROPS);
  constexpr int WAIT_G = 2 * ROPS + QOPS + PW;     // MODE 1, in front of g'(j)'s read in Q(j): behind its pieces (the last operations of Q(j-2)) came R(j-2), Q(j-1), R(j-1) and Q(j)'s B pieces
  constexpr int DSTEP = 2;                         // MFMA slots between two DMA pieces
  constexpr int GOFF = 2 * NBUF * PART;            // MODE 0: tables; MODE 1: the g' slots [2][4 waves][2 KiB]
  static_assert(NCH % 2 == 0 && (PW == 4 || PW == 8) && WAIT_Q < 64 && WAIT_R < 64 && WAIT_G < 64 && DSTEP * PW <= NKS / 2, "shapes");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const b1tab = reinterpret_cast<float*>(smem + GOFF);
  float* const satab = b1tab + HID;
  float* const sbtab = satab + C;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, kh = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)(mt_lptr_t)smem;
  const unsigned voff = (unsigned)(wave * PW * 1024 + lane * 16);
  if ((int)blockIdx.x >= a.n_tiles) return;

  if constexpr (MODE == 0) {
    for (int i = t; i < HID; i += 256) b1tab[i] = a.b1 ? a.b1[i] : 0.f;
    if (a.XN)
      for (int i = t; i < C; i += 256) { satab[i] = a.sa[i]; sbtab[i] = a.sb[i]; }
  }

  // stream state (wave-uniform): content index and buffer of the next part to issue / to read
  int aIssC = 0, aIssB = 0, bIssC = 0, bIssB = 0, aRdB = 0, bRdB = 0;
  auto a_src = [&]() { return a.imgA + (size_t)aIssC * PART; };
  auto a_dst = [&]() { return lds0 + aIssB * PART + wave * PW * 1024; };
  auto a_adv = [&]() { aIssC = aIssC == NCH - 1 ? 0 : aIssC + 1; aIssB = aIssB == NBUF - 1 ? 0 : aIssB + 1; };
  auto b_src = [&]() { return a.imgB + (size_t)bIssC * PART; };
  auto b_dst = [&]() { return lds0 + (NBUF + bIssB) * PART + wave * PW * 1024; };
  auto b_adv = [&]() { bIssC = bIssC == NCH - 1 ? 0 : bIssC + 1; bIssB = bIssB == NBUF - 1 ? 0 : bIssB + 1; };
  auto issueA_all = [&]() { mt_dma<PW>(voff, a_src(), a_dst()); a_adv(); };
  auto issueB_all = [&]() { mt_dma<PW>(voff, b_src(), b_dst()); b_adv(); };
  auto issueA_piece = [&](int pc) { mt_dma1(voff + pc * 1024, a_src(), a_dst() + pc * 1024); if (pc == PW - 1) a_adv(); };
  auto issueB_piece = [&](int pc) { mt_dma1(voff + pc * 1024, b_src(), b_dst() + pc * 1024); if (pc == PW - 1) b_adv(); };
#pragma unroll
  for (int i = 0; i < NBUF; ++i) issueA_all();
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i) issueB_all();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  mt_bar();

  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int m = tile * MT_BM + wave * 32 + r;
    const int mrow = m < a.M ? m : a.M - 1;
    const bf16* xrow = a.X + (size_t)mrow * C + 16 * kh;
    bf16* hrow = a.H + (size_t)m * HID + 16 * kh;                     // (padded rows exist)
    bf16* gsrow = a.G + (size_t)m * HID + 16 * kh;                    // MODE 0: written
    // MODE 1: this lane's g' run of chunk j, half h = a.G + goff + (32 j + 8 h) elements; slot sl of this wave at smem + GOFF + (sl * 4 + wave) * 2048
    const unsigned goff = (unsigned)(((size_t)mrow * HID + 16 * kh) * 2);
    auto issueG = [&](const int j, const int sl) {
      const unsigned dst = lds0 + GOFF + (sl * MT_NW + wave) * 2048;
      mt_dma1(goff, a.G + 32 * j, dst);
      mt_dma1(goff, a.G + 32 * j + 8, dst + 1024);
    };
    u32x4 xr[NKS];
    [&]<int... S>(std::integer_sequence<int, S...>) { ((xr[S] = mt_gload16<(32 * (S >> 1) + 8 * (S & 1)) * 2>(xrow)), ...); }(std::make_integer_sequence<int, NKS>{});
    if constexpr (MODE == 1) { issueG(0, 0); issueG(1, 1); }
    // one drain per tile: everything older (DMAs, the previous tile's stores) has long landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < NKS; s += 4) asm volatile("" : "+v"(xr[s]), "+v"(xr[s + 1]), "+v"(xr[s + 2]), "+v"(xr[s + 3]) :: "memory");
    mt_bar();                     // (parts issued by every wave: wait, barrier, P0's barrier, read)

    f32x16 yacc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) mt_zero_a(yacc[ct]);

    f32x16 hacc[2];
    u32x4 hp[2];
    // GEMM1 of one chunk into `acc` (+ `acc2`, summed at the end) from the A buffer being read; `slot(s)` runs behind MFMA s
    auto gemm1 = [&](f32x16& acc, auto&& slot) {
      const unsigned char* const fa = smem + aRdB * PART + lane * 16;
      u32x4 fr[D];
      f32x16 acc2;
#pragma unroll
      for (int q = 0; q < D; ++q) fr[q] = *reinterpret_cast<const u32x4*>(fa + q * 1024);
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        if (s == 0) mt_mfma_v_z(fr[0], xr[0], acc);
        else if (s == 1) mt_mfma_v_z(fr[1 % D], xr[1], acc2);
        else if (s & 1) mt_mfma_v(fr[s % D], xr[s], acc2);
        else mt_mfma_v(fr[s % D], xr[s], acc);
        if (s + D < NKS) fr[s % D] = *reinterpret_cast<const u32x4*>(fa + (s + D) * 1024);
        slot(s);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc), "+v"(acc2));      // wait states MFMA -> VALU read
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
      aRdB = aRdB == NBUF - 1 ? 0 : aRdB + 1;
    };
    // ---- mid (j) in the gaps of Q (j)
    MtGelu gs[8];
    f32x2 gh[8], gd[8];
    // MODE 0: micro-step tt (0 .. 31) of chunk j: pair tt / 4 (hidden units 2 (tt / 4), + 1 of the lane's 16), step tt % 4; a half is packed and stored behind its last step
    auto gelu_ms = [&](const f32x16& hc, const int j, auto ttc) {
      constexpr int tt = decltype(ttc)::value, pr = tt >> 2, k = tt & 3;
      if constexpr (k == 0) {
        const f32x2 bq = *reinterpret_cast<const f32x2*>(b1tab + 32 * j + 16 * kh + 2 * pr);
        gs[pr].x = f32x2{hc[2 * pr] + bq[0], hc[2 * pr + 1] + bq[1]};
      }
      mt_gelu_step<k>(gs[pr], gh[pr], gd[pr]);
      if constexpr (tt == 15 || tt == 31) {
        constexpr int hf = tt == 31, p0 = 4 * hf;
        const float hv[8] = {gh[p0][0], gh[p0][1], gh[p0 + 1][0], gh[p0 + 1][1], gh[p0 + 2][0], gh[p0 + 2][1], gh[p0 + 3][0], gh[p0 + 3][1]};
        const float dv[8] = {gd[p0][0], gd[p0][1], gd[p0 + 1][0], gd[p0 + 1][1], gd[p0 + 2][0], gd[p0 + 2][1], gd[p0 + 3][0], gd[p0 + 3][1]};
        hp[hf] = mt_pack8(hv);
        mt_gstore16<16 * hf>(hrow + 32 * j, hp[hf]);
        mt_gstore16<16 * hf>(gsrow + 32 * j, mt_pack8(dv));
      }
    };
    // MODE 1: half hf of chunk j: dh = hacc * g' (g' from this wave's LDS slot), packed and stored; behind the second half the slot is refilled with g'(j+2)
    auto mul_half = [&](const f32x16& hc, const int j, auto hfc) {
      constexpr int hf = decltype(hfc)::value;
      const unsigned char* const gsl = smem + GOFF + ((j & 1) * MT_NW + wave) * 2048 + lane * 16;
      if constexpr (hf == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAIT_G) : "memory");
      const bf16x8 g8 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(gsl + hf * 1024));
      float hv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) hv[i] = hc[8 * hf + i] * (float)g8[i];
      hp[hf] = mt_pack8(hv);
      mt_gstore16<16 * hf>(hrow + 32 * j, hp[hf]);
      if constexpr (hf == 1) {
        asm volatile("" :: "v"(hp[0]), "v"(hp[1]) : "memory");         // (both halves of the slot have been read: the multiplies above consumed them)
        issueG(j + 2 < NCH ? j + 2 : j + 2 - NCH, j & 1);              // (the last two refills are dummies: counts stay uniform)
      }
    };
    // what runs behind MFMA slot s of Q(j)'s GEMM1 (for the last chunk: on its own): the B pieces in the first half (DSTEP slots apart; every store /
    // g' piece of the phase is issued behind the last of them - the order the wait counts assume), the steps of mid (j) spread over all slots
    auto q_slot = [&](const f32x16& hc, const int j, auto sc) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s % DSTEP == 0 && s / DSTEP < PW) issueB_piece(s / DSTEP);
      if constexpr (MODE == 0) {
        constexpr int MS = 32 / NKS;          // micro-steps per slot: 2 at C = 256, 1 at C = 512
        gelu_ms(hc, j, std::integral_constant<int, MS * s>{});
        if constexpr (MS == 2) gelu_ms(hc, j, std::integral_constant<int, MS * s + 1>{});
      } else {
        if constexpr (s == NKS / 2) mul_half(hc, j, std::integral_constant<int, 0>{});
        if constexpr (s == 3 * NKS / 4) mul_half(hc, j, std::integral_constant<int, 1>{});
      }
    };
    auto q_slots_alone = [&](const f32x16& hc, const int j) {
      [&]<int... S>(std::integer_sequence<int, S...>) { ((q_slot(hc, j, std::integral_constant<int, S>{}), __builtin_amdgcn_sched_barrier(0)), ...); }(std::make_integer_sequence<int, NKS>{});
    };

    // ---------------- P0: GEMM1 (0)   (everything in flight was drained at the tile start; nothing is issued here)
    mt_bar();
    gemm1(hacc[0], [&](int) {});

    auto chunk = [&](const int j, auto parity) {
      constexpr int P = decltype(parity)::value;
      // ---------------- Q (j): GEMM1 (j+1) -> hacc[1 - P]  ||  mid (j) on hacc[P]
      mt_bar();
      if (j == 0) issueA_all();                          // the buffer P0 read
      if (j + 1 < NCH) {
        const unsigned char* const fa = smem + aRdB * PART + lane * 16;
        f32x16& acc = hacc[1 - P];
        u32x4 fr[D];
        f32x16 acc2;
#pragma unroll
        for (int q = 0; q < D; ++q) fr[q] = *reinterpret_cast<const u32x4*>(fa + q * 1024);
        [&]<int... S>(std::integer_sequence<int, S...>) {
          (([&] {
             if constexpr (S == 0) mt_mfma_v_z(fr[0], xr[0], acc);
             else if constexpr (S == 1) mt_mfma_v_z(fr[1 % D], xr[1], acc2);
             else if constexpr (S & 1) mt_mfma_v(fr[S % D], xr[S], acc2);
             else mt_mfma_v(fr[S % D], xr[S], acc);
             if constexpr (S + D < NKS) fr[S % D] = *reinterpret_cast<const u32x4*>(fa + (S + D) * 1024);
             q_slot(hacc[P], j, std::integral_constant<int, S>{});
             __builtin_amdgcn_sched_barrier(0);
           }()),
           ...);
        }(std::make_integer_sequence<int, NKS>{});
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc), "+v"(acc2));      // wait states MFMA -> VALU read
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
        aRdB = aRdB == NBUF - 1 ? 0 : aRdB + 1;
      } else {
        q_slots_alone(hacc[P], j);
      }
      mt_settle_b(hp[0], hp[1]);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAIT_Q) : "memory");
      // ---------------- R (j): GEMM2 (j)
      mt_bar();
      {
        const unsigned char* const fb = smem + (NBUF + bRdB) * PART + lane * 16;
        // iteration order (half s2 outer, channel tile ct inner); the image holds fragment (ct, s2) at index 2 ct + s2
        auto fidx = [](int it) { return 2 * (it % NCT) + it / NCT; };
        u32x4 fr[D];
#pragma unroll
        for (int q = 0; q < D; ++q) fr[q] = *reinterpret_cast<const u32x4*>(fb + fidx(q) * 1024);
#pragma unroll
        for (int it = 0; it < NKS; ++it) {
          mt_mfma_a(fr[it % D], hp[it / NCT], yacc[it % NCT]);
          if (it + D < NKS) fr[it % D] = *reinterpret_cast<const u32x4*>(fb + fidx(it + D) * 1024);
          if (j + 1 < NCH && it % DSTEP == 0 && it / DSTEP < PW) issueA_piece(it / DSTEP);
          __builtin_amdgcn_sched_barrier(0);
        }
        bRdB = bRdB == NBUF - 1 ? 0 : bRdB + 1;
      }
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAIT_R) : "memory");
    };
    for (int j = 0; j < NCH; j += 2) {
      chunk(j, std::integral_constant<int, 0>{});
      chunk(j + 1, std::integral_constant<int, 1>{});
    }

    // ---------------- tile epilogue
    {     // wait states MFMA -> v_accvgpr_read, accumulators threaded through
      static_assert(NCT == 8 || NCT == 16, "operand list");
      if constexpr (NCT == 16)
        asm volatile("s_nop 15\n\ts_nop 3"
                     : "+a"(yacc[0]), "+a"(yacc[1]), "+a"(yacc[2]), "+a"(yacc[3]), "+a"(yacc[4]), "+a"(yacc[5]), "+a"(yacc[6]), "+a"(yacc[7]), "+a"(yacc[NCT - 8]),
                       "+a"(yacc[NCT - 7]), "+a"(yacc[NCT - 6]), "+a"(yacc[NCT - 5]), "+a"(yacc[NCT - 4]), "+a"(yacc[NCT - 3]), "+a"(yacc[NCT - 2]), "+a"(yacc[NCT - 1]));
      else
        asm volatile("s_nop 15\n\ts_nop 3"
                     : "+a"(yacc[0]), "+a"(yacc[1]), "+a"(yacc[2]), "+a"(yacc[3]), "+a"(yacc[4]), "+a"(yacc[5]), "+a"(yacc[6]), "+a"(yacc[7]));
    }
    bf16* orow = a.OUT + (size_t)m * C + 16 * kh;
    asm volatile("" : "+v"(orow));              // (opaque: the addresses are formed here, from one pointer and immediates)
    if constexpr (MODE == 0) {
      const float sc = a.scale ? a.scale[mrow / a.rows_per_img] : 1.0f;
      bf16* nrow = a.XN ? a.XN + (size_t)m * C + 16 * kh : nullptr;
      asm volatile("" : "+v"(nrow));
      const float *sat = satab + 16 * kh, *sbt = sbtab + 16 * kh;
      asm volatile("" : "+v"(sat), "+v"(sbt));
      auto tile_out = [&](auto ctc) {
        constexpr int ct = decltype(ctc)::value;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const bf16x8 xv = __builtin_bit_cast(bf16x8, xr[2 * ct + hf]);
          float o[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] = fmaf(sc, yacc[ct][8 * hf + q], (float)xv[q]);
          if (hf == 0) mt_gstore16<64 * ct>(orow, mt_pack8(o));
          else mt_gstore16<64 * ct + 16>(orow, mt_pack8(o));
          if (nrow) {
            const int c0 = 32 * ct + 8 * hf;
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(sat + c0), s1 = *reinterpret_cast<const f32x4*>(sat + c0 + 4);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(sbt + c0), t1 = *reinterpret_cast<const f32x4*>(sbt + c0 + 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) { o[q] = fmaf((float)xv[q], s0[q], t0[q]); o[4 + q] = fmaf((float)xv[4 + q], s1[q], t1[q]); }
            if (hf == 0) mt_gstore16<64 * ct>(nrow, mt_pack8(o));
            else mt_gstore16<64 * ct + 16>(nrow, mt_pack8(o));
          }
        }
      };
      [&]<int... CT>(std::integer_sequence<int, CT...>) { (tile_out(std::integral_constant<int, CT>{}), ...); }(std::make_integer_sequence<int, NCT>{});
    } else {
      auto tile_out = [&](auto ctc) {
        constexpr int ct = decltype(ctc)::value;
        float o[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = yacc[ct][i];
        mt_gstore16<64 * ct>(orow, mt_pack8(o));
        mt_gstore16<64 * ct + 16>(orow, mt_pack8(o + 8));
      };
      [&]<int... CT>(std::integer_sequence<int, CT...>) { (tile_out(std::integral_constant<int, CT>{}), ...); }(std::make_integer_sequence<int, NCT>{});
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the wrapped-around prefetches of the last tile land before the LDS allocation goes away
}

// imgA of the forward: fragments of W1 diag(sa) (rows = hidden units, k = channels; same element order as pack mode 3) and the folded bias W1 sb.
// One block per 32-unit chunk.
template <int C>
__global__ __launch_bounds__(256) void mlp_fold_pack_kernel(const float* __restrict__ W, const float* __restrict__ sa, const float* __restrict__ sb,
                                                            bf16* __restrict__ img, float* __restrict__ bf) {
  constexpr int NKS = C / 16;
  const int j = blockIdx.x, t = threadIdx.x;
  for (int piece = t; piece < NKS * 64; piece += 256) {
    const int s = piece >> 6, lane = piece & 63, r = lane & 31, kh = lane >> 5;
    const int row = 32 * j + mt_hperm(r), k0 = 32 * (s >> 1) + 16 * kh + 8 * (s & 1);
    const float* w = W + (size_t)row * C + k0;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = w[q] * sa[k0 + q];
    *reinterpret_cast<u32x4*>(img + ((size_t)j * NKS * 64 + piece) * 8) = mt_pack8(v);
  }
  // bias: 8 threads per row
  const int row = 32 * j + (t >> 3), sub = t & 7;
  float acc = 0.f;
  for (int c = sub; c < C; c += 8) acc += W[(size_t)row * C + c] * sb[c];
  acc += __shfl_xor(acc, 1);
  acc += __shfl_xor(acc, 2);
  acc += __shfl_xor(acc, 4);
  if (sub == 0) bf[row] = acc;
}

bool mlp_train_supported(int dtype, int C, int hid) { return dtype == 1 && ((C == 256 && hid == 1024) || (C == 512 && hid == 2048)); }
// Which shapes the trainer routes through this kernel.  Measured at 800 images (profiles/r05_mlp_train_*): stage 2 (C = 256, 80 000 rows, 625 tiles) 205 us
// forward / 155 us data gradient against 190 / 147 us for the launches it replaces - a draw that removes 164 MB of re-reads and 3 launches per block;
// stage 3 (C = 512, 20 000 rows = 157 tiles on 256 CUs, 4 MB of weight images against the 4 MB L2 of an XCD) 180 / 135 us against 145 / 100 us - a
// loss, so stage 3 keeps its two gemm256 launches unless FSVIT_MLP_TRAIN_FUSED=2 asks for the kernel at both stages.
bool mlp_train_preferred(int C, int hid) {
  static const int mode = [] { const char* e = getenv("FSVIT_MLP_TRAIN_FUSED"); return e ? atoi(e) : 1; }();
  return mode == 2 || (mode == 1 && C == 256 && hid == 1024);
}
int mlp_train_rows_pad(int M) { return (M + MT_BM - 1) / MT_BM * MT_BM; }
size_t mlp_train_image_bytes(int C, int hid) { return (size_t)C * hid * 2; }

int launch_mlp_fold_pack(const float* W1, const float* sa, const float* sb, void* imgA, float* b1f, int C, int hid, hipStream_t s) {
  if (C == 256) hipLaunchKernelGGL(mlp_fold_pack_kernel<256>, dim3(hid / 32), dim3(256), 0, s, W1, sa, sb, (bf16*)imgA, b1f);
  else if (C == 512) hipLaunchKernelGGL(mlp_fold_pack_kernel<512>, dim3(hid / 32), dim3(256), 0, s, W1, sa, sb, (bf16*)imgA, b1f);
  else return -1;
  return (int)hipGetLastError();
}

template <int C, int HID, int MODE>
static int mlp_pair_launch(const MlpPairArgs& a, hipStream_t s) {
  constexpr size_t lds = (size_t)2 * (C == 256 ? 4 : 2) * (C / 16) * 1024 + (MODE == 0 ? (size_t)(HID + 2 * C) * 4 : (size_t)2 * MT_NW * 2048);
  static_assert(lds <= 160 * 1024, "LDS");
  static bool once = false;
  if (!once) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_pair_kernel<C, HID, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    once = true;
  }
  static const int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  const int grid = a.n_tiles < ncu ? a.n_tiles : ncu;
  hipLaunchKernelGGL((mlp_pair_kernel<C, HID, MODE>), dim3(grid), dim3(256), lds, s, a);
  return (int)hipGetLastError();
}

// forward: out / xn [Mp][C], h / g [Mp][hid] with Mp = mlp_train_rows_pad(M); scale / xn may be null
int launch_mlp_train_fwd(const void* xa, const void* imgA, const float* b1f, const void* imgB, const float* sa, const float* sb, const float* scale, int rows_per_img,
                         void* out, void* xn, void* h, void* g, int M, int C, int hid, hipStream_t s) {
  MlpPairArgs a{(const bf16*)xa, (const unsigned char*)imgA, (const unsigned char*)imgB, b1f, sa, sb, scale, (bf16*)out, (bf16*)xn, (bf16*)h, (bf16*)g, M,
                (M + MT_BM - 1) / MT_BM, rows_per_img > 0 ? rows_per_img : 1};
  if (C == 256 && hid == 1024) return mlp_pair_launch<256, 1024, 0>(a, s);
  if (C == 512 && hid == 2048) return mlp_pair_launch<512, 2048, 0>(a, s);
  return -1;
}
// data gradient: dz [M][C], g [M][hid] -> dh [Mp][hid], dxn [Mp][C]
int launch_mlp_train_bwd(const void* dz, const void* imgA, const void* imgB, const void* g, void* dh, void* dxn, int M, int C, int hid, hipStream_t s) {
  MlpPairArgs a{(const bf16*)dz, (const unsigned char*)imgA, (const unsigned char*)imgB, nullptr, nullptr, nullptr, nullptr, (bf16*)dxn, nullptr, (bf16*)dh,
                (bf16*)const_cast<void*>(g), M, (M + MT_BM - 1) / MT_BM, 1};
  if (C == 256 && hid == 1024) return mlp_pair_launch<256, 1024, 1>(a, s);
  if (C == 512 && hid == 2048) return mlp_pair_launch<512, 2048, 1>(a, s);
  return -1;
}

}  // namespace fsvit
