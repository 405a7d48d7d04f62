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
// LDS by LDS-DMA, double buffered per part: every chunk is two phases (GEMM1 | GEMM2) with one barrier each; a part is issued one phase after its
// buffer's last read, its counted vmcnt wait sits before the NEXT barrier and its first read after the one behind that (cdna_hip_programming.md:
// "read a staged buffer one phase after the wait that retires it").  Every VMEM instruction inside the chunk loop is issued from inline asm, so the
// counts below are exact:   phase 1 (j): DMA B(j+1) [PW]                                          | wait A(j+1): vmcnt(PW)
//                           phase 2 (j): stores mid(j) [NS], loads g'(j+1) [NL], DMA A(j+2) [PW]   | wait B(j+1): vmcnt(NS + NL + PW)
// The chunk stream is periodic, so the prefetch runs across the tiles of the persistent workgroup.  Rows beyond M are computed on a clamped row and
// STORED (counts stay uniform): every output has room for n_tiles * 128 rows (mlp_train_rows_pad).
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
__device__ __forceinline__ f32x16 mt_mfma(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// The output accumulators (32 rows x C channels: 256 registers at C = 512) are pinned to AGPRs by issuing their MFMAs from inline asm: left to hipcc's
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

template <int C, int HID, int MODE>
__global__ __launch_bounds__(256, 1) void mlp_pair_kernel(const MlpPairArgs a) {
  constexpr int NKS = C / 16, NCT = C / 32, NCH = HID / 32;
  constexpr int PART = NKS * 1024;                 // one chunk of one image: NKS fragments
  constexpr int PW = NKS / MT_NW;                  // LDS-DMA pieces per wave and part
  constexpr int NL = MODE == 1 ? 2 : 0;            // g' loads per chunk
  constexpr int NS = MODE == 1 ? 2 : 4;            // by-product stores per chunk
  static_assert(NCH % 2 == 0 && (PW == 4 || PW == 8), "shapes");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const b1tab = reinterpret_cast<float*>(smem + 4 * PART);
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

  int nextA = 0, nextB = 0;       // next chunk (mod NCH) to issue; the buffer is its parity
  auto issueA = [&]() {
    mt_dma<PW>(voff, a.imgA + (size_t)nextA * PART, lds0 + (nextA & 1) * PART + wave * PW * 1024);
    nextA = nextA == NCH - 1 ? 0 : nextA + 1;
  };
  auto issueB = [&]() {
    mt_dma<PW>(voff, a.imgB + (size_t)nextB * PART, lds0 + (2 + (nextB & 1)) * PART + wave * PW * 1024);
    nextB = nextB == NCH - 1 ? 0 : nextB + 1;
  };
  issueA();
  issueB();
  issueA();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  mt_bar();

  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int m = tile * MT_BM + wave * 32 + r;
    const int mrow = m < a.M ? m : a.M - 1;
    const bf16* xrow = a.X + (size_t)mrow * C + 16 * kh;
    bf16* const hrow = a.H + (size_t)m * HID + 16 * kh;                     // (padded rows exist)
    const bf16* const glrow = a.G + (size_t)mrow * HID + 16 * kh;           // MODE 1: read
    bf16* const gsrow = a.G + (size_t)m * HID + 16 * kh;                    // MODE 0: written
    u32x4 xr[NKS];
    [&]<int... S>(std::integer_sequence<int, S...>) { ((xr[S] = mt_gload16<(32 * (S >> 1) + 8 * (S & 1)) * 2>(xrow)), ...); }(std::make_integer_sequence<int, NKS>{});
    u32x4 gl[2];                  // MODE 1: g' of the next chunk (one buffer: reloaded right behind the multiply that consumed it)
    if constexpr (MODE == 1) {
      gl[0] = mt_gload16<0>(glrow);
      gl[1] = mt_gload16<16>(glrow);
    }
    // one drain per tile: everything older (DMAs, the previous tile's stores) has long landed
#pragma unroll
    for (int s = 0; s < NKS; s += 4) asm volatile("" : "+v"(xr[s]), "+v"(xr[s + 1]), "+v"(xr[s + 2]), "+v"(xr[s + 3]) :: "memory");
    if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(gl[0]), "+v"(gl[1]) :: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < NKS; s += 4) asm volatile("" : "+v"(xr[s]), "+v"(xr[s + 1]), "+v"(xr[s + 2]), "+v"(xr[s + 3]) :: "memory");

    f32x16 yacc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) mt_zero_a(yacc[ct]);

    auto chunk = [&](const int j, auto parity) {
      constexpr int P = decltype(parity)::value;
      const unsigned char* const fa = smem + P * PART + lane * 16;
      const unsigned char* const fb = smem + (2 + P) * PART + lane * 16;
      // ---------------- phase 1: GEMM1 (j)
      mt_bar();
      issueB();
      f32x16 hacc;
      {     // fragments are read one group of 4 ahead of the MFMAs that consume them (left alone, hipcc hoists all NKS reads: 128 VGPRs at C = 512, spills)
        constexpr int FG = C == 512 ? 2 : 4;       // (C = 512 has 128 x + 256 y registers: 16 fragment registers instead of 32)
        u32x4 fr[2][FG];
#pragma unroll
        for (int q = 0; q < FG; ++q) fr[0][q] = *reinterpret_cast<const u32x4*>(fa + q * 1024);
#pragma unroll
        for (int g = 0; g < NKS / FG; ++g) {
          if (g + 1 < NKS / FG) {
#pragma unroll
            for (int q = 0; q < FG; ++q) fr[(g + 1) & 1][q] = *reinterpret_cast<const u32x4*>(fa + (FG * (g + 1) + q) * 1024);
          }
#pragma unroll
          for (int q = 0; q < FG; ++q) {
            if (g == 0 && q == 0) mt_mfma_v_z(fr[0][0], xr[0], hacc);
            else mt_mfma_v(fr[g & 1][q], xr[FG * g + q], hacc);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        mt_settle_v(hacc);
      }
      // A (j+1) has landed (issued a phase ago; behind it in the queue: B (j+1)) - and with it this chunk's g' (older)
      if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(gl[0]), "+v"(gl[1]) : "n"(PW) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PW) : "memory");
      // ---------------- phase 2: mid (j), GEMM2 (j)
      mt_bar();
      u32x4 hp[2];
      if constexpr (MODE == 0) {
        // (one half of the lane's 16 hidden units at a time, each stored as soon as it is packed: the C = 512 variant has 256 - 128 (x) VGPRs for everything)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          float hv[8], dv[8];
#pragma unroll
          for (int i = 0; i < 8; i += 4) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(b1tab + 32 * j + 16 * kh + 8 * hf + i);
            f32x2 d0, d1;
            const f32x2 h0 = gelu_sig2_d(f32x2{hacc[8 * hf + i] + b[0], hacc[8 * hf + i + 1] + b[1]}, d0);
            const f32x2 h1 = gelu_sig2_d(f32x2{hacc[8 * hf + i + 2] + b[2], hacc[8 * hf + i + 3] + b[3]}, d1);
            hv[i] = h0[0]; hv[i + 1] = h0[1]; hv[i + 2] = h1[0]; hv[i + 3] = h1[1];
            dv[i] = d0[0]; dv[i + 1] = d0[1]; dv[i + 2] = d1[0]; dv[i + 3] = d1[1];
          }
          hp[hf] = mt_pack8(hv);
          if (hf == 0) { mt_gstore16<0>(hrow + 32 * j, hp[0]); mt_gstore16<0>(gsrow + 32 * j, mt_pack8(dv)); }
          else { mt_gstore16<16>(hrow + 32 * j, hp[1]); mt_gstore16<16>(gsrow + 32 * j, mt_pack8(dv)); }
        }
      } else {
        float hv[16];
        const bf16x8 g0 = __builtin_bit_cast(bf16x8, gl[0]), g1 = __builtin_bit_cast(bf16x8, gl[1]);
#pragma unroll
        for (int i = 0; i < 8; ++i) { hv[i] = hacc[i] * (float)g0[i]; hv[8 + i] = hacc[8 + i] * (float)g1[i]; }
        hp[0] = mt_pack8(hv);
        hp[1] = mt_pack8(hv + 8);
        mt_gstore16<0>(hrow + 32 * j, hp[0]);
        mt_gstore16<16>(hrow + 32 * j, hp[1]);
        asm volatile("" :: "v"(hp[0]), "v"(hp[1]) : "memory");         // (the multiply above has consumed gl)
        const int jn = j + 1 < NCH ? j + 1 : 0;       // (the last chunk's prefetch is a dummy: counts stay uniform)
        gl[0] = mt_gload16<0>(glrow + 32 * jn);
        gl[1] = mt_gload16<16>(glrow + 32 * jn);
      }
      issueA();
      mt_settle_b(hp[0], hp[1]);
      {
        if constexpr (C == 512) {
          u32x4 fr[2][2];
#pragma unroll
          for (int q = 0; q < 2; ++q) fr[0][q] = *reinterpret_cast<const u32x4*>(fb + q * 1024);
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            if (ct + 1 < NCT) {
#pragma unroll
              for (int q = 0; q < 2; ++q) fr[(ct + 1) & 1][q] = *reinterpret_cast<const u32x4*>(fb + (2 * (ct + 1) + q) * 1024);
            }
            mt_mfma_a(fr[ct & 1][0], hp[0], yacc[ct]);
            mt_mfma_a(fr[ct & 1][1], hp[1], yacc[ct]);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          u32x4 fr[2][4];
#pragma unroll
          for (int q = 0; q < 4; ++q) fr[0][q] = *reinterpret_cast<const u32x4*>(fb + q * 1024);
#pragma unroll
          for (int g = 0; g < NCT / 2; ++g) {
            if (g + 1 < NCT / 2) {
#pragma unroll
              for (int q = 0; q < 4; ++q) fr[(g + 1) & 1][q] = *reinterpret_cast<const u32x4*>(fb + (4 * (g + 1) + q) * 1024);
            }
            // two independent accumulators alternate
            mt_mfma_a(fr[g & 1][0], hp[0], yacc[2 * g]);
            mt_mfma_a(fr[g & 1][2], hp[0], yacc[2 * g + 1]);
            mt_mfma_a(fr[g & 1][1], hp[1], yacc[2 * g]);
            mt_mfma_a(fr[g & 1][3], hp[1], yacc[2 * g + 1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      // B (j+1) has landed (behind it: this chunk's stores, the g' prefetch, A (j+2))
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NS + NL + PW) : "memory");
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
  constexpr size_t lds = (size_t)4 * (C / 16) * 1024 + (size_t)(HID + 2 * C) * 4;
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
