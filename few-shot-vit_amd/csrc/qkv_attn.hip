// Fused qkv conv + attention core of the Visformer stage-2 Attention block (bf16):
//   ctx = softmax((q k^T) * hd^-0.5) v   with   [q | k | v] = qkv( BN(x) )          test_phase/models/visformer.py:172-190
// (the eval BatchNorm of the block is folded into the qkv conv by the weight packer: column scale + bias).
//
// As two launches the block is bound by the qkv tensor: at C = 256 the conv writes 3.4 x its input (864 channels per token,
// 1.1 GB per 6400-image launch) only for the attention kernel to read it back one (image, head) at a time - both launches sit at about
// half of their HBM bounds (0.50 + 0.58 ms).  Here qkv never leaves the chip.  One 8-wave workgroup owns TWO images per pass (waves
// 0-3 / 4-7; S <= 112 tokens, 32 per wave, held in registers as MFMA operands for the whole pass) and walks their 6 heads:
//   * q_h, k_h: D^T = W X^T (v_mfma_f32_16x16x32_bf16, A = weight fragment, B = the wave's tokens): a lane ends up with 4 consecutive
//     channels of one token per 16-channel tile.  q stays in REGISTERS: packed to bf16 the three tiles are already a valid B operand
//     of the score MFMA in the k order (tile 0 rows | tile 1 rows), (tile 2 rows | zeros); k is stored to LDS in exactly that order
//     (8-byte stores, the zero half written once), so q . k is unchanged and no permutation is ever applied;
//     v_h: the same two registers with the operands SWAPPED (D = X W^T): 4 consecutive tokens of one channel -> 8-byte stores into
//     the V^T image.  No transposing pass, no scalar LDS writes;
//   * the attention of the head then runs as in attention_v2_kernel (register softmax, P fed back as the second MFMA's B operand);
//     each wave owns the 32 queries whose q it just computed;
//   * only weights stream: 442 KB per pass (L2-resident), as 18 slot images of 24 fragment-major 1 KB fragments (one per
//     (head, q|k|v)) through a 3-slot LDS ring filled by linear LDS-DMA with counted vmcnt (the ring of mlp_rows.hip), one piece at
//     a time between MFMAs; a fragment feeds the wave's two token tiles;
//   * the next pass's token rows are requested during the last head.
// Two waves per SIMD: one wave's softmax VALU runs under the other's MFMAs.  Padding tokens (rows S..127) carry x = 0, so their
// q / k / v equal the bias: finite, masked as keys, never stored as queries.
#include <stdlib.h>

#include "fsvit_common.h"
#include "kernels.h"

namespace FSVIT_NS {

typedef __attribute__((address_space(3))) void* lptrq_t;

namespace qa {
constexpr int C = 256, HEADS = 6, HDP = 48;
constexpr int NCT = HDP / 16, NKS = C / 32;           // channel tiles of a head part, k-steps of 32 input channels
constexpr int NW = 8, IMGS = 2, WPI = NW / IMGS;       // waves, images per pass, waves per image
constexpr int TT = 2, TOK = WPI * TT * 16;             // token tiles per wave, token rows per image (128)
constexpr int TOKK = 112, NKT = TOKK / 16;             // key rows held / key tiles scored: S <= 112 (keys 112..127 of the last PV chunk carry P = 0)
constexpr int FRAGS = NCT * NKS, SLOT = FRAGS * 1024;  // one (head, part) weight image: 24 KB
constexpr int NST = 4, NIMG = HEADS * 3;
constexpr int PPW = FRAGS / NW;                        // LDS-DMA pieces per wave and slot image
constexpr int KS = 128 + 32;                           // K row stride (bytes): 64 k positions (48 real) + pad.  32 mod 64: a ds_read_b128 is serviced in four
                                                       // 16-lane groups mixing the rows {0-3, 12-15} of one lq with {4-11} of the next (MI355X_MICROARCH.md, LDS):
                                                       // strides of 32 mod 64 bytes are the conflict-free ones, an odd multiple of 16 (144 before) reads 2-way
constexpr int VS = TOK * 2 + 16;                       // V^T row stride
constexpr int OFF_K = NST * SLOT, OFF_V = OFF_K + IMGS * TOKK * KS, OFF_B = OFF_V + IMGS * HDP * VS;
static_assert(OFF_B + 3 * HEADS * HDP * 4 <= 160 * 1024, "LDS budget");
constexpr int LDS_BYTES = OFF_B + 3 * HEADS * HDP * 4;
static_assert(FRAGS % NW == 0, "whole pieces per wave");
}  // namespace qa

namespace {

__device__ __forceinline__ void qa_dma1(unsigned voff, const void* sbase, unsigned lds) {
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
// x rows through inline asm: a compiler-visible global load in the pass loop would make hipcc's waitcnt pass drain the ring
__device__ __forceinline__ u32x4 qa_gload16(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned qa_pk2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) bf16 bf16x2_t;
  const bf16x2_t v = {(bf16)a, (bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

}  // namespace

__global__ __launch_bounds__(qa::NW * 64) void qkv_attn_kernel(const bf16* __restrict__ X, bf16* __restrict__ CTX, const unsigned char* __restrict__ wimg,
                                                               const float* __restrict__ bias, const int B, const int S, const float scale) {
  using namespace qa;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const btab = reinterpret_cast<float*>(smem + OFF_B);
  const unsigned lds0 = (unsigned)(size_t)(lptrq_t)smem;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int sub = wave / WPI, tokbase = (wave % WPI) * (TT * 16);       // image of the pass, first token of this wave
  unsigned char* const Ks = smem + OFF_K + sub * (TOKK * KS);
  unsigned char* const Vt = smem + OFF_V + sub * (HDP * VS);
  const int m = lane & 15, lq = lane >> 4;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const int n_pass = (B + IMGS - 1) / IMGS;
  if ((int)blockIdx.x >= n_pass) return;

  for (int i = t; i < 3 * HEADS * HDP; i += NW * 64) btab[i] = bias ? bias[i] : 0.0f;
  for (int i = t; i < (OFF_B - OFF_K) / 16; i += NW * 64) *reinterpret_cast<u32x4*>(smem + OFF_K + i * 16) = zero4;   // the zero halves of the K rows, V^T columns 112..127
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // weight ring (mlp_rows.hip): slot image n is issued after barrier n-3, waited for before barrier n-1 and first read after barrier
  // n.  Every wave issues PPW pieces per image; vmcnt(PPW) = at most PPW operations outstanding.  Loads return in issue order, so if
  // a piece of image n+1 were still in flight all PPW pieces of image n+2 would be too, plus the piece itself: the wait certifies
  // image n+1 whatever the ctx stores (which may complete out of order with loads) do - they can only make it stricter, which is why
  // they are issued right after the q part's barrier, a whole attention phase ahead of the next wait.
  int issue_img = 0, issue_slot = 0;
  const unsigned voff = (unsigned)(wave * PPW * 1024 + lane * 16);
  auto issue1 = [&](int piece) {
    qa_dma1(voff + piece * 1024, wimg + (size_t)issue_img * SLOT, lds0 + issue_slot * SLOT + (wave * PPW + piece) * 1024);
    if (piece == PPW - 1) {
      issue_img = issue_img == NIMG - 1 ? 0 : issue_img + 1;
      issue_slot = issue_slot == NST - 1 ? 0 : issue_slot + 1;
    }
  };
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
#pragma unroll
    for (int pc = 0; pc < PPW; ++pc) issue1(pc);
  int slot = 0;
  asm volatile("s_barrier" ::: "memory");                  // bias table / zeroed K rows visible

  const float sscale = scale * 1.44269504088896340736f;    // exp runs on v_exp_f32 (2^x): log2(e) folded into the score scale

  // token rows of (pass, this wave) -> registers: xr[tt][ks] = channels 32 ks + 8 lq .. +7 of token tokbase + 16 tt + m (A or B operand alike)
  u32x4 xr[TT][NKS];
  auto load_x = [&](int pass) {
    const int img = pass * IMGS + sub;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int tok = tokbase + 16 * tt + m;
      const bool ok = img < B && tok < S;
      const bf16* src = X + ((size_t)(img < B ? img : B - 1) * S + (tok < S ? tok : S - 1)) * C + lq * 8;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) xr[tt][ks] = ok ? qa_gload16(src + ks * 32) : zero4;
    }
  };
  load_x(blockIdx.x);

  for (int pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
    const int img = pass * IMGS + sub;
    // the rows requested one pass ago (or just above) have landed; registers threaded through so no use moves above the wait
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(xr[0][0]), "+v"(xr[0][1]), "+v"(xr[0][2]), "+v"(xr[0][3]) :: "memory");
    asm volatile("" : "+v"(xr[0][4]), "+v"(xr[0][5]), "+v"(xr[0][6]), "+v"(xr[0][7]) :: "memory");
    asm volatile("" : "+v"(xr[1][0]), "+v"(xr[1][1]), "+v"(xr[1][2]), "+v"(xr[1][3]) :: "memory");
    asm volatile("" : "+v"(xr[1][4]), "+v"(xr[1][5]), "+v"(xr[1][6]), "+v"(xr[1][7]) :: "memory");
    bf16* const obase = CTX + (size_t)img * S * (HEADS * HDP);
    // ctx rows of the previous head, stored one head late (see the ring comment)
    u32x2 pend[TT][NCT];
    int pend_h = -1;
    auto flush_ctx = [&]() {
      if (pend_h < 0) return;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int q = tokbase + 16 * tt + m;
        if (img < B && q < S) {
#pragma unroll
          for (int dt = 0; dt < NCT; ++dt) *reinterpret_cast<u32x2*>(obase + (size_t)q * (HEADS * HDP) + pend_h * HDP + dt * 16 + lq * 4) = pend[tt][dt];
        }
      }
      pend_h = -1;
    };

#pragma unroll 1
    for (int h = 0; h < HEADS; ++h) {
      u32x4 qf[TT][2];
      // ---- q_h (registers), k_h, v_h (LDS images) of the wave's tokens
      // Stream order per head: k, v, q.  The ring barrier in front of the q part also publishes the K / V^T images every wave wrote
      // after its k / v parts (lgkmcnt(0) in front of each barrier), so the attention needs no barrier of its own; the ring barrier in
      // front of the next head's k part is passed only after every wave has finished reading them.
#pragma unroll
      for (int pi = 0; pi < 3; ++pi) {
        const int p = pi == 2 ? 0 : pi + 1;
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(PPW) : "memory");
        asm volatile("s_barrier" ::: "memory");
        if (pi == 2) flush_ctx();                          // the previous head's ctx rows: a q part and an attention phase until the next wait
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sp = smem + slot * SLOT + lane * 16;
        slot = slot == NST - 1 ? 0 : slot + 1;
        const float* bp = btab + (p * HEADS + h) * HDP;
        f32x4 acc[NCT][TT];
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) {
            if (p < 2) acc[c][tt] = *reinterpret_cast<const f32x4*>(bp + 16 * c + 4 * lq);      // rows = channels 16 c + 4 lq + e
            else { const float bv = bp[16 * c + m]; acc[c][tt] = f32x4{bv, bv, bv, bv}; }        // column = channel 16 c + m
          }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
          for (int c = 0; c < NCT; ++c) {
            const u32x4 wf = *reinterpret_cast<const u32x4*>(sp + (ks * NCT + c) * 1024);
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
              acc[c][tt] = p < 2 ? mma_chunk<bf16>(wf, xr[tt][ks], acc[c][tt]) : mma_chunk<bf16>(xr[tt][ks], wf, acc[c][tt]);
          }
          if (ks % 3 == 0 && ks / 3 < PPW) issue1(ks / 3);
        }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          const int tok = tokbase + 16 * tt + m;
          if (p == 0) {
            qf[tt][0] = u32x4{qa_pk2(acc[0][tt][0], acc[0][tt][1]), qa_pk2(acc[0][tt][2], acc[0][tt][3]),
                              qa_pk2(acc[1][tt][0], acc[1][tt][1]), qa_pk2(acc[1][tt][2], acc[1][tt][3])};
            qf[tt][1] = u32x4{qa_pk2(acc[2][tt][0], acc[2][tt][1]), qa_pk2(acc[2][tt][2], acc[2][tt][3]), 0u, 0u};
          } else if (tokbase + 16 * tt >= TOKK) {
            // token tile beyond the 112 key rows kept (never valid for S <= 112): no K / V^T entry
          } else if (p == 1) {      // k positions: chunk 0 = (tile 0 rows 4 lq.. | tile 1 rows 4 lq..) per lq, chunk 1 = (tile 2 rows | zeros)
            // Round 5: the 16-byte column of a row is XORed with bit 2 of the row (KSW): the 8-byte stores of the first version put the 16 rows of a
            // store group on 4 bank positions (row stride 40 dwords = 8 mod 32: 4-way, 21.8 % of the kernel's LDS cycles were conflicts,
            // profiles/r04_mfma_pmc.txt); rows m and m + 4 now sit one column apart, the two tiles of chunk 0 go out as ONE 16-byte store
            // (conflict-free in its 8-lane groups), and the b128 reads of the score product stay conflict-free (bank sets checked in tools/lds_conflicts.py)
            const int ksw = 16 * (lq ^ ((m >> 2) & 1));
            *reinterpret_cast<u32x4*>(Ks + tok * KS + ksw) = u32x4{qa_pk2(acc[0][tt][0], acc[0][tt][1]), qa_pk2(acc[0][tt][2], acc[0][tt][3]),
                                                                   qa_pk2(acc[1][tt][0], acc[1][tt][1]), qa_pk2(acc[1][tt][2], acc[1][tt][3])};
            *reinterpret_cast<u32x2*>(Ks + tok * KS + 64 + ksw) = u32x2{qa_pk2(acc[2][tt][0], acc[2][tt][1]), qa_pk2(acc[2][tt][2], acc[2][tt][3])};
          } else {                  // rows = tokens tokbase + 16 tt + 4 lq + e of channel 16 c + m
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
              const u32x2 o = {qa_pk2(acc[c][tt][0], acc[c][tt][1]), qa_pk2(acc[c][tt][2], acc[c][tt][3])};
              *reinterpret_cast<u32x2*>(Vt + (16 * c + m) * VS + (tokbase + 16 * tt + 4 * lq) * 2) = o;
            }
          }
        }
      }
      if (h == HEADS - 1 && pass + (int)gridDim.x < n_pass) load_x(pass + gridDim.x);     // xr is dead from here on: next pass's rows

      // ---- attention of head h for the wave's two query tiles (attention_v2_kernel's body on the LDS images).  Both tiles run in one
      // straight-line block - padding tiles compute on finite bias-valued rows and are simply not stored - so that the scheduler can
      // put one tile's softmax under the other tile's MFMAs.
      {
        f32x4 sc[TT][NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          const unsigned char* ka = Ks + (kt * 16 + m) * KS + 16 * (lq ^ ((m >> 2) & 1));        // (the store's column swizzle)
          const u32x4 k0 = *reinterpret_cast<const u32x4*>(ka), k1 = *reinterpret_cast<const u32x4*>(ka + 64);
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = mma_chunk<bf16>(k0, qf[tt][0], a);
            sc[tt][kt] = mma_chunk<bf16>(k1, qf[tt][1], a);               // keys kt*16 + lq*4 + r  x  query m
          }
        }
        float inv[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          // The softmax costs as many issue slots as the head's three GEMM parts: keys are masked only in the key tiles that can reach
          // past S (a scalar test per tile), and the score scale rides in the exp's argument (one fma) instead of a multiply per score.
          float mx = -INFINITY;
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt) {
            if (kt * 16 + 16 > S) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (kt * 16 + lq * 4 + r >= S) sc[tt][kt][r] = -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[tt][kt][r]);
          }
          mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          const float mxs = mx * sscale;
          float sum = 0.f;
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float e = __builtin_amdgcn_exp2f(fmaf(sc[tt][kt][r], sscale, -mxs));   // -inf for masked keys -> 0
              sc[tt][kt][r] = e;
              sum += e;
            }
          sum += __shfl_xor(sum, 16, 64);
          sum += __shfl_xor(sum, 32, 64);
          inv[tt] = 1.0f / sum;
        }
        pend_h = h;
        // ctx^T[d][q] = sum_key V^T[d][key] P[q][key]; P in the C layout of S^T is the B operand, the k permutation it implies is
        // applied to the V^T fragment read (two 8-byte reads per 32-key chunk)
        constexpr int NPC = (NKT + 1) / 2;                  // 32-key chunks of the PV product; the last one is half empty
        u32x4 pb[TT][NPC];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
          for (int kc = 0; kc < NPC; ++kc) {
            const bool two = 2 * kc + 1 < NKT;
            pb[tt][kc] = u32x4{qa_pk2(sc[tt][2 * kc][0], sc[tt][2 * kc][1]), qa_pk2(sc[tt][2 * kc][2], sc[tt][2 * kc][3]),
                               two ? qa_pk2(sc[tt][two ? 2 * kc + 1 : 0][0], sc[tt][two ? 2 * kc + 1 : 0][1]) : 0u,
                               two ? qa_pk2(sc[tt][two ? 2 * kc + 1 : 0][2], sc[tt][two ? 2 * kc + 1 : 0][3]) : 0u};
          }
#pragma unroll
        for (int dt = 0; dt < NCT; ++dt) {
          f32x4 a[TT];
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) a[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
          const unsigned char* va = Vt + (dt * 16 + m) * VS;
#pragma unroll
          for (int kc = 0; kc < NPC; ++kc) {
            const u32x2 v0 = *reinterpret_cast<const u32x2*>(va + (32 * kc + lq * 4) * 2);
            const u32x2 v1 = *reinterpret_cast<const u32x2*>(va + (32 * kc + 16 + lq * 4) * 2);
            const u32x4 vf = {v0[0], v0[1], v1[0], v1[1]};
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) a[tt] = mma_chunk<bf16>(vf, pb[tt][kc], a[tt]);
          }
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) {
            const f32x4 o = a[tt] * inv[tt];
            pend[tt][dt] = u32x2{qa_pk2(o[0], o[1]), qa_pk2(o[2], o[3])};
          }
        }
      }
    }
    flush_ctx();                                            // the last head's rows
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // no DMA may be in flight into the LDS of a finished workgroup
}

// Fragment-major weight image from the engine's packed qkv layer (w [3 * HEADS * HDP][kw] bf16, rows (part, head, z)):
//   image 3 h + (k, v, q), fragment ks * NCT + c, lane (m, lq), 8 elements: row (p, h, 16 c + m), columns 32 ks + 8 lq .. +7
__global__ void qkv_attn_pack_kernel(const bf16* __restrict__ w, int kw, bf16* __restrict__ img) {
  using namespace qa;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)NIMG * FRAGS * 512) return;
  const int e8 = (int)idx & 7, lane = (int)(idx >> 3) & 63, f = (int)(idx >> 9) % FRAGS, n = (int)(idx >> 9) / FRAGS;
  const int h = n / 3, p = (n % 3 + 1) % 3, ks = f / NCT, c = f % NCT, m = lane & 15, lq = lane >> 4;      // stream order per head: k, v, q
  img[idx] = w[(size_t)((p * HEADS + h) * HDP + 16 * c + m) * kw + 32 * ks + 8 * lq + e8];
}

bool qkv_attn_supported(int dtype, int C, int heads, int hdp, int S) {
  constexpr int on = 1;
  return on && dtype == 1 && C == qa::C && heads == qa::HEADS && hdp == qa::HDP && S >= 1 && S <= qa::TOKK;
}
size_t qkv_attn_image_bytes() { return (size_t)qa::NIMG * qa::SLOT; }

int launch_qkv_attn_pack(const void* w, int kw, void* img, hipStream_t s) {
  const long n = (long)qa::NIMG * qa::FRAGS * 512;
  hipLaunchKernelGGL(qkv_attn_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16*)w, kw, (bf16*)img);
  return (int)hipGetLastError();
}

int launch_qkv_attn(const void* x, void* ctx, const void* wimg, const float* bias, int B, int S, float scale, hipStream_t s) {
  if (B <= 0) return 0;
  {    // per launch: the attribute is per DEVICE (a process-wide "done" flag skipped it on a second GPU), and the call is cheap
    hipError_t e = hipFuncSetAttribute((const void*)qkv_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, qa::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
  }
  const int n_pass = (B + qa::IMGS - 1) / qa::IMGS;
  const int grid = n_pass < 256 ? n_pass : 256;
  hipLaunchKernelGGL(qkv_attn_kernel, dim3(grid), dim3(qa::NW * 64), qa::LDS_BYTES, s, (const bf16*)x, (bf16*)ctx, (const unsigned char*)wimg, bias, B, S,
                     scale);
  return (int)hipGetLastError();
}

}  // namespace FSVIT_NS
