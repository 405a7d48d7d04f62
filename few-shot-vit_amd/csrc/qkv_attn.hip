// Fused qkv conv + attention core of the Visformer stage-2 Attention block (bf16):
//   ctx = softmax((q k^T) * hd^-0.5) v   with   [q | k | v] = qkv( BN(x) )          test_phase/models/visformer.py:172-190
// (the eval BatchNorm of the block is folded into the qkv conv by the weight packer: column scale + bias).
//
// As two launches the block is bound by the qkv tensor: at C = 256 the conv writes 3.4 x its input (864 channels per token,
// 1.1 GB per 6400-image launch) only for the attention kernel to read it back one (image, head) at a time - both launches sit at about
// half of their HBM bounds (0.50 + 0.58 ms).  Here qkv never leaves the chip.  One 8-wave workgroup owns an IMAGE (S <= 128
// tokens, 16 per wave, held in registers as MFMA operands for the whole image) and walks its 6 heads:
//   * q_h, k_h: D^T = W X^T (v_mfma_f32_16x16x32_bf16, A = weight fragment, B = the wave's tokens): a lane ends up with 4 consecutive
//     channels of one token -> 8-byte stores into the row-major Q / K images in LDS;
//     v_h: the same two registers with the operands SWAPPED (D = X W^T): 4 consecutive tokens of one channel -> 8-byte stores into
//     the V^T image.  No transposing pass, no scalar LDS writes;
//   * the attention of the head then runs on those images exactly as attention_v2_kernel does on its staged copies (register
//     softmax, P fed back as the second MFMA's B operand); each wave owns the 16 queries whose q it just computed;
//   * only weights stream: 442 KB per image (L2-resident), as 18 slot images of 24 fragment-major 1 KB fragments (one per
//     (head, q|k|v)) through a 4-slot LDS ring filled by linear LDS-DMA with counted vmcnt (the ring of mlp_rows.hip), one piece at
//     a time between MFMAs.
// Two waves per SIMD: one wave's softmax VALU runs under the other's MFMAs.  Padding tokens (rows S..127) carry x = 0, so their
// q / k / v equal the bias: finite, masked as keys, never stored as queries.
#include <stdlib.h>

#include "fsvit_common.h"
#include "kernels.h"

namespace fsvit {

typedef __attribute__((address_space(3))) void* lptrq_t;

namespace qa {
constexpr int C = 256, HEADS = 6, HDP = 48;
constexpr int NCT = HDP / 16, NKS = C / 32;           // channel tiles of a head part, k-steps of 32 input channels
constexpr int NW = 8, TOK = NW * 16;                   // waves, token rows per workgroup
constexpr int NKT = TOK / 16;                          // key tiles
constexpr int FRAGS = NCT * NKS, SLOT = FRAGS * 1024;  // one (head, part) weight image: 24 KB
constexpr int NST = 4, NIMG = HEADS * 3;
constexpr int PPW = FRAGS / NW;                        // LDS-DMA pieces per wave and slot image
constexpr int QS = HDP * 2 + 16;                       // Q / K row stride (bytes): odd multiple of 16
constexpr int VS = TOK * 2 + 16;                       // V^T row stride
constexpr int OFF_Q = NST * SLOT, OFF_K = OFF_Q + TOK * QS, OFF_V = OFF_K + TOK * QS, OFF_B = OFF_V + HDP * VS;
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
// x rows through inline asm: a compiler-visible global load in the image loop would make hipcc's waitcnt pass drain the ring
__device__ __forceinline__ u32x4 qa_gload16(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void qa_lds_barrier() {      // LDS writes of this wave done, then the workgroup barrier (no vmcnt: the ring stays in flight)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace

__global__ __launch_bounds__(qa::NW * 64) void qkv_attn_kernel(const bf16* __restrict__ X, bf16* __restrict__ CTX, const unsigned char* __restrict__ wimg,
                                                               const float* __restrict__ bias, const int B, const int S, const float scale) {
  using namespace qa;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Qs = smem + OFF_Q;
  unsigned char* const Ks = smem + OFF_K;
  unsigned char* const Vt = smem + OFF_V;
  float* const btab = reinterpret_cast<float*>(smem + OFF_B);
  const unsigned lds0 = (unsigned)(size_t)(lptrq_t)smem;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int m = lane & 15, lq = lane >> 4;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  if ((int)blockIdx.x >= B) return;

  for (int i = t; i < 3 * HEADS * HDP; i += NW * 64) btab[i] = bias ? bias[i] : 0.0f;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // weight ring (mlp_rows.hip): slot image n is issued after barrier n-3, waited for (this wave's counted vmcnt) before barrier n-1 and
  // first read after barrier n.  Every wave issues PPW pieces per image; vmcnt(PPW) = everything but the newest PPW operations has
  // landed (the ctx stores of the attention phase share the queue and only make the wait stricter).
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
  asm volatile("s_barrier" ::: "memory");                  // bias table visible

  const int tok = wave * 16 + m;                           // this lane's token as an MFMA row / column
  const float sscale = scale * 1.44269504088896340736f;    // exp runs on v_exp_f32 (2^x): log2(e) folded into the score scale

  for (int img = blockIdx.x; img < B; img += gridDim.x) {
    // ---- the wave's 16 tokens -> registers: xr[ks] = channels 32 ks + 8 lq .. +7 of token `tok` (A or B operand alike)
    u32x4 xr[NKS];
    {
      const bf16* src = X + ((size_t)img * S + (tok < S ? tok : S - 1)) * C + lq * 8;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) xr[ks] = qa_gload16(src + ks * 32);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]) :: "memory");
      asm volatile("" : "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]) :: "memory");
      if (tok >= S) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) xr[ks] = zero4;
      }
    }
    bf16* const obase = CTX + (size_t)img * S * (HEADS * HDP);

#pragma unroll 1
    for (int h = 0; h < HEADS; ++h) {
      // ---- q_h, k_h, v_h of the wave's tokens -> the Q / K / V^T images
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW) : "memory");
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sp = smem + slot * SLOT + lane * 16;
        slot = slot == NST - 1 ? 0 : slot + 1;
        const float* bp = btab + (p * HEADS + h) * HDP;
        f32x4 acc[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          if (p < 2) acc[c] = *reinterpret_cast<const f32x4*>(bp + 16 * c + 4 * lq);      // rows = channels 16 c + 4 lq + e
          else { const float bv = bp[16 * c + m]; acc[c] = f32x4{bv, bv, bv, bv}; }        // column = channel 16 c + m
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
          for (int c = 0; c < NCT; ++c) {
            const u32x4 wf = *reinterpret_cast<const u32x4*>(sp + (ks * NCT + c) * 1024);
            acc[c] = p < 2 ? mma_chunk<bf16>(wf, xr[ks], acc[c]) : mma_chunk<bf16>(xr[ks], wf, acc[c]);
          }
          if (ks % 3 == 0 && ks / 3 < PPW) issue1(ks / 3);
        }
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          const bf16x4 o = {(bf16)acc[c][0], (bf16)acc[c][1], (bf16)acc[c][2], (bf16)acc[c][3]};
          if (p < 2) *reinterpret_cast<bf16x4*>((p == 0 ? Qs : Ks) + tok * QS + (16 * c + 4 * lq) * 2) = o;
          else *reinterpret_cast<bf16x4*>(Vt + (16 * c + m) * VS + (wave * 16 + 4 * lq) * 2) = o;   // rows = tokens 16 wave + 4 lq + e
        }
      }
      qa_lds_barrier();

      // ---- attention of head h for the wave's query tile (attention_v2_kernel's body on the LDS images)
      if (wave * 16 < S) {
        const int q = tok;
        u32x4 qf[2];
        qf[0] = *reinterpret_cast<const u32x4*>(Qs + q * QS + lq * 16);
        qf[1] = lq < 2 ? *reinterpret_cast<const u32x4*>(Qs + q * QS + (4 + lq) * 16) : zero4;   // head dim 48 = 1.5 MFMA k-chunks
        f32x4 sc[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          const unsigned char* ka = Ks + (kt * 16 + m) * QS + lq * 16;
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
          a = mma_chunk<bf16>(*reinterpret_cast<const u32x4*>(ka), qf[0], a);
          u32x4 kf = *reinterpret_cast<const u32x4*>(ka + 64);
          if (lq >= 2) kf = zero4;                                       // beyond the row: pad / next row
          sc[kt] = mma_chunk<bf16>(kf, qf[1], a);                        // keys kt*16 + lq*4 + r  x  query m
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = kt * 16 + lq * 4 + r < S;
            sc[kt][r] = ok ? sc[kt][r] * sscale : -INFINITY;
            mx = fmaxf(mx, sc[kt][r]);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(sc[kt][r] - mx);      // -inf for masked keys -> 0
            sc[kt][r] = e;
            sum += e;
          }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        // ctx^T[d][q] = sum_key V^T[d][key] P[q][key]; P in the C layout of S^T is the B operand, the k permutation it implies is
        // applied to the V^T fragment read (two 8-byte reads per 32-key chunk)
#pragma unroll
        for (int dt = 0; dt < NCT; ++dt) {
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
          const unsigned char* va = Vt + (dt * 16 + m) * VS;
#pragma unroll
          for (int kc = 0; kc < NKT / 2; ++kc) {
            const u32x2 v0 = *reinterpret_cast<const u32x2*>(va + (32 * kc + lq * 4) * 2);
            const u32x2 v1 = *reinterpret_cast<const u32x2*>(va + (32 * kc + 16 + lq * 4) * 2);
            const u32x4 vf = {v0[0], v0[1], v1[0], v1[1]};
            const bf16x8 pb = {(bf16)sc[2 * kc][0], (bf16)sc[2 * kc][1], (bf16)sc[2 * kc][2], (bf16)sc[2 * kc][3],
                               (bf16)sc[2 * kc + 1][0], (bf16)sc[2 * kc + 1][1], (bf16)sc[2 * kc + 1][2], (bf16)sc[2 * kc + 1][3]};
            a = mma_chunk<bf16>(vf, __builtin_bit_cast(u32x4, pb), a);
          }
          if (q < S) store4<bf16>(obase + (size_t)q * (HEADS * HDP) + h * HDP + dt * 16 + lq * 4, a * inv);
        }
      }
      // the next part's ring barrier is passed only after every wave has finished these reads: it also releases the images
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // no DMA may be in flight into the LDS of a finished workgroup
}

// Fragment-major weight image from the engine's packed qkv layer (w [3 * HEADS * HDP][kw] bf16, rows (part, head, z)):
//   image (h, p), fragment ks * NCT + c, lane (m, lq), 8 elements: row (p, h, 16 c + m), columns 32 ks + 8 lq .. +7
__global__ void qkv_attn_pack_kernel(const bf16* __restrict__ w, int kw, bf16* __restrict__ img) {
  using namespace qa;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)NIMG * FRAGS * 512) return;
  const int e8 = (int)idx & 7, lane = (int)(idx >> 3) & 63, f = (int)(idx >> 9) % FRAGS, n = (int)(idx >> 9) / FRAGS;
  const int h = n / 3, p = n % 3, ks = f / NCT, c = f % NCT, m = lane & 15, lq = lane >> 4;
  img[idx] = w[(size_t)((p * HEADS + h) * HDP + 16 * c + m) * kw + 32 * ks + 8 * lq + e8];
}

bool qkv_attn_supported(int dtype, int C, int heads, int hdp, int S) {
  static const int on = [] { const char* e = getenv("FSVIT_QKV_ATTN"); return e ? atoi(e) : 1; }();
  return on && dtype == 1 && C == qa::C && heads == qa::HEADS && hdp == qa::HDP && S >= 1 && S <= qa::TOK;
}
size_t qkv_attn_image_bytes() { return (size_t)qa::NIMG * qa::SLOT; }

int launch_qkv_attn_pack(const void* w, int kw, void* img, hipStream_t s) {
  const long n = (long)qa::NIMG * qa::FRAGS * 512;
  hipLaunchKernelGGL(qkv_attn_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16*)w, kw, (bf16*)img);
  return (int)hipGetLastError();
}

int launch_qkv_attn(const void* x, void* ctx, const void* wimg, const float* bias, int B, int S, float scale, hipStream_t s) {
  if (B <= 0) return 0;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)qkv_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, qa::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const int grid = B < 256 ? B : 256;
  hipLaunchKernelGGL(qkv_attn_kernel, dim3(grid), dim3(qa::NW * 64), qa::LDS_BYTES, s, (const bf16*)x, (bf16*)ctx, (const unsigned char*)wimg, bias, B, S,
                     scale);
  return (int)hipGetLastError();
}

}  // namespace fsvit
