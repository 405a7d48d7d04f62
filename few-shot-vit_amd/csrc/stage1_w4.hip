// Fused Visformer stage-1 block, third design ("w4"): ONE wave per SIMD, 32x32x16 MFMAs, the GELUs hand-slotted into the MFMA gaps.
//   y = x + conv3( GELU( conv2_g8_3x3( GELU( conv1( BN(x) ) ) ) ) )     (test_phase/models/visformer.py:259-263, Mlp :152-163; eval BatchNorm folded)
//
// stage1_ring.hip (8 waves, two per SIMD, wave = channel group) sat at 26 % of the bf16 MFMA peak for two rounds: per 64-pixel chunk a SIMD
// issues 2 x 136 MFMAs (4.4 k cycles) and 2 x ~816 VALU, and on gfx950 the MFMA block of one wave and the VALU block of its SIMD partner
// overlap only ~45 % (tools/probes/wave_overlap.hip) - the chunk takes 14.8 k cycles.  What does overlap is VALU issued by the SAME wave in the
// shadow of its own MFMA (MI355X_MICROARCH.md: <= 5 single-issue instructions per 32x32x16 gap at one wave per SIMD; mlp_rows.hip runs on it).
// This kernel is that structure for the stage-1 block:
//   * 4 waves (256 threads, 512 registers each).  Wave w owns channel groups 2w, 2w + 1 (64 hidden channels) for conv1 / conv2 and output
//     channels 32w .. 32w + 31 for conv3.  The weight fragments live in AGPRs (MFMA SrcA reads them there): conv1 64 + conv2 144 + conv3 36 + bias 12;
//     conv3's last seven fragments are re-read from L1 / L2 at the start of every conv3 segment (28 arch VGPRs for one segment);
//   * v_mfma_f32_32x32x16: A = 32 channels x 16 k (weights), B = 32 pixels x 16 k (activations from LDS: [plane of 8 channels][pixel slot][16 B]),
//     C = 32 x 32.  MFMA row R carries channel 16 ((R >> 2) & 1) + 4 (R >> 3) + (R & 3) of its 32-channel tile, so a lane's 16 accumulators are
//     16 CONSECUTIVE channels of one pixel = two 16-byte slots (ds_write_b128 / global_store_dwordx4);
//   * conv1's bias is one more k step (A = the (hi, lo) limbs of the fp32 bias, B = ones): the chain starts with it, no accumulator is initialised
//     by VALU / LDS and none is live before its segment.  The residual is the INITIAL value of conv3's accumulators (copied from the x ring one
//     segment ahead, fp32): no residual registers, no output tile in LDS - the result goes from the accumulators to global memory;
//   * software pipeline over the 64-pixel chunks, four segments per body, each = one MFMA phase with one half (bt = 32-pixel tile) of a GELU
//     pass sliced between its MFMAs:
//        S1  conv1(q)            [36 MFMA]  |  GELU2(q-1) / bt1  -> h2;   residual(q-1) -> acc3
//        --- barrier C: h2(q-1) complete; then the x pixels of chunk q+1 are staged (192-slot ring: the slots of pixels 64q-85 .. 64q-22)
//        S2  conv3(q-1)          [32 MFMA]  |  GELU1(q)   / bt0  -> h1 ring
//        S3  conv2(q) / bt0      [36 MFMA]  |  GELU1(q)   / bt1  -> h1 ring;   y(q-1) stored
//        --- barrier D: every wave is done with h2(q-1); the staged pixels are visible
//        S4  conv2(q) / bt1      [36 MFMA]  |  GELU2(q)   / bt0  -> h2;   loads of chunk q+2
//     conv2 is grouped, so h1 is written and read by the same wave (no barrier).
// Numerics: as stage1_ring - h1 / h2 rounded to the 16-bit storage type where they are stored, everything else fp32, gelu_sig; a pixel's value
// does not depend on its position in a chunk or a launch.
#include <stdlib.h>
#include <utility>
#include <type_traits>

#include "fsvit_common.h"
#include "kernels.h"

namespace FSVIT_NS {

namespace s1w {
constexpr int C1 = 128, HID = 256, CH = 64, RING = 128, XRING = 192, HALO = 21, KW2 = 320;
constexpr int XP = XRING * 16;                   // x ring plane pitch: 192 pixel slots (the residual of chunk q-1 is still there in S1(q))
constexpr int RINGB = RING * 16;                 // bytes of one h1 ring plane's 128 pixel slots (the ring wraps at this)
constexpr int PITCH = RINGB + 16;                // h1 ring plane pitch: the ring + ONE ZERO SLOT (slot 128) - an out-of-image conv2 tap reads it under every
                                                 // immediate plane offset of its fragment (round 6; a separate 12 KB zero region until round 5)
constexpr int XR = 0;                            // x ring: 16 planes
constexpr int H1R = XR + (C1 / 8) * XP;          // h1 ring: 32 planes
constexpr int H2P = CH * 16;
constexpr int H2 = H1R + (HID / 8) * PITCH;      // h2 of one chunk: 32 planes x 64 pixels
// GELU table (bf16 build only, round 6; fsvit_common.h gelu_tab): 7680 bytes behind h2
constexpr bool TABLE = gelu_tab::ON;
constexpr int TAB = H2 + (HID / 8) * H2P;
constexpr int TABC = TAB + 2 * gelu_tab::TN;       // the table's centre: entry i at TABC + 2 i
constexpr int LDS_BYTES = TAB + gelu_tab::BYTES;   // 155 648 (bf16) / 147 968 (f16)
}  // namespace s1w

typedef float f32x16w __attribute__((ext_vector_type(16)));

// compile-time slot loops: the slot index is a type, every schedule test an `if constexpr` (a `#pragma unroll` loop over a body this size is
// refused by the unroller's cost model - and the register arrays it indexes then live in scratch)
template <int... I, typename F>
__device__ __forceinline__ void w4_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void w4_for(F&& f) { w4_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }
#define W4_IC(v) std::integral_constant<int, (v)>{}

// weights in AGPRs (MFMA SrcA), accumulators in arch VGPRs (the GELU reads them)
__device__ __forceinline__ void w4_mfma(const u32x4& w, const u32x4& x, f32x16w& c) {
  asm volatile(FSVIT_MFMA_32x32x16 " %0, %1, %2, %0" : "+v"(c) : "a"(w), "v"(x));
}
__device__ __forceinline__ void w4_mfma_z(const u32x4& w, const u32x4& x, f32x16w& c) {      // first MFMA of a chain that starts from zero
  asm volatile(FSVIT_MFMA_32x32x16 " %0, %1, %2, 0" : "=&v"(c) : "a"(w), "v"(x));
}
__device__ __forceinline__ void w4_mfma_vw(const u32x4& w, const u32x4& x, f32x16w& c) {     // SrcA from arch VGPRs (conv3's re-read fragments)
  asm volatile(FSVIT_MFMA_32x32x16 " %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(x));
}
__device__ __forceinline__ void w4_mfma_az(const u32x4& w, const u32x4& x, f32x16w& c) {     // both operands in AGPRs (the bias step), chain start
  asm volatile(FSVIT_MFMA_32x32x16 " %0, %1, %2, 0" : "=&v"(c) : "a"(w), "a"(x));
}
// workgroup barrier for LDS hand-offs only: the LDS operations of this wave are complete, outstanding GLOBAL loads / stores are not waited for
// (__syncthreads() drains vmcnt too: the prefetches of the next segment would be exposed at every barrier)
__device__ __forceinline__ void w4_bar() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ unsigned w4_pk2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) bf16 bf16x2_t;
  const bf16x2_t v = {(bf16)a, (bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

// PIPE = false: the same data flow with every GELU pass run en bloc behind its MFMA phase (bring-up / reference for the slotted schedule)
// SC: `w3` is the caller's PRE-SCALED copy 8 x W3 (exact; launch_stage1_w4_prescale) - conv3 then accumulates x + W3 h2 itself, and neither the residual
// (x / 8 in) nor the output (8 x acc out) is touched by VALU: -64 VALU per chunk and wave
template <bool PIPE, bool SC>
__global__ __launch_bounds__(256, 1) void stage1_w4_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, const bf16* __restrict__ w1,
                                                           const float* __restrict__ b1, const bf16* __restrict__ w2, const bf16* __restrict__ w3, int M,
                                                           int H, int W, int n_chunks, int chunks_per_wg) {
  using namespace s1w;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, p = lane & 31, kh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int q0 = blockIdx.x * chunks_per_wg;
  int q1 = q0 + chunks_per_wg; q1 = q1 < n_chunks ? q1 : n_chunks;
  if (q0 >= q1) return;
  if (t < 4 * (HID / 8)) reinterpret_cast<unsigned*>(smem + H1R + (t >> 2) * PITCH + RINGB)[t & 3] = 0u;      // the zero slot of every h1 plane
  if constexpr (TABLE) gelu_tab::fill(smem + TAB, t, 256);      // exact-erf form (closer to the reference's nn.GELU than gelu_sig)

  // ---- this wave's weights.  MFMA row R = lane & 31 carries channel cR of its 32-channel tile; the lane's k half = kh
  const int cR = 16 * ((p >> 2) & 1) + 4 * (p >> 3) + (p & 3);
  constexpr int W3A = 9;                                       // conv3 fragments resident in AGPRs; W3A .. 15 are re-read per conv3 segment
  u32x4 wf1[2][8], wf2[2][9][2], wf3[W3A];
  const bf16* const w3row = w3 + (size_t)(32 * w + cR) * HID + 8 * kh;
#pragma unroll
  for (int at = 0; at < 2; ++at)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      // W1 / 8 and b1 / 8 (exact: a power of two): the accumulators are z / 8, min(z^2, 64) / 64 is ONE multiply with the clamp modifier, and what the
      // GELU stores is h / 8 - conv2 then accumulates z2 / 8 by itself, and conv3's 8 x is taken at the output (residual / 8 in, 8 x acc out)
      bf16x8 v = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(w1 + (size_t)(64 * w + 32 * at + cR) * C1 + 16 * ks + 8 * kh));
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] * 0.125f);
      wf1[at][ks] = __builtin_bit_cast(u32x4, v);
    }
#pragma unroll
  for (int gi = 0; gi < 2; ++gi)
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wf2[gi][tp][ks] = *reinterpret_cast<const u32x4*>(w2 + (size_t)(64 * w + 32 * gi + cR) * KW2 + tp * 32 + 16 * ks + 8 * kh);
#pragma unroll
  for (int ks = 0; ks < W3A; ++ks) wf3[ks] = *reinterpret_cast<const u32x4*>(w3row + 16 * ks);
  // conv1's bias as a k step: A = (hi, lo) limbs of the fp32 bias in k slots 0 / 1 of the kh = 0 lanes, B = ones in those slots
  u32x4 wb[2], ones;
  {
    const bf16 one = (bf16)1.0f;
    const unsigned o2 = (unsigned)__builtin_bit_cast(unsigned short, one) * 0x10001u;
    ones = kh == 0 ? u32x4{o2, 0u, 0u, 0u} : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int at = 0; at < 2; ++at) {
      const float b = b1[64 * w + 32 * at + cR] * 0.125f;
      const bf16 hi = (bf16)b, lo = (bf16)(b - (float)hi);
      const unsigned v = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
      wb[at] = kh == 0 ? u32x4{v, 0u, 0u, 0u} : u32x4{0u, 0u, 0u, 0u};
    }
  }
  // pin the fragments to AGPRs once (an "a" INPUT alone leaves the value in an arch VGPR and copies it in front of every MFMA)
#pragma unroll
  for (int at = 0; at < 2; ++at)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+a"(wf1[at][ks]));
#pragma unroll
  for (int gi = 0; gi < 2; ++gi)
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+a"(wf2[gi][tp][ks]));
#pragma unroll
  for (int ks = 0; ks < W3A; ++ks) asm volatile("" : "+a"(wf3[ks]));
  asm volatile("" : "+a"(wb[0]), "+a"(wb[1]), "+a"(ones));

  // ---- x ring slot of pixel 64 q + d: the ring base of chunk q walks 0, 64, 128 (sb, wave-uniform); -85 <= d < 235
  int sb = 0;                                                  // slot of pixel 64 q for the body in flight (q = q0: 0)
  auto xslot16 = [&](int d) {                                  // byte offset of the slot inside a plane
    int s = sb + d;
    s += s < 0 ? XRING : 0;
    s -= s >= XRING ? XRING : 0;
    return s * 16;
  };
  // x batches: 64 consecutive pixels 64 q + d0 + (0 .. 63) (zeros outside [0, M)); 4 x 16 B per thread
  u32x4 px[4];
  unsigned pxok = 0;
  auto gload = [&](int q, int d0) {
    pxok = 0;
#pragma unroll
    for (int u0 = 0; u0 < 4; ++u0) {
      const int u = t + 256 * u0, pp = (u & 7) + 8 * (u >> 7), c8 = (u >> 3) & 15;
      const long m = (long)q * CH + d0 + pp;
      const bool ok = m >= 0 && m < M;
      px[u0] = *reinterpret_cast<const u32x4*>(x + (size_t)(ok ? m : 0) * C1 + c8 * 8);
      pxok |= ok ? (1u << u0) : 0u;
    }
  };
  auto lstore = [&](int d0) {                                  // ... into the ring, relative to the chunk in flight
#pragma unroll
    for (int u0 = 0; u0 < 4; ++u0) {
      const int u = t + 256 * u0, pp = (u & 7) + 8 * (u >> 7), c8 = (u >> 3) & 15;
      const u32x4 v = ((pxok >> u0) & 1u) ? px[u0] : u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(smem + XR + c8 * XP + xslot16(d0 + pp)) = v;
    }
  };

  // ---- per-lane LDS bases
  const unsigned char* const xb = smem + XR + kh * XP;                     // conv1 B fragment: + 2 ks planes + slot
  unsigned char* const h1w = smem + H1R + (8 * w + 2 * kh) * PITCH;        // h1 write: + (4 at + o) planes + slot
  const int h1r_off = H1R + (8 * w + kh) * PITCH;                          // conv2 B fragment: + (4 gi + 2 ks) planes + tap slot
  unsigned char* const h2w = smem + H2 + (8 * w + 2 * kh) * H2P + p * 16;  // h2 write: + (4 gi + o) planes + 32 bt pixels
  const unsigned char* const h2b = smem + H2 + kh * H2P + p * 16;          // conv3 B fragment: + 2 ks planes + 32 bt pixels
  const unsigned char* const xres = smem + XR + (4 * w + 2 * kh) * XP;     // residual: + o planes + slot
  bf16* const yl = y + 32 * w + 16 * kh;                                   // output: + pixel * 128 + 8 o
  const int HW = H * W;

  f32x16w acc1[2][2], acc2[2][2], acc3[2];          // conv1 [at][bt], conv2 [gi][bt], conv3 [bt]
  // conv1 of the 64 pixels 64 q + d0 + (0 .. 63) (bias step first)
  auto conv1_mfma = [&](int d0) {
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      const int slot = xslot16(d0 + 32 * bt + p);
      w4_mfma_az(wb[0], ones, acc1[0][bt]);
      w4_mfma_az(wb[1], ones, acc1[1][bt]);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const u32x4 xf = *reinterpret_cast<const u32x4*>(xb + 2 * ks * XP + slot);
        w4_mfma(wf1[0][ks], xf, acc1[0][bt]);
        w4_mfma(wf1[1][ks], xf, acc1[1][bt]);
      }
    }
  };
  // conv2 of pixel tile bt of chunk q (pixels 64 q + 32 bt + p) for both groups
  auto conv2_mfma = [&](int q, int bt) {
    const int m = q * CH + 32 * bt + p;
    const int rem = m % HW;
    const int oy = m < M ? rem / W : -4, ox = rem % W;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int dy = tp / 3 - 1, dx = tp % 3 - 1;
      const bool ok = (unsigned)(oy + dy) < (unsigned)H && (unsigned)(ox + dx) < (unsigned)W;
      const unsigned char* base = smem + h1r_off + (ok ? ((m + dy * W + dx) & (RING - 1)) * 16 : RINGB);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          const u32x4 hf = *reinterpret_cast<const u32x4*>(base + (4 * gi + 2 * ks) * PITCH);
          if (tp == 0 && ks == 0) w4_mfma_z(wf2[gi][tp][ks], hf, acc2[gi][bt]);
          else w4_mfma(wf2[gi][tp][ks], hf, acc2[gi][bt]);
        }
    }
  };
  // conv3's accumulators := the residual x of the chunk BEFORE the one in flight (pixels 64 q - 64 + 32 bt + p), fp32
  auto res_init = [&]() {
#pragma unroll
    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const bf16x8 r = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xres + o * XP + xslot16(-CH + 32 * bt + p)));
#pragma unroll
        for (int e = 0; e < 8; ++e) acc3[bt][8 * o + e] = SC ? (float)r[e] : (float)r[e] * 0.125f;
      }
  };
  auto conv3_mfma = [&]() {
    u32x4 wv[16 - W3A];
#pragma unroll
    for (int i = 0; i < 16 - W3A; ++i) wv[i] = *reinterpret_cast<const u32x4*>(w3row + 16 * (W3A + i));
    asm volatile("s_nop 3" : "+v"(acc3[0]), "+v"(acc3[1]));                // VALU-written accumulators -> MFMA SrcC
#pragma unroll
    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const u32x4 hf = *reinterpret_cast<const u32x4*>(h2b + 2 * ks * H2P + 32 * bt * 16);
        if (ks < W3A) w4_mfma(wf3[ks], hf, acc3[bt]);
        else w4_mfma_vw(wv[ks - W3A], hf, acc3[bt]);
      }
  };
  auto store_out = [&](int q) {                                            // y of chunk q from the accumulators
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      const long m = (long)q * CH + 32 * bt + p;
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        u32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = SC ? w4_pk2(acc3[bt][8 * o + 2 * j], acc3[bt][8 * o + 2 * j + 1]) : w4_pk2(8.0f * acc3[bt][8 * o + 2 * j], 8.0f * acc3[bt][8 * o + 2 * j + 1]);
        if (m < M) *reinterpret_cast<u32x4*>(yl + (size_t)m * C1 + 8 * o) = v;
      }
    }
  };

  // ================================================================ the steady-state body, slot by slot (q0 < q, q + 2 < q1)
  // Slot m = one MFMA + the fragment read three slots ahead + the GELU micro-stages scheduled at m, pinned by sched_barrier(0).
  // GELU pass = 2 accumulator tiles = 16 pairs; pair k (tile k >> 3, registers 2 (k & 7), + 1) runs the stages
  //   A1 x, u = min(x^2, 64) | A2 p(u), z = x p | E exp2 | Ba 1 + e | Br rcp | C x r, pack      one slot apart, a new pair every 7 / 4 slots
  // (S1: must be done before barrier C) or 2 slots; an octet (4 pairs) is written to LDS when its last pair is packed.
  float gx[4][16][2], gu[4][16][2];
  unsigned gp[4][16];
  auto g_a1 = [&](int J, int k, const f32x16w& a) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      gx[J][k][h] = a[2 * (k & 7) + h];
      asm("v_mul_f32_e64 %0, %1, %1 clamp" : "=v"(gu[J][k][h]) : "v"(gx[J][k][h]));      // min(z^2, 64) / 64 on z / 8
    }
  };
  auto g_a2 = [&](int J, int k) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float pp = fmaf(1.0153755e-3f * 32768.0f, gu[J][k][h], -1.0678257e-1f * 512.0f);      // gelu_sig's polynomial on the scaled arguments: the
      pp = fmaf(pp, gu[J][k][h], -2.3011138f * 8.0f);                                        // coefficients absorb the powers of two (bit-identical)
      gu[J][k][h] = gx[J][k][h] * pp;
    }
  };
  auto g_e = [&](int J, int k) {
#pragma unroll
    for (int h = 0; h < 2; ++h) gu[J][k][h] = __builtin_amdgcn_exp2f(gu[J][k][h]);
  };
  auto g_ba = [&](int J, int k) {
#pragma unroll
    for (int h = 0; h < 2; ++h) gu[J][k][h] = 1.0f + gu[J][k][h];
  };
  auto g_br = [&](int J, int k) {
#pragma unroll
    for (int h = 0; h < 2; ++h) gu[J][k][h] = __builtin_amdgcn_rcpf(gu[J][k][h]);
  };
  auto g_c = [&](int J, int k) { gp[J][k] = w4_pk2(gx[J][k][0] * gu[J][k][0], gx[J][k][1] * gu[J][k][1]); };
  // the micro-stages of job J due at relative slot r; pair k starts at k * NUM / 4 (NUM = 7: 1.75 slots per pair, 8: 2); dst(tile, octet) = LDS address
  auto g_slot = [&](auto jt, auto numt, auto rt, const f32x16w& a0, const f32x16w& a1, auto dst) {
    constexpr int J = decltype(jt)::value, NUM = decltype(numt)::value, r = decltype(rt)::value;
    w4_for<16>([&](auto kc) {
      constexpr int k = decltype(kc)::value, st = k * NUM / 4;
      if constexpr (r == st) g_a1(J, k, k < 8 ? a0 : a1);
      if constexpr (r == st + 1) g_a2(J, k);
      if constexpr (r == st + 2) g_e(J, k);
      if constexpr (r == st + 3) g_ba(J, k);
      if constexpr (r == st + 4) g_br(J, k);
      if constexpr (r == st + 5) {
        g_c(J, k);
        if constexpr ((k & 3) == 3) *reinterpret_cast<u32x4*>(dst(k >> 3, (k >> 2) & 1)) = u32x4{gp[J][k - 3], gp[J][k - 2], gp[J][k - 1], gp[J][k]};
      }
    });
  };
  // ---- round 6: the same passes as TABLE LOOK-UPS (bf16 build).  stage1_w4 was VALU-issue-bound on its 512 GELUs per token (17 VALU per pair, 4 of them
  // quarter-rate) while the LDS pipe idled (r05: VALU busy 62 %, LDS 8 %).  The hidden maps are stored as bf16 anyway, so the GELU is a function of the
  // bf16-ROUNDED pre-activation: 9 VALU + 2 ds_read_u16 per pair (tools/probes/gelu_lds.hip: -18 % cycles per 32-MFMA segment at this kernel's ratio),
  //   T1 code pair = cvt_pk_bf16(z0 / 8, z1 / 8)  | T2 magnitudes (& 0x7fff7fff), re-based to 2^-13 with a saturating packed u16 subtraction
  //   T3 clamp at the last entry; sign mask code >>a 15 | T4 sign as the one's complement i = a ^ mask; byte addresses tabc + 2 i (v_mad_i32_i16, op_sel for the
  //   upper half) | T5 the two gathers | T6 (two slots later) v_lshl_or_b32 joins the halves; octet store as before
  // (d16 loads cannot merge the halves: with SRAM ECC on they zero the other half of the register - measured, gelu_lds.out.txt.)
#ifndef W4_TJOBS
#define W4_TJOBS 0xf                                // jobs (GELU passes) that run as table look-ups: bit 0 GELU2 / bt1, 1 GELU1 / bt0, 2 GELU1 / bt1, 3 GELU2 / bt0
#endif
#ifndef W4_PF
#define W4_PF 3                                     // fragment read-ahead of S2 / S3 / S4 in slots (<= 4: the prologue's fragments are one tap's)
#endif
#ifndef W4_TLAT
#define W4_TLAT 4                                   // slots between a pair's gathers and their use
#endif
  unsigned tc[4][16], ta[4][16], tm[4][16];
  unsigned ad0[4][16], ad1[4][16];
  const unsigned tabc = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + TABC);      // LDS address of entry 0
  unsigned tl[4][16], th[4][16];
  auto t_slot = [&](auto jt, auto numt, auto rt, const f32x16w& a0, const f32x16w& a1, auto dst) {
    constexpr int J = decltype(jt)::value, NUM = decltype(numt)::value, r = decltype(rt)::value;
    w4_for<16>([&](auto kc) {
      constexpr int k = decltype(kc)::value, st = k * NUM / 4;
      // (stage boundaries keep a packed 16-bit result and its reader in different slots: gfx950 needs a wait state between such a pair, hipcc
      // fills it with an s_nop 0)
      if constexpr (r == st) {
        const f32x16w& a = k < 8 ? a0 : a1;
        tc[J][k] = w4_pk2(a[2 * (k & 7)], a[2 * (k & 7) + 1]);
        asm("" : "+v"(tc[J][k]));                            // (opaque: otherwise the sign shift below converts the two floats again)
      }
      if constexpr (r == st + 1) ta[J][k] = gelu_tab::rebase(tc[J][k]);
      if constexpr (r == st + 2) {
        tm[J][k] = gelu_tab::sign_mask(tc[J][k]);
        ta[J][k] = gelu_tab::clamp_top(ta[J][k]);
      }
      if constexpr (r == st + 3) gelu_tab::addresses(ta[J][k] ^ tm[J][k], tabc, ad0[J][k], ad1[J][k]);
      if constexpr (r == st + 4) {
        tl[J][k] = gelu_tab::gather(ad0[J][k]);
        th[J][k] = gelu_tab::gather(ad1[J][k]);
      }
      if constexpr (r == st + 4 + W4_TLAT) {
        gp[J][k] = tl[J][k] | (th[J][k] << 16);
        if constexpr ((k & 3) == 3) *reinterpret_cast<u32x4*>(dst(k >> 3, (k >> 2) & 1)) = u32x4{gp[J][k - 3], gp[J][k - 2], gp[J][k - 1], gp[J][k]};
      }
    });
  };
  // one GELU pass slot: the table form (bf16) or the VALU form (fp16: a 10-bit mantissa would need an 8 x larger table).  NT / NV = pair spacing of
  // the two forms in quarter slots (the table form finishes 6 slots after a pair's start, the VALU form 5)
  auto gelu_slot = [&](auto jt, auto nt, auto nv, auto rt, const f32x16w& a0, const f32x16w& a1, auto dst) {
    if constexpr (TABLE && ((W4_TJOBS >> decltype(jt)::value) & 1)) t_slot(jt, nt, rt, a0, a1, dst);
    else g_slot(jt, nv, rt, a0, a1, dst);
  };
  typedef std::integral_constant<int, 0> J0;
  typedef std::integral_constant<int, 1> J1;
  typedef std::integral_constant<int, 2> J2;
  typedef std::integral_constant<int, 3> J3;
  typedef std::integral_constant<int, 6> N6;
  typedef std::integral_constant<int, 7> N7;
  typedef std::integral_constant<int, 8> N8;

  // pixel coordinates of the lane's two conv2 pixels (m = 64 q + 32 bt + p), advanced by one chunk per body.  Tap validity as LANE MASKS in SGPRs: three row
  // masks (dy = -1, 0, +1) and three column masks per pixel tile, formed once per body (6 compares); a tap then costs one scalar AND and one v_cndmask
  // on the mask instead of two adds, two compares and a select
  const int divW = (65536 + W - 1) / W;
  int remq[2], tms[2];
  unsigned long long rmk[2][3], cmk[2][3];
  // tap addresses are LDS ADDRESSES (the base of `smem` folded into the per-lane constants): a read through `smem + offset` costs a v_add of the base
  // per fragment base, 18 per chunk
  typedef __attribute__((address_space(3))) const u32x4 lds_frag;
  typedef lds_frag* lds_frag_t;
  const int h1r_lds = h1r_off + (int)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int zero_off = h1r_lds + RINGB;                     // the zero slot of the lane's plane (a VGPR: v_cndmask reads the mask over the constant bus)
  auto tap_setup = [&](int q, int bt) {                       // remq[bt] = m % HW is current
    const int m = q * CH + 32 * bt + p;
    const int oy = m < M ? (remq[bt] * divW) >> 16 : -4;
    const int ox = remq[bt] - ((remq[bt] * divW) >> 16) * W;
    tms[bt] = (m & (RING - 1)) * 16;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      rmk[bt][d] = __builtin_amdgcn_ballot_w64((unsigned)(oy + d - 1) < (unsigned)H);
      cmk[bt][d] = __builtin_amdgcn_ballot_w64((unsigned)(ox + d - 1) < (unsigned)W);
    }
  };
  auto tap_addr = [&](int bt, int tp) {                       // LDS address of the tap's fragment (plane 8 w + kh), or of the plane's zero slot
    const int dy = tp / 3 - 1, dx = tp % 3 - 1;
    const unsigned long long mk = rmk[bt][tp / 3] & cmk[bt][tp % 3];
    const int in = h1r_lds + ((tms[bt] + (dy * W + dx) * 16) & (RINGB - 1));
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(zero_off), "v"(in), "s"(mk));
    return r;
  };
  auto rem_advance = [&](int bt) {
    remq[bt] += CH;
#pragma unroll
    for (int i = 0; i < 4; ++i) remq[bt] -= remq[bt] >= HW ? HW : 0;       // HW >= 16: at most four wraps per 64 pixels
  };

  // a GELU pass on its own (prologue, first / last bodies): the same micro-stages in the same order, pinned, so that no more than three pairs are
  // in flight (the en-bloc form let the scheduler interleave 64 GELUs: 209 live VGPRs in a kernel that has 256)
  auto gelu1 = [&](long P0, int bt) {
    const int slot = (int)((P0 + 32 * bt + p) & (RING - 1)) * 16;
    auto dst = [&](int at, int o) { return h1w + (4 * at + o) * PITCH + slot; };
    w4_for<36 + W4_TLAT>([&](auto rc) {
      gelu_slot(J1{}, N8{}, N8{}, rc, acc1[0][bt], acc1[1][bt], dst);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  auto gelu2 = [&](int bt) {
    auto dst = [&](int gi, int o) { return h2w + (4 * gi + o) * H2P + 32 * bt * 16; };
    w4_for<36 + W4_TLAT>([&](auto rc) {
      gelu_slot(J3{}, N8{}, N8{}, rc, acc2[0][bt], acc2[1][bt], dst);
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- prologue: the ring holds pixels [64 q0 - 21, 64 q0 + 107); h1 of the first batch; conv1 of the second batch = S1 of the first body
  gload(q0, -HALO);
  __syncthreads();                                          // zero region visible, LDS free
  lstore(-HALO);
  gload(q0, -HALO + CH);
  __syncthreads();
  conv1_mfma(-HALO);
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc1[0][0]), "+v"(acc1[0][1]), "+v"(acc1[1][0]), "+v"(acc1[1][1]));   // MFMA -> VALU read of the accumulators
  gelu1((long)q0 * CH - HALO, 0);
  gelu1((long)q0 * CH - HALO, 1);
  lstore(-HALO + CH);
  if (q0 + 1 < q1) gload(q0, -HALO + 2 * CH);
  __syncthreads();

#define W4_STAMP(i) do { } while (0)
  auto body_pipe = [&](int q) {
    const long Pn = (long)q * CH - HALO + CH;               // linear index of the first of the 64 pixels chunk q adds (h1 ring slots)
    const int sn0 = (int)((Pn + p) & (RING - 1)) * 16, sn1 = (int)((Pn + 32 + p) & (RING - 1)) * 16;
    const int xn0 = xslot16(CH - HALO + p), xn1 = xslot16(CH - HALO + 32 + p);      // ... and their x ring slots
    auto h2dst1 = [&](int gi, int o) { return h2w + (4 * gi + o) * H2P + 32 * 16; };
    auto h2dst0 = [&](int gi, int o) { return h2w + (4 * gi + o) * H2P; };
    auto h1dst0 = [&](int at, int o) { return h1w + (4 * at + o) * PITCH + sn0; };
    auto h1dst1 = [&](int at, int o) { return h1w + (4 * at + o) * PITCH + sn1; };
    u32x4 wv[16 - W3A];                                     // conv3's fragments W3A .. 15: requested late in S1 (L2 latency under ~6 slots), used in S2
    // ---------------- S1: conv1(q), slots 0..35 (per pixel tile: 2 bias MFMAs + 16) | GELU2(q-1)/bt1 (job 0, from slot 1) | residual(q-1) -> acc3
    {
      u32x4 xf[3], rr[4];
      xf[0] = *reinterpret_cast<const u32x4*>(xb + xn0);
      xf[1] = *reinterpret_cast<const u32x4*>(xb + 2 * XP + xn0);
      __builtin_amdgcn_sched_barrier(0);
      w4_for<36>([&](auto mc) {
        constexpr int m = decltype(mc)::value, bt = m / 18, j = m % 18;
        if constexpr (j < 2) w4_mfma_az(wb[j], ones, acc1[j][bt]);
        else {
          constexpr int ks = (j - 2) >> 1, at = (j - 2) & 1, f = 8 * bt + ks;
          w4_mfma(wf1[at][ks], xf[f % 3], acc1[at][bt]);
          if constexpr (at == 1 && f + 3 < 16) {           // (fragment f is dead: f + 3 takes its register)
            constexpr int f3 = f + 3;
            xf[f3 % 3] = *reinterpret_cast<const u32x4*>(xb + 2 * (f3 & 7) * XP + (f3 >> 3 ? xn1 : xn0));
          }
        }
        if constexpr (m == 0) {
          asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc2[0][1]), "+v"(acc2[1][1]));      // last MFMA of the body before -> first GELU read (behind this slot's MFMA)
          xf[2] = *reinterpret_cast<const u32x4*>(xb + 4 * XP + xn0);
        }
        if constexpr (m >= 1 && m <= 35) gelu_slot(J0{}, N7{}, N7{}, W4_IC(m - 1), acc2[0][1], acc2[1][1], h2dst1);      // (table: last octet stored at m = 33)
        // residual of chunk q-1 (x ring, still intact: its slots are re-staged behind barrier C of the NEXT body) -> conv3's accumulators, one octet per
        // visit at the END of the segment (GELU2(q-1) has read acc2[.][1] for the last time at slot 27: acc3 can take those registers): read at
        // slots 28 .. 31, converted at 32 .. 35
        if constexpr (m >= 28 && m < 32) {
          constexpr int i = m - 28, bt2 = i >> 1, o = i & 1;
          rr[i] = *reinterpret_cast<const u32x4*>(xres + o * XP + xslot16(-CH + 32 * bt2 + p));
        }
        if constexpr (m >= 36 - (16 - W3A)) wv[m - (36 - (16 - W3A))] = *reinterpret_cast<const u32x4*>(w3row + 16 * (W3A + m - (36 - (16 - W3A))));
        if constexpr (m >= 32) {
          constexpr int i = m - 32, bt2 = i >> 1, o = i & 1;
          const bf16x8 r = __builtin_bit_cast(bf16x8, rr[i]);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc3[bt2][8 * o + e] = SC ? (float)r[e] : (float)r[e] * 0.125f;
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    W4_STAMP(0);
    w4_bar();                                               // C
    W4_STAMP(1);
    lstore(2 * CH - HALO);                                  // x pixels of chunk q+1 (loaded in S4 of the body before)
    // ---------------- S2: conv3(q-1), slots 0..31 | GELU1(q)/bt0 (job 1)
    {
      u32x4 hf[W4_PF + 1];
#pragma unroll
      for (int i = 0; i < W4_PF; ++i) hf[i] = *reinterpret_cast<const u32x4*>(h2b + 2 * i * H2P);
      __builtin_amdgcn_sched_barrier(0);
      w4_for<32>([&](auto mc) {
        constexpr int m = decltype(mc)::value, bt = m >> 4, ks = m & 15;
        if constexpr (ks < W3A) w4_mfma(wf3[ks], hf[m % (W4_PF + 1)], acc3[bt]);
        else w4_mfma_vw(wv[ks - W3A], hf[m % (W4_PF + 1)], acc3[bt]);
        if constexpr (m + W4_PF < 32) hf[(m + W4_PF) % (W4_PF + 1)] = *reinterpret_cast<const u32x4*>(h2b + 2 * ((m + W4_PF) & 15) * H2P + ((m + W4_PF) >> 4) * 32 * 16);
        gelu_slot(J1{}, N6{}, N7{}, mc, acc1[0][0], acc1[1][0], h1dst0);      // (table: a pair every 1.5 slots, last octet stored at m = 28)
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    W4_STAMP(2);
    // ---------------- S3: conv2(q)/bt0, slots 0..35 | GELU1(q)/bt1 (job 2) | y(q-1) from the accumulators | tap coordinates of bt1
    {
      u32x4 hf[W4_PF + 1];
      int ta = tap_addr(0, 0), tb = tap_addr(0, 1);          // current / next tap's base
#pragma unroll
      for (int i = 0; i < W4_PF; ++i) hf[i] = *(lds_frag_t)(size_t)(unsigned)(ta + (4 * (i & 1) + 2 * ((i >> 1) & 1)) * PITCH);
      __builtin_amdgcn_sched_barrier(0);
      w4_for<36>([&](auto mc) {
        constexpr int m = decltype(mc)::value, tp = m >> 2, ks = (m >> 1) & 1, gi = m & 1;
        if constexpr (m < 2) w4_mfma_z(wf2[gi][0][0], hf[m % (W4_PF + 1)], acc2[gi][0]);
        else w4_mfma(wf2[gi][tp][ks], hf[m % (W4_PF + 1)], acc2[gi][0]);
        if constexpr (m + W4_PF < 36) {                    // the read W4_PF slots ahead: fragment m3 = (tap m3 >> 2, ks, gi); a tap's base is formed one tap early
          constexpr int m3 = m + W4_PF;
          if constexpr ((m3 & 3) == 0) ta = tb;
          hf[m3 % (W4_PF + 1)] = *(lds_frag_t)(size_t)(unsigned)(ta + (4 * (m3 & 1) + 2 * ((m3 >> 1) & 1)) * PITCH);
          if constexpr ((m3 & 3) == 2 && (m3 >> 2) + 1 < 9 && (m3 >> 2) >= 1) tb = tap_addr(0, (m3 >> 2) + 1);
        }
        gelu_slot(J2{}, N7{}, N8{}, mc, acc1[0][1], acc1[1][1], h1dst1);
        if constexpr (m == 4 || m == 10 || m == 16 || m == 22) {     // y of chunk q-1: one 16-byte octet per visit (acc3 was complete 4+ slots ago)
          constexpr int i = (m - 4) / 6, bt = i >> 1, o = i & 1;
          const long mm = (long)(q - 1) * CH + 32 * bt + p;
          u32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = SC ? w4_pk2(acc3[bt][8 * o + 2 * j], acc3[bt][8 * o + 2 * j + 1]) : w4_pk2(8.0f * acc3[bt][8 * o + 2 * j], 8.0f * acc3[bt][8 * o + 2 * j + 1]);
          if (mm < M) *reinterpret_cast<u32x4*>(yl + (size_t)mm * C1 + 8 * o) = v;
        }
        if constexpr (m == 28) tap_setup(q, 1);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    W4_STAMP(3);
    w4_bar();                                               // D
    W4_STAMP(4);
    // ---------------- S4: conv2(q)/bt1, slots 0..35 | GELU2(q)/bt0 (job 3) | loads of chunk q+2 | tap coordinates of q+1 / bt0
    {
      u32x4 hf[W4_PF + 1];
      int ta = tap_addr(1, 0), tb = tap_addr(1, 1);
#pragma unroll
      for (int i = 0; i < W4_PF; ++i) hf[i] = *(lds_frag_t)(size_t)(unsigned)(ta + (4 * (i & 1) + 2 * ((i >> 1) & 1)) * PITCH);
      __builtin_amdgcn_sched_barrier(0);
      w4_for<36>([&](auto mc) {
        constexpr int m = decltype(mc)::value, tp = m >> 2, ks = (m >> 1) & 1, gi = m & 1;
        if constexpr (m < 2) w4_mfma_z(wf2[gi][0][0], hf[m % (W4_PF + 1)], acc2[gi][1]);
        else w4_mfma(wf2[gi][tp][ks], hf[m % (W4_PF + 1)], acc2[gi][1]);
        if constexpr (m == 0) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc2[0][0]), "+v"(acc2[1][0]));      // S3's last MFMA -> first GELU read
        if constexpr (m + W4_PF < 36) {
          constexpr int m3 = m + W4_PF;
          if constexpr ((m3 & 3) == 0) ta = tb;
          hf[m3 % (W4_PF + 1)] = *(lds_frag_t)(size_t)(unsigned)(ta + (4 * (m3 & 1) + 2 * ((m3 >> 1) & 1)) * PITCH);
          if constexpr ((m3 & 3) == 2 && (m3 >> 2) + 1 < 9 && (m3 >> 2) >= 1) tb = tap_addr(1, (m3 >> 2) + 1);
        }
        if constexpr (m >= 1) gelu_slot(J3{}, N7{}, N8{}, W4_IC(m - 1), acc2[0][0], acc2[1][0], h2dst0);
        if constexpr (m == 20) gload(q, 3 * CH - HALO);    // pixels of chunk q+2: staged behind barrier C of the next body (one S1 = ~1.8 k cycles away)
        if constexpr (m == 30) { rem_advance(0); rem_advance(1); }
        if constexpr (m == 32) tap_setup(q + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (!(TABLE && ((W4_TJOBS >> 3) & 1))) g_slot(J3{}, N8{}, W4_IC(35), acc2[0][0], acc2[1][0], h2dst0);
    }
    W4_STAMP(5);
  };

  for (int q = q0; q <= q1; ++q, sb = sb + CH >= XRING ? sb + CH - XRING : sb + CH) {
    if constexpr (PIPE) {
      if (q > q0 && q + 2 < q1) {                               // the steady state; the first body, the last two and the tail run the plain sequence below
        if (q == q0 + 1) {
#pragma unroll
          for (int bt = 0; bt < 2; ++bt) remq[bt] = (q * CH + 32 * bt + p) % HW;
          tap_setup(q, 0);
        }
        body_pipe(q);
        continue;
      }
    }
    // Body q, plain (first body, last two, tail): the same work and the same state at its end, phase after phase - one accumulator set live at a time
    //   GELU2(q-1)/bt1, residual(q-1) -> acc3;  C;  staging;  conv3(q-1), y(q-1);  conv1(q), GELU1(q);  D;  conv2(q)/bt0, GELU2(q)/bt0, conv2(q)/bt1
    // (q == q0: no chunk q-1;  q == q1: only the tail of chunk q1-1)
    const bool cur = q < q1, prev = q > q0;
    const long Pn = (long)q * CH - HALO + CH;               // the 64 pixels chunk q adds
    if (prev) {
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc2[0][1]), "+v"(acc2[1][1]));
      gelu2(1);
      res_init();
    }
    __syncthreads();                                        // C
    if (cur && q + 1 < q1) lstore(2 * CH - HALO);           // x pixels of chunk q+1
    if (prev) {
      conv3_mfma();
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc3[0]), "+v"(acc3[1]));
      store_out(q - 1);
    }
    if (cur) {
      conv1_mfma(CH - HALO);
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc1[0][0]), "+v"(acc1[0][1]), "+v"(acc1[1][0]), "+v"(acc1[1][1]));
      gelu1(Pn, 0);
      gelu1(Pn, 1);
    }
    __syncthreads();                                        // D
    if (cur) {
      if (q + 2 < q1) gload(q, 3 * CH - HALO);
      conv2_mfma(q, 0);
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc2[0][0]), "+v"(acc2[1][0]));
      gelu2(0);
      conv2_mfma(q, 1);
    }
  }
}

// The engines' stage-1 kernel since round 4 (round 6: 1.91 ms per 12 800-image launch, stage1_ring's eight-wave kernel 2.22).  The FSVIT_STAGE1_W4=0 switch
// left the sources in round 6 (tools/probes/variants/dispatch_switches.r06.patch re-adds it and the other retired dispatch switches)
bool stage1_w4_enabled() {
  constexpr bool off = false;
  return !off;
}

__global__ __launch_bounds__(256) void stage1_w4_prescale_kernel(const bf16* __restrict__ w, bf16* __restrict__ ws, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) ws[i] = (bf16)((float)w[i] * 8.0f);      // exact: a power of two
}
// ws := 8 x w (the conv3 weights the engines hand to launch_stage1_w4 as `w3s`); n elements of the storage type
int launch_stage1_w4_prescale(const void* w, void* ws, int n, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(stage1_w4_prescale_kernel, dim3((n + 255) / 256 < 256 ? (n + 255) / 256 : 256), dim3(256), 0, s, (const bf16*)w, (bf16*)ws, n);
  return (int)hipGetLastError();
}

int launch_stage1_w4(const void* x, void* y, const void* w1, const float* b1, const void* w2, const void* w3, int B, int H, int W, hipStream_t s, const void* w3s) {
  const long Ml = (long)B * H * W;
  if (Ml <= 0) return 0;
  if (Ml >= (1L << 31) - 256 || W > 20 || H * W < 16) return (int)hipErrorInvalidValue;
  const int M = (int)Ml, n_chunks = (M + s1w::CH - 1) / s1w::CH;
  int wgs = n_chunks < 256 ? n_chunks : 256;              // one 4-wave workgroup per CU (156 KB of LDS)
  const int cpw = (n_chunks + wgs - 1) / wgs;
  wgs = (n_chunks + cpw - 1) / cpw;
  // (the unpipelined variant <false, false> and the FSVIT_STAGE1_W4_PIPE / _PRESCALE switches of round 4 are out of the shipped dispatch:
  // tools/probes/variants/stage1_switches.r05.patch re-adds them for timing)
  const bool sc = w3s != nullptr;
  auto kern = sc ? stage1_w4_kernel<true, true> : stage1_w4_kernel<true, false>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, s1w::LDS_BYTES);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), s1w::LDS_BYTES, s, (const bf16*)x, (bf16*)y, (const bf16*)w1, b1, (const bf16*)w2, (const bf16*)(sc ? w3s : w3), M, H, W, n_chunks, cpw);
  return (int)hipGetLastError();
}

}  // namespace FSVIT_NS
