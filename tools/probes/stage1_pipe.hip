// EXPERIMENT, NOT BUILT INTO libfsvit.so (round 2; measured 3-4 % SLOWER than csrc/stage1_fused.hip: 1500 vs 1455 us per 6400 images
// on the same box, correct against tests/test_gpu_ops.py).  To reproduce: copy into csrc/, add to DUAL in the Makefile (with
// -fno-slp-vectorize), declare stage1_pipe_supported / launch_stage1_pipe in kernels_decl.inc and call them from engine.hip.
// What was learned: per channel group and SIMD the three resources are about equal - MFMA 1.8 k cycles, LDS reads
// 1.7 k, GELU VALU ~2.4 k (7 plain + 2 quarter-rate transcendental instructions per value) - and they overlap only partially
// even with roles in different phases (4.7 k cycles per step against 5.6 k for the two lock-step intervals of stage1_fused); the
// longer prologue (x through the staging area, then registers) and the global residual re-read eat the difference.  In-kernel
// timers (-DS1_CLK): whichever role has the higher s_setprio finishes its step early and waits for the other - the work is
// zero-sum on the SIMD; reading the taps one ahead (two register sets) does not help, two ahead spills at 128 VGPRs.
//
// Fused Visformer stage-1 block, round-2 design ("wave roles pipelined over the channel groups"), 16-bit:
//   y = x + conv3( GELU( conv2_g8_3x3( GELU( conv1( BN(x) ) ) ) ) )      test_phase/models/visformer.py:259-263, Mlp :152-163
// (eval BN folded into conv1 by the weight packer).  Same contract, LDS plane layouts and per-tile arithmetic as stage1_fused.hip.
//
// Why a second design.  stage1_fused runs every channel group g as two barrier intervals in which ALL 16 waves do the same thing:
// A = {P3(g-1), P1(g): reads -> MFMAs -> GELU -> LDS stores}, B = {P2(g): reads -> MFMAs -> GELU -> stores}.  Per group the LDS reads
// (752 KB = 2.9 k cycles of the 256 B/clk LDS), the MFMAs (450 = 1.8 k cycles of the four pipes) and the GELU VALU (~2 k cycles) run
// one after the other because every wave is in the same phase: 5.8 k cycles per group, MFMA pipe 25 % busy (rocprof PMC).
// Software-pipelining the fragment reads inside a phase did not help (the phases are throughput-, not latency-bound).  Here:
//   * ROLES: waves 0-7 (A) run the grouped 3x3 conv P2(g) while waves 8-15 (B) run conv1 of the NEXT group, P1(g+1), and both
//     accumulate conv3 of the PREVIOUS group, P3(g-1) - one barrier per group step (10 instead of 16 intervals), and inside a step the
//     two role sets are in different phases by construction, so one set's GELU issues under the other's MFMAs on every SIMD
//     (2 A-waves + 2 B-waves each);
//   * conv1's input never comes from LDS inside the group loop: a B wave keeps the x fragments of its two m-tiles in 32 VGPRs for all
//     8 groups (x is staged once through the region that later holds the hidden maps, so the loads are full-line LDS-DMA, not
//     fragment-shaped global loads); the residual is re-read from global (L2) in the epilogue;
//   * P2 register tiles are 2 m-tiles x 2 n-tiles (both halves of the 32-channel group): 1.0 instead of 1.5 ds_read_b128 per MFMA;
//   * H1 / H2 are double-buffered across the steps (P1 writes H1[(g+1)&1] while P2 reads H1[g&1]); the three weight slices of the
//     step after next stream in by LDS-DMA one step ahead, issued mostly by the B waves.
// Per group step: 445 KB of LDS reads (1.7 k cycles), 450 MFMAs (1.8 k), GELU overlapped.  LDS 149 KB.
#include <stdlib.h>

#include "fsvit_common.h"
#include "kernels.h"

// GELU of a register pair.  Scalar gelu_sig, not the packed-fp32 gelu_sig2 of stage1_fused: packed fp32 (v_pk_fma_f32 ...) does not issue
// beside MFMAs on a SIMD (tools/probes/mfma_valu_overlap.hip), and this design lives on one role's GELU running under the other's MFMAs
// (the file is built with -fno-slp-vectorize so that hipcc does not re-pack the pair).
#if defined(S1_NO_GELU)      // timing diagnostics only
#define S1P_GELU2(v) (v)
#elif defined(S1P_PACKED_GELU)
#define S1P_GELU2(v) gelu_sig2(v)
#else
#define S1P_GELU2(v) (f32x2{gelu_sig((v)[0]), gelu_sig((v)[1])})
#endif

namespace FSVIT_NS {

namespace s1p {
constexpr int C1 = 128, HID = 256, G = 8, CG = 32, W = 20;
constexpr int NW = 16;
constexpr int XT = 220, XTP = 224;        // input tokens held (11 rows), padded to 14 m-tiles
constexpr int OT = 200, OTP = 208;        // output tokens (10 rows), padded to 13 m-tiles
constexpr int PW = 36;                    // pitch of the zero-bordered H1 pixel grid (stage1_fused.hip: 20 + 16, no slot aliasing across row ends)
constexpr int H1_PLANE = 12 * PW * 16, H1_BUF = 4 * H1_PLANE;       // 6912, 27648
constexpr int H2_PLANE = OTP * 16, H2_BUF = 4 * H2_PLANE;           // 3328, 13312
constexpr int W1_SLOT = 16 * 32 * 16, W2_SLOT = 36 * 32 * 16, W3_SLOT = 4 * 128 * 16;   // 8192, 18432, 8192
constexpr int OFF_H1 = 0;
constexpr int OFF_H2 = OFF_H1 + 2 * H1_BUF;        //  55296
constexpr int OFF_W1 = OFF_H2 + 2 * H2_BUF;        //  81920   [slot][16 k-chunks][32 n][16 B]
constexpr int OFF_W2 = OFF_W1 + 2 * W1_SLOT;       //  98304   [slot][9 taps * 4 k-chunks][32 n][16 B]
constexpr int OFF_W3 = OFF_W2 + 2 * W2_SLOT;       // 135168   [slot][4 k-chunks][128 n][16 B]
constexpr int OFF_B1 = OFF_W3 + 2 * W3_SLOT;       // 151552   conv1 folded bias, 256 fp32
constexpr int LDS_BYTES = OFF_B1 + HID * 4;        // 152576
static_assert(XTP * 256 <= OFF_W1, "the x staging area overlays the H1 / H2 buffers");
constexpr int KW2 = 320;                  // packed conv2 row length (9*32 = 288 rounded up to the 64-element K slice)
constexpr int P3B = 7;                    // conv3 m-tiles accumulated by a B wave (0..6); the A wave of the same n-tile takes 7..12
constexpr int P3A = 13 - P3B;
static_assert(P3A <= P3B, "the accumulator array is sized for the B waves");
}  // namespace s1p

__device__ __forceinline__ void s1p_dma16(const void* gsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_byte_addr)
      : "memory");
}
__device__ __forceinline__ void s1p_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(1024) void stage1_pipe_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                           const bf16* __restrict__ w1, const float* __restrict__ b1,
                                                           const bf16* __restrict__ w2, const bf16* __restrict__ w3) {
  using namespace s1p;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Xs = smem;                       // prologue only (overlays H1 / H2)
  float* const B1s = reinterpret_cast<float*>(smem + OFF_B1);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lrow = lane & 15, lq = lane >> 4;
  const int b = blockIdx.x >> 1, hsel = blockIdx.x & 1;
  const int xr0 = hsel ? 9 : 0;                       // first image row held
  const int r0 = hsel * 10;                           // first output row
  const bf16* xin = x + ((size_t)b * 400 + xr0 * W) * C1;
  const bool roleB = w >= 8;
  const int j = w & 7;                                // index inside the role set; also the conv3 n-tile of this wave
  const int wq = (w + 8) & 15;                        // DMA issue order: B waves first

  // LDS-DMA of one weight slice into a slot: instruction i fills 16-byte slots [64 i, 64 i + 64); lane -> slot -> source
  auto dma_w1 = [&](int g, int slot) {
    for (int i = wq; i < 8; i += NW) {
      const int sl = i * 64 + lane, ch = sl >> 5, n = sl & 31;
      s1p_dma16(w1 + (size_t)(g * CG + n) * C1 + ch * 8, __builtin_amdgcn_readfirstlane(lds0 + OFF_W1 + slot * W1_SLOT + i * 1024));
    }
  };
  auto dma_w2 = [&](int g, int slot) {
    for (int i = wq; i < 18; i += NW) {
      const int sl = i * 64 + lane, qq = sl >> 5, n = sl & 31;
      s1p_dma16(w2 + (size_t)(g * CG + n) * KW2 + (qq >> 2) * CG + (qq & 3) * 8, __builtin_amdgcn_readfirstlane(lds0 + OFF_W2 + slot * W2_SLOT + i * 1024));
    }
  };
  auto dma_w3 = [&](int g, int slot) {
    for (int i = wq; i < 8; i += NW) {
      const int sl = i * 64 + lane, ch = sl >> 7, n = sl & 127;
      s1p_dma16(w3 + (size_t)n * HID + g * CG + ch * 8, __builtin_amdgcn_readfirstlane(lds0 + OFF_W3 + slot * W3_SLOT + i * 1024));
    }
  };

#ifdef S1_CLK      // timing diagnostics (tools/build_variant.sh): cycles of wave 0 (role A) and wave 8 (role B) of workgroups 0 and 5001
  long long ck0 = __builtin_readcyclecounter(), ckl = ck0, ckP = 0, ckC = 0, ckW = 0, ckE = 0;
#define S1P_STAMP(acc_) { const long long c_ = __builtin_readcyclecounter(); acc_ += c_ - ckl; ckl = c_; }
#define S1P_REPORT(role_) if (lane == 0 && j == 0 && (blockIdx.x == 0 || blockIdx.x == 5001)) \
    printf("[stage1_pipe wg %d role %s] total %lld  prologue %lld  compute %lld  wait+barrier %lld  tail+epilogue %lld\n", (int)blockIdx.x, role_, \
           __builtin_readcyclecounter() - ck0, ckP, ckC, ckW, ckE);
#else
#define S1P_STAMP(acc_)
#define S1P_REPORT(role_)
#endif
  // ---- prologue 1: x tokens (+ halo row) through the staging area, W1(0), W2(0), bias table
  for (int grp = w; grp < XT / 4; grp += NW) {
    const int tk = grp * 4 + (lane >> 4);
    const int ch = (lane & 15) ^ (tk & 15);
    s1p_dma16(xin + (size_t)tk * C1 + ch * 8, __builtin_amdgcn_readfirstlane(lds0 + grp * 1024));
  }
  dma_w1(0, 0);
  dma_w2(0, 0);
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
    if (t < (XTP - XT) * 16) *reinterpret_cast<u32x4*>(Xs + XT * 256 + t * 16) = z;      // pad tokens 220..223
    if (t < HID) B1s[t] = b1[t];
  }
  s1p_dma_wait();
  __syncthreads();

  // ---- prologue 2: a B wave keeps the conv1 input fragments of its m-tiles (j, j + 8) for the whole kernel
  const int p1m0 = j, p1m1 = j + 8;
  const bool p1two = p1m1 < XTP / 16;                 // 14 m-tiles: B waves 0..5 own two
  u32x4 xr[2][4];
  if (roleB) {
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const int sw = (((kc * 4 + lq) ^ lrow) << 4);
      xr[0][kc] = *reinterpret_cast<const u32x4*>(Xs + (p1m0 * 16 + lrow) * 256 + sw);
      xr[1][kc] = *reinterpret_cast<const u32x4*>(Xs + ((p1two ? p1m1 : p1m0) * 16 + lrow) * 256 + sw);
    }
  }
  __syncthreads();                                    // everyone is done with the staging area
  {   // both H1 buffers: the border cells of the 12 x 22 grids must stay zero; every other cell is rewritten per group
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = t; i < (2 * H1_BUF) / 16; i += NW * 64) *reinterpret_cast<u32x4*>(smem + OFF_H1 + i * 16) = z;
  }
  __syncthreads();

  auto h1_store = [&](int buf, int mt, int nt, f32x4 a, f32x4 bias) {
    const int tk = mt * 16 + lrow;
    if (tk < XT) {
      const int pr = tk / W, pc = tk - pr * W;
      const int pix = (pr + (hsel ? 0 : 1)) * PW + pc + 1;
      a += bias;
      const f32x2 g0 = S1P_GELU2((f32x2{a[0], a[1]})), g1 = S1P_GELU2((f32x2{a[2], a[3]}));
      const bf16x4 o = {(bf16)g0[0], (bf16)g0[1], (bf16)g1[0], (bf16)g1[1]};
      *reinterpret_cast<bf16x4*>(smem + OFF_H1 + buf * H1_BUF + (nt * 2 + (lq >> 1)) * H1_PLANE + pix * 16 + (lq & 1) * 8) = o;
    }
  };
  auto h2_store = [&](int buf, int mt, int nt, f32x4 a) {
    const f32x2 g0 = S1P_GELU2((f32x2{a[0], a[1]})), g1 = S1P_GELU2((f32x2{a[2], a[3]}));
    const bf16x4 o = {(bf16)g0[0], (bf16)g0[1], (bf16)g1[0], (bf16)g1[1]};
    *reinterpret_cast<bf16x4*>(smem + OFF_H2 + buf * H2_BUF + (nt * 2 + (lq >> 1)) * H2_PLANE + (mt * 16 + lrow) * 16 + (lq & 1) * 8) = o;
  };
  // weights of the step after next: W1(s+2), W2(s+1), W3(s) (all consumed in step s+1)
  auto step_dma = [&](int s) {
    if (s + 2 < G) dma_w1(s + 2, s & 1);
    if (s + 1 < G) dma_w2(s + 1, (s + 1) & 1);
    dma_w3(s, s & 1);
  };
  // y = acc + x for one conv3 tile: lane holds channels 16 j + 4 lq .. +3 of token mt * 16 + lrow; the residual comes from global (L2)
  const size_t row0 = (size_t)b * 400 + r0 * W;
  auto out_tile = [&](int mt, f32x4 v) {
    const int tk = mt * 16 + lrow;
    if (tk < OT) {
      const size_t off = (row0 + tk) * C1 + j * 16 + lq * 4;
      const bf16x4 r = *reinterpret_cast<const bf16x4*>(x + off);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
      store4<bf16>(y + off, v);
    }
  };

  // The two roles are two separate instruction streams with the same barrier sequence (1 + 8 barriers): a B wave's x fragments and an
  // A wave's tap fragments never share a live range (one loop with role branches inside kept both alive: 42 spilled registers).
  S1P_STAMP(ckP);
  if (roleB) {
    __builtin_amdgcn_s_setprio(2);                      // the B waves are the younger half of the workgroup and lose the VALU arbitration to the A waves otherwise
    // ---- B: conv1 of group s+1 from the register-resident x fragments -> H1[(s+1) & 1]; conv3 m-tiles 0 .. P3B-1 of n-tile j
    f32x4 acc[P3B];
#pragma unroll
    for (int i = 0; i < P3B; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto p1 = [&](int g) {
      const int buf = g & 1;
      const unsigned char* wb = smem + OFF_W1 + buf * W1_SLOT;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(B1s + g * CG + nt * 16 + lq * 4);
        u32x4 wf[4];
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) wf[kc] = *reinterpret_cast<const u32x4*>(wb + ((kc * 4 + lq) * 32 + nt * 16 + lrow) * 16);
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          a0 = mma_chunk<bf16>(wf[kc], xr[0][kc], a0);
          a1 = mma_chunk<bf16>(wf[kc], xr[1][kc], a1);
        }
        h1_store(buf, p1m0, nt, a0, bias);
        if (p1two) h1_store(buf, p1m1, nt, a1, bias);
      }
    };
    auto p3 = [&](int g) {
      const int buf = g & 1;
      const u32x4 wf = *reinterpret_cast<const u32x4*>(smem + OFF_W3 + buf * W3_SLOT + (lq * 128 + j * 16 + lrow) * 16);
      const unsigned char* hb = smem + OFF_H2 + buf * H2_BUF + lq * H2_PLANE + lrow * 16;
#pragma unroll
      for (int c0 = 0; c0 < P3B; c0 += 4) {            // fragment reads in chunks of 4 (register budget: 128 VGPRs at 4 waves per SIMD)
        u32x4 af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) if (c0 + i < P3B) af[i] = *reinterpret_cast<const u32x4*>(hb + (c0 + i) * 256);
#pragma unroll
        for (int i = 0; i < 4; ++i) if (c0 + i < P3B) acc[c0 + i] = mma_chunk<bf16>(wf, af[i], acc[c0 + i]);
      }
    };
    dma_w1(1, 1);                                       // step -1: conv1 of group 0; W1(1) streams in
    p1(0);
    S1P_STAMP(ckC);
    s1p_dma_wait();
    __syncthreads();
    S1P_STAMP(ckW);
#pragma unroll 1
    for (int s = 0; s < G; ++s) {
      step_dma(s);
      if (s + 1 < G) p1(s + 1);
      if (s >= 1) p3(s - 1);
      S1P_STAMP(ckC);
      s1p_dma_wait();
      __syncthreads();
      S1P_STAMP(ckW);
    }
    p3(G - 1);
#pragma unroll
    for (int i = 0; i < P3B; ++i) out_tile(i, acc[i]);
    S1P_STAMP(ckE);
    S1P_REPORT("B");
  } else {
    // ---- A: grouped 3x3 conv of group s, H1[s & 1] -> H2[s & 1], m-tiles (j, j + 8) x both n-tiles; conv3 m-tiles P3B .. 12 of n-tile j
    const int p2m0 = j, p2m1 = j + 8;
    const bool p2two = p2m1 < OTP / 16;                 // 13 m-tiles: A waves 0..4 own two
    int hp0, hp1;                                       // H1 pixel (top-left tap) of this lane's output token
    {
      int tk = p2m0 * 16 + lrow; tk = tk < OT ? tk : OT - 1;
      hp0 = (tk / W) * PW + tk % W;
      tk = (p2two ? p2m1 : p2m0) * 16 + lrow; tk = tk < OT ? tk : OT - 1;      // padded output rows recompute token 199 (ignored later)
      hp1 = (tk / W) * PW + tk % W;
    }
    f32x4 acc[P3A];
#pragma unroll
    for (int i = 0; i < P3A; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto p2 = [&](int g) {
      const int buf = g & 1;
      const unsigned char* wb = smem + OFF_W2 + buf * W2_SLOT + (lq * 32 + lrow) * 16;
      const unsigned char* h0 = smem + OFF_H1 + buf * H1_BUF + lq * H1_PLANE + hp0 * 16;
      const unsigned char* h1 = smem + OFF_H1 + buf * H1_BUF + lq * H1_PLANE + hp1 * 16;
      f32x4 a00 = {0.f, 0.f, 0.f, 0.f}, a01 = a00, a10 = a00, a11 = a00;
      // fragment reads ONE tap ahead of the MFMAs (two register sets, order pinned with sched_group_barrier; three sets spill at 128
      // VGPRs): with only two A waves per SIMD the LDS latency of a read -> wait -> 4 MFMAs chain per tap is no longer hidden by other
      // waves in the same phase
      u32x4 wf0[2], wf1[2], f0[2], f1[2];
      auto ld = [&](int tap, int r) {
        const int toff = ((tap / 3) * PW + tap % 3) * 16;
        wf0[r] = *reinterpret_cast<const u32x4*>(wb + tap * 2048);
        wf1[r] = *reinterpret_cast<const u32x4*>(wb + tap * 2048 + 256);
        f0[r] = *reinterpret_cast<const u32x4*>(h0 + toff);
        f1[r] = *reinterpret_cast<const u32x4*>(h1 + toff);
      };
      ld(0, 0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) ld(tap + 1, (tap + 1) & 1);
        const int r = tap & 1;
        a00 = mma_chunk<bf16>(wf0[r], f0[r], a00);
        a01 = mma_chunk<bf16>(wf1[r], f0[r], a01);
        a10 = mma_chunk<bf16>(wf0[r], f1[r], a10);
        a11 = mma_chunk<bf16>(wf1[r], f1[r], a11);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);        // taps 0, 1
#pragma unroll
      for (int tap = 0; tap < 7; ++tap) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // MFMAs of tap
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);      // reads of tap + 2
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      h2_store(buf, p2m0, 0, a00);
      h2_store(buf, p2m0, 1, a01);
      if (p2two) {
        h2_store(buf, p2m1, 0, a10);
        h2_store(buf, p2m1, 1, a11);
      }
    };
    auto p3 = [&](int g) {
      const int buf = g & 1;
      const u32x4 wf = *reinterpret_cast<const u32x4*>(smem + OFF_W3 + buf * W3_SLOT + (lq * 128 + j * 16 + lrow) * 16);
      const unsigned char* hb = smem + OFF_H2 + buf * H2_BUF + lq * H2_PLANE + lrow * 16;
#pragma unroll
      for (int c0 = 0; c0 < P3A; c0 += 3) {
        u32x4 af[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) if (c0 + i < P3A) af[i] = *reinterpret_cast<const u32x4*>(hb + (P3B + c0 + i) * 256);
#pragma unroll
        for (int i = 0; i < 3; ++i) if (c0 + i < P3A) acc[c0 + i] = mma_chunk<bf16>(wf, af[i], acc[c0 + i]);
      }
    };
    dma_w1(1, 1);
    S1P_STAMP(ckC);
    s1p_dma_wait();
    __syncthreads();
    S1P_STAMP(ckW);
#pragma unroll 1
    for (int s = 0; s < G; ++s) {
      step_dma(s);
      if (s >= 1) p3(s - 1);
      p2(s);
      S1P_STAMP(ckC);
      s1p_dma_wait();
      __syncthreads();
      S1P_STAMP(ckW);
    }
    p3(G - 1);
#pragma unroll
    for (int i = 0; i < P3A; ++i) out_tile(P3B + i, acc[i]);
    S1P_STAMP(ckE);
    S1P_REPORT("A");
  }
}

bool stage1_pipe_supported(int dtype, int C1, int hid, int group, int H1) {
  static const bool off = [] { const char* e = getenv("FSVIT_STAGE1_PIPE"); return e && e[0] == '0'; }();      // A/B against stage1_fused
  return !off && dtype == 1 && C1 == s1p::C1 && hid == s1p::HID && group == s1p::G && H1 == s1p::W;
}

int launch_stage1_pipe(const void* x, void* y, const void* w1, const float* b1, const void* w2, const void* w3, int B, hipStream_t s) {
  if (B <= 0) return 0;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)stage1_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, s1p::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(stage1_pipe_kernel, dim3(B * 2), dim3(s1p::NW * 64), s1p::LDS_BYTES, s, (const bf16*)x, (bf16*)y, (const bf16*)w1, b1,
                     (const bf16*)w2, (const bf16*)w3);
  return (int)hipGetLastError();
}

}  // namespace FSVIT_NS
