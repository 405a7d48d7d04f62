// Fused Visformer stage-1 block (bf16 / f16):   y = x + conv3( GELU( conv2_g8_3x3( GELU( conv1( BN(x) ) ) ) ) )
// (test_phase/models/visformer.py:259-263 Block.forward with attn_disabled, Mlp :152-163; eval BN
// folded into conv1 by the weight packer).
//
// As three conv_gemm launches this block is bound by its intermediates, not by MFMA: the two 256-channel
// hidden maps cost 4 x 205 KB of HBM traffic per image and the grouped 3x3 (N = 32 per group) cannot
// fill a 128-wide tile.  Here one 512-thread workgroup owns one half image (10 x 20 output tokens) and keeps
// everything on chip.  v3 (round 2): the 16-wave v2 spent 46 % of its cycles with the LDS array busy and 25 %
// with the MFMA pipe busy (profiles/r02_mfma_pmc.json) - every 16x16x32 MFMA was fed by 1.5 KB of fragment reads,
// each wave owning one 16 x 16 output tile per phase.  v3 is organised around LDS bytes per MFMA:
//   * wave w (0..6) owns the token tiles 2w, 2w+1 and BOTH 16-channel tiles of the current group: 2 x 2 register
//     blocking, 4 fragment reads per 4 MFMAs;
//   * its input-token fragments (2 tiles x 4 K chunks, the B operand of conv1) are loaded from Xs ONCE and stay in
//     32 VGPRs for all 8 groups;
//   * conv2's output never goes to LDS: the rows of W2g are brought into LDS in the order that makes the two
//     accumulator tiles of a token tile, GELU'd and packed, exactly the B operand (32 hidden channels of the group)
//     of conv3 - the wave multiplies them straight into its 2 x 8 output-channel accumulators (64 VGPRs);
//   * wave 7 does no arithmetic: it issues the LDS-DMA of the NEXT group's three weight slices (34 KB, double
//     buffered) and is the only wave that ever waits on vmcnt.
//   Xs   input tokens incl. one halo row, [224][128] in LDS (row-major, 16-B chunks XOR (token & 15)); residual source
//   per group g of 32 hidden channels:
//     P1  H1g = GELU(X . W1g^T + b1g)             224 tokens incl. halo  -> LDS planes, zero-bordered 12 x 22 pixels
//     ---- barrier
//     P2  H2g = GELU(conv3x3(H1g, W2g))           9 taps x (2 + 2) fragment reads, 36 MFMAs   -> registers
//     P3  acc += W3[:, g] . H2g^T                 8 fragment reads, 16 MFMAs
//     ---- barrier
//   y = acc + Xs, written once.
// HBM traffic per image: 102 KB in + 102 KB out; LDS fragment traffic per group 52 KB per wave (7 waves) instead of
// 43 KB per wave (16 waves).
#include <stdlib.h>

#include "fsvit_common.h"
#include "kernels.h"

// GELU of a register pair: packed (gelu_sig2) or two scalar gelu_sig (-DS1_SCALAR_GELU: packed fp32 does not issue behind MFMAs)
#if defined(S1_NO_GELU)      // timing diagnostics only
#define S1_GELU2(v) (v)
#elif defined(S1_SCALAR_GELU)
#define S1_GELU2(v) (f32x2{gelu_sig((v)[0]), gelu_sig((v)[1])})
#else
#define S1_GELU2(v) gelu_sig2(v)
#endif

namespace FSVIT_NS {

namespace s1 {
constexpr int C1 = 128, HID = 256, G = 8, CG = 32, W = 20;
constexpr int NW = 8;                     // waves per workgroup: 7 compute waves + the weight-DMA wave (one workgroup per CU)
constexpr int XT = 220;                   // input tokens (10 rows + one halo row): 14 m-tiles, the last 4 tokens empty
constexpr int OT = 200;                   // output tokens (10 rows): 13 m-tiles, the last half empty
constexpr int PW = 36;                    // pitch of the zero-bordered H1 pixel grid (12 rows; columns -1 .. 20 used).  36 = 20 + 16: the 16 tokens of an
                                          // m-tile usually straddle a row end, and with the natural pitch 22 the tokens after the wrap land on the 16-byte
                                          // slots (mod 16) of the ones before it - 6.5 LDS cycles per ds_read_b128 instead of 4 (tools/lds_conflicts.py)
constexpr int H1_PLANE = 12 * PW * 16;    // 6912 = 27 * 256: the plane stride stays 0 mod 256 B
constexpr int OFF_H1 = 0;
constexpr int OFF_WS = OFF_H1 + 4 * H1_PLANE;      //  27648  two weight sets
constexpr int WS_W1 = 0;                           //  [16 k-chunks][32 n][16 B]
constexpr int WS_W2 = 16 * 32 * 16;                //  8192   [9 taps * 4 k-chunks][32 n][16 B], row n = hidden channel sigma(n)
constexpr int WS_W3 = WS_W2 + 36 * 32 * 16;        //  26624  [4 k-chunks][128 n][16 B]
constexpr int WSET = WS_W3 + 4 * 128 * 16;         //  34816 = 34 LDS-DMA pieces of 1 KB
constexpr int NPIECE = WSET / 1024;
constexpr int OFF_B1 = OFF_WS + 2 * WSET;          //  97280  conv1 folded bias, 256 fp32
constexpr int LDS_BYTES = OFF_B1 + HID * 4;        //  98304
constexpr int KW2 = 320;                  // packed conv2 row length (9*32 = 288 rounded up to the 64-element K slice)
}  // namespace s1

// workgroup barrier that orders LDS traffic only: global loads (the next image's tokens, the next weight set) stay in flight across it
__device__ __forceinline__ void s1_bar_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Weight image: [8 groups][34 pieces of 1 KB] = the three slices of a group exactly as they sit in LDS, so a piece is a linear 1 KB LDS-DMA copy
// (a piece gathered from the row-major layer weights touches 64 cache lines for 1 KB of payload: the helper wave then needed 290 cycles per piece
// and set the length of every interval).  Slot sl of group g:
//   W1 [16 k-chunks][32 n]: hidden channel 32 g + n, input channels 8 ch ..;   W2 [36 = tap * 4 + k-chunk][32 n]: ROW PERMUTED, LDS row n = 16 t + R
//   holds hidden channel sigma(n) = 8 (R >> 2) + 4 t + (R & 3), so that the MFMA output rows 4 lq + e of tile t are the channels 8 lq + 4 t + e: tile 0 /
//   tile 1 of one token tile, side by side, ARE conv3's B fragment (k = 8 lq + j) in natural order;   W3 [4 k-chunks][128 n]: hidden channels 32 g + 8 ch ..
__global__ __launch_bounds__(256) void stage1_pack_kernel(const bf16* __restrict__ w1, const bf16* __restrict__ w2, const bf16* __restrict__ w3,
                                                          u32x4* __restrict__ img) {
  using namespace s1;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= G * (WSET / 16)) return;
  const int g = idx / (WSET / 16), sl = idx % (WSET / 16);
  const bf16* src;
  if (sl < 512) {
    const int ch = sl >> 5, n = sl & 31;
    src = w1 + (size_t)(g * CG + n) * C1 + ch * 8;
  } else if (sl < 512 + 1152) {
    const int s2 = sl - 512, qq = s2 >> 5, n = s2 & 31;
    const int sg = 8 * ((n & 15) >> 2) + 4 * (n >> 4) + (n & 3);
    src = w2 + (size_t)(g * CG + sg) * KW2 + (qq >> 2) * CG + (qq & 3) * 8;
  } else {
    const int s3 = sl - 1664, ch = s3 >> 7, n = s3 & 127;
    src = w3 + (size_t)n * HID + g * CG + ch * 8;
  }
  img[idx] = *reinterpret_cast<const u32x4*>(src);
}

__global__ __launch_bounds__(512) void stage1_block_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, const unsigned char* __restrict__ wimg,
                                                           const float* __restrict__ b1, int n_img) {
  using namespace s1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const H1 = smem + OFF_H1;
  float* const B1s = reinterpret_cast<float*>(smem + OFF_B1);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;

  const int t = threadIdx.x, lane = t & 63;
#ifdef S1_CLK
  long long ck0 = __builtin_readcyclecounter(), ckA = 0, ckB = 0, ckP = 0, ckl = 0, ckE = 0, ckA1 = 0, ckA2 = 0, ckB2 = 0, ckB3 = 0;
#endif
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lrow = lane & 15, lq = lane >> 4;
  // Persistent workgroup: a fixed image half (hsel: the zero border row of H1 differs between the halves) of the images first, first + stride, ...
  // With a grid that fills the XCDs evenly the two halves of an image run on the same XCD at the same time (their shared halo rows meet in its L2).
  int hsel, first, stride;
  if ((gridDim.x & 15) == 0) {
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    hsel = q & 1; first = (q >> 1) * 8 + xcd; stride = gridDim.x >> 1;
  } else {
    hsel = blockIdx.x & 1; first = blockIdx.x >> 1; stride = (gridDim.x + 1) >> 1;
  }
  // Token order inside the workgroup: tokens 0..199 = the 10 output rows r0 .. r0+9, tokens 200..219 = the halo row (image row 10 for the upper
  // half, 9 for the lower): output token == input token, so the x fragments a wave holds for conv1 are also its residual.
  const int r0 = hsel * 10;                           // first output row
  const int hrow = hsel ? 9 : 10;                     // halo row

  // ---- once per workgroup: H1 zeroed (its border must stay 0), the bias table
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = t; i < (4 * H1_PLANE) / 16; i += NW * 64) *reinterpret_cast<u32x4*>(H1 + i * 16) = z;
    if (t < HID) B1s[t] = b1[t];        // bias table: an in-loop global load would cost a full round trip per tile
  }

  if (w == NW - 1) {
    // ================================================================ the weight wave
    // Streams the next group's 34 KB weight set into the free LDS set through its own registers: global_load_dwordx4 x 34 one group ahead,
    // ds_write_b128 x 34 when the set's last readers have passed the barrier.  (LDS-DMA, global_load_lds_dwordx4, costs a wave ~100-200 cycles of
    // issue per 1 KB piece whatever the source - measured here with an L1-hot source: 13 pieces = 2400 cycles on this wave, 3 pieces = 300 cycles on
    // each compute wave - so the wave that issued a third of the pieces set the length of every interval.)
    u32x4 stg[NPIECE];
    auto ld = [&](int g) {
#pragma unroll
      for (int i = 0; i < NPIECE; ++i) stg[i] = *reinterpret_cast<const u32x4*>(wimg + (size_t)g * WSET + i * 1024 + lane * 16);
    };
    auto wr = [&](int set) {
#pragma unroll
      for (int i = 0; i < NPIECE; ++i) *reinterpret_cast<u32x4*>(smem + OFF_WS + set * WSET + i * 1024 + lane * 16) = stg[i];
    };
    ld(0);
    wr(0);
    ld(1);
    s1_bar_lds();
    int gg = 0;
#pragma unroll 1
    for (int b = first; b < n_img; b += stride) {
      const bool has_next = b + stride < n_img;
#pragma unroll 1
      for (int g = 0; g < G; ++g, ++gg) {
        if (g + 1 < G || has_next) {
          wr((gg + 1) & 1);                                   // that set's last readers finished before the previous barrier
          if (g + 2 < G || has_next) ld((g + 2) & (G - 1));
        }
        __builtin_amdgcn_s_barrier();
        s1_bar_lds();
      }
    }
    return;
  }

  // ================================================================ compute waves: token tiles 2w, 2w+1
  const int mt0 = 2 * w;
  // conv1 B fragments of the wave's two token tiles (all of K = 128), straight from global memory: lane = (token lrow, 8 channels (4 kc + lq) * 8 ..)
  auto load_x = [&](int b, u32x4 (&xf)[2][4]) {
    const bf16* xin = x + (size_t)b * 400 * C1;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int tk = (mt0 + m) * 16 + lrow;
      const int gt = tk < OT ? r0 * W + tk : hrow * W + (tk - OT);      // image token
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
        xf[m][kc] = tk < XT ? *reinterpret_cast<const u32x4*>(xin + (size_t)gt * C1 + (kc * 4 + lq) * 8) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  u32x4 xr[2][4], xn[2][4];
  if (first < n_img) load_x(first, xr);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) xn[m][kc] = u32x4{0u, 0u, 0u, 0u};
  int hp[2];                                           // H1 pixel (top-left tap) of this lane's output token
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    int tk = (mt0 + m) * 16 + lrow; tk = tk < OT ? tk : OT - 1;      // padded output rows recompute token 199 (ignored later)
    hp[m] = (tk / W) * PW + tk % W;
  }
  int h1pix[2];                                        // H1 pixel this lane's conv1 token is stored to (-1: pad token)
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int tk = (mt0 + m) * 16 + lrow;
    const int pr = tk / W, pc = tk - pr * W;
    const int hr = pr < 10 ? pr + 1 : (hsel ? 0 : 11);              // H1 rows 1..10 = the output rows; the halo row above (lower half) / below them
    h1pix[m] = tk < XT ? hr * PW + pc + 1 : -1;
  }
  // residual: y = x + ...  The output accumulators start as E_p . X^T, E_p[r][k] = (k == 16 p + r): an exact copy of the wave's own x fragments
  // into the accumulator layout (channel 16 n + r <- chunk n >> 1, k = 16 (n & 1) + r), 16 MFMAs per tile instead of an LDS copy of x.
  u32x4 eye[2];
  {
    const unsigned one = (unsigned)__builtin_bit_cast(unsigned short, (bf16)1.0f) << (16 * (lrow & 1));
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      eye[p] = u32x4{0u, 0u, 0u, 0u};
      if (lq == 2 * p + (lrow >> 3)) eye[p][(lrow & 7) >> 1] = one;
    }
  }
  s1_bar_lds();
#ifdef S1_CLK
  ckP = __builtin_readcyclecounter() - ck0; ckl = __builtin_readcyclecounter();
#endif

  int gg = 0;
#pragma unroll 1
  for (int b = first; b < n_img; b += stride) {
    const bool has_next = b + stride < n_img;
    f32x4 acc[2][8];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 8; ++n) acc[m][n] = mma_chunk<bf16>(eye[n & 1], xr[m][n >> 1], f32x4{0.f, 0.f, 0.f, 0.f});

#pragma unroll 1
    for (int g = 0; g < G; ++g, ++gg) {
      const unsigned char* const Wb = smem + OFF_WS + (gg & 1) * WSET;
      if (g == 2 && has_next) load_x(b + stride, xn);     // the next image's tokens: in flight for two groups
#ifdef S1_CLK
      ckA1 += __builtin_readcyclecounter() - ckl;
#endif
      // ---- interval A: P1  H1g = GELU(conv1 + bias)
      {
        const unsigned char* wr = Wb + WS_W1 + lq * 512 + lrow * 16;
        f32x4 a[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m) a[m][0] = a[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          const u32x4 wf0 = *reinterpret_cast<const u32x4*>(wr + kc * 2048);
          const u32x4 wf1 = *reinterpret_cast<const u32x4*>(wr + kc * 2048 + 256);
          a[0][0] = mma_chunk<bf16>(wf0, xr[0][kc], a[0][0]);
          a[0][1] = mma_chunk<bf16>(wf1, xr[0][kc], a[0][1]);
          a[1][0] = mma_chunk<bf16>(wf0, xr[1][kc], a[1][0]);
          a[1][1] = mma_chunk<bf16>(wf1, xr[1][kc], a[1][1]);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const f32x4 bias = *reinterpret_cast<const f32x4*>(B1s + g * CG + nt * 16 + lq * 4);
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            if (h1pix[m] >= 0) {
              const f32x4 v = a[m][nt] + bias;
              const f32x2 g0 = S1_GELU2((f32x2{v[0], v[1]})), g1 = S1_GELU2((f32x2{v[2], v[3]}));
              const bf16x4 o = {(bf16)g0[0], (bf16)g0[1], (bf16)g1[0], (bf16)g1[1]};
              *reinterpret_cast<bf16x4*>(H1 + (nt * 2 + (lq >> 1)) * H1_PLANE + h1pix[m] * 16 + (lq & 1) * 8) = o;
            }
          }
        }
      }
#ifdef S1_CLK
      ckA2 += __builtin_readcyclecounter() - ckl;
#endif
      s1_bar_lds();
#ifdef S1_CLK
      { long long c = __builtin_readcyclecounter(); ckA += c - ckl; ckl = c; }
#endif
      // ---- interval B: P2  H2g = GELU(grouped 3x3 conv of H1g)  ->  P3  acc += W3[:, g] . H2g^T
      {
        const unsigned char* wr = Wb + WS_W2 + lq * 512 + lrow * 16;
        const unsigned char* h0 = H1 + lq * H1_PLANE + hp[0] * 16;
        const unsigned char* h1 = H1 + lq * H1_PLANE + hp[1] * 16;
        f32x4 a[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m) a[m][0] = a[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int toff = ((tap / 3) * PW + tap % 3) * 16;
          const u32x4 wf0 = *reinterpret_cast<const u32x4*>(wr + tap * 2048);
          const u32x4 wf1 = *reinterpret_cast<const u32x4*>(wr + tap * 2048 + 256);
          const u32x4 f0 = *reinterpret_cast<const u32x4*>(h0 + toff);
          const u32x4 f1 = *reinterpret_cast<const u32x4*>(h1 + toff);
          a[0][0] = mma_chunk<bf16>(wf0, f0, a[0][0]);
          a[0][1] = mma_chunk<bf16>(wf1, f0, a[0][1]);
          a[1][0] = mma_chunk<bf16>(wf0, f1, a[1][0]);
          a[1][1] = mma_chunk<bf16>(wf1, f1, a[1][1]);
        }
        u32x4 pb[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const f32x2 g00 = S1_GELU2((f32x2{a[m][0][0], a[m][0][1]})), g01 = S1_GELU2((f32x2{a[m][0][2], a[m][0][3]}));
          const f32x2 g10 = S1_GELU2((f32x2{a[m][1][0], a[m][1][1]})), g11 = S1_GELU2((f32x2{a[m][1][2], a[m][1][3]}));
          const bf16x8 o = {(bf16)g00[0], (bf16)g00[1], (bf16)g01[0], (bf16)g01[1], (bf16)g10[0], (bf16)g10[1], (bf16)g11[0], (bf16)g11[1]};
          pb[m] = __builtin_bit_cast(u32x4, o);
        }
        const unsigned char* w3r = Wb + WS_W3 + lq * 2048 + lrow * 16;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          const u32x4 wf = *reinterpret_cast<const u32x4*>(w3r + n * 256);
          acc[0][n] = mma_chunk<bf16>(wf, pb[0], acc[0][n]);
          acc[1][n] = mma_chunk<bf16>(wf, pb[1], acc[1][n]);
        }
      }
#ifdef S1_CLK
      ckB2 += __builtin_readcyclecounter() - ckl;
#endif
      s1_bar_lds();
#ifdef S1_CLK
      { long long c = __builtin_readcyclecounter(); ckB += c - ckl; ckl = c; }
#endif
    }

    // ---- y (the residual is already in acc); lane holds channels 16 n + 4 lq .. +3 of token mt*16 + lrow
    bf16* yout = y + ((size_t)b * 400 + r0 * W) * C1 + lq * 4;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int tk = (mt0 + m) * 16 + lrow;
      if (tk < OT) {
#pragma unroll
        for (int n = 0; n < 8; ++n) store4<bf16>(yout + (size_t)tk * C1 + n * 16, acc[m][n]);
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) xr[m][kc] = xn[m][kc];
#ifdef S1_CLK
    { long long c = __builtin_readcyclecounter(); ckE += c - ckl; ckl = c; }
#endif
  }
#ifdef S1_CLK
  if (t == 0 && (blockIdx.x == 0 || blockIdx.x == 101)) {
    const long long c = __builtin_readcyclecounter();
    printf("[stage1 wg %d] total %lld  prologue %lld  intervals A %lld (dma issued %lld, computed %lld)  B %lld (computed %lld, - %lld)  epilogues %lld\n",
           (int)blockIdx.x, c - ck0, ckP, ckA, ckA1, ckA2, ckB, ckB2, ckB3, ckE);
  }
#endif
}

bool stage1_fused_supported(int dtype, int C1, int hid, int group, int H1) {
  return dtype == 1 && C1 == s1::C1 && hid == s1::HID && group == s1::G && H1 == s1::W;
}

size_t stage1_image_bytes() { return (size_t)s1::G * s1::WSET; }

int launch_stage1_pack(const void* w1, const void* w2, const void* w3, void* wimg, hipStream_t s) {
  const int n = s1::G * (s1::WSET / 16);
  hipLaunchKernelGGL(stage1_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const bf16*)w1, (const bf16*)w2, (const bf16*)w3, (u32x4*)wimg);
  return (int)hipGetLastError();
}

int launch_stage1_block(const void* x, void* y, const void* wimg, const float* b1, int B, hipStream_t s) {
  if (B <= 0) return 0;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)stage1_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, s1::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int grid = 2 * B < 256 ? 2 * B : 256;            // persistent: one workgroup per CU, a fixed image half each
  hipLaunchKernelGGL(stage1_block_kernel, dim3(grid), dim3(s1::NW * 64), s1::LDS_BYTES, s, (const bf16*)x, (bf16*)y, (const unsigned char*)wimg, b1, B);
  return (int)hipGetLastError();
}

}  // namespace FSVIT_NS
