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

#include <utility>

#include "fsvit_common.h"
#include "kernels.h"

// GELU of a register pair: packed (gelu_sig2) or two scalar gelu_sig (-DS1_SCALAR_GELU: packed fp32 does not issue behind MFMAs)
#if defined(S1_NO_GELU)      // timing diagnostics only
#define S1_GELU(v) (v)
#else
#define S1_GELU(v) gelu_sig(v)
#endif
#if defined(S1_NO_GELU)
#define S1_GELU2(v) (v)
#elif defined(S1_SCALAR_GELU)
#define S1_GELU2(v) (f32x2{gelu_sig((v)[0]), gelu_sig((v)[1])})
#else
#define S1_GELU2(v) gelu_sig2(v)
#endif

#ifndef S1_LEAD
#define S1_LEAD 8
#endif
#ifndef S1_VPM
#define S1_VPM 5
#endif

namespace FSVIT_NS {

namespace s1 {
constexpr int C1 = 128, HID = 256, G = 8, CG = 32, W = 20;
constexpr int NW = 8;                     // waves per workgroup: 7 compute waves + the weight wave (one workgroup per CU)
constexpr int XT = 220;                   // input tokens (10 rows + one halo row): 14 m-tiles, the last 4 tokens empty
constexpr int OT = 200;                   // output tokens (10 rows): 13 m-tiles, the last half empty
constexpr int PW = 36;                    // pitch of the zero-bordered H1 pixel grid (12 rows; columns -1 .. 20 used).  36 = 20 + 16: the 16 tokens of an
                                          // m-tile usually straddle a row end, and with the natural pitch 22 the tokens after the wrap land on the 16-byte
                                          // slots (mod 16) of the ones before it - 6.5 LDS cycles per ds_read_b128 instead of 4 (tools/lds_conflicts.py)
constexpr int DUMMY_PIX = 30;             // row 0, column 29: never read (pad tokens store here instead of branching)
constexpr int H1_PLANE = 12 * PW * 16;    // 6912 = 27 * 256: the plane stride stays 0 mod 256 B
constexpr int H1_BYTES = 4 * H1_PLANE;    // 27648
constexpr int W1_BYTES = 16 * 32 * 16;    //  8192  [16 k-chunks][32 n][16 B]
constexpr int W2_BYTES = 36 * 32 * 16;    // 18432  [9 taps * 4 k-chunks][32 n][16 B], row n = hidden channel sigma(n)
constexpr int W3_BYTES = 4 * 128 * 16;    //  8192  [4 k-chunks][128 n][16 B]
constexpr int WS_W2 = W1_BYTES, WS_W3 = W1_BYTES + W2_BYTES;
constexpr int WSET = WS_W3 + W3_BYTES;    // 34816: one group of the weight image = 34 pieces of 1 KB
constexpr int NPIECE = WSET / 1024;
constexpr int OFF_H1 = 0;                          // two H1 buffers
constexpr int OFF_W1 = OFF_H1 + 2 * H1_BYTES;      //  55296  two buffers each of W1, W2, W3 (their read intervals differ, see the kernel)
constexpr int OFF_W2 = OFF_W1 + 2 * W1_BYTES;      //  71680
constexpr int OFF_W3 = OFF_W2 + 2 * W2_BYTES;      // 108544
constexpr int OFF_B1 = OFF_W3 + 2 * W3_BYTES;      // 124928  conv1 folded bias, 256 fp32
constexpr int LDS_BYTES = OFF_B1 + HID * 4;        // 125952
constexpr int KW2 = 320;                  // packed conv2 row length (9*32 = 288 rounded up to the 64-element K slice)
}  // namespace s1

template <int... I, typename F>
__device__ __forceinline__ void s1_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { s1_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

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

// Software pipeline over the workgroup's group sequence k = 8 * tile + g (one barrier per group).  Interval k:
//   matrix pipe:  P1(k+1)  H1 pre-activation of the NEXT group (16 MFMAs, x fragments from registers)
//                 P2(k)    grouped 3x3 conv of this group from H1[k & 1]                (36 MFMAs)
//                 P3(k-2)  conv3 of the group before last from its packed GELU'd tiles  (16 MFMAs)
//   VALU:         GELU(P2(k-1)) -> packed B fragments for P3 in the next interval;  bias + GELU(P1(k+1)) -> H1[(k+1) & 1]
// so every MFMA of an interval has its operands ready when the interval starts (or after the first 16 MFMAs) and the two GELU passes run in the
// shadow of the 68 MFMAs instead of between them (v3, two barriers per group and phases in series: 6.7k cycles per group for 2.2k cycles of MFMA and
// ~2.5k cycles of VALU per SIMD).  The tile's output is stored, and the accumulators re-seeded with the residual, at k & 7 == 1, after P3 of the
// previous tile's last group; the next tile's x fragments are loaded at k & 7 == 6 after P1 has consumed the current ones.
// Weight buffers (read in interval k: W1(k+1), W2(k), W3(k-2)): the weight wave writes W1(k+2), W2(k+1), W3(k-1) during interval k into the other
// buffer of each kind, from registers it filled one interval earlier.
__global__ __launch_bounds__(512) void stage1_block_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, const unsigned char* __restrict__ wimg,
                                                           const float* __restrict__ b1, int n_img) {
  using namespace s1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const B1s = reinterpret_cast<float*>(smem + OFF_B1);

  const int t = threadIdx.x, lane = t & 63;
#ifdef S1_CLK
  long long ck0 = __builtin_readcyclecounter(), ckW = 0, ckB = 0, ckl = 0;
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
  const int n_tiles = first < n_img ? (n_img - first + stride - 1) / stride : 0;
  const int k_end = 8 * n_tiles + 1;                  // intervals k = -1 .. k_end
  // Token order inside the workgroup: tokens 0..199 = the 10 output rows r0 .. r0+9, tokens 200..219 = the halo row (image row 10 for the upper
  // half, 9 for the lower): output token == input token, so the x fragments a wave holds for conv1 are also its residual.
  const int r0 = hsel * 10;                           // first output row
  const int hrow = hsel ? 9 : 10;                     // halo row

  // ---- once per workgroup: both H1 buffers zeroed (their borders must stay 0), the bias table
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = t; i < (2 * H1_BYTES) / 16; i += NW * 64) *reinterpret_cast<u32x4*>(smem + OFF_H1 + i * 16) = z;
    if (t < HID) B1s[t] = b1[t];        // bias table: an in-loop global load would cost a full round trip per tile
  }

  if (w == NW - 1) {
    // ================================================================ the weight wave
    // Streams 34 KB per interval through its own registers: global_load_dwordx4 x 34 one interval ahead, ds_write_b128 x 34 into the buffers whose
    // readers passed the last barrier.  (LDS-DMA, global_load_lds_dwordx4, costs a wave ~100-200 cycles of issue per 1 KB piece whatever the source -
    // measured here with an L1-hot source: 13 pieces = 2400 cycles on this wave, 3 pieces = 300 cycles on each compute wave.)
    u32x4 stg[NPIECE];
    auto ld = [&](int j) {               // the set written during interval j: W1(j+2), W2(j+1), W3(j-1)
      const unsigned char* p1 = wimg + (size_t)((j + 2) & (G - 1)) * WSET + lane * 16;
      const unsigned char* p2 = wimg + (size_t)((j + 1) & (G - 1)) * WSET + WS_W2 + lane * 16;
      const unsigned char* p3 = wimg + (size_t)((j - 1) & (G - 1)) * WSET + WS_W3 + lane * 16;
#pragma unroll
      for (int i = 0; i < 8; ++i) stg[i] = *reinterpret_cast<const u32x4*>(p1 + i * 1024);
#pragma unroll
      for (int i = 0; i < 18; ++i) stg[8 + i] = *reinterpret_cast<const u32x4*>(p2 + i * 1024);
#pragma unroll
      for (int i = 0; i < 8; ++i) stg[26 + i] = *reinterpret_cast<const u32x4*>(p3 + i * 1024);
    };
    auto wr = [&](int j) {
      unsigned char* q1 = smem + OFF_W1 + (j & 1) * W1_BYTES + lane * 16;
      unsigned char* q2 = smem + OFF_W2 + ((j + 1) & 1) * W2_BYTES + lane * 16;
      unsigned char* q3 = smem + OFF_W3 + ((j + 1) & 1) * W3_BYTES + lane * 16;
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(q1 + i * 1024) = stg[i];
#pragma unroll
      for (int i = 0; i < 18; ++i) *reinterpret_cast<u32x4*>(q2 + i * 1024) = stg[8 + i];
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(q3 + i * 1024) = stg[26 + i];
    };
    ld(-2);
    wr(-2);                               // W1(0) for the prologue interval
    ld(-1);
    s1_bar_lds();
#pragma unroll 1
    for (int k = -1; k <= k_end; ++k) {
#ifdef S1_CLK
      ckl = __builtin_readcyclecounter();
#endif
      wr(k);
#ifdef S1_CLK
      ckB += __builtin_readcyclecounter() - ckl;
#endif
      ld(k + 1);
#ifdef S1_CLK
      ckW += __builtin_readcyclecounter() - ckl;
#endif
      s1_bar_lds();
    }
#ifdef S1_CLK
    if (lane == 0 && blockIdx.x == 0) printf("[stage1 weight wave] written after %lld, loads issued after %lld (sums over %d intervals)\n", ckB, ckW, k_end + 2);
#endif
    return;
  }

  // ================================================================ compute waves: token tiles 2w, 2w+1
  const int mt0 = 2 * w;
  // conv1 B fragments of the wave's two token tiles (all of K = 128), straight from global memory: lane = (token lrow, 8 channels (4 kc + lq) * 8 ..)
  u32x4 xr[2][4];
  auto load_x = [&](int tile) {
    int b = first + tile * stride;
    b = b < n_img ? b : (first < n_img ? first : 0);              // past the end: any valid image (the result is never stored)
    const bf16* xin = x + (size_t)b * 400 * C1;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int tk = (mt0 + m) * 16 + lrow;
      const int gt = tk < OT ? r0 * W + tk : hrow * W + (tk - OT);      // image token
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
        xr[m][kc] = tk < XT ? *reinterpret_cast<const u32x4*>(xin + (size_t)gt * C1 + (kc * 4 + lq) * 8) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  load_x(0);
  int hoff[2];                                         // P2: byte offset (in an H1 buffer) of the top-left tap pixel of this lane's output token, plane lq
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    int tk = (mt0 + m) * 16 + lrow; tk = tk < OT ? tk : OT - 1;      // padded output rows recompute token 199 (ignored later)
    hoff[m] = lq * H1_PLANE + ((tk / W) * PW + tk % W) * 16;
  }
  int h1off[2];                                        // P1: byte offset this lane's conv1 token is stored to (tile nt adds 2 planes)
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int tk = (mt0 + m) * 16 + lrow;
    const int pr = tk / W, pc = tk - pr * W;
    const int hr = pr < 10 ? pr + 1 : (hsel ? 0 : 11);              // H1 rows 1..10 = the output rows; the halo row above (lower half) / below them
    const int pix = tk < XT ? hr * PW + pc + 1 : DUMMY_PIX;
    h1off[m] = (lq >> 1) * H1_PLANE + pix * 16 + (lq & 1) * 8;
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
  f32x4 acc[2][8], a2[2][2];
  u32x4 pbq[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    pbq[m] = u32x4{0u, 0u, 0u, 0u};
    a2[m][0] = a2[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  s1_bar_lds();
#ifdef S1_CLK
  ckl = __builtin_readcyclecounter();
#endif

#pragma unroll 1
  for (int k = -1; k <= k_end; ++k) {
    const int par = k & 1;
    const unsigned char* const W1r = smem + OFF_W1 + (par ^ 1) * W1_BYTES + lq * 512 + lrow * 16;
    const unsigned char* const W2r = smem + OFF_W2 + par * W2_BYTES + lq * 512 + lrow * 16;
    const unsigned char* const W3r = smem + OFF_W3 + par * W3_BYTES + lq * 2048 + lrow * 16;
    const unsigned char* const H1r = smem + OFF_H1 + par * H1_BYTES;
    unsigned char* const H1w = smem + OFF_H1 + (par ^ 1) * H1_BYTES;
    const float* const bias = B1s + ((k + 1) & (G - 1)) * CG + lq * 4;

    // ---- the interval as 68 pinned steps: MFMA i, the fragment reads of MFMA i + LEAD, and a slice of the two GELU passes.  (Left to itself - and
    // with sched_group_barrier patterns too - hipcc issues the MFMAs first and the ~400 VALU after them; the two waves of a SIMD pass the same barrier,
    // so their MFMA blocks collide and their VALU blocks collide: measured 5-7k cycles of own work per wave and interval.)
    //   MFMA  0..15  P1(k+1): a1[m][nt] += W1[kc][nt] . x[m][kc]          i = 4 kc + 2 m + nt
    //   MFMA 16..51  P2(k):   a2n[m][nt] += W2[tap][nt] . H1[tap][m]      i = 16 + 4 tap + 2 m + nt
    //   MFMA 52..67  P3(k-2): acc[m][n] += W3[n] . pbq[m]                 i = 52 + 2 n + m
    //   GELU eval e after MFMA 2e: e < 16 of a2 (P2(k-1), -> pbn), e >= 16 of a1 + bias (-> H1[(k+1) & 1])
    f32x4 a1[2][2], a2n[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      a1[m][0] = a1[m][1] = a2n[m][0] = a2n[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      asm volatile("" : "+v"(a2[m][0]), "+v"(a2[m][1]));      // opaque: keeps hipcc from computing GELU(a2) at the end of the previous interval
    }
    u32x4 f1[4][2], f2w[9][2], f2h[9][2], f3[8];
    f32x4 bv[2];
    float ga[2][8];                                      // GELU'd a2, [m][4 nt + e]
    float gb[4];
    u32x4 pbn[2];
    auto reads_of = [&](auto ic) {                        // the fragment reads first used by MFMA I
      constexpr int I = decltype(ic)::value;
      if constexpr (I < 16) {
        if constexpr ((I & 2) == 0) f1[I >> 2][I & 1] = *reinterpret_cast<const u32x4*>(W1r + (I >> 2) * 2048 + (I & 1) * 256);
      } else if constexpr (I < 52) {
        constexpr int J = I - 16, TAP = J >> 2, R = J & 3, TOFF = ((TAP / 3) * PW + TAP % 3) * 16;
        if constexpr (R == 0) {
          f2w[TAP][0] = *reinterpret_cast<const u32x4*>(W2r + TAP * 2048);
          f2h[TAP][0] = *reinterpret_cast<const u32x4*>(H1r + hoff[0] + TOFF);
        } else if constexpr (R == 1) {
          f2w[TAP][1] = *reinterpret_cast<const u32x4*>(W2r + TAP * 2048 + 256);
        } else if constexpr (R == 2) {
          f2h[TAP][1] = *reinterpret_cast<const u32x4*>(H1r + hoff[1] + TOFF);
        }
      } else if constexpr (I < 68) {
        if constexpr (((I - 52) & 1) == 0) f3[(I - 52) >> 1] = *reinterpret_cast<const u32x4*>(W3r + ((I - 52) >> 1) * 256);
      }
      if constexpr (I == 30) { bv[0] = *reinterpret_cast<const f32x4*>(bias); bv[1] = *reinterpret_cast<const f32x4*>(bias + 16); }
    };
    static_for<S1_LEAD>([&](auto ic) { reads_of(ic); });
    __builtin_amdgcn_sched_barrier(0);
    static_for<68>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      if constexpr (I < 16) {
        constexpr int KC = I >> 2, M = (I >> 1) & 1, NT = I & 1;
        a1[M][NT] = mma_chunk<bf16>(f1[KC][NT], xr[M][KC], a1[M][NT]);
      } else if constexpr (I < 52) {
        constexpr int J = I - 16, TAP = J >> 2, M = (J >> 1) & 1, NT = J & 1;
        a2n[M][NT] = mma_chunk<bf16>(f2w[TAP][NT], f2h[TAP][M], a2n[M][NT]);
      } else {
        constexpr int J = I - 52, N = J >> 1, M = J & 1;
        acc[M][N] = mma_chunk<bf16>(f3[N], pbq[M], acc[M][N]);
      }
      reads_of(std::integral_constant<int, I + S1_LEAD>{});
#ifdef S1_SCALAR_GELU
      if constexpr ((I & 1) == 0 && I < 64) {
        constexpr int E = I >> 1;
        if constexpr (E < 16) {                           // a2 of the previous interval: m = E / 8, tile nt = (E / 4) & 1, element E & 3
          constexpr int M = E >> 3, Q = E & 7;
          ga[M][Q] = S1_GELU(a2[M][Q >> 2][Q & 3]);
          asm volatile("" : "+v"(ga[M][Q]));             // pinned here: LLVM's sink pass would move the whole pass to the block that uses pbn (the loop latch)
          if constexpr (Q == 7) {
            const bf16x8 o = {(bf16)ga[M][0], (bf16)ga[M][1], (bf16)ga[M][2], (bf16)ga[M][3], (bf16)ga[M][4], (bf16)ga[M][5], (bf16)ga[M][6], (bf16)ga[M][7]};
            pbn[M] = __builtin_bit_cast(u32x4, o);
            asm volatile("" : "+v"(pbn[M]));
          }
        } else {                                          // a1 + bias: tile nt = (E - 16) / 8, m = ((E - 16) / 4) & 1, element E & 3
          constexpr int D = E - 16, NT = D >> 3, M = (D >> 2) & 1, EL = D & 3;
          gb[EL] = S1_GELU(a1[M][NT][EL] + bv[NT][EL]);
          if constexpr (EL == 3) {
            const bf16x4 o = {(bf16)gb[0], (bf16)gb[1], (bf16)gb[2], (bf16)gb[3]};
            *reinterpret_cast<bf16x4*>(H1w + NT * 2 * H1_PLANE + h1off[M]) = o;
          }
        }
      }
#else
      // packed GELU on value pairs (a wave issues one VALU instruction per ~6 cycles whatever its width - tools/probes/valu_rates.hip - so the
      // v_pk_* forms halve the issue slots of the arithmetic): unit U after MFMA 4 U + 1
      if constexpr ((I & 3) == 1 && I < 64) {
        constexpr int U = I >> 2;
        if constexpr (U < 8) {                            // a2 of the previous interval: m = U / 4, elements 2 (U & 3), +1 of the 8 (tile nt = element / 4)
          constexpr int M = U >> 2, Q = 2 * (U & 3);
          f32x2 g = S1_GELU2((f32x2{a2[M][Q >> 2][Q & 3], a2[M][Q >> 2][(Q & 3) + 1]}));
          asm volatile("" : "+v"(g));                    // pinned here: LLVM's sink pass would move the whole pass to the block that uses pbn (the loop latch)
          ga[M][Q] = g[0]; ga[M][Q + 1] = g[1];
          if constexpr (Q == 6) {
            const bf16x8 o = {(bf16)ga[M][0], (bf16)ga[M][1], (bf16)ga[M][2], (bf16)ga[M][3], (bf16)ga[M][4], (bf16)ga[M][5], (bf16)ga[M][6], (bf16)ga[M][7]};
            pbn[M] = __builtin_bit_cast(u32x4, o);
            asm volatile("" : "+v"(pbn[M]));
          }
        } else {                                          // a1 + bias: tile nt = (U - 8) / 4, m = ((U - 8) / 2) & 1, elements 2 (U & 1), +1
          constexpr int D = U - 8, NT = D >> 2, M = (D >> 1) & 1, EL = 2 * (D & 1);
          const f32x2 g = S1_GELU2((f32x2{a1[M][NT][EL] + bv[NT][EL], a1[M][NT][EL + 1] + bv[NT][EL + 1]}));
          gb[EL] = g[0]; gb[EL + 1] = g[1];
          if constexpr (EL == 2) {
            const bf16x4 o = {(bf16)gb[0], (bf16)gb[1], (bf16)gb[2], (bf16)gb[3]};
            *reinterpret_cast<bf16x4*>(H1w + NT * 2 * H1_PLANE + h1off[M]) = o;
          }
        }
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      pbq[m] = pbn[m];
      a2[m][0] = a2n[m][0];
      a2[m][1] = a2n[m][1];
    }

    if ((k & 7) == 1) {                   // P3 of a tile's last group is in: store it (k > 1), seed the accumulators of the current tile with its residual
      if (k > 1) {
        const int b = first + ((k - 2) >> 3) * stride;
        bf16* yout = y + ((size_t)b * 400 + r0 * W) * C1 + lq * 4;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int tk = (mt0 + m) * 16 + lrow;
          if (tk < OT) {
#pragma unroll
            for (int n = 0; n < 8; ++n) store4<bf16>(yout + (size_t)tk * C1 + n * 16, acc[m][n]);
          }
        }
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[m][n] = mma_chunk<bf16>(eye[n & 1], xr[m][n >> 1], f32x4{0.f, 0.f, 0.f, 0.f});
    }
    if ((k & 7) == 6) load_x((k >> 3) + 1);        // P1 of this tile's last group has consumed xr: the next tile's tokens (one interval in flight)
#ifdef S1_CLK
    ckW += __builtin_readcyclecounter() - ckl;
#endif
    s1_bar_lds();
#ifdef S1_CLK
    { long long c = __builtin_readcyclecounter(); ckB += c - ckl; ckl = c; }
#endif
  }
#ifdef S1_CLK
  if (lane == 0 && blockIdx.x == 0) {
    const long long c = __builtin_readcyclecounter();
    printf("[stage1 wg %d wave %d] total %lld  intervals %lld (%d)  own work done after %lld\n", (int)blockIdx.x, w, c - ck0, ckB, k_end + 2, ckW);
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
