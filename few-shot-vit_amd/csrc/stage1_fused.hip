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

#ifndef S1_MT
#define S1_MT 1
#endif
#ifndef S1_LEAD
#define S1_LEAD (S1_MT == 1 ? 3 : 8)        // fragment reads run this many MFMA steps ahead (MT = 1: 128-VGPR budget)
#endif
#ifndef S1_VPM
#define S1_VPM 5
#endif

namespace FSVIT_NS {

typedef __attribute__((ext_vector_type(2))) bf16 bf16x2;

namespace s1 {
constexpr int C1 = 128, HID = 256, G = 8, CG = 32, W = 20;
constexpr int MT = S1_MT;                 // token tiles per compute wave
constexpr int NCW = 14 / MT;              // compute waves (14 token tiles incl. the halo row)
constexpr int NHW = 3 - MT;               // weight waves (34 staging registers x 4 would not fit the 128-VGPR budget of a 16-wave workgroup)
constexpr int NW = NCW + NHW;             // 8 (MT = 2) or 16 (MT = 1) waves, one workgroup per CU
constexpr int NSTEP = 34 * MT;            // MFMAs per wave and interval
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
constexpr int WS_W3 = W1_BYTES + W2_BYTES;
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
__global__ __launch_bounds__(s1::NW * 64) void stage1_block_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, const unsigned char* __restrict__ wimg,
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

  if (w >= NCW) {
    // ================================================================ the weight wave(s)
    // Stream 34 KB per interval through their own registers: global_load_dwordx4 one interval ahead, ds_write_b128 into the buffers whose readers
    // passed the last barrier.  (LDS-DMA, global_load_lds_dwordx4, costs a wave ~100-200 cycles of issue per 1 KB piece whatever the source -
    // measured here with an L1-hot source: 13 pieces = 2400 cycles on one wave, 3 pieces = 300 cycles on each compute wave.)
    constexpr int PPW = NPIECE / NHW;                   // pieces per weight wave
    const int p0 = (w - NCW) * PPW;
    u32x4 stg[PPW];
    auto src_of = [&](int j, int i) -> const unsigned char* {      // piece i of the set written during interval j: W1(j+2), W2(j+1), W3(j-1)
      const int g = i < 8 ? j + 2 : i < 26 ? j + 1 : j - 1;
      return wimg + (size_t)(g & (G - 1)) * WSET + i * 1024 + lane * 16;
    };
    auto dst_of = [&](int j, int i) -> unsigned char* {
      return smem + (i < 8 ? OFF_W1 + (j & 1) * W1_BYTES + i * 1024
                     : i < 26 ? OFF_W2 + ((j + 1) & 1) * W2_BYTES + (i - 8) * 1024 : OFF_W3 + ((j + 1) & 1) * W3_BYTES + (i - 26) * 1024) + lane * 16;
    };
    auto ld = [&](int j) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) stg[i] = *reinterpret_cast<const u32x4*>(src_of(j, p0 + i));
    };
    auto wr = [&](int j) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) *reinterpret_cast<u32x4*>(dst_of(j, p0 + i)) = stg[i];
    };
    ld(-2);
    wr(-2);                               // W1(0) for the prologue interval
    ld(-1);
    s1_bar_lds();
#pragma unroll 1
    for (int k = -1; k <= k_end; ++k) {
#ifndef S1_NO_WEIGHTS       // timing diagnostics only
#ifndef S1_NO_WWRITE
      wr(k);
#endif
#ifndef S1_NO_WLOAD
      ld(k + 1);
#endif
#endif
      s1_bar_lds();
    }
    return;
  }

  // ================================================================ compute waves: token tiles MT w .. MT w + MT - 1
  const int mt0 = MT * w;
  // conv1 B fragments of the wave's token tiles (all of K = 128), straight from global memory: lane = (token lrow, 8 channels (4 kc + lq) * 8 ..)
  u32x4 xr[MT][4];
  auto load_x = [&](int tile) {
    int b = first + tile * stride;
    b = b < n_img ? b : (first < n_img ? first : 0);              // past the end: any valid image (the result is never stored)
    const bf16* xin = x + (size_t)b * 400 * C1;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int tk = (mt0 + m) * 16 + lrow;
      const int gt = tk < OT ? r0 * W + tk : hrow * W + (tk - OT);      // image token
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
        xr[m][kc] = tk < XT ? *reinterpret_cast<const u32x4*>(xin + (size_t)gt * C1 + (kc * 4 + lq) * 8) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  load_x(0);
  int hoff[MT];                                        // P2: byte offset (in an H1 buffer) of the top-left tap pixel of this lane's output token, plane lq
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    int tk = (mt0 + m) * 16 + lrow; tk = tk < OT ? tk : OT - 1;      // padded output rows recompute token 199 (ignored later)
    hoff[m] = lq * H1_PLANE + ((tk / W) * PW + tk % W) * 16;
  }
  int h1off[MT];                                       // P1: byte offset this lane's conv1 token is stored to (tile nt adds 2 planes)
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int tk = (mt0 + m) * 16 + lrow;
    const int pr = tk / W, pc = tk - pr * W;
    const int hr = pr < 10 ? pr + 1 : (hsel ? 0 : 11);              // H1 rows 1..10 = the output rows; the halo row above (lower half) / below them
    const int pix = tk < XT ? hr * PW + pc + 1 : DUMMY_PIX;
    h1off[m] = (lq >> 1) * H1_PLANE + pix * 16 + (lq & 1) * 8;
  }
  // residual: y = x + ...  The output accumulators start as E_p . X^T, E_p[r][k] = (k == 16 p + r): an exact copy of the wave's own x fragments
  // into the accumulator layout (channel 16 n + r <- chunk n >> 1, k = 16 (n & 1) + r), 8 MFMAs per tile instead of an LDS copy of x.
  const unsigned eye_one = (unsigned)__builtin_bit_cast(unsigned short, (bf16)1.0f) << (16 * (lrow & 1));
  auto eye = [&](int p) {
    u32x4 e = {0u, 0u, 0u, 0u};
    if (lq == 2 * p + (lrow >> 3)) e[(lrow & 7) >> 1] = eye_one;
    return e;
  };
  f32x4 acc[MT][8], a2[MT][2];
  u32x4 pb[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    pb[m] = u32x4{0u, 0u, 0u, 0u};
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

    // ---- the interval as 34 MT pinned steps: MFMA i, the fragment reads of MFMA i + LEAD, and a slice of the two GELU passes.  (Left to itself - and
    // with sched_group_barrier patterns too - hipcc issues the MFMAs first and the VALU after them; the waves of a SIMD pass the same barrier, so
    // their MFMA blocks collide and their VALU blocks collide.)  With S = 2 MT:
    //   MFMA        0 ..  4S-1  P3(k-2): acc[m][n] += W3[n] . pb[m]                  i = MT n + m
    //   MFMA       4S ..  8S-1  P1(k+1): a1[m][nt] += W1[kc][nt] . x[m][kc]          i = 4S + S kc + 2 m + nt
    //   MFMA       8S .. 17S-1  P2(k):   a2[m][nt] += W2[tap][nt] . H1[tap][m]       i = 8S + S tap + 2 m + nt
    //   GELU: one value per step - steps 4S .. 8S-1: a2 (= P2(k-1)) -> pb, dead as P3's operand by then and P2(k) restarts a2 only at step 8S;
    //         steps 8S .. 12S-1: a1 + bias -> H1[(k+1) & 1]
    constexpr int S = 2 * MT;
    f32x4 a1[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
      asm volatile("" : "+v"(a2[m][0]), "+v"(a2[m][1]));      // opaque: keeps hipcc from computing GELU(a2) at the end of the previous interval
    u32x4 f1[4][2], f2w[9][2], f2h[9][MT], f3[8];
    f32x4 bv[2];
    float ga[2];
    unsigned hb[2];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto reads_of = [&](auto ic) {                        // the fragment reads first used by MFMA I
      constexpr int I = decltype(ic)::value;
      if constexpr (I < 4 * S) {
        if constexpr (I % MT == 0) f3[I / MT] = *reinterpret_cast<const u32x4*>(W3r + (I / MT) * 256);
      } else if constexpr (I < 8 * S) {
        constexpr int J = I - 4 * S;
        if constexpr (J % S < 2) f1[J / S][J % S] = *reinterpret_cast<const u32x4*>(W1r + (J / S) * 2048 + (J % S) * 256);
      } else if constexpr (I < 17 * S) {
        constexpr int J = I - 8 * S, TAP = J / S, R = J % S, TOFF = ((TAP / 3) * PW + TAP % 3) * 16;
        if constexpr (R < 2) f2w[TAP][R] = *reinterpret_cast<const u32x4*>(W2r + TAP * 2048 + R * 256);
        if constexpr ((R & 1) == 0) f2h[TAP][R >> 1] = *reinterpret_cast<const u32x4*>(H1r + hoff[R >> 1] + TOFF);
      }
      if constexpr (I == 8 * S) bv[0] = *reinterpret_cast<const f32x4*>(bias);
      if constexpr (I == 10 * S) bv[1] = *reinterpret_cast<const f32x4*>(bias + 16);
    };
    static_for<S1_LEAD>([&](auto ic) { reads_of(ic); });
    __builtin_amdgcn_sched_barrier(0);
    static_for<NSTEP>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      if constexpr (I < 4 * S) {
        constexpr int N = I / MT, M = I % MT;
        acc[M][N] = mma_chunk<bf16>(f3[N], pb[M], acc[M][N]);
      } else if constexpr (I < 8 * S) {
        constexpr int J = I - 4 * S, KC = J / S, M = (J % S) >> 1, NT = J & 1;
        a1[M][NT] = mma_chunk<bf16>(f1[KC][NT], xr[M][KC], KC == 0 ? zero4 : a1[M][NT]);
      } else {
        constexpr int J = I - 8 * S, TAP = J / S, M = (J % S) >> 1, NT = J & 1;
        a2[M][NT] = mma_chunk<bf16>(f2w[TAP][NT], f2h[TAP][M], TAP == 0 ? zero4 : a2[M][NT]);
      }
      reads_of(std::integral_constant<int, I + S1_LEAD>{});
      if constexpr (I >= 4 * S && I < 8 * S) {            // a2 of the previous interval: m = E / 8, tile nt = (E / 4) & 1, element E & 3
        constexpr int E = I - 4 * S, M = E >> 3, Q = E & 7;
        ga[Q & 1] = S1_GELU(a2[M][Q >> 2][Q & 3]);
        asm volatile("" : "+v"(ga[Q & 1]));              // pinned here: LLVM's sink pass would move the whole pass to the block that uses pb (the loop latch)
        if constexpr (Q & 1) {
          const bf16x2 o = {(bf16)ga[0], (bf16)ga[1]};
          pb[M][Q >> 1] = __builtin_bit_cast(unsigned, o);
          asm volatile("" : "+v"(pb[M][Q >> 1]));
        }
      } else if constexpr (I >= 8 * S && I < 12 * S) {    // a1 + bias: tile nt = D / (4 MT), m = (D / 4) % MT, element D & 3
        constexpr int D = I - 8 * S, NT = D / (4 * MT), M = (D >> 2) % MT, EL = D & 3;
        ga[EL & 1] = S1_GELU(a1[M][NT][EL] + bv[NT][EL]);
        if constexpr (EL & 1) {
          const bf16x2 o = {(bf16)ga[0], (bf16)ga[1]};
          hb[EL >> 1] = __builtin_bit_cast(unsigned, o);
        }
        if constexpr (EL == 3) *reinterpret_cast<u32x2*>(H1w + NT * 2 * H1_PLANE + h1off[M]) = u32x2{hb[0], hb[1]};
      }
      __builtin_amdgcn_sched_barrier(0);
    });

    if ((k & 7) == 1) {                   // P3 of a tile's last group is in: store it (k > 1), seed the accumulators of the current tile with its residual
      if (k > 1) {
        const int b = first + ((k - 2) >> 3) * stride;
        bf16* yout = y + ((size_t)b * 400 + r0 * W) * C1 + lq * 4;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int tk = (mt0 + m) * 16 + lrow;
          if (tk < OT) {
#pragma unroll
            for (int n = 0; n < 8; ++n) store4<bf16>(yout + (size_t)tk * C1 + n * 16, acc[m][n]);
          }
        }
      }
      const u32x4 e0 = eye(0), e1 = eye(1);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[m][n] = mma_chunk<bf16>((n & 1) ? e1 : e0, xr[m][n >> 1], zero4);
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
  {    // per launch: the attribute is per DEVICE (a process-wide "done" flag skipped it on a second GPU), and the call is cheap
    hipError_t e = hipFuncSetAttribute((const void*)stage1_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, s1::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
  }
  const int grid = 2 * B < 256 ? 2 * B : 256;            // persistent: one workgroup per CU, a fixed image half each
  hipLaunchKernelGGL(stage1_block_kernel, dim3(grid), dim3(s1::NW * 64), s1::LDS_BYTES, s, (const bf16*)x, (bf16*)y, (const unsigned char*)wimg, b1, B);
  return (int)hipGetLastError();
}

}  // namespace FSVIT_NS
