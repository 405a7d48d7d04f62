// 3x3 / stride 1 / pad 1 convolution with the input tile + halo RESIDENT IN LDS (stem conv2, conv3 of the Visformer,
// test_phase/models/visformer.py:211-213,224-237), bf16.  Round-2 design ("channel-half phases").
//
// A workgroup owns 8 output rows x 40 columns x 128 output channels.  Round 1 staged the whole 10 x 42-pixel halo tile of all input
// channels once per tile (107 KB at 128 channels) and walked the nine taps as shifted LDS reads of it.  rocprof / PMC of that kernel
// (profiles/r02_a_*): MFMA pipe 39 % busy, 33 % of the LDS cycles bank conflicts, 3.7 VALU per MFMA, and three structural costs:
//   (1) the next tile's halo could only be requested after the K loop (one buffer fills the LDS): ~9k of a tile's 56k cycles exposed;
//   (2) the XOR swizzle of the halo pixels made every fragment address depend on the tap: ~40 VALU per 40 MFMAs, issued between the
//       MFMA blocks, and hipcc's waitcnt pass put lgkmcnt(0) in front of the loop-carried first MFMA block, i.e. waited for the
//       fragment reads it was meant to overlap (found in the -save-temps ISA);
//   (3) with the 2x2-window-major pixel order of the pooled variant the 16 pixels of a fragment wrapped onto the same 16-byte slots.
// This version:
//   * K loop order (channel half, tap) instead of (tap, channel half): a PHASE is one 64-channel half of the input under all nine
//     taps, its halo tile 54 KB - TWO such buffers fit, so the next phase's halo (the other half of this tile, or the first half of
//     the workgroup's next tile) streams in under the current phase's 9 K tiles (one 1 KB LDS-DMA piece per wave and K tile, issued
//     behind the K tile's barrier so that no weight wait ever queues behind an HBM load that was just issued);
//   * the halo buffer is chunk-PLANAR, [8 k-chunks][10 rows x 44 pixels][16 B] (plane stride 0 mod 256 B), and an MFMA m-tile is a
//     4 x 4 pixel block: its 16 pixels sit on 16 different 16-byte slots whatever the tap shift (row pitch 44 = 12 mod 16), with no
//     XOR - a tap is a compile-time ADDITIVE offset, the nine taps of a phase are unrolled and every fragment read is
//     `ds_read_b128 base offset:imm` on five per-lane bases: no address arithmetic in the loop;
//   * fragments are not carried across the (runtime) phase loop: the first read of a phase is exposed once per 9 K tiles, and inside
//     the unrolled body the compiler's own counted lgkmcnt waits are exact;
//   * 16-byte stores: weight rows are fetched in the order 8 (r / 4) + 4 (tile & 1) + r % 4 inside each 32-channel block, so the
//     accumulators a lane holds of an n-tile pair are 8 consecutive channels (the gemm256 epilogue trick).
// Weights stream as before: 16 KB per 64-wide K tile through a 3-stage ring with counted vmcnt, two K tiles ahead.
// LDS: 2 x 56 KB halo + 3 x 16 KB weights = 160 KB exactly.
//   * FUSE_TAIL (conv3): the downsample/identity conv rides as one extra K tile whose A fragments come straight from the im2col rows
//     in global memory; LeakyReLU -> MaxPool2d(2) is a max over 4 adjacent lanes (the 4 x 4 block is enumerated 2x2-window-major) and
//     pos_embed1 is added in the same epilogue (same contract as conv_gemm_v2's x2 / pool2 / pos).
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "conv_gemm.h"
#include "fsvit_common.h"

namespace FSVIT_NS {

__device__ __attribute__((aligned(256))) unsigned char g_zero_page_halo[256];
typedef __attribute__((address_space(3))) void* lptrh_t;

__device__ __forceinline__ void hdma1(const void* gsrc, unsigned lds_byte_addr) {       // one 1 KiB LDS-DMA, per-lane 64-bit source
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
__device__ __forceinline__ void hdma2s(unsigned off0, unsigned off1, const void* sbase, unsigned lds0, unsigned lds1) {   // scalar base + 32-bit offsets
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %3\n\t"
      "s_mov_b32 m0, %5\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %3\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(off0), "v"(off1), "s"(sbase), "s"(lds0), "s"(lds1)
      : "memory");
}
__device__ __forceinline__ u32x4 hgload16(const void* p) {      // hidden from hipcc's waitcnt pass like the DMAs: counted by hand
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
#define HWAIT(vm) asm volatile("s_waitcnt vmcnt(" #vm ") lgkmcnt(0)" ::: "memory")
__device__ __forceinline__ void hbar() {
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int CTRL>
__device__ __forceinline__ float quad_xor(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

namespace halo {
constexpr int TR = 8, TW = 40;                 // output tile
constexpr int HR = TR + 2, HP = 44;            // halo rows, halo row pitch in pixels (42 used; 44 = 12 mod 16, see the header)
constexpr int PLANE = 7168;                    // one k-chunk plane: 448 16-byte slots (440 used) = 7 LDS-DMA pieces; 28 x 256 B
constexpr int NPIECE = 7;                      // pieces per plane = per wave (wave w fills plane w)
constexpr int HBUF = 8 * PLANE;                // 57344: one 64-channel halo buffer
constexpr int NST = 3, STAGE = 16384;          // weight ring
constexpr int OFF_W = 2 * HBUF;
constexpr int LDS_BYTES = OFF_W + NST * STAGE; // 163840 = 160 KiB
static_assert(LDS_BYTES == 160 * 1024, "the two halo buffers and the weight ring fill the LDS exactly");
}  // namespace halo

template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

template <int CIN, bool FUSE_TAIL, bool STATS = false>
__global__ __launch_bounds__(512, 2) void conv3x3_halo_kernel(const ConvGemmParams p, const int n_tiles) {
  using namespace halo;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NH = CIN / 64;                                  // phases (64-channel halves) per tile
  constexpr int NKM = 9 * NH, NKT = NKM + (FUSE_TAIL ? 1 : 0);  // weight K tiles per output tile (64 k each; the tail tile uses 32)
  constexpr int MT = 5, NT = 4;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = lane & 15, lq = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)(lptrh_t)smem;
  unsigned char* const wst0 = smem + OFF_W;

  const unsigned char* const Xb = reinterpret_cast<const unsigned char*>(p.x);
  const unsigned char* const Wb = reinterpret_cast<const unsigned char*>(p.w);
  bf16* __restrict__ Y = reinterpret_cast<bf16*>(p.y);
  const int tiles_per_img = p.H / TR;

  // ---- this lane's 5 output pixels: m-tile = 4 x 4 pixel block (wm * 5 + i), lane lrow = window * 4 + dy * 2 + dx inside it.
  // Pixel (row, column) inside the tile = (4 br + fr, 4 bc + fc) with the block position (br, bc) wave-uniform.
  const int fr = ((lrow >> 3) & 1) * 2 + ((lrow >> 1) & 1), fc = ((lrow >> 2) & 1) * 2 + (lrow & 1);
  auto blk_r = [&](int i) { return (wm * MT + i) / (TW / 4); };
  auto blk_c = [&](int i) { return (wm * MT + i) % (TW / 4); };
  f32x4 st0[STATS ? NT : 1], st1[STATS ? NT : 1];      // STATS: sum / sum of squares of this lane's 4 x 4 output channels over all its pixels
#pragma unroll
  for (int j = 0; j < (STATS ? NT : 1); ++j) st0[j] = st1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned xb[MT];                         // LDS byte offset of the tap-(0,0) halo pixel in plane lq of the CURRENT halo buffer
#pragma unroll
  for (int i = 0; i < MT; ++i) xb[i] = (unsigned)(((blk_r(i) * 4 + fr) * HP + blk_c(i) * 4 + fc) * 16 + lq * PLANE);
  // ---- weight DMA rows of this wave (128 rows of 128 bytes per K tile = 16 groups of 8); LDS row R holds output channel
  // (R & ~31) + 8 ((R & 15) >> 2) + 4 ((R >> 4) & 1) + (R & 3): see the 16-byte epilogue
  const int srow = lane >> 3;
  const unsigned schunk = (unsigned)(((lane & 7) ^ srow) << 4);
  unsigned offB[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int R = (2 * wave + j) * 8 + srow;
    const int ch = (R & ~31) + 8 * ((R & 15) >> 2) + 4 * ((R >> 4) & 1) + (R & 3);
    offB[j] = (unsigned)ch * (unsigned)(p.Kw * 2) + schunk;
  }
  const unsigned b_rd = (unsigned)((wn * 64 + lrow) * 128);
  const unsigned sw0 = (unsigned)((lq ^ (lrow & 7)) << 4), sw1 = (unsigned)(((4 + lq) ^ (lrow & 7)) << 4);

  // ---- halo pieces of this wave: plane `wave`, piece j = slots [64 j, 64 j + 64) of the plane; lane's slot -> halo pixel (hr, hc)
  // -> source byte offset relative to pixel (image b, row r0 - 1, column 0); padding slots and the left / right border read zeros
  auto issue_w = [&](int wk, int stage) {
    const unsigned d = lds0 + OFF_W + stage * STAGE;
    hdma2s(offB[0], offB[1], Wb + (size_t)wk * 128, d + (2 * wave) * 1024, d + (2 * wave + 1) * 1024);
  };
  // source of (halo row hr, halo column hc) of this wave's chunk plane = base + hr * x_rs + (hc - 1) * x_ps bytes: NHWC (pixel = CIN * 2 bytes, the wave's
  // chunk 16 bytes into the half) or row-chunk-planar (conv_gemm.h x_planar: a row of one chunk = W * 16 contiguous bytes)
  const int x_rs = p.x_planar ? (CIN / 8) * p.W * 16 : p.W * CIN * 2, x_ps = p.x_planar ? 16 : CIN * 2;
  // one piece of the halo tile whose pixel (row r0 - 1, column 0, first channel of the half) is at `base` into buffer nb; base == nullptr:
  // nothing to fetch (zero page) - the piece is still issued so that the counted waits below see the same number of operations in
  // flight on every path.  top / bot: the tile touches the upper / lower image border (halo row 0 / 9 is zero padding).
  auto issue_piece = [&](int j, int lane_o, const unsigned char* base, bool top, bool bot, int nb) {
    const int s = j * 64 + lane_o, hr = s / HP, hc = s - hr * HP;
    // branch-free (bitwise & / one select): an exec-masked branch here is a scheduling boundary that keeps the whole address computation
    // in front of the MFMA block instead of between its MFMAs
    const bool ok = (base != nullptr) & (s < HR * HP) & (hc >= 1) & (hc <= TW) & !(top & (hr == 0)) & !(bot & (hr == HR - 1));
    const unsigned long a = (unsigned long)base + (unsigned long)(unsigned)(hr * x_rs + (hc - 1) * x_ps);
    const unsigned long z = (unsigned long)g_zero_page_halo;
    hdma1(reinterpret_cast<const void*>(ok ? a : z), lds0 + nb * HBUF + wave * PLANE + j * 1024);
  };
  auto halo_base = [&](int tile, int h) {
    const int bb = tile / tiles_per_img, rr = (tile - bb * tiles_per_img) * TR;
    return Xb + (long)(bb * p.H + rr - 1) * x_rs + (p.x_planar ? (h * 8 + wave) * p.W * 16 : h * 128 + wave * 16);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int tix = blockIdx.x;
  if (tix >= n_tiles) return;
  const int my_tiles = (n_tiles - tix + (int)gridDim.x - 1) / (int)gridDim.x;
  int steps_left = my_tiles * NKT;                              // K-tile steps this workgroup still has to run (incl. the current one)

  // weight K tile of step q of a tile (steps run phase-major, weights are packed tap-major): q -> (tap, half) -> tap * NH + half
  auto wk_of = [&](int q) { return q >= NKM ? NKM : (NH == 2 ? (q >= 9 ? (q - 9) * 2 + 1 : q * 2) : q); };

  // ---- prologue: whole halo of (tile, half 0) into buffer 0, weight K tiles of steps 0 and 1
  {
    const int r00 = (tix % tiles_per_img) * TR;
    const unsigned char* b0 = halo_base(tix, 0);
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) issue_piece(j, lane, b0, r00 == 0, r00 + TR == p.H, 0);
  }
  issue_w(wk_of(0), 0);
  issue_w(wk_of(1 % NKT), 1);
  HWAIT(0);
  hbar();
  int st_cur = 0, st_pre = 2, q_pre = 2 % NKT;                  // ring state: stage being consumed, stage / step-in-tile being prefetched
  int buf = 0;                                                  // halo buffer of the current phase
#define H_STAMP(acc_)

  auto advance = [&]() {
    st_cur = st_cur == NST - 1 ? 0 : st_cur + 1;
    st_pre = st_pre == NST - 1 ? 0 : st_pre + 1;
    q_pre = q_pre == NKT - 1 ? 0 : q_pre + 1;
    --steps_left;
  };

  while (true) {
    const int tnext = tix + (int)gridDim.x;
    const bool has_next = tnext < n_tiles;
    const int b = tix / tiles_per_img, r0 = (tix - b * tiles_per_img) * TR;
    u32x4 xf0[MT], wf0[NT];                                      // chunk-0 fragments; after the last tap they hold the tail K tile's (FUSE_TAIL)

#pragma unroll 1
    for (int h = 0; h < NH; ++h) {
      // the phase after this one: the other half of this tile, or half 0 of the next tile, or none (dummy pieces)
      const int n_tix = h + 1 < NH ? tix : (has_next ? tnext : -1), n_h = h + 1 < NH ? h + 1 : 0;
      const unsigned char* nbase = n_tix >= 0 ? halo_base(n_tix, n_h) : nullptr;
      const int n_r0 = n_tix >= 0 ? (n_tix % tiles_per_img) * TR : 0;
      const bool n_top = n_r0 == 0, n_bot = n_r0 + TR == p.H;
      // Register pressure: everything derived from the lane index for the pieces / the tail loads / the epilogue is loop-invariant, and
      // hipcc would hoist ~80 VGPRs of addresses out of the tile loop (spills in the K loop); an opaque copy per phase keeps the
      // arithmetic where it is used.
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      u32x4 xf1[MT], wf1[NT];
      // chunk 0 of tap 0 (exposed once per phase: fragments are not carried over the phase loop, see the header)
      {
        const unsigned char* wb = wst0 + st_cur * STAGE + b_rd;
#pragma unroll
        for (int j = 0; j < NT; ++j) wf0[j] = *reinterpret_cast<const u32x4*>(wb + j * 2048 + sw0);
#pragma unroll
        for (int i = 0; i < MT; ++i) xf0[i] = *reinterpret_cast<const u32x4*>(smem + xb[i]);
      }
      H_STAMP(ckP);
      static_for<9>([&](auto tapc) {
        constexpr int TAP = decltype(tapc)::value;
        constexpr int TOFF = ((TAP / 3) * HP + TAP % 3) * 16;                       // this tap's shift inside a plane
        constexpr int TOFF_N = (((TAP + 1) / 3) * HP + (TAP + 1) % 3) * 16;         // the next tap's
        const bool pre_w = steps_left > 2;
        if (pre_w) issue_w(wk_of(q_pre), st_pre);                                    // weights of the step two ahead
        __builtin_amdgcn_sched_barrier(0);
        // ---- region A: the 20 MFMAs of chunk 0, with the 9 fragment reads of chunk 1 (k 32..63 of the half) issued one behind each
        // of the first 9 MFMAs.  All eight waves pass the same barriers, so whatever a wave issues OUTSIDE an MFMA block is time the
        // SIMD's matrix pipe idles (both of its waves are in that section together): every other instruction rides between MFMAs.
        {
          const unsigned char* wb = wst0 + st_cur * STAGE + b_rd;
#pragma unroll
          for (int j = 0; j < NT; ++j) wf1[j] = *reinterpret_cast<const u32x4*>(wb + j * 2048 + sw1);
#pragma unroll
          for (int i = 0; i < MT; ++i) xf1[i] = *reinterpret_cast<const u32x4*>(smem + xb[i] + 4 * PLANE + TOFF);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mma_chunk<bf16>(wf0[j], xf0[i], acc[i][j]);
#pragma unroll
        for (int k = 0; k < MT + NT; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                         // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                         // 1 DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - (MT + NT), 0);
        __builtin_amdgcn_sched_barrier(0);
        H_STAMP(ckA);
        // Everything older than the youngest operations has landed: the weights of the NEXT step (issued one step ago), and at tap 8
        // every halo piece of the next phase.  Younger: the halo piece issued behind the previous step's barrier (taps 1..7) and the
        // two weight DMAs just issued.  lgkmcnt(0): this wave's reads of the current weight stage are complete before anyone may
        // overwrite it.
        if (TAP >= 1 && TAP <= NPIECE) { if (pre_w) HWAIT(3); else HWAIT(1); }
        else { if (pre_w) HWAIT(2); else HWAIT(0); }
        H_STAMP(ckV);
        hbar();
        H_STAMP(ckW);
        // ---- region B: the 20 MFMAs of chunk 1; between them the halo piece of the next phase (address arithmetic + one LDS-DMA) and
        // the 9 fragment reads of the next tap's chunk 0
        if constexpr (TAP < NPIECE) issue_piece(TAP, lane_o, nbase, n_top, n_bot, buf ^ 1);
        const int st_next = st_cur == NST - 1 ? 0 : st_cur + 1;
        if constexpr (TAP < 8) {                                                     // chunk 0 of the next tap
          const unsigned char* wb = wst0 + st_next * STAGE + b_rd;
#pragma unroll
          for (int j = 0; j < NT; ++j) wf0[j] = *reinterpret_cast<const u32x4*>(wb + j * 2048 + sw0);
#pragma unroll
          for (int i = 0; i < MT; ++i) xf0[i] = *reinterpret_cast<const u32x4*>(smem + xb[i] + TOFF_N);
        } else if (FUSE_TAIL) {
          if (h == NH - 1) {                                                         // im2col rows of the tile's pixels for the tail K tile
            const bf16* X2 = reinterpret_cast<const bf16*>(p.x2);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              const int lr = lane_o & 15;
              const int fr_o = ((lr >> 3) & 1) * 2 + ((lr >> 1) & 1), fc_o = ((lr >> 2) & 1) * 2 + (lr & 1);
              const size_t pix = (size_t)(b * p.H + r0 + blk_r(i) * 4 + fr_o) * p.W + blk_c(i) * 4 + fc_o;
              xf0[i] = hgload16(X2 + pix * p.x2_cstride + lq * 8);      // asynchronous: certified by the counted wait of the tail step
            }
          }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mma_chunk<bf16>(wf1[j], xf1[i], acc[i][j]);
        if constexpr (TAP < NPIECE) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                       // 3 VALU of the piece's address
          }
#pragma unroll
          for (int k = 0; k < MT + NT; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - 8 - (MT + NT), 0);
        } else if constexpr (TAP < 8) {
#pragma unroll
          for (int k = 0; k < MT + NT; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - (MT + NT), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        H_STAMP(ckB);
        advance();
      });
      buf ^= 1;
#pragma unroll
      for (int i = 0; i < MT; ++i) xb[i] = buf ? xb[i] + (unsigned)HBUF : xb[i] - (unsigned)HBUF;
    }

    if constexpr (FUSE_TAIL) {   // tail K tile: 32 im2col taps of the downsample conv, one k-chunk
      // The five im2col loads (asynchronous, into xf0) were issued behind the previous barrier; everything older - the weights of this
      // step and of the next - was issued a whole K tile ago, so one unconditional vmcnt(0) costs nothing.  ONE asm statement carries
      // the registers: with a wait per branch hipcc tied each branch's operands to different registers and copied the fragments
      // BEFORE the wait on one path (stale data in the last tile of every workgroup).
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(xf0[0]), "+v"(xf0[1]), "+v"(xf0[2]), "+v"(xf0[3]), "+v"(xf0[4]) :: "memory");
      const bool pre_w = steps_left > 2;
      if (pre_w) issue_w(wk_of(q_pre), st_pre);
      {
        const unsigned char* wb = wst0 + st_cur * STAGE + b_rd;
#pragma unroll
        for (int j = 0; j < NT; ++j) wf0[j] = *reinterpret_cast<const u32x4*>(wb + j * 2048 + sw0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mma_chunk<bf16>(wf0[j], xf0[i], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      hbar();
      advance();
      H_STAMP(ckT);
    }

    // ---- epilogue: per n-tile pair a lane holds 8 consecutive channels (wn * 64 + 32 jp + 8 lq .. + 7) of pixel lrow
    {
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));                          // see the phase loop: keeps the store addresses out of the K loop's registers
      const int lrow_e = lane_e & 15, lq_e = lane_e >> 4;
      const int fr_e = ((lrow_e >> 3) & 1) * 2 + ((lrow_e >> 1) & 1), fc_e = ((lrow_e >> 2) & 1) * 2 + (lrow_e & 1);
      f32x4 bv[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j)
        bv[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + wn * 64 + (j >> 1) * 32 + lq_e * 8 + (j & 1) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (STATS) {               // BatchNorm statistics of the stored map (training forward): per-lane running sums over every tile of this workgroup
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const f32x4 v = acc[i][j] + bv[j];
            st0[j] += v;
            st1[j] += v * v;
          }
      }
      auto finish = [&](auto actf) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          size_t orow;
          const float* posrow = nullptr;
          const int prow_i = blk_r(i) * 4 + fr_e, pcol_i = blk_c(i) * 4 + fc_e;
          if (FUSE_TAIL) {
            const int ph = p.H >> 1, pw = p.W >> 1;
            const int py = (r0 + prow_i) >> 1, px = pcol_i >> 1;
            orow = (size_t)(b * ph + py) * pw + px;
            posrow = p.pos + (size_t)(py * pw + px) * p.y_cstride;
          } else {
            orow = (size_t)(b * p.H + r0 + prow_i) * p.W + pcol_i;
          }
#pragma unroll
          for (int jp = 0; jp < NT / 2; ++jp) {
            const int n = wn * 64 + jp * 32 + lq_e * 8;
            f32x4 v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              v[u] = acc[i][2 * jp + u] + bv[2 * jp + u];
              acc[i][2 * jp + u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int e = 0; e < 4; ++e) v[u][e] = actf(v[u][e]);
              if (FUSE_TAIL) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {        // max over the window's 4 lanes with DPP quad permutes (VALU): __shfl_xor is an LDS ds_bpermute per value
                  v[u][e] = fmaxf(v[u][e], quad_xor<0xB1>(v[u][e]));      // lanes ^ 1: quad_perm [1,0,3,2]
                  v[u][e] = fmaxf(v[u][e], quad_xor<0x4E>(v[u][e]));      // lanes ^ 2: quad_perm [2,3,0,1]
                }
                v[u] += *reinterpret_cast<const f32x4*>(posrow + n + 4 * u);
              }
            }
            if (!FUSE_TAIL || (lrow_e & 3) == 0) {
              const bf16x8 o = {(bf16)v[0][0], (bf16)v[0][1], (bf16)v[0][2], (bf16)v[0][3], (bf16)v[1][0], (bf16)v[1][1], (bf16)v[1][2], (bf16)v[1][3]};
              if (!FUSE_TAIL && p.y_planar) *reinterpret_cast<bf16x8*>(Y + (((size_t)(b * p.H + r0 + prow_i) * (p.y_cstride >> 3) + (n >> 3)) * p.W + pcol_i) * 8) = o;
              else *reinterpret_cast<bf16x8*>(Y + orow * p.y_cstride + n) = o;
            }
          }
        }
      };
      // Pooled variant with a monotone activation (LeakyReLU / none; the bias is per channel): max(act(a_i + b)) = act(max(a_i) + b), so
      // the window maximum is taken on the raw accumulators (2 DPP + 2 max per value), after which the window's four lanes hold the SAME
      // four f32x4 - lane u = lrow & 3 finishes only the u-th of them (bias, activation, pos_embed, 8-byte store): the per-value tail
      // work and the bias / pos loads drop 4 x (the epilogue was 27 % of this kernel: 16 values x 8.5 VALU per m-tile and lane).
      auto finish_pooled = [&](auto actf) {
        const int u_e = lrow_e & 3;
        const int nq = wn * 64 + (u_e >> 1) * 32 + lq_e * 8 + (u_e & 1) * 4;             // the 4 channels this lane finishes
        const f32x4 bq = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nq) : f32x4{0.f, 0.f, 0.f, 0.f};
        const int ph = p.H >> 1, pw = p.W >> 1;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int py = (r0 + blk_r(i) * 4 + fr_e) >> 1, px = (blk_c(i) * 4 + fc_e) >> 1;
          const f32x4 pq = *reinterpret_cast<const f32x4*>(p.pos + (size_t)(py * pw + px) * p.y_cstride + nq);
          f32x4 m[NT];
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            m[j] = acc[i][j];
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              m[j][e] = fmaxf(m[j][e], quad_xor<0xB1>(m[j][e]));      // lanes ^ 1: quad_perm [1,0,3,2]
              m[j][e] = fmaxf(m[j][e], quad_xor<0x4E>(m[j][e]));      // lanes ^ 2: quad_perm [2,3,0,1]
            }
          }
          f32x4 v = u_e == 0 ? m[0] : (u_e == 1 ? m[1] : (u_e == 2 ? m[2] : m[3]));
          v += bq;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = actf(v[e]);
          v += pq;
          store4<bf16>(Y + ((size_t)(b * ph + py) * pw + px) * p.y_cstride + nq, v);
        }
      };
      if (FUSE_TAIL && p.act == ACT_LRELU) finish_pooled([](float x) { return fmaxf(x, 0.1f * x); });
      else if (FUSE_TAIL && p.act == ACT_NONE) finish_pooled([](float x) { return x; });
      else if (p.act == ACT_GELU) finish([](float x) { return gelu_sig(x); });
      else if (p.act == ACT_LRELU) finish([](float x) { return fmaxf(x, 0.1f * x); });
      else finish([](float x) { return x; });
    }
    H_STAMP(ckE);
    if (!has_next) break;
    tix = tnext;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // no DMA (dummy pieces) may be in flight into the LDS of a finished workgroup
  if constexpr (STATS) {
    // workgroup partial: the 16 pixel lanes of a channel quad (shuffles), then the 4 waves that share the 64-channel half, in fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);             // [8 waves][2][64 channels]
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = st0[j][e], q = st1[j][e];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); q += __shfl_xor(q, o); }
        if (lrow == 0) {
          const int cl = (j >> 1) * 32 + lq * 8 + (j & 1) * 4 + e;     // channel inside the wave's 64 (the epilogue's 16-byte order)
          red[(wave * 2 + 0) * 64 + cl] = a;
          red[(wave * 2 + 1) * 64 + cl] = q;
        }
      }
    __syncthreads();
    if (t < 256) {
      const int which = t >> 7, c = t & 127, h = c >> 6, cl = c & 63;
      float a = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) a += red[((m * 2 + h) * 2 + which) * 64 + cl];      // waves (wm = m, wn = h)
      p.stats[((size_t)blockIdx.x * 2 + which) * p.y_cstride + c] = a;
    }
  }
}

bool conv3x3_halo_eligible(const ConvGemmParams& p, int dtype) {
  constexpr bool off = false;
  if (off || dtype != 1) return false;
  if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.groups != 1) return false;
  if (p.N != 128 || p.y_cstride != 128 || p.W != halo::TW || (p.H % halo::TR) || p.OH != p.H || p.OW != p.W) return false;
  if (p.res || p.y2 || p.act == ACT_MUL || p.y_rpi || p.out_f32 || p.w_rstride || p.w_gstride) return false;
  if (p.Cin != p.x_cstride || p.K != 9 * p.Cin) return false;
  if (p.stats && (p.pool2 || p.act != ACT_NONE)) return false;
  if (p.y_planar && (p.pool2 || p.stats)) return false;      // the planar output exists for the plain epilogue only (conv2 -> conv3 inside the eval stem)
  if (p.pool2) return p.Cin == 128 && p.x2 && p.K2 == 32 && p.x2_cstride >= 32 && p.pos && p.Kw == p.K + 64;
  return (p.Cin == 64 || p.Cin == 128) && !p.x2 && !p.pos && p.Kw == p.K;
}

static inline int halo_grid(const ConvGemmParams& p) { const int n_tiles = p.B * (p.H / halo::TR); return n_tiles < 256 ? n_tiles : 256; }
// (the statistics rows = the persistent grid: one partial per workgroup)
int conv3x3_halo_stats_rows(const ConvGemmParams& p, int dtype) {
  ConvGemmParams q = p;
  float dummy;
  q.stats = &dummy;
  return conv3x3_halo_eligible(q, dtype) ? halo_grid(p) : 0;
}

template <int CIN, bool FUSE_TAIL, bool STATS = false>
static int launch_halo_t(const ConvGemmParams& p, hipStream_t stream) {
  const int n_tiles = p.B * (p.H / halo::TR);
  auto kern = conv3x3_halo_kernel<CIN, FUSE_TAIL, STATS>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, halo::LDS_BYTES);
  if (e != hipSuccess) return (int)e;
  const int grid = n_tiles < 256 ? n_tiles : 256;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), halo::LDS_BYTES, stream, p, n_tiles);
  return (int)hipGetLastError();
}

int launch_conv3x3_halo(const ConvGemmParams& p, hipStream_t stream) {
  if (p.pool2) return launch_halo_t<128, true>(p, stream);
  if (p.stats) return p.Cin == 64 ? launch_halo_t<64, false, true>(p, stream) : launch_halo_t<128, false, true>(p, stream);
  return p.Cin == 64 ? launch_halo_t<64, false>(p, stream) : launch_halo_t<128, false>(p, stream);
}

}  // namespace FSVIT_NS
