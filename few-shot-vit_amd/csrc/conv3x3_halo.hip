// 3x3 / stride 1 / pad 1 convolution with the input tile + halo RESIDENT IN LDS (stem conv2, conv3 of the Visformer,
// test_phase/models/visformer.py:211-213,224-237), bf16.
//
// Why: conv_gemm_v2 runs a 3x3 conv as an implicit GEMM that re-stages every input pixel once per tap - 9 x the
// activation bytes through the LDS-DMA path, which delivers only ~16 B/clk/CU (tools/bench_gemm256.py) - so it sits on
// the fill roof of a 128x128 tile (64 flop per staged byte, ~620 TFLOP/s; measured 627).  Here a workgroup owns 8 output
// rows x 40 columns x 128 channels: the 10 x 42 pixel halo tile is staged ONCE (107 KB at 128 input channels) and the nine
// taps are shifted LDS reads of it; only the weights stream (16 KB per 64-wide K tile, 3-stage ring, counted vmcnt,
// shared by every tile of the persistent workgroup).  235 flop per staged byte: the kernel is MFMA / LDS-read bound.
//   * 8 waves = 4 (pixels) x 2 (channels): a wave holds 5 x 4 accumulator tiles (80 pixels x 64 channels), per K tile
//     10 + 8 ds_read_b128 and 40 MFMAs;
//   * halo pixels are 128 / 256 bytes; the 16-byte chunk position inside a pixel is XOR-swizzled with the low bits of the
//     halo pixel index (applied to the per-lane DMA source, as in conv_gemm_v2), so the 16 pixels of a fragment read hit
//     16 different bank groups whatever the tap shift;
//   * FUSE_TAIL (conv3): the downsample/identity conv rides as one extra K tile whose A fragments come straight from the
//     im2col rows in global memory, rows are enumerated 2x2-window-major so LeakyReLU -> MaxPool2d(2) is a max over 4
//     adjacent lanes, and pos_embed1 is added in the same epilogue (same contract as conv_gemm_v2's x2 / pool2 / pos).
#include <stdlib.h>

#include <type_traits>

#include "conv_gemm.h"
#include "fsvit_common.h"

namespace fsvit {

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
#define HWAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
__device__ __forceinline__ void hbar() {
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

constexpr int H_TR = 8, H_TW = 40, H_HW = H_TW + 2, H_HR = H_TR + 2, H_HP = H_HR * H_HW;   // 8 x 40 outputs, 10 x 42 halo
constexpr int H_NST = 3, H_STAGE = 16384;
__host__ __device__ constexpr int halo_bytes(int cin) { return (H_HP * cin * 2 + 1023) / 1024 * 1024; }

template <int CIN, bool FUSE_TAIL>
__global__ __launch_bounds__(512, 1) void conv3x3_halo_kernel(const ConvGemmParams p, const int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PXB = CIN * 2, CPP = CIN / 8;                 // bytes / 16-byte chunks per pixel
  constexpr int HALO = halo_bytes(CIN), ND = HALO / 1024;     // halo DMAs per tile
  constexpr int KH64 = CIN / 64, NKM = 9 * KH64, NKT = NKM + (FUSE_TAIL ? 1 : 0);
  constexpr int MT = 5, NT = 4;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = lane & 15, lq = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)(lptrh_t)smem;
  unsigned char* const wst0 = smem + HALO;

  const bf16* __restrict__ X = reinterpret_cast<const bf16*>(p.x);
  const unsigned char* const Wb = reinterpret_cast<const unsigned char*>(p.w);
  bf16* __restrict__ Y = reinterpret_cast<bf16*>(p.y);
  const int tiles_per_img = p.H / H_TR;

  // this lane's 5 output pixels (tile-local) -> halo pixel index of tap (0, 0)
  int hp0[MT], prow[MT], pcol[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int q = (wm * MT + i) * 16 + lrow;
    int r, c;
    if (FUSE_TAIL) {                 // 2x2-window-major: q = window * 4 + dy * 2 + dx
      const int win = q >> 2, lpy = win / (H_TW / 2), lpx = win - lpy * (H_TW / 2);
      r = 2 * lpy + ((q >> 1) & 1);
      c = 2 * lpx + (q & 1);
    } else {
      r = q / H_TW;
      c = q - r * H_TW;
    }
    prow[i] = r; pcol[i] = c;
    hp0[i] = r * H_HW + c;
  }
  // weight DMA rows of this wave (128 rows of 128 bytes per K tile = 16 groups of 8)
  const int srow = lane >> 3;
  const unsigned schunk = (unsigned)(((lane & 7) ^ srow) << 4);
  unsigned offB[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) offB[j] = (unsigned)((2 * wave + j) * 8 + srow) * (unsigned)(p.Kw * 2) + schunk;
  const unsigned b_rd = (unsigned)((wn * 64 + lrow) * 128);
  const unsigned sw0 = (unsigned)((lq ^ (lrow & 7)) << 4), sw1 = (unsigned)(((4 + lq) ^ (lrow & 7)) << 4);

  auto issue_w = [&](int wk, int stage) {
    const unsigned d = lds0 + HALO + stage * H_STAGE;
    hdma2s(offB[0], offB[1], Wb + (size_t)wk * 128, d + (2 * wave) * 1024, d + (2 * wave + 1) * 1024);
  };
  auto issue_halo = [&](int tix) {
    const int b = tix / tiles_per_img, r0 = (tix - b * tiles_per_img) * H_TR;
    for (int d = wave; d < ND; d += 8) {
      const int slot = d * 64 + lane;
      const int hp = slot / CPP, pos = slot - hp * CPP;
      const int c = pos ^ (hp & (CPP - 1));
      const int hr = hp / H_HW, hc = hp - hr * H_HW;
      const int iy = r0 + hr - 1, ix = hc - 1;
      const bool ok = hp < H_HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const void* src = ok ? static_cast<const void*>(X + ((size_t)(b * p.H + iy) * p.W + ix) * CIN + c * 8) : static_cast<const void*>(g_zero_page_halo);
      hdma1(src, lds0 + d * 1024);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int tix = blockIdx.x;
  if (tix >= n_tiles) return;
  issue_halo(tix);
  issue_w(0, 0);
  issue_w(1 % NKT, 1);
  HWAIT_VM(0);
  hbar();
  // wave-uniform ring state, advanced incrementally (no divisions in the loop): stage of the K tile being consumed, stage and
  // weight K-tile index of the one being prefetched (two ahead)
  int st_cur = 0, st_pre = 2, wk_pre = 2 % NKT;
  while (true) {
    const int tnext = tix + gridDim.x;
    const bool has_next = tnext < n_tiles;
    const int b = tix / tiles_per_img, r0 = (tix - b * tiles_per_img) * H_TR;
    // K loop, rotated by half a K tile so that LDS reads always run under MFMAs: the two 32-wide k-chunks of a K tile live
    // in two fragment sets; while chunk 0 of tile g multiplies, chunk 1 of g is read, and while chunk 1 multiplies, chunk 0
    // of g+1 is read (after the counted wait + barrier that publishes the weight stage of g+1).  With all 8 waves in
    // lockstep the unrotated loop exposed the whole read burst (144 ds_read_b128 per K tile per CU) before every MFMA block.
    u32x4 xf0[MT], xf1[MT], wf0[NT], wf1[NT];
    // Fragment addresses of the current K tile: computed once by read_k0 (6 VALU each), chunk 1 of the same tile is the same
    // address with bit 6 flipped.  Recomputing them in read_k1 put 2.6 VALU instructions beside every 16-cycle MFMA (PMC: VALU
    // 26 % busy) - more than issue for free behind it (tools/probes/mfma_valu_overlap.hip).
    unsigned xa[MT];
    auto read_k0 = [&](int toff, int kh, int stage) {
      const unsigned char* wb = wst0 + stage * H_STAGE + b_rd;
#pragma unroll
      for (int j = 0; j < NT; ++j) wf0[j] = *reinterpret_cast<const u32x4*>(wb + j * 2048 + sw0);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int hp = hp0[i] + toff;
        xa[i] = (unsigned)hp * PXB + (unsigned)((((kh * 8 + lq) ^ hp) & (CPP - 1)) << 4);
        xf0[i] = *reinterpret_cast<const u32x4*>(smem + xa[i]);
      }
    };
    auto read_k1 = [&](int stage) {
      const unsigned char* wb = wst0 + stage * H_STAGE + b_rd;
#pragma unroll
      for (int j = 0; j < NT; ++j) wf1[j] = *reinterpret_cast<const u32x4*>(wb + j * 2048 + sw1);
#pragma unroll
      for (int i = 0; i < MT; ++i) xf1[i] = *reinterpret_cast<const u32x4*>(smem + (xa[i] ^ 64u));   // chunk + 4
    };
    auto read_tail0 = [&](int stage) {   // identity / downsample conv: A rows = the 27 (padded to 32) im2col taps, straight from global
      const unsigned char* wb = wst0 + stage * H_STAGE + b_rd;
#pragma unroll
      for (int j = 0; j < NT; ++j) wf0[j] = *reinterpret_cast<const u32x4*>(wb + j * 2048 + sw0);
      const bf16* X2 = reinterpret_cast<const bf16*>(p.x2);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const size_t pix = (size_t)(b * p.H + r0 + prow[i]) * p.W + pcol[i];
        xf0[i] = *reinterpret_cast<const u32x4*>(X2 + pix * p.x2_cstride + lq * 8);
      }
    };
    auto mma0 = [&]() {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mma_chunk<bf16>(wf0[j], xf0[i], acc[i][j]);
    };
    auto mma1 = [&]() {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mma_chunk<bf16>(wf1[j], xf1[i], acc[i][j]);
    };
    auto advance = [&]() {
      st_cur = st_cur == H_NST - 1 ? 0 : st_cur + 1;
      st_pre = st_pre == H_NST - 1 ? 0 : st_pre + 1;
      wk_pre = wk_pre == NKT - 1 ? 0 : wk_pre + 1;
    };
    int ky = 0, kx = 0, kh = 0;                              // tap / 64-channel slice of the current K tile
    read_k0(0, 0, st_cur);
    // One K tile.  NEXT: 0 = another main K tile follows, 1 = the tail K tile follows, 2 = nothing follows.  The steady-state
    // instance (MORE known, NEXT = 0) is straight-line code: with a branch around the reads hipcc's waitcnt pass falls back
    // to lgkmcnt(4..0) in front of the second MFMA block, i.e. waits for the reads it was meant to overlap.
    auto ktile = [&](auto more_c, auto next_c, bool more_rt) {
      constexpr int MORE = decltype(more_c)::value, NEXT = decltype(next_c)::value;     // MORE: 1 yes, 0 no, 2 runtime
      const bool more = MORE == 2 ? more_rt : MORE == 1;
      if (more) issue_w(wk_pre, st_pre);                     // the K tile two ahead (of this output tile, or of the next one)
      read_k1(st_cur);
      __builtin_amdgcn_sched_barrier(0);                     // reads are ISSUED before the MFMA block they run under
      mma0();
      if (more) HWAIT_VM(2); else HWAIT_VM(0);               // everything but the K tile just issued has landed
      hbar();
      int kh2 = kh + 1, kx2 = kx, ky2 = ky;
      if (kh2 == KH64) { kh2 = 0; ++kx2; if (kx2 == 3) { kx2 = 0; ++ky2; } }
      const int st_next = st_cur == H_NST - 1 ? 0 : st_cur + 1;
      if constexpr (NEXT == 0) read_k0(ky2 * H_HW + kx2, kh2, st_next);
      else if constexpr (NEXT == 1) read_tail0(st_next);
      __builtin_amdgcn_sched_barrier(0);
      mma1();
      kh = kh2; kx = kx2; ky = ky2;
      advance();
    };
#pragma unroll 1
    for (int kt = 0; kt + 2 < NKM; ++kt) ktile(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, true);
    if constexpr (FUSE_TAIL) {
      ktile(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, true);           // kt = NKM-2: the tail is 2 ahead
      ktile(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, has_next);       // kt = NKM-1
      if (has_next) issue_w(wk_pre, st_pre);                                                     // tail K tile: 32 taps, one k-chunk
      mma0();
      if (has_next) HWAIT_VM(2); else HWAIT_VM(0);
      advance();
    } else {
      ktile(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, has_next);       // kt = NKM-2
      ktile(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, has_next);       // kt = NKM-1
    }
    hbar();                                               // every wave is done reading this halo tile
    if (has_next) issue_halo(tnext);                      // lands under the epilogue stores

    // ---- epilogue: lane holds 4 consecutive channels of pixel lrow per 16x16 tile
    {
      f32x4 bv[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + wn * 64 + j * 16 + lq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      auto finish = [&](auto actf) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          size_t orow;
          const float* posrow = nullptr;
          if (FUSE_TAIL) {
            const int ph = p.H >> 1, pw = p.W >> 1;
            const int py = (r0 + prow[i]) >> 1, px = pcol[i] >> 1;
            orow = (size_t)(b * ph + py) * pw + px;
            posrow = p.pos + (size_t)(py * pw + px) * p.y_cstride;
          } else {
            orow = (size_t)(b * p.H + r0 + prow[i]) * p.W + pcol[i];
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int n = wn * 64 + j * 16 + lq * 4;
            f32x4 v = acc[i][j] + bv[j];
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = actf(v[e]);
            if (FUSE_TAIL) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[e] = fmaxf(v[e], __shfl_xor(v[e], 1, 64));
                v[e] = fmaxf(v[e], __shfl_xor(v[e], 2, 64));
              }
              v += *reinterpret_cast<const f32x4*>(posrow + n);
              if ((lrow & 3) == 0) store4<bf16>(Y + orow * p.y_cstride + n, v);
            } else {
              store4<bf16>(Y + orow * p.y_cstride + n, v);
            }
          }
        }
      };
      if (p.act == ACT_GELU) finish([](float x) { return gelu_sig(x); });
      else if (p.act == ACT_LRELU) finish([](float x) { return x > 0.0f ? x : 0.1f * x; });
      else finish([](float x) { return x; });
    }
    if (!has_next) break;
    HWAIT_VM(0);                                          // halo + the two prefetched weight tiles of the next output tile
    hbar();
    tix = tnext;
  }
}

bool conv3x3_halo_eligible(const ConvGemmParams& p, int dtype) {
  static const bool off = [] { const char* e = getenv("FSVIT_HALO"); return e && e[0] == '0'; }();
  if (off || dtype != 1) return false;
  if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.groups != 1) return false;
  if (p.N != 128 || p.y_cstride != 128 || p.W != H_TW || (p.H % H_TR) || p.OH != p.H || p.OW != p.W) return false;
  if (p.res || p.y_rpi || p.out_f32 || p.w_rstride || p.w_gstride) return false;
  if (p.Cin != p.x_cstride || p.K != 9 * p.Cin) return false;
  if (p.pool2) return p.Cin == 128 && p.x2 && p.K2 == 32 && p.x2_cstride >= 32 && p.pos && p.Kw == p.K + 64;
  return (p.Cin == 64 || p.Cin == 128) && !p.x2 && !p.pos && p.Kw == p.K;
}

template <int CIN, bool FUSE_TAIL>
static int launch_halo_t(const ConvGemmParams& p, hipStream_t stream) {
  const int n_tiles = p.B * (p.H / H_TR);
  const int lds = halo_bytes(CIN) + H_NST * H_STAGE;
  auto kern = conv3x3_halo_kernel<CIN, FUSE_TAIL>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return (int)e;
  const int grid = n_tiles < 256 ? n_tiles : 256;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p, n_tiles);
  return (int)hipGetLastError();
}

int launch_conv3x3_halo(const ConvGemmParams& p, hipStream_t stream) {
  if (p.pool2) return launch_halo_t<128, true>(p, stream);
  return p.Cin == 64 ? launch_halo_t<64, false>(p, stream) : launch_halo_t<128, false>(p, stream);
}

}  // namespace fsvit
