// Dense bf16 GEMM for the attention + MLP blocks (1x1 convs of Visformer stages 2/3, every Linear of DeiT):
//   y[m][n] = epi( sum_k x[m][k] * w[n][k] ),  x [M][x_cstride] and w [N][Kw] both K-contiguous, fused epilogue as conv_gemm.h.
//
// conv_gemm_v2 (128x128 tile, 4 waves of 64x64, two barriers per K slice) spends ~4 non-MFMA instructions per MFMA and
// measured 31 % MFMA-busy; this kernel follows the structure cdna_hip_programming.md describes for its 256^2 template:
//   * 256x256 output tile, 8 waves (2 x 4), 128x64 per wave = 128 accumulator VGPRs, BK = 64: per K tile a wave runs
//     4 phases of 16 MFMAs (one 64x32 quadrant x K 64) and re-reads only the register sub-tile that changes
//     (A top 8 x b128, B left 4, B right 4, A bottom 8): 28 ds_read_b128 + 8 LDS-DMA per 64 MFMAs;
//   * LDS-DMA addressing is scalar: per-lane 32-bit row offsets are computed once per output tile, the K position
//     lives in an SGPR base that is bumped by 128 bytes per K tile - no VALU address work in the loop;
//   * two LDS stages of 64 KB; the four half-tiles of K tile t+1 are issued one per phase while tile t is computed
//     and are never drained: each phase waits `vmcnt(4)` (everything but the two newest half-tiles), so a half-tile has
//     three phases to land;
//   * one barrier per phase, between the counted wait and the MFMAs; the pipeline runs over the flattened
//     (output tile, K tile) sequence of the persistent workgroup, so the first K tile of the next output tile streams in
//     under the last K tile and the epilogue of the current one.
// Measured (tools/bench_gemm256.py): the LDS-DMA path delivers ~16 B/clk/CU, so at 128 flops per staged byte the kernel is
// fill-bound at ~50 % of the MFMA peak (compute-only loop: 1.5 PFLOP/s at 4096^3, with the DMA 1.1); a half-phase stagger
// of the two wave groups and a second barrier per phase were measured and dropped (-6 %).
// Hazards (LDS-DMA is ordered for a reader only by the issuing wave's counted vmcnt followed by a barrier the reader
// passes): every wave waits, before the barrier of phase g, for all half-tiles issued up to phase g-2 - which include
// everything phase g+1 reads; a slot is rewritten by a DMA issued 2+ phases (hence 1+ barriers that every wave reaches
// only after its MFMAs consumed the reads) after its last read.
#include <stdlib.h>

#include "conv_gemm.h"
#include <type_traits>

#include "fsvit_common.h"

namespace FSVIT_NS {

typedef __attribute__((address_space(3))) void* lptr256_t;

// two 1 KiB LDS-DMAs: per-lane 32-bit byte offsets on one scalar base
__device__ __forceinline__ void dma2(unsigned off0, unsigned off1, const void* sbase, unsigned lds0, unsigned lds1) {
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
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
__device__ __forceinline__ void bar() {
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

constexpr int G_BM = 256, G_BN = 256, G_STAGE = 65536, G_BOFF = 32768;

// LIMBS = the two-limb mode (`bf16x2` / `f16x2`, conv_gemm_v2.hip): fp32 activations / outputs, weights as (hi, lo) limb words.  The LDS
// geometry is byte-identical (a 128-byte row = 32 fp32 k instead of 64 16-bit ones), so the whole DMA / phase / vmcnt structure is shared; the
// activation fragments are split into limbs right after their LDS read (x2_split), the weight fragments half-swapped, and every MFMA of the
// 16-bit kernel becomes the pair (w_swapped, x) + (w, x) = all four limb products.  64 flop per staged byte (128 in the 16-bit kernel).
template <bool LIMBS>
__device__ __forceinline__ void gemm256_body(const ConvGemmParams& p, const int tiles_m, const int tiles_n) {
  constexpr int ES = LIMBS ? 4 : 2;                 // bytes per operand element
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // 16-bit kernel: 2 x 4 waves of 128 (m) x 64 (n).  Two-limb kernel: 4 x 2 waves of 64 x 128 - an activation fragment is split into limbs by
  // every wave that reads it, and this layout halves both the fragments a wave reads per K tile (8 instead of 16) and the waves sharing them.
  constexpr int WM_ROWS = LIMBS ? 64 : 128, WN_COLS = LIMBS ? 128 : 64;
  constexpr int TMH = WM_ROWS / 32, TNH = WN_COLS / 32;        // 16-row tiles per A half / 16-column tiles per B half of a wave
  const int wm = LIMBS ? wave >> 1 : wave >> 2, wn = LIMBS ? wave & 1 : wave & 3;
  const int lrow = lane & 15, lq = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)(lptr256_t)smem;

  // LDS read bases of this lane (row part + the two swizzled 16-byte k-chunk slots of the 128-byte row)
  const unsigned sw0 = (unsigned)((lq ^ (lrow & 7)) << 4), sw1 = (unsigned)(((4 + lq) ^ (lrow & 7)) << 4);
  const unsigned a_rd = (unsigned)((wm * WM_ROWS + lrow) * 128), b_rd = (unsigned)(G_BOFF + (wn * WN_COLS + lrow) * 128);

  // DMA geometry of this wave: half-tile h of A / B = 128 rows = 16 groups of 8 rows, this wave stages groups 2w, 2w+1
  const int srow = lane >> 3;                       // row inside the 8-row group
  const unsigned schunk = (unsigned)(((lane & 7) ^ srow) << 4);   // pre-swizzled source chunk (bytes)
  int rbA[2][2], rbB[2][2];                         // first row of the 8-row group in the 256-row tile
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int hr0 = (2 * wave + j) * 8;
      const int two = h == 0 ? (hr0 < 64 ? hr0 : hr0 + 64) : (hr0 < 64 ? hr0 + 64 : hr0 + 128);   // 2 wave rows / columns of 128: {0-63,128-191} | {64-127,192-255}
      const int four = (hr0 >> 5) * 64 + h * 32 + (hr0 & 31);                                      // 4 wave rows / columns of 64: each one's first | second 32
      rbA[h][j] = LIMBS ? four : two;
      rbB[h][j] = LIMBS ? two : four;
    }

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, P = gridDim.x >> 3;
  const int cnt = tiles_m > xcd ? ((tiles_m - xcd + 7) / 8) * tiles_n : 0;
  if (slot >= cnt) return;
  const int nk = p.Kw / (128 / ES);
  const unsigned char* const Xb = reinterpret_cast<const unsigned char*>(p.x);
  const unsigned char* const Wb = reinterpret_cast<const unsigned char*>(p.w);
  bf16* __restrict__ Y = reinterpret_cast<bf16*>(p.y);
  const bf16* __restrict__ R = reinterpret_cast<const bf16*>(p.res);

  // per-lane byte offsets of the 8 DMA rows of an output tile (clamped: tail rows re-read the last valid row, their
  // results are never stored)
  unsigned offA[2][2], offB[2][2], offAn[2][2], offBn[2][2];
  auto setup = [&](int it, unsigned (&oa)[2][2], unsigned (&ob)[2][2]) {
    const int mt = xcd + 8 * (it / tiles_n), nt = it % tiles_n;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int m = mt * G_BM + rbA[h][j] + srow;
        m = m < p.M ? m : p.M - 1;
        // LDS weight row R holds output channel (R & ~31) + perm(R & 31): row r of the even / odd 16-row MFMA tile of a
        // 32-channel block is channel 8 (r / 4) + 4 (tile & 1) + (r & 3), so that the 4 + 4 accumulators a lane holds of
        // the tile pair are 8 CONSECUTIVE channels of one output row -> one 16-byte store (and bias / residual load)
        const int R = rbB[h][j] + srow, r32 = R & 31;
        int n = nt * G_BN + (R & ~31) + 8 * ((r32 & 15) >> 2) + 4 * (r32 >> 4) + (r32 & 3);
        n = n < p.N ? n : p.N - 1;
        oa[h][j] = (unsigned)m * (unsigned)(p.x_cstride * ES) + schunk;
        ob[h][j] = (unsigned)n * (unsigned)(p.Kw * ES) + schunk;
      }
  };
  // mode 1: K tile kt of the current output tile; mode 2: K tile 0 of the NEXT output tile
  auto issueA = [&](int h, int mode, int kt, int stage) {
    const unsigned d = lds0 + stage * G_STAGE;
    const bool nx = mode == 2;
    dma2(nx ? offAn[h][0] : offA[h][0], nx ? offAn[h][1] : offA[h][1], Xb + (size_t)(nx ? 0 : kt) * 128, d + rbA[h][0] * 128, d + rbA[h][1] * 128);
  };
  auto issueB = [&](int h, int mode, int kt, int stage) {
    const unsigned d = lds0 + stage * G_STAGE + G_BOFF;
    const bool nx = mode == 2;
    dma2(nx ? offBn[h][0] : offB[h][0], nx ? offBn[h][1] : offB[h][1], Wb + (size_t)(nx ? 0 : kt) * 128, d + rbB[h][0] * 128, d + rbB[h][1] * 128);
  };

  f32x4 acc[2 * TMH][2 * TNH];
#pragma unroll
  for (int i = 0; i < 2 * TMH; ++i)
#pragma unroll
    for (int j = 0; j < 2 * TNH; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 xa[TMH][2], wb[TNH][2];
  bool xa_raw = false;                             // LIMBS: xa holds fp32 values not yet split into limb words
  auto readA = [&](int half, int stage) {          // rows wm*128 + half*64 + i*16 + lrow
    const unsigned char* base = smem + stage * G_STAGE + a_rd + half * (TMH * 16 * 128);
#pragma unroll
    for (int i = 0; i < TMH; ++i) {
      xa[i][0] = *reinterpret_cast<const u32x4*>(base + i * 2048 + sw0);
      xa[i][1] = *reinterpret_cast<const u32x4*>(base + i * 2048 + sw1);
    }
    xa_raw = true;
  };
  auto readB = [&](int half, int stage) {          // weight rows wn*64 + half*32 + j*16 + lrow
    const unsigned char* base = smem + stage * G_STAGE + b_rd + half * (TNH * 16 * 128);
#pragma unroll
    for (int j = 0; j < TNH; ++j) {
      wb[j][0] = *reinterpret_cast<const u32x4*>(base + j * 2048 + sw0);
      wb[j][1] = *reinterpret_cast<const u32x4*>(base + j * 2048 + sw1);
    }
  };
  auto mma = [&](int ah, int bh) {                 // quadrant (ah, bh): 4 x 2 tiles x 2 k-chunks
    if constexpr (LIMBS) {
      if (xa_raw) {                                // (after the phase's wait + barrier: the split sits next to the MFMAs that consume it)
#pragma unroll
        for (int i = 0; i < TMH; ++i)
#pragma unroll
          for (int kc = 0; kc < 2; ++kc) {
            u32x4 xs, xr;
            x2_split(xa[i][kc], xs, xr);
            xa[i][kc] = xs;
          }
        xa_raw = false;
      }
      // the cross terms pair the weight words with the half-swapped activation words; the swap is redone per phase on the (fewer) activation
      // fragments instead of kept in registers (254 VGPRs as it is)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int i = 0; i < TMH; ++i) {
          u32x4 xr;
#pragma unroll
          for (int e = 0; e < 4; ++e) xr[e] = __builtin_amdgcn_alignbit(xa[i][kc][e], xa[i][kc][e], 16);
#pragma unroll
          for (int j = 0; j < TNH; ++j) {
            acc[ah * TMH + i][bh * TNH + j] = mma_chunk<bf16>(wb[j][kc], xr, acc[ah * TMH + i][bh * TNH + j]);          // lo x hi + hi x lo
            acc[ah * TMH + i][bh * TNH + j] = mma_chunk<bf16>(wb[j][kc], xa[i][kc], acc[ah * TMH + i][bh * TNH + j]);   // hi x hi + lo x lo
          }
        }
    } else {
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[ah * 4 + i][bh * 2 + j] = mma_chunk<bf16>(wb[j][kc], xa[i][kc], acc[ah * 4 + i][bh * 2 + j]);
    }
  };

  // ---- ONE software pipeline over the flattened (output tile, K tile) sequence of this workgroup
  int it = slot;
  setup(it, offA, offB);
  issueA(0, 1, 0, 0); issueB(0, 1, 0, 0); issueB(1, 1, 0, 0); issueA(1, 1, 0, 0);
  WAIT_VM(0);
  bar();
  int g = 0;                                       // K tiles done so far: stage = g & 1
  while (true) {
    const int itn = it + P;
    const bool has_next = itn < cnt;
    if (has_next) setup(itn, offAn, offBn);
    for (int kt = 0; kt < nk; ++kt, ++g) {
      const int st = g & 1, sn = st ^ 1;
      const int mode = kt + 1 < nk ? 1 : (has_next ? 2 : 0);       // what streams in under this K tile
      // phase 1: quadrant (top, left)
      if (mode) issueA(0, mode, kt + 1, sn);
      readA(0, st); readB(0, st);
      if (mode) WAIT_VM(4); else WAIT_VM(2);
      bar();
      mma(0, 0);
      // phase 2: (top, right)
      if (mode) issueB(0, mode, kt + 1, sn);
      readB(1, st);
      if (mode) WAIT_VM(4); else WAIT_VM(0);
      bar();
      mma(0, 1);
      // phase 3: (bottom, right)
      if (mode) issueB(1, mode, kt + 1, sn);
      readA(1, st);
      if (mode) WAIT_VM(4); else WAIT_VM(0);
      bar();
      mma(1, 1);
      // phase 4: (bottom, left)
      if (mode) issueA(1, mode, kt + 1, sn);
      readB(0, st);
      if (mode) WAIT_VM(4); else WAIT_VM(0);
      bar();
      mma(1, 0);
    }

    // ---- epilogue of output tile `it` (the next tile's first K tile is landing meanwhile): per 16x16 tile PAIR a lane holds
    // 8 consecutive n (see `setup`) of row m = lrow: 16-byte bias / residual loads and stores (the store tail of this
    // kernel is issue-bound - 32 dwordx2 stores per lane cost more than the whole K = 256 main loop)
    {
      const int mt = xcd + 8 * (it / tiles_n), nt = it % tiles_n;
      const int mb = mt * G_BM + wm * WM_ROWS + lrow, nb = nt * G_BN + wn * WN_COLS + lq * 8;
      int ncl[TNH];
      bool nok[TNH];
      f32x4 bv[TNH][2];
#pragma unroll
      for (int jp = 0; jp < TNH; ++jp) {
        const int n = nb + jp * 32;
        nok[jp] = n < p.N;                         // N % 8 == 0: a group of 8 is valid or not as a whole
        ncl[jp] = nok[jp] ? n : 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) bv[jp][q] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + ncl[jp] + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      auto finish = [&](auto actf, auto mode) {
        constexpr int MODE = decltype(mode)::value;        // 0: plain, 1: also store the GELU derivative (p.y2), 2: ACT_MUL (res multiplies) - conv_gemm.h
        if constexpr (LIMBS) {                     // fp32 rows: the lane's 8 consecutive channels = two 16-byte accesses
          float* const Yf = reinterpret_cast<float*>(p.y);
          const float* const Rf = reinterpret_cast<const float*>(p.res);
#pragma unroll
          for (int i = 0; i < 2 * TMH; ++i) {
            const int m = mb + i * 16;
            const bool mok = m < p.M;
            const size_t rowoff = (size_t)(mok ? m : p.M - 1) * p.y_cstride;
            f32x4 rf[TNH][2];
#pragma unroll
            for (int jp = 0; jp < TNH; ++jp)
#pragma unroll
              for (int q = 0; q < 2; ++q) rf[jp][q] = Rf ? *reinterpret_cast<const f32x4*>(Rf + rowoff + ncl[jp] + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jp = 0; jp < TNH; ++jp)
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                f32x4 v = acc[i][2 * jp + q] + bv[jp][q];
                acc[i][2 * jp + q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  float x = p.res_first ? v[e] + rf[jp][q][e] : v[e];
                  x = actf(x);
                  if (!p.res_first) x += rf[jp][q][e];
                  v[e] = x;
                }
                if (mok && nok[jp]) *reinterpret_cast<f32x4*>(Yf + rowoff + ncl[jp] + 4 * q) = v;
              }
          }
        } else {
#pragma unroll
        for (int i = 0; i < 2 * TMH; ++i) {
          const int m = mb + i * 16;
          const bool mok = m < p.M;
          const size_t rowoff = (size_t)(mok ? m : p.M - 1) * p.y_cstride;
          u32x4 rv[TNH];
#pragma unroll
          for (int jp = 0; jp < TNH; ++jp) rv[jp] = R ? *reinterpret_cast<const u32x4*>(R + rowoff + ncl[jp]) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
          for (int jp = 0; jp < TNH; ++jp) {
            const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[jp]);
            bf16x8 o, o2;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              f32x4 v = acc[i][2 * jp + q] + bv[jp][q];
              acc[i][2 * jp + q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float r = (float)r8[4 * q + e];
                if constexpr (MODE == 2) o[4 * q + e] = (bf16)(v[e] * r);
                else if constexpr (MODE == 1) { float d; o[4 * q + e] = (bf16)gelu_sig_d(v[e], d); o2[4 * q + e] = (bf16)d; }
                else {
                  float x = p.res_first ? v[e] + r : v[e];
                  x = actf(x);
                  if (!p.res_first) x += r;
                  o[4 * q + e] = (bf16)x;
                }
              }
            }
            if (mok && nok[jp]) {
              *reinterpret_cast<bf16x8*>(Y + rowoff + ncl[jp]) = o;
              if constexpr (MODE == 1) *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.y2) + rowoff + ncl[jp]) = o2;
            }
          }
        }
        }
      };
      typedef std::integral_constant<int, 0> M0_t;
      typedef std::integral_constant<int, 1> M1_t;
      typedef std::integral_constant<int, 2> M2_t;
      if (p.act == ACT_GELU && p.y2 && !LIMBS) finish([](float x) { return x; }, M1_t{});
      else if (p.act == ACT_GELU) finish([](float x) { return LIMBS ? gelu_erfc(x) : gelu_sig(x); }, M0_t{});
      else if (p.act == ACT_LRELU) finish([](float x) { return x > 0.0f ? x : 0.1f * x; }, M0_t{});
      else if (p.act == ACT_MUL && !LIMBS) finish([](float x) { return x; }, M2_t{});
      else finish([](float x) { return x; }, M0_t{});
    }
    if (!has_next) break;
    it = itn;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) { offA[h][j] = offAn[h][j]; offB[h][j] = offBn[h][j]; }
  }
}

__global__ __launch_bounds__(512, 1) void gemm256_kernel(const ConvGemmParams p, const int tiles_m, const int tiles_n) { gemm256_body<false>(p, tiles_m, tiles_n); }
__global__ __launch_bounds__(512, 1) void gemm256_x2_kernel(const ConvGemmParams p, const int tiles_m, const int tiles_n) { gemm256_body<true>(p, tiles_m, tiles_n); }

// dtype 1: the 16-bit kernel; dtype 2: the two-limb kernel on fp32 storage (4-byte operand elements)
bool gemm256_eligible(const ConvGemmParams& p, int dtype) {
  constexpr bool off = false;
  if (off || (dtype != 1 && dtype != 2)) return false;
  const int es = dtype == 2 ? 4 : 2, bke = 128 / es;
  if (p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad != 0 || p.groups != 1) return false;
  if (p.x2 || p.K2 || p.pool2 || p.y_rpi || p.pos || p.out_f32 || p.w_rstride || p.w_gstride) return false;
  if (p.N < 192 || p.M < 1024 || (p.N & 7) || (p.y_cstride & 7)) return false;   // 16-byte epilogue accesses
  if (p.K != p.Kw || (p.Kw % bke) || p.Kw < 2 * bke) return false;            // whole 128-byte K tiles on both operands
  if ((size_t)p.M * p.x_cstride * es >= (1ull << 32) || (size_t)p.N * p.Kw * es >= (1ull << 32)) return false;   // 32-bit DMA offsets
  if (dtype == 2 && (p.y2 || p.act == ACT_MUL)) return false;
  if (dtype == 2) {
    // two-limb mode: conv_gemm_v2's 128 x 128 tile stages 32 flop per byte and runs at 100 .. 270 TFLOP/s; every dense 1x1 layer wide enough for
    // the 256-wide tile comes here
    return true;
  }
  // One 8-wave workgroup per CU cannot hide its epilogue behind another workgroup's MFMAs, so the HBM-bound layers
  // (few flops per streamed byte: the N = 256 / residual projections) stay on conv_gemm_v2 (2-3 workgroups per CU);
  // measured crossover ~200 flop/B (profiles/r01_gemm256_layers.txt); 190 keeps the stage-2 qkv layer with 48-wide heads (N = 864, 197.5 flop/B) here.
  const double min_ai = 190.0;
  const double ai = (double)p.N * p.K / ((double)p.K + (double)p.N * ((p.res && p.act != ACT_MUL) ? 2.0 : 1.0));
  return ai >= min_ai;
}

template <typename KernT>
static int launch_gemm256_t(KernT kern, const ConvGemmParams& p, hipStream_t stream) {
  const int tiles_m = (p.M + G_BM - 1) / G_BM, tiles_n = (p.N + G_BN - 1) / G_BN;
  const long items = (long)tiles_m * tiles_n;
  long grid = 256;                                  // persistent: one 8-wave workgroup per CU
  if (items < grid) grid = (items + 7) / 8 * 8;
  {    // per launch: the attribute is per device
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G_STAGE);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), 2 * G_STAGE, stream, p, tiles_m, tiles_n);
  return (int)hipGetLastError();
}
int launch_gemm256(const ConvGemmParams& p, hipStream_t stream) {
  return launch_gemm256_t(gemm256_kernel, p, stream);
}
int launch_gemm256_x2(const ConvGemmParams& p, hipStream_t stream) {
  return launch_gemm256_t(gemm256_x2_kernel, p, stream);
}

}  // namespace FSVIT_NS
