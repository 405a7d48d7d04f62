// Implicit-GEMM convolution, v2: persistent workgroups + direct-to-LDS (LDS-DMA) staging.
//
// Round 1's first kernel (register-staged, one tile per workgroup; rocprof r01_v1: 515 TF/s at K=N=2048,
// 100-260 TF/s on the K<=256 layers) was bound by its data movement, which is what this design is about:
//   * both operand tiles are staged with `global_load_lds_dwordx4` (16 B per lane, 1 KiB per wave
//     instruction, no VGPR round trip, no ds_write pass).  The LDS image is lane-linear, so the
//     (row & 7) XOR swizzle that keeps ds_read_b128 conflict-free is applied to the per-lane SOURCE
//     chunk instead: lane l of a wave fills chunk slot (l & 7) of row (l >> 3) with global chunk
//     (l & 7) ^ (l >> 3).  Out-of-image taps / tile tails read a 16-byte zero page.
//   * workgroups are persistent: each walks a list of (m-tile, n-tile) items and runs ONE software
//     pipeline over the flattened (item, k-slice) sequence, so the first K slice of the next tile
//     streams in under the last MFMAs and the epilogue of the current one — the prologue/epilogue
//     bubbles that dominated the small-K layers disappear.
//   * the item list is XCD-aware: workgroup b runs on XCD b % 8 (observed placement, used for speed
//     only), and each XCD owns whole m-tiles and visits all their n-tiles back to back, so an
//     activation tile is fetched from HBM once and re-read by its other n-tiles from that XCD's L2.
#include <stdlib.h>

#include "conv_gemm.h"
#include <type_traits>

#include "fsvit_common.h"

namespace FSVIT_NS {

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA issued through inline asm so hipcc's waitcnt pass does not see it: with the builtin, any
// ordinary global load in the loop (the epilogue's bias fetch was enough) made the pass put
// `s_waitcnt vmcnt(0)` in front of the first ds_read of EVERY k-step, i.e. drain the DMA right after
// issuing it (cdna_hip_programming.md 5 "three .s-level traps" (b)).  Hidden from the pass, the DMA
// is ordered by hand: dma_wait_all() before the barrier that publishes a buffer.  The compiler's own
// counted waits for its ordinary loads stay safe: extra in-flight operations only make vmcnt(N)
// stricter.  M0 (the LDS destination base) is written in the same statement that consumes it.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_byte_addr) {
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
// N LDS-DMAs to destinations lds_byte_addr + i * 4096 in ONE statement: M0 is saved / restored once and
// advanced with s_add (the per-DMA form spends 5 scalar instructions on M0 for every load).
__device__ __forceinline__ void dma16x4(const void* s0, const void* s1, const void* s2, const void* s3, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %5\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_add_u32 m0, m0, 0x1000\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, off\n\t"
      "s_add_u32 m0, m0, 0x1000\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %3, off\n\t"
      "s_add_u32 m0, m0, 0x1000\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %4, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "s"(lds_byte_addr)
      : "memory", "scc");
}
__device__ __forceinline__ void dma16x2(const void* s0, const void* s1, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_add_u32 m0, m0, 0x1000\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(s0), "v"(s1), "s"(lds_byte_addr)
      : "memory", "scc");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(lptr_t)p; }

// ---- two-limb ("x2") arithmetic on fp32 storage: every fp32 operand value v is fed to the 16-bit MFMA as the pair (hi, lo), hi = v rounded
// to the 16-bit type, lo = (v - hi) rounded to it (v - hi is exact in fp32), interleaved along K: one 16-byte chunk = 4 fp32 values = 8
// MFMA k-slots [lo0, hi0, lo1, hi1, ...].  MFMA(w, x) then sums w_lo x_lo + w_hi x_hi and MFMA(w, rot16(x)) sums w_lo x_hi + w_hi x_lo:
// two 16x16x32 MFMAs per chunk give the full (w_hi + w_lo)(x_hi + x_lo) with fp32 accumulation, where the exact-fp32 path needs four
// v_mfma_f32_16x16x4_f32 at 1/16 of the rate (dense peaks: 2500 / 4 = 625 TFLOP/s against 157).  Operand precision 16 significand bits
// with bf16 limbs, 22 with fp16 limbs.  The weights arrive pre-split (engine.hip upload()); the activations stay plain fp32
// in HBM - every other kernel of the fp32 path is shared - and are split on their way into LDS (stash() in the kernel): x2_split() returns
// the chunk in both slot orders.
// STATS (training forward, ConvGemmParams::stats): per-lane running sums of the stored values and their squares over every item of the
// workgroup, reduced to ONE partial row per workgroup at the end - the BatchNorm behind the layer needs no reduce pass.  A workgroup's items
// j = slot, slot + P, ... all have the same n-tile when P % tiles_n == 0 (the launcher checks), so the sums stay per channel.
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int MINB, bool STATS = false>
__global__ __launch_bounds__(256, MINB) void conv_gemm_v2_kernel(const ConvGemmParams p, const int tiles_m, const int tiles_n) {
  constexpr int EPC = Elem<T>::kPerChunk;
  constexpr int BKE = Elem<T>::kBK;
  constexpr int TM = BM / WAVES_M / 16;
  constexpr int TN = BN / WAVES_N / 16;
  constexpr int A_IT = BM / 32;
  constexpr int B_IT = BN / 32;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr bool LIMBS = sizeof(T) == 4 && !__is_same(T, float);      // T = f32x2l: fp32 storage, two-limb MFMA arithmetic

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * 128];
  unsigned char* const ldsA0 = smem;
  unsigned char* const ldsB0 = smem + 2 * BM * 128;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lrow = lane & 15, lq = lane >> 4, rswz = lane & 7;
  const int srow = lane >> 3;                  // staging: row inside the wave's 8-row group
  const int sc = (lane & 7) ^ srow;            // staging: global 16-byte chunk this lane fetches (pre-swizzled)

  // ---- work list of this workgroup
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, P = gridDim.x >> 3;
  const int MG = p.groups * tiles_m;                            // (group, m-tile) pairs, dealt round-robin to XCDs
  const int cnt = MG > xcd ? ((MG - xcd + 7) / 8) * tiles_n : 0;   // items owned by this XCD
  if (slot >= cnt) {
    if constexpr (STATS) {                                        // an idle workgroup still owns a partial row
      for (int i = t; i < 2 * p.y_cstride; i += 256) p.stats[(size_t)blockIdx.x * 2 * p.y_cstride + i] = 0.f;
    }
    return;
  }

  const int ohw = p.OH * p.OW;
  const int nk_main = p.Kw / BKE - (p.K2 > 0 ? 1 : 0);   // K slices of the main operand
  const int nk = p.Kw / BKE;                             // + the optional tail slice
  const bool multi_tap = (p.KH * p.KW) > 1;
  const T* const zero = reinterpret_cast<const T*>(g_zero_page);

  // ---- load cursor state (one pipeline step ahead of the compute cursor)
  int iy0[A_IT], ix0[A_IT], pixbase[A_IT], pixnat[A_IT];
  bool rowok[A_IT];
  const T* wrow[B_IT];
  bool nok[B_IT];
  const T* Xg = nullptr;
  auto setup = [&](int j) {
    const int mg = xcd + 8 * (j / tiles_n), nt = j % tiles_n;
    const int g = mg / tiles_m, mt = mg - g * tiles_m;
    Xg = reinterpret_cast<const T*>(p.x) + (size_t)g * p.Cin;
    const long wrs = p.w_rstride ? p.w_rstride : (long)p.Kw;
    const T* Wg = reinterpret_cast<const T*>(p.w) + (p.w_rstride ? (long)g * p.w_gstride : (long)g * p.N * p.Kw);
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int m = mt * BM + 32 * i + 8 * wave + srow;
      rowok[i] = m < p.M;
      const int mm = rowok[i] ? m : 0;
      const int b = mm / ohw;
      const int rem = mm - b * ohw;
      int oy, ox;
      if (p.pool2) {                       // rem = (py * PW + px) * 4 + dy * 2 + dx
        const int win = rem >> 2, pw2 = p.OW >> 1;
        const int py = win / pw2;
        oy = 2 * py + ((rem >> 1) & 1);
        ox = 2 * (win - py * pw2) + (rem & 1);
      } else {
        oy = rem / p.OW;
        ox = rem - oy * p.OW;
      }
      pixnat[i] = b * ohw + oy * p.OW + ox;
      iy0[i] = oy * p.stride - p.pad;
      ix0[i] = ox * p.stride - p.pad;
      pixbase[i] = b * p.H * p.W;
    }
#pragma unroll
    for (int jj = 0; jj < B_IT; ++jj) {
      const int n = nt * BN + 32 * jj + 8 * wave + srow;
      nok[jj] = n < p.N;
      wrow[jj] = Wg + (long)(nok[jj] ? n : 0) * wrs + sc * EPC;
    }
  };
  // Two-limb mode: the activation slice goes through registers - fetched with ordinary loads in issue(), split into (hi, lo) limb words
  // ONCE per element and written to the slot the DMA would have filled in stash() - so the LDS-DMA engine (the ~16 B/clk/CU fill limit of
  // this kernel family, gemm256.hip) carries only the weight half of the tile and no wave repeats another's split.
  u32x4 areg[A_IT];
  auto stash = [&](int buf) {
    if constexpr (LIMBS) {
      unsigned char* const dst = ldsA0 + buf * (BM * 128) + wave * (8 * 128) + lane * 16;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        u32x4 xs, xr;
        x2_split(areg[i], xs, xr);
        *reinterpret_cast<u32x4*>(dst + i * (32 * 128)) = xs;
      }
    }
  };
  auto issue = [&](int kt, int buf) {
    const int k = kt * BKE + sc * EPC;
    int ky = 0, kx = 0, cc = k;
    if (multi_tap) {
      const int tap = k >> p.log2Cin;
      cc = k & (p.Cin - 1);
      ky = tap / p.KW;
      kx = tap - ky * p.KW;
    }
    const bool kok = k < p.K;            // (main operand only; the tail slice has its own bound)
    const unsigned la = __builtin_amdgcn_readfirstlane(lds_addr(ldsA0) + buf * (BM * 128) + wave * (8 * 128));
    const unsigned lb = __builtin_amdgcn_readfirstlane(lds_addr(ldsB0) + buf * (BN * 128) + wave * (8 * 128));
    if (kt < nk_main) {
      const T* srcs[A_IT];
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int iy = iy0[i] + ky, ix = ix0[i] + kx;
        const bool ok = kok && rowok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        srcs[i] = ok ? Xg + (size_t)(pixbase[i] + iy * p.W + ix) * p.x_cstride + cc : zero;
      }
      static_assert(A_IT == 4, "A tile = 128 rows = 4 row groups per wave");
      if constexpr (LIMBS) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) areg[i] = *reinterpret_cast<const u32x4*>(srcs[i]);
      } else dma16x4(srcs[0], srcs[1], srcs[2], srcs[3], la);
    } else {                               // tail operand: row m of x2, columns sc*EPC .. (no spatial gather)
      const T* X2 = reinterpret_cast<const T*>(p.x2);
      const bool cok = sc * EPC < p.K2;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const T* src = (cok && rowok[i]) ? X2 + (size_t)pixnat[i] * p.x2_cstride + sc * EPC : zero;
        if constexpr (LIMBS) areg[i] = *reinterpret_cast<const u32x4*>(src);
        else dma16(src, la + i * (32 * 128));
      }
    }
    const T* bs[B_IT];
#pragma unroll
    for (int jj = 0; jj < B_IT; ++jj) bs[jj] = nok[jj] ? wrow[jj] + (size_t)kt * BKE : zero;
    if constexpr (B_IT == 4) dma16x4(bs[0], bs[1], bs[2], bs[3], lb);
    else if constexpr (B_IT == 2) dma16x2(bs[0], bs[1], lb);
    else {
#pragma unroll
      for (int jj = 0; jj < B_IT; ++jj) dma16(bs[jj], lb + jj * (32 * 128));
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 st0[STATS ? TN : 1], st1[STATS ? TN : 1];
#pragma unroll
  for (int j = 0; j < (STATS ? TN : 1); ++j) st0[j] = st1[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const unsigned char* a = ldsA0 + buf * (BM * 128) + (wm * TM * 16 + lrow) * 128;
    const unsigned char* b = ldsB0 + buf * (BN * 128) + (wn * TN * 16 + lrow) * 128;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      const int off = (((kc * 4 + lq) ^ rswz) << 4);
      u32x4 xf[TM], wf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const u32x4*>(a + i * 16 * 128 + off);
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const u32x4*>(b + j * 16 * 128 + off);
      if constexpr (LIMBS) {              // xf = limb words [hi | lo] (stash()); the cross terms pair them with the half-swapped weight words
        u32x4 wr[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) wr[j][e] = __builtin_amdgcn_alignbit(wf[j][e], wf[j][e], 16);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = mma_chunk<bf16>(wr[j], xf[i], acc[i][j]);     // lo x hi + hi x lo first, hi x hi + lo x lo last
            acc[i][j] = mma_chunk<bf16>(wf[j], xf[i], acc[i][j]);
          }
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mma_chunk<T>(wf[j], xf[i], acc[i][j]);
      }
    }
  };

  T* __restrict__ Y = reinterpret_cast<T*>(p.y);
  const T* __restrict__ R = reinterpret_cast<const T*>(p.res);
  // Epilogue in two passes so no load sits behind a dependent wait: (1) issue every bias / residual /
  // pos-embed load of the tile back to back, (2) arithmetic + stores.  The activation is selected by
  // ONE wave-uniform branch around the whole pass (a per-element switch compiles to a branch tree).
  auto epilogue = [&](int j) {
    const int mg = xcd + 8 * (j / tiles_n), nt = j % tiles_n;
    const int g = mg / tiles_m, mt = mg - g * tiles_m;
    const int cg = g * p.N;
    const int nb = nt * BN + wn * TN * 16 + lq * 4;
    const int mb = mt * BM + wm * TM * 16 + lrow;
    // Loads are UNCONDITIONAL on clamped (always in-bounds) coordinates so they issue back to back
    // with a single wait; a per-lane `ok ? load : 0` becomes an exec-masked branch + vmcnt(0) per
    // load.  Only the stores are predicated.
    int ncl[TN];
    bool nk_[TN];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      const int n = nb + jn * 16;
      nk_[jn] = n < p.N;
      ncl[jn] = cg + (nk_[jn] ? n : p.N - 4);
    }
    f32x4 bv[TN];
    if (p.bias) {
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) bv[jn] = *reinterpret_cast<const f32x4*>(p.bias + ncl[jn]);
    } else {
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) bv[jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    constexpr int HB = TM > 2 ? 2 : TM;
    auto batch = [&](int i0, auto actf, auto mode) {
      constexpr int MODE = decltype(mode)::value;          // 0: plain, 1: also store the GELU derivative (y2), 2: ACT_MUL (res multiplies)
      f32x4 rv[HB][TN];
      size_t rowoff[HB];
      bool mk_[HB];
#pragma unroll
      for (int i = 0; i < HB; ++i) {
        const int m = mb + (i0 + i) * 16;
        mk_[i] = m < p.M;
        const int mc = mk_[i] ? m : p.M - 1;
        int orow = p.pool2 ? (mc >> 2) : mc;                            // pool2: 4 consecutive rows = one 2x2 window
        if (p.y_rpi) orow = (orow / ohw) * p.y_rpi + p.y_row0 + orow % ohw;
        rowoff[i] = (size_t)orow * p.y_cstride;
      }
      if (R) {
#pragma unroll
        for (int i = 0; i < HB; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) rv[i][jn] = load4<T>(R + rowoff[i] + ncl[jn]);
      } else {
#pragma unroll
        for (int i = 0; i < HB; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) rv[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < HB; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) { acc[i0 + i][jn] = (acc[i0 + i][jn] + bv[jn]) * rv[i][jn] - bv[jn]; rv[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      } else if (p.res_first) {
#pragma unroll
        for (int i = 0; i < HB; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) { acc[i0 + i][jn] += rv[i][jn]; rv[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      }
      if (p.pos) {
#pragma unroll
        for (int i = 0; i < HB; ++i) {
          const int m = mk_[i] ? mb + (i0 + i) * 16 : p.M - 1;
          const float* posrow = p.pos + (size_t)(p.pool2 ? (m >> 2) % (ohw >> 2) : m % ohw) * p.y_cstride;
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) rv[i][jn] += *reinterpret_cast<const f32x4*>(posrow + ncl[jn]);
        }
      }
#pragma unroll
      for (int i = 0; i < HB; ++i) {
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
          f32x4 v = acc[i0 + i][jn] + bv[jn];
          acc[i0 + i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (STATS) {
            const float mk = mk_[i] ? 1.0f : 0.0f;                 // rows past M are computed on clamped coordinates: not part of the map
            st0[jn] += v * mk;
            st1[jn] += v * v * mk;
          }
          f32x4 dv = {0.f, 0.f, 0.f, 0.f};
          if constexpr (MODE == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float d; v[e] = sizeof(T) == 2 ? gelu_sig_d(v[e], d) : gelu_erf_d(v[e], d); dv[e] = d; }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = actf(v[e]);
          }
          if (p.pool2) {                   // the window's 4 pixels sit in lanes lrow = 4j .. 4j+3 of the same lq
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = fmaxf(v[e], __shfl_xor(v[e], 1, 64));
              v[e] = fmaxf(v[e], __shfl_xor(v[e], 2, 64));
            }
          }
          v += rv[i][jn];
          if (mk_[i] && nk_[jn] && (!p.pool2 || (lrow & 3) == 0)) {
            if (p.out_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + rowoff[i] + ncl[jn]) = v;
            else store4<T>(Y + rowoff[i] + ncl[jn], v);
            if constexpr (MODE == 1) store4<T>(reinterpret_cast<T*>(p.y2) + rowoff[i] + ncl[jn], dv);
          }
        }
      }
    };
    auto finish = [&](auto actf, auto mode) {
#pragma unroll
      for (int i0 = 0; i0 < TM; i0 += HB) batch(i0, actf, mode);
    };
    typedef std::integral_constant<int, 0> M0_t;
    typedef std::integral_constant<int, 1> M1_t;
    typedef std::integral_constant<int, 2> M2_t;
    if (p.act == ACT_GELU && p.y2) finish([](float x) { return x; }, M1_t{});
    else if (p.act == ACT_GELU) finish([](float x) { return sizeof(T) == 2 ? gelu_sig(x) : (LIMBS ? gelu_erfc(x) : gelu_erf(x)); }, M0_t{});
    else if (p.act == ACT_LRELU) finish([](float x) { return x > 0.0f ? x : 0.1f * x; }, M0_t{});
    else if (p.act == ACT_MUL) finish([](float x) { return x; }, M2_t{});
    else finish([](float x) { return x; }, M0_t{});
  };

  // ---- one software pipeline over the flattened (item, k-slice) sequence.  The inner k-loop holds
  // only DMA issue + LDS reads + MFMA (no ordinary VMEM access may sit in it, or hipcc drains vmcnt
  // at its header every step); the first slice of the NEXT item is issued before the last MFMAs and
  // the epilogue of the current one.
  int buf = 0;
  setup(slot);
  issue(0, 0);
  stash(0);
  dma_wait_all();
  __syncthreads();                       // buffer 0 landed in every wave's view
  for (int j = slot; j < cnt; j += P) {
    for (int kt = 0; kt + 1 < nk; ++kt) {
      issue(kt + 1, buf ^ 1);
      compute(buf);
      stash(buf ^ 1);
      dma_wait_all();                    // this wave's share of the next buffer has landed ...
      __syncthreads();                   // ... so has everyone's, and everyone is done reading `buf`
      buf ^= 1;
    }
    const int jn = j + P;
    if (jn < cnt) {
      setup(jn);
      issue(0, buf ^ 1);
    }
    compute(buf);
    epilogue(j);
    if (jn < cnt) stash(buf ^ 1);
    dma_wait_all();
    __syncthreads();
    buf ^= 1;
  }
  if constexpr (STATS) {
    // the 16 pixel lanes of a channel quad (shuffles), then the WAVES_M waves that share the channels, in fixed order (bit-reproducible)
    float* red = reinterpret_cast<float*>(smem);                 // [4 waves][2][TN * 16 channels]
    constexpr int CW = TN * 16;
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = st0[jn][e], q = st1[jn][e];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); q += __shfl_xor(q, o); }
        if (lrow == 0) {
          red[(wave * 2 + 0) * CW + jn * 16 + lq * 4 + e] = a;
          red[(wave * 2 + 1) * CW + jn * 16 + lq * 4 + e] = q;
        }
      }
    __syncthreads();
    const int nt0 = (slot % tiles_n) * BN;                       // this workgroup's n-tile; the other channels of its partial row are zero
    for (int i = t; i < 2 * p.y_cstride; i += 256) {
      const int which = i / p.y_cstride, n = i % p.y_cstride, nl = n - nt0;
      float a = 0.f;
      if (nl >= 0 && nl < BN) {
        const int w_n = nl / CW, cl = nl % CW;
#pragma unroll
        for (int m = 0; m < WAVES_M; ++m) a += red[((m * WAVES_N + w_n) * 2 + which) * CW + cl];
      }
      p.stats[((size_t)blockIdx.x * 2 + which) * p.y_cstride + n] = a;
    }
  }
}

template <int BM, int BN, int MINB>
static long v2_grid(const ConvGemmParams& p) {
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const long items = (long)p.groups * tiles_m * tiles_n;
  long grid = 256L * MINB;                       // persistent: MINB workgroups per CU
  if (items < grid) grid = (items + 7) / 8 * 8;
  return grid;
}
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int MINB>
static int launch_cfg(const ConvGemmParams& p, hipStream_t stream) {
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const long grid = v2_grid<BM, BN, MINB>(p);
  hipLaunchKernelGGL((conv_gemm_v2_kernel<T, BM, BN, WAVES_M, WAVES_N, MINB>), dim3((unsigned)grid), dim3(256), 0, stream, p, tiles_m, tiles_n);
  return (int)hipGetLastError();
}
// The statistics variant: 16-bit, 128 x 64 tile at two workgroups per CU (the running sums need 16 more registers than the three-per-CU budget has;
// the 128 x 128 tile has none to spare).  Partial rows it writes into ConvGemmParams::stats, 0 = not produced for this layer.
int conv_gemm_v2_stats_rows(const ConvGemmParams& p, int dtype) {
  if (dtype != 1 || p.groups != 1 || p.pool2 || p.act != ACT_NONE || p.res || p.pos || p.y_rpi || p.out_f32 || p.y2 || (p.N % 64) || p.N > 128 || p.N != p.y_cstride || p.M <= 0) return 0;
  const long grid = v2_grid<128, 64, 2>(p);
  return (grid / 8) % (p.N / 64) == 0 ? (int)grid : 0;
}
static int launch_v2_stats(const ConvGemmParams& p, hipStream_t stream) {
  const int tiles_m = (p.M + 127) / 128, tiles_n = p.N / 64;
  const long grid = v2_grid<128, 64, 2>(p);
  hipLaunchKernelGGL((conv_gemm_v2_kernel<bf16, 128, 64, 2, 2, 2, true>), dim3((unsigned)grid), dim3(256), 0, stream, p, tiles_m, tiles_n);
  return (int)hipGetLastError();
}

template <typename T>
static int launch_v2_t(const ConvGemmParams& p, hipStream_t stream) {
  if (p.N > 64) return launch_cfg<T, 128, 128, 2, 2, 2>(p, stream);
  if (p.N > 32) return launch_cfg<T, 128, 64, 2, 2, 3>(p, stream);
  return launch_cfg<T, 128, 32, 4, 1, 3>(p, stream);
}

// 1: 128x64, 2: 128x32, 3: 128x128 (0 is reserved for a 256x128 / 8-wave / 3-stage-ring variant that was
// measured and dropped in round 1: +6 % at K = N = 2048 but -10..-25 % on the conv / small-K layers) -- mirrors launch_v2_t
int conv_gemm_v2_config(const ConvGemmParams& p) { return p.N > 64 ? 3 : (p.N > 32 ? 1 : 2); }

int launch_conv_gemm_v2(const ConvGemmParams& p, int dtype, hipStream_t stream) {
  if (p.M <= 0) return 0;
  if (p.pool2 && (p.res || p.y_rpi || (p.OH & 1) || (p.OW & 1) || p.stride != 1)) return (int)hipErrorInvalidValue;
  if (dtype == 2) return launch_v2_t<f32x2l>(p, stream);      // fp32 storage, two-limb 16-bit MFMA arithmetic
  if (p.stats) return conv_gemm_v2_stats_rows(p, dtype) ? launch_v2_stats(p, stream) : (int)hipErrorInvalidValue;
  return dtype == 0 ? launch_v2_t<float>(p, stream) : launch_v2_t<bf16>(p, stream);
}

// The dispatcher every conv / GEMM launch goes through: the LDS-resident halo kernel for the stem's 3x3 convs, the 256x256 tile for
// the dense 1x1 layers, this file's persistent implicit GEMM for everything else.
// rows per gemm256 slice of an over-long plain 1x1 layer, 0: not sliced (see launch_conv_gemm)
static int gemm256_slice_rows(const ConvGemmParams& p, int dtype) {
  const size_t es = dtype == 2 ? 4 : 2;
  const size_t row_bytes = (size_t)p.x_cstride * es;
  if ((dtype != 1 && dtype != 2) || p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad != 0 || p.groups != 1 || p.pos || p.y_rpi || p.x2 || p.pool2) return 0;
  if ((size_t)p.M * row_bytes < (1ull << 32)) return 0;
  const int rows_max = (int)((((1ull << 32) - 1) / row_bytes) / 256 * 256);
  ConvGemmParams q = p;
  q.B = rows_max < p.M ? rows_max : p.M; q.H = q.W = q.OH = q.OW = 1; q.M = q.B;
  return rows_max >= 1024 && gemm256_eligible(q, dtype) ? rows_max : 0;
}
static int run_gemm256(const ConvGemmParams& p, int dtype, hipStream_t stream) { return dtype == 2 ? launch_gemm256_x2(p, stream) : launch_gemm256(p, stream); }
// which kernel launch_conv_gemm runs for p: 0 conv3x3_halo, 1 gemm256 (whole or in row slices), 2 conv_gemm_v2, 3 gconv3x3_x2 (two-limb grouped 3x3)
int conv_gemm_route(const ConvGemmParams& p, int dtype) {
  if (conv3x3_halo_eligible(p, dtype)) return 0;
  if (gconv3x3_x2_eligible(p, dtype)) return 3;
  if (gemm256_eligible(p, dtype) || gemm256_slice_rows(p, dtype)) return 1;
  return 2;
}

// partial rows ConvGemmParams::stats receives when launch_conv_gemm runs p with stats set (follows the routing below); 0: the layer's kernel has no
// statistics epilogue and the caller must not set `stats`
int conv_stats_rows(const ConvGemmParams& p, int dtype) {
  ConvGemmParams q = p;
  float dummy;
  q.stats = &dummy;
  if (conv3x3_halo_eligible(q, dtype)) return conv3x3_halo_stats_rows(p, dtype);
  if (gconv3x3_x2_eligible(q, dtype) || gemm256_eligible(q, dtype) || gemm256_slice_rows(q, dtype)) return 0;
  return conv_gemm_v2_stats_rows(p, dtype);
}

int launch_conv_gemm(const ConvGemmParams& p, int dtype, hipStream_t stream) {
  if (conv3x3_halo_eligible(p, dtype)) return launch_conv3x3_halo(p, stream);
  if (p.x_planar || p.y_planar) return (int)hipErrorInvalidValue;      // only conv3x3_halo reads / writes the row-chunk-planar layout
  if (gconv3x3_x2_eligible(p, dtype)) return launch_gconv3x3_x2(p, stream);
  if (gemm256_eligible(p, dtype)) return run_gemm256(p, dtype, stream);
  // A plain 1x1 layer whose activation matrix is past gemm256's 32-bit DMA offsets (a 12 800-image ViT chunk: 2.5 M rows x 1536 columns)
  // runs as row slices of < 4 GB each, if a slice is eligible.
  if (const int rows_max = gemm256_slice_rows(p, dtype)) {
    const size_t es = dtype == 2 ? 4 : 2;
    ConvGemmParams q = p;
    q.H = q.W = q.OH = q.OW = 1;
    for (int m0 = 0; m0 < p.M; m0 += rows_max) {
      q.M = q.B = p.M - m0 < rows_max ? p.M - m0 : rows_max;
      q.x = (const unsigned char*)p.x + (size_t)m0 * p.x_cstride * es;
      q.y = (unsigned char*)p.y + (size_t)m0 * p.y_cstride * es;
      q.res = p.res ? (const unsigned char*)p.res + (size_t)m0 * p.y_cstride * es : nullptr;
      q.y2 = p.y2 ? (unsigned char*)p.y2 + (size_t)m0 * p.y_cstride * es : nullptr;
      const int rc = gemm256_eligible(q, dtype) ? run_gemm256(q, dtype, stream) : launch_conv_gemm_v2(q, dtype, stream);
      if (rc) return rc;
    }
    return 0;
  }
  return launch_conv_gemm_v2(p, dtype, stream);
}

}  // namespace FSVIT_NS
