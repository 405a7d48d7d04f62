// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions of the meta-tuning step (stem conv2 / conv3, the grouped stage-1 conv2;
// the reference gets it from autograd: meta_tuning_sun_m/train_meta.py:228-232 loss.backward() through visformer.py:209-237, :152-163):
//     dW[o][i][ky][kx] = sum_m dz[m][o] * x[pix(m) + (ky-1, kx-1)][i]            m = (b, oy, ox), zero outside the image
// computed directly from the NHWC activations.  Round 1 fed a split-K GEMM with TRANSPOSED operands: dz^T [N][M] and a transposed
// im2col [9 C][M] written to HBM first (profiles/r02_train_kernel_stats.csv: im2col_t 4.75 ms + transposes 1.6 ms of a 34.5 ms step, and
// the grouped conv ran as a dense GEMM with 8 x the useful flops).  The contraction index of this product is the ROW index of both
// operands, i.e. both MFMA fragments want 8 consecutive rows of one column: that is what gfx950's ds_read_b64_tr_b16 delivers from a
// row-major LDS tile (tools/probes/tr_read_probe.hip: within a 16-lane group lane i supplies the address of 4 contiguous columns
// 4 (i & 3) .. of row i >> 2 and receives rows 0..3 of column i).  So:
//   * a workgroup (8 waves) stages 64 rows of dz and the matching window of x pixels (64 + 2 (W + 1), so that every tap's shifted row is
//     present) in LDS as [16-column subtile][row][32 B] - the conflict-free image for the transposing read - through registers, the next
//     chunk's global loads in flight while the current one is multiplied;
//   * wave j owns one (32 output channels) x (32 input channels) x 9 taps block = 36 accumulator tiles (144 VGPRs): a group of the grouped
//     conv, or an (n-block, c-block) pair of a dense one; per 32 rows: 4 transposing reads for dz, 4 per tap for x, 36 MFMAs;
//   * a tap is an ADDRESS: the lane that supplies row r's address adds the tap's pixel shift, or points at a zero chunk when the tap
//     leaves the image - no im2col, no padded copy;
//   * split over row chunks across workgroups: fp32 partials [split][job][tap][32][32], summed in fixed order by wgrad3x3_finalize_kernel
//     into the PyTorch weight layout (deterministic, like the round-1 path).
#include <stdlib.h>

#include "conv_gemm.h"
#include "fsvit_common.h"
#include "kernels.h"

namespace FSVIT_NS {

namespace wg3 {
constexpr int NW = 8;            // waves = (n-block, c-block) jobs per workgroup
constexpr int CH = 64;           // rows per stage = 2 MFMA K steps
constexpr int JOB = 9 * 32 * 32; // fp32 partials per job
typedef short s4 __attribute__((ext_vector_type(4)));
}  // namespace wg3

// NC / CC: dz / x columns staged per workgroup.  (256, 256): grouped conv, wave j = group j.  (128, 64): dense conv, wave j = n-block j & 3,
// c-block j >> 2 of the 64 input channels blockIdx.y * 64 ..
template <int NC, int CC, int MAXW>
__global__ __launch_bounds__(512) void wgrad3x3_kernel(const bf16* __restrict__ x, int xld, const bf16* __restrict__ dz, int zld, float* __restrict__ part,
                                                       int M, int H, int W, int n_chunks, int chunks_per_wg) {
  using namespace wg3;
  constexpr int WINP = CH + 2 * (MAXW + 1) + 3;               // pixels of the x window (W <= MAXW), rounded up to an ODD count: the staging stores walk the
                                                               // 16-column subtiles, and with an even pitch they met on 2 of the 16 bank groups
  constexpr int ZT_BYTES = (NC / 16) * CH * 32;
  constexpr int XW_BYTES = (CC / 16) * WINP * 32;
  constexpr int NPZ = (CH * NC / 8) / 512;                     // 16-byte units of dz per thread and chunk
  constexpr int NPX = (WINP * (CC / 8) + 511) / 512;           // of the x window
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const ZT = smem;
  unsigned char* const XW = smem + ZT_BYTES;
  unsigned char* const ZERO = smem + ZT_BYTES + XW_BYTES;      // a zero row (32 B) here and another one subtile pitch further (the c-tile offset is an immediate)

  const int t = threadIdx.x, lane = t & 63, i = lane & 15, lq = lane >> 4;
  const int j = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nb = NC == 256 ? j : (j & 3), cb = NC == 256 ? j : (j >> 2);
  const int xc0 = NC == 256 ? 0 : blockIdx.y * CC;
  const int halo = W + 1, winp = CH + 2 * halo;                // pixels actually used (<= WINP)
  const int q0 = blockIdx.x * chunks_per_wg;
  int q1 = q0 + chunks_per_wg; q1 = q1 < n_chunks ? q1 : n_chunks;

  if (t < 8) { reinterpret_cast<unsigned*>(ZERO)[t] = 0u; reinterpret_cast<unsigned*>(ZERO + WINP * 32)[t] = 0u; }   // one zero row (32 B) each

  u32x4 pz[NPZ], px[NPX];
  unsigned okm = 0;                                            // validity bits of pz / px (rows past the end / outside the window read a clamped row)
  auto gload = [&](int q) {                                    // chunk q -> registers
    const long m0 = (long)q * CH;
    // UNCONDITIONAL loads on clamped rows, the validity kept as a bit and applied by lstore(): a per-lane `ok ? load : 0` compiles to an exec-masked
    // branch (or a select right behind the load) with its own vmcnt wait, i.e. the chunk's 11 loads went out one latency after the other
    okm = 0;
#pragma unroll
    for (int u0 = 0; u0 < NPZ; ++u0) {
      const int u = t + 512 * u0, r = u / (NC / 8), c8 = u % (NC / 8);
      const long m = m0 + r;
      pz[u0] = *reinterpret_cast<const u32x4*>(dz + (size_t)(m < M ? m : M - 1) * zld + c8 * 8);
      okm |= m < M ? 1u << u0 : 0u;
    }
#pragma unroll
    for (int u0 = 0; u0 < NPX; ++u0) {
      const int u = t + 512 * u0, p = u / (CC / 8), c8 = u % (CC / 8);
      const long m = m0 - halo + p;
      const bool ok = p < winp && m >= 0 && m < M;
      px[u0] = *reinterpret_cast<const u32x4*>(x + (size_t)(ok ? m : 0) * xld + xc0 + c8 * 8);
      okm |= ok ? 1u << (NPZ + u0) : 0u;
    }
  };
  auto lstore = [&]() {                                        // registers -> the subtile images
#pragma unroll
    for (int u0 = 0; u0 < NPZ; ++u0) {
      const int u = t + 512 * u0, r = u / (NC / 8), c8 = u % (NC / 8);
      *reinterpret_cast<u32x4*>(ZT + (c8 >> 1) * (CH * 32) + r * 32 + (c8 & 1) * 16) = (okm >> u0) & 1u ? pz[u0] : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u0 = 0; u0 < NPX; ++u0) {
      const int u = t + 512 * u0, p = u / (CC / 8), c8 = u % (CC / 8);
      if (p < WINP) *reinterpret_cast<u32x4*>(XW + (c8 >> 1) * (WINP * 32) + p * 32 + (c8 & 1) * 16) = (okm >> (NPZ + u0)) & 1u ? px[u0] : u32x4{0u, 0u, 0u, 0u};
    }
  };

  f32x4 acc[9][2][2];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[tp][a >> 1][a & 1] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned zt_lane = (unsigned)(size_t)(__attribute__((address_space(3))) void*)ZT + (nb * 2) * (CH * 32) + (lq * 4 + (i >> 2)) * 32 + (i & 3) * 8;
  const unsigned xw_lane = (unsigned)(size_t)(__attribute__((address_space(3))) void*)XW + (cb * 2) * (WINP * 32) + (i & 3) * 8;
  const unsigned zero_lane = (unsigned)(size_t)(__attribute__((address_space(3))) void*)ZERO + (i & 3) * 8;
  auto tr = [&](unsigned addr) -> u32x2 {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(size_t)addr));
  };
  const int HW = H * W;

  if (q0 < q1) gload(q0);
  for (int q = q0; q < q1; ++q) {
    __syncthreads();                                            // the previous chunk's fragment reads are done
    lstore();
    __syncthreads();
    if (q + 1 < q1) gload(q + 1);
    const long m0 = (long)q * CH;
#pragma unroll 1
    for (int ks = 0; ks < 2; ++ks) {      // (not unrolled: with both steps' tap addresses live the grouped variant spilled 20 registers)
      // dz^T fragments: rows (= K slots) ks*32 + h*16 + lq*4 + 0..3, column 16 nt + i
      u32x4 af[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const u32x2 r1 = tr(zt_lane + nt * (CH * 32) + ks * 1024), r2 = tr(zt_lane + nt * (CH * 32) + ks * 1024 + 512);
        af[nt] = u32x4{r1[0], r1[1], r2[0], r2[1]};
      }
      // this lane SUPPLIES the addresses of rows ks*32 + h*16 + lq*4 + (i >> 2), h = 0, 1
      int oy[2], ox[2];
      unsigned rowa[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int rr = ks * 32 + h * 16 + lq * 4 + (i >> 2);
        const int m = (int)m0 + rr;
        const int rem = m % HW;
        oy[h] = rem / W; ox[h] = rem - oy[h] * W;
        if (m >= M) oy[h] = -4;                                 // rows past the end: every tap invalid (their dz rows are zero anyway)
        rowa[h] = xw_lane + (rr + halo) * 32;
      }
      // per tap: its 4 transposing reads first, then its 4 MFMAs (hipcc had put every MFMA PAIR behind its own two reads and an lgkmcnt(0))
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int dy = tp / 3 - 1, dx = tp % 3 - 1;
        const int shift = (dy * W + dx) * 32;
        unsigned ad[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool ok = (unsigned)(oy[h] + dy) < (unsigned)H && (unsigned)(ox[h] + dx) < (unsigned)W;
          ad[h] = ok ? rowa[h] + shift : zero_lane;
        }
        u32x4 bf[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const u32x2 r1 = tr(ad[0] + ct * (WINP * 32)), r2 = tr(ad[1] + ct * (WINP * 32));
          bf[ct] = u32x4{r1[0], r1[1], r2[0], r2[1]};
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          acc[tp][0][ct] = mma_chunk<bf16>(af[0], bf[ct], acc[tp][0][ct]);
          acc[tp][1][ct] = mma_chunk<bf16>(af[1], bf[ct], acc[tp][1][ct]);
        }
      }
    }
  }

  // fp32 partials of this row range: [split][job][tap][n 32][c 32]; lane holds n = 16 nt + 4 lq + e, c = 16 ct + i
  const int job = (NC == 256 ? 0 : blockIdx.y * NW) + j, njobs = (NC == 256 ? 1 : gridDim.y) * NW;
  float* out = part + ((size_t)blockIdx.x * njobs + job) * JOB;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) out[(tp * 32 + nt * 16 + lq * 4 + e) * 32 + ct * 16 + i] = acc[tp][nt][ct][e];
}

// ---- the same kernel for the two-limb trainers (fp32 activations, `bf16x2` / `f16x2`): a row of dz / a pixel of x becomes two 16-bit LDS rows
// (2 r = its lo limbs, 2 r + 1 = its hi limbs; see wgrad1x1_x2_kernel), a tap shifts by 2 rows per pixel, and a 16-row K step is two MFMAs per
// accumulator tile.  CHX = rows per stage: 64 for the dense layers, 32 for the grouped one (its 256-column window would not fit twice).
template <int NC, int CC, int MAXW, int CHX>
__global__ __launch_bounds__(512) void wgrad3x3_x2_kernel(const float* __restrict__ x, int xld, const float* __restrict__ dz, int zld, float* __restrict__ part,
                                                          int M, int H, int W, int n_chunks, int chunks_per_wg) {
  using namespace wg3;
  constexpr int WINP = CHX + 2 * (MAXW + 1) + 3;
  constexpr int ZSUB = 2 * CHX * 32, XSUB = 2 * WINP * 32;     // subtile pitches (limb rows)
  constexpr int ZT_BYTES = (NC / 16) * ZSUB;
  constexpr int XW_BYTES = (CC / 16) * XSUB;
  constexpr int NPZ = (CHX * NC / 8) / 512;
  constexpr int NPX = (WINP * (CC / 8) + 511) / 512;
  static_assert(NPZ + NPX <= 32, "validity bits");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const ZT = smem;
  unsigned char* const XW = smem + ZT_BYTES;
  unsigned char* const ZERO = smem + ZT_BYTES + XW_BYTES;      // a zero row (32 B) here and another one subtile pitch further

  const int t = threadIdx.x, lane = t & 63, i = lane & 15, lq = lane >> 4;
  const int j = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nb = NC == 256 ? j : (j & 3), cb = NC == 256 ? j : (j >> 2);
  const int xc0 = NC == 256 ? 0 : blockIdx.y * CC;
  const int halo = W + 1, winp = CHX + 2 * halo;
  const int q0 = blockIdx.x * chunks_per_wg;
  int q1 = q0 + chunks_per_wg; q1 = q1 < n_chunks ? q1 : n_chunks;

  if (t < 8) { reinterpret_cast<unsigned*>(ZERO)[t] = 0u; reinterpret_cast<unsigned*>(ZERO + XSUB)[t] = 0u; }

  u32x4 pz[NPZ][2], px[NPX][2];
  unsigned okm = 0;
  auto gload = [&](int q) {
    const long m0 = (long)q * CHX;
    okm = 0;
#pragma unroll
    for (int u0 = 0; u0 < NPZ; ++u0) {
      const int u = t + 512 * u0, r = u / (NC / 8), c8 = u % (NC / 8);
      const long m = m0 + r;
      const float* src = dz + (size_t)(m < M ? m : M - 1) * zld + c8 * 8;
      pz[u0][0] = *reinterpret_cast<const u32x4*>(src);
      pz[u0][1] = *reinterpret_cast<const u32x4*>(src + 4);
      okm |= m < M ? 1u << u0 : 0u;
    }
#pragma unroll
    for (int u0 = 0; u0 < NPX; ++u0) {
      const int u = t + 512 * u0, p = u / (CC / 8), c8 = u % (CC / 8);
      const long m = m0 - halo + p;
      const bool ok = p < winp && m >= 0 && m < M;
      const float* src = x + (size_t)(ok ? m : 0) * xld + xc0 + c8 * 8;
      px[u0][0] = *reinterpret_cast<const u32x4*>(src);
      px[u0][1] = *reinterpret_cast<const u32x4*>(src + 4);
      okm |= ok ? 1u << (NPZ + u0) : 0u;
    }
  };
  auto limbs8 = [&](const u32x4 a, const u32x4 b, u32x4& lo, u32x4& hi) {
    u32x4 sa, sb, r_;
    x2_split(a, sa, r_);
    x2_split(b, sb, r_);
    lo = u32x4{__builtin_amdgcn_perm(sa[1], sa[0], 0x05040100u), __builtin_amdgcn_perm(sa[3], sa[2], 0x05040100u),
               __builtin_amdgcn_perm(sb[1], sb[0], 0x05040100u), __builtin_amdgcn_perm(sb[3], sb[2], 0x05040100u)};
    hi = u32x4{__builtin_amdgcn_perm(sa[1], sa[0], 0x07060302u), __builtin_amdgcn_perm(sa[3], sa[2], 0x07060302u),
               __builtin_amdgcn_perm(sb[1], sb[0], 0x07060302u), __builtin_amdgcn_perm(sb[3], sb[2], 0x07060302u)};
  };
  auto lstore = [&]() {
    const u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int u0 = 0; u0 < NPZ; ++u0) {
      const int u = t + 512 * u0, r = u / (NC / 8), c8 = u % (NC / 8);
      u32x4 lo, hi;
      limbs8(pz[u0][0], pz[u0][1], lo, hi);
      const bool ok = (okm >> u0) & 1u;
      unsigned char* d = ZT + (c8 >> 1) * ZSUB + (2 * r) * 32 + (c8 & 1) * 16;
      *reinterpret_cast<u32x4*>(d) = ok ? lo : z4;
      *reinterpret_cast<u32x4*>(d + 32) = ok ? hi : z4;
    }
#pragma unroll
    for (int u0 = 0; u0 < NPX; ++u0) {
      const int u = t + 512 * u0, p = u / (CC / 8), c8 = u % (CC / 8);
      if (p < WINP) {
        u32x4 lo, hi;
        limbs8(px[u0][0], px[u0][1], lo, hi);
        const bool ok = (okm >> (NPZ + u0)) & 1u;
        unsigned char* d = XW + (c8 >> 1) * XSUB + (2 * p) * 32 + (c8 & 1) * 16;
        *reinterpret_cast<u32x4*>(d) = ok ? lo : z4;
        *reinterpret_cast<u32x4*>(d + 32) = ok ? hi : z4;
      }
    }
  };

  f32x4 acc[9][2][2];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[tp][a >> 1][a & 1] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned zt_lane = (unsigned)(size_t)(__attribute__((address_space(3))) void*)ZT + (nb * 2) * ZSUB + (lq * 4 + (i >> 2)) * 32 + (i & 3) * 8;
  const unsigned xw_lane = (unsigned)(size_t)(__attribute__((address_space(3))) void*)XW + (cb * 2) * XSUB + (i & 3) * 8;
  const unsigned zero_lane = (unsigned)(size_t)(__attribute__((address_space(3))) void*)ZERO + (i & 3) * 8;
  auto tr = [&](unsigned addr) -> u32x2 {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(size_t)addr));
  };
  const int HW = H * W;

  if (q0 < q1) gload(q0);
  for (int q = q0; q < q1; ++q) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (q + 1 < q1) gload(q + 1);
    const long m0 = (long)q * CHX;
#pragma unroll 1
    for (int ks = 0; ks < CHX / 16; ++ks) {                      // 32 limb rows = 16 rows of dz per step (not unrolled: the tap addresses of all steps at once spill)
      u32x4 af[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const u32x2 r1 = tr(zt_lane + nt * ZSUB + ks * 1024), r2 = tr(zt_lane + nt * ZSUB + ks * 1024 + 512);
        af[nt] = u32x4{r1[0], r1[1], r2[0], r2[1]};
      }
      // this lane SUPPLIES the addresses of limb rows ks*32 + h*16 + lq*4 + (i >> 2), h = 0, 1: row rr = that >> 1 of the chunk, limb = its low bit
      int oy[2], ox[2];
      unsigned rowa[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int lr = ks * 32 + h * 16 + lq * 4 + (i >> 2), rr = lr >> 1;
        const int m = (int)m0 + rr;
        const int rem = m % HW;
        oy[h] = rem / W; ox[h] = rem - oy[h] * W;
        if (m >= M) oy[h] = -4;
        rowa[h] = xw_lane + (lr + 2 * halo) * 32;
      }
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int dy = tp / 3 - 1, dx = tp % 3 - 1;
        const int shift = (dy * W + dx) * 64;
        unsigned ad[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool ok = (unsigned)(oy[h] + dy) < (unsigned)H && (unsigned)(ox[h] + dx) < (unsigned)W;
          ad[h] = ok ? rowa[h] + shift : zero_lane;
        }
        u32x4 bf[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const u32x2 r1 = tr(ad[0] + ct * XSUB), r2 = tr(ad[1] + ct * XSUB);
          bf[ct] = u32x4{r1[0], r1[1], r2[0], r2[1]};
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          u32x4 br;
#pragma unroll
          for (int e = 0; e < 4; ++e) br[e] = __builtin_amdgcn_alignbit(bf[ct][e], bf[ct][e], 16);
          acc[tp][0][ct] = mma_chunk<bf16>(af[0], br, acc[tp][0][ct]);
          acc[tp][1][ct] = mma_chunk<bf16>(af[1], br, acc[tp][1][ct]);
          acc[tp][0][ct] = mma_chunk<bf16>(af[0], bf[ct], acc[tp][0][ct]);
          acc[tp][1][ct] = mma_chunk<bf16>(af[1], bf[ct], acc[tp][1][ct]);
        }
      }
    }
  }

  const int job = (NC == 256 ? 0 : blockIdx.y * NW) + j, njobs = (NC == 256 ? 1 : gridDim.y) * NW;
  float* out = part + ((size_t)blockIdx.x * njobs + job) * JOB;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) out[(tp * 32 + nt * 16 + lq * 4 + e) * 32 + ct * 16 + i] = acc[tp][nt][ct][e];
}

// dW[o][ig][ky][kx] (PyTorch layout, overwrite) = sum over splits, in split order.  One thread per partial element in PARTIAL order (c fastest: the
// reads of a wave are contiguous 128-byte rows, split after split); the 4-byte scatter into the weight layout is 74 k .. 147 k stores per layer.
// grouped: o = 32 job + n, ig = c;  dense: job = (ig / 64) * 8 + (ig % 64 / 32) * 4 + o / 32
__global__ __launch_bounds__(256) void wgrad3x3_finalize_kernel(const float* __restrict__ part, float* __restrict__ dw, int O, int Ig, int grouped, int njobs,
                                                                int splits) {
  // a thread owns 4 consecutive c (one 16-byte load per split slab) and keeps four slabs in flight: the one-element-per-thread loop with its
  // dependent 4-byte loads ran at 0.3 TB/s over the 19 MB of slabs (60 us per layer)
  const int total = njobs * wg3::JOB;
  for (int i4 = blockIdx.x * 256 + threadIdx.x; i4 < total / 4; i4 += gridDim.x * 256) {
    const int idx = i4 * 4;
    const int c = idx & 31, n = (idx >> 5) & 31, tp = (idx >> 10) % 9, job = idx / wg3::JOB;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int sp = 0;
    for (; sp + 4 <= splits; sp += 4) {
      s0 += *reinterpret_cast<const f32x4*>(part + (size_t)sp * total + idx);
      s1 += *reinterpret_cast<const f32x4*>(part + (size_t)(sp + 1) * total + idx);
      s2 += *reinterpret_cast<const f32x4*>(part + (size_t)(sp + 2) * total + idx);
      s3 += *reinterpret_cast<const f32x4*>(part + (size_t)(sp + 3) * total + idx);
    }
    for (; sp < splits; ++sp) s0 += *reinterpret_cast<const f32x4*>(part + (size_t)sp * total + idx);
    const f32x4 s = (s0 + s1) + (s2 + s3);
    const int o = grouped ? job * 32 + n : (job & 3) * 32 + n;
    const int ig = grouped ? c : (job >> 3) * 64 + ((job >> 2) & 1) * 32 + c;
#pragma unroll
    for (int e = 0; e < 4; ++e) dw[((size_t)o * Ig + ig + e) * 9 + tp] = s[e];
  }
}

// Supported: 3x3 / stride 1 / pad 1, 16-bit storage; grouped with 32 -> 32 channels per group and 8 groups, or dense with 128 output
// channels and 64 / 128 input channels.
bool wgrad3x3_supported(int dtype, int O, int Ig, int groups, int W) {
  constexpr bool off = false;
  if (off || (dtype != 1 && dtype != 2)) return false;         // dtype 2: fp32 rows, two-limb arithmetic (wgrad3x3_x2_kernel)
  if (groups == 8) return O == 256 && Ig == 32 && W <= 20;
  return groups == 1 && O == 128 && (Ig == 64 || Ig == 128) && W <= 40;
}
static int wgrad3x3_rows(int groups, int dtype) { return dtype == 2 && groups == 8 ? 32 : wg3::CH; }      // rows per stage
static int wgrad3x3_plan(int O, int Ig, int groups, int M, int* splits, int* cpw, int* njobs, int dtype) {
  const int ch = wgrad3x3_rows(groups, dtype);
  const int n_chunks = (M + ch - 1) / ch;
  const int gy = groups == 8 ? 1 : Ig / 64;
  int s = 256 / gy;                     // one workgroup per CU: fewer partials to write and sum
  if (s > n_chunks) s = n_chunks;
  *cpw = (n_chunks + s - 1) / s;
  *splits = (n_chunks + *cpw - 1) / *cpw;
  *njobs = gy * wg3::NW;
  return n_chunks;
}
size_t wgrad3x3_scratch_bytes(int O, int Ig, int groups, int M, int dtype) {
  int splits, cpw, njobs;
  wgrad3x3_plan(O, Ig, groups, M, &splits, &cpw, &njobs, dtype);
  return (size_t)splits * njobs * wg3::JOB * sizeof(float);
}
// defer != nullptr: the split slabs are left in `scratch` for a later batched finalize (train_kernels.hip wgrad_finalize_multi, kind 3) and
// defer[0..1] = {njobs, splits}
int launch_wgrad3x3(const void* x, int xld, const void* dz, int zld, float* dw, float* scratch, int B, int H, int W, int O, int Ig, int groups, hipStream_t s,
                    int* defer, int dtype) {
  const int M = B * H * W;
  int splits, cpw, njobs;
  const int n_chunks = wgrad3x3_plan(O, Ig, groups, M, &splits, &cpw, &njobs, dtype);
  if (dtype == 2 && groups == 8) {
    constexpr int CHX = 32, WINP = CHX + 2 * 21 + 3, lds = (256 / 16) * 2 * CHX * 32 + (256 / 16) * 2 * WINP * 32 + 2 * WINP * 32 + 32;
    auto kern = wgrad3x3_x2_kernel<256, 256, 20, CHX>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(splits), dim3(512), lds, s, (const float*)x, xld, (const float*)dz, zld, scratch, M, H, W, n_chunks, cpw);
  } else if (dtype == 2) {
    constexpr int CHX = 64, WINP = CHX + 2 * 41 + 3, lds = (128 / 16) * 2 * CHX * 32 + (64 / 16) * 2 * WINP * 32 + 2 * WINP * 32 + 32;
    auto kern = wgrad3x3_x2_kernel<128, 64, 40, CHX>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(splits, Ig / 64), dim3(512), lds, s, (const float*)x, xld, (const float*)dz, zld, scratch, M, H, W, n_chunks, cpw);
  } else if (groups == 8) {
    constexpr int WINP = wg3::CH + 2 * 21 + 3, lds = (256 / 16) * wg3::CH * 32 + (256 / 16) * WINP * 32 + WINP * 32 + 32;
    {    // per launch: the attribute is per DEVICE (a process-wide "done" flag skipped it on a second GPU), and the call is cheap
      hipError_t e = hipFuncSetAttribute((const void*)wgrad3x3_kernel<256, 256, 20>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((wgrad3x3_kernel<256, 256, 20>), dim3(splits), dim3(512), lds, s, (const bf16*)x, xld, (const bf16*)dz, zld, scratch, M, H, W, n_chunks, cpw);
  } else {
    constexpr int WINP = wg3::CH + 2 * 41 + 3, lds = (128 / 16) * wg3::CH * 32 + (64 / 16) * WINP * 32 + WINP * 32 + 32;
    hipLaunchKernelGGL((wgrad3x3_kernel<128, 64, 40>), dim3(splits, Ig / 64), dim3(512), lds, s, (const bf16*)x, xld, (const bf16*)dz, zld, scratch, M, H, W, n_chunks, cpw);
  }
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  if (defer) { defer[0] = njobs; defer[1] = splits; return 0; }
  const int total = njobs * wg3::JOB;
  hipLaunchKernelGGL(wgrad3x3_finalize_kernel, dim3((total / 4 + 255) / 256), dim3(256), 0, s, scratch, dw, O, Ig, groups == 8 ? 1 : 0, njobs, splits);
  return (int)hipGetLastError();
}

// ---- the grouped 3x3 / s1 / p1 convolution itself (stage-1 conv2 in the training step: forward, and the data gradient with the transposed +
// tap-flipped weights): 8 groups of 32 -> 32 channels.  As an implicit GEMM with N = 32 per group it ran on conv_gemm_v2's 128 x 32 tile at
// 220 TF/s (8 launches of 213 us per step).  Here wave g IS group g: its 18 weight fragments (9 taps x 2 channel tiles) stay in 72 VGPRs for the
// whole launch, the activations go through the same staged pixel window as in wgrad3x3 (planes of 8 channels, a tap = an address, invalid taps
// read a zero slot): per 16 pixels 9 fragment reads feed 18 MFMAs.
//   y[m][32 g + n] = sum_{tap, c} x[pix(m) + tap][32 g + c] * w[32 g + n][tap * 32 + c]        w = the trainer's packed layer ([256][Kw], K order (ky, kx, c))
// Training epilogues (conv_gemm.h): y2 != nullptr - y = GELU(conv), y2 = the GELU's derivative at the pre-activation (forward); mul != nullptr -
// y = conv * mul (data gradient times the saved derivative of the layer in front)
__global__ __launch_bounds__(512) void gconv3x3_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w, int Kw, bf16* __restrict__ y, int M, int H, int W,
                                                       int n_chunks, int chunks_per_wg, bf16* __restrict__ y2, const bf16* __restrict__ mul) {
  using namespace wg3;
  constexpr int C = 256, MAXW = 20, WINP = CH + 2 * (MAXW + 1) + 3;      // 109 pixels (odd: the staging stores walk the planes)
  constexpr int NPX = (WINP * (C / 8) + 511) / 512;
  __shared__ __attribute__((aligned(16))) unsigned char smem[(C / 8) * WINP * 16 + 16];     // [32 channel chunks][pixel][16 B] + a zero slot
  unsigned char* const ZERO = smem + (C / 8) * WINP * 16;
  const int t = threadIdx.x, lane = t & 63, lrow = lane & 15, lq = lane >> 4;
  const int g = __builtin_amdgcn_readfirstlane(t >> 6);
  const int halo = W + 1, winp = CH + 2 * halo;
  const int q0 = blockIdx.x * chunks_per_wg;
  int q1 = q0 + chunks_per_wg; q1 = q1 < n_chunks ? q1 : n_chunks;
  if (t < 4) reinterpret_cast<unsigned*>(ZERO)[t] = 0u;

  // this wave's weights: A fragments (rows = output channels 16 nt + lrow of the group, k = 8 lq .. of the tap's 32 input channels)
  u32x4 wf[9][2];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) wf[tp][nt] = *reinterpret_cast<const u32x4*>(w + (size_t)(g * 32 + nt * 16 + lrow) * Kw + tp * 32 + lq * 8);

  u32x4 px[NPX];
  auto gload = [&](int q) {
    const long m0 = (long)q * CH;
#pragma unroll
    for (int u0 = 0; u0 < NPX; ++u0) {
      const int u = t + 512 * u0, p = u / (C / 8), c8 = u % (C / 8);
      const long m = m0 - halo + p;
      px[u0] = (p < winp && m >= 0 && m < M) ? *reinterpret_cast<const u32x4*>(x + (size_t)m * C + c8 * 8) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int u0 = 0; u0 < NPX; ++u0) {
      const int u = t + 512 * u0, p = u / (C / 8), c8 = u % (C / 8);
      if (p < WINP) *reinterpret_cast<u32x4*>(smem + c8 * (WINP * 16) + p * 16) = px[u0];
    }
  };
  const unsigned char* const plane = smem + (g * 4 + lq) * (WINP * 16);     // the 8 channels this lane feeds as k = 8 lq ..
  const int HW = H * W;

  if (q0 < q1) gload(q0);
  for (int q = q0; q < q1; ++q) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (q + 1 < q1) gload(q + 1);
    const int m0 = q * CH;
#pragma unroll
    for (int mt = 0; mt < CH / 16; ++mt) {
      const int m = m0 + mt * 16 + lrow;                              // this lane's pixel (B operand column)
      const int rem = m % HW, oy = m < M ? rem / W : -4, ox = rem - (rem / W) * W;
      const unsigned char* const base = plane + (mt * 16 + lrow + halo) * 16;
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int dy = tp / 3 - 1, dx = tp % 3 - 1;
        const bool ok = (unsigned)(oy + dy) < (unsigned)H && (unsigned)(ox + dx) < (unsigned)W;
        const u32x4 xf = *reinterpret_cast<const u32x4*>(ok ? base + (dy * W + dx) * 16 : ZERO);
        acc[0] = mma_chunk<bf16>(wf[tp][0], xf, acc[0]);
        acc[1] = mma_chunk<bf16>(wf[tp][1], xf, acc[1]);
      }
      if (m < M) {                                                    // lane holds channels 32 g + 16 nt + 4 lq .. + 3 of pixel m
        const size_t o0 = (size_t)m * C + g * 32 + lq * 4;
        if (y2) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            f32x4 dv;
#pragma unroll
            for (int e = 0; e < 4; ++e) { float d; acc[nt][e] = gelu_sig_d(acc[nt][e], d); dv[e] = d; }
            store4<bf16>(y2 + o0 + 16 * nt, dv);
          }
        } else if (mul) {
          acc[0] = acc[0] * load4<bf16>(mul + o0);
          acc[1] = acc[1] * load4<bf16>(mul + o0 + 16);
        }
        store4<bf16>(y + o0, acc[0]);
        store4<bf16>(y + o0 + 16, acc[1]);
      }
    }
  }
}

bool gconv3x3_supported(int dtype, int O, int Ig, int groups, int KH, int KW, int stride, int pad, int W) {
  constexpr bool off = false;
  return !off && dtype == 1 && O == 256 && Ig == 32 && groups == 8 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && W <= 20;
}
int launch_gconv3x3(const void* x, const void* w_packed, int Kw, void* y, int B, int H, int W, hipStream_t s, void* y2, const void* mul) {
  const int M = B * H * W, n_chunks = (M + wg3::CH - 1) / wg3::CH;
  int wgs = n_chunks < 512 ? n_chunks : 512;
  const int cpw = (n_chunks + wgs - 1) / wgs;
  wgs = (n_chunks + cpw - 1) / cpw;
  hipLaunchKernelGGL(gconv3x3_kernel, dim3(wgs), dim3(512), 0, s, (const bf16*)x, (const bf16*)w_packed, Kw, (bf16*)y, M, H, W, n_chunks, cpw,
                     (bf16*)y2, (const bf16*)mul);
  return (int)hipGetLastError();
}

// ---- the same grouped conv in the two-limb numerics modes (`bf16x2` / `f16x2`, conv_gemm_v2.hip): fp32 activations in and out, weights as (hi, lo)
// limb words [256][Kw].  On conv_gemm_v2's 128 x 32 two-limb tile this layer was the slowest of the mode (7.2 ms per block at 12800 images, 105
// TFLOP/s: 16 MFMAs per staged K slice and wave).  Wave g = group g again: its 36 weight fragments (9 taps x 2 k-chunks of 16 channels x 2 channel
// tiles) stay in 144 VGPRs.  The pixels live in a RING of 128 slots per 4-channel plane ([64 planes][128 pixels][16 B] = 128 KB of LDS, slot =
// linear pixel index & 127), split into limb words ONCE while they are staged: consecutive 64-pixel chunks share 42 pixels of their windows, so a
// chunk brings in only its 64 new pixels (8 x 16 B per thread in flight under the previous chunk's MFMAs - with the whole 106-pixel window in
// registers the kernel spilled and the prefetch serialised).  A fragment read delivers 8 k-slots = 4 channels x (lo, hi), its half-swapped copy feeds
// the cross terms: 18 reads + 72 MFMAs per 16 pixels.
__global__ __launch_bounds__(512, 1) void gconv3x3_x2_kernel(const float* __restrict__ x, const unsigned* __restrict__ w, int Kw, float* __restrict__ y, int M, int H,
                                                             int W, int n_chunks, int chunks_per_wg, int act) {
  using namespace wg3;
  constexpr int C = 256, NPL = C / 4, RING = 128, HALO = 21;              // 64 planes of 4 channels; the window of a chunk = its 64 pixels +- 21
  constexpr int NPX = CH * NPL / 512;                                      // 8: 16-byte units per thread and batch of 64 pixels
  constexpr int PLANE = RING * 16;                                         // plane pitch 0 mod 256 B: ds_read_b128's 16-lane groups mix the rows {0-3, 12-15} of one
                                                                           // plane (lq) with {4-11} of the next - one pad slot per plane read 2-way
  static_assert(CH == 64 && CH + 2 * HALO + (CH - 2 * HALO) <= RING, "ring geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];    // [64 planes][RING][16 B] + a zero slot
  unsigned char* const ZERO = smem + NPL * PLANE;
  const int t = threadIdx.x, lane = t & 63, lrow = lane & 15, lq = lane >> 4;
  const int g = __builtin_amdgcn_readfirstlane(t >> 6);
  const int q0 = blockIdx.x * chunks_per_wg;
  int q1 = q0 + chunks_per_wg; q1 = q1 < n_chunks ? q1 : n_chunks;
  if (t < 4) reinterpret_cast<unsigned*>(ZERO)[t] = 0u;

  // this wave's weights: rows = output channels 16 nt + lrow of the group, limb words of channels 16 kc + 4 lq .. + 3 of tap tp
  u32x4 wf[9][2][2];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) wf[tp][kc][nt] = *reinterpret_cast<const u32x4*>(w + (size_t)(g * 32 + nt * 16 + lrow) * Kw + tp * 32 + kc * 16 + lq * 4);

  // a batch = 64 consecutive pixels starting at linear index P0 (may start before 0 / end past M: those are stored as zeros).  The loads are
  // UNCONDITIONAL on a clamped pixel: a per-lane `ok ? load : 0` compiles to an exec-masked branch with its own vmcnt(0) per load.
  u32x4 px[NPX];
  unsigned pxok = 0;
  auto gload = [&](long P0) {
    pxok = 0;
#pragma unroll
    for (int u0 = 0; u0 < NPX; ++u0) {
      const int u = t + 512 * u0, p = (u & 7) | ((u >> 9) << 3), c4 = (u >> 3) & (NPL - 1);      // 8 consecutive lanes = 8 consecutive pixels of one plane
      const long m = P0 + p;                                                                       // (their 16-byte stores fill 128 contiguous LDS bytes)
      const bool ok = m >= 0 && m < M;
      px[u0] = *reinterpret_cast<const u32x4*>(x + (size_t)(ok ? m : 0) * C + c4 * 4);
      pxok |= ok ? (1u << u0) : 0u;
    }
  };
  auto lstore = [&](long P0) {
#pragma unroll
    for (int u0 = 0; u0 < NPX; ++u0) {
      const int u = t + 512 * u0, p = (u & 7) | ((u >> 9) << 3), c4 = (u >> 3) & (NPL - 1);
      u32x4 xs, xr;
      x2_split(px[u0], xs, xr);
      if (!((pxok >> u0) & 1u)) xs = u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(smem + c4 * PLANE + (int)((P0 + p) & (RING - 1)) * 16) = xs;
    }
  };
  const unsigned char* const plane = smem + (g * 8 + lq) * PLANE;           // k-chunk kc adds 4 planes
  const int HW = H * W;

  if (q0 < q1) {                           // the first window of this workgroup: pixels [64 q0 - 21, 64 q0 + 107) = the whole ring, in two batches
    gload((long)q0 * CH - HALO);
    lstore((long)q0 * CH - HALO);
    gload((long)q0 * CH - HALO + CH);
  }
  for (int q = q0; q < q1; ++q) {
    const long Pn = (long)q * CH - HALO + CH + (q == q0 ? 0 : CH - 2 * HALO);   // first pixel of the batch in px: second half of the first window, else the 64 new pixels
    __syncthreads();                       // every wave is done with the previous chunk: its oldest 64 slots may be overwritten
    lstore(q == q0 ? Pn : (long)q * CH + HALO + (CH - 2 * HALO));
    __syncthreads();
    if (q + 1 < q1) gload((long)(q + 1) * CH + HALO + (CH - 2 * HALO));          // the 64 pixels the next window adds: [64 (q+1) + 43, 64 (q+1) + 107)
    const int m0 = q * CH;
    // (oy, ox) of this lane's pixel: divided out once per chunk, then stepped by 16 pixels per tile (the three integer divisions per tile
    // cost about as many VALU issues as the tile's 72 half-swaps)
    int oyc, oxc;
    {
      const int mm = m0 + lrow, rem = mm % HW;
      oyc = rem / W;
      oxc = rem - oyc * W;
    }
#pragma unroll 2
    for (int mt = 0; mt < CH / 16; ++mt) {
      const int m = m0 + mt * 16 + lrow;                              // this lane's pixel (B operand column)
      const int oy = m < M ? oyc : -4, ox = oxc;
      oxc += 16;
      while (oxc >= W) { oxc -= W; ++oyc; }
      if (oyc >= H) oyc -= H;                                         // (H * W >= 16: at most one image boundary per step)
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int dy = tp / 3 - 1, dx = tp % 3 - 1;
        const bool ok = (unsigned)(oy + dy) < (unsigned)H && (unsigned)(ox + dx) < (unsigned)W;
        const unsigned char* const src = ok ? plane + ((m + dy * W + dx) & (RING - 1)) * 16 : ZERO;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
          const u32x4 xf = *reinterpret_cast<const u32x4*>(ok ? src + kc * (4 * PLANE) : ZERO);
          u32x4 xr;
#pragma unroll
          for (int e = 0; e < 4; ++e) xr[e] = __builtin_amdgcn_alignbit(xf[e], xf[e], 16);
          acc[0] = mma_chunk<bf16>(wf[tp][kc][0], xr, acc[0]);          // lo x hi + hi x lo   (the two accumulators alternate: no back-to-back dependence)
          acc[1] = mma_chunk<bf16>(wf[tp][kc][1], xr, acc[1]);
          acc[0] = mma_chunk<bf16>(wf[tp][kc][0], xf, acc[0]);          // hi x hi + lo x lo
          acc[1] = mma_chunk<bf16>(wf[tp][kc][1], xf, acc[1]);
        }
      }
      if (m < M) {                                                    // lane holds channels 32 g + 16 nt + 4 lq .. + 3 of pixel m
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          if (act == ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[nt][e] = gelu_erfc(acc[nt][e]);
          }
          *reinterpret_cast<f32x4*>(y + (size_t)m * C + g * 32 + nt * 16 + lq * 4) = acc[nt];
        }
      }
    }
  }
}

// p: a conv_gemm launch in the two-limb modes (dtype 2) that this kernel covers
bool gconv3x3_x2_eligible(const ConvGemmParams& p, int dtype) {
  if (dtype != 2) return false;
  if (p.groups != 8 || p.N != 32 || p.Cin != 32 || p.x_cstride != 256 || p.y_cstride != 256 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1) return false;
  if (p.W > 20 || p.Kw < 288 || (p.Kw & 3) || p.bias || p.res || p.y2 || p.pos || p.x2 || p.K2 || p.pool2 || p.y_rpi || p.out_f32 || p.w_rstride || p.w_gstride) return false;
  return (p.act == ACT_NONE || p.act == ACT_GELU) && (long)p.B * p.H * p.W < (1L << 31) && p.H * p.W >= 16;
}
int launch_gconv3x3_x2(const ConvGemmParams& p, hipStream_t s) {
  const int M = p.B * p.H * p.W, n_chunks = (M + wg3::CH - 1) / wg3::CH;
  int wgs = n_chunks < 256 ? n_chunks : 256;              // one 8-wave workgroup per CU (128 KB of LDS, 2 waves per SIMD with 256 VGPRs)
  const int cpw = (n_chunks + wgs - 1) / wgs;
  wgs = (n_chunks + cpw - 1) / cpw;
  const int lds = 64 * 128 * 16 + 16;                  // the pixel ring + the zero slot
  {    // per launch: the attribute is per DEVICE (a process-wide "done" flag skipped it on a second GPU), and the call is cheap
    hipError_t e = hipFuncSetAttribute((const void*)gconv3x3_x2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(gconv3x3_x2_kernel, dim3(wgs), dim3(512), lds, s, (const float*)p.x, (const unsigned*)p.w, p.Kw, (float*)p.y, M, p.H, p.W, n_chunks, cpw, p.act);
  return (int)hipGetLastError();
}

// ---- 1x1 convolutions (conv1 / conv3 of the Mlps, qkv, proj):  Y[n][split * Kc_pad + c] = sum_{m in split} dz[m][n] * x[m][c]
// The same transposing-read scheme without taps: a workgroup owns an NT (n) x CT (c) block (256 where the layer has >= 256 columns, else 128), its
// 8 waves (4 x 2) NT/4 x CT/2 each (up to 32 accumulator tiles); per 64 rows it stages (NT + CT) x 128 B and issues up to 64 MFMAs per wave.  Output is the split-K partial layout wgrad_finalize_kernel already sums (it also
// undoes the head-dim padding of qkv rows / proj columns).
template <int NT, int CT>
__global__ __launch_bounds__(512) void wgrad1x1_kernel(const bf16* __restrict__ x, int xld, int C, const bf16* __restrict__ dz, int zld, int N,
                                                       float* __restrict__ y, int M, int n_chunks, int chunks_per_wg, int splits, int Kc_pad) {
  using namespace wg3;
  constexpr int WN = NT / 4, WC = CT / 2;                          // a wave's block: 4 x 2 waves
  constexpr int TN = WN / 16, TC = WC / 16;
  // subtile pitch 2048 + 32 bytes: the staging stores of a lane group hit 4 .. 8 DIFFERENT 16-column subtiles at the same row - with a pitch
  // of 0 mod 256 bytes they all landed on the same banks (SQ_LDS_BANK_CONFLICT 54 % of this kernel's LDS cycles, profiles/r03_train_mfma_pmc.txt)
  constexpr int SUB = CH * 32 + 32;
  constexpr int ZT_BYTES = (NT / 16) * SUB, XT_BYTES = (CT / 16) * SUB;
  constexpr int NPZ = (CH * NT / 8) / 512, NPX = (CH * CT / 8) / 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const ZT = smem;
  unsigned char* const XT = smem + ZT_BYTES;
  const int t = threadIdx.x, lane = t & 63, i = lane & 15, lq = lane >> 4;
  const int j = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = j & 3, wc = j >> 2;
  const int n0 = blockIdx.y * NT, c0 = blockIdx.z * CT;
  const int q0 = blockIdx.x * chunks_per_wg;
  int q1 = q0 + chunks_per_wg; q1 = q1 < n_chunks ? q1 : n_chunks;

  u32x4 pz[NPZ], px[NPX];
  auto gload = [&](int q) {
    const long m0 = (long)q * CH;
#pragma unroll
    for (int u0 = 0; u0 < NPZ; ++u0) {
      const int u = t + 512 * u0, r = u / (NT / 8), c8 = u % (NT / 8);
      const long m = m0 + r;
      pz[u0] = (m < M && n0 + c8 * 8 < N) ? *reinterpret_cast<const u32x4*>(dz + (size_t)m * zld + n0 + c8 * 8) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u0 = 0; u0 < NPX; ++u0) {
      const int u = t + 512 * u0, r = u / (CT / 8), c8 = u % (CT / 8);
      const long m = m0 + r;
      px[u0] = (m < M && c0 + c8 * 8 < C) ? *reinterpret_cast<const u32x4*>(x + (size_t)m * xld + c0 + c8 * 8) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int u0 = 0; u0 < NPZ; ++u0) {
      const int u = t + 512 * u0, r = u / (NT / 8), c8 = u % (NT / 8);
      *reinterpret_cast<u32x4*>(ZT + (c8 >> 1) * SUB + r * 32 + (c8 & 1) * 16) = pz[u0];
    }
#pragma unroll
    for (int u0 = 0; u0 < NPX; ++u0) {
      const int u = t + 512 * u0, r = u / (CT / 8), c8 = u % (CT / 8);
      *reinterpret_cast<u32x4*>(XT + (c8 >> 1) * SUB + r * 32 + (c8 & 1) * 16) = px[u0];
    }
  };
  f32x4 acc[TN][TC];
#pragma unroll
  for (int a = 0; a < TN * TC; ++a) acc[a / TC][a % TC] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned lane_off = (lq * 4 + (i >> 2)) * 32 + (i & 3) * 8;
  const unsigned zt_lane = (unsigned)(size_t)(__attribute__((address_space(3))) void*)ZT + (wn * TN) * SUB + lane_off;
  const unsigned xt_lane = (unsigned)(size_t)(__attribute__((address_space(3))) void*)XT + (wc * TC) * SUB + lane_off;
  auto tr = [&](unsigned addr) -> u32x2 {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(size_t)addr));
  };

  if (q0 < q1) gload(q0);
  for (int q = q0; q < q1; ++q) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (q + 1 < q1) gload(q + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 af[TN];
#pragma unroll
      for (int k4 = 0; k4 < TN; ++k4) {
        const u32x2 a1 = tr(zt_lane + k4 * SUB + ks * 1024), a2 = tr(zt_lane + k4 * SUB + ks * 1024 + 512);
        af[k4] = u32x4{a1[0], a1[1], a2[0], a2[1]};
      }
#pragma unroll
      for (int ct = 0; ct < TC; ++ct) {
        const u32x2 b1 = tr(xt_lane + ct * SUB + ks * 1024), b2 = tr(xt_lane + ct * SUB + ks * 1024 + 512);
        const u32x4 bf = {b1[0], b1[1], b2[0], b2[1]};
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) acc[nt][ct] = mma_chunk<bf16>(af[nt], bf, acc[nt][ct]);
      }
    }
  }
  // partials: lane holds n = n0 + WN wn + 16 nt + 4 lq + e, c = c0 + WC wc + 16 ct + i
  const size_t ldy = (size_t)splits * Kc_pad;
#pragma unroll
  for (int nt = 0; nt < TN; ++nt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + wn * WN + nt * 16 + lq * 4 + e;
      if (n < N) {
#pragma unroll
        for (int ct = 0; ct < TC; ++ct) {
          const int c = c0 + wc * WC + ct * 16 + i;
          if (c < C) y[(size_t)n * ldy + (size_t)blockIdx.x * Kc_pad + c] = acc[nt][ct][e];
        }
      }
    }
}

// ---- the same for the two-limb trainers (`bf16x2` / `f16x2`: fp32 storage).  x and dz are fp32 rows; the limb split happens on the way into LDS:
// row m of a chunk becomes TWO 16-bit rows, 2 m = lo(m, :) and 2 m + 1 = hi(m, :) (x2_split: hi = the leading bits, lo = the rounded remainder), so the
// transposing read hands every lane 32-bit words (lo, hi) of ONE value - the k-slot order of conv_gemm_v2's two-limb arithmetic - and a 16-value k step is
// two MFMAs: (a, b) sums lo lo + hi hi, (a, b with its halves swapped) the cross terms.  No transposed / limb-packed copies of dz and x in HBM
// (round 3 route: transpose_cols + im2col_t + a limb GEMM, 12 ms of the 57 ms step in transposes alone).
template <int NT, int CT>
__global__ __launch_bounds__(512) void wgrad1x1_x2_kernel(const float* __restrict__ x, int xld, int C, const float* __restrict__ dz, int zld, int N,
                                                          float* __restrict__ y, int M, int n_chunks, int chunks_per_wg, int splits, int Kc_pad) {
  using namespace wg3;
  constexpr int WN = NT / 4, WC = CT / 2;
  constexpr int TN = WN / 16, TC = WC / 16;
  constexpr int SUB = 2 * CH * 32 + 32;                            // a 16-column subtile of 2 CH limb rows (+ 32: see wgrad1x1_kernel)
  constexpr int ZT_BYTES = (NT / 16) * SUB;
  constexpr int NPZ = (CH * NT / 8) / 512, NPX = (CH * CT / 8) / 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const ZT = smem;
  unsigned char* const XT = smem + ZT_BYTES;
  const int t = threadIdx.x, lane = t & 63, i = lane & 15, lq = lane >> 4;
  const int j = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = j & 3, wc = j >> 2;
  const int n0 = blockIdx.y * NT, c0 = blockIdx.z * CT;
  const int q0 = blockIdx.x * chunks_per_wg;
  int q1 = q0 + chunks_per_wg; q1 = q1 < n_chunks ? q1 : n_chunks;

  u32x4 pz[NPZ][2], px[NPX][2];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto gload = [&](int q) {
    const long m0 = (long)q * CH;
#pragma unroll
    for (int u0 = 0; u0 < NPZ; ++u0) {
      const int u = t + 512 * u0, r = u / (NT / 8), c8 = u % (NT / 8);
      const long m = m0 + r;
      const bool ok = m < M && n0 + c8 * 8 < N;
      const float* src = dz + (size_t)(ok ? m : 0) * zld + (ok ? n0 + c8 * 8 : 0);
      pz[u0][0] = ok ? *reinterpret_cast<const u32x4*>(src) : zero4;
      pz[u0][1] = ok ? *reinterpret_cast<const u32x4*>(src + 4) : zero4;
    }
#pragma unroll
    for (int u0 = 0; u0 < NPX; ++u0) {
      const int u = t + 512 * u0, r = u / (CT / 8), c8 = u % (CT / 8);
      const long m = m0 + r;
      const bool ok = m < M && c0 + c8 * 8 < C;
      const float* src = x + (size_t)(ok ? m : 0) * xld + (ok ? c0 + c8 * 8 : 0);
      px[u0][0] = ok ? *reinterpret_cast<const u32x4*>(src) : zero4;
      px[u0][1] = ok ? *reinterpret_cast<const u32x4*>(src + 4) : zero4;
    }
  };
  // 8 fp32 values -> their 8 lo limbs and 8 hi limbs, packed in column order
  auto limbs8 = [&](const u32x4 a, const u32x4 b, u32x4& lo, u32x4& hi) {
    u32x4 sa, sb, r_;
    x2_split(a, sa, r_);
    x2_split(b, sb, r_);
    lo = u32x4{__builtin_amdgcn_perm(sa[1], sa[0], 0x05040100u), __builtin_amdgcn_perm(sa[3], sa[2], 0x05040100u),
               __builtin_amdgcn_perm(sb[1], sb[0], 0x05040100u), __builtin_amdgcn_perm(sb[3], sb[2], 0x05040100u)};
    hi = u32x4{__builtin_amdgcn_perm(sa[1], sa[0], 0x07060302u), __builtin_amdgcn_perm(sa[3], sa[2], 0x07060302u),
               __builtin_amdgcn_perm(sb[1], sb[0], 0x07060302u), __builtin_amdgcn_perm(sb[3], sb[2], 0x07060302u)};
  };
  auto lstore = [&]() {
#pragma unroll
    for (int u0 = 0; u0 < NPZ; ++u0) {
      const int u = t + 512 * u0, r = u / (NT / 8), c8 = u % (NT / 8);
      u32x4 lo, hi;
      limbs8(pz[u0][0], pz[u0][1], lo, hi);
      unsigned char* d = ZT + (c8 >> 1) * SUB + (2 * r) * 32 + (c8 & 1) * 16;
      *reinterpret_cast<u32x4*>(d) = lo;
      *reinterpret_cast<u32x4*>(d + 32) = hi;
    }
#pragma unroll
    for (int u0 = 0; u0 < NPX; ++u0) {
      const int u = t + 512 * u0, r = u / (CT / 8), c8 = u % (CT / 8);
      u32x4 lo, hi;
      limbs8(px[u0][0], px[u0][1], lo, hi);
      unsigned char* d = XT + (c8 >> 1) * SUB + (2 * r) * 32 + (c8 & 1) * 16;
      *reinterpret_cast<u32x4*>(d) = lo;
      *reinterpret_cast<u32x4*>(d + 32) = hi;
    }
  };
  f32x4 acc[TN][TC];
#pragma unroll
  for (int a = 0; a < TN * TC; ++a) acc[a / TC][a % TC] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned lane_off = (lq * 4 + (i >> 2)) * 32 + (i & 3) * 8;
  const unsigned zt_lane = (unsigned)(size_t)(__attribute__((address_space(3))) void*)ZT + (wn * TN) * SUB + lane_off;
  const unsigned xt_lane = (unsigned)(size_t)(__attribute__((address_space(3))) void*)XT + (wc * TC) * SUB + lane_off;
  auto tr = [&](unsigned addr) -> u32x2 {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(size_t)addr));
  };

  if (q0 < q1) gload(q0);
  for (int q = q0; q < q1; ++q) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (q + 1 < q1) gload(q + 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                               // 32 limb rows = 16 fp32 rows per step
      u32x4 af[TN];
#pragma unroll
      for (int k4 = 0; k4 < TN; ++k4) {
        const u32x2 a1 = tr(zt_lane + k4 * SUB + ks * 1024), a2 = tr(zt_lane + k4 * SUB + ks * 1024 + 512);
        af[k4] = u32x4{a1[0], a1[1], a2[0], a2[1]};
      }
#pragma unroll
      for (int ct = 0; ct < TC; ++ct) {
        const u32x2 b1 = tr(xt_lane + ct * SUB + ks * 1024), b2 = tr(xt_lane + ct * SUB + ks * 1024 + 512);
        const u32x4 bf = {b1[0], b1[1], b2[0], b2[1]};
        u32x4 br;
#pragma unroll
        for (int e = 0; e < 4; ++e) br[e] = __builtin_amdgcn_alignbit(bf[e], bf[e], 16);
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          acc[nt][ct] = mma_chunk<bf16>(af[nt], br, acc[nt][ct]);      // the cross terms first, hi hi + lo lo last (as conv_gemm_v2)
          acc[nt][ct] = mma_chunk<bf16>(af[nt], bf, acc[nt][ct]);
        }
      }
    }
  }
  const size_t ldy = (size_t)splits * Kc_pad;
#pragma unroll
  for (int nt = 0; nt < TN; ++nt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + wn * WN + nt * 16 + lq * 4 + e;
      if (n < N) {
#pragma unroll
        for (int ct = 0; ct < TC; ++ct) {
          const int c = c0 + wc * WC + ct * 16 + i;
          if (c < C) y[(size_t)n * ldy + (size_t)blockIdx.x * Kc_pad + c] = acc[nt][ct][e];
        }
      }
    }
}

bool wgrad1x1_supported(int dtype, int N, int C) {
  constexpr bool off = false;
  return !off && (dtype == 1 || dtype == 2) && (N % 8) == 0 && (C % 8) == 0;      // dtype 2: the two-limb kernel on fp32 rows
}
// block shape per layer: 256 along a dimension that has >= 256 columns, else 128 (wave blocks 64 / 32 rows x 128 / 64 columns)
// (two-limb kernel, dtype 2: 256 x 128 at most - the fp32 staging registers of a 256 x 256 block spill)
static void wgrad1x1_tile(int N, int C, int* NT, int* CT, int dtype) { *NT = N >= 256 ? 256 : 128; *CT = (C >= 256 && !(dtype == 2 && *NT == 256)) ? 256 : 128; }
// number of row splits (= partial slabs of Y) the launch will use
int wgrad1x1_splits(int N, int C, int M, int dtype) {
  int NT, CT;
  wgrad1x1_tile(N, C, &NT, &CT, dtype);
  const int n_chunks = (M + wg3::CH - 1) / wg3::CH;
  const int tiles = ((N + NT - 1) / NT) * ((C + CT - 1) / CT);
  static const int target = [] { const char* e = getenv("FSVIT_WGRAD_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 128; }();
  int s = (target + tiles - 1) / tiles;        // 128 workgroups per launch (two launches run side by side, train_engine.hip): the partial slabs (N x C x 4 B per split) are written and summed once each
  if (s > n_chunks) s = n_chunks;
  if (s < 1) s = 1;
  const int cpw = (n_chunks + s - 1) / s;
  return (n_chunks + cpw - 1) / cpw;
}
int launch_wgrad1x1(const void* x, int xld, int C, const void* dz, int zld, int N, float* y, int M, int Kc_pad, hipStream_t s, int dtype) {
  int NT, CT;
  wgrad1x1_tile(N, C, &NT, &CT, dtype);
  const int n_chunks = (M + wg3::CH - 1) / wg3::CH;
  const int splits = wgrad1x1_splits(N, C, M, dtype);
  const int cpw = (n_chunks + splits - 1) / splits;
  const dim3 grid(splits, (N + NT - 1) / NT, (C + CT - 1) / CT);
#define WG1_LAUNCH(A, B) do { const int lds = ((A) / 16 + (B) / 16) * (wg3::CH * 32 + 32);                                                                  \
    hipError_t e = hipFuncSetAttribute((const void*)wgrad1x1_kernel<A, B>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                            \
    if (e != hipSuccess) return (int)e;                                                                                                                  \
    hipLaunchKernelGGL((wgrad1x1_kernel<A, B>), grid, dim3(512), lds, s, (const bf16*)x, xld, C, (const bf16*)dz, zld, N, y, M, n_chunks, cpw, splits, Kc_pad); } while (0)
#define WG1X_LAUNCH(A, B) do { const int lds = ((A) / 16 + (B) / 16) * (2 * wg3::CH * 32 + 32);                                                             \
    hipError_t e = hipFuncSetAttribute((const void*)wgrad1x1_x2_kernel<A, B>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                         \
    if (e != hipSuccess) return (int)e;                                                                                                                  \
    hipLaunchKernelGGL((wgrad1x1_x2_kernel<A, B>), grid, dim3(512), lds, s, (const float*)x, xld, C, (const float*)dz, zld, N, y, M, n_chunks, cpw, splits, Kc_pad); } while (0)
  if (dtype == 2) {
    if (NT == 256) WG1X_LAUNCH(256, 128);
    else if (CT == 256) WG1X_LAUNCH(128, 256);
    else WG1X_LAUNCH(128, 128);
    return (int)hipGetLastError();
  }
  if (NT == 256 && CT == 256) WG1_LAUNCH(256, 256);
  else if (NT == 256) WG1_LAUNCH(256, 128);
  else if (CT == 256) WG1_LAUNCH(128, 256);
  else WG1_LAUNCH(128, 128);
#undef WG1_LAUNCH
  return (int)hipGetLastError();
}

}  // namespace FSVIT_NS
