// Fused Visformer stage-1 block (bf16):   y = x + conv3( GELU( conv2_g8_3x3( GELU( conv1( BN(x) ) ) ) ) )
// (test_phase/models/visformer.py:259-263 Block.forward with attn_disabled, Mlp :152-163; eval BN
// folded into conv1 by the weight packer).
//
// As three conv_gemm launches this block is bound by its intermediates, not by MFMA: the two 256-channel
// hidden maps cost 4 x 205 KB of HBM traffic per image and the grouped 3x3 (N = 32 per group) cannot
// fill a 128-wide tile (rocprof r01: 140-260 TF/s, 31 % of the step).  Here one 512-thread workgroup
// owns one half image (10 x 20 output tokens) and keeps everything on chip:
//   Xs   input tokens incl. one halo row, [224][128] bf16 in LDS (row-major, 16-B chunks XOR (token & 15))
//   per group g of 32 hidden channels (the grouped conv makes the groups independent until conv3):
//     P1  H1g = GELU(Xs . W1g^T + b1g)            240 tokens incl. halo  -> LDS, zero-bordered 12 x 22 pixels
//     P2  H2g = GELU(conv3x3(H1g, W2g))           9 taps = 9 K-chunks of 32  -> LDS
//     P3  acc += H2g . W3[:, g]^T                 wave w owns output channels 16w..16w+15 of all 13 m-tiles
//   y = acc + Xs (residual straight from LDS), written once.
// HBM traffic per image: 102 KB in + 102 KB out (vs ~1 MB unfused); halo recompute + tile padding cost 8 % MFMAs.
// H1g / H2g use a chunk-plane layout [k-chunk][pixel][16 B] so the b128 fragment reads of 16 consecutive
// pixels hit 16 distinct 16-byte slots (a 64-byte-per-pixel image cannot be made conflict-free by XOR).
// The three weight slices of a group (34 KB; 278 KB per block, L2-resident) are LDS-DMA'd one interval ahead.
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
constexpr int NW = 16;                    // waves per workgroup (1024 threads, one workgroup per CU)
constexpr int XT = 220, XTP = 224;        // input tokens held (11 rows), padded to 14 m-tiles
constexpr int OT = 200, OTP = 208;        // output tokens (10 rows), padded to 13 m-tiles
constexpr int PW = 36;                    // pitch of the zero-bordered H1 pixel grid (12 rows; columns -1 .. 20 used).  36 = 20 + 16: the 16 tokens of an
                                          // m-tile usually straddle a row end, and with the natural pitch 22 the tokens after the wrap land on the 16-byte
                                          // slots (mod 16) of the ones before it - 6.5 LDS cycles per ds_read_b128 instead of 4 (tools/lds_conflicts.py)
constexpr int H1_PLANE = 12 * PW * 16;    // 6912 = 27 * 256: the plane stride stays 0 mod 256 B
constexpr int H2_PLANE = OTP * 16;        // 3328
constexpr int OFF_H1 = XTP * 256;                  //  57344
constexpr int OFF_H2 = OFF_H1 + 4 * H1_PLANE;      //  84992
constexpr int OFF_W1 = OFF_H2 + 4 * H2_PLANE;      //  98304  [16 k-chunks][32 n][16 B]
constexpr int OFF_W2 = OFF_W1 + 16 * 32 * 16;      // 106496  [9 taps * 4 k-chunks][32 n][16 B]
constexpr int OFF_W3 = OFF_W2 + 36 * 32 * 16;      // 124928  [4 k-chunks][128 n][16 B]
constexpr int OFF_B1 = OFF_W3 + 4 * 128 * 16;      // 133120  conv1 folded bias, 256 fp32
constexpr int LDS_BYTES = OFF_B1 + HID * 4;        // 134144
constexpr int KW2 = 320;                  // packed conv2 row length (9*32 = 288 rounded up to the 64-element K slice)
}  // namespace s1

__device__ __forceinline__ void s1_dma16(const void* gsrc, unsigned lds_byte_addr) {
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
__device__ __forceinline__ void s1_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// v2 of the kernel: 16 waves; the three weight slices of a group (34 KB) are brought into LDS by LDS-DMA
// one barrier interval before use (single-buffered: each slice's previous consumer finished an interval
// earlier) in a [k-chunk][n][16 B] plane layout, so every fragment read in the kernel is a conflict-free
// ds_read_b128 of 16 consecutive 16-byte slots.  v1 (8 waves, per-wave weight fragments from global)
// spent its time in dependent L2 / LDS round trips: 70 us per workgroup for 7 us of MFMA work.
__global__ __launch_bounds__(1024) void stage1_block_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                            const bf16* __restrict__ w1, const float* __restrict__ b1,
                                                            const bf16* __restrict__ w2, const bf16* __restrict__ w3) {
  using namespace s1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Xs = smem;
  unsigned char* const H1 = smem + OFF_H1;
  unsigned char* const H2 = smem + OFF_H2;
  unsigned char* const W1b = smem + OFF_W1;
  unsigned char* const W2b = smem + OFF_W2;
  unsigned char* const W3b = smem + OFF_W3;
  float* const B1s = reinterpret_cast<float*>(smem + OFF_B1);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;

  const int t = threadIdx.x, lane = t & 63;
#ifdef S1_CLK
  long long ck0 = __builtin_readcyclecounter(), ckA = 0, ckB = 0, ckP = 0, ckl = 0;
#endif
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lrow = lane & 15, lq = lane >> 4;
  const int b = blockIdx.x >> 1, hsel = blockIdx.x & 1;
  const int xr0 = hsel ? 9 : 0;                       // first image row held in Xs
  const int r0 = hsel * 10;                           // first output row
  const bf16* xin = x + ((size_t)b * 400 + xr0 * W) * C1;

  // LDS-DMA of one weight slice: instruction i fills 16-byte slots [64 i, 64 i + 64); lane -> slot -> source
  auto dma_w1 = [&](int g) {
    for (int i = w; i < 8; i += NW) {
      const int sl = i * 64 + lane, ch = sl >> 5, n = sl & 31;
      s1_dma16(w1 + (size_t)(g * CG + n) * C1 + ch * 8, __builtin_amdgcn_readfirstlane(lds0 + OFF_W1 + i * 1024));
    }
  };
  auto dma_w2 = [&](int g) {
    for (int i = w; i < 18; i += NW) {
      const int sl = i * 64 + lane, qq = sl >> 5, n = sl & 31;
      s1_dma16(w2 + (size_t)(g * CG + n) * KW2 + (qq >> 2) * CG + (qq & 3) * 8, __builtin_amdgcn_readfirstlane(lds0 + OFF_W2 + i * 1024));
    }
  };
  auto dma_w3 = [&](int g) {
    for (int i = w; i < 8; i += NW) {
      const int sl = i * 64 + lane, ch = sl >> 7, n = sl & 127;
      s1_dma16(w3 + (size_t)n * HID + g * CG + ch * 8, __builtin_amdgcn_readfirstlane(lds0 + OFF_W3 + i * 1024));
    }
  };

  // ---- stage the input tokens + W1(0), zero H1 (its border must stay 0) and the Xs pad rows
  for (int grp = w; grp < XT / 4; grp += NW) {
    const int tk = grp * 4 + (lane >> 4);
    const int ch = (lane & 15) ^ (tk & 15);
    s1_dma16(xin + (size_t)tk * C1 + ch * 8, __builtin_amdgcn_readfirstlane(lds0 + grp * 1024));
  }
  dma_w1(0);
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = t; i < (4 * H1_PLANE) / 16; i += NW * 64) *reinterpret_cast<u32x4*>(H1 + i * 16) = z;
    if (t < (XTP - XT) * 16) *reinterpret_cast<u32x4*>(Xs + XT * 256 + t * 16) = z;
    if (t < HID) B1s[t] = b1[t];        // bias table: an in-loop global load would cost a full round trip + vmcnt(0) per tile
  }
  s1_dma_wait();
  __syncthreads();
#ifdef S1_CLK
  ckP = __builtin_readcyclecounter() - ck0; ckl = __builtin_readcyclecounter();
#endif

  // P3 ownership: output channels 16 (w & 7) .. +15, m-tiles (w >> 3), +2, ...  (7 tiles for the even half, 6 for the odd)
  const int n3 = w & 7, m3 = w >> 3;
  f32x4 acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto p3 = [&]() {
    const u32x4 wf = *reinterpret_cast<const u32x4*>(W3b + (lq * 128 + n3 * 16 + lrow) * 16);
    const unsigned char* hb = H2 + lq * H2_PLANE + (m3 * 16 + lrow) * 16;
    u32x4 af[7];
#pragma unroll
    for (int i = 0; i < 6; ++i) af[i] = *reinterpret_cast<const u32x4*>(hb + i * 512);
    af[6] = m3 == 0 ? *reinterpret_cast<const u32x4*>(hb + 6 * 512) : u32x4{0u, 0u, 0u, 0u};     // m-tile 12 exists only for the even half
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] = mma_chunk<bf16>(wf, af[i], acc[i]);
  };

  // P1 / P2 work split: (m-tile, n-tile) pairs p and p + 16; both pairs of a wave have the same n-tile (w & 1), so
  // they share the weight fragment and run as two independent MFMA chains
  const int nt = w & 1;
  const int mtA = w >> 1, mtB = (w + NW) >> 1;
  const bool p1B = w + NW < 2 * (XTP / 16);            // second conv1 pair exists  (28 pairs)
  const bool p2B = w + NW < 2 * (OTP / 16);            // second conv2 pair exists  (26 pairs)
  int hpA, hpB;                                        // H1 pixel (top-left tap) of this lane's output token, pairs A / B
  {
    int tk = mtA * 16 + lrow; tk = tk < OT ? tk : OT - 1;
    hpA = (tk / W) * PW + tk % W;
    tk = mtB * 16 + lrow; tk = tk < OT ? tk : OT - 1;  // padded output rows recompute token 199 (ignored later)
    hpB = (tk / W) * PW + tk % W;
  }
  auto h1_store = [&](int mt, f32x4 a, f32x4 bias) {
    const int tk = mt * 16 + lrow;
    if (tk < XT) {
      const int pr = tk / W, pc = tk - pr * W;
      const int pix = (pr + (hsel ? 0 : 1)) * PW + pc + 1;
      a += bias;
      const f32x2 g0 = S1_GELU2((f32x2{a[0], a[1]})), g1 = S1_GELU2((f32x2{a[2], a[3]}));
      const bf16x4 o = {(bf16)g0[0], (bf16)g0[1], (bf16)g1[0], (bf16)g1[1]};
      *reinterpret_cast<bf16x4*>(H1 + (nt * 2 + (lq >> 1)) * H1_PLANE + pix * 16 + (lq & 1) * 8) = o;
    }
  };
  auto h2_store = [&](int mt, f32x4 a) {
    const f32x2 g0 = S1_GELU2((f32x2{a[0], a[1]})), g1 = S1_GELU2((f32x2{a[2], a[3]}));
    const bf16x4 o = {(bf16)g0[0], (bf16)g0[1], (bf16)g1[0], (bf16)g1[1]};
    *reinterpret_cast<bf16x4*>(H2 + (nt * 2 + (lq >> 1)) * H2_PLANE + (mt * 16 + lrow) * 16 + (lq & 1) * 8) = o;
  };

#pragma unroll 1
  for (int g = 0; g < G; ++g) {
    // ---- interval A: DMA W2(g); P3(g-1); P1: H1g = GELU(conv1)
    dma_w2(g);
    if (g > 0) p3();
    {
      const f32x4 bias = *reinterpret_cast<const f32x4*>(B1s + g * CG + nt * 16 + lq * 4);
      const unsigned char* wr = W1b + (nt * 16 + lrow) * 16;
      const unsigned char* xa = Xs + (mtA * 16 + lrow) * 256;
      const unsigned char* xb = Xs + (mtB * 16 + lrow) * 256;
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      if (p1B) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          const int sw = (((kc * 4 + lq) ^ lrow) << 4);
          const u32x4 wf = *reinterpret_cast<const u32x4*>(wr + (kc * 4 + lq) * 512);
          const u32x4 x0 = *reinterpret_cast<const u32x4*>(xa + sw);
          const u32x4 x1 = *reinterpret_cast<const u32x4*>(xb + sw);
          a0 = mma_chunk<bf16>(wf, x0, a0);
          a1 = mma_chunk<bf16>(wf, x1, a1);
        }
        h1_store(mtA, a0, bias);
        h1_store(mtB, a1, bias);
      } else {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          const u32x4 wf = *reinterpret_cast<const u32x4*>(wr + (kc * 4 + lq) * 512);
          const u32x4 x0 = *reinterpret_cast<const u32x4*>(xa + (((kc * 4 + lq) ^ lrow) << 4));
          a0 = mma_chunk<bf16>(wf, x0, a0);
        }
        h1_store(mtA, a0, bias);
      }
    }
    s1_dma_wait();
    __syncthreads();
#ifdef S1_CLK
    { long long c = __builtin_readcyclecounter(); ckA += c - ckl; ckl = c; }
#endif
    // ---- interval B: DMA W1(g+1), W3(g); P2: H2g = GELU(grouped 3x3 conv of H1g)
    if (g + 1 < G) dma_w1(g + 1);
    dma_w3(g);
    {
      const unsigned char* wr = W2b + (lq * 32 + nt * 16 + lrow) * 16;
      const unsigned char* ha = H1 + lq * H1_PLANE + hpA * 16;
      const unsigned char* hb = H1 + lq * H1_PLANE + hpB * 16;
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      if (p2B) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int toff = ((tap / 3) * PW + tap % 3) * 16;
          const u32x4 wf = *reinterpret_cast<const u32x4*>(wr + tap * 2048);
          const u32x4 f0 = *reinterpret_cast<const u32x4*>(ha + toff);
          const u32x4 f1 = *reinterpret_cast<const u32x4*>(hb + toff);
          a0 = mma_chunk<bf16>(wf, f0, a0);
          a1 = mma_chunk<bf16>(wf, f1, a1);
        }
        h2_store(mtA, a0);
        h2_store(mtB, a1);
      } else {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const u32x4 wf = *reinterpret_cast<const u32x4*>(wr + tap * 2048);
          const u32x4 f0 = *reinterpret_cast<const u32x4*>(ha + ((tap / 3) * PW + tap % 3) * 16);
          a0 = mma_chunk<bf16>(wf, f0, a0);
        }
        h2_store(mtA, a0);
      }
    }
    s1_dma_wait();
    __syncthreads();
#ifdef S1_CLK
    { long long c = __builtin_readcyclecounter(); ckB += c - ckl; ckl = c; }
#endif
  }
  p3();

  // ---- y = acc + x (residual from LDS); lane holds channels 16 n3 + 4 lq .. +3 of token mt*16 + lrow
  bf16* yout = y + ((size_t)b * 400 + r0 * W) * C1 + n3 * 16 + lq * 4;
  const int xshift = hsel ? W : 0;                    // output token -> Xs token
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int mt = m3 + 2 * i;
    const int tk = mt * 16 + lrow;
    if (mt < 13 && tk < OT) {
      const int xt = tk + xshift;
      const bf16x4 r = *reinterpret_cast<const bf16x4*>(Xs + xt * 256 + (((n3 * 2 + (lq >> 1)) ^ (xt & 15)) << 4) + (lq & 1) * 8);
      f32x4 v = acc[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
      store4<bf16>(yout + (size_t)tk * C1, v);
    }
  }
#ifdef S1_CLK
  if (t == 0 && (blockIdx.x == 0 || blockIdx.x == 5001)) {
    const long long c = __builtin_readcyclecounter();
    printf("[stage1 wg %d] total %lld  prologue %lld  intervals A %lld  B %lld  tail+epilogue %lld\n", (int)blockIdx.x, c - ck0, ckP, ckA, ckB, c - ckl);
  }
#endif
}

bool stage1_fused_supported(int dtype, int C1, int hid, int group, int H1) {
  return dtype == 1 && C1 == s1::C1 && hid == s1::HID && group == s1::G && H1 == s1::W;
}

int launch_stage1_block(const void* x, void* y, const void* w1, const float* b1, const void* w2, const void* w3, int B, hipStream_t s) {
  if (B <= 0) return 0;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)stage1_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, s1::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(stage1_block_kernel, dim3(B * 2), dim3(s1::NW * 64), s1::LDS_BYTES, s, (const bf16*)x, (bf16*)y, (const bf16*)w1, b1,
                     (const bf16*)w2, (const bf16*)w3);
  return (int)hipGetLastError();
}

}  // namespace FSVIT_NS
