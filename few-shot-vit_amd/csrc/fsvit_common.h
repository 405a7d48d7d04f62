// Shared device/host helpers for the fsvit gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "conv_gemm.h"

// The 16-bit storage / MFMA operand type of this translation unit.  Every kernel source that handles 16-bit activations is compiled
// TWICE by the Makefile: into namespace fsvit with __bf16 (the `bf16` numerics mode) and, under -DFSVIT_HALF_F16, into namespace
// fsvit_f16 with _Float16 (the `f16` mode: same width, same MFMA rate, 3 more mantissa bits - 8 x smaller logit deviation on this
// network, whose bf16 error is dominated by the rounding of the WEIGHTS).  Inside a source the alias `bf16` means
// "this build's 16-bit type"; nothing else changes between the two builds except the MFMA opcode.
#ifdef FSVIT_HALF_F16
#define FSVIT_NS fsvit_f16
#define FSVIT_MFMA_16x16x32 "v_mfma_f32_16x16x32_f16"
#define FSVIT_MFMA_32x32x16 "v_mfma_f32_32x32x16_f16"
namespace fsvit_f16 { typedef _Float16 bf16; }
#else
#define FSVIT_NS fsvit
#define FSVIT_MFMA_16x16x32 "v_mfma_f32_16x16x32_bf16"
#define FSVIT_MFMA_32x32x16 "v_mfma_f32_32x32x16_bf16"
namespace fsvit { typedef __bf16 bf16; }
#endif

namespace FSVIT_NS {

typedef __attribute__((ext_vector_type(8))) bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kPerChunk = 4;      // elements per 16-byte chunk
  static constexpr int kBK = 32;           // elements per 128-byte K slice
};
// fp32 storage whose GEMMs run as two-limb 16-bit MFMAs (conv_gemm_v2.hip x2_split): same layout as float, a distinct type so that the
// kernel instantiation carries its own name in a profile
struct f32x2l { float v; };
template <> struct Elem<f32x2l> : Elem<float> {};
template <> struct Elem<bf16> {
  static constexpr int kPerChunk = 8;
  static constexpr int kBK = 64;
};

// One 64-byte K chunk of a 16x16 output tile: `a` and `b` are the 16 bytes this lane read from
// row (lane&15) at K offset (lane>>4)*16 bytes of the two K-contiguous operands.
// bf16: one v_mfma_f32_16x16x32_bf16.  f32: four v_mfma_f32_16x16x4_f32 (exact fp32 fma chain);
// MFMA j consumes element j of every lane's chunk, i.e. a fixed permutation of k shared by
// both operands, so the dot product covers each k exactly once.
template <typename T> __device__ __forceinline__ f32x4 mma_chunk(u32x4 a, u32x4 b, f32x4 acc);
template <> __device__ __forceinline__ f32x4 mma_chunk<bf16>(u32x4 a, u32x4 b, f32x4 acc) {
#ifdef FSVIT_HALF_F16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
#endif
}
template <> __device__ __forceinline__ f32x4 mma_chunk<float>(u32x4 a, u32x4 b, f32x4 acc) {
  f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], acc, 0, 0, 0);
  return acc;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// erf-form GELU through the Abramowitz-Stegun 7.1.26 erfc polynomial: |erfc error| <= 1.5e-7 (|gelu error| <= ~1e-7 |x|, fp32 rounding
// level) in ~18 VALU issues, against ~60 for erff.  The two-limb GEMM modes use it (their GEMMs deviate by 1e-5 .. 1e-4 already); `parity`
// keeps erff.
__device__ __forceinline__ float gelu_erfc(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float q = fmaf(1.061405429f, t, -1.453152027f);
  q = fmaf(q, t, 1.421413741f);
  q = fmaf(q, t, -0.284496736f);
  q = fmaf(q, t, 0.254829592f);
  const float h = 0.5f * x * (q * t) * __builtin_amdgcn_exp2f(z * z * -1.44269504088896340736f);    // 0.5 x erfc(|x| / sqrt 2)
  return x > 0.0f ? x - h : h;
}

// GELU as x * sigmoid(x * (c0 + c1 x^2 + c2 x^4)): minimax fit of the three coefficients against the exact erf form,
// max |gelu_sig - gelu_erf| = 2.6e-5 over the whole real line (x^2 is clamped at 64, where the sigmoid has long saturated
// and before the quartic turns over) - 300x below the bf16 rounding of the stored activation (2^-9 relative).  9 VALU
// issues incl. 2 transcendentals vs 19 for the Abramowitz-Stegun 7.1.26 erfc form used before (4e-7): the GEMM epilogues of the K <= 512 layers are VALU / store-issue
// bound (an 8-wave 256x256 tile spends 2 x 128 GELUs per lane per SIMD), so this is the bf16 throughput mode's GELU.
// The coefficients carry the -log2(e) of exp2.
__device__ __forceinline__ float gelu_sig(float x) {
  const float u = fminf(x * x, 64.0f);
  float p = fmaf(1.0153755e-3f, u, -1.0678257e-1f);      // -log2e * (c2 u + c1)
  p = fmaf(p, u, -2.3011138f);                           // -log2e * c0
  const float e = __builtin_amdgcn_exp2f(x * p);         // exp(-x * poly)
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// gelu_sig and its derivative d/dx [x s(x)], s = sigmoid(x * poly(x^2)):  s * (1 + x * (1 - s) * q(x^2)),  q(u) = c0 + 3 c1 u + 5 c2 u^2
// (= d/dx [x poly(x^2)]; beyond the clamp of u the sigmoid has saturated and s (1 - s) vanishes).  Training forward epilogues store it next
// to the activation (conv_gemm.h y2).
__device__ __forceinline__ float gelu_sig_d(float x, float& d) {
  const float u = fminf(x * x, 64.0f);
  float p = fmaf(1.0153755e-3f, u, -1.0678257e-1f);
  p = fmaf(p, u, -2.3011138f);
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
  float q = fmaf(5.0f * 1.0153755e-3f, u, 3.0f * -1.0678257e-1f);
  q = fmaf(q, u, -2.3011138f) * -0.69314718055994530942f;      // the coefficients carry -log2(e)
  d = s * fmaf(x * (1.0f - s), q, 1.0f);
  return x * s;
}
// exact erf form and its derivative Phi(x) + x phi(x) (fp32 / parity mode)
__device__ __forceinline__ float gelu_erf_d(float x, float& d) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  d = cdf + x * 0.39894228040143267794f * expf(-0.5f * x * x);
  return x * cdf;
}

// Two GELUs at once on packed fp32 (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 process a register pair per issue; exp2, rcp and
// min stay scalar): 14 VALU issues per pair instead of 18.  The GELU-heavy kernels (stage-1 block: 256 hidden channels x 2
// GELUs per token, fused Mlp) spend more issue slots on GELU than on MFMA, so this is an end-to-end lever there.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_sig2(f32x2 x) {
  f32x2 u = x * x;
  u[0] = fminf(u[0], 64.0f);
  u[1] = fminf(u[1], 64.0f);
  f32x2 p = u * f32x2{1.0153755e-3f, 1.0153755e-3f} + f32x2{-1.0678257e-1f, -1.0678257e-1f};
  p = p * u + f32x2{-2.3011138f, -2.3011138f};
  const f32x2 z = x * p;
  f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(z[0]);
  e[1] = __builtin_amdgcn_exp2f(z[1]);
  e = e + f32x2{1.0f, 1.0f};
  f32x2 r;
  r[0] = __builtin_amdgcn_rcpf(e[0]);
  r[1] = __builtin_amdgcn_rcpf(e[1]);
  return x * r;
}

// gelu_sig2 with the derivative of gelu_sig_d, packed (the training variant of the stage-1 ring kernel stores both)
__device__ __forceinline__ f32x2 gelu_sig2_d(f32x2 x, f32x2& d) {
  f32x2 u = x * x;
  u[0] = fminf(u[0], 64.0f);
  u[1] = fminf(u[1], 64.0f);
  f32x2 p = u * f32x2{1.0153755e-3f, 1.0153755e-3f} + f32x2{-1.0678257e-1f, -1.0678257e-1f};
  p = p * u + f32x2{-2.3011138f, -2.3011138f};
  const f32x2 z = x * p;
  f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(z[0]);
  e[1] = __builtin_amdgcn_exp2f(z[1]);
  e = e + f32x2{1.0f, 1.0f};
  f32x2 s;
  s[0] = __builtin_amdgcn_rcpf(e[0]);
  s[1] = __builtin_amdgcn_rcpf(e[1]);
  f32x2 q = u * f32x2{5.0f * 1.0153755e-3f, 5.0f * 1.0153755e-3f} + f32x2{3.0f * -1.0678257e-1f, 3.0f * -1.0678257e-1f};
  q = (q * u + f32x2{-2.3011138f, -2.3011138f}) * f32x2{-0.69314718055994530942f, -0.69314718055994530942f};
  d = s * ((x * (f32x2{1.0f, 1.0f} - s)) * q + f32x2{1.0f, 1.0f});
  return x * s;
}

// GELU as an LDS table look-up on the bf16-ROUNDED pre-activation (round 6; the bf16 build of stage1_w4.hip and mlp_rows.hip, whose GELU micro-stages
// competed with the MFMAs for the wave's issue slots: 9 VALU + 2 ds_read_u16 per pair of values instead of 17 VALU, 4 of them transcendental -
// tools/probes/gelu_lds.hip).  The kernels carry their pre-activations at 1 / 8 scale (z / 8; W1 / 8 and 8 W2 are exact in bf16), so the table maps
// the bf16 code of z / 8 to bf16(gelu_erf(z) / 8): entry i in [-TN, TN) at byte 2 i from the table's centre, magnitude code TLO + a with a = i
// (z >= 0) or -1 - i (z < 0) - the sign folded in as the one's complement, one contiguous table.  |z| in [2^-10, 32): 15 binades x 128 codes per sign,
// 7680 bytes; smaller magnitudes share the first entry (|gelu| <= 2^-11 there), larger ones the last (z = 31.9).  The fp16 build keeps the VALU
// form: a 10-bit mantissa would need an 8 x larger table.
namespace gelu_tab {
#ifdef FSVIT_HALF_F16
constexpr bool ON = false;
constexpr int NB = 0;
#else
constexpr bool ON = true;
constexpr int NB = 15;
#endif
constexpr int TN = NB * 128, TLO = (127 - 13) << 7, BYTES = 4 * TN;
// every thread of the workgroup (t of nthreads) fills its share of the table at `tab`; the caller orders the stores before the first look-up
__device__ __forceinline__ void fill(unsigned char* tab, int t, int nthreads) {
  for (int e = t; e < 2 * TN; e += nthreads) {
    const int i = e - TN, a = i >= 0 ? i : -1 - i;
    const float z8 = __builtin_bit_cast(float, (unsigned)(TLO + a) << 16);
    const float z = i >= 0 ? 8.0f * z8 : -8.0f * z8;
    reinterpret_cast<bf16*>(tab)[e] = (bf16)(gelu_erf(z) * 0.125f);
  }
}
// packed bf16 code pair of (z0 / 8, z1 / 8) -> packed signed table indices (steps T2 .. T4 of the kernels' micro-stages, here in one piece)
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
typedef short ss2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned rebase(unsigned codes) {       // magnitudes re-based to 2^-13 (saturating at 0)
  constexpr unsigned klo = (unsigned)TLO * 0x10001u;
  return __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(us2_t, codes & 0x7fff7fffu), __builtin_bit_cast(us2_t, klo)));
}
__device__ __forceinline__ unsigned clamp_top(unsigned a) {        // ... and clamped at the last entry
  constexpr unsigned kmax = (unsigned)(TN > 0 ? TN - 1 : 0) * 0x10001u;
  return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, kmax)));
}
__device__ __forceinline__ unsigned sign_mask(unsigned codes) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(ss2_t, codes) >> 15); }
// LDS byte addresses of the two entries: centre + 2 i (v_mad_i32_i16, op_sel picks the upper half)
__device__ __forceinline__ void addresses(unsigned idx, unsigned centre, unsigned& a0, unsigned& a1) {
  asm("v_mad_i32_i16 %0, %1, 2, %2" : "=v"(a0) : "v"(idx), "s"(centre));
  asm("v_mad_i32_i16 %0, %1, 2, %2 op_sel:[1,0,0,0]" : "=v"(a1) : "v"(idx), "s"(centre));
}
typedef __attribute__((address_space(3))) const unsigned short lds_u16;
__device__ __forceinline__ unsigned gather(unsigned addr) { return *(lds_u16*)(size_t)addr; }      // (d16 loads zero the other half under SRAM ECC: two registers, one v_lshl_or)
}  // namespace gelu_tab

// FAST selects gelu_sig (bf16 storage); the fp32 parity path keeps the exact erff form.
template <bool FAST>
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == ACT_GELU) return FAST ? gelu_sig(v) : gelu_erf(v);
  if (act == ACT_LRELU) return v > 0.0f ? v : 0.1f * v;
  return v;
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// 4 consecutive elements <-> 4 floats
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 load4<bf16>(const bf16* p) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
template <> __device__ __forceinline__ f32x4 load4<f32x2l>(const f32x2l* p) { return *reinterpret_cast<const f32x4*>(p); }
template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<f32x2l>(f32x2l* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, f32x4 v) {
  bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  *reinterpret_cast<bf16x4*>(p) = o;
}

// Two-limb split of one 16-byte chunk of fp32 values (conv_gemm_v2.hip / gemm256.hip, the `bf16x2` / `f16x2` modes): s = limb words
// [hi | lo] (upper / lower half), r = the same words half-swapped.
__device__ __forceinline__ void x2_split(const u32x4 v, u32x4& s, u32x4& r) {
  // (the floats come from ONE bit_cast of the whole vector: hipcc 7.2 compiles __builtin_bit_cast(float, v[e]) of a vector element to
  // element 0 for every e)
  const f32x4 fv = __builtin_bit_cast(f32x4, v);
#pragma unroll
  for (int e = 0; e < 4; e += 2) {
    const float x0 = fv[e], x1 = fv[e + 1];
#ifdef FSVIT_HALF_F16
    typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
    const h2_t h = __builtin_bit_cast(h2_t, __builtin_amdgcn_cvt_pkrtz(x0, x1));           // any 11-bit rounding of x keeps x - hi exact
    const h2_t l = __builtin_bit_cast(h2_t, __builtin_amdgcn_cvt_pkrtz(x0 - (float)h[0], x1 - (float)h[1]));
    const unsigned ch = __builtin_bit_cast(unsigned, h), cl = __builtin_bit_cast(unsigned, l);
    s[e] = (ch << 16) | (cl & 0xffffu);             r[e] = (cl << 16) | (ch & 0xffffu);
    s[e + 1] = (ch & 0xffff0000u) | (cl >> 16);     r[e + 1] = (cl & 0xffff0000u) | (ch >> 16);
#else
    typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
    const unsigned u0 = v[e], u1 = v[e + 1];               // hi = the upper 16 bits (truncation; lo picks up the remainder exactly)
    const b2_t l = {(__bf16)(x0 - __builtin_bit_cast(float, u0 & 0xffff0000u)), (__bf16)(x1 - __builtin_bit_cast(float, u1 & 0xffff0000u))};
    const unsigned cl = __builtin_bit_cast(unsigned, l);
    s[e] = (u0 & 0xffff0000u) | (cl & 0xffffu);     r[e] = (cl << 16) | (u0 >> 16);
    s[e + 1] = (u1 & 0xffff0000u) | (cl >> 16);     r[e + 1] = (cl & 0xffff0000u) | (u1 >> 16);
#endif
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace FSVIT_NS
