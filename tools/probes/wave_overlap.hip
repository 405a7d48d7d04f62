// Probe: do the MFMA block of one wave and the VALU block of ANOTHER wave of the same SIMD overlap on gfx950?
// A workgroup of 64 * 4 * W threads = W waves per SIMD (one workgroup per CU, no barriers).  Each wave loops over
//   [A independent v_mfma_f32_16x16x32_bf16 (A accumulators)]  [B VALU instructions of a class, 8 independent chains]
// The waves of a SIMD start half a period apart when STAGGER = 1 (odd waves run their VALU block first).
// Prints shader cycles per iteration per wave; sum model = A * 16 + B * c_valu per wave and iteration (x W on the SIMD),
// overlap model = max over the two pipes.
//   class 0: v_fma_f32   1: v_pk_fma_f32   2: v_exp_f32   3: v_cndmask / v_cmp pair   4: v_cvt_pk_bf16_f32
//   round 4 (a GELU on packed fp16 pairs?): 5: v_pk_fma_f16   6: v_pk_mul_f16   7: v_exp_f16   8: v_cvt_pkrtz_f16_f32   9: v_fma_f16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CLS> __device__ __forceinline__ void valu(f32x2& v, float c) {
  if (CLS == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[0]) : "v"(c));
  if (CLS == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(f32x2{c, c}));
  if (CLS == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[0]));
  if (CLS == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[0]) : "v"(c));
  if (CLS == 4) { unsigned o; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o) : "v"(v[0]), "v"(v[1])); v[0] = __builtin_bit_cast(float, o); }
  if (CLS == 5) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(v[0]) : "v"(c));
  if (CLS == 6) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v[0]) : "v"(c));
  if (CLS == 7) asm volatile("v_exp_f16 %0, %0" : "+v"(v[0]));
  if (CLS == 8) { unsigned o; asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(o) : "v"(v[0]), "v"(v[1])); v[0] = __builtin_bit_cast(float, o); }
  if (CLS == 9) asm volatile("v_fma_f16 %0, %0, %1, %0" : "+v"(v[0]) : "v"(c));
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CLS, int A, int B, int STAGGER, int M32 = 0>
__global__ __launch_bounds__(1024, 1) void probe(float* out, long long* clk, int iters, float c) {
  f32x4 acc[A > 0 ? A : 1];
  f32x16 acc32[A > 0 ? A : 1];
  for (int i = 0; i < (A > 0 ? A : 1); ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; for (int e = 0; e < 16; ++e) acc32[i][e] = 0.f; }
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2{c * (threadIdx.x + i), c};
  const int wave = threadIdx.x >> 6;
  const bool valu_first = STAGGER && ((wave >> 2) & 1);
  const long long t0 = __builtin_readcyclecounter();
  if (valu_first) {
#pragma unroll
    for (int k = 0; k < B; ++k) valu<CLS>(v[k & 7], c);
  }
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < A; ++i) {
      if (M32) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc32[i]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
#pragma unroll
    for (int k = 0; k < B; ++k) valu<CLS>(v[k & 7], c);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < (A > 0 ? A : 1); ++i) s += acc[i][0] + acc[i][3] + acc32[i][0] + acc32[i][15];
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int CLS, int A, int B, int STAGGER, int M32 = 0> void run(const char* name, int W, float* out, long long* clk) {
  const int iters = 2000;
  probe<CLS, A, B, STAGGER, M32><<<256, 256 * W>>>(out, clk, iters, 1.0001f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  probe<CLS, A, B, STAGGER, M32><<<256, 256 * W>>>(out, clk, iters, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  printf("%-10s A=%2d B=%3d W=%d stagger=%d: %8.1f s_memtime ticks / iteration (wave 0), %8.1f ns / iteration (event)\n", name, A, B, W, STAGGER, (double)h / iters, ms * 1e6 / iters);
}

int main() {
  float* out; long long* clk;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 8);
#define SET(CLS, NAME)                                                                                       \
  run<CLS, 8, 0, 0>(NAME " mfma only", 1, out, clk); run<CLS, 8, 0, 0>(NAME " mfma only", 2, out, clk);      \
  run<CLS, 0, 32, 0>(NAME " valu only", 1, out, clk); run<CLS, 0, 32, 0>(NAME " valu only", 2, out, clk);    \
  run<CLS, 8, 32, 0>(NAME, 1, out, clk); run<CLS, 8, 32, 0>(NAME, 2, out, clk); run<CLS, 8, 32, 1>(NAME, 2, out, clk); \
  run<CLS, 8, 32, 0>(NAME, 4, out, clk); run<CLS, 8, 32, 1>(NAME, 4, out, clk);
  // 32x32x16 MFMAs (32 cycles each): 4 per iteration = the same 128 matrix cycles
  run<3, 4, 0, 0, 1>("max m32 mfma only", 1, out, clk); run<3, 4, 0, 0, 1>("max m32 mfma only", 2, out, clk);
  run<3, 4, 32, 0, 1>("max m32", 1, out, clk); run<3, 4, 32, 0, 1>("max m32", 2, out, clk); run<3, 4, 32, 0, 1>("max m32", 4, out, clk);
  run<3, 4, 64, 0, 1>("max m32 B64", 1, out, clk); run<3, 4, 64, 0, 1>("max m32 B64", 2, out, clk);
  run<3, 8, 64, 0, 0>("max m16 B64", 1, out, clk); run<3, 8, 64, 0, 0>("max m16 B64", 2, out, clk);
  run<2, 4, 32, 0, 1>("exp m32", 2, out, clk); run<1, 4, 32, 0, 1>("pk_fma m32", 2, out, clk);
  if (getenv("WO_ALL")) {
  SET(0, "fma")
  SET(1, "pk_fma")
  SET(2, "exp")
  SET(3, "max")
  SET(4, "cvt_pk")
  }
  SET(5, "pk_fma_f16")
  SET(6, "pk_mul_f16")
  SET(7, "exp_f16")
  SET(8, "cvt_pkrtz")
  SET(9, "fma_f16")
  run<5, 4, 32, 0, 1>("pk_fma_f16 m32", 1, out, clk); run<5, 4, 32, 0, 1>("pk_fma_f16 m32", 2, out, clk);
  return 0;
}
