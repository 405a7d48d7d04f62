// Probe: do VALU instructions of the SAME wave issue in the shadow of its MFMAs on gfx950, and what do the GELU's instruction
// classes cost?  One wave per SIMD (256 threads, one workgroup per CU).  Each loop iteration = 4 independent
// v_mfma_f32_32x32x16_bf16 (4 accumulators) with K VALU instructions of a class after every MFMA.
//   class 0: v_pk_fma_f32   class 1: v_exp_f32   class 2: v_rcp_f32   class 3: v_fma_f32   class 4: v_cvt_pk_bf16_f32   class 5: v_min_f32
//   class 6: v_pk_fma_f16   class 7: v_exp_f16   class 8: v_rcp_f16   class 9: v_cvt_pkrtz_f16_f32
// Prints shader cycles per (MFMA + its K VALU ops) from s_memtime.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int CLS> __device__ __forceinline__ void valu(f32x2& v, float c) {
  if (CLS == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(f32x2{c, c}));
  if (CLS == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[0]));
  if (CLS == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[0]));
  if (CLS == 3) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[0]) : "v"(c));
  if (CLS == 4) { unsigned o; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o) : "v"(v[0]), "v"(v[1])); v[0] = __builtin_bit_cast(float, o); }
  if (CLS == 5) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[0]) : "v"(c));
  if (CLS == 6) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(v[0]) : "v"(c));
  if (CLS == 7) asm volatile("v_exp_f16 %0, %0" : "+v"(v[0]));
  if (CLS == 8) asm volatile("v_rcp_f16 %0, %0" : "+v"(v[0]));
  if (CLS == 9) { unsigned o; asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(o) : "v"(v[0]), "v"(v[1])); v[0] = __builtin_bit_cast(float, o); }
}

template <int CLS, int K, bool MFMA>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* clk, int iters, float c) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2{c * (threadIdx.x + i), c};
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < K; ++k) valu<CLS>(v[(i * K + k) & 7], c);      // 8 independent chains
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int CLS, int K, bool MFMA> static void run(const char* cls, float* out, long long* clk) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<CLS, K, MFMA>), dim3(256), dim3(256), 0, 0, out, clk, 100, 1e-9f);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<CLS, K, MFMA>), dim3(256), dim3(256), 0, 0, out, clk, iters, 1e-9f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  printf("%-18s K=%2d %s  %7.1f ns per MFMA slot   (%6.1f clk @2.4GHz)  s_memtime/slot %.1f\n", cls, K, MFMA ? "MFMA+VALU" : "VALU only", ms * 1e6 / (iters * 4.0),
         ms * 1e6 / (iters * 4.0) * 2.4, (double)c / (iters * 4.0));
}

int main() {
  float* out; long long* clk;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 64);
  run<0, 0, true>("-", out, clk);
#define ROW(CLS, NAME) \
  run<CLS, 2, true>(NAME, out, clk); run<CLS, 4, true>(NAME, out, clk); run<CLS, 6, true>(NAME, out, clk); run<CLS, 8, true>(NAME, out, clk); \
  run<CLS, 4, false>(NAME, out, clk); run<CLS, 8, false>(NAME, out, clk);
  ROW(0, "v_pk_fma_f32")
  ROW(1, "v_exp_f32")
  ROW(2, "v_rcp_f32")
  ROW(3, "v_fma_f32")
  ROW(4, "v_cvt_pk_bf16_f32")
  ROW(5, "v_min_f32")
  ROW(6, "v_pk_fma_f16")
  ROW(7, "v_exp_f16")
  ROW(8, "v_rcp_f16")
  ROW(9, "v_cvt_pkrtz_f16_f32")
  return 0;
}
