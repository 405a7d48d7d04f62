// Probe: issue rate of v_mfma_f32_32x32x16_bf16 as a function of the number of independent accumulator chains (1, 2, 4) on gfx950,
// with VGPR and with AGPR accumulators.  One wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NCH, bool AGPR>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* clk, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i % NCH]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i % NCH]) : "v"(a), "v"(b));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int NCH, bool AGPR> static void run(float* out, long long* clk) {
  const int iters = 20000;
  hipLaunchKernelGGL((probe<NCH, AGPR>), dim3(256), dim3(256), 0, 0, out, clk, 100);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((probe<NCH, AGPR>), dim3(256), dim3(256), 0, 0, out, clk, iters);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  printf("chains=%d %s accumulators: %.1f shader cycles per MFMA\n", NCH, AGPR ? "AGPR" : "VGPR", (double)c / (iters * 8.0));
}
int main() {
  float* out; long long* clk;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 64);
  run<1, false>(out, clk); run<2, false>(out, clk); run<4, false>(out, clk);
  run<1, true>(out, clk); run<2, true>(out, clk); run<4, true>(out, clk);
  return 0;
}
