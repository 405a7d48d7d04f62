// Probe: issue rate of v_mfma_f32_32x32x16_bf16 as a function of the number of independent accumulator chains (1, 2, 4) on gfx950,
// with VGPR and with AGPR accumulators, and of v_mfma_f32_16x16x32_bf16 (1..8 chains, 1 and 2 waves per SIMD).  One wave per SIMD unless noted.
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
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NCH, int NWAVE>
__global__ __launch_bounds__(NWAVE * 64, 1) void probe16(float* out, long long* clk, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i % NCH]) : "v"(a), "v"(b));
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  out[blockIdx.x * NWAVE * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int NCH, int NWAVE> static void run16(float* out, long long* clk) {
  // wall-clock timing: with several waves per SIMD the oldest wave wins the issue arbitration, so one wave's own cycle counter says
  // nothing about the throughput of the SIMD
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe16<NCH, NWAVE>), dim3(256), dim3(NWAVE * 64), 0, 0, out, clk, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe16<NCH, NWAVE>), dim3(256), dim3(NWAVE * 64), 0, 0, out, clk, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_simd_ns = ms * 1e6 / (iters * 16.0 * (NWAVE / 4));
  printf("16x16x32: chains=%d, %d wave(s) per SIMD: %.2f ns per MFMA per SIMD = %.0f TFLOP/s over 1024 SIMDs\n", NCH, NWAVE / 4, per_simd_ns,
         16384.0 * 1024 / per_simd_ns / 1e3);
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
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 64);
  run<1, false>(out, clk); run<2, false>(out, clk); run<4, false>(out, clk);
  run<1, true>(out, clk); run<2, true>(out, clk); run<4, true>(out, clk);
  run16<1, 4>(out, clk); run16<2, 4>(out, clk); run16<4, 4>(out, clk); run16<8, 4>(out, clk); run16<4, 8>(out, clk); run16<8, 8>(out, clk); run16<8, 16>(out, clk);
  return 0;
}
