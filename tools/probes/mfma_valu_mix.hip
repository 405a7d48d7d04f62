// Does VALU work hide behind MFMAs on gfx950?  Every wave runs R repetitions of { 1 x v_mfma_f32_16x16x32_bf16 ; N x v_fma_f32 } on independent
// registers (4 accumulators, 8 VALU chains); W waves per SIMD on all 256 CUs; wall clock per repetition per SIMD.
//   hipcc -O3 --offload-arch=gfx950 mfma_valu_mix.hip -o mfma_valu_mix && ./mfma_valu_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

template <int N, bool TRANS>
__global__ void mix(float* out) {
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  b8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * threadIdx.x); b[i] = (__bf16)(0.02f * i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.5f + 0.001f * (threadIdx.x + i);
  for (int r = 0; r < 2048; ++r) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < N; ++n) {
        if (TRANS && (n % 4) == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(j * N + n) & 7]));
        else asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(v[(j * N + n) & 7]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int N, bool TRANS>
void run(float* out) {
  for (int waves = 1; waves <= 4; waves *= 2) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mix<N, TRANS>), dim3(256), dim3(256 * waves), 0, 0, out); hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((mix<N, TRANS>), dim3(256), dim3(256 * waves), 0, 0, out);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / (2048.0 * 4) / waves;            // ns per (MFMA + N VALU) per SIMD
    printf("1 MFMA + %d VALU%s, %d wave(s)/SIMD: %7.2f ns per unit per SIMD  (MFMA alone at peak = 6.7 ns @ 2.4 GHz)\n", N, TRANS ? " (1 in 4 v_exp)" : "", waves, per);
  }
}

int main() {
  float* out; hipMalloc(&out, 4 << 20);
  run<0, false>(out); run<2, false>(out); run<4, false>(out); run<6, false>(out); run<8, false>(out); run<12, false>(out); run<8, true>(out);
  return 0;
}
