// Issue cost of the VALU instructions a GELU epilogue is made of, one wave alone on a SIMD (gfx950): cycles per wave-instruction, measured over
// 8 independent dependency chains x 64 repetitions with s_memtime.    hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(NAME, ASM)                                                                                   \
  __global__ void NAME(float* out, long long* cyc) {                                                        \
    float v[8];                                                                                             \
    for (int i = 0; i < 8; ++i) v[i] = 0.5f + 0.001f * (threadIdx.x + i);                                   \
    long long t0 = __builtin_readcyclecounter();                                                            \
    for (int r = 0; r < 4096; ++r) {                                                                        \
      asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                   \
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));   \
    }                                                                                                       \
    long long t1 = __builtin_readcyclecounter();                                                            \
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i];                                                     \
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;                                                         \
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                                \
  }

#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %" #i ", 1.0\n\t"
#define A_MUL(i) "v_mul_f32 %" #i ", %" #i ", %" #i "\n\t"
#define A_EXP(i) "v_exp_f32 %" #i ", %" #i "\n\t"
#define A_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n\t"
#define A_EXPH(i) "v_exp_f16 %" #i ", %" #i "\n\t"
#define A_RCPH(i) "v_rcp_f16 %" #i ", %" #i "\n\t"
#define A_MIN(i) "v_min_f32 %" #i ", %" #i ", 4.0\n\t"
#define A_CVT(i) "v_cvt_pk_bf16_f32 %" #i ", %" #i ", %" #i "\n\t"
#define A_PKFMAH(i) "v_pk_fma_f16 %" #i ", %" #i ", %" #i ", %" #i "\n\t"
#define A_PKMULH(i) "v_pk_mul_f16 %" #i ", %" #i ", %" #i "\n\t"
#define A_PKMINH(i) "v_pk_min_f16 %" #i ", %" #i ", %" #i "\n\t"
#define A_FMAH(i) "v_fma_f16 %" #i ", %" #i ", %" #i ", %" #i "\n\t"

KERNEL(k_fma, A_FMA)
KERNEL(k_mul, A_MUL)
KERNEL(k_exp, A_EXP)
KERNEL(k_rcp, A_RCP)
KERNEL(k_exph, A_EXPH)
KERNEL(k_rcph, A_RCPH)
KERNEL(k_min, A_MIN)
KERNEL(k_cvt, A_CVT)
KERNEL(k_pkfmah, A_PKFMAH)
KERNEL(k_pkmulh, A_PKMULH)
KERNEL(k_pkminh, A_PKMINH)
KERNEL(k_fmah, A_FMAH)

// packed f32 needs register pairs
__global__ void k_pkfma(float* out, long long* cyc) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f2{0.5f + 0.001f * threadIdx.x, 0.25f + i};
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < 4096; ++r) {
#define A_PK(i) "v_pk_fma_f32 %" #i ", %" #i ", %" #i ", %" #i "\n\t"
    asm volatile(A_PK(0) A_PK(1) A_PK(2) A_PK(3) A_PK(4) A_PK(5) A_PK(6) A_PK(7)
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  out[threadIdx.x + blockIdx.x * blockDim.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 8);
  struct { const char* name; void (*k)(float*, long long*); } ks[] = {
      {"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_min_f32", k_min}, {"v_exp_f32", k_exp}, {"v_rcp_f32", k_rcp}, {"v_exp_f16", k_exph}, {"v_rcp_f16", k_rcph},
      {"v_cvt_pk_bf16_f32", k_cvt}, {"v_pk_fma_f32", k_pkfma}, {"v_fma_f16", k_fmah}, {"v_pk_fma_f16", k_pkfmah}, {"v_pk_mul_f16", k_pkmulh}, {"v_pk_min_f16", k_pkminh}};
  for (int waves = 1; waves <= 4; ++waves)          // waves per SIMD (block of 256 threads = 1 per SIMD, 512 = 2)
    for (auto& k : ks) {
      long long h = 0;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k.k, dim3(256), dim3(256 * waves), 0, 0, out, cyc); hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k.k, dim3(256), dim3(256 * waves), 0, 0, out, cyc);
      hipEventRecord(e1, 0); hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("%-20s %d wave(s)/SIMD: %6.2f s_memtime ticks per wave-instruction;  %7.3f ns per wave-instruction per wave, %7.3f ns per instruction per SIMD\n", k.name, waves,
             h / 32768.0, ms * 1e6 / 32768.0, ms * 1e6 / 32768.0 / waves);
    }
  return 0;
}
