// Probe: does the immediate offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void probe(const unsigned* src, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned smem[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) smem[i] = 0xdeadbeefu;
  __syncthreads();
  unsigned lds = (unsigned)(size_t)(lptr_t)smem;
  unsigned voff = threadIdx.x * 16, keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
      "s_mov_b32 m0, %0\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&s"(keep) : "v"(voff), "s"(src), "s"(lds) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) out[i] = smem[i];
}
int main() {
  unsigned *src, *out, h[2048], hs[4096];
  for (int i = 0; i < 4096; ++i) hs[i] = i;
  hipMalloc(&src, sizeof(hs)); hipMalloc(&out, sizeof(h));
  hipMemcpy(src, hs, sizeof(hs), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out);
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("lds[0]=%u lds[255]=%u lds[256]=%u lds[511]=%u lds[512]=%x\n", h[0], h[255], h[256], h[511], h[512]);
  printf(h[0] == 256 ? "RESULT: offset moves the GLOBAL source only\n" : h[256] == 256 ? "RESULT: offset moves BOTH source and LDS destination\n" : "RESULT: unexpected\n");
  return 0;
}
