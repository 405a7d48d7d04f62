// Semantics probe: __builtin_amdgcn_permlane32_swap(x, x) on gfx950 - which lanes end up where (used by xhalf_max / xhalf_sum in mlp_rows.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned x = threadIdx.x;
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  out[threadIdx.x] = r[0];
  out[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; unsigned h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("r[0]: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[0], h[31], h[32], h[63]);
  printf("r[1]: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[64], h[95], h[96], h[127]);
  return 0;
}
