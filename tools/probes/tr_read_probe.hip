#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
  __shared__ unsigned short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;      // element value = its LDS index
  __syncthreads();
  // lane l supplies the address of element index a(l); print what each lane gets
  const int l = threadIdx.x;
  const int a = ((l >> 4) * 4 + ((l & 15) >> 2)) * 16 + (l & 3) * 4;           // [rows][16 cols] image: row = 4*(l>>4) + (i>>2), col chunk = 4*(i&3)
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + a));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d (supplied elem %4d = row %2d col %2d): got %4d %4d %4d %4d  = (r,c) (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l,
     ((l >> 4) * 4 + ((l & 15) >> 2)) * 16 + (l & 3) * 4, (l >> 4) * 4 + ((l & 15) >> 2), (l & 3) * 4, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3],
     h[l*4]/16, h[l*4]%16, h[l*4+1]/16, h[l*4+1]%16, h[l*4+2]/16, h[l*4+2]%16, h[l*4+3]/16, h[l*4+3]%16);
  return 0;
}
