// Probe: how fast can one CU bring an L2-resident weight image into LDS, and by which instruction?
// Every workgroup (one per CU, 256 workgroups) sweeps the same 256 KB image REPS times in 32 KB slabs into a 2-slab LDS ring.
//   mode 0  global_load_lds_dwordx4 (LDS-DMA, 1 KB per wave instruction), vmcnt(0) per slab
//   mode 1  global_load_dwordx4 -> VGPR -> ds_write_b128 (8 loads in flight per lane)
//   mode 2  global_load_lds_dword   (LDS-DMA, 256 B per wave instruction)
//   mode 3  like 0 but each workgroup reads its own 256 KB image (no sharing in L2 / MALL between CUs)
// NW = waves per workgroup (4, 8, 16).  Prints bytes / clock / CU from the wall time at the measured shader clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int IMG = 256 * 1024, SLAB = 32 * 1024;

__device__ __forceinline__ void dma16(const void* g, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ void dma4(const void* g, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void fill(const unsigned char* img, unsigned* sink, int reps, long long* clk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const unsigned char* src = img + (MODE == 3 ? (size_t)blockIdx.x * IMG : 0);
  unsigned acc = 0;
  const long long c0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll 1
    for (int s = 0; s < IMG / SLAB; ++s) {
      const unsigned char* g = src + s * SLAB;
      const unsigned dst = lds0 + (s & 1) * SLAB;
      if (MODE == 0 || MODE == 3) {
#pragma unroll
        for (int i = 0; i < SLAB / 1024 / NW; ++i) dma16(g + (i * NW + w) * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(dst + (i * NW + w) * 1024));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (MODE == 2) {
#pragma unroll 8
        for (int i = 0; i < SLAB / 256 / NW; ++i) dma4(g + (i * NW + w) * 256 + lane * 4, __builtin_amdgcn_readfirstlane(dst + (i * NW + w) * 256));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        constexpr int N = SLAB / 1024 / NW;
        u32x4 v[N];
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = *reinterpret_cast<const u32x4*>(g + (i * NW + w) * 1024 + lane * 16);
#pragma unroll
        for (int i = 0; i < N; ++i) *reinterpret_cast<u32x4*>(smem + (s & 1) * SLAB + (i * NW + w) * 1024 + lane * 16) = v[i];
      }
      __syncthreads();
      acc += *reinterpret_cast<unsigned*>(smem + (s & 1) * SLAB + ((t * 68) & (SLAB - 4)));
    }
  }
  const long long c1 = clock64();
  if (acc == 0x12345678u) sink[0] = acc;
  if (t == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}

template <int MODE, int NW>
static void run(const char* name, const unsigned char* img, unsigned* sink, long long* clk, int reps, int mhz) {
  hipFuncSetAttribute((const void*)fill<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLAB);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((fill<MODE, NW>), dim3(256), dim3(NW * 64), 2 * SLAB, 0, img, sink, 2, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((fill<MODE, NW>), dim3(256), dim3(NW * 64), 2 * SLAB, 0, img, sink, reps, clk);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long c = 0;
  hipMemcpy(&c, clk, sizeof(c), hipMemcpyDeviceToHost);
  const double bytes = (double)IMG * reps;
  printf("%-44s NW=%2d  %8.3f ms  %6.1f B/clk/CU (wall @ %d MHz)   %6.2f TB/s aggregate   s_memtime ticks %lld\n", name, NW, ms,
         bytes / (ms * 1e-3 * mhz * 1e6), mhz, bytes * 256 / (ms * 1e-3) / 1e12, c);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  int mhz = 2400;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  if (p.clockRate > 0) mhz = p.clockRate / 1000;
  unsigned char* img; unsigned* sink; long long* clk;
  hipMalloc(&img, (size_t)IMG * 256); hipMemset(img, 1, (size_t)IMG * 256);
  hipMalloc(&sink, 64); hipMalloc(&clk, 64);
  printf("device %s, %d CUs, clockRate %d MHz\n", p.name, p.multiProcessorCount, mhz);
  run<0, 4>("LDS-DMA dwordx4, shared image", img, sink, clk, reps, mhz);
  run<0, 8>("LDS-DMA dwordx4, shared image", img, sink, clk, reps, mhz);
  run<0, 16>("LDS-DMA dwordx4, shared image", img, sink, clk, reps, mhz);
  run<1, 4>("global_load_dwordx4 + ds_write_b128, shared", img, sink, clk, reps, mhz);
  run<1, 8>("global_load_dwordx4 + ds_write_b128, shared", img, sink, clk, reps, mhz);
  run<1, 16>("global_load_dwordx4 + ds_write_b128, shared", img, sink, clk, reps, mhz);
  run<2, 4>("LDS-DMA dword, shared image", img, sink, clk, reps, mhz);
  run<2, 16>("LDS-DMA dword, shared image", img, sink, clk, reps, mhz);
  run<3, 4>("LDS-DMA dwordx4, private image per CU", img, sink, clk, reps, mhz);
  run<3, 16>("LDS-DMA dwordx4, private image per CU", img, sink, clk, reps, mhz);
  return 0;
}
