// Probe (round 6, VERDICT r05 item 4 "conv3x3_halo 32x32x16: prototype or strike"): the INNER LOOP of conv3x3_halo.hip in both MFMA shapes, nothing else
// (no DMA, no epilogue): 8 waves (two per SIMD, one workgroup per CU), per 64-wide K tile and wave
//   S16 (the shipped loop): 2 regions x [20 v_mfma_f32_16x16x32_bf16 on a 5 x 4 block of 16 x 16 tiles + 9 ds_read_b128 (5 pixel + 4 weight fragments)],
//                           a workgroup barrier per K tile (the weight ring's);
//   S32: the same 320 pixels x 128 channels over 8 waves is FIVE 32 x 32 tiles per wave - 5 is prime, the only register blocking is 5 x 1
//        (160 pixels x 32 channels): per 16-wide k chunk 5 MFMAs + 6 fragment reads, 4 chunks per K tile = 20 v_mfma_f32_32x32x16_bf16 + 24 ds_read_b128.
//   (A 2 x 2 blocking - 12 reads per 16 MFMAs - needs 4096 outputs per wave, i.e. a 256-pixel tile: 40-pixel rows do not split that way.)
// Fragments are read from the kernel's own layouts (chunk-planar pixels, 128-byte weight rows) so the LDS traffic is the real one; the reads of a
// region ride behind its first MFMAs as in the kernel.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/halo_tile_shape.hip -o tools/probes/halo_tile_shape && tools/probes/halo_tile_shape
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PLANE = 7168, HBUF = 8 * PLANE, OFF_W = 2 * HBUF, LDS = 160 * 1024;

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void loop(float* out, long long* clk, int ktiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < LDS / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + (i & 0xff);
  __syncthreads();
  long long t0 = 0;
  float s = 0.f;
  if constexpr (SHAPE == 16) {
    const int lrow = lane & 15, lq = lane >> 4, wm = wave >> 1, wn = wave & 1;
    f32x4 acc[5][4];
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned xb[5];
    for (int i = 0; i < 5; ++i) xb[i] = (unsigned)((((wm * 5 + i) / 10 * 4 + (lrow >> 2)) * 44 + ((wm * 5 + i) % 10) * 4 + (lrow & 3)) * 16 + lq * PLANE);
    const unsigned b_rd = (unsigned)((wn * 64 + lrow) * 128), sw0 = (unsigned)((lq ^ (lrow & 7)) << 4), sw1 = (unsigned)(((4 + lq) ^ (lrow & 7)) << 4);
    u32x4 xf0[5], wf0[4], xf1[5], wf1[4];
    for (int j = 0; j < 4; ++j) wf0[j] = *reinterpret_cast<const u32x4*>(smem + OFF_W + b_rd + j * 2048 + sw0);
    for (int i = 0; i < 5; ++i) xf0[i] = *reinterpret_cast<const u32x4*>(smem + xb[i]);
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int kt = 0; kt < ktiles; ++kt) {
      const unsigned toff = (unsigned)((kt % 9) / 3 * 44 + (kt % 9) % 3) * 16, st = (unsigned)(kt % 3) * 16384;
      for (int j = 0; j < 4; ++j) wf1[j] = *reinterpret_cast<const u32x4*>(smem + OFF_W + st + b_rd + j * 2048 + sw1);
      for (int i = 0; i < 5; ++i) xf1[i] = *reinterpret_cast<const u32x4*>(smem + xb[i] + 4 * PLANE + toff);
      for (int i = 0; i < 5; ++i) for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf0[j]), __builtin_bit_cast(bf16x8, xf0[i]), acc[i][j], 0, 0, 0);
      for (int k = 0; k < 9; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
      __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      for (int j = 0; j < 4; ++j) wf0[j] = *reinterpret_cast<const u32x4*>(smem + OFF_W + st + b_rd + j * 2048 + sw0);
      for (int i = 0; i < 5; ++i) xf0[i] = *reinterpret_cast<const u32x4*>(smem + xb[i] + toff + 16);
      for (int i = 0; i < 5; ++i) for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf1[j]), __builtin_bit_cast(bf16x8, xf1[i]), acc[i][j], 0, 0, 0);
      for (int k = 0; k < 9; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
      __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  } else {
    const int p = lane & 31, kh = lane >> 5, wm = wave >> 2, wn = wave & 3;      // wave: 160 pixels (5 blocks of 4 x 8) x 32 channels
    f32x16 acc[5];
    for (int i = 0; i < 5; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    unsigned xb[5];
    for (int i = 0; i < 5; ++i) xb[i] = (unsigned)((((wm * 5 + i) / 5 * 4 + (p >> 3)) * 44 + ((wm * 5 + i) % 5) * 8 + (p & 7)) * 16 + kh * PLANE);
    const unsigned b_rd = (unsigned)((wn * 32 + p) * 128);
    u32x4 xf[2][5], wf[2];
    auto rd = [&](int slot, unsigned st, unsigned toff, int c) {                // k chunk c (16 k = planes 2 c, 2 c + 1) of the tile
      wf[slot] = *reinterpret_cast<const u32x4*>(smem + OFF_W + st + b_rd + (((2 * c + kh) ^ (p & 7)) << 4));
      for (int i = 0; i < 5; ++i) xf[slot][i] = *reinterpret_cast<const u32x4*>(smem + xb[i] + 2 * c * PLANE + toff);
    };
    rd(0, 0, 0, 0);
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int kt = 0; kt < ktiles; ++kt) {
      const unsigned toff = (unsigned)((kt % 9) / 3 * 44 + (kt % 9) % 3) * 16, st = (unsigned)(kt % 3) * 16384;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        rd((c + 1) & 1, st, c == 3 ? toff + 16 : toff, (c + 1) & 3);
        for (int i = 0; i < 5; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[c & 1]), __builtin_bit_cast(bf16x8, xf[c & 1][i]), acc[i], 0, 0, 0);
        for (int k = 0; k < 4; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (c == 1) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
      }
    }
    for (int i = 0; i < 5; ++i) s += acc[i][0] + acc[i][15];
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + t] = s;
  if (t == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int SHAPE> void run(const char* name, float* out, long long* clk) {
  const int kt = 9 * 200;
  (void)hipFuncSetAttribute((const void*)loop<SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  loop<SHAPE><<<256, 512, LDS>>>(out, clk, kt);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  loop<SHAPE><<<256, 512, LDS>>>(out, clk, kt);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h; (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  // a K tile of the workgroup = 320 pixels x 128 channels x 64 k = 5.24 MFLOP; MFMA time per SIMD: 2 waves x 40 x 16 (or 20 x 32) = 1280 cycles
  printf("%-40s %8.1f cycles / K tile (wave 0; 1280 = the matrix pipe's time), %7.1f ns / K tile, %6.1f TFLOP/s on 256 CUs\n", name, (double)h / kt, ms * 1e6 / kt,
         256 * 2.0 * 320 * 128 * 64 / (ms * 1e-3 / kt) / 1e12);
}

int main() {
  float* out; long long* clk;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&clk, 8);
  for (int rep = 0; rep < 2; ++rep) {
    run<16>("S16: 5 x 4 tiles of 16x16x32 (shipped)", out, clk);
    run<32>("S32: 5 x 1 tiles of 32x32x16", out, clk);
  }
  return 0;
}
