// Probe (round 6): can the GELU of stage1_w4 leave the VALU port?
// One wave per SIMD (256 threads, 512 registers, one workgroup per CU), a 32-slot segment like S2 of stage1_w4.hip:
//   slot = one v_mfma_f32_32x32x16_bf16 + the ds_read_b128 of the fragment three slots ahead + the GELU micro-stages due at the slot;
//   NP pairs of GELUs per segment (16 = the kernel's ratio: 140 MFMAs and 64 pairs per chunk and wave; 64 = VERDICT's "32 MFMAs + 128 GELUs"),
//   results packed to 16 bits and stored with one ds_write_b128 per four pairs.
// MODE 0: gelu_sig on the VALU exactly as stage1_w4's micro-stages (17 VALU per pair, 4 of them transcendental).
// MODE 1: TABLE.  The pre-activation pair is rounded to bf16 (v_cvt_pk_bf16_f32), the 15-bit magnitude code re-based and clamped with two
//         saturating packed u16 operations, the sign folded in as the one's complement (code ^ (code >>a 15)): a signed index in [-N, N) into ONE
//         contiguous table centred at `tab` (entry i = bf16(gelu) of the code, 2 bytes), fetched with ds_read_u16_d16 / _d16_hi so that the pair
//         8 VALU + 2 LDS gathers + 1 v_or_b32 per pair, no transcendental.
//         Table: |z / 8| in [2^-13, 1) = 13 binades x 128 codes per sign = 3328 entries = 6656 bytes; REPL replicas (lane & (REPL - 1)) spread the banks.
// MODE 2: no GELU at all (the segment's floor);  MODE 3: table arithmetic without the gathers (what the 8 VALU cost);
// Inputs: DIST 0 = z ~ N(0, 1) (realistic: indices cluster in a few binades), 1 = one value for the whole wave (broadcast, no conflicts),
//         2 = log-uniform over the whole table (worst case for the banks).
// Prints shader cycles per segment (s_memtime of wave 0 of workgroup 0), ns per segment from HIP events over the grid, and the largest
// difference between the table's result and bf16(gelu_erf(bf16(z))) - the index arithmetic is checked, not just timed.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/gelu_lds.hip -o /tmp/gelu_lds && /tmp/gelu_lds
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <utility>
#include <type_traits>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int... I, typename F>
__device__ __forceinline__ void for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void cfor(F&& f) { for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

constexpr int NB = 13, NENT = NB * 128;            // codes per sign
constexpr int LO_CODE = (127 - 13) << 7;           // bf16 code of 2^-13
constexpr int TAB_BYTES = 2 * NENT * 2;            // 6656
constexpr int FRAG = 0;                            // 16 fragments x 64 lanes x 16 B = 16 KB
constexpr int OUT = 16384;                         // 64 pairs x 4 B x 64 lanes ... 16 KB per wave x 4
constexpr int TAB = OUT + 4 * 16384;               // REPL x 6656
constexpr int LDS = 160 * 1024 - 256;              // one workgroup per CU

// ---- compile-time DS bookkeeping: program order of every LDS operation of a segment (fragment reads, gathers, octet stores)
template <int NP> struct Sched {
  // pair k starts at slot st(k); table stages: T1 st, T2 st+1, T3 st+2, T4 (addresses) st+3, gather st+4, consume st+4+LAT
  static constexpr int LAT = 3;
  static constexpr int st(int k) { return k * 28 / NP; }      // 16 pairs: 1.75 slots per pair (stage1_w4's N7); 64 pairs: 0.4375
  int frag_seq[32 + 3] = {}, gath_seq[NP] = {}, n_at_wait_frag[32] = {}, n_at_wait_gath[NP] = {}, total = 0;
  constexpr Sched(bool table) {
    int seq = 0;
    for (int f = 0; f < 3; ++f) frag_seq[f] = seq++;                       // prologue
    for (int r = 0; r < 32 + LAT + 6; ++r) {
      if (r < 32) { n_at_wait_frag[r] = seq; if (r + 3 < 32) frag_seq[r + 3] = seq++; }
      for (int k = 0; k < NP; ++k) {                                          // (the kernel's order: by pair; a pair's gather, then its octet's store)
        if (table && r == st(k) + 4) { gath_seq[k] = seq + 1; seq += 2; }         // (seq of the SECOND gather of the pair)
        if (table && r == st(k) + 4 + LAT && (k & 3) == 3) { n_at_wait_gath[k] = seq; seq++; }    // wait, then the octet's store
        if (!table && r == st(k) + 5 && (k & 3) == 3) seq++;                                       // VALU mode: the store only
      }
    }
    total = seq;
  }
};

__device__ __forceinline__ void lgkm_wait(int n) {            // n is a compile-time constant at every call site
  if (n > 14) n = 14;                                          // (4-bit counter; a smaller count only waits for more)
  switch (n) {
#define C(N) case N: asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); break;
    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14)
#undef C
  }
}

template <int MODE, int NP, int REPL, int NACC = 4>
__global__ __launch_bounds__(256, 1) void probe(const float* __restrict__ xin, const unsigned short* __restrict__ tab_g, unsigned* __restrict__ out,
                                                long long* __restrict__ clk, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int i = t; i < REPL * TAB_BYTES / 2; i += 256) reinterpret_cast<unsigned short*>(smem + TAB)[i] = tab_g[i % (TAB_BYTES / 2)];
  for (int i = t; i < 16384 / 4; i += 256) reinterpret_cast<unsigned*>(smem + FRAG)[i] = 0u;        // zero B fragments: the accumulators keep their inputs
  __syncthreads();
  constexpr int NT = NP / 8;                                   // accumulator tiles read by the GELU (16 values = 8 pairs each)
  f32x16 acc[4], zin[NT];                                      // (the GELU reads a tile the segment's MFMAs do not write, as in the kernel)
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int i = 0; i < NT; ++i)
    for (int e = 0; e < 16; ++e) zin[i][e] = xin[((size_t)(blockIdx.x * 256 + t) * 8 + (i & 7)) * 16 + e] * 0.125f;      // z / 8 as in stage1_w4
  u32x4 wf = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  asm volatile("" : "+a"(wf));
  const unsigned fragb = FRAG + lane * 16;
  const unsigned outb = OUT + w * 16384 + lane * 16;
  // table centre of this lane's replica: entry i (signed) at tabc + 2 i
  const int tabc = TAB + (lane & (REPL - 1)) * TAB_BYTES + NENT * 2;
  constexpr Sched<NP> S(MODE == 1);
  constexpr bool TABLE = MODE == 1 || MODE == 3;

  float gx[NP][2], gu[NP][2];
  unsigned gp[NP], gc[NP], ga[NP], gm[NP], ad0[NP], ad1[NP], glo[NP], ghi[NP];
  const unsigned k7fff = 0x7fff7fffu, klo = (unsigned)LO_CODE * 0x10001u, kmax = (unsigned)(NENT - 1) * 0x10001u;
  u32x4 fr[4];

  long long t0 = 0, t1 = 0;
#pragma unroll 1
  for (int it = -1; it < iters; ++it) {
    if (it == 0) t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < NT; ++i) asm volatile("" : "+v"(zin[i]));          // (the pre-activations are new every segment: nothing of the GELU is loop-invariant)
    for (int f = 0; f < 3; ++f) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[f]) : "v"(fragb), "n"(0));
    __builtin_amdgcn_sched_barrier(0);
    cfor<32 + S.LAT + 6>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      if constexpr (r < 32) {
        lgkm_wait(S.n_at_wait_frag[r] - S.frag_seq[r] - 1);
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[r % NACC]) : "a"(wf), "v"(fr[r & 3]));
        if constexpr (r + 3 < 32) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(r + 3) & 3]) : "v"(fragb), "n"(((r + 3) & 15) * 1024));
      }
      if constexpr (MODE != 2) cfor<NP>([&](auto kc) {
        constexpr int k = decltype(kc)::value, st = Sched<NP>::st(k);
        (void)outb; (void)tabc; (void)klo; (void)kmax; (void)k7fff;      // (captures named outside the discarded branches)
        if constexpr (!TABLE) {
          if constexpr (r == st) {
#pragma unroll
            for (int h = 0; h < 2; ++h) { gx[k][h] = zin[k >> 3][2 * (k & 7) + h]; asm("v_mul_f32_e64 %0, %1, %1 clamp" : "=v"(gu[k][h]) : "v"(gx[k][h])); }
          }
          if constexpr (r == st + 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              float pp = fmaf(1.0153755e-3f * 32768.0f, gu[k][h], -1.0678257e-1f * 512.0f);
              pp = fmaf(pp, gu[k][h], -2.3011138f * 8.0f);
              gu[k][h] = gx[k][h] * pp;
            }
          }
          if constexpr (r == st + 2) { gu[k][0] = __builtin_amdgcn_exp2f(gu[k][0]); gu[k][1] = __builtin_amdgcn_exp2f(gu[k][1]); }
          if constexpr (r == st + 3) { gu[k][0] = 1.0f + gu[k][0]; gu[k][1] = 1.0f + gu[k][1]; }
          if constexpr (r == st + 4) { gu[k][0] = __builtin_amdgcn_rcpf(gu[k][0]); gu[k][1] = __builtin_amdgcn_rcpf(gu[k][1]); }
          if constexpr (r == st + 5) {
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(gp[k]) : "v"(gx[k][0] * gu[k][0]), "v"(gx[k][1] * gu[k][1]));
            if constexpr ((k & 3) == 3) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(outb), "v"(u32x4{gp[k - 3], gp[k - 2], gp[k - 1], gp[k]}), "n"((k >> 2) * 1024) : "memory");
          }
        }
      });
      // table stages: two passes per slot - the FIRST operation of every due stage, then the SECOND - so that no packed 16-bit result is read by the
      // next instruction (gfx950 needs a wait state there: hipcc puts an s_nop 0 between such a pair)
      if constexpr (TABLE) cfor<2 * NP>([&](auto kc) {
        constexpr int k = decltype(kc)::value % NP, ps = decltype(kc)::value / NP, st = Sched<NP>::st(k);
        (void)outb; (void)tabc; (void)klo; (void)kmax; (void)k7fff; (void)ad0; (void)ad1; (void)ga; (void)gc; (void)gm; (void)zin;
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        typedef short ss2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bb2 __attribute__((ext_vector_type(2)));
        if constexpr (r == st) {                        // T1: round the pair to bf16; magnitude codes
          if constexpr (ps == 0) gc[k] = __builtin_bit_cast(unsigned, bb2{(__bf16)zin[k >> 3][2 * (k & 7)], (__bf16)zin[k >> 3][2 * (k & 7) + 1]});
          else { asm("" : "+v"(gc[k])); ga[k] = gc[k] & k7fff; }     // (opaque: otherwise the sign shift below re-converts the two floats)
        }
        if constexpr (r == st + 1) {                    // T2: re-base to 2^-13 (saturating at 0) and clamp at the last entry
          if constexpr (ps == 0) ga[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, ga[k]), __builtin_bit_cast(us2, klo)));
          else ga[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(us2, ga[k]), __builtin_bit_cast(us2, kmax)));
        }
        if constexpr (r == st + 2) {                    // T3: sign as the one's complement: i = a (x >= 0), -1 - a (x < 0)
          if constexpr (ps == 0) gm[k] = __builtin_bit_cast(unsigned, __builtin_bit_cast(ss2, gc[k]) >> 15);
          else ga[k] ^= gm[k];
        }
        if constexpr (r == st + 3) {                    // T4: byte addresses tabc + 2 i
          if constexpr (ps == 0) asm("v_mad_i32_i16 %0, %1, 2, %2" : "=v"(ad0[k]) : "v"(ga[k]), "v"(tabc));
          else asm("v_mad_i32_i16 %0, %1, 2, %2 op_sel:[1,0,0,0]" : "=v"(ad1[k]) : "v"(ga[k]), "v"(tabc));
        }
      });
      if constexpr (TABLE) cfor<NP>([&](auto kc) {
        constexpr int k = decltype(kc)::value, st = Sched<NP>::st(k);
        (void)outb; (void)ad0; (void)ad1; (void)glo; (void)ghi;
        {
          if constexpr (MODE == 1) {
            if constexpr (r == st + 4) {                // the gathers: the pair lands packed
              // (MI355X runs with SRAM ECC: a d16 load ZEROES the other half of its destination instead of keeping it - measured with this probe -
              // so the pair cannot land packed in one register: two registers and one v_or_b32)
              asm volatile("ds_read_u16_d16 %0, %2\n\tds_read_u16_d16_hi %1, %3" : "=&v"(glo[k]), "=&v"(ghi[k]) : "v"(ad0[k]), "v"(ad1[k]));
            }
            if constexpr (r == st + 4 + S.LAT && (k & 3) == 3) {
              lgkm_wait(S.n_at_wait_gath[k] - S.gath_seq[k] - 1);
              asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(outb), "v"(u32x4{glo[k - 3] | ghi[k - 3], glo[k - 2] | ghi[k - 2], glo[k - 1] | ghi[k - 1], glo[k] | ghi[k]}), "n"((k >> 2) * 1024) : "memory");
            }
          } else {
            if constexpr (r == st + 4 && (k & 3) == 3)
              asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(outb), "v"(u32x4{ad0[k - 3] ^ ad1[k - 3], ad0[k - 2] ^ ad1[k - 2], ad0[k - 1] ^ ad1[k - 1], ad0[k] ^ ad1[k]}), "n"((k >> 2) * 1024) : "memory");
          }
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  t1 = __builtin_amdgcn_s_memtime();
  // results of the last segment: the packed pairs as stored
  __syncthreads();
  for (int o = 0; o < NP / 4; ++o) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + outb + o * 1024);
    for (int j = 0; j < 4; ++j) out[((size_t)(blockIdx.x * 256 + t) * (NP / 4) + o) * 4 + j] = v[j];
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  if (s == 12345.678f) out[0] = 1;
  if (t == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; __builtin_memcpy(&f, &u, 4); return f; }
static unsigned short f2bf(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static double gelu(double x) { return 0.5 * x * (1.0 + erf(x * 0.70710678118654752440)); }

template <int MODE, int NP, int REPL, int NACC = 4> void run(const char* name, int dist, const float* xin, const unsigned short* tab, unsigned* out, long long* clk, const std::vector<float>& hx) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)probe<MODE, NP, REPL, NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  probe<MODE, NP, REPL, NACC><<<256, 256, LDS>>>(xin, tab, out, clk, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  probe<MODE, NP, REPL, NACC><<<256, 256, LDS>>>(xin, tab, out, clk, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  double maxd = -1.0;
  if (MODE == 0 || MODE == 1) {                               // compare with bf16(gelu(bf16 z)) / 8 (table) or bf16(gelu(z) / 8) (VALU)
    std::vector<unsigned> ho((size_t)256 * 256 * NP);
    hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    maxd = 0.0;
    for (size_t th = 0; th < 256 * 256; th += 7)
      for (int k = 0; k < NP; ++k)
        for (int hh = 0; hh < 2; ++hh) {
          const float z8 = hx[(th * 8 + ((k >> 3) & 7)) * 16 + 2 * (k & 7) + hh] * 0.125f;
          const unsigned pk = ho[(th * (NP / 4) + (k >> 2)) * 4 + (k & 3)];
          const float got = bf2f((unsigned short)(hh ? pk >> 16 : pk & 0xffffu));
          double zz = (MODE == 1 ? (double)bf2f(f2bf(z8)) : (double)z8) * 8.0;
          if (MODE == 1) { const double a = fmin(fmax(fabs(zz), ldexp(1.0, -10)), 8.0 * (1.0 - 1.0 / 256)); zz = zz < 0 ? -a : a; }
          const float want = bf2f(f2bf((float)(gelu(zz) / 8.0)));
          const double d = fabs((double)got - (double)want) * 8.0;
          if (d > maxd) maxd = d;
        }
  }
  printf("%-22s NP=%2d REPL=%d dist=%d: %8.1f cycles / segment (wave 0), %8.1f ns / segment (event), max |d gelu| = %.2e\n", name, NP, REPL, dist, (double)h / iters, ms * 1e6 / iters, maxd);
}

int main() {
  const size_t nx = (size_t)256 * 256 * 8 * 16;
  std::vector<float> hx(nx);
  std::vector<unsigned short> htab(2 * NENT);
  for (int i = -NENT; i < NENT; ++i) {                        // entry i: code = LO_CODE + a, a = i (i >= 0) or -1 - i, value / 8 domain
    const int a = i >= 0 ? i : -1 - i;
    const double z = bf2f((unsigned short)(LO_CODE + a)) * 8.0 * (i >= 0 ? 1.0 : -1.0);
    htab[i + NENT] = f2bf((float)(gelu(z) / 8.0));
  }
  float* xin; unsigned short* tab; unsigned* out; long long* clk;
  hipMalloc(&xin, nx * 4); hipMalloc(&tab, htab.size() * 2); hipMalloc(&out, (size_t)256 * 256 * 64 * 4); hipMalloc(&clk, 8);
  hipMemcpy(tab, htab.data(), htab.size() * 2, hipMemcpyHostToDevice);
  for (int dist = 0; dist < 3; ++dist) {
    srand(1234);
    for (size_t i = 0; i < nx; ++i) {
      const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
      if (dist == 0) hx[i] = (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
      else if (dist == 1) hx[i] = 0.7f;
      else hx[i] = (float)((u2 < 0.5 ? -1.0 : 1.0) * exp2(-10.0 + 13.0 * u1));
    }
    hipMemcpy(xin, hx.data(), nx * 4, hipMemcpyHostToDevice);
    if (dist == 0) {
      run<2, 16, 1>("no GELU (floor)", dist, xin, tab, out, clk, hx);
      run<2, 16, 1, 2>("floor, 2 accumulators", dist, xin, tab, out, clk, hx);      // (round 6: is a chain of MFMAs on ONE accumulator paced by its latency?)
      run<2, 16, 1, 1>("floor, 1 accumulator", dist, xin, tab, out, clk, hx);
      run<1, 16, 1, 2>("LDS table, 2 acc", dist, xin, tab, out, clk, hx);
      run<1, 16, 1, 1>("LDS table, 1 acc", dist, xin, tab, out, clk, hx);
      run<3, 16, 1>("table VALU, no gather", dist, xin, tab, out, clk, hx);
      run<3, 64, 1>("table VALU, no gather", dist, xin, tab, out, clk, hx);
    }
    run<0, 16, 1>("VALU gelu_sig", dist, xin, tab, out, clk, hx);
    run<1, 16, 1>("LDS table", dist, xin, tab, out, clk, hx);
    run<1, 16, 4>("LDS table", dist, xin, tab, out, clk, hx);
    run<0, 64, 1>("VALU gelu_sig", dist, xin, tab, out, clk, hx);
    run<1, 64, 1>("LDS table", dist, xin, tab, out, clk, hx);
    run<1, 64, 4>("LDS table", dist, xin, tab, out, clk, hx);
  }
  return 0;
}
