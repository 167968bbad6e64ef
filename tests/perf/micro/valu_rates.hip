// Issue cost of single VALU instructions on gfx950 with 8 waves per SIMD (the walk kernel's residency):
// every wave runs the same unrolled block of 32 independent copies of one instruction; the cost per
// instruction and SIMD is (kernel time x clock) / (instructions per wave x 8).  Measurement helper of
// DESIGN.md section 3 (which instructions of the walk loops are not "4 cycles per wave64").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))

template <int OP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) k(double *out, int iters, int sidx)
{
  double a = threadIdx.x * 1.0 + 0.25, b = 1.0000001, c = 0.5;
  int i0 = threadIdx.x, i1 = 3;
  unsigned long long m = 0x5555555555555555ull;
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(i0), "+v"(i1), "+s"(m), "+s"(sidx));
  for (int it = 0; it < iters; ++it)
  {
    if (OP == 0) { REP32(asm volatile("v_add_f64 %0, %1, %2" : "=v"(c) : "v"(a), "v"(b));) }
    if (OP == 1) { REP32(asm volatile("v_mul_f64 %0, %1, %2" : "=v"(c) : "v"(a), "v"(b));) }
    if (OP == 2) { REP32(asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(c) : "v"(i0));) }
    if (OP == 3) { REP32(asm volatile("v_rndne_f64 %0, %1" : "=v"(c) : "v"(a));) }
    if (OP == 4) { { int si; REP32(asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(si) : "v"(i0), "s"(sidx));) asm volatile("" :: "s"(si)); } }
    if (OP == 5) { REP32(asm volatile("s_mov_b32 m0, %1\n v_writelane_b32 %0, %1, m0" : "+v"(i0) : "s"(sidx) : "m0");) }
    if (OP == 6) { REP32(asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(i1) : "v"(i0), "v"(i0), "s"(m));) }
    if (OP == 7) { REP32(asm volatile("v_cmp_le_f64 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");) }
    if (OP == 8) { REP32(asm volatile("v_mov_b64 %0, %1" : "=v"(c) : "v"(a));) }
    if (OP == 9) { REP32(asm volatile("v_add_u32 %0, %1, %2" : "=v"(i1) : "v"(i0), "v"(i0));) }
    if (OP == 10) { REP32(asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(c) : "v"(a), "v"(b), "v"(a));) }
    if (OP == 11) { REP32(asm volatile("s_add_i32 %0, %0, 1" : "+s"(sidx) : : "scc");) }
    if (OP == 12) { REP32(asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(c) : "s"(sidx));) }
    if (OP == 13) { REP32(asm volatile("v_rndne_f64 %0, %1" : "=v"(c) : "s"(m));) }
  }
  asm volatile("" : "+v"(c), "+v"(i0), "+v"(i1));
  if (c == 123.456 && i0 == 77 && i1 == 99)
    out[0] = c;
}

template <int OP>
static void run(const char *name, double *d, double ghz)
{
  const int iters = 4096, blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 64, 5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 5);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_wave = (double)iters * 32.0;
  const double cyc      = ms * 1e-3 * ghz * 1e9 / (per_wave * 8.0);
  printf("%-28s %8.3f ms   %6.2f cycles per instruction and SIMD (at %.2f GHz)\n", name, ms, cyc, ghz);
}

int main(int argc, char **argv)
{
  const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
  double *d;
  hipMalloc(&d, 64);
  run<0>("v_add_f64", d, ghz);
  run<1>("v_mul_f64", d, ghz);
  run<10>("v_fma_f64", d, ghz);
  run<2>("v_cvt_f64_i32 (vgpr src)", d, ghz);
  run<12>("v_cvt_f64_i32 (sgpr src)", d, ghz);
  run<3>("v_rndne_f64 (vgpr src)", d, ghz);
  run<13>("v_rndne_f64 (sgpr src)", d, ghz);
  run<4>("v_readlane_b32", d, ghz);
  run<5>("s_mov m0 + v_writelane_b32", d, ghz);
  run<6>("v_cndmask_b32_e64 (sgpr mask)", d, ghz);
  run<7>("v_cmp_le_f64", d, ghz);
  run<8>("v_mov_b64", d, ghz);
  run<9>("v_add_u32", d, ghz);
  run<11>("s_add_i32", d, ghz);
  return 0;
}
