// GPU-box microbenchmark: how many wave64 integer VALU instructions a gfx950 SIMD issues per second, with
// 1 .. 8 waves per SIMD, for the instruction kinds k_poa is made of.  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void __launch_bounds__(64) k_rate(unsigned *out, int iters, unsigned long long *ticks)
{
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned b = blockIdx.x | 0x00010001u;
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) { REP16(asm volatile("v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (KIND == 1) { REP16(asm volatile("v_pk_mad_i16 %0, %0, %8, %8\n v_pk_mad_i16 %1, %1, %8, %8\n v_pk_mad_i16 %2, %2, %8, %8\n v_pk_mad_i16 %3, %3, %8, %8\n v_pk_mad_i16 %4, %4, %8, %8\n v_pk_mad_i16 %5, %5, %8, %8\n v_pk_mad_i16 %6, %6, %8, %8\n v_pk_mad_i16 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (KIND == 2) { REP16(asm volatile("v_bfi_b32 %0, %8, %0, %8\n v_bfi_b32 %1, %8, %1, %8\n v_bfi_b32 %2, %8, %2, %8\n v_bfi_b32 %3, %8, %3, %8\n v_bfi_b32 %4, %8, %4, %8\n v_bfi_b32 %5, %8, %5, %8\n v_bfi_b32 %6, %8, %6, %8\n v_bfi_b32 %7, %8, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (KIND == 3) { REP16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (KIND == 4) { REP16(asm volatile("v_pk_sub_i16 %0, %0, %8\n v_pk_sub_i16 %1, %1, %8\n v_pk_sub_i16 %2, %2, %8\n v_pk_sub_i16 %3, %3, %8\n v_pk_sub_i16 %4, %4, %8\n v_pk_sub_i16 %5, %5, %8\n v_pk_sub_i16 %6, %6, %8\n v_pk_sub_i16 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (KIND == 5) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (KIND == 6) { REP16(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long w1 = wall_clock64();
  out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  if (blockIdx.x == 0 && threadIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = w1 - w0; }
}

template <int KIND>
static void run(const char *name, unsigned *d_out, unsigned long long *d_ticks)
{
  const int iters = 2000;                       // x 128 instructions
  for (int wps : {1, 2, 3, 4, 8}) {
    const int blocks = 256 * 4 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, 10, d_ticks);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, iters, d_ticks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long tk[2];
    hipMemcpy(tk, d_ticks, sizeof tk, hipMemcpyDeviceToHost);
    const double insts = (double)blocks * iters * 128.0;
    std::printf("%-14s waves/SIMD %d: %.3f ms, %.1f G wave-insts/s chip, %.2f insts/ns/SIMD; block 0: %llu s_memtime ticks, %llu wall_clock64 ticks (100 MHz) -> s_memtime %.0f MHz; issue interval %.2f ns per inst per SIMD\n",
                name, wps, ms, insts / ms / 1e6, insts / 1024.0 / (ms * 1e6), tk[0], tk[1], (double)tk[0] / ((double)tk[1] / 100.0),
                ms * 1e6 / (insts / 1024.0));
  }
}

int main()
{
  unsigned *d_out; unsigned long long *d_ticks;
  hipMalloc(&d_out, 256 * 4 * 8 * 64 * 4);
  hipMalloc(&d_ticks, 16);
  run<3>("v_add_u32", d_out, d_ticks);
  run<0>("v_pk_max_i16", d_out, d_ticks);
  run<4>("v_pk_sub_i16", d_out, d_ticks);
  run<1>("v_pk_mad_i16", d_out, d_ticks);
  run<2>("v_bfi_b32", d_out, d_ticks);
  run<6>("v_mov_dpp", d_out, d_ticks);
  run<5>("v_fma_f32", d_out, d_ticks);
  return 0;
}
