// GPU-box microbenchmark, second sheet: issue interval per wave64 instruction and SIMD for the VOP2 / VOP3 / VOP3P
// kinds a rewrite of k_poa's row could be made of, at 4 waves per SIMD (k_poa's occupancy), and for two mixed
// streams.  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP16(x) x x x x x x x x x x x x x x x x
// eight independent chains of one instruction; T = operand tail behind "dst, dst"
#define I8(OP, T)                                                                                              \
  OP " %0, %0" T "\n" OP " %1, %1" T "\n" OP " %2, %2" T "\n" OP " %3, %3" T "\n" OP " %4, %4" T "\n" OP " %5, %5" T \
     "\n" OP " %6, %6" T "\n" OP " %7, %7" T
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sb)

#define KERNEL(NAME, BODY)                                                                                     \
  __global__ void __launch_bounds__(64) NAME(unsigned *out, int iters)                                         \
  {                                                                                                            \
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,    \
             a7 = a0 + 7;                                                                                      \
    unsigned b = threadIdx.x | 0x00010001u;                                                                    \
    unsigned sb = blockIdx.x | 0x00010001u;                                                                    \
    for (int i = 0; i < iters; ++i) { REP16(asm volatile(BODY OPS);) }                                         \
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                \
  }

KERNEL(k_add, I8("v_add_u32", ", %8"))
KERNEL(k_sub, I8("v_sub_u32", ", %8"))
KERNEL(k_sub_s, "v_sub_u32 %0, %9, %0\n v_sub_u32 %1, %9, %1\n v_sub_u32 %2, %9, %2\n v_sub_u32 %3, %9, %3\n v_sub_u32 %4, %9, %4\n v_sub_u32 %5, %9, %5\n v_sub_u32 %6, %9, %6\n v_sub_u32 %7, %9, %7")
KERNEL(k_subrev_s, "v_subrev_u32 %0, %9, %0\n v_subrev_u32 %1, %9, %1\n v_subrev_u32 %2, %9, %2\n v_subrev_u32 %3, %9, %3\n v_subrev_u32 %4, %9, %4\n v_subrev_u32 %5, %9, %5\n v_subrev_u32 %6, %9, %6\n v_subrev_u32 %7, %9, %7")
KERNEL(k_max32, I8("v_max_i32", ", %8"))
KERNEL(k_minu32, I8("v_min_u32", ", %8"))
KERNEL(k_xor, I8("v_xor_b32", ", %8"))
KERNEL(k_and, I8("v_and_b32", ", %8"))
KERNEL(k_or, I8("v_or_b32", ", %8"))
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7")
KERNEL(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0")
KERNEL(k_cndmask, I8("v_cndmask_b32", ", %8, vcc"))
KERNEL(k_max16, I8("v_max_i16", ", %8"))
KERNEL(k_sub16, I8("v_sub_u16", ", %8"))
KERNEL(k_mul24, I8("v_mul_u32_u24", ", %8"))
KERNEL(k_lshl_or, I8("v_lshl_or_b32", ", 1, %8"))
KERNEL(k_and_or, I8("v_and_or_b32", ", %8, %8"))
KERNEL(k_lshl_add, I8("v_lshl_add_u32", ", 1, %8"))
KERNEL(k_add3, I8("v_add3_u32", ", %8, %8"))
KERNEL(k_max3, I8("v_max3_i32", ", %8, %8"))
KERNEL(k_med3, I8("v_med3_i32", ", %8, %8"))
KERNEL(k_perm, I8("v_perm_b32", ", %8, %8"))
KERNEL(k_mad24, I8("v_mad_u32_u24", ", %8, %8"))
KERNEL(k_pk_min_u16, I8("v_pk_min_u16", ", %8"))
KERNEL(k_pk_min_u16_s, I8("v_pk_min_u16", ", %9"))
KERNEL(k_pk_max_i16, I8("v_pk_max_i16", ", %8"))
KERNEL(k_pk_add_u16, I8("v_pk_add_u16", ", %8"))
KERNEL(k_pk_lshr, "v_pk_lshrrev_b16 %0, 1, %0\n v_pk_lshrrev_b16 %1, 1, %1\n v_pk_lshrrev_b16 %2, 1, %2\n v_pk_lshrrev_b16 %3, 1, %3\n v_pk_lshrrev_b16 %4, 1, %4\n v_pk_lshrrev_b16 %5, 1, %5\n v_pk_lshrrev_b16 %6, 1, %6\n v_pk_lshrrev_b16 %7, 1, %7")
KERNEL(k_pk_mad, I8("v_pk_mad_i16", ", %9, %8"))
KERNEL(k_bfi, "v_bfi_b32 %0, %8, %0, %8\n v_bfi_b32 %1, %8, %1, %8\n v_bfi_b32 %2, %8, %2, %8\n v_bfi_b32 %3, %8, %3, %8\n v_bfi_b32 %4, %8, %4, %8\n v_bfi_b32 %5, %8, %5, %8\n v_bfi_b32 %6, %8, %6, %8\n v_bfi_b32 %7, %8, %7, %8")
KERNEL(k_dpp_mov, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_dpp_add, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %4, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %6, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %0, %7 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_sdwa_max, "v_max_i16_sdwa %0, %0, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n v_max_i16_sdwa %1, %1, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n v_max_i16_sdwa %2, %2, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n v_max_i16_sdwa %3, %3, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n v_max_i16_sdwa %4, %4, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n v_max_i16_sdwa %5, %5, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n v_max_i16_sdwa %6, %6, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n v_max_i16_sdwa %7, %7, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1")
// mixed streams: packed and plain alternating (4 + 4 of the 8 chains)
KERNEL(k_mix_pk_add, "v_pk_max_i16 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_pk_max_i16 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_add_u32 %7, %7, %8")
// ... and with every instruction reading the result of the one before it (the row of k_poa is such a chain)
KERNEL(k_chain_pk, "v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %0, %0, %8")
KERNEL(k_chain_add, "v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8")
KERNEL(k_chain_mix, "v_pk_max_i16 %0, %0, %8\n v_sub_u32 %0, %0, %8\n v_pk_max_i16 %0, %0, %8\n v_sub_u32 %0, %0, %8\n v_pk_max_i16 %0, %0, %8\n v_sub_u32 %0, %0, %8\n v_pk_max_i16 %0, %0, %8\n v_sub_u32 %0, %0, %8")

typedef void (*kern_t)(unsigned *, int);

static void run(const char *name, kern_t k, unsigned *d_out)
{
  const int iters = 2000;                       // x 128 instructions
  for (int wps : {1, 4, 8}) {
    const int blocks = 256 * 4 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d_out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * iters * 128.0;
    std::printf("%-16s waves/SIMD %d: %7.3f ms, %6.1f G wave-insts/s chip, issue interval %.2f ns per inst per SIMD\n", name, wps, ms,
                insts / ms / 1e6, ms * 1e6 / (insts / 1024.0));
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
}

int main()
{
  unsigned *d_out;
  hipMalloc(&d_out, 256 * 4 * 8 * 64 * 4);
#define RUN(k) run(#k, k, d_out)
  RUN(k_add); RUN(k_sub); RUN(k_sub_s); RUN(k_subrev_s); RUN(k_max32); RUN(k_minu32); RUN(k_xor); RUN(k_and); RUN(k_or);
  RUN(k_lshl); RUN(k_mov); RUN(k_cndmask); RUN(k_max16); RUN(k_sub16); RUN(k_mul24);
  RUN(k_lshl_or); RUN(k_and_or); RUN(k_lshl_add); RUN(k_add3); RUN(k_max3); RUN(k_med3); RUN(k_perm); RUN(k_mad24);
  RUN(k_pk_min_u16); RUN(k_pk_min_u16_s); RUN(k_pk_max_i16); RUN(k_pk_add_u16); RUN(k_pk_lshr); RUN(k_pk_mad); RUN(k_bfi);
  RUN(k_dpp_mov); RUN(k_dpp_add); RUN(k_sdwa_max);
  RUN(k_mix_pk_add); RUN(k_chain_pk); RUN(k_chain_add); RUN(k_chain_mix);
  return 0;
}
