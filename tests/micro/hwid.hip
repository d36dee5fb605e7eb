// GPU-box microbenchmark: is (XCC_ID, HW_ID without the pipe bits) a unique index of a resident wave slot on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(64) k_hwid(int *occ, int *maxocc, unsigned *seen_bits, int spin, int lds_bytes)
{
  extern __shared__ int lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  const unsigned idx = ((hw & 0x3Fu) | ((hw >> 2) & 0x3FC0u)) & 0x3FFFu;   // wave_id[3:0] simd[5:4] | cu[11:8] sh[12] se[15:13] shifted over the pipe bits
  if (threadIdx.x == 0) {
    const int now = atomicAdd(&occ[xcc * 16384 + idx], 1) + 1;
    atomicMax(&maxocc[xcc * 16384 + idx], now);
    atomicOr(&seen_bits[0], hw);
    lds[0] = now;
  }
  __syncthreads();
  int v = lds[0];
  for (int i = 0; i < spin; ++i) { v = v * 1664525 + 1013904223; __builtin_amdgcn_s_sleep(4); }
  if (v == 0x12345) lds[1] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicSub(&occ[xcc * 16384 + idx], 1);
}
int main()
{
  int *occ, *mx; unsigned *seen;
  hipMalloc(&occ, 8 * 16384 * 4); hipMalloc(&mx, 8 * 16384 * 4); hipMalloc(&seen, 4);
  hipMemset(occ, 0, 8 * 16384 * 4); hipMemset(mx, 0, 8 * 16384 * 4); hipMemset(seen, 0, 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_hwid), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  for (int lds : {1024, 10240, 20480})
    hipLaunchKernelGGL(k_hwid, dim3(200000), dim3(64), lds, 0, occ, mx, seen, 200, lds);
  hipDeviceSynchronize();
  std::vector<int> h(8 * 16384); unsigned hs;
  hipMemcpy(h.data(), mx, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(&hs, seen, 4, hipMemcpyDeviceToHost);
  int used = 0, worst = 0, maxidx = 0; int per_xcc[8] = {0};
  for (size_t i = 0; i < h.size(); ++i) if (h[i]) { ++used; worst = h[i] > worst ? h[i] : worst; per_xcc[i / 16384]++; if ((int)(i % 16384) > maxidx) maxidx = (int)(i % 16384); }
  std::printf("slots used %d (per XCC: %d %d %d %d %d %d %d %d), largest index %d, most waves ever on one index %d, OR of HW_ID 0x%08x\n",
              used, per_xcc[0], per_xcc[1], per_xcc[2], per_xcc[3], per_xcc[4], per_xcc[5], per_xcc[6], per_xcc[7], maxidx, worst, hs);
  return 0;
}
