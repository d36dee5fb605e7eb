// elector_amd/csrc/poa_classify.hip -- the per-window bookkeeping of a batch on the device: window status, launch
// class (geometry class x LDS slot tier), size key, the per-class maxima the launches are sized from, and the class
// lists in processing order (largest window first).
//
// Up to round 3 the host walked all 1.4 M windows of a batch for this (class search, counting sort: 5.5 ms of a
// 6.4 ms step on sixteen threads) before the first launch could be queued.  Now the host reads back 29 KB of per-class
// totals, decides which classes become launches (poa_host.hip) and hands a class -> list table to the sort.
//
// What the reference does at this point: nothing -- poa takes the windows in file order, one at a time
// (src/poa-graph/main.c:265-284).  The classes exist because a wavefront is fastest when its windows look alike.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "elector_poa.h"
#include "poa_classes.h"

namespace elector {

constexpr int kClsChunk = 2048;         // windows per block

__global__ void __launch_bounds__(256) k_classify(ClassifyArgs a)
{
  __shared__ int32_t acc[kAccRows * kBins];
  __shared__ unsigned long long s_left;
  __shared__ int s_gen, s_po, s_bad;
  for (int i = threadIdx.x; i < kAccRows * kBins; i += 256) acc[i] = 0;
  if (threadIdx.x == 0) { s_left = 0; s_gen = 0; s_po = 0; s_bad = 0; }
  __syncthreads();
  const int64_t w0 = (int64_t)blockIdx.x * kClsChunk;
  unsigned long long left = 0;
  int gen = 0, po = 0, bad = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0 && (a.off[0] != 0 || a.off[3 * a.n] != a.total)) bad = 1;
  // the thread's eight windows: their offsets are all asked for before any is looked at (the loop below was eight
  // dependent trips to memory otherwise)
  constexpr int PER = kClsChunk / 256;
  int64_t o[PER][4];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int64_t w = min(w0 + threadIdx.x + 256 * j, a.n - 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) o[j][q] = a.off[3 * w + q];
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int64_t w = w0 + threadIdx.x + 256 * j;
    if (w >= a.n) break;
    const int64_t o0 = o[j][0], o1 = o[j][1], o2 = o[j][2], o3 = o[j][3];
    const int64_t lr = o1 - o0, lc = o2 - o1, lu = o3 - o2;
    if (lr < 0 || lc < 0 || lu < 0) { bad = 1; a.status[w] = ELECTOR_W_TOOLONG; a.bin[w] = -1; a.wkey[w] = 0; continue; }
    const int st = window_status(lr, lc, lu, a.pen_abs_max, ELECTOR_MAX_SEQ, a.window_moves_max);
    a.status[w] = st;
    WindowClass wc{-1, 0, 0, 0};
    if (!st && a.use_fused) wc = window_class(a.kp, lr, lc, lu, a.force_cls);
    a.bin[w] = (int16_t)wc.bin;
    a.wkey[w] = (uint8_t)window_size_key(lr, lu, a.coarse != 0);
    // Neighbouring windows mostly share a bin: the wavefront's lanes that hold one bin count themselves with ONE
    // atomic (the bin's first lane adds the population count), and a maximum is only sent when it beats what the
    // block already holds -- after the first few windows hardly ever.  (One atomic per window and field on the same
    // few LDS words took 0.7 ms per 1.8 M windows.)
    {
      int mine = wc.bin;
      unsigned long long todo = __ballot(mine >= 0);
      while (todo) {
        const int lead = __builtin_ctzll(todo);
        const int b = __shfl(mine, lead);
        const unsigned long long same = __ballot(mine == b);
        if ((int)(threadIdx.x & 63) == lead) atomicAdd(&acc[b], (int)__popcll(same));
        todo &= ~same;
      }
    }
    if (wc.bin >= 0) {
      auto raise = [&](int row, int v) { if (v > acc[row * kBins + wc.bin]) atomicMax(&acc[row * kBins + wc.bin], v); };
      raise(1, wc.need_a);
      raise(2, (int)lr);
      raise(3, (int)lc);
      raise(4, (int)lu);
      raise(5, (int)(lr + lc));
      raise(6, wc.need_pack);
      raise(7, wc.need_triv);
      left += (unsigned long long)((int64_t)n_strips((int)lu) * mv_tw((int)(lr + lc)) * 64);
    } else ++gen;
    if (!st) po = max(po, (int)(lr + lc));
  }
  if (left) atomicAdd(&s_left, left);
  if (gen) atomicAdd(&s_gen, gen);
  if (po) atomicMax(&s_po, po);
  if (bad) s_bad = 1;
  __syncthreads();
  for (int i = threadIdx.x; i < kAccRows * kBins; i += 256) {
    const int32_t v = acc[i];
    if (!v) continue;
    if (i < kBins) atomicAdd(a.acc + i, v); else atomicMax(a.acc + i, v);
  }
  if (threadIdx.x == 0) {
    if (s_gen) atomicAdd(a.glob, (unsigned long long)s_gen);
    if (s_left) atomicAdd(a.glob + 1, s_left);
    if (s_po) atomicMax(a.glob + 2, (unsigned long long)s_po);
    if (s_bad) atomicMax(a.glob + 3, 1ull);
  }
}

// ---- counting sort of the windows by (list, size key) ----
__global__ void __launch_bounds__(256) k_sort_hist(SortArgs a)
{
  extern __shared__ uint32_t cnt[];
  const int nb = a.ndest * kKeys;
  for (int i = threadIdx.x; i < nb; i += 256) cnt[i] = 0;
  __syncthreads();
  const int64_t w0 = (int64_t)blockIdx.x * kClsChunk;
  for (int k = threadIdx.x; k < kClsChunk; k += 256) {
    const int64_t w = w0 + k;
    if (w >= a.n) break;
    const int b = a.bin[w];
    atomicAdd(&cnt[(int)a.dest_of[b >= 0 ? b : kBins] * kKeys + a.wkey[w]], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += 256)
    if (cnt[i]) atomicAdd(a.hist + i, cnt[i]);
}

// counts -> first position of every (list, key) bucket, relative to the list's array (one block)
__global__ void __launch_bounds__(256) k_sort_scan(SortArgs a)
{
  __shared__ uint32_t part[256];
  for (int d = 0; d < a.ndest; ++d) {
    const uint32_t v = a.hist[d * kKeys + threadIdx.x];
    part[threadIdx.x] = v;
    __syncthreads();
    for (int s = 1; s < 256; s <<= 1) {
      const uint32_t t = threadIdx.x >= (unsigned)s ? part[threadIdx.x - s] : 0u;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    a.hist[d * kKeys + threadIdx.x] = (uint32_t)a.dest_first[d] + part[threadIdx.x] - v;
    __syncthreads();
  }
}

// A block reserves a range of every bucket its windows fall into and places them there: the windows of a bucket keep
// their order inside a block's share (neighbours in the batch stay neighbours in the list: k_gather's reads stay
// local), the blocks' shares follow in the order the reservations happen to come.  Every window of a bucket has the
// same reference length, so the order inside one is of no consequence for the kernels.
__global__ void __launch_bounds__(256) k_sort_scatter(SortArgs a)
{
  extern __shared__ uint32_t sm[];
  const int nb = a.ndest * kKeys;
  uint32_t *cnt = sm, *base = sm + nb;
  for (int i = threadIdx.x; i < nb; i += 256) cnt[i] = 0;
  __syncthreads();
  const int64_t w0 = (int64_t)blockIdx.x * kClsChunk;
  constexpr int PER = kClsChunk / 256;
  int bucket[PER];
  uint32_t rank[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int64_t w = w0 + (int64_t)threadIdx.x * PER + j;       // a thread's windows are consecutive
    bucket[j] = -1;
    if (w < a.n) {
      const int b = a.bin[w];
      bucket[j] = (int)a.dest_of[b >= 0 ? b : kBins] * kKeys + a.wkey[w];
    }
  }
  // ranks inside the block in window order: thread by thread (consecutive windows), so two passes -- counts per
  // thread group are not needed, an LDS atomic per window hands out the ranks in an order that is stable per thread
  // and arbitrary between threads; a bucket's windows of one block stay within the block's share either way
#pragma unroll
  for (int j = 0; j < PER; ++j)
    if (bucket[j] >= 0) rank[j] = atomicAdd(&cnt[bucket[j]], 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += 256)
    if (cnt[i]) base[i] = atomicAdd(a.hist + i, cnt[i]);
  __syncthreads();
  const int gen_first = (a.ndest - 1) * kKeys;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    if (bucket[j] < 0) continue;
    const uint32_t w = (uint32_t)(w0 + (int64_t)threadIdx.x * PER + j);
    const uint32_t pos = base[bucket[j]] + rank[j];
    if (bucket[j] >= gen_first) a.generic[pos] = w; else a.lists[pos] = w;
  }
}

// ---- the generic list's windows for the host (few): lengths and status, list order ----
__global__ void __launch_bounds__(256) k_generic_info(const uint32_t *glist, int64_t ng, const int64_t *off, const int32_t *status,
                                                      int32_t *info)
{
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= ng) return;
  const int64_t w = glist[k];
  info[4 * k] = (int32_t)(off[3 * w + 1] - off[3 * w]);
  info[4 * k + 1] = (int32_t)(off[3 * w + 2] - off[3 * w + 1]);
  info[4 * k + 2] = (int32_t)(off[3 * w + 3] - off[3 * w + 2]);
  info[4 * k + 3] = status[w];
}

// ... and their moves offsets back into the per-window arrays
__global__ void __launch_bounds__(256) k_generic_moves(const uint32_t *glist, int64_t ng, const int64_t *gmv, int64_t *mv1, int64_t *mv2)
{
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= ng) return;
  const int64_t w = glist[k];
  mv1[w] = gmv[2 * k];
  mv2[w] = gmv[2 * k + 1];
}

// Small results on their way to the host WITHOUT the copy engine: a kernel stores them into page-locked host memory the
// device can address.  The copy engine is busy with the merged rows of earlier batches (hundreds of megabytes each):
// a 29 KB read-back queued behind one of those made the host wait milliseconds for totals it needs before it can
// queue the next batch.  (dst and src 4-byte aligned, bytes a multiple of 4.)
__global__ void __launch_bounds__(256) k_words_to_host(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, int64_t n)
{
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

int launch_words_to_host(void *host_dst, const void *src, size_t bytes, hipStream_t st)
{
  if (!bytes) return 0;
  void *d = nullptr;
  if (hipHostGetDevicePointer(&d, host_dst, 0) != hipSuccess || !d) { (void)hipGetLastError(); return -1; }
  const int64_t n = (int64_t)(bytes + 3) / 4;
  const unsigned blocks = (unsigned)std::min<int64_t>(1024, (n + 255) / 256);
  hipLaunchKernelGGL(k_words_to_host, dim3(blocks), dim3(256), 0, st, reinterpret_cast<uint32_t *>(d),
                     reinterpret_cast<const uint32_t *>(src), n);
  return 0;
}

void launch_classify(const ClassifyArgs &a, hipStream_t st)
{
  if (a.n <= 0) return;
  hipLaunchKernelGGL(k_classify, dim3((unsigned)((a.n + kClsChunk - 1) / kClsChunk)), dim3(256), 0, st, a);
}

int launch_sort(const SortArgs &a, hipStream_t st)
{
  if (a.n <= 0) return 0;
  if (a.ndest < 1 || a.ndest > kSortDestMax) return -1;
  const unsigned blocks = (unsigned)((a.n + kClsChunk - 1) / kClsChunk);
  const size_t lds = (size_t)a.ndest * kKeys * 4;
  static DeviceOnce once;
  if (once.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_sort_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * kSortDestMax * kKeys * 4) != hipSuccess)
      return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_sort_hist), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kSortDestMax * kKeys * 4) != hipSuccess)
      return -1;
    once.done();
  }
  hipLaunchKernelGGL(k_sort_hist, dim3(blocks), dim3(256), lds, st, a);
  hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_sort_scatter, dim3(blocks), dim3(256), 2 * lds, st, a);
  return 0;
}

void launch_generic_info(const uint32_t *glist, int64_t ng, const int64_t *off, const int32_t *status, int32_t *info, hipStream_t st)
{
  if (ng <= 0) return;
  hipLaunchKernelGGL(k_generic_info, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, st, glist, ng, off, status, info);
}

void launch_generic_moves(const uint32_t *glist, int64_t ng, const int64_t *gmv, int64_t *mv1, int64_t *mv2, hipStream_t st)
{
  if (ng <= 0) return;
  hipLaunchKernelGGL(k_generic_moves, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, st, glist, ng, gmv, mv1, mv2);
}

}  // namespace elector
