// elector_amd/csrc/stats.hip -- the second half of the MSA stage on the device:
// merging the windows of a read into one record (k_merge, SURVEY.md section 8(f)
// row 2) and the per-read MSA statistics (k_stats, section 8 row a14).
//
// k_merge: one 256-thread block per PIECE (one 6-line record of msa.fa: all
// windows of a read, or of one part of a split read).  It replaces `Donatello`
// (reference: src/split/Donatello.cpp:13-31 clean_msa, :48-93 concatenation): the
// windows' column-interleaved MSAs (the output layout of elector_poa_batch_device)
// are concatenated into three rows, dropping the columns whose corrected letter is
// 'n'.  Pure byte movement: 3 bytes per column in, 3 out.
//
// k_stats: one 256-thread block per READ walks the read's pieces and produces the
// integer counters of include/elector_stats.h.  It replaces the per-column Python
// loops of the reference (elector/computeStats.py:61-189, 291-328, 371-440,
// 472-498, 712-752).  One pass over the letters turns the piece into six bit rows
// in LDS (gap in reference / corrected / uncorrected, letters pairwise equal: one
// ballot per 64 columns); the end-gap scans, the search for runs of corrected gaps,
// the interval list logic of findGapStretches (one entry per run of >= 5 corrected
// gaps, on single lanes), the mask and the masked counters (population counts of
// word combinations) work on those words.  Pieces beyond the LDS budget keep
// reading the letters from HBM (the first form of this kernel).  Latency bound.
// Floats never appear here.
//
// elector_homopolymer_pairs: host-side integer state machine for the one read
// whose homopolymer ratio the reference reports (computeStats.py:671-674).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <cstdio>
#include <thread>
#include <vector>
#include <unistd.h>

#include "ctx.h"
#include "poa_classes.h"
#include "elector_stats.h"
#include "elector_split.h"

namespace elector {

constexpr int kThresh = 5;     // THRESH  computeStats.py:40
constexpr int kThresh2 = 20;   // THRESH2 computeStats.py:41
constexpr int kStatsThreads = 256;
constexpr int kStatsThreadsMax = 1024;  // k_stats on batches of long reads: a block per read of tens of thousands of columns

// --------------------------------------------------------------------- merge ---

struct MergeArgs {
  int64_t n_pieces;
  const int64_t *piece_first;   // n_pieces + 1 window ids
  const int64_t *off;           // [3n+1] window offsets of the POA batch
  const uint8_t *cols_in;       // window w: 3 * ncol[w] bytes at 3 * off[3w], column-interleaved
  const int32_t *ncol;
  const int32_t *status;
  uint8_t *rows;                // out: piece p's three rows at rows + row_off[p]
  int64_t *row_off;             // out: 3 * off[3 * piece_first[p]]  (room for every column of its windows)
  int64_t *cols;                // out: surviving columns per piece
  int32_t *woff;                // scratch, one per window
};

constexpr int kMergeWin = 1024;          // windows of a piece whose descriptors and offsets stay in LDS

__global__ void __launch_bounds__(kStatsThreads) k_merge(MergeArgs a)
{
  __shared__ int s_wave[kStatsThreads / 64];
  __shared__ int64_t s_carry;
  __shared__ int s_nc[kMergeWin], s_wo[kMergeWin];
  __shared__ int64_t s_src[kMergeWin];
  const int64_t p = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t w0 = a.piece_first[p], W = a.piece_first[p + 1] - w0;
  // The kernel is bound by the latency of its dependent loads (window descriptor -> columns).  The descriptors
  // of all windows are fetched first, one window per thread; then every wave works on FOUR windows at a time,
  // 16 lanes each, with the first 64 columns (nearly always all of them) of TWO windows per lane group in
  // flight before it looks at any.
  for (int64_t i = tid; i < W && i < kMergeWin; i += kStatsThreads) {
    const int64_t w = w0 + i;
    s_nc[i] = a.status[w] == 0 ? a.ncol[w] : 0;
    s_src[i] = 3 * a.off[3 * w];
  }
  if (tid == 0) s_carry = 0;
  __syncthreads();
  auto win_nc = [&](int64_t i) { return i < kMergeWin ? s_nc[i] : (a.status[w0 + i] == 0 ? a.ncol[w0 + i] : 0); };
  auto win_src = [&](int64_t i) { return a.cols_in + (i < kMergeWin ? s_src[i] : 3 * a.off[3 * (w0 + i)]); };
  auto get_wo = [&](int64_t i) { return i < kMergeWin ? s_wo[i] : a.woff[w0 + i]; };
  auto set_wo = [&](int64_t i, int v) { if (i < kMergeWin) s_wo[i] = v; else a.woff[w0 + i] = v; };
  constexpr int kGroups = 4 * (kStatsThreads / 64);            // windows in flight per block and turn
  const int grp = tid >> 4, gl = tid & 15, gshift = (lane >> 4) * 16;
  // surviving columns per window (Donatello.cpp:13-31)
  auto load_y = [&](const uint8_t *src, int nc, int c0, uint8_t (&y)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int c = c0 + 16 * u + gl; y[u] = c < nc ? src[3 * c + 1] : (uint8_t)'n'; }
  };
  auto count_y = [&](const uint8_t (&y)[4]) {
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) cnt += __popcll((__ballot(y[u] != 'n') >> gshift) & 0xFFFFull);
    return cnt;
  };
  for (int64_t i = grp; i < W; i += 2 * kGroups) {
    const int64_t i2 = i + kGroups;
    const bool two = i2 < W;
    const int nc1 = win_nc(i), nc2 = two ? win_nc(i2) : 0;
    const uint8_t *src1 = win_src(i), *src2 = two ? win_src(i2) : src1;
    uint8_t y1[4], y2[4];
    load_y(src1, nc1, 0, y1);
    load_y(src2, nc2, 0, y2);
    int cnt1 = count_y(y1), cnt2 = count_y(y2);
    for (int c0 = 64; c0 < nc1; c0 += 64) { load_y(src1, nc1, c0, y1); cnt1 += count_y(y1); }
    for (int c0 = 64; c0 < nc2; c0 += 64) { load_y(src2, nc2, c0, y2); cnt2 += count_y(y2); }
    if (gl == 0) { set_wo(i, cnt1); if (two) set_wo(i2, cnt2); }
  }
  __syncthreads();
  // exclusive scan of the counts over the piece's windows
  for (int64_t base = 0; base < W; base += kStatsThreads) {
    const int64_t i = base + tid;
    const int v = i < W ? get_wo(i) : 0;
    int inc = v;
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(inc, d);
      if (lane >= d) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int before = 0, total = 0;
    for (int k = 0; k < kStatsThreads / 64; ++k) {
      if (k < wave) before += s_wave[k];
      total += s_wave[k];
    }
    const int64_t carry = s_carry;
    if (i < W) set_wo(i, (int32_t)(carry + before + inc - v));
    __syncthreads();
    if (tid == 0) s_carry = carry + total;
    __syncthreads();
  }
  const int64_t n = s_carry;
  const int64_t rb = 3 * a.off[3 * w0];
  if (tid == 0) { a.row_off[p] = rb; a.cols[p] = n; }
  uint8_t *d0 = a.rows + rb, *d1 = d0 + n, *d2 = d1 + n;
  auto load_xyz = [&](const uint8_t *src, int nc, int c0, uint8_t (&x)[4], uint8_t (&y)[4], uint8_t (&z)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + 16 * u + gl;
      x[u] = 0; y[u] = 'n'; z[u] = 0;
      if (c < nc) { x[u] = src[3 * c]; y[u] = src[3 * c + 1]; z[u] = src[3 * c + 2]; }
    }
  };
  auto store_xyz = [&](int64_t k, const uint8_t (&x)[4], const uint8_t (&y)[4], const uint8_t (&z)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool keep = y[u] != 'n';
      const unsigned long long m = (__ballot(keep) >> gshift) & 0xFFFFull;
      if (keep) {
        const int64_t at = k + __popcll(m & ((1ull << gl) - 1ull));
        d0[at] = x[u]; d1[at] = y[u]; d2[at] = z[u];
      }
      k += __popcll(m);
    }
    return k;
  };
  for (int64_t i = grp; i < W; i += 2 * kGroups) {
    const int64_t i2 = i + kGroups;
    const bool two = i2 < W;
    const int nc1 = win_nc(i), nc2 = two ? win_nc(i2) : 0;
    const uint8_t *src1 = win_src(i), *src2 = two ? win_src(i2) : src1;
    int64_t k1 = get_wo(i), k2 = two ? get_wo(i2) : 0;
    uint8_t x1[4], y1[4], z1[4], x2[4], y2[4], z2[4];
    load_xyz(src1, nc1, 0, x1, y1, z1);
    load_xyz(src2, nc2, 0, x2, y2, z2);
    k1 = store_xyz(k1, x1, y1, z1);
    k2 = store_xyz(k2, x2, y2, z2);
    for (int c0 = 64; c0 < nc1; c0 += 64) { load_xyz(src1, nc1, c0, x1, y1, z1); k1 = store_xyz(k1, x1, y1, z1); }
    for (int c0 = 64; c0 < nc2; c0 += 64) { load_xyz(src2, nc2, c0, x2, y2, z2); k2 = store_xyz(k2, x2, y2, z2); }
  }
}

// ---- the merge with the WINDOWS as the unit of parallel work (round 4) ----
// k_merge above gives a block to a piece: a batch of 50 kb reads is 2,000 pieces of 850 windows, twelve blocks per CU
// working through hundreds of windows each (3.5 ms per batch); the pieces of 8 kb reads have 140.  Here every window
// has its 16-lane group whatever piece it belongs to: count its surviving columns (Donatello.cpp:13-31), one exclusive
// scan over all windows of the batch, per piece its column total and its place, then every window copies its
// columns to (its scan value less its piece's).
struct MergeWinArgs {
  int64_t n_windows, n_pieces;
  const int64_t *piece_first;
  const int64_t *off;
  const uint8_t *cols_in;
  int64_t cols_bytes;           // 3 * off[3 * n_windows]: no load may reach beyond
  const int32_t *ncol;
  const int32_t *status;
  int64_t *cnt;                 // n_windows + 1: surviving columns per window, then their exclusive scan
  int64_t *wdst;                // n_windows: where the window's first kept column goes in row 0, from `rows` (k_merge_pieces)
  int32_t *wn;                  // n_windows: columns of the window's piece
  uint8_t *rows;
  int64_t *row_off, *cols;
};

// Four columns of a window per lane: 12 consecutive bytes x0 y0 z0 x1 | y1 z1 x2 y2 | z2 x3 y3 z3 as ONE load (the
// address is byte-aligned: the hardware takes unaligned dwords), the three rows' letters sorted into a dword each by
// byte permutes.  A wavefront instruction moves 768 bytes where the byte-wide form moved 64: the texture addresser, which
// takes a wavefront's request at 64 lanes per 16 cycles whatever its width, was what the byte-wide kernels waited for
// (11 M requests per yeast -split batch: 0.29 of k_merge_copy's 0.51 ms).
struct __attribute__((packed)) Cols12 { uint32_t a, b, c; };

// the letters of columns c .. c + 3 of a window (nv of them valid, 1 <= nv <= 4); y's bytes beyond nv come back as 'n'
__device__ __forceinline__ void load_cols4(const uint8_t *src, int64_t src_at, int64_t limit, int c, int nv, uint32_t &x, uint32_t &y, uint32_t &z)
{
  if (src_at + 3 * (int64_t)c + 12 <= limit) {
    Cols12 v;
    __builtin_memcpy(&v, src + 3 * (int64_t)c, 12);
    x = __builtin_amdgcn_perm(v.c, __builtin_amdgcn_perm(v.b, v.a, 0x00060300u), 0x05020100u);
    y = __builtin_amdgcn_perm(v.c, __builtin_amdgcn_perm(v.b, v.a, 0x00070401u), 0x06020100u);
    z = __builtin_amdgcn_perm(v.c, __builtin_amdgcn_perm(v.b, v.a, 0x00000502u), 0x07040100u);
  } else {                                            // the batch's last bytes: letter by letter
    x = 0; y = 0; z = 0;
    for (int u = 0; u < nv; ++u) {
      x |= (uint32_t)src[3 * (c + u)] << (8 * u); y |= (uint32_t)src[3 * (c + u) + 1] << (8 * u); z |= (uint32_t)src[3 * (c + u) + 2] << (8 * u);
    }
  }
  if (nv < 4) { const uint32_t m = (1u << (8 * nv)) - 1u; y = (y & m) | (0x6E6E6E6Eu & ~m); }
}

// bit 7 of every byte of y that is not 'n' (Donatello.cpp:13-31: those columns stay)
__device__ __forceinline__ uint32_t kept_bytes(uint32_t y)
{
  const uint32_t t = y ^ 0x6E6E6E6Eu;
  return (((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u;
}

// A 16-lane group takes kMergeNW consecutive windows, the first 128 columns of each (nearly always all of them) asked for
// before any is looked at: the kernels wait for trips to memory, not for bandwidth, and a group with one window in
// flight had two of them (the window's descriptor, then its columns) for 200 bytes.
constexpr int kMergeNW = 2;
constexpr int kMergeWinPerBlock = kMergeNW * (kStatsThreads / 16);

__global__ void __launch_bounds__(kStatsThreads) k_merge_count(MergeWinArgs a)
{
  const int gl = threadIdx.x & 15;
  const int64_t wb = ((int64_t)blockIdx.x * (kStatsThreads / 16) + (threadIdx.x >> 4)) * kMergeNW;
  int nc[kMergeNW];
  int64_t at[kMergeNW];
#pragma unroll
  for (int j = 0; j < kMergeNW; ++j) {
    const int64_t w = wb + j;
    const bool on = w < a.n_windows;
    nc[j] = on && a.status[on ? w : 0] == 0 ? a.ncol[w] : 0;
    at[j] = on ? 3 * a.off[3 * w] : 0;
  }
  uint32_t y[kMergeNW][2];
#pragma unroll
  for (int j = 0; j < kMergeNW; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = 64 * h + 4 * gl;
      uint32_t x, z;
      y[j][h] = 0x6E6E6E6Eu;
      if (c < nc[j]) load_cols4(a.cols_in + at[j], at[j], a.cols_bytes, c, min(4, nc[j] - c), x, y[j][h], z);
    }
#pragma unroll
  for (int j = 0; j < kMergeNW; ++j) {
    int cnt = __popc(kept_bytes(y[j][0])) + __popc(kept_bytes(y[j][1]));
    for (int c = 128 + 4 * gl; c < nc[j]; c += 64) {
      uint32_t x, yy, z;
      load_cols4(a.cols_in + at[j], at[j], a.cols_bytes, c, min(4, nc[j] - c), x, yy, z);
      cnt += __popc(kept_bytes(yy));
    }
    for (int d = 1; d < 16; d <<= 1) cnt += __shfl_xor(cnt, d, 16);
    if (gl == 0 && wb + j < a.n_windows) a.cnt[wb + j] = cnt;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) a.cnt[a.n_windows] = 0;
}

// (the rows go where the scan over ALL windows of the batch says: piece after piece without a gap -- what the copy to the
// host wants; the block-per-piece form leaves every piece the room of its windows' letters and k_compact packs)
// Per window: where its first kept column goes in the reference's row, and the piece's columns (the distance to the next
// row) -- k_merge_copy then has one trip to memory in front of its columns instead of three.
__global__ void __launch_bounds__(kStatsThreads) k_merge_pieces(MergeWinArgs a)
{
  const int64_t p = blockIdx.x;
  const int64_t w0 = a.piece_first[p], w1 = a.piece_first[p + 1];
  const int64_t e0 = a.cnt[w0], n = a.cnt[w1] - e0;
  for (int64_t w = w0 + threadIdx.x; w < w1; w += kStatsThreads) { a.wdst[w] = 2 * e0 + a.cnt[w]; a.wn[w] = (int32_t)n; }
  if (threadIdx.x == 0) { a.cols[p] = n; a.row_off[p] = 3 * e0; }
}

__global__ void __launch_bounds__(kStatsThreads) k_merge_copy(MergeWinArgs a)
{
  const int gl = threadIdx.x & 15;
  const int64_t wb = ((int64_t)blockIdx.x * (kStatsThreads / 16) + (threadIdx.x >> 4)) * kMergeNW;
  int nc[kMergeNW];
  int64_t at[kMergeNW], to[kMergeNW], n[kMergeNW];
#pragma unroll
  for (int j = 0; j < kMergeNW; ++j) {
    const int64_t w = wb + j;
    const bool on = w < a.n_windows;
    nc[j] = on && a.status[on ? w : 0] == 0 ? a.ncol[w] : 0;
    at[j] = on ? 3 * a.off[3 * w] : 0;
    to[j] = on ? a.wdst[w] : 0;
    n[j] = on ? a.wn[w] : 0;
  }
  uint32_t x[kMergeNW][2], y[kMergeNW][2], z[kMergeNW][2];
#pragma unroll
  for (int j = 0; j < kMergeNW; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = 64 * h + 4 * gl;
      x[j][h] = 0; y[j][h] = 0x6E6E6E6Eu; z[j][h] = 0;
      if (c < nc[j]) load_cols4(a.cols_in + at[j], at[j], a.cols_bytes, c, min(4, nc[j] - c), x[j][h], y[j][h], z[j][h]);
    }
  // four columns of this lane go out; -> the columns the window's sixteen lanes kept together (they share nc: all of
  // them are here, and the scan over the DPP row sees every one)
  auto emit = [&](uint8_t *d0, int64_t nn, int64_t k, uint32_t xx, uint32_t yy, uint32_t zz) {
    const uint32_t keep = kept_bytes(yy);
    const int mine = __popc(keep);
    int inc = mine;
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, true);
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, true);
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, true);
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, true);
    uint8_t *p0 = d0 + k + (inc - mine), *p1 = p0 + nn, *p2 = p1 + nn;
    if (mine == 4) {
      __builtin_memcpy(p0, &xx, 4); __builtin_memcpy(p1, &yy, 4); __builtin_memcpy(p2, &zz, 4);
    } else if (mine > 0) {
      for (int u = 0; u < 4; ++u)
        if ((keep >> (8 * u + 7)) & 1u) { *p0++ = (uint8_t)(xx >> (8 * u)); *p1++ = (uint8_t)(yy >> (8 * u)); *p2++ = (uint8_t)(zz >> (8 * u)); }
    }
    return (int64_t)__shfl(inc, 15, 16);
  };
#pragma unroll
  for (int j = 0; j < kMergeNW; ++j) {
    if (nc[j] <= 0) continue;
    uint8_t *d0 = a.rows + to[j];
    int64_t k = emit(d0, n[j], 0, x[j][0], y[j][0], z[j][0]);
    if (nc[j] > 64) k += emit(d0, n[j], k, x[j][1], y[j][1], z[j][1]);
    for (int c0 = 128; c0 < nc[j]; c0 += 64) {
      const int c = c0 + 4 * gl, nv = min(4, nc[j] - c);
      uint32_t xx = 0, yy = 0x6E6E6E6Eu, zz = 0;
      if (nv > 0) load_cols4(a.cols_in + at[j], at[j], a.cols_bytes, c, nv, xx, yy, zz);
      k += emit(d0, n[j], k, xx, yy, zz);
    }
  }
}

// copy the pieces' rows from their roomy device layout to a dense one (out_off[p] = 3 * sum of cols before p).  The
// destination may be page-locked HOST memory (elector_msa_stats_enqueue_rows: the rows cross PCIe as this kernel's
// stores, no copy engine, no staging): 16-byte stores to 16-byte aligned addresses, a wavefront's stores side by
// side; the source's misalignment is taken out with a funnel shift over aligned dwords.
struct CompactArgs {
  const uint8_t *rows;
  const int64_t *row_off, *cols, *out_off;
  uint8_t *out;
};

typedef uint4 __attribute__((aligned(4))) uint4_a4;

__global__ void __launch_bounds__(kStatsThreads) k_compact(CompactArgs a)
{
  const int64_t p = blockIdx.x;
  const uint8_t *src = a.rows + a.row_off[p];
  uint8_t *dst = a.out + a.out_off[p];
  const int64_t nb = 3 * a.cols[p];
  const int64_t head = min(nb, (int64_t)((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15));
  for (int64_t i = threadIdx.x; i < head; i += kStatsThreads) dst[i] = src[i];
  const uint8_t *s2 = src + head;
  uint8_t *d2 = dst + head;
  const int64_t n2 = nb - head, nv = n2 >> 4;
  const int mis = (int)(reinterpret_cast<uintptr_t>(s2) & 3);
  const uint32_t *s4 = reinterpret_cast<const uint32_t *>(s2 - mis);
  for (int64_t v = threadIdx.x; v < nv; v += kStatsThreads) {
    // (the rows buffer has slack behind its last byte: the dword behind a piece's end may be read)
    const uint4 lo = *reinterpret_cast<const uint4_a4 *>(s4 + 4 * v);
    const uint32_t hi = s4[4 * v + 4];
    const int sh = 8 * mis;
    uint4 o;
    o.x = (uint32_t)(((((uint64_t)lo.y) << 32) | lo.x) >> sh);
    o.y = (uint32_t)(((((uint64_t)lo.z) << 32) | lo.y) >> sh);
    o.z = (uint32_t)(((((uint64_t)lo.w) << 32) | lo.z) >> sh);
    o.w = (uint32_t)(((((uint64_t)hi) << 32) | lo.w) >> sh);
    reinterpret_cast<uint4 *>(d2)[v] = o;
  }
  for (int64_t i = (nv << 4) + threadIdx.x; i < n2; i += kStatsThreads) d2[i] = s2[i];
}

// out_off[p] = 3 * (columns of the pieces before p), out_off[n] = 3 * all columns (one block)
__global__ void __launch_bounds__(1024) k_scan_cols(const int64_t *cols, int64_t n, int64_t *out_off)
{
  __shared__ int64_t part[1024];
  __shared__ int64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const int64_t v = i < n ? 3 * cols[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int s = 1; s < 1024; s <<= 1) {
      const int64_t t = (int)threadIdx.x >= s ? part[threadIdx.x - s] : 0;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < n) out_off[i] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) out_off[n] = carry;
}

// --------------------------------------------------------------------- stats ---

struct StatsArgs {
  int64_t n_reads;
  const int64_t *read_first;   // n_reads + 1 piece ids
  const uint8_t *rows;
  const int64_t *row_off;      // per piece: start of its three rows
  const int64_t *cols;         // per piece
  const int32_t *clips;        // may be null
  int64_t *counters;
  uint8_t *mask;               // 1 byte per column; piece p at mask + row_off[p] / 3 (written for the LAST read's pieces, and for pieces on the HBM path)
  int32_t *pool;               // bump-allocated scratch: interval lists, union bytes of split reads
  unsigned long long pool_cap; // in int32 units
  unsigned long long *pool_used;
  int32_t *overflow;           // set when the pool ran out: the host grows it and runs again
  int bit_words;               // 64-column words per bit row in LDS; longer pieces read their letters from HBM throughout
};

// computeStats.py:61-77
__device__ int left_gaps(const uint8_t *row, int n)
{
  int gaps = 0, nts = 0, total = 0, i = 0;
  while (i < n && nts <= kThresh) {
    if (row[i] == '.') { ++gaps; nts = 0; }
    else { if (gaps >= kThresh) total = i; gaps = 0; ++nts; }
    ++i;
  }
  return total;
}

// computeStats.py:82-98
__device__ int right_gaps(const uint8_t *row, int n)
{
  int gaps = 0, nts = 0, total = 0, i = n - 1;
  while (i >= 0 && nts <= kThresh) {
    if (row[i] == '.') { ++gaps; nts = 0; }
    else { if (gaps >= kThresh) total = n - i; gaps = 0; ++nts; }
    --i;
  }
  return total;
}

// findGapStretches (computeStats.py:104-189), its per-column state machine:
//   countGapsCor  (cg)  = k for the k-th column of a run of corrected gaps, except 0 on the run's first
//                         column unless the run starts the record (:113-127)
//   countGapsRef  (cgr) is advanced on a reference gap only when the PREVIOUS corrected column was a gap,
//                         from 0 to 2 and then by one (from 0 to 1 on the record's first column), and is
//                         cleared by a reference letter
// A column is a "hit" when cg >= THRESH and cgr < THRESH2 (:133); hits exist only inside runs of at least
// THRESH corrected gaps.  One lane walks one such run and reports its first and last hit.
// Runs longer than kWalkCap columns (the part of the reference a trimmed or split piece does not cover:
// thousands of columns) are not walked by one lane: it returns first_hit = kRunLong with the run's entry
// count in *last_hit, and the whole block finishes the run (block_gap_run).
constexpr int kWalkCap = 192;
constexpr int kRunLong = -2;

__device__ void walk_gap_run(const uint8_t *cor, const uint8_t *ref, int n, int s, int *first_hit, int *last_hit)
{
  // reference-gap count on entry: replay the reference gap run that reaches column s - 1
  int rb = s;
  while (rb > 0 && ref[rb - 1] == '.') --rb;
  int cgr = 0;
  auto step = [&](int x) {
    const bool rgap = ref[x] == '.';
    if (x == 0) { if (rgap) cgr = 1; }
    else if (cor[x - 1] == '.' && rgap) cgr = cgr > 0 ? cgr + 1 : 2;
    if (!rgap) cgr = 0;
  };
  for (int x = rb; x < s; ++x) step(x);
  const int entry = cgr;
  int fh = -1, lh = -1;
  for (int x = s; x < n && cor[x] == '.'; ++x) {
    if (x - s >= kWalkCap) { *first_hit = kRunLong; *last_hit = entry; return; }
    step(x);
    if (x - s + 1 >= kThresh && cgr < kThresh2) { if (fh < 0) fh = x; lh = x; }
  }
  *first_hit = fh;
  *last_hit = lh;
}

// The same for one long run, by the whole block.  Inside a run every corrected column is a gap, so the
// count at column x depends only on the reference gaps that reach x without interruption:
//   reference letter at x                        -> 0
//   streak of reference gaps starting at a > s    -> 2 + (x - a)
//   streak starting at the run's first column s   -> c0 + (x - s) when c0 > 0, else x - s + 1 (0 at x = s)
// with c0 = the count on entry (1 for a run that starts the record).  A column is a hit when it is at least
// the THRESH-th of the run and the count is below THRESH2, so looking back THRESH2 columns decides it.
__device__ void block_gap_run(const uint8_t *cor, const uint8_t *ref, int n, int s, int entry, int *sh /* 3 ints, shared */,
                              int *first_hit, int *last_hit)
{
  const int tid = threadIdx.x;
  // the run's end
  int end = -1;
  for (int base = s; end < 0; base += (int)blockDim.x) {
    __syncthreads();
    if (tid == 0) sh[0] = 0x7fffffff;
    __syncthreads();
    const int x = base + tid;
    if (x >= n || cor[x] != '.') atomicMin(&sh[0], x);
    __syncthreads();
    if (sh[0] != 0x7fffffff) end = sh[0] - 1;
  }
  __syncthreads();
  if (tid == 0) { sh[1] = 0x7fffffff; sh[2] = -1; }
  __syncthreads();
  const int c0 = s == 0 ? 1 : entry;
  int fh = 0x7fffffff, lh = -1;
  for (int x = s + kThresh - 1 + tid; x <= end; x += (int)blockDim.x) {
    int cgr;
    if (ref[x] != '.') cgr = 0;
    else {
      int a = -1;                                        // start of the streak of reference gaps that reaches x
      for (int j = 1; j <= kThresh2 && a < 0; ++j) {
        if (x - j < s) a = s;
        else if (ref[x - j] != '.') a = x - j + 1;
      }
      if (a < 0) cgr = kThresh2;                          // a longer streak: the count is past THRESH2 whatever c0 is
      else if (a > s) cgr = 2 + (x - a);
      else cgr = c0 > 0 ? c0 + (x - s) : x - s + 1;
    }
    if (cgr < kThresh2) { fh = min(fh, x); lh = max(lh, x); }
  }
  if (lh >= 0) { atomicMin(&sh[1], fh); atomicMax(&sh[2], lh); }
  __syncthreads();
  *first_hit = sh[2] >= 0 ? sh[1] : -1;
  *last_hit = sh[2];
  __syncthreads();
}

// the interval-list part of findGapStretches (:146-188) on the runs that had hits: pairs in `runs`
// (n_ne of them), n_total = length of the reference's position list including its empty entries.
// Returns the number of kept stretches, written as pairs to dict.
__device__ int stretch_intervals(int32_t *runs, int n_ne, int n_total, int n, int32_t *tmp, int32_t *mrg, int32_t *dict)
{
  // borders (:146-162)
  int nt = 0;
  for (int k = 0; k < n_ne; ++k) {
    const int s0 = runs[2 * k], s1 = runs[2 * k + 1];
    if (n_total > 1) {
      if (s0 <= kThresh2) { tmp[2 * nt] = 0; tmp[2 * nt + 1] = s1; ++nt; }
      if (n - s1 <= kThresh2) { tmp[2 * nt] = s0; tmp[2 * nt + 1] = n - 1; ++nt; }
      else { tmp[2 * nt] = s0; tmp[2 * nt + 1] = s1; ++nt; }
    } else {
      if (s0 <= kThresh2) { tmp[2 * nt] = 0; tmp[2 * nt + 1] = s1; ++nt; }
      else { tmp[2 * nt] = s0; tmp[2 * nt + 1] = s1; ++nt; }
      if (n - s1 <= kThresh2) tmp[2 * (nt - 1) + 1] = n - 1;
    }
  }
  // merge neighbours (:165-177)
  int nm = 0;
  bool merge = false;
  for (int i = 0; i + 1 < nt; ++i) {
    if (tmp[2 * (i + 1)] - tmp[2 * i + 1] <= kThresh) { mrg[2 * nm] = tmp[2 * i]; mrg[2 * nm + 1] = tmp[2 * (i + 1) + 1]; merge = true; }
    else { mrg[2 * nm] = tmp[2 * i]; mrg[2 * nm + 1] = tmp[2 * i + 1]; merge = false; }
    ++nm;
  }
  if (!merge && nt > 0) { mrg[2 * nm] = tmp[2 * (nt - 1)]; mrg[2 * nm + 1] = tmp[2 * (nt - 1) + 1]; ++nm; }
  // keep only long stretches touching an end; dict semantics: same start overwrites (:182-188)
  int nd = 0;
  for (int i = 0; i < nm; ++i) {
    const int s0 = mrg[2 * i], s1 = mrg[2 * i + 1];
    if ((s0 == 0 || s1 == n - 1) && s1 - s0 > kThresh2) {
      int j = 0;
      while (j < nd && dict[2 * j] != s0) ++j;
      dict[2 * j] = s0; dict[2 * j + 1] = s1;
      if (j == nd) ++nd;
    }
  }
  return nd;
}

__device__ inline bool is_gc(uint8_t c) { return c == 'g' || c == 'c' || c == 'G' || c == 'C'; }

__device__ inline int32_t *pool_take(const StatsArgs &a, unsigned long long ints)
{
  const unsigned long long at = atomicAdd(a.pool_used, ints);
  if (at + ints > a.pool_cap) { *a.overflow = 1; return nullptr; }
  return a.pool + at;
}

// sum over the block; every thread gets the result
__device__ inline int block_sum(int v, int *red)
{
  for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
  for (int k = 0; k < (int)blockDim.x / 64; ++k) t += red[k];
  return t;
}

enum { kAccN = 18 };

// ---- shared state of k_stats (file scope, so that the per-piece helpers address it as LDS) ----
constexpr int kLdsRuns = 64;             // gap runs of a piece whose interval lists stay in LDS
constexpr int kBitArrays = 6;
enum { kGx = 0, kGc = 1, kGu = 2, kExc = 3, kExu = 4, kEuc = 5, kMk = kGu /* the mask takes the place of Gu */,
       kUn = 6 /* a split read's realNotMissing: the OR of its pieces' masks, column index by column index */ };
constexpr int kBitRows = kBitArrays + 1;
__shared__ int s_red[kStatsThreadsMax / 64];
__shared__ int s_end[4];                 // end-gap scans: left ref, left unc, right ref, right unc
__shared__ int s_acc[kAccN];
__shared__ int s_long[3];               // block_gap_run: end of the run, first and last hit
__shared__ int s_n[8];                   // 0: run starts filled  1: stretches kept  2: clip left  3: clip right  4: earliest qualifying run end  5: union size  6: run starts filled (pool)  7: the union is a bit row in LDS
__shared__ int32_t *s_lists;
__shared__ uint8_t *s_uni;
__shared__ int32_t s_lst[14 * (kLdsRuns + 2)];
__shared__ int32_t s_start[kLdsRuns];
// Bit rows of the piece at hand (dynamic LDS; StatsArgs::bit_words words per row and a zero guard word either
// side): bit i of row kGx/kGc/kGu = "reference / corrected / uncorrected has a gap in column i", of
// kExc/kExu/kEuc = "the two letters are equal".  Everything after the one pass that reads the letters --
// end scans, gap runs, interval lists, mask, the masked counters -- works on these words.
extern __shared__ unsigned long long s_bits[];

__device__ __forceinline__ unsigned long long bwr(int arr, int W2, int j) { return s_bits[arr * W2 + 1 + j]; }
__device__ __forceinline__ void bww(int arr, int W2, int j, unsigned long long v) { s_bits[arr * W2 + 1 + j] = v; }
__device__ __forceinline__ bool bbit(int arr, int W2, int i) { return (bwr(arr, W2, i >> 6) >> (i & 63)) & 1ull; }

// the bits of word j that stand for columns lo..hi
__device__ __forceinline__ unsigned long long range_word(int j, int lo, int hi)
{
  int x = lo - 64 * j, y = hi - 64 * j;
  if (y < 0 || x > 63 || hi < lo) return 0;
  x = max(x, 0); y = min(y, 63);
  return (~0ull << x) & (~0ull >> (63 - y));
}

// set bits of a row in columns lo..hi; every thread gets the result
__device__ int count_range(int arr, int W2, int lo, int hi)
{
  int v = 0;
  if (hi >= lo)
    for (int j = (lo >> 6) + (int)threadIdx.x; j <= (hi >> 6); j += (int)blockDim.x) v += __popcll(bwr(arr, W2, j) & range_word(j, lo, hi));
  return block_sum(v, s_red);
}

// left_gaps / right_gaps on a bit row
__device__ int left_gaps_bits(int arr, int W2, int n)
{
  int gaps = 0, nts = 0, total = 0;
  unsigned long long w = 0;
  for (int i = 0; i < n && nts <= kThresh; ++i) {
    if ((i & 63) == 0) w = bwr(arr, W2, i >> 6);
    if ((w >> (i & 63)) & 1ull) { ++gaps; nts = 0; }
    else { if (gaps >= kThresh) total = i; gaps = 0; ++nts; }
  }
  return total;
}

__device__ int right_gaps_bits(int arr, int W2, int n)
{
  int gaps = 0, nts = 0, total = 0;
  unsigned long long w = bwr(arr, W2, (n - 1) >> 6);
  for (int i = n - 1; i >= 0 && nts <= kThresh; --i) {
    if ((i & 63) == 63) w = bwr(arr, W2, i >> 6);
    if ((w >> (i & 63)) & 1ull) { ++gaps; nts = 0; }
    else { if (gaps >= kThresh) total = n - i; gaps = 0; ++nts; }
  }
  return total;
}

// walk_gap_run on the bit rows.  The run's end comes from the words first, so that a long run (the part of
// the reference a trimmed or split piece does not cover) is recognised without walking its first kWalkCap
// columns; *run_end = its last column.
__device__ void walk_gap_run_bits(int W2, int n, int s, int32_t *first_hit, int32_t *last_hit, int32_t *run_end)
{
  int j = s >> 6;
  unsigned long long w = ~bwr(kGc, W2, j) & (~0ull << (s & 63));
  while (!w) { ++j; w = ~bwr(kGc, W2, j); }                              // the guard word after the row ends the search
  const int e = min(64 * j + (int)__builtin_ctzll(w), n);                // first column after the run
  int rb = s;
  while (rb > 0 && bbit(kGx, W2, rb - 1)) --rb;
  int cgr = 0;
  auto step = [&](int x) {
    const bool rgap = bbit(kGx, W2, x);
    if (x == 0) { if (rgap) cgr = 1; }
    else if (bbit(kGc, W2, x - 1) && rgap) cgr = cgr > 0 ? cgr + 1 : 2;
    if (!rgap) cgr = 0;
  };
  for (int x = rb; x < s; ++x) step(x);
  *run_end = e - 1;
  if (e - s > kWalkCap) { *first_hit = kRunLong; *last_hit = cgr; return; }
  int fh = -1, lh = -1;
  for (int x = s; x < e; ++x) {
    step(x);
    if (x - s + 1 >= kThresh && cgr < kThresh2) { if (fh < 0) fh = x; lh = x; }
  }
  *first_hit = fh;
  *last_hit = lh;
}

// block_gap_run on the bit rows, for the run s..end
__device__ void block_gap_run_bits(int W2, int s, int end, int entry, int *first_hit, int *last_hit)
{
  const int tid = threadIdx.x;
  __syncthreads();
  if (tid == 0) { s_long[1] = 0x7fffffff; s_long[2] = -1; }
  __syncthreads();
  const int c0 = s == 0 ? 1 : entry;
  int fh = 0x7fffffff, lh = -1;
  for (int x = s + kThresh - 1 + tid; x <= end; x += (int)blockDim.x) {
    int cgr;
    if (!bbit(kGx, W2, x)) cgr = 0;
    else {
      int a = -1;
      for (int j = 1; j <= kThresh2 && a < 0; ++j) {
        if (x - j < s) a = s;
        else if (!bbit(kGx, W2, x - j)) a = x - j + 1;
      }
      if (a < 0) cgr = kThresh2;
      else if (a > s) cgr = 2 + (x - a);
      else cgr = c0 > 0 ? c0 + (x - s) : x - s + 1;
    }
    if (cgr < kThresh2) { fh = min(fh, x); lh = max(lh, x); }
  }
  if (lh >= 0) { atomicMin(&s_long[1], fh); atomicMax(&s_long[2], lh); }
  __syncthreads();
  *first_hit = s_long[2] >= 0 ? s_long[1] : -1;
  *last_hit = s_long[2];
  __syncthreads();
}

// starts of the runs of >= THRESH corrected gaps in word j of kGc
__device__ __forceinline__ unsigned long long long_run_starts(int W2, int j, unsigned long long *first_out)
{
  const unsigned long long g = bwr(kGc, W2, j), gp = bwr(kGc, W2, j - 1) >> 63, gn = bwr(kGc, W2, j + 1);
  const unsigned long long first = g & ~((g << 1) | gp);
  *first_out = first;
  return first & ((g >> 1) | (gn << 63)) & ((g >> 2) | (gn << 62)) & ((g >> 3) | (gn << 61)) & ((g >> 4) | (gn << 60));
}

__global__ void __launch_bounds__(kStatsThreadsMax) k_stats(StatsArgs a)
{
  const int64_t r = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t p0 = a.read_first[r], p1 = a.read_first[r + 1];
  const int nfrag = (int)(p1 - p0);
  const int W2 = a.bit_words + 2;
  const bool last_read = r == a.n_reads - 1;

  // realNotMissing of a split read (:589-591): a bit per column of its longest piece in LDS (row kUn) when every piece
  // has its bit rows there -- the pieces' mask words are ORed in, the last piece counts from the words; a byte per column
  // in the pool otherwise (round 4 did that for every split read: three byte-wide passes over global memory per piece)
  int ucap = 0;
  bool uni_lds = false;
  if (nfrag > 1) {
    if (tid == 0) {
      int64_t maxc = 0;
      for (int64_t p = p0; p < p1; ++p) maxc = a.cols[p] > maxc ? a.cols[p] : maxc;
      s_n[5] = (int)maxc;
      s_n[7] = (maxc + 63) / 64 <= a.bit_words ? 1 : 0;
      s_uni = s_n[7] ? nullptr : reinterpret_cast<uint8_t *>(pool_take(a, (unsigned long long)(maxc + 3) / 4 + 1));
    }
    __syncthreads();
    ucap = s_n[5];
    uni_lds = s_n[7] != 0;
    if (uni_lds) { for (int j = tid; j < (ucap + 63) / 64; j += (int)blockDim.x) bww(kUn, W2, j, 0ull); }
    else {
      if (!s_uni) return;
      for (int i = tid; i < ucap; i += (int)blockDim.x) s_uni[i] = 0;
    }
  }
  uint8_t *uni = nfrag > 1 && !uni_lds ? s_uni : nullptr;

  int64_t missing = 0;
  for (int64_t p = p0; p < p1; ++p) {
    int64_t *out = a.counters + p * ES_NCOUNTERS;
    const int n = (int)a.cols[p];
    __syncthreads();
    if (n <= 10) {                                                   // computeStats.py:577,624
      if (tid == 0) {
        for (int k = 0; k < ES_NCOUNTERS; ++k) out[k] = 0;
        out[ES_EXT_LEFT] = out[ES_EXT_RIGHT] = -1;
        out[ES_MISSING_LAST] = -1;
        out[ES_MISSING] = missing;
      }
      continue;
    }
    const uint8_t *ref = a.rows + a.row_off[p], *cor = ref + n, *unc = cor + n;
    uint8_t *mask = a.mask + a.row_off[p] / 3;
    const int nw = (n + 63) >> 6;
    const bool fast = nw <= a.bit_words;
    int gl, gr;
    int64_t ext_left = -1, ext_right = -1;
    if (fast) {
      // ---- the one pass over the letters: bit rows, G/C counts ----
      if (tid < kBitArrays) { bww(tid, W2, -1, 0); bww(tid, W2, nw, 0); }
      if (tid < kAccN) s_acc[tid] = 0;
      if (tid == 0) { s_n[0] = 0; s_n[1] = 0; s_n[2] = 0; s_n[3] = n - 1; s_n[4] = 0x7fffffff; s_n[6] = 0; s_lists = nullptr; }
      int gcx = 0, gcc = 0;
      constexpr int kU = 8;                                           // 24 byte loads in flight per thread
      for (int c0 = 0; c0 < n; c0 += kU * (int)blockDim.x) {
        uint8_t x[kU], c[kU], u[kU];
#pragma unroll
        for (int k = 0; k < kU; ++k) {
          const int i = min(c0 + k * (int)blockDim.x + tid, n - 1);
          x[k] = ref[i]; c[k] = cor[i]; u[k] = unc[i];
        }
#pragma unroll
        for (int k = 0; k < kU; ++k) {
          const bool v = c0 + k * (int)blockDim.x + tid < n;
          const unsigned long long bx = __ballot(v && x[k] == '.'), bc = __ballot(v && c[k] == '.'), bu = __ballot(v && u[k] == '.'),
                                   exc = __ballot(v && x[k] == c[k]), exu = __ballot(v && x[k] == u[k]), euc = __ballot(v && u[k] == c[k]);
          gcx += __popcll(__ballot(v && is_gc(x[k])));
          gcc += __popcll(__ballot(v && is_gc(c[k])));
          const int j = ((c0 + k * (int)blockDim.x) >> 6) + wave;
          if (lane == 0 && j < nw) {
            bww(kGx, W2, j, bx); bww(kGc, W2, j, bc); bww(kGu, W2, j, bu);
            bww(kExc, W2, j, exc); bww(kExu, W2, j, exu); bww(kEuc, W2, j, euc);
          }
        }
      }
      __syncthreads();
      if (lane == 0) { atomicAdd(&s_acc[7], gcx); atomicAdd(&s_acc[8], gcc); }

      // ---- gapsAndExtensions (:472-498) ----
      // (the four end scans on the first lanes of the block's wavefronts, in turn where it has fewer than four)
      for (int k = wave; k < 4; k += (int)blockDim.x >> 6)
        if (lane == 0) s_end[k] = k == 0 ? left_gaps_bits(kGx, W2, n) : k == 1 ? left_gaps_bits(kGu, W2, n)
                                : k == 2 ? right_gaps_bits(kGx, W2, n) : right_gaps_bits(kGu, W2, n);
      __syncthreads();
      gl = min(s_end[0], s_end[1]); gr = min(s_end[2], s_end[3]);
      if (gl >= kThresh && gl >= kThresh2) ext_left = gl - count_range(kGc, W2, 0, gl - 1);
      if (gr >= kThresh && gr >= kThresh2) ext_right = gr - count_range(kGc, W2, n - gr + 1, n - 1);

      // ---- findGapStretches (:104-189) ----
      int nlong = 0, nq = 0, qmin = 0x7fffffff;
      for (int j = tid; j < nw; j += (int)blockDim.x) {
        unsigned long long first;
        unsigned long long ls = long_run_starts(W2, j, &first);
        nlong += __popcll(ls);
        const unsigned long long g = bwr(kGc, W2, j), e1 = (g >> 1) | (bwr(kGc, W2, j + 1) << 63);
        const int rem = n - 1 - 64 * j;                                 // columns i of this word with i + 1 < n
        const unsigned long long valid1 = rem >= 64 ? ~0ull : rem <= 0 ? 0ull : ((1ull << rem) - 1ull);
        const unsigned long long q = g & ~e1 & valid1 & (~first | (j == 0 ? 1ull : 0ull));
        nq += __popcll(q);
        if (q) qmin = min(qmin, 64 * j + (int)__builtin_ctzll(q) + 1);
        while (ls) {
          const int b = (int)__builtin_ctzll(ls);
          ls &= ls - 1;
          const int at = atomicAdd(&s_n[0], 1);
          if (at < kLdsRuns) s_start[at] = 64 * j + b;
        }
      }
      const int m = block_sum(nlong, s_red);
      nq = block_sum(nq, s_red);
      int nd = 0;
      int32_t *dict = nullptr;
      if (m > 0) {
        atomicMin(&s_n[4], qmin);
        if (tid == 0) s_lists = m <= kLdsRuns ? s_lst : pool_take(a, 14ull * (unsigned long long)(m + 2));
        __syncthreads();
        int32_t *lists = s_lists;
        if (!lists) return;                                             // uniform: the host runs again with a larger pool
        int32_t *runs = lists, *tmp = runs + 2 * (m + 2), *mrg = tmp + 4 * (m + 2);
        dict = mrg + 4 * (m + 2);
        if (m <= kLdsRuns) {
          for (int e = tid; e < m; e += (int)blockDim.x) tmp[e] = s_start[e];
        } else {
          for (int j = tid; j < nw; j += (int)blockDim.x) {
            unsigned long long first;
            unsigned long long ls = long_run_starts(W2, j, &first);
            while (ls) {
              const int b = (int)__builtin_ctzll(ls);
              ls &= ls - 1;
              tmp[atomicAdd(&s_n[6], 1)] = 64 * j + b;
            }
          }
        }
        __syncthreads();
        for (int e = tid; e < m; e += (int)blockDim.x) {
          const int v = tmp[e];
          int rank = 0;
          for (int x = 0; x < m; ++x) rank += tmp[x] < v;
          runs[rank] = v;
        }
        __syncthreads();
        for (int e = tid; e < m; e += (int)blockDim.x) walk_gap_run_bits(W2, n, runs[e], &mrg[2 * e], &mrg[2 * e + 1], &tmp[e]);
        __syncthreads();
        for (int e = 0; e < m; ++e) {                                   // long runs: all threads together
          if (mrg[2 * e] != kRunLong) continue;
          int fh, lh;
          block_gap_run_bits(W2, runs[e], tmp[e], mrg[2 * e + 1], &fh, &lh);
          if (tid == 0) { mrg[2 * e] = fh; mrg[2 * e + 1] = lh; }
          __syncthreads();
        }
        if (tid == 0) {
          int n_ne = 0, first_hit = 0x7fffffff;
          for (int e = 0; e < m; ++e) {
            const int fh = mrg[2 * e], lh = mrg[2 * e + 1];
            if (fh < 0) continue;
            if (n_ne == 0) first_hit = fh;
            runs[2 * n_ne] = fh - kThresh + 1;
            runs[2 * n_ne + 1] = lh;
            ++n_ne;
          }
          const int n_total = nq + ((n_ne > 0 && s_n[4] > first_hit) ? 1 : 0);
          s_n[1] = stretch_intervals(runs, n_ne, n_total, n, tmp, mrg, dict);
        }
        __syncthreads();
        nd = s_n[1];
      }
      for (int k = 0; k < nd; ++k) {
        const int s0 = dict[2 * k], s1 = dict[2 * k + 1];
        missing += s1 - s0 - count_range(kGx, W2, s0, s1);
      }
      missing -= gl + gr;
      if (missing < 0) missing = 0;

      // ---- getCorrectedPositions (:712-752) ----
      if (a.clips && tid == 0) {
        const int lc = a.clips[2 * p], rc = a.clips[2 * p + 1];
        int i = 0, j = 0;
        while (j < lc && i < n) { if (!bbit(kGc, W2, i)) ++j; ++i; }
        s_n[2] = i;
        if (lc != 0 || rc != 0) {
          const int right_clip = n - rc;
          i = n - 1; j = n - 1;
          while (j >= right_clip && i >= 0) { if (!bbit(kGc, W2, i)) --j; --i; }
          s_n[3] = i;
        }
      }
      __syncthreads();
      const int clip_l = s_n[2], clip_r = s_n[3];

      // ---- the mask and the masked counters (:399-440 with :291-328 and :371-393), 64 columns per step ----
      const int lo = max(clip_l, gl >= kThresh ? gl : 0), hi = min(clip_r, gr >= kThresh ? n - gr : n - 1);
      int acc[kAccN];
#pragma unroll
      for (int k = 0; k < kAccN; ++k) acc[k] = 0;
      for (int j = tid; j < nw; j += (int)blockDim.x) {
        unsigned long long M = range_word(j, lo, hi);
        for (int k = 0; k < nd; ++k) M &= ~range_word(j, dict[2 * k], dict[2 * k + 1]);
        const unsigned long long gx = bwr(kGx, W2, j), gc = bwr(kGc, W2, j), gu = bwr(kGu, W2, j),
                                 exc = bwr(kExc, W2, j), exu = bwr(kExu, W2, j), euc = bwr(kEuc, W2, j);
        bww(kMk, W2, j, M);
        if (uni_lds) bww(kUn, W2, j, bwr(kUn, W2, j) | M);                        // realNotMissing (:589-591)
        acc[15] += __popcll(gx); acc[16] += __popcll(gc); acc[17] += __popcll(gu);
        const unsigned long long dc = M & ~exc, du = M & ~exu, eq = M & exu;
        acc[12] += __popcll(dc & gx); acc[14] += __popcll(dc & ~gx & ~gc); acc[13] += __popcll(dc & ~gx & gc);
        acc[9] += __popcll(du & gx); acc[11] += __popcll(du & ~gx & ~gu); acc[10] += __popcll(du & ~gx & gu);
        int t = __popcll(eq & ~euc); acc[1] += t; acc[4] += t;
        t = __popcll(eq & euc); acc[0] += t; acc[3] += t;
        acc[5] += __popcll(eq);
        acc[6] += __popcll(du);
        t = __popcll(du & exc); acc[0] += t; acc[3] += t;
        acc[4] += __popcll(du & ~exc);
        t = __popcll(du & ~exc & euc); acc[2] += t; acc[1] += t;
      }
#pragma unroll
      for (int k = 0; k < kAccN; ++k) {
        int v = acc[k];
        for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d);
        if (lane == 0 && v) atomicAdd(&s_acc[k], v);
      }
      __syncthreads();
      if (tid == 0) { s_acc[15] = n - s_acc[15]; s_acc[16] = n - s_acc[16]; s_acc[17] = n - s_acc[17]; }
      // the mask as bytes: only the batch's LAST read's is ever taken out (call site #2's report), a split read's
      // pieces also fold theirs into the read's union -- for every other read the words in LDS were all of it
      // (a byte per column of every read was 119 MB of byte-wide stores per yeast -split batch)
      if (last_read || uni)
        for (int i = tid; i < n; i += (int)blockDim.x) {
          const uint8_t mk = bbit(kMk, W2, i) ? 1 : 0;
          if (last_read) mask[i] = mk;
          if (uni) uni[i] |= mk;                                              // realNotMissing (:589-591)
        }
    } else {

      // ---- gapsAndExtensions (:472-498): the four end scans on four lanes ----
      for (int k = wave; k < 4; k += (int)blockDim.x >> 6)
        if (lane == 0) s_end[k] = k == 0 ? left_gaps(ref, n) : k == 1 ? left_gaps(unc, n) : k == 2 ? right_gaps(ref, n) : right_gaps(unc, n);
      if (tid < kAccN) s_acc[tid] = 0;
      if (tid == 0) { s_n[0] = 0; s_n[1] = 0; s_n[2] = 0; s_n[3] = n - 1; s_n[4] = 0x7fffffff; s_lists = nullptr; }
      __syncthreads();
      gl = min(s_end[0], s_end[1]); gr = min(s_end[2], s_end[3]);
      if (gl >= kThresh && gl >= kThresh2) {
        int dots = 0;
        for (int i = tid; i < gl; i += (int)blockDim.x) dots += cor[i] == '.';
        ext_left = gl - block_sum(dots, s_red);
      }
      if (gr >= kThresh && gr >= kThresh2) {
        int dots = 0;
        for (int i = n - gr + 1 + tid; i < n; i += (int)blockDim.x) dots += cor[i] == '.';
        ext_right = gr - block_sum(dots, s_red);
      }

      // ---- findGapStretches (:104-189) ----
      // pass 1: runs of >= THRESH corrected gaps, and the ends of runs that leave countGapsCor > 0
      int nlong = 0, nq = 0, qmin = 0x7fffffff;
      for (int i = tid; i < n; i += (int)blockDim.x) {
        if (cor[i] != '.') continue;
        const bool first = i == 0 || cor[i - 1] != '.';
        if (first && i + kThresh - 1 < n) {
          bool all = true;
          for (int k = 1; k < kThresh; ++k) all = all && cor[i + k] == '.';
          nlong += all;
        }
        if (i + 1 < n && cor[i + 1] != '.' && (i == 0 || !first)) { ++nq; qmin = min(qmin, i + 1); }
      }
      const int m = block_sum(nlong, s_red);
      nq = block_sum(nq, s_red);
      int nd = 0;
      int32_t *dict = nullptr;
      if (m > 0) {
        atomicMin(&s_n[4], qmin);
        if (tid == 0) s_lists = pool_take(a, 14ull * (unsigned long long)(m + 2));
        __syncthreads();
        int32_t *lists = s_lists;
        if (!lists) return;                                             // uniform: the host runs again with a larger pool
        int32_t *runs = lists, *tmp = runs + 2 * (m + 2), *mrg = tmp + 4 * (m + 2);
        dict = mrg + 4 * (m + 2);
        // pass 2: the run starts, in any order, then sorted by rank
        for (int i = tid; i < n; i += (int)blockDim.x) {
          if (cor[i] != '.' || !(i == 0 || cor[i - 1] != '.') || i + kThresh - 1 >= n) continue;
          bool all = true;
          for (int k = 1; k < kThresh; ++k) all = all && cor[i + k] == '.';
          if (all) tmp[atomicAdd(&s_n[0], 1)] = i;
        }
        __syncthreads();
        for (int e = tid; e < m; e += (int)blockDim.x) {
          const int v = tmp[e];
          int rank = 0;
          for (int x = 0; x < m; ++x) rank += tmp[x] < v;
          runs[rank] = v;
        }
        __syncthreads();
        for (int e = tid; e < m; e += (int)blockDim.x) walk_gap_run(cor, ref, n, runs[e], &mrg[2 * e], &mrg[2 * e + 1]);
        __syncthreads();
        for (int e = 0; e < m; ++e) {                                   // long runs: all threads together
          if (mrg[2 * e] != kRunLong) continue;
          int fh, lh;
          block_gap_run(cor, ref, n, runs[e], mrg[2 * e + 1], s_long, &fh, &lh);
          if (tid == 0) { mrg[2 * e] = fh; mrg[2 * e + 1] = lh; }
          __syncthreads();
        }
        if (tid == 0) {
          int n_ne = 0, first_hit = 0x7fffffff;
          for (int e = 0; e < m; ++e) {
            const int fh = mrg[2 * e], lh = mrg[2 * e + 1];
            if (fh < 0) continue;
            if (n_ne == 0) first_hit = fh;
            runs[2 * n_ne] = fh - kThresh + 1;
            runs[2 * n_ne + 1] = lh;
            ++n_ne;
          }
          // the reference's list also holds one empty entry per ended run; the very first hit opens the
          // list itself when no run has ended before it (:134-137)
          const int n_total = nq + ((n_ne > 0 && s_n[4] > first_hit) ? 1 : 0);
          s_n[1] = stretch_intervals(runs, n_ne, n_total, n, tmp, mrg, dict);
        }
        __syncthreads();
        nd = s_n[1];
      }
      for (int k = 0; k < nd; ++k) {
        const int s0 = dict[2 * k], s1 = dict[2 * k + 1];
        int dots = 0;
        for (int i = s0 + tid; i <= s1; i += (int)blockDim.x) dots += ref[i] == '.';
        missing += s1 - s0 - block_sum(dots, s_red);
      }
      missing -= gl + gr;
      if (missing < 0) missing = 0;

      // ---- getCorrectedPositions (:712-752): soft clips walk the corrected row from both ends ----
      if (a.clips && tid == 0) {
        const int lc = a.clips[2 * p], rc = a.clips[2 * p + 1];
        int i = 0, j = 0;
        while (j < lc && i < n) { if (cor[i] != '.') ++j; ++i; }
        s_n[2] = i;                                                     // columns < i are clipped
        if (lc != 0 || rc != 0) {
          const int right_clip = n - rc;
          i = n - 1; j = n - 1;
          while (j >= right_clip && i >= 0) { if (cor[i] != '.') --j; --i; }
          s_n[3] = i;                                                   // columns > i are clipped
        }
      }
      __syncthreads();
      const int clip_l = s_n[2], clip_r = s_n[3];

      // ---- per-column counters (:399-440 with :291-328 and :371-393) ----
      int acc[kAccN];
  #pragma unroll
      for (int k = 0; k < kAccN; ++k) acc[k] = 0;
      for (int i = tid; i < n; i += (int)blockDim.x) {
        const uint8_t x = ref[i], c = cor[i], u = unc[i];
        bool mk = i >= clip_l && i <= clip_r;
        if (gl >= kThresh && i < gl) mk = false;
        if (gr >= kThresh && i > n - gr) mk = false;
        for (int k = 0; k < nd; ++k) if (i >= dict[2 * k] && i <= dict[2 * k + 1]) mk = false;
        mask[i] = mk ? 1 : 0;
        acc[7] += is_gc(x); acc[8] += is_gc(c);
        acc[15] += x != '.'; acc[16] += c != '.'; acc[17] += u != '.';
        if (!mk) continue;
        if (c != x) { if (x == '.') ++acc[12]; else if (c != '.') ++acc[14]; else ++acc[13]; }
        if (u != x) { if (x == '.') ++acc[9]; else if (u != '.') ++acc[11]; else ++acc[10]; }
        if (x == u) { if (u != c) { ++acc[1]; ++acc[4]; } else { ++acc[0]; ++acc[3]; } ++acc[5]; }
        else { if (x == c) { ++acc[0]; ++acc[3]; } else { if (u == c) { ++acc[2]; ++acc[1]; } ++acc[4]; } ++acc[6]; }
      }
  #pragma unroll
      for (int k = 0; k < kAccN; ++k) {
        int v = acc[k];
        for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d);
        if (lane == 0 && v) atomicAdd(&s_acc[k], v);
      }
      __syncthreads();
      if (uni)
        for (int i = tid; i < n; i += (int)blockDim.x) uni[i] |= mask[i];      // realNotMissing (:589-591)
    }
    int64_t missing_last = -1;
    if (nfrag > 1) {
      if (p == p1 - 1) {                                                    // last piece (:595-599)
        __syncthreads();
        int miss = 0;
        if (uni_lds)                                                        // (every piece was on the fast path: the rows are the last piece's)
          for (int j = tid; j < nw; j += (int)blockDim.x) miss += __popcll(~bwr(kUn, W2, j) & ~bwr(kGx, W2, j) & range_word(j, 0, n - 1));
        else
          for (int i = tid; i < n; i += (int)blockDim.x) miss += (!uni[i] && ref[i] != '.');
        missing_last = block_sum(miss, s_red);
      }
    }
    if (tid == 0) {
      // s_acc order = ES_TP .. ES_LEN_UNC
      for (int k = 0; k < kAccN; ++k) out[k] = s_acc[k];
      out[ES_GAPS_LEFT] = gl;
      out[ES_GAPS_RIGHT] = gr;
      out[ES_EXT_LEFT] = ext_left;
      out[ES_EXT_RIGHT] = ext_right;
      out[ES_MISSING] = missing;
      out[ES_MISSING_LAST] = missing_last;
      out[ES_PROCESSED] = 1;
    }
  }
}

}  // namespace elector

using namespace elector;

namespace {

// scratch pool of k_stats: [used counter u64][overflow i32][pad][ints...]
struct PoolView { unsigned long long *used; int32_t *overflow; int32_t *ints; unsigned long long cap; };

int pool_prepare(elector_ctx *c, unsigned long long ints, PoolView *v)
{
  if (c->d_st_scr.ensure((size_t)ints * 4 + 64)) return ELECTOR_E_NOMEM;
  uint8_t *base = c->d_st_scr.as<uint8_t>();
  v->used = reinterpret_cast<unsigned long long *>(base);
  v->overflow = reinterpret_cast<int32_t *>(base + 8);
  v->ints = reinterpret_cast<int32_t *>(base + 16);
  v->cap = (c->d_st_scr.cap - 16) / 4;
  return 0;
}

// k_stats with bit rows for pieces of up to `max_cols` columns (as far as LDS goes)
constexpr int kBitWordsMax = 2560;       // 163,840 columns: 7 rows = 143 KB, one workgroup per CU
int launch_stats(elector_ctx *c, StatsArgs a, int64_t max_cols, hipStream_t st)
{
  const char *force = std::getenv("ELECTOR_STATS_BITWORDS");            // tests: 0 = every piece on the HBM path
  int64_t words = (std::max<int64_t>(max_cols, 1) + 63) / 64;
  words = std::min<int64_t>(std::max<int64_t>(words, 64), kBitWordsMax);
  if (force) words = std::min<int64_t>(std::max(0, std::atoi(force)), kBitWordsMax);
  a.bit_words = (int)words;
  const size_t lds = (size_t)kBitRows * (size_t)(words + 2) * 8;
  static bool big_ok = false;
  if (lds > 48 * 1024 && !big_ok) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_stats), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)((size_t)kBitRows * (kBitWordsMax + 2) * 8));
    if (e != hipSuccess) return elector_fail(c, ELECTOR_E_HIP, "hipFuncSetAttribute(k_stats)", e);
    big_ok = true;
  }
  // threads per read: 256 for reads of a few thousand columns; 1024 when the batch's reads run to tens of thousands
  // (50 kb reads: 2,000 blocks of ~60 k columns each kept a CU's three resident blocks busy for 0.8 ms apiece).
  // ELECTOR_STATS_THREADS=256|512|1024 forces one (A/B)
  static const int forced = std::getenv("ELECTOR_STATS_THREADS") ? std::atoi(std::getenv("ELECTOR_STATS_THREADS")) : 0;
  int threads = max_cols >= 64000 ? kStatsThreadsMax : kStatsThreads;
  if (forced == 64 || forced == 128 || forced == 256 || forced == 512 || forced == 1024) threads = forced;
  hipLaunchKernelGGL(k_stats, dim3((unsigned)a.n_reads), dim3((unsigned)threads), lds, st, a);
  return 0;
}

// runs k_stats, growing the pool to the worst case once if the first size was too small;
// total_cols bounds the sum of the pieces' columns
int run_stats(elector_ctx *c, StatsArgs a, int64_t n_pieces, int64_t total_cols, int64_t max_cols)
{
  hipStream_t st = c->stream;
  unsigned long long want = std::max<unsigned long long>(1ull << 20, (unsigned long long)total_cols / 4);   // pool_first_size
  for (int attempt = 0; attempt < 2; ++attempt) {
    PoolView pv;
    if (pool_prepare(c, want, &pv)) return elector_fail(c, ELECTOR_E_NOMEM, "statistics scratch pool");
    HIPCHK(c, hipMemsetAsync(c->d_st_scr.p, 0, 16, st));
    a.pool = pv.ints; a.pool_cap = pv.cap; a.pool_used = pv.used; a.overflow = pv.overflow;
    timed_begin(c, 3, st);
    if (launch_stats(c, a, max_cols, st)) return ELECTOR_E_HIP;
    timed_end(c, st);
    HIPCHK(c, hipGetLastError());
    int32_t over = 0;
    HIPCHK(c, hipMemcpyAsync(&over, pv.overflow, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (!over) return ELECTOR_OK;
    // worst case: a run list entry per THRESH columns (14 ints each) + a union byte per column
    want = 14ull * ((unsigned long long)total_cols / kThresh + 3ull * (unsigned long long)n_pieces) +
           (unsigned long long)total_cols / 4 + 2ull * (unsigned long long)a.n_reads + 1024;
  }
  return elector_fail(c, ELECTOR_E_NOMEM, "statistics scratch pool overflow");
}

}  // namespace

extern "C" int elector_stats_batch(elector_ctx *c, int64_t n_reads, const int64_t *read_first, int64_t n_pieces,
                                   const uint8_t *rows, const int64_t *row_off, const int64_t *cols,
                                   const int32_t *clips, int64_t *counters, uint8_t *last_mask)
{
  if (!c) return ELECTOR_E_INVAL;
  if (n_reads < 0 || n_pieces < 0 || !read_first || (n_pieces > 0 && (!rows || !row_off || !cols || !counters)))
    return elector_fail(c, ELECTOR_E_INVAL, "bad arguments");
  if (n_reads == 0 || n_pieces == 0) return ELECTOR_OK;
  if (n_reads > 0x7fffffff) return elector_fail(c, ELECTOR_E_INVAL, "too many reads in one call");
  if (read_first[0] != 0 || read_first[n_reads] != n_pieces) return elector_fail(c, ELECTOR_E_INVAL, "read_first must cover all pieces");
  std::lock_guard<std::mutex> lock(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  int64_t total_cols = 0;
  for (int64_t p = 0; p < n_pieces; ++p) {
    if (cols[p] < 0 || cols[p] > 0x3fffffff || row_off[p + 1] - row_off[p] != 3 * cols[p])
      return elector_fail(c, ELECTOR_E_INVAL, "row_off/cols mismatch");
    total_cols += cols[p];
  }
  for (int64_t r = 0; r < n_reads; ++r)
    if (read_first[r + 1] < read_first[r]) return elector_fail(c, ELECTOR_E_INVAL, "read_first must be non-decreasing");
  const int64_t r0 = row_off[0], total_rows = row_off[n_pieces] - r0;
  int rc = c->d_st_rows.ensure((size_t)total_rows + r0 % 3 + 64) | c->d_st_rowoff.ensure((size_t)(n_pieces + 1) * 8) |
           c->d_st_cols.ensure((size_t)n_pieces * 8) | c->d_st_first.ensure((size_t)(n_reads + 1) * 8) |
           c->d_st_cnt.ensure((size_t)n_pieces * ES_NCOUNTERS * 8) | c->d_st_mask.ensure((size_t)total_cols + 64) |
           (clips ? c->d_st_clips.ensure((size_t)n_pieces * 8) : 0);
  if (rc) return elector_fail(c, ELECTOR_E_NOMEM, "statistics workspace");
  hipStream_t st = c->stream;
  // device offsets are relative to the first piece
  std::vector<int64_t> rel((size_t)n_pieces + 1);
  for (int64_t p = 0; p <= n_pieces; ++p) rel[(size_t)p] = row_off[p] - r0;
  HIPCHK(c, hipMemcpyAsync(c->d_st_rows.p, rows + r0, (size_t)total_rows, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->d_st_rowoff.p, rel.data(), (size_t)(n_pieces + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->d_st_cols.p, cols, (size_t)n_pieces * 8, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->d_st_first.p, read_first, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, st));
  if (clips) HIPCHK(c, hipMemcpyAsync(c->d_st_clips.p, clips, (size_t)n_pieces * 8, hipMemcpyHostToDevice, st));
  StatsArgs a{};
  a.n_reads = n_reads;
  a.read_first = c->d_st_first.as<int64_t>();
  a.rows = c->d_st_rows.as<uint8_t>();
  a.row_off = c->d_st_rowoff.as<int64_t>();
  a.cols = c->d_st_cols.as<int64_t>();
  a.clips = clips ? c->d_st_clips.as<int32_t>() : nullptr;
  a.counters = c->d_st_cnt.as<int64_t>();
  a.mask = c->d_st_mask.as<uint8_t>();
  int64_t max_cols = 0;
  for (int64_t p = 0; p < n_pieces; ++p) max_cols = std::max(max_cols, cols[p]);
  rc = run_stats(c, a, n_pieces, total_cols, max_cols);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(counters, c->d_st_cnt.p, (size_t)n_pieces * ES_NCOUNTERS * 8, hipMemcpyDeviceToHost, st));
  if (last_mask) {
    const int64_t pl = read_first[n_reads - 1];
    const int64_t nb = (rel[(size_t)n_pieces] - rel[(size_t)pl]) / 3;
    if (nb > 0)
      HIPCHK(c, hipMemcpyAsync(last_mask, c->d_st_mask.as<uint8_t>() + rel[(size_t)pl] / 3, (size_t)nb, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(c, hipStreamSynchronize(st));
  return ELECTOR_OK;
}

namespace {

unsigned long long pool_first_size(int64_t total_cols) { return std::max<unsigned long long>(1ull << 20, (unsigned long long)total_cols / 4); }

// worst case: a run list entry per THRESH columns (14 ints each) + a union byte per column
unsigned long long pool_worst_size(int64_t total_cols, int64_t n_pieces, int64_t n_reads)
{
  return 14ull * ((unsigned long long)total_cols / kThresh + 3ull * (unsigned long long)n_pieces) +
         (unsigned long long)total_cols / 4 + 2ull * (unsigned long long)n_reads + 1024;
}

StatsArgs slot_args(const elector::StatsSlot &s)
{
  StatsArgs a{};
  a.n_reads = s.n_reads;
  a.read_first = s.first.as<int64_t>();
  a.rows = s.rows.as<uint8_t>();
  a.row_off = s.rowoff.as<int64_t>();
  a.cols = s.cols.as<int64_t>();
  a.clips = s.has_clips ? s.clips.as<int32_t>() : nullptr;
  a.counters = s.cnt.as<int64_t>();
  a.mask = s.mask.as<uint8_t>();
  return a;
}

// k_stats of a slot on the context's stream with a pool of `ints`, then its results to the slot's
// pinned staging: [overflow i32, pad][counters][cols]
int enqueue_stats(elector_ctx *c, elector::StatsSlot &s, unsigned long long ints)
{
  hipStream_t st = c->stream;
  PoolView pv;
  if (pool_prepare(c, ints, &pv)) return elector_fail(c, ELECTOR_E_NOMEM, "statistics scratch pool");
  HIPCHK(c, hipMemsetAsync(c->d_st_scr.p, 0, 16, st));
  StatsArgs a = slot_args(s);
  a.pool = pv.ints; a.pool_cap = pv.cap; a.pool_used = pv.used; a.overflow = pv.overflow;
  timed_begin(c, 3, st);
  // columns of the longest piece, unknown on the host: 96 per window (the splitter's windows average 57 bases)
  if (launch_stats(c, a, 96 * s.max_windows, st)) return ELECTOR_E_HIP;
  timed_end(c, st);
  HIPCHK(c, hipGetLastError());
  uint8_t *h = s.h.as<uint8_t>();
  // (kernel stores into the slot's page-locked block: the copy engine belongs to the merged rows, and the counters of
  // this batch must not wait behind an earlier batch's rows)
  if (elector::launch_words_to_host(h, pv.overflow, 4, st) ||
      elector::launch_words_to_host(h + 16, s.cnt.p, (size_t)s.n_pieces * ES_NCOUNTERS * 8, st) ||
      elector::launch_words_to_host(h + 16 + (size_t)s.n_pieces * ES_NCOUNTERS * 8, s.cols.p, (size_t)s.n_pieces * 8, st)) {
    HIPCHK(c, hipMemcpyAsync(h, pv.overflow, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(h + 16, s.cnt.p, (size_t)s.n_pieces * ES_NCOUNTERS * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(h + 16 + (size_t)s.n_pieces * ES_NCOUNTERS * 8, s.cols.p, (size_t)s.n_pieces * 8, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(c, hipEventRecord(s.done, st));
  return ELECTOR_OK;
}

}  // namespace

static int stats_enqueue(elector_ctx *c, int64_t n_windows, const uint8_t *d_cols, const int32_t *d_ncol,
                         const int32_t *d_status, int64_t n_pieces, const int64_t *piece_first,
                         int64_t n_reads, const int64_t *read_first, const int32_t *clips, uint8_t *rows_out, int64_t rows_cap);

extern "C" int elector_msa_stats_enqueue(elector_ctx *c, int64_t n_windows, const uint8_t *d_cols, const int32_t *d_ncol,
                                         const int32_t *d_status, int64_t n_pieces, const int64_t *piece_first,
                                         int64_t n_reads, const int64_t *read_first, const int32_t *clips)
{
  return stats_enqueue(c, n_windows, d_cols, d_ncol, d_status, n_pieces, piece_first, n_reads, read_first, clips, nullptr, 0);
}

// Host function in the context's stream behind the kernels that packed a job's rows: the statistics kernel has left the
// pieces' column counts in the slot's page-locked block, so the byte count is known HERE, without the caller -- the copy
// starts on the DMA engine the moment the rows exist instead of when the job is collected (a caller with four batches in
// flight collects a job three batches later, and at the end of a run all at once: the last four copies then queued up one
// behind the other behind the last kernel, 4 x 6.3 ms on the yeast `-split` batch).  HSA calls only (a host function may
// not call into HIP).  Leaves the job as it is when the statistics have to be run again (collect then sends the rows).
static void rows_go(void *p)
{
  elector::StatsSlot &s = *static_cast<elector::StatsSlot *>(p);
  const uint8_t *h = s.h.as<uint8_t>();
  if (!s.rows_host || *reinterpret_cast<const int32_t *>(h) != 0) return;
  const int64_t *hcols = reinterpret_cast<const int64_t *>(h + 16 + (size_t)s.n_pieces * ES_NCOUNTERS * 8);
  int64_t nbytes = 0;
  for (int64_t k = 0; k < s.n_pieces; ++k) nbytes += 3 * hcols[k];
  if (nbytes <= 0 || nbytes > (int64_t)s.rows_src_cap) return;
  if (elector::rows_dma_start(s.device, s.rows_host, s.rows_src, (size_t)nbytes, &s.rows_sig) != 0) return;
  s.rows_by_dma = true;
  s.rows_inflight = true;
  s.rows_host = nullptr;
}

// the fence of a slot's rows copy (see elector_msa_stats_collect)
static int rows_fence(elector_ctx *c, elector::StatsSlot &s)
{
  if (!s.rows_inflight) return 0;
  s.rows_inflight = false;
  if (s.rows_by_dma) return elector::rows_dma_wait(s.rows_sig);
  return hipEventSynchronize(s.rows_done) != hipSuccess;
}

extern "C" int elector_msa_stats_enqueue_rows(elector_ctx *c, int64_t n_windows, const uint8_t *d_cols, const int32_t *d_ncol,
                                              const int32_t *d_status, int64_t n_pieces, const int64_t *piece_first,
                                              int64_t n_reads, const int64_t *read_first, const int32_t *clips,
                                              uint8_t *rows_out, int64_t rows_cap)
{
  if (!c) return ELECTOR_E_INVAL;
  if (!rows_out || rows_cap < 0) return elector_fail(c, ELECTOR_E_INVAL, "bad arguments");
  return stats_enqueue(c, n_windows, d_cols, d_ncol, d_status, n_pieces, piece_first, n_reads, read_first, clips, rows_out, rows_cap);
}

static int stats_enqueue(elector_ctx *c, int64_t n_windows, const uint8_t *d_cols, const int32_t *d_ncol,
                         const int32_t *d_status, int64_t n_pieces, const int64_t *piece_first,
                         int64_t n_reads, const int64_t *read_first, const int32_t *clips, uint8_t *rows_out, int64_t rows_cap)
{
  if (!c) return ELECTOR_E_INVAL;
  if (n_windows < 0 || n_pieces < 0 || n_reads < 0 || !piece_first || !read_first ||
      (n_windows > 0 && (!d_cols || !d_ncol || !d_status)))
    return elector_fail(c, ELECTOR_E_INVAL, "bad arguments");
  if (n_windows != c->last_n) return elector_fail(c, ELECTOR_E_INVAL, "the windows must be those of the context's last elector_poa_batch_device call");
  if (n_reads > 0x7fffffff || n_pieces > 0x7fffffff) return elector_fail(c, ELECTOR_E_INVAL, "too many reads in one call");
  if (piece_first[0] != 0 || piece_first[n_pieces] != n_windows) return elector_fail(c, ELECTOR_E_INVAL, "piece_first must cover all windows");
  if (read_first[0] != 0 || read_first[n_reads] != n_pieces) return elector_fail(c, ELECTOR_E_INVAL, "read_first must cover all pieces");
  for (int64_t p = 0; p < n_pieces; ++p)
    if (piece_first[p + 1] < piece_first[p]) return elector_fail(c, ELECTOR_E_INVAL, "piece_first must be non-decreasing");
  for (int64_t r = 0; r < n_reads; ++r)
    if (read_first[r + 1] < read_first[r]) return elector_fail(c, ELECTOR_E_INVAL, "read_first must be non-decreasing");
  std::lock_guard<std::mutex> lock(c->mu);
  if (c->st_inflight >= elector_ctx::kStatsSlots) return elector_fail(c, ELECTOR_E_INVAL, "two statistics jobs in flight: collect one first");
  HIPCHK(c, hipSetDevice(c->device));
  uint8_t *rows_dev = nullptr;                // a device destination
  bool rows_is_host = false;
  if (rows_out) {
    if (rows_cap < 3 * c->last_total) return elector_fail(c, ELECTOR_E_INVAL, "rows_cap must hold 3 bytes per base of the batch");
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, rows_out) != hipSuccess) { (void)hipGetLastError(); return elector_fail(c, ELECTOR_E_INVAL, "rows_out must be page-locked host memory or device memory"); }
    if (at.type == hipMemoryTypeDevice) rows_dev = rows_out;
    else if (at.type == hipMemoryTypeHost) rows_is_host = true;
    else return elector_fail(c, ELECTOR_E_INVAL, "rows_out must be page-locked host memory or device memory");
  }
  elector::StatsSlot &s = c->st_slot[c->st_head];
  s.rows_host = nullptr;
  if (!s.done) HIPCHK(c, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  s.n_pieces = n_pieces; s.n_reads = n_reads; s.total = c->last_total; s.has_clips = clips != nullptr;
  s.last_piece = n_reads > 0 ? read_first[n_reads - 1] : 0;
  s.max_windows = 0;
  for (int64_t p = 0; p < n_pieces; ++p) s.max_windows = std::max(s.max_windows, piece_first[p + 1] - piece_first[p]);
  const int64_t total = s.total;            // bases of the batch: an upper bound of its MSA columns
  if (n_pieces > 0 && n_reads > 0) {
    hipStream_t st = c->stream;
    // The slot's rows buffer is about to be written again: a copy to the host that still reads from it (the merge with
    // the windows as the unit leaves the rows packed, and they go out from where they lie) has to be through.
    if (rows_fence(c, s)) return elector_fail(c, ELECTOR_E_HIP, "rows copy");
    // both slots grow together: the second job of a pipelined caller must not pay for allocations
    int rc = 0;
    for (int k = 0; k < elector_ctx::kStatsSlots; ++k) {
      elector::StatsSlot &z = c->st_slot[k];
      if (z.rows_inflight && (size_t)3 * total + 64 > z.rows.cap) {     // (it grows = is freed: not under a copy out of it)
        HIPCHK(c, hipStreamSynchronize(st));
        if (rows_fence(c, z)) return elector_fail(c, ELECTOR_E_HIP, "rows copy");
      }
      rc |= z.rows.ensure((size_t)3 * total + 64) | z.rowoff.ensure((size_t)(n_pieces + 1) * 8) |
            z.cols.ensure((size_t)n_pieces * 8) | z.first.ensure((size_t)(n_reads + 1 + n_pieces + 1) * 8) |
            z.cnt.ensure((size_t)n_pieces * ES_NCOUNTERS * 8) | z.mask.ensure((size_t)total + 64) |
            z.woff.ensure((size_t)(n_windows + 1) * 4) | (clips ? z.clips.ensure((size_t)n_pieces * 8) : 0) |
            z.h.ensure(16 + (size_t)n_pieces * (ES_NCOUNTERS + 1) * 8 + (size_t)(n_reads + n_pieces + 2) * 8 +
                       (clips ? (size_t)n_pieces * 8 : 0));
    }
    if (rc) return elector_fail(c, ELECTOR_E_NOMEM, "statistics workspace");
    // inputs go through the slot's pinned block so that the uploads do not wait for the device
    uint8_t *hin = s.h.as<uint8_t>() + 16 + (size_t)n_pieces * (ES_NCOUNTERS + 1) * 8;
    std::memcpy(hin, read_first, (size_t)(n_reads + 1) * 8);
    std::memcpy(hin + (size_t)(n_reads + 1) * 8, piece_first, (size_t)(n_pieces + 1) * 8);
    int64_t *d_read_first = s.first.as<int64_t>(), *d_piece_first = d_read_first + (n_reads + 1);
    HIPCHK(c, hipMemcpyAsync(d_read_first, hin, (size_t)(n_reads + n_pieces + 2) * 8, hipMemcpyHostToDevice, st));
    if (clips) {
      uint8_t *hc = hin + (size_t)(n_reads + n_pieces + 2) * 8;
      std::memcpy(hc, clips, (size_t)n_pieces * 8);
      HIPCHK(c, hipMemcpyAsync(s.clips.p, hc, (size_t)n_pieces * 8, hipMemcpyHostToDevice, st));
    }
    MergeArgs m{};
    m.n_pieces = n_pieces;
    m.piece_first = d_piece_first;
    m.off = c->d_off.as<int64_t>();
    m.cols_in = d_cols;
    m.ncol = d_ncol;
    m.status = d_status;
    m.rows = s.rows.as<uint8_t>();
    m.row_off = s.rowoff.as<int64_t>();
    m.cols = s.cols.as<int64_t>();
    m.woff = s.woff.as<int32_t>();
    // A block per piece (k_merge) for small batches; a lane group per window (k_merge_count, one scan over the batch's
    // windows, k_merge_pieces, k_merge_copy) for the rest.  Round 4 drew the line at pieces of 400 windows (50 kb reads: the
    // blocks' long serial loops were 3.45 ms of merge + counters per batch, 2.83 with the windows as the unit; on 8 kb reads
    // the four launches and the scan cost 0.2 ms more than they saved).  Round 5: with twelve-byte loads, the windows'
    // destinations worked out per piece and the rows written without gaps the window form is level with k_merge on 8 kb
    // reads (0.40 against 0.44 ms) and the rows need no packing pass on their way to the host: every batch of 32,768
    // windows or more takes it.  ELECTOR_MERGE_PER_PIECE=1 / =0 force one or the other (A/B, tests).
    const char *force = std::getenv("ELECTOR_MERGE_PER_PIECE");
    const bool per_piece = force ? std::atoi(force) != 0 : n_windows < 32768;
    timed_begin(c, 3, st);
    if (per_piece || n_windows == 0)
      hipLaunchKernelGGL(k_merge, dim3((unsigned)n_pieces), dim3(kStatsThreads), 0, st, m);
    else {
      MergeWinArgs mw{};
      mw.n_windows = n_windows; mw.n_pieces = n_pieces;
      mw.piece_first = d_piece_first; mw.off = m.off; mw.cols_in = d_cols; mw.ncol = d_ncol; mw.status = d_status;
      mw.cols_bytes = 3 * total;
      size_t tmp_bytes = 0;
      (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (int64_t *)nullptr, (int64_t *)nullptr, (int)(n_windows + 1), st);
      for (int k = 0; k < elector_ctx::kStatsSlots; ++k)
        if (c->st_slot[k].wcnt.ensure((size_t)(n_windows + 1) * 8 * 2 + tmp_bytes + 256) | c->st_slot[k].wpiece.ensure((size_t)n_windows * 4 + 64))
          return elector_fail(c, ELECTOR_E_NOMEM, "merge workspace");
      int64_t *cnt_in = s.wcnt.as<int64_t>(), *cnt_ex = cnt_in + (n_windows + 1);
      void *tmp = cnt_ex + (n_windows + 1);
      mw.cnt = cnt_in; mw.wn = s.wpiece.as<int32_t>();
      mw.wdst = cnt_in;                  // (the counts are dead once they are scanned)
      mw.rows = m.rows; mw.row_off = m.row_off; mw.cols = m.cols;
      const unsigned wblocks = (unsigned)((n_windows + kMergeWinPerBlock - 1) / kMergeWinPerBlock);
      hipLaunchKernelGGL(k_merge_count, dim3(wblocks), dim3(kStatsThreads), 0, st, mw);
      if (hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, cnt_in, cnt_ex, (int)(n_windows + 1), st) != hipSuccess)
        return elector_fail(c, ELECTOR_E_HIP, "merge scan");
      mw.cnt = cnt_ex;
      hipLaunchKernelGGL(k_merge_pieces, dim3((unsigned)n_pieces), dim3(kStatsThreads), 0, st, mw);
      hipLaunchKernelGGL(k_merge_copy, dim3(wblocks), dim3(kStatsThreads), 0, st, mw);
    }
    timed_end(c, st);
    HIPCHK(c, hipGetLastError());
    const int rc2 = enqueue_stats(c, s, pool_first_size(total));
    if (rc2) return rc2;
    if (rows_out) {
      // The merged rows, packed, to where the caller wants them -- in the queue, behind the counters.  Device memory:
      // the packing kernel writes there.  Host memory: it packs into a device buffer, and the moment the job is
      // collected (the host then knows the byte count) ONE copy of exactly that size goes out on the context's copy
      // stream: the copy engine moves it (57 GB/s on this box, measured; a kernel's own stores to host memory reach
      // 20) while the kernels of the following batches run.  elector_msa_rows_wait() waits for it.
      // (The merge with the windows as the unit puts the pieces' rows one behind the other already: a host destination
      // then gets them from where they lie, no packing pass -- 0.16 ms and 714 MB of traffic per yeast -split batch.)
      const bool packed = !(per_piece || n_windows == 0) && rows_is_host;
      if (s.outoff.ensure((size_t)(n_pieces + 1) * 8)) return elector_fail(c, ELECTOR_E_NOMEM, "row offsets");
      uint8_t *dst = rows_dev;
      if (rows_is_host && packed) {
        s.rows_src = s.rows.p; s.rows_src_cap = s.rows.cap;
        if (!s.rows_done) HIPCHK(c, hipEventCreateWithFlags(&s.rows_done, hipEventDisableTiming));
        if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
      } else if (rows_is_host) {
        for (int k = 0; k < elector_ctx::kStatsSlots; ++k) {
          elector::StatsSlot &o = c->st_slot[k];
          if ((size_t)3 * total + 64 > o.dense.cap) {
            // the buffer grows (= is freed): not under a copy that is on its way out of it or about to start
            HIPCHK(c, hipStreamSynchronize(st));
            if (rows_fence(c, o)) return elector_fail(c, ELECTOR_E_HIP, "rows copy");
          }
          if (o.dense.ensure((size_t)3 * total + 64)) return elector_fail(c, ELECTOR_E_NOMEM, "packed rows");
        }
        dst = s.dense.as<uint8_t>();
        s.rows_src = s.dense.p; s.rows_src_cap = s.dense.cap;
        if (!s.rows_done) HIPCHK(c, hipEventCreateWithFlags(&s.rows_done, hipEventDisableTiming));
        if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
      }
      if (!(rows_is_host && packed)) {
        timed_begin(c, 3, st);
        hipLaunchKernelGGL(k_scan_cols, dim3(1), dim3(1024), 0, st, s.cols.as<int64_t>(), n_pieces, s.outoff.as<int64_t>());
        CompactArgs ca{s.rows.as<uint8_t>(), s.rowoff.as<int64_t>(), s.cols.as<int64_t>(), s.outoff.as<int64_t>(), dst};
        hipLaunchKernelGGL(k_compact, dim3((unsigned)n_pieces), dim3(kStatsThreads), 0, st, ca);
        timed_end(c, st);
        HIPCHK(c, hipGetLastError());
      }
      s.rows_host = rows_is_host ? rows_out : nullptr;
      s.device = c->device;
      if (rows_is_host && !std::getenv("ELECTOR_ROWS_AT_COLLECT") && elector::rows_dma_ready(c->device, &s.rows_sig))
        HIPCHK(c, hipLaunchHostFunc(st, rows_go, &s));
      HIPCHK(c, hipEventRecord(s.done, st));
    }
  }
  c->st_head = (c->st_head + 1) % elector_ctx::kStatsSlots;
  ++c->st_inflight;
  return ELECTOR_OK;
}

extern "C" int elector_msa_stats_collect(elector_ctx *c, int64_t n_pieces, int64_t *counters, int64_t *piece_cols,
                                         uint8_t *last_rows, uint8_t *last_mask, int64_t last_cap)
{
  if (!c) return ELECTOR_E_INVAL;
  std::lock_guard<std::mutex> lock(c->mu);
  if (c->st_inflight <= 0) return elector_fail(c, ELECTOR_E_INVAL, "no statistics job in flight");
  elector::StatsSlot &s = c->st_slot[c->st_tail];
  if (n_pieces != s.n_pieces || (n_pieces > 0 && !counters)) return elector_fail(c, ELECTOR_E_INVAL, "n_pieces differs from the job's");
  HIPCHK(c, hipSetDevice(c->device));
  {   // (a bundle search noted for the context's last batch: queued here, where the caller waits for the context anyway)
    const int rcb = elector_bundles_flush(c);
    if (rcb) return rcb;
  }
  c->st_last = c->st_tail;
  c->st_tail = (c->st_tail + 1) % elector_ctx::kStatsSlots;
  --c->st_inflight;
  if (s.n_pieces == 0 || s.n_reads == 0) return ELECTOR_OK;
  hipStream_t st = c->stream;
  HIPCHK(c, hipEventSynchronize(s.done));
  const uint8_t *h = s.h.as<uint8_t>();
  if (*reinterpret_cast<const int32_t *>(h) != 0) {
    // the interval scratch ran out: run the statistics kernel again with the worst-case pool
    const int rc = enqueue_stats(c, s, pool_worst_size(s.total, s.n_pieces, s.n_reads));
    if (rc) return rc;
    HIPCHK(c, hipEventSynchronize(s.done));
    if (*reinterpret_cast<const int32_t *>(h) != 0) return elector_fail(c, ELECTOR_E_NOMEM, "statistics scratch pool overflow");
  }
  std::memcpy(counters, h + 16, (size_t)n_pieces * ES_NCOUNTERS * 8);
  const int64_t *hcols = reinterpret_cast<const int64_t *>(h + 16 + (size_t)n_pieces * ES_NCOUNTERS * 8);
  if (piece_cols) std::memcpy(piece_cols, hcols, (size_t)n_pieces * 8);
  if (s.rows_host) {
    // the packed rows leave for the host now that their size is known; nothing waits here (elector_msa_rows_wait)
    int64_t nbytes = 0;
    for (int64_t p = 0; p < n_pieces; ++p) nbytes += 3 * hcols[p];
    if (nbytes > 0) {
      // On the DMA engine through the HSA runtime (rows_dma.cpp: the HIP runtime ran four such copies in five as blit kernels
      // on the compute units); through HIP on the copy stream where that is not to be had.  Nothing waits here: the kernels
      // that packed the rows are through (s.done above).
      s.rows_by_dma = elector::rows_dma_start(c->device, s.rows_host, s.rows_src, (size_t)nbytes, &s.rows_sig) == 0;
      if (!s.rows_by_dma) {
        HIPCHK(c, hipMemcpyAsync(s.rows_host, s.rows_src, (size_t)nbytes, hipMemcpyDeviceToHost, c->copy_stream));
        HIPCHK(c, hipEventRecord(s.rows_done, c->copy_stream));
      }
      s.rows_inflight = true;
    }
    s.rows_host = nullptr;
  }
  const int64_t pl = s.last_piece, npl = n_pieces - pl;
  if ((last_rows || last_mask) && npl > 0) {
    int64_t need = 0;
    for (int64_t k = 0; k < npl; ++k) need += hcols[pl + k];
    if (need > last_cap) return elector_fail(c, ELECTOR_E_INVAL, "last_cap is smaller than the last read's columns");
    std::vector<int64_t> loff((size_t)npl);
    HIPCHK(c, hipMemcpyAsync(loff.data(), s.rowoff.as<int64_t>() + pl, (size_t)npl * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    int64_t at = 0;
    for (int64_t k = 0; k < npl; ++k) {
      const int64_t nc = hcols[pl + k];
      if (nc > 0 && last_rows)
        HIPCHK(c, hipMemcpyAsync(last_rows + 3 * at, s.rows.as<uint8_t>() + loff[(size_t)k], (size_t)3 * nc, hipMemcpyDeviceToHost, st));
      if (nc > 0 && last_mask)
        HIPCHK(c, hipMemcpyAsync(last_mask + at, s.mask.as<uint8_t>() + loff[(size_t)k] / 3, (size_t)nc, hipMemcpyDeviceToHost, st));
      at += nc;
    }
    HIPCHK(c, hipStreamSynchronize(st));
  }
  return ELECTOR_OK;
}

extern "C" int elector_msa_rows_wait(elector_ctx *c)
{
  if (!c) return ELECTOR_E_INVAL;
  std::lock_guard<std::mutex> lock(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  for (auto &s : c->st_slot)
    if (rows_fence(c, s)) return elector_fail(c, ELECTOR_E_HIP, "rows copy");
  return ELECTOR_OK;
}

extern "C" int elector_msa_stats_device(elector_ctx *c, int64_t n_windows, const uint8_t *d_cols, const int32_t *d_ncol,
                                        const int32_t *d_status, int64_t n_pieces, const int64_t *piece_first,
                                        int64_t n_reads, const int64_t *read_first, const int32_t *clips,
                                        int64_t *counters, int64_t *piece_cols, uint8_t *last_rows, uint8_t *last_mask,
                                        int64_t last_cap)
{
  if (!c) return ELECTOR_E_INVAL;
  if (c->st_inflight != 0) return elector_fail(c, ELECTOR_E_INVAL, "statistics jobs in flight: collect them first");
  if (n_pieces > 0 && !counters) return elector_fail(c, ELECTOR_E_INVAL, "bad arguments");
  const int rc = elector_msa_stats_enqueue(c, n_windows, d_cols, d_ncol, d_status, n_pieces, piece_first, n_reads, read_first, clips);
  if (rc) return rc;
  return elector_msa_stats_collect(c, n_pieces, counters, piece_cols, last_rows, last_mask, last_cap);
}

extern "C" int elector_msa_rows_fetch(elector_ctx *c, int64_t n_pieces, const int64_t *piece_cols, uint8_t *rows)
{
  if (!c) return ELECTOR_E_INVAL;
  if (n_pieces < 0 || (n_pieces > 0 && (!piece_cols || !rows))) return elector_fail(c, ELECTOR_E_INVAL, "bad arguments");
  std::lock_guard<std::mutex> lock(c->mu);
  if (c->st_last < 0 || n_pieces != c->st_slot[c->st_last].n_pieces)
    return elector_fail(c, ELECTOR_E_INVAL, "the pieces must be those of the context's last collected statistics job");
  if (n_pieces == 0) return ELECTOR_OK;
  elector::StatsSlot &s = c->st_slot[c->st_last];
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<int64_t> out_off((size_t)n_pieces + 1);
  out_off[0] = 0;
  for (int64_t p = 0; p < n_pieces; ++p) {
    if (piece_cols[p] < 0) return elector_fail(c, ELECTOR_E_INVAL, "negative column count");
    out_off[(size_t)p + 1] = out_off[(size_t)p] + 3 * piece_cols[p];
  }
  const int64_t total = out_off[(size_t)n_pieces];
  if (total == 0) return ELECTOR_OK;
  if (total > 3 * s.total) return elector_fail(c, ELECTOR_E_INVAL, "piece_cols exceed the job's columns");
  if (c->d_st_dense.ensure((size_t)total + 64) | c->d_st_outoff.ensure((size_t)(n_pieces + 1) * 8))
    return elector_fail(c, ELECTOR_E_NOMEM, "row staging");
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemcpyAsync(c->d_st_outoff.p, out_off.data(), (size_t)(n_pieces + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipStreamSynchronize(st));
  CompactArgs a{s.rows.as<uint8_t>(), s.rowoff.as<int64_t>(), s.cols.as<int64_t>(), c->d_st_outoff.as<int64_t>(),
                c->d_st_dense.as<uint8_t>()};
  hipLaunchKernelGGL(k_compact, dim3((unsigned)n_pieces), dim3(kStatsThreads), 0, st, a);
  HIPCHK(c, hipGetLastError());
  // to page-locked memory on the DMA engine (rows_dma.cpp; the writer threads of getPOA pass the context's pinned buffer),
  // anywhere else through the HIP runtime
  hipPointerAttribute_t at;
  bool pinned = hipPointerGetAttributes(&at, rows) == hipSuccess && at.type == hipMemoryTypeHost;
  if (!pinned) (void)hipGetLastError();
  if (pinned) {
    HIPCHK(c, hipStreamSynchronize(st));
    if (elector::rows_dma_start(c->device, rows, c->d_st_dense.p, (size_t)total, &c->fetch_sig) == 0) {
      if (elector::rows_dma_wait(c->fetch_sig)) return elector_fail(c, ELECTOR_E_HIP, "rows copy");
      return ELECTOR_OK;
    }
  }
  HIPCHK(c, hipMemcpyAsync(rows, c->d_st_dense.p, (size_t)total, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return ELECTOR_OK;
}

// The merged records of the context's last collected statistics job straight into msa.fa: rows device -> pinned
// host memory, formatted as Donatello's records (elector_msa_format, io_host.cpp) by `nthreads` host threads,
// one write() to `fd`.  Pieces with drop[p] != 0 are left out.  Returns the bytes written or a negative code.
extern "C" int64_t elector_msa_records_write(elector_ctx *c, int64_t n_pieces, const int64_t *piece_cols, const uint8_t *hdr,
                                              const int64_t *hdr_off, const uint8_t *drop, int fd, int nthreads)
{
  if (!c) return ELECTOR_E_INVAL;
  if (n_pieces < 0 || (n_pieces > 0 && (!piece_cols || !hdr || !hdr_off)) || fd < 0) return elector_fail(c, ELECTOR_E_INVAL, "bad arguments");
  if (n_pieces == 0) return 0;
  int64_t total = 0;
  for (int64_t p = 0; p < n_pieces; ++p) {
    if (piece_cols[p] < 0) return elector_fail(c, ELECTOR_E_INVAL, "negative column count");
    total += 3 * piece_cols[p];
  }
  {
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->h_rows.ensure((size_t)total + 64)) return elector_fail(c, ELECTOR_E_NOMEM, "pinned rows");
  }
  const int rc = elector_msa_rows_fetch(c, n_pieces, piece_cols, c->h_rows.as<uint8_t>());
  if (rc) return rc;
  const int64_t need = elector_msa_format(n_pieces, c->h_rows.as<uint8_t>(), piece_cols, hdr, hdr_off, drop, nullptr, 0, 1);
  if (need < 0) return elector_fail(c, (int)need, "records");
  if ((int64_t)c->h_text.size() < need) c->h_text.resize((size_t)need + (size_t)need / 8);
  const int64_t got = elector_msa_format(n_pieces, c->h_rows.as<uint8_t>(), piece_cols, hdr, hdr_off, drop, c->h_text.data(),
                                         (int64_t)c->h_text.size(), nthreads);
  if (got != need) return elector_fail(c, ELECTOR_E_INVAL, "records");
  int64_t at = 0;
  while (at < got) {
    const ssize_t w = ::write(fd, c->h_text.data() + at, (size_t)(got - at));
    if (w < 0) { if (errno == EINTR) continue; return elector_fail(c, ELECTOR_E_INVAL, "write to msa.fa failed"); }
    at += w;
  }
  return got;
}

// The same records written at a given offset of the file (a descriptor opened WITHOUT O_APPEND: the caller keeps
// the end-of-file position itself), by `nthreads` threads with one pwrite() each: the copy into the page cache is what
// a single write() of a 265 MB batch spends its time on, and it parallelises over pages.
extern "C" int64_t elector_msa_records_pwrite(elector_ctx *c, int64_t n_pieces, const int64_t *piece_cols, const uint8_t *hdr,
                                               const int64_t *hdr_off, const uint8_t *drop, int fd, int64_t offset, int nthreads)
{
  if (!c) return ELECTOR_E_INVAL;
  if (n_pieces < 0 || (n_pieces > 0 && (!piece_cols || !hdr || !hdr_off)) || fd < 0 || offset < 0) return elector_fail(c, ELECTOR_E_INVAL, "bad arguments");
  if (n_pieces == 0) return 0;
  int64_t total = 0;
  for (int64_t p = 0; p < n_pieces; ++p) {
    if (piece_cols[p] < 0) return elector_fail(c, ELECTOR_E_INVAL, "negative column count");
    total += 3 * piece_cols[p];
  }
  {
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->h_rows.ensure((size_t)total + 64)) return elector_fail(c, ELECTOR_E_NOMEM, "pinned rows");
  }
  static const bool prof = std::getenv("ELECTOR_DEBUG_HOST") != nullptr;
  auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
  const double t0 = now_ms();
  const int rc = elector_msa_rows_fetch(c, n_pieces, piece_cols, c->h_rows.as<uint8_t>());
  if (rc) return rc;
  const double t1 = now_ms();
  const int64_t need = elector_msa_format(n_pieces, c->h_rows.as<uint8_t>(), piece_cols, hdr, hdr_off, drop, nullptr, 0, 1);
  if (need < 0) return elector_fail(c, (int)need, "records");
  if ((int64_t)c->h_text.size() < need) c->h_text.resize((size_t)need + (size_t)need / 8);
  const int64_t got = elector_msa_format(n_pieces, c->h_rows.as<uint8_t>(), piece_cols, hdr, hdr_off, drop, c->h_text.data(),
                                         (int64_t)c->h_text.size(), nthreads);
  if (got != need) return elector_fail(c, ELECTOR_E_INVAL, "records");
  const double t2 = now_ms();
  // (buffered writes to one file go through one lock: the page cache takes ~10 GB/s from one thread or from sixteen, a
  // couple of threads keep it fed while other writers are in the same call; with sixteen each the end-to-end run was 10 % slower)
  static const int pw_max = std::getenv("ELECTOR_PWRITE_THREADS") ? std::max(1, std::atoi(std::getenv("ELECTOR_PWRITE_THREADS"))) : 2;
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(std::max(1, nthreads), pw_max), got >> 22));      // at least 4 MB per thread
  std::vector<int> bad((size_t)nt, 0);
  auto work = [&](int t) {
    int64_t at = got * t / nt;
    const int64_t end = got * (t + 1) / nt;
    while (at < end) {
      const ssize_t w = ::pwrite(fd, c->h_text.data() + at, (size_t)(end - at), (off_t)(offset + at));
      if (w < 0) { if (errno == EINTR) continue; bad[(size_t)t] = 1; return; }
      at += w;
    }
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
  }
  for (int t = 0; t < nt; ++t) if (bad[(size_t)t]) return elector_fail(c, ELECTOR_E_INVAL, "write to msa.fa failed");
  if (prof)
    std::fprintf(stderr, "[elector] records: rows to the host %.1f ms, format %.1f ms, pwrite x%d %.1f ms (%lld bytes)\n", t1 - t0, t2 - t1, nt,
                 now_ms() - t2, (long long)got);
  return got;
}

// ------------------------------------------------------- homopolymers (host) ---
// The reference keeps two growing lists (`reported`) and, when a homopolymer
// ends, picks the most frequent reference letter and measures its longest run
// in both lists (computeStats.py:344-363).  Here the lists are never stored:
// per candidate letter we keep its count, first position, current run and
// longest run in each list, which is all those lines read.  Ties in "most
// frequent" go to the letter seen first (the reference's tie order depends on
// Python's per-process string hashing).

namespace {

struct HomoState {
  // one entry per distinct byte seen in either list since the last reset
  struct Cand { uint8_t ch; int64_t count_r, first_r; int64_t cur_r, max_r, cur_c, max_c; };
  std::vector<Cand> cands;
  int64_t len = 0;
  uint8_t last_r = 0, last_c = 0;
  void reset(uint8_t r, uint8_t c) { cands.clear(); len = 0; push(r, c); }
  Cand &find(uint8_t ch)
  {
    for (auto &k : cands) if (k.ch == ch) return k;
    // first appearance in either list: every earlier non-gap letter differed, so all runs are 0
    cands.push_back({ch, 0, -1, 0, 0, 0, 0});
    return cands.back();
  }
  void push(uint8_t r, uint8_t c)
  {
    find(r); find(c);
    for (auto &k : cands) {
      if (k.ch == r) { if (k.count_r++ == 0) k.first_r = len; ++k.cur_r; if (k.cur_r > k.max_r) k.max_r = k.cur_r; }
      else if (r != '.') k.cur_r = 0;
      if (k.ch == c) { ++k.cur_c; if (k.cur_c > k.max_c) k.max_c = k.cur_c; }
      else if (c != '.') k.cur_c = 0;
    }
    ++len;
    last_r = r; last_c = c;
  }
};

}  // namespace

extern "C" int64_t elector_homopolymer_pairs(int64_t n_pieces, const uint8_t *rows, const int64_t *row_off,
                                             const int64_t *cols, const uint8_t *mask, int32_t threshold,
                                             int32_t *pairs, int64_t cap)
{
  if (n_pieces < 0 || (n_pieces > 0 && (!rows || !row_off || !cols || !mask))) return ELECTOR_E_INVAL;
  int64_t npairs = 0, moff = 0;
  for (int64_t p = 0; p < n_pieces; ++p) {
    const int64_t n = cols[p];
    if (n <= 10) { moff += n; continue; }
    const uint8_t *ref = rows + row_off[p], *cor = ref + n;
    const uint8_t *mk = mask + moff;
    HomoState st;
    st.reset('x', 'x');                                   // reported = [['x'],['x']] (:419)
    bool ok_to_report = false, end_ref = false;
    for (int64_t i = 0; i < n; ++i) {
      const uint8_t r = ref[i], c = cor[i];
      bool app_r = false, app_c = false, end_cor = true;
      if (mk[i]) {
        if (r != '.') {
          if (r == st.last_r) { app_r = true; if (st.len + 1 >= threshold) ok_to_report = true; }
          else if (ok_to_report) end_ref = true;
        }
        if (c != '.' && c == st.last_c) { app_c = true; end_cor = false; }
      }
      if (app_c || app_r) st.push(r, c);
      else if (!(end_ref && end_cor) && !end_ref && r != '.') st.reset(r, c);
      if (end_ref && end_cor) {
        // most frequent letter of the reference list, gaps only if nothing else is there
        const HomoState::Cand *best = nullptr, *best_ng = nullptr;
        for (auto &k : st.cands) {
          if (k.count_r == 0) continue;
          if (!best || k.count_r > best->count_r || (k.count_r == best->count_r && k.first_r < best->first_r)) best = &k;
          if (k.ch != '.' && (!best_ng || k.count_r > best_ng->count_r ||
                              (k.count_r == best_ng->count_r && k.first_r < best_ng->first_r)))
            best_ng = &k;
        }
        const HomoState::Cand *pick = (best && best->ch == '.') ? best_ng : best;
        if (pick) {
          if (npairs < cap && pairs) { pairs[2 * npairs] = (int32_t)pick->max_c; pairs[2 * npairs + 1] = (int32_t)pick->max_r; }
          ++npairs;
        }
        ok_to_report = false; end_ref = false;
        st.reset(r, c);
      }
    }
    moff += n;
  }
  return npairs;
}
