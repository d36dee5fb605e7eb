// elector_amd/csrc/split_dev.hip -- the window splitter on the device.
//
// What bin/masterSplitter does per read (src/split/Master_Splitter.cpp:175-332: split, best_split; :396-446 the
// read loop), restated for one workgroup per read:
//   * 2-bit k-mer codes of every position, with the reference's two letter maps (str2num for the first k
//     letters, :26-38; the rolling update for the rest, :41-49);
//   * three hash tables (k-mers unique in the reference read; of those, unique in the uncorrected read; of those,
//     unique in the corrected read) -- "unique" does not depend on insertion order.  Reads of up to 12.5 kb: 16-bit
//     slots in LDS, checked against a 2-bit packed copy of the sequences (tables_lds); reads of up to 123 kb: 32-bit
//     slots in LDS, one partition of the k-mer space at a time (tables_long); longer ones: 64-bit slots in HBM;
//     every table phase a FLAT loop (one probe per lane and turn, look-ups and insertions in separate loops);
//   * anchors: the greedy "more than minSize after the last one" pass over the candidate bitmap as per-word exit
//     tables chained by fixed-point rounds (anchors_bitmap); the longest chain with steps < 1000 in all three reads
//     (:79-126) back to front by one wavefront, every finished anchor offering itself to the waiting ones; while
//     that wavefront works, the others prepare best_split's next round; windows from the chain by hopping along
//     per-anchor links, with the reference's re-split of a missing start / end of the corrected read
//     (:268-277,295-301) as a second, workgroup-wide pass;
//   * best_split's loop over k = 15, 13, 11, 9 while the largest fragment shrinks (:310-332);
//   * the reads of a batch drawn from a counter, longest first, in up to three launches side by side (short / medium / long reads).
// A read whose anchors or windows do not fit the arrays sized for its launch gets a second try with an anchor per
// base; one that still does not fit is flagged and the batch split by the host code (splitter.cpp) instead: same
// result, never silently different.
// Byte / integer work, latency-bound; no MFMA.  The unsigned-conversion quirks of the reference that decide
// which windows come out are kept (marked "ref:").
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <thread>
#include <type_traits>
#include <vector>

#include "elector_poa.h"
#include "elector_split.h"
#include "ctx.h"

namespace elector {

#ifndef ELECTOR_SPLIT_THREADS
#define ELECTOR_SPLIT_THREADS 512
#endif
constexpr int kSplitThreads = ELECTOR_SPLIT_THREADS;
static_assert(kSplitThreads >= 128 && kSplitThreads % 64 == 0, "wavefront 0 and at least one more (split_core: the next round's preparation)");
constexpr int kMaxAnchors = 3000;          // LDS: 5 ints per anchor

struct DSeq { int64_t base; uint32_t n; };   // base: byte offset into the reads buffer

// v, known to be the same in every lane of the wavefront, as a value the compiler knows to be uniform
template <class T>
__device__ __forceinline__ T uniform(const T &v)
{
  static_assert(sizeof(T) % 4 == 0, "uniform(): whole dwords");
  uint32_t w[sizeof(T) / 4];
  __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
  for (size_t i = 0; i < sizeof(T) / 4; ++i) w[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)w[i]);
  T r;
  __builtin_memcpy(&r, w, sizeof(T));
  return r;
}

struct SplitArgs {
  int64_t n_reads;
  const uint8_t *reads;
  const int64_t *read_off;
  const int32_t *hdr_len;
  double thr;
  unsigned long long *ent;       // [block][3][tab_cap] table entries
  int32_t *ca; int32_t *cb; int32_t *wl;
  int64_t tab_cap, maxlen, maxwin;
  int32_t *out_win;              // [sum of caps][8]
  const int64_t *out_first;      // per read: first window slot
  int32_t *out_cnt, *out_kind;   // kind: 0 windows, 1 small, 2 wrong, -1 skipped, -2 host fallback
  int32_t *anc; int64_t maxanc;
  unsigned long long *stamps;
  int lds_tables;                // the dynamic LDS holds the on-chip tables behind the anchor arrays
  int lds_long;                  // the dynamic LDS holds the long reads' on-chip tables (tables_long): 1 first size only, 2 both
  const int32_t *order;          // the reads, longest first: the order the workgroups take them in
  int32_t *next;                 // ... through this counter
};

// window record (8 ints): ref off, ref len, S1 off, S1 len, S2 off, S2 len, S2 is the 'N' filler, unused;
// offsets relative to the start of the read's own sequence
__device__ __forceinline__ int ldg(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stg(int32_t *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the same through a pointer the compiler knows to be global memory: a store through a generic pointer is a FLAT
// instruction, which also counts in lgkmcnt -- the next LDS read then waits for the store's trip to memory
typedef __attribute__((address_space(1))) int32_t global_i32;
__device__ __forceinline__ void stg_global(int32_t *p, int v)
{
  __hip_atomic_store((global_i32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t map1(uint8_t c) { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u; }   // ref: str2num
__device__ __forceinline__ uint32_t map2(uint8_t c) { return c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u; }   // ref: rolling update

constexpr int kU = 4;                         // table operations a thread keeps in flight
constexpr unsigned long long kEmptyEnt = ~0ull;

__device__ __forceinline__ unsigned long long ld64(const unsigned long long *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// open-addressing table in HBM, one 64-bit entry per slot: k-mer code in the high word, in the low word the
// position of its only occurrence so far, or -1 once it has been seen twice.  One compare-and-swap per insertion;
// a thread issues kU of them (or kU look-ups) before it waits for the first -- every one is a trip to L2 or HBM.
struct Tab {
  unsigned long long *ent; uint32_t mask;
  __device__ __forceinline__ static unsigned long long pack(uint32_t key, int pos) { return ((unsigned long long)key << 32) | (uint32_t)pos; }
  __device__ __forceinline__ void add(const uint32_t (&key)[kU], const uint32_t (&pos)[kU], const bool (&on)[kU]) const
  {
    uint32_t h[kU];
    unsigned long long prev[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      h[u] = (key[u] * 2654435761u) & mask;
      prev[u] = kEmptyEnt;
      if (on[u]) prev[u] = atomicCAS(ent + h[u], kEmptyEnt, pack(key[u], (int)pos[u]));
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (!on[u]) continue;
      while (prev[u] != kEmptyEnt) {
        if ((uint32_t)(prev[u] >> 32) == key[u]) { atomicOr(ent + h[u], 0xFFFFFFFFull); break; }   // a second occurrence: repeated
        h[u] = (h[u] + 1) & mask;
        prev[u] = atomicCAS(ent + h[u], kEmptyEnt, pack(key[u], (int)pos[u]));
      }
    }
  }
  // position of the k-mer when it occurs exactly once, else -1
  __device__ __forceinline__ void unique_pos(const uint32_t (&key)[kU], const bool (&on)[kU], int (&out)[kU]) const
  {
    uint32_t h[kU];
    unsigned long long e[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      h[u] = (key[u] * 2654435761u) & mask;
      e[u] = kEmptyEnt;
      if (on[u]) e[u] = ld64(ent + h[u]);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      out[u] = -1;
      if (!on[u]) continue;
      while (e[u] != kEmptyEnt) {
        if ((uint32_t)(e[u] >> 32) == key[u]) { out[u] = (int)(uint32_t)e[u]; break; }
        h[u] = (h[u] + 1) & mask;
        e[u] = ld64(ent + h[u]);
      }
    }
  }
};

// anchors and chain of one split() level: five int arrays of `cap` entries -- in LDS for the usual reads, in the
// workgroup's HBM scratch for reads with more than kMaxAnchors possible anchors (written and read by this
// workgroup only) -- and four scalars in LDS
struct LvlState { int n, start, nchain, fail; };
// The anchor arrays are reached through accessors rather than plain pointers: a pointer that travels through a
// struct into a function that is not inlined is a generic one, and every access through it a FLAT instruction --
// for LDS several times slower than ds_read / ds_write.  LdsArr rebuilds the address from the dynamic-LDS symbol.
template <class T>
struct LdsArr {
  using elem = T;
  uint32_t off;                   // in units of T from the start of the dynamic LDS
  __device__ __forceinline__ T &operator[](int i) const
  {
    extern __shared__ int32_t dyn_lds_[];
    return reinterpret_cast<T *>(dyn_lds_)[off + (uint32_t)i];
  }
};
template <class T>
struct GlbArr {
  using elem = T;
  T *p;
  __device__ __forceinline__ T &operator[](int i) const { return p[i]; }
};
template <class A>
struct LvlT {                     // uint16_t entries in LDS (reads below 64 kb), int32_t in HBM
  using elem = typename A::elem;
  A ar, aa, ab, cl, cn;
  int cap;
  LvlState *s;
  static constexpr int kNoNext = (int)(elem)(-1);     // end of a chain in cn
};
using Lvl16 = LvlT<LdsArr<uint16_t>>;
using Lvl32 = LvlT<GlbArr<int32_t>>;

struct WG {
  const uint8_t *reads;
  Tab tab[3];
  int32_t *ca, *cb;
  int32_t *wl;                     // [3][maxwin][8]
  int64_t tab_cap, maxwin;
  unsigned long long *stamps;     // debug (ELECTOR_DEBUG_SPLIT): cycles per phase, summed over reads
  int lds_tab;                    // LDS tables (tables_lds): dword offset in the dynamic LDS, or -1
  int lds_long;                   // LDS tables for reads of up to 65,535 bases (tables_long): dword offset, or -1
  int long_b;                     // ... and the second size, up to 122,879 bases
};

#define SP_STAMP(idx)                                                                      \
  do {                                                                                     \
    if (g.stamps && threadIdx.x == 0) {                                                    \
      const unsigned long long now_ = __builtin_readcyclecounter();                        \
      atomicAdd(g.stamps + (idx), now_ - sp_t_);                                           \
      sp_t_ = now_;                                                                        \
    }                                                                                      \
  } while (0)

// k-mer code at position p of s (p + k <= n, or p == 0 for a sequence shorter than k): the first k letters of the
// sequence go through map1, the others through map2
__device__ __forceinline__ uint32_t code_at(const uint8_t *s, uint32_t n, uint32_t p, int k)
{
  uint32_t r = 0;
  const uint32_t m = min(n - p, (uint32_t)k);
  for (uint32_t i = 0; i < m; ++i) {
    const uint32_t q = p + i;
    r = (r << 2) | (q < (uint32_t)k ? map1(s[q]) : map2(s[q]));
  }
  return r & ((1u << (2 * k)) - 1u);
}

// number of k-mer positions the reference visits: the first k-mer always, then one per j with j + k < n
__device__ __forceinline__ uint32_t n_kmers(uint32_t n, int k) { return n > (uint32_t)k ? n - (uint32_t)k + 1u : 1u; }

__device__ void reset_tab(const Tab &t, int64_t cap)
{
  for (int64_t i = threadIdx.x; i < cap; i += kSplitThreads) t.ent[i] = kEmptyEnt;
}

// ---- the same three tables in LDS, for reads of up to kLdsMaxN bases ----
// A 64-bit entry per slot does not fit on chip; a 16-bit one does: [15] seen twice, [14:0] position of the first
// occurrence, 0xFFFF empty -- the k-mer itself is read back from a 2-bit packed copy of the sequence (16 letters
// per dword, letter q coded as the reference codes it: first k letters by str2num, the rest by the rolling update).
// The reference's table gets 16,384 slots (most look-ups of uncorrected k-mers miss: short probe sequences matter),
// the other two 8,192 each -- they only hold k-mers shared with the reference -- and the third lives where the first
// was (nothing looks the reference's k-mers up once the second table is built), which with the 16-bit anchor arrays
// is about 72 KB: two workgroups per CU.  When they fill beyond kLdsFill (an uncorrected read that is nearly
// error-free) the call falls back to the HBM tables.
constexpr uint32_t kLdsMaxN = 12500;
constexpr uint32_t kLdsCapRef = 16384, kLdsCapOther = 8192, kLdsFill = 6400;
constexpr uint32_t kLdsSeqWords = kLdsMaxN / 16 + 3;
constexpr uint32_t kLdsBitWords = 2 * ((kLdsMaxN + 63) / 64 + 1);       // candidate bitmap: one bit per reference position, as dwords
constexpr uint32_t kLdsAncTab = 6 * (kLdsBitWords / 2);                   // per bitmap word: the anchor walk's exit table (anchors_lds), six dwords,
static_assert(kLdsCapOther * 2 + kLdsAncTab * 4 <= kLdsCapRef * 2, "the third table and the exit tables overlay the reference's k-mer table");   // which is dead by then
constexpr size_t kLdsTabBytes = (size_t)(kLdsCapRef + kLdsCapOther) * 2 + 3 * (size_t)kLdsSeqWords * 4 + (size_t)kLdsBitWords * 4 + 16 + 40;   // + four words (fill counter, what is prepared for which k, a barrier count, how far) and the key of the packed sequences

constexpr uint32_t kLdsKeyWord = (kLdsCapRef + kLdsCapOther) / 2 + 3 * kLdsSeqWords + kLdsBitWords + 4;   // dwords from the tables' start

struct LTab {
  uint32_t *w;            // slots, two per word
  uint32_t mask;
  const uint32_t *seq;    // packed sequence the positions refer to
  uint32_t kmsk;
  __device__ __forceinline__ static uint32_t bits(const uint32_t *seq, uint32_t p, uint32_t kmsk)
  {
    const uint32_t bit = 2u * p, i = bit >> 5, sh = bit & 31u;
    const unsigned long long v = ((unsigned long long)seq[i + 1] << 32) | seq[i];
    return (uint32_t)(v >> sh) & kmsk;
  }
  __device__ __forceinline__ static uint32_t slot_of(uint32_t code, uint32_t mask) { return ((code * 2654435761u) >> 7) & mask; }
  // Double hashing: the probe sequence of a k-mer advances by an odd step of its own (the table sizes are powers of
  // two, so it visits every slot).  With linear probing at half load the probe chains cluster, and a wavefront
  // waits for the longest chain among its 64 look-ups -- which of the k-mers are unique does not depend on the
  // probe order, so this changes no result.
  __device__ __forceinline__ static uint32_t step_of(uint32_t code) { return ((code * 0x9E3779B1u) >> 15) | 1u; }
  // true when the k-mer got a slot of its own (a new entry)
  __device__ __forceinline__ bool add(uint32_t code, uint32_t pos) const
  {
    uint32_t h = slot_of(code, mask);
    const uint32_t step = step_of(code);
    for (uint32_t probes = 0; probes <= mask; ++probes) {
      uint32_t *word = w + (h >> 1);
      const uint32_t sh = (h & 1u) * 16u;
      for (;;) {
        const uint32_t old = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t cur = (old >> sh) & 0xFFFFu;
        if (cur == 0xFFFFu) {
          if (atomicCAS(word, old, (old & ~(0xFFFFu << sh)) | (pos << sh)) == old) return true;
          continue;                                            // the word changed under us: look again
        }
        if (bits(seq, cur & 0x7FFFu, kmsk) == code) {          // a second occurrence: repeated
          if (!(cur & 0x8000u)) atomicOr(word, 0x8000u << sh);
          return false;
        }
        break;
      }
      h = (h + step) & mask;
    }
    return false;
  }
  // position of the k-mer when it occurs exactly once, else -1
  __device__ __forceinline__ int find(uint32_t code) const
  {
    uint32_t h = slot_of(code, mask);
    const uint32_t step = step_of(code);
    for (uint32_t probes = 0; probes <= mask; ++probes) {
      const uint32_t e = (w[h >> 1] >> ((h & 1u) * 16u)) & 0xFFFFu;
      if (e == 0xFFFFu) return -1;
      if (bits(seq, e & 0x7FFFu, kmsk) == code) return (e & 0x8000u) ? -1 : (int)(e & 0x7FFFu);
      h = (h + step) & mask;
    }
    return -1;
  }
};

// Table phases as FLAT loops.  A thread has some sixteen k-mers to look up per phase; written as "for every k-mer:
// probe until found", a wavefront stays in the inner loop until the longest of its 64 probe chains ends (about seven
// probes at half load) before any lane may take its next k-mer: sixteen times seven probe rounds.  Here every lane is
// a small state machine -- one probe per loop turn, and a lane whose look-up ends takes its next k-mer in that very
// turn -- so a wavefront runs for the longest SUM of probe chains among its lanes, about sixteen times two.  For the
// same reason a phase that looks up and then inserts does so in two loops: an insertion nested in the look-up's turn
// (a probe loop of its own, with a CAS) was paid by the whole wavefront in nearly every turn.
// A lane's k-mers are those at positions tid + q * kSplitThreads; which of them a loop handles is a mask over q.
static_assert(kLdsMaxN <= 32u * (uint32_t)kSplitThreads, "a lane's positions as one 32-bit mask");
// (me, stride: the thread's number among the threads that share the work and how many they are -- the whole
// workgroup, or the seven wavefronts that prepare the next pass during the chain phase)
__device__ __forceinline__ uint32_t lane_positions(uint32_t np, uint32_t me, uint32_t stride)
{
  if (me >= np) return 0u;
  const uint32_t nq = (np - me + stride - 1u) / stride;
  return nq >= 32u ? 0xFFFFFFFFu : (1u << nq) - 1u;
}
__device__ __forceinline__ uint32_t lane_positions(uint32_t np) { return lane_positions(np, threadIdx.x, (uint32_t)kSplitThreads); }

// the lane's k-mers of seq named by todo that occur exactly once in t -> mask of the same kind
__device__ __forceinline__ uint32_t flat_find(const LTab &t, const uint32_t *seq, uint32_t todo, uint32_t kmsk, unsigned long long *turns = nullptr,
                                             uint32_t tid = threadIdx.x, uint32_t stride = (uint32_t)kSplitThreads)
{
  uint32_t hits = 0, nturn = 0;
  bool active = todo != 0u;
  uint32_t q = active ? (uint32_t)__builtin_ctz(todo) : 0u;
  uint32_t code = active ? LTab::bits(seq, tid + q * stride, kmsk) : 0u;
  // the code of the lane's next k-mer is fetched a look-up ahead: its LDS read is not on the path of the turn that
  // ends a look-up
  uint32_t rest = todo & (todo - 1u);
  uint32_t code_next = rest ? LTab::bits(seq, tid + (uint32_t)__builtin_ctz(rest) * stride, kmsk) : 0u;
  uint32_t h = LTab::slot_of(code, t.mask), step = LTab::step_of(code), probes = 0;
  while (__builtin_amdgcn_ballot_w64(active) != 0) {
    ++nturn;
    if (active) {
      const uint32_t e = (t.w[h >> 1] >> ((h & 1u) * 16u)) & 0xFFFFu;
      int res = -2;                                                  // -2: go on probing
      if (e == 0xFFFFu || probes > t.mask) res = -1;
      else if (LTab::bits(t.seq, e & 0x7FFFu, t.kmsk) == code) res = (e & 0x8000u) ? -1 : (int)(e & 0x7FFFu);
      if (res == -2) { h = (h + step) & t.mask; ++probes; }
      else {
        if (res >= 0) hits |= 1u << q;
        todo = rest;
        active = todo != 0u;
        q = active ? (uint32_t)__builtin_ctz(todo) : 0u;
        code = code_next;
        h = LTab::slot_of(code, t.mask); step = LTab::step_of(code); probes = 0;
        rest = todo & (todo - 1u);
        code_next = rest ? LTab::bits(seq, tid + (uint32_t)__builtin_ctz(rest) * stride, kmsk) : 0u;
      }
    }
  }
  if (turns && threadIdx.x == 0) atomicAdd(turns, (unsigned long long)nturn);
  return hits;
}

// the lane's k-mers of t's own sequence named by todo into t (LTab::add as a flat loop).  fill (LDS) counts the new
// entries when given; once it passes limit every lane stops (the caller then gives the tables up), looked at every
// eighth turn -- at most a few hundred entries late, and a table that has filled up meanwhile only costs those turns.
__device__ __forceinline__ void flat_add(const LTab &t, uint32_t todo, int *fill = nullptr, int limit = 0, uint32_t tid = threadIdx.x,
                                         uint32_t stride = (uint32_t)kSplitThreads)
{
  bool active = todo != 0u;
  uint32_t p = active ? tid + (uint32_t)__builtin_ctz(todo) * stride : 0u;
  uint32_t code = active ? LTab::bits(t.seq, p, t.kmsk) : 0u;
  uint32_t rest = todo & (todo - 1u);
  uint32_t code_next = rest ? LTab::bits(t.seq, tid + (uint32_t)__builtin_ctz(rest) * stride, t.kmsk) : 0u;
  uint32_t h = LTab::slot_of(code, t.mask), step = LTab::step_of(code), probes = 0, nturn = 0;
  while (__builtin_amdgcn_ballot_w64(active) != 0) {
    if (active) {
      uint32_t *word = t.w + (h >> 1);
      const uint32_t sh = (h & 1u) * 16u;
      const uint32_t old = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t cur = (old >> sh) & 0xFFFFu;
      bool fin = false;
      if (cur == 0xFFFFu) {
        fin = atomicCAS(word, old, (old & ~(0xFFFFu << sh)) | (p << sh)) == old;   // lost the race: look at the slot again
        if (fin && fill) atomicAdd(fill, 1);
      } else if (LTab::bits(t.seq, cur & 0x7FFFu, t.kmsk) == code) { if (!(cur & 0x8000u)) atomicOr(word, 0x8000u << sh); fin = true; }
      else { h = (h + step) & t.mask; fin = ++probes > t.mask; }
      if (fin) {
        todo = rest;
        active = todo != 0u;
        p = active ? tid + (uint32_t)__builtin_ctz(todo) * stride : 0u;
        code = code_next;
        h = LTab::slot_of(code, t.mask); step = LTab::step_of(code); probes = 0;
        rest = todo & (todo - 1u);
        code_next = rest ? LTab::bits(t.seq, tid + (uint32_t)__builtin_ctz(rest) * stride, t.kmsk) : 0u;
      }
    }
    if (fill && (++nturn & 7u) == 0u && __builtin_amdgcn_readfirstlane(*(volatile int *)fill) > limit) break;
  }
}

// ---- on-chip tables for reads of up to 65,535 bases ----
// The HBM tables cost a read of 50 kb some ten million cycles per split() pass: a million outstanding compare-and-swaps on
// random 64-byte lines.  Which k-mers are shared and unique can be decided one PARTITION of the k-mer space at a time
// -- a k-mer belongs to one partition (a hash of its code), and the three tables of a partition never meet another
// partition's k-mers -- so the whole pipeline (reference table, uncorrected look-ups -> second table, corrected
// look-ups -> third table, candidates) runs P times over tables that fit in LDS, P chosen so that a partition holds
// at most ~6,000 k-mers of a read.  32-bit slots ([31] seen twice, [30:0] position, all ones empty), the k-mer read
// back from the 2-bit packed sequences, which also live in LDS (3 x 16 KB); the candidates go to the dense arrays
// ca / cb in HBM that the HBM tables write, and everything behind (anchor walk, chain) is shared with that path.
// One workgroup per CU (148 KB of LDS).
// Two sizes: reads of up to 65,535 bases with tables of 16,384 / 8,192 slots (a partition holds up to ~6,000 k-mers),
// reads of up to 122,879 bases with half the tables and twice the partitions -- the packed sequences take the room.
template <int NW_, uint32_t MAXN_, uint32_t CAPREF_, uint32_t CAPOTHER_, uint32_t FILL_, uint32_t PART_, bool S16_ = false>
struct LongCfg {
  static constexpr bool kS16 = S16_;                               // 16-bit slots (positions below 32,768), two to a dword
  static constexpr uint32_t kRefWords = CAPREF_ / (S16_ ? 2 : 1), kOtherWords = CAPOTHER_ / (S16_ ? 2 : 1);
  static constexpr int NW = NW_;                                   // 32-bit words of a lane's position mask
  static constexpr uint32_t kMaxN = MAXN_, kCapRef = CAPREF_, kCapOther = CAPOTHER_, kFill = FILL_, kPart = PART_;
  static constexpr uint32_t kSeqWords = MAXN_ / 16 + 3;
  static constexpr uint32_t kBitWords = 2 * ((MAXN_ + 63) / 64 + 1);                  // candidate bitmap, as dwords
  static constexpr uint32_t kBmWord = kRefWords + kOtherWords + 3 * kSeqWords + 2;     // dwords from the region's start
  static constexpr uint32_t kDwords = kBmWord + kBitWords;
  static_assert(MAXN_ <= 32u * NW_ * (uint32_t)kSplitThreads, "a lane's positions as one mask");
  static_assert(6 * (kBitWords / 2) <= kRefWords + kOtherWords, "the exit tables of the anchor walk overlay the dead tables");
  static_assert(!S16_ || MAXN_ <= 32767u, "a position and the seen-twice bit in 16 bits");
  static_assert(kDwords * 4 <= 160 * 1024 - 1024, "one workgroup's LDS");
};
using LongS = LongCfg<2, 32767u, 16384u, 8192u, 7200u, 6000u, true>;    // reads of up to 32,767 bases: 16-bit slots, 78 KB, two workgroups per CU
using LongA = LongCfg<4, 65535u, 16384u, 8192u, 7200u, 6000u>;
using LongB = LongCfg<8, 122879u, 8192u, 4096u, 3600u, 3000u>;
constexpr uint32_t kLongMaxN = LongB::kMaxN;
constexpr size_t kLongBytes = (size_t)(LongA::kDwords > LongB::kDwords ? LongA::kDwords : LongB::kDwords) * 4;
constexpr size_t kLongSBytes = (size_t)LongS::kDwords * 4;
static_assert(2 * (kLongSBytes + 128) <= 160 * 1024, "two workgroups of the medium class per CU");

template <int NW>
struct MaskN {
  uint32_t m[NW];
  __device__ __forceinline__ bool any() const
  {
    uint32_t o = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) o |= m[w];
    return o != 0u;
  }
  __device__ __forceinline__ uint32_t first() const                    // index of the lowest bit (any())
  {
    uint32_t r = 0;
#pragma unroll
    for (int w = NW - 1; w >= 0; --w) if (m[w]) r = 32u * (uint32_t)w + (uint32_t)__builtin_ctz(m[w]);
    return r;
  }
  __device__ __forceinline__ void drop_first()
  {
    bool done = false;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const bool here = !done && m[w] != 0u;
      if (here) m[w] &= m[w] - 1u;
      done = done || here;
    }
  }
  __device__ __forceinline__ void set(uint32_t q)
  {
    const uint32_t b = 1u << (q & 31u);
#pragma unroll
    for (int w = 0; w < NW; ++w) if ((q >> 5) == (uint32_t)w) m[w] |= b;
  }
};

// a partition's table: 32-bit slots ([31] seen twice, [30:0] position, all ones empty), or 16-bit ones, two to a dword
// ([15] seen twice, [14:0] position, 0xFFFF empty) -- the accessors below give both the same face
template <bool S16>
struct LTabP {
  uint32_t *w;
  uint32_t mask;
  const uint32_t *seq;    // packed sequence the positions refer to
  uint32_t kmsk;
  static constexpr uint32_t kEmpty = 0xFFFFFFFFu, kTwice = 0x80000000u, kPos = 0x7FFFFFFFu;
  __device__ __forceinline__ uint32_t get(uint32_t h) const             // the slot in the 32-bit form
  {
    if constexpr (S16) {
      const uint32_t e = (__hip_atomic_load(w + (h >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> ((h & 1u) * 16u)) & 0xFFFFu;
      return e == 0xFFFFu ? kEmpty : ((e & 0x8000u) << 16) | (e & 0x7FFFu);
    } else return __hip_atomic_load(w + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ bool claim(uint32_t h, uint32_t pos) const   // an empty slot takes pos; false: look at the slot again
  {
    if constexpr (S16) {
      uint32_t *word = w + (h >> 1);
      const uint32_t sh = (h & 1u) * 16u;
      const uint32_t old = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (((old >> sh) & 0xFFFFu) != 0xFFFFu) return false;
      return atomicCAS(word, old, (old & ~(0xFFFFu << sh)) | (pos << sh)) == old;
    } else return atomicCAS(w + h, kEmpty, pos) == kEmpty;
  }
  __device__ __forceinline__ void twice(uint32_t h) const
  {
    if constexpr (S16) atomicOr(w + (h >> 1), 0x8000u << ((h & 1u) * 16u));
    else atomicOr(w + h, kTwice);
  }
  __device__ __forceinline__ int find(uint32_t code) const              // position of the k-mer when it occurs exactly once, else -1
  {
    uint32_t h = LTab::slot_of(code, mask);
    const uint32_t step = LTab::step_of(code);
    for (uint32_t probes = 0; probes <= mask; ++probes) {
      const uint32_t e = get(h);
      if (e == kEmpty) return -1;
      if (LTab::bits(seq, e & kPos, kmsk) == code) return (e & kTwice) ? -1 : (int)e;
      h = (h + step) & mask;
    }
    return -1;
  }
};

// which partition of the k-mer space a code belongs to (lg: log2 of the number of partitions); a multiplier of its
// own, so that the slots of a partition's k-mers are spread over the whole table
__device__ __forceinline__ uint32_t part_of(uint32_t code, uint32_t lg) { return lg ? (code * 0x85EBCA6Bu) >> (32u - lg) : 0u; }

// In the long reads' tables a lane's positions are CONTIGUOUS (base .. base + cnt - 1, bit q of a mask is position
// base + q): which of them lie in partition part is then a matter of sliding a window over the packed sequence -- a
// shift, a multiply and a compare per position, one LDS read per eight -- instead of two LDS reads per position and
// pass; with P partitions every position is looked at P times.
template <int NW>
__device__ __forceinline__ MaskN<NW> lane_positions_of(const uint32_t *seq, uint32_t base, uint32_t cnt, uint32_t kmsk, uint32_t part, uint32_t lg)
{
  MaskN<NW> r;
#pragma unroll
  for (int wd = 0; wd < NW; ++wd) {
    uint32_t bitsw = 0;
#pragma nounroll
    for (uint32_t b0 = 0; b0 < 32u && 32u * (uint32_t)wd + b0 < cnt; b0 += 8u) {
      const uint32_t p0 = base + 32u * (uint32_t)wd + b0, bit = 2u * p0, i = bit >> 5, sh = bit & 31u;
      // 46 bits from position p0 on: eight positions of up to 30 bits each, two bits apart
      unsigned long long win = ((((unsigned long long)seq[i + 1] << 32) | seq[i]) >> sh) | (sh ? (unsigned long long)seq[i + 2] << (64u - sh) : 0ull);
#pragma unroll
      for (uint32_t j = 0; j < 8u; ++j) {
        if (32u * (uint32_t)wd + b0 + j < cnt && part_of((uint32_t)win & kmsk, lg) == part) bitsw |= 1u << (b0 + j);
        win >>= 2;
      }
    }
    r.m[wd] = bitsw;
  }
  return r;
}

// flat loops (see flat_find / flat_add) on a 32-bit table and a lane's position mask
// (found(q, position in the table's sequence) is called for every k-mer that occurs exactly once)
template <int NW, class TAB, class Found>
__device__ __forceinline__ MaskN<NW> flat_find32(const TAB &t, const uint32_t *seq, uint32_t base, MaskN<NW> todo, uint32_t kmsk, Found &&found)
{
  MaskN<NW> hits;
#pragma unroll
  for (int w = 0; w < NW; ++w) hits.m[w] = 0u;
  bool active = todo.any();
  uint32_t q = active ? todo.first() : 0u;
  uint32_t code = active ? LTab::bits(seq, base + q, kmsk) : 0u;
  uint32_t h = LTab::slot_of(code, t.mask), step = LTab::step_of(code), probes = 0;
  while (__builtin_amdgcn_ballot_w64(active) != 0) {
    if (active) {
      const uint32_t e = t.get(h);
      int res = -2;                                                  // -2: go on probing
      if (e == TAB::kEmpty || probes > t.mask) res = -1;
      else if (LTab::bits(t.seq, e & TAB::kPos, t.kmsk) == code) res = (e & TAB::kTwice) ? -1 : 0;
      if (res == -2) { h = (h + step) & t.mask; ++probes; }
      else {
        if (res >= 0) { hits.set(q); found(q, e & TAB::kPos); }
        todo.drop_first();
        active = todo.any();
        q = active ? todo.first() : 0u;
        code = active ? LTab::bits(seq, base + q, kmsk) : 0u;
        h = LTab::slot_of(code, t.mask); step = LTab::step_of(code); probes = 0;
      }
    }
  }
  return hits;
}
template <int NW, class TAB>
__device__ __forceinline__ MaskN<NW> flat_find32(const TAB &t, const uint32_t *seq, uint32_t base, MaskN<NW> todo, uint32_t kmsk)
{
  return flat_find32<NW>(t, seq, base, todo, kmsk, [](uint32_t, uint32_t) {});
}

template <int NW, class TAB>
__device__ __forceinline__ void flat_add32(const TAB &t, uint32_t base, MaskN<NW> todo, int *fill = nullptr, int limit = 0)
{
  bool active = todo.any();
  uint32_t p = active ? base + todo.first() : 0u;
  uint32_t code = active ? LTab::bits(t.seq, p, t.kmsk) : 0u;
  uint32_t h = LTab::slot_of(code, t.mask), step = LTab::step_of(code), probes = 0, nturn = 0;
  while (__builtin_amdgcn_ballot_w64(active) != 0) {
    if (active) {
      const uint32_t cur = t.get(h);
      bool fin = false;
      if (cur == TAB::kEmpty) {
        fin = t.claim(h, p);                                         // lost the race: look at the slot again
        if (fin && fill) atomicAdd(fill, 1);
      } else if (LTab::bits(t.seq, cur & TAB::kPos, t.kmsk) == code) { if (!(cur & TAB::kTwice)) t.twice(h); fin = true; }
      else { h = (h + step) & t.mask; fin = ++probes > t.mask; }
      if (fin) {
        todo.drop_first();
        active = todo.any();
        p = active ? base + todo.first() : 0u;
        code = active ? LTab::bits(t.seq, p, t.kmsk) : 0u;
        h = LTab::slot_of(code, t.mask); step = LTab::step_of(code); probes = 0;
      }
    }
    if (fill && (++nturn & 7u) == 0u && __builtin_amdgcn_readfirstlane(*(volatile int *)fill) > limit) break;
  }
}

// the table phases of split_core for a read of up to C::kMaxN bases: ca / cb written for the candidates as the HBM
// tables write them, the candidates as a bitmap in LDS for the anchor walk; false (uniform) when a partition's
// second table filled up (the HBM tables then take the call)
template <class C>
__device__ bool tables_long(const WG &g, unsigned long long &sp_t_, int lds_off, const uint8_t *pr, uint32_t nr,
                            const uint8_t *p1, uint32_t n1, const uint8_t *p2, uint32_t n2, int k)
{
  constexpr int NW = C::NW;
  const int tid = threadIdx.x;
  extern __shared__ int32_t dyn_lds_[];
  uint32_t *lds = reinterpret_cast<uint32_t *>(dyn_lds_) + lds_off;
  uint32_t *wr = lds, *w1 = wr + C::kRefWords, *w2 = wr;         // the third table takes the place of the first, which is dead by then
  uint32_t *sr = w1 + C::kOtherWords, *s1 = sr + C::kSeqWords, *s2 = s1 + C::kSeqWords;
  int *flag = reinterpret_cast<int *>(s2 + C::kSeqWords);
  uint32_t *bm = lds + C::kBmWord;                                // the candidates as a bitmap, for the anchor walk
  const uint32_t kmsk = (1u << (2 * k)) - 1u;
  auto pack = [&](const uint8_t *s, uint32_t n, uint32_t *dst) {
    const uint32_t nw = (n + 15) / 16 + 2;
    for (uint32_t wdx = tid; wdx < nw; wdx += kSplitThreads) {
      uint8_t ch[16];
#pragma unroll
      for (uint32_t i = 0; i < 16; ++i) { const uint32_t q = 16 * wdx + i; ch[i] = s[q < n ? q : 0]; }   // all sixteen loads in flight
      uint32_t v = 0;
#pragma unroll
      for (uint32_t i = 0; i < 16; ++i) {
        const uint32_t q = 16 * wdx + i;
        if (q < n) v |= (q < (uint32_t)k ? map1(ch[i]) : map2(ch[i])) << (2 * i);
      }
      dst[wdx] = v;
    }
  };
  pack(pr, nr, sr); pack(p1, n1, s1); pack(p2, n2, s2);
  const uint32_t npr = n_kmers(nr, k), np1 = n_kmers(n1, k), np2 = n_kmers(n2, k);
  for (uint32_t i = tid; i < 2 * ((npr + 63) / 64 + 1); i += kSplitThreads) bm[i] = 0u;
  // a lane's share of a sequence's positions: a contiguous run (lane_positions_of)
  auto share = [&](uint32_t np, uint32_t &base, uint32_t &cnt) {
    const uint32_t per = (np + (uint32_t)kSplitThreads - 1u) / (uint32_t)kSplitThreads;
    base = min((uint32_t)tid * per, np);
    cnt = min(per, np - base);
  };
  uint32_t br, cr, b1, c1, b2, c2;
  share(npr, br, cr); share(np1, b1, c1); share(np2, b2, c2);
  uint32_t lg = 0;
  while (max(max(npr, np1), np2) > (C::kPart << lg)) ++lg;
  using TAB = LTabP<C::kS16>;
  const TAB tr{wr, C::kCapRef - 1, sr, kmsk}, t1{w1, C::kCapOther - 1, s1, kmsk}, t2{w2, C::kCapOther - 1, s2, kmsk};
  __syncthreads();
  SP_STAMP(0);
  for (uint32_t part = 0; part < (1u << lg); ++part) {
    for (uint32_t i = tid; i < C::kRefWords + C::kOtherWords; i += kSplitThreads) wr[i] = 0xFFFFFFFFu;
    if (tid == 0) flag[0] = 0;
    __syncthreads();
    SP_STAMP(0);
    const MaskN<NW> mine_r = lane_positions_of<NW>(sr, br, cr, kmsk, part, lg);
    flat_add32<NW>(tr, br, mine_r);
    __syncthreads();
    SP_STAMP(1);
    flat_add32<NW>(t1, b1, flat_find32<NW>(tr, s1, b1, lane_positions_of<NW>(s1, b1, c1, kmsk, part, lg), kmsk), flag, (int)C::kFill);
    __syncthreads();
    SP_STAMP(2);
    if (flag[0] > (int)C::kFill) return false;
    for (uint32_t i = tid; i < C::kOtherWords; i += kSplitThreads) w2[i] = 0xFFFFFFFFu;   // nobody looks the reference's k-mers up any more
    __syncthreads();
    flat_add32<NW>(t2, b2, flat_find32<NW>(t1, s2, b2, lane_positions_of<NW>(s2, b2, c2, kmsk, part, lg), kmsk));   // no more distinct k-mers than the table before holds
    __syncthreads();
    SP_STAMP(3);
    // candidates: the reference positions whose k-mer is unique in all three reads, with their partner positions
    // (two flat look-ups that write the partner positions as they find them; a k-mer only enters the third table when
    // it is unique in the second, so the second look-up finds every k-mer the first one found)
    const MaskN<NW> in2 = flat_find32<NW>(t2, sr, br, mine_r, kmsk, [&](uint32_t q, uint32_t pos) { stg_global(g.cb + br + q, (int)pos); });
    flat_find32<NW>(t1, sr, br, in2, kmsk, [&](uint32_t q, uint32_t pos) {
      const uint32_t p = br + q;
      stg_global(g.ca + p, (int)pos);
      atomicOr(bm + (p >> 5), 1u << (p & 31u));
    });
    __syncthreads();
    SP_STAMP(4);
  }
  __threadfence();
  __syncthreads();
  return true;
}

// every k-mer of t's own sequence into t by SOME of the workgroup's threads (thread `me` of `stride`): flat_add for
// the wavefronts that prepare the next pass's reference table while wavefront 0 works on the chain
__device__ __forceinline__ void flat_add_by(const LTab &t, uint32_t np, uint32_t me, uint32_t stride)
{
  bool active = me < np;
  uint32_t p = me;
  uint32_t code = active ? LTab::bits(t.seq, p, t.kmsk) : 0u;
  uint32_t h = LTab::slot_of(code, t.mask), step = LTab::step_of(code), probes = 0;
  while (__builtin_amdgcn_ballot_w64(active) != 0) {
    if (active) {
      uint32_t *word = t.w + (h >> 1);
      const uint32_t sh = (h & 1u) * 16u;
      const uint32_t old = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t cur = (old >> sh) & 0xFFFFu;
      bool fin = false;
      if (cur == 0xFFFFu) fin = atomicCAS(word, old, (old & ~(0xFFFFu << sh)) | (p << sh)) == old;   // lost the race: look at the slot again
      else if (LTab::bits(t.seq, cur & 0x7FFFu, t.kmsk) == code) { if (!(cur & 0x8000u)) atomicOr(word, 0x8000u << sh); fin = true; }
      else { h = (h + step) & t.mask; fin = ++probes > t.mask; }
      if (fin) {
        p += stride;
        active = p < np;
        code = active ? LTab::bits(t.seq, p, t.kmsk) : 0u;
        h = LTab::slot_of(code, t.mask); step = LTab::step_of(code); probes = 0;
      }
    }
  }
}

// the three table phases and the candidate arrays of split_core on the LDS tables; false (uniform) when a table
// filled up: nothing has been written to ca / cb then and the HBM tables take the call
__device__ bool tables_lds(const WG &g, unsigned long long &sp_t_, int lds_off, const uint8_t *pr, uint32_t nr,
                           const uint8_t *p1, uint32_t n1, const uint8_t *p2, uint32_t n2, int k)
{
  const int tid = threadIdx.x;
  extern __shared__ int32_t dyn_lds_[];
  uint32_t *lds = reinterpret_cast<uint32_t *>(dyn_lds_) + lds_off;
  uint32_t *wr = lds, *w1 = wr + kLdsCapRef / 2, *w2 = wr;     // the third table takes the place of the first, which is dead by then
  uint32_t *sr = w1 + kLdsCapOther / 2, *s1 = sr + kLdsSeqWords, *s2 = s1 + kLdsSeqWords;
  const uint32_t kmsk = (1u << (2 * k)) - 1u;
  auto pack = [&](const uint8_t *s, uint32_t n, uint32_t *dst) {
    const uint32_t nw = (n + 15) / 16 + 2;
    for (uint32_t wdx = tid; wdx < nw; wdx += kSplitThreads) {
      uint8_t ch[16];
#pragma unroll
      for (uint32_t i = 0; i < 16; ++i) { const uint32_t q = 16 * wdx + i; ch[i] = s[q < n ? q : 0]; }   // all sixteen loads in flight
      uint32_t v = 0;
#pragma unroll
      for (uint32_t i = 0; i < 16; ++i) {
        const uint32_t q = 16 * wdx + i;
        if (q < n) v |= (q < (uint32_t)k ? map1(ch[i]) : map2(ch[i])) << (2 * i);
      }
      dst[wdx] = v;
    }
  };
  uint32_t *bm = s2 + kLdsSeqWords;
  // the fill counter and the "table full" flag live behind the bitmap, in the dynamic LDS like everything else here:
  // through a plain pointer to a __shared__ variable the counter's atomicAdd became a FLAT atomic (the compiler no
  // longer knows the address space), a far slower path than ds_add, once per k-mer that enters the second table
  int *flag = reinterpret_cast<int *>(bm + kLdsBitWords);
  // best_split runs the same three sequences through here with k = 15, 13, 11, 9: their packed form only differs in
  // the first word (which letters go through map1), so a pass that finds its sequences packed -- the key names them
  // -- redoes that word alone and skips two thirds of this phase's trips to memory.  (A re-split packs other ranges
  // and the key with them; k_split clears it at every read.)
  uint32_t *key = lds + kLdsKeyWord;
  const uint32_t want[9] = {(uint32_t)(uintptr_t)pr, (uint32_t)((uintptr_t)pr >> 32), nr, (uint32_t)(uintptr_t)p1, (uint32_t)((uintptr_t)p1 >> 32), n1,
                            (uint32_t)(uintptr_t)p2, (uint32_t)((uintptr_t)p2 >> 32), n2};
  bool packed = true;
#pragma unroll
  for (int i = 0; i < 9; ++i) packed = packed && key[i] == want[i];
  packed = __builtin_amdgcn_readfirstlane((int)packed) != 0;
  // ... and when the pass before ran on the same sequences, the wavefronts that had nothing to do during its chain
  // phase have already cleared the tables, fitted the first words to this k and entered the reference's k-mers
  // (split_core): flag[1] says for which k
  const bool prepared = packed && __builtin_amdgcn_readfirstlane(flag[1]) == k && __builtin_amdgcn_readfirstlane(flag[3]) >= 1;
  const bool prepared2 = prepared && __builtin_amdgcn_readfirstlane(flag[3]) >= 2;      // ... and the second table built from the uncorrected read's
  if (prepared) {
  } else if (packed) {
    if (tid < 3) {
      const uint8_t *s = tid == 0 ? pr : tid == 1 ? p1 : p2;
      const uint32_t n = tid == 0 ? nr : tid == 1 ? n1 : n2;
      uint32_t *dst = tid == 0 ? sr : tid == 1 ? s1 : s2;
      uint32_t v = 0;
      for (uint32_t i = 0; i < 16; ++i)
        if (i < n) v |= (i < (uint32_t)k ? map1(s[i]) : map2(s[i])) << (2 * i);
      dst[0] = v;
    }
  } else { pack(pr, nr, sr); pack(p1, n1, s1); pack(p2, n2, s2); }
  if (!prepared) {
    for (uint32_t i = tid; i < kLdsBitWords; i += kSplitThreads) bm[i] = 0u;
    for (uint32_t i = tid; i < (kLdsCapRef + kLdsCapOther) / 2; i += kSplitThreads) wr[i] = 0xFFFFFFFFu;
  }
  if (tid == 0) flag[0] = 0;
  __syncthreads();
  if (tid == 0) { flag[1] = 0; flag[3] = 0; }                // (whatever was prepared is used up, or was not for this pass)
  if (!packed && tid == 0) {                                 // (read by the next pass, barriers away)
#pragma unroll
    for (int i = 0; i < 9; ++i) key[i] = want[i];
  }
  SP_STAMP(0);
  const LTab tr{wr, kLdsCapRef - 1, sr, kmsk}, t1{w1, kLdsCapOther - 1, s1, kmsk}, t2{w2, kLdsCapOther - 1, s2, kmsk};
  const uint32_t npr = n_kmers(nr, k), np1 = n_kmers(n1, k), np2 = n_kmers(n2, k);
  const uint32_t all_r = lane_positions(npr), all_1 = lane_positions(np1), all_2 = lane_positions(np2);
  if (!prepared) flat_add(tr, all_r);
  __syncthreads();
  SP_STAMP(1);
  if (!prepared2) flat_add(t1, flat_find(tr, s1, all_1, kmsk, g.stamps ? g.stamps + 13 : nullptr), flag, (int)kLdsFill);
  __syncthreads();
  SP_STAMP(2);
  if (flag[0] > (int)kLdsFill) return false;
  for (uint32_t i = tid; i < kLdsCapOther / 2; i += kSplitThreads) w2[i] = 0xFFFFFFFFu;     // nobody looks the reference's k-mers up any more
  __syncthreads();
  flat_add(t2, flat_find(t1, s2, all_2, kmsk));                // no more distinct k-mers than the table before holds
  __syncthreads();
  SP_STAMP(3);
  // candidates: the reference positions whose k-mer is unique in all three reads, one bit each in the (cleared)
  // bitmap; the partner positions are looked up again for the few positions that become anchors (anchors_lds)
  for (uint32_t c = flat_find(t1, sr, flat_find(t2, sr, all_r, kmsk), kmsk); c; c &= c - 1u) {
    const uint32_t p = (uint32_t)tid + (uint32_t)__builtin_ctz(c) * kSplitThreads;
    atomicOr(bm + (p >> 5), 1u << (p & 31u));
  }
  return true;
}

// anchors of a split() level from the candidate bitmap of tables_lds (:234-251): position 0 without a distance test,
// then greedily every candidate more than minSize loop steps after the last one taken (loop index j = position - 1,
// last_indexed starts at 0).  Wavefront 0 holds the bitmap in registers and walks it with scalar code; all threads
// then look the anchors' partner positions up in the two on-chip tables.  Workgroup-wide.
// bm: the candidate bitmap (LDS, one bit per reference position, 64-bit words as dword pairs); atab: room for six
// dwords per bitmap word (LDS, dead memory); partner(position, a, b): the anchor's positions in the other two reads.
template <class LV, class Partner>
__device__ void anchors_bitmap(const LV &L, const uint32_t *bm, uint32_t *atab, uint32_t np, uint32_t minSize, Partner &&partner,
                               unsigned long long *stamps = nullptr)
{
  unsigned long long st_t_ = stamps ? __builtin_readcyclecounter() : 0;
#define AN_STAMP(idx) do { if (stamps && threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); atomicAdd(stamps + (idx), now_ - st_t_); st_t_ = now_; } } while (0)

  const int tid = threadIdx.x;
  const uint32_t nwords = (np + 63) >> 6;
  // The usual case, minSize = 20 (any minSize up to 29): the walk as a chain of small functions.  What the walk does
  // inside one 64-bit word of the bitmap depends on the word and on ONE number, the first position of the word it may
  // take (0 .. minSize + 2: what the anchor before left over); what it hands to the next word is again such a number
  // (0 .. minSize).  The threads tabulate that function for every word (32 entries of 5 bits); wavefront 0 then only
  // chains the tables, 64 words at a time, and its lanes write the anchors of their words, at offsets from a prefix
  // sum of their counts.
  const bool chained = minSize <= 29u;
  // exit tables: six 5-bit entries to a dword, six dwords to a word of the bitmap; a thread per dword
  const uint32_t ne = minSize + 3u, nd = (ne + 5u) / 6u;
  if (chained) {
    for (uint32_t idx = (uint32_t)tid; idx < nwords * nd; idx += kSplitThreads) {
      const uint32_t w = idx / nd, d = idx - w * nd;
      const unsigned long long B = ((unsigned long long)bm[2 * w + 1] << 32) | bm[2 * w];
      uint32_t v = 0;
      for (uint32_t j = 0; j < 6u; ++j) {
        const uint32_t e = 6u * d + j;
        if (e >= ne) break;
        unsigned long long x = B & (~0ull << e);
        int t = -1;
        while (x) {
          t = __builtin_ctzll(x);
          const uint32_t nx = (uint32_t)t + minSize + 1u;
          x = nx >= 64u ? 0ull : x & (~0ull << nx);
        }
        const uint32_t ex = t >= 0 ? (uint32_t)max(0, t + (int)minSize + 1 - 64) : 0u;
        v |= ex << (5u * j);
      }
      atab[6 * w + d] = v;
    }
    __syncthreads();
  }
  AN_STAMP(16);
  if (tid < 64) {
    int n = 0;
    bool fail = false;
    if (bm[0] & 1u) { if (tid == 0) L.ar[0] = 0; n = 1; }
    // the walk starts behind loop index 0 + minSize whether position 0 was taken or not (last_indexed = 0): the
    // first candidate that counts has j = p - 1 > minSize
    const unsigned long long from0 = (unsigned long long)minSize + 2ull;
    if (chained) {
      // Chaining the tables looks serial -- a word's entry state is the exit state of the word before -- but nearly
      // every word that holds a candidate forgets how it was entered.  So every lane guesses, applies its word's
      // table, takes its neighbour's exit state as the new entry state, and the wavefront repeats that until no
      // lane's entry state changes any more: correct for lanes 0 .. t after t rounds whatever the guesses were,
      // and in practice done after two or three.  Then every lane: the anchors of its word, counted, placed behind
      // the lanes (words) before it, written.
      uint32_t st = (uint32_t)from0;                       // <= 31: inside word 0
      for (uint32_t q0 = 0; q0 < nwords; q0 += 64u) {
        const uint32_t wd = q0 + (uint32_t)tid;
        const uint32_t wq = min(64u, nwords - q0);
        uint32_t T[6];
#pragma unroll
        for (uint32_t d = 0; d < 6u; ++d) T[d] = wd < nwords && d < nd ? atab[6 * wd + d] : 0u;
        auto exit_of = [&](uint32_t e) {
          const uint32_t sel = (e * 43u) >> 8;              // e / 6, e < 64
          const uint32_t tv = sel == 0u ? T[0] : sel == 1u ? T[1] : sel == 2u ? T[2] : sel == 3u ? T[3] : sel == 4u ? T[4] : T[5];
          return (tv >> (5u * (e - 6u * sel))) & 31u;
        };
        uint32_t in = tid == 0 ? st : 0u, out = exit_of(in);
        for (;;) {
          uint32_t prev = (uint32_t)__shfl_up((int)out, 1);
          if (tid == 0) prev = st;
          if (__builtin_amdgcn_ballot_w64((uint32_t)tid < wq && prev != in) == 0) break;
          in = prev;
          out = exit_of(in);
        }
        st = (uint32_t)__builtin_amdgcn_readlane((int)out, (int)(wq - 1u));
        const unsigned long long B = wd < nwords ? (((unsigned long long)bm[2 * wd + 1] << 32) | bm[2 * wd]) : 0ull;
        const unsigned long long x0 = B & (~0ull << in);
        int c = 0;
        for (unsigned long long x = x0; x;) {
          const uint32_t nx = (uint32_t)__builtin_ctzll(x) + minSize + 1u;
          ++c;
          x = nx >= 64u ? 0ull : x & (~0ull << nx);
        }
        int inc = c;
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (tid >= d) inc += t; }
        int at = n + inc - c;
        for (unsigned long long x = x0; x;) {
          const uint32_t t = (uint32_t)__builtin_ctzll(x), nx = t + minSize + 1u;
          if (at < L.cap) L.ar[at] = (typename LV::elem)(64u * wd + t);
          ++at;
          x = nx >= 64u ? 0ull : x & (~0ull << nx);
        }
        n += __shfl(inc, 63);
      }
      AN_STAMP(17);
      if (n > L.cap) fail = true;
    } else {
    auto word = [&](uint32_t wd) -> unsigned long long { return ((unsigned long long)bm[2 * wd + 1] << 32) | bm[2 * wd]; };   // uniform wd: one broadcast read
    uint32_t from = from0 >= (unsigned long long)np ? np : (uint32_t)from0;
    uint32_t wd = from >> 6;
    unsigned long long cur = wd < nwords ? word(wd) & (~0ull << (from & 63u)) : 0ull;
    while (wd < nwords) {
      if (cur == 0ull) {
        ++wd;
        if (wd >= nwords) break;
        cur = word(wd);
        const uint32_t base = wd << 6;
        if (from > base) cur = from - base >= 64u ? 0ull : cur & (~0ull << (from - base));
        continue;
      }
      const uint32_t p = (wd << 6) + (uint32_t)__builtin_ctzll(cur);
      if (p >= np) break;
      // taken: loop index j = p - 1; the next one must have j' > j + minSize, i.e. p' >= p + minSize + 1
      if (n < L.cap) { if (tid == 0) L.ar[n] = (typename LV::elem)p; }
      else fail = true;
      ++n;
      const unsigned long long next64 = (unsigned long long)p + minSize + 1ull;
      if (next64 >= (unsigned long long)np) break;
      from = (uint32_t)next64;
      if ((from >> 6) == wd) cur &= ~0ull << (from & 63u);
      else {
        wd = from >> 6;
        if (wd >= nwords) break;
        cur = word(wd) & (~0ull << (from & 63u));
      }
    }
    }
    if (tid == 0) { L.s->n = fail ? 0 : n; if (fail) L.s->fail = 1; }
  }
  __syncthreads();
  AN_STAMP(18);
  // partner positions of the anchors: the k-mer's only occurrence in the uncorrected / corrected read
  const int n = L.s->n;
  for (int i = tid; i < n; i += kSplitThreads) {
    int a, b;
    partner((uint32_t)L.ar[i], a, b);
    L.aa[i] = (typename LV::elem)a;
    L.ab[i] = (typename LV::elem)b;
  }
  __syncthreads();
}

// the anchors of a pass that went through tables_lds: bitmap, exit tables (over the dead reference table, behind the
// third table) and the two tables the partner positions are looked up in
template <class LV>
__device__ void anchors_lds(const LV &L, int lds_off, uint32_t np, uint32_t minSize, int k, unsigned long long *stamps = nullptr)
{
  extern __shared__ int32_t dyn_lds_[];
  uint32_t *lds = reinterpret_cast<uint32_t *>(dyn_lds_) + lds_off;
  uint32_t *w1 = lds + kLdsCapRef / 2, *w2 = lds;
  uint32_t *sr = w1 + kLdsCapOther / 2, *s1 = sr + kLdsSeqWords, *s2 = s1 + kLdsSeqWords;
  const uint32_t *bm = s2 + kLdsSeqWords;
  const uint32_t kmsk = (1u << (2 * k)) - 1u;
  const LTab t1{w1, kLdsCapOther - 1, s1, kmsk}, t2{w2, kLdsCapOther - 1, s2, kmsk};
  anchors_bitmap(L, bm, lds + kLdsCapOther / 2, np, minSize, [&](uint32_t p, int &a, int &b) {
    const uint32_t c = LTab::bits(sr, p, kmsk);
    a = t1.find(c);
    b = t2.find(c);
  }, stamps);
}

// maximum over the wavefront, uniform (DPP row shifts + four v_readlane: no LDS traffic)
__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
  // after the four shifts lane 15 of every row of 16 holds the row's maximum (a lane without a source takes 0)
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 31),
                 c = (uint32_t)__builtin_amdgcn_readlane((int)v, 47), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
  return max(max(a, b), max(c, d));
}

// tables, anchors and the best chain of one split() call (ref: split :175-255, best_chain :79-126).
// Workgroup-wide.  On return L.s->n anchors, L.s->nchain chain entries (indices of the chain in L.cl[0 .. nchain), reused).
// LONG: which kernel's copy of this function -- 0: no partitioned tables; 1: the kernel of the medium reads (up to
// 32,767 bases, two workgroups per CU); 2: the kernel of the long reads (one workgroup per CU there: twice the registers)
template <int LONG, class LV>
__device__ void split_core(const WG &g_in, const LV &L_in, DSeq ref, DSeq S1, DSeq S2, int k, uint32_t minSize, int next_k = 0)
{
  // This function is not inlined (three call sites, a long body): behind the references its arguments live in the
  // caller's scratch memory, and every use of a member after a store the compiler cannot tell apart from it is a
  // FLAT load from there -- once per anchor in the chain loop below, on its critical path.  Copies in registers.
  // The arguments of such a function also arrive in vector registers, and the compiler must take everything computed
  // from them for divergent: loop counters, addresses and branch conditions that are the same in all lanes went
  // through the vector ALU and exec-mask loops.  uniform() (v_readfirstlane) says what they are.
  const WG g = uniform(g_in);
  const LV L = uniform(L_in);
  ref = uniform(ref); S1 = uniform(S1); S2 = uniform(S2);
  k = uniform(k); minSize = uniform(minSize); next_k = uniform(next_k);
  const uint8_t *pr = g.reads + ref.base, *p1 = g.reads + S1.base, *p2 = g.reads + S2.base;
  const int tid = threadIdx.x;
  // table sizes: power of two >= 2 n + 2 (splitter.cpp), within the scratch
  auto cap_for = [&](uint32_t n) { int64_t c = 64; while (c < 2 * (int64_t)n + 2) c <<= 1; return c; };
  Tab tr = g.tab[0], t1 = g.tab[1], t2 = g.tab[2];
  const int64_t cr = cap_for(ref.n), c1 = cap_for(S1.n), c2 = cap_for(S2.n);
  tr.mask = (uint32_t)cr - 1; t1.mask = (uint32_t)c1 - 1; t2.mask = (uint32_t)c2 - 1;
  unsigned long long sp_t_ = g.stamps ? __builtin_readcyclecounter() : 0;
  if (tid == 0) { L.s->n = 0; L.s->nchain = 0; L.s->start = -1; }
  // (a sequence shorter than k has one k-mer whose code depends on its length: only the HBM tables' codes, built
  // like the reference's, tell such k-mers apart the way the reference does)
  bool on_chip = g.lds_tab >= 0 && ref.n <= kLdsMaxN && S1.n <= kLdsMaxN && S2.n <= kLdsMaxN &&
                 ref.n >= (uint32_t)k && S1.n >= (uint32_t)k && S2.n >= (uint32_t)k;
  if (g.stamps && tid == 0) { atomicAdd(g.stamps + 11, (unsigned long long)ref.n); atomicAdd(g.stamps + 12, (unsigned long long)S1.n); }
  if (on_chip) on_chip = tables_lds(g, sp_t_, g.lds_tab, pr, ref.n, p1, S1.n, p2, S2.n, k);
  bool long_chip = false, long_small = true;
  if constexpr (LONG == 2) {                                    // (only the kernel of the long batches carries that code)
    const uint32_t longest = max(max(ref.n, S1.n), S2.n);
    long_chip = !on_chip && g.lds_long >= 0 && longest <= (g.long_b ? LongB::kMaxN : LongA::kMaxN) && ref.n >= (uint32_t)k && S1.n >= (uint32_t)k && S2.n >= (uint32_t)k;
    long_small = longest <= LongA::kMaxN;
    if (long_chip) long_chip = long_small ? tables_long<LongA>(g, sp_t_, g.lds_long, pr, ref.n, p1, S1.n, p2, S2.n, k)
                                          : tables_long<LongB>(g, sp_t_, g.lds_long, pr, ref.n, p1, S1.n, p2, S2.n, k);
    if (long_chip && g.stamps && tid == 0) atomicAdd(g.stamps + 10, 1ull);
  } else if constexpr (LONG == 1) {
    const uint32_t longest = max(max(ref.n, S1.n), S2.n);
    long_chip = !on_chip && g.lds_long >= 0 && longest <= LongS::kMaxN && ref.n >= (uint32_t)k && S1.n >= (uint32_t)k && S2.n >= (uint32_t)k;
    if (long_chip) long_chip = tables_long<LongS>(g, sp_t_, g.lds_long, pr, ref.n, p1, S1.n, p2, S2.n, k);
    if (long_chip && g.stamps && tid == 0) atomicAdd(g.stamps + 10, 1ull);
  }
  if (on_chip) {
    if (g.stamps && tid == 0) atomicAdd(g.stamps + 10, 1ull);
    __syncthreads();
    SP_STAMP(4);
    anchors_lds(L, g.lds_tab, n_kmers(ref.n, k), minSize, k, g.stamps);
    SP_STAMP(5);
  } else if (long_chip) {
    // the anchors from the bitmap tables_long left in LDS (exit tables over its dead tables), their partner positions
    // from the dense arrays -- only the anchors' entries of those are ever read
    extern __shared__ int32_t dyn_lds_[];
    uint32_t *lds = reinterpret_cast<uint32_t *>(dyn_lds_) + g.lds_long;
    const uint32_t *bm = lds + (LONG == 1 ? LongS::kBmWord : long_small ? LongA::kBmWord : LongB::kBmWord);
    anchors_bitmap(L, bm, lds, n_kmers(ref.n, k), minSize, [&](uint32_t p, int &a, int &b) { a = ldg(g.ca + p); b = ldg(g.cb + p); }, g.stamps);
    SP_STAMP(5);
  } else {
  reset_tab(tr, cr); reset_tab(t1, c1); reset_tab(t2, c2);
  __threadfence();
  __syncthreads();
  SP_STAMP(0);
  // contiguous chunk of positions per thread, rolling code inside the chunk, kU positions handed out at a time
  auto for_kmers = [&](const uint8_t *s, uint32_t n, auto &&fn) {
    const uint32_t np = n_kmers(n, k);
    const uint32_t chunk = (np + kSplitThreads - 1) / kSplitThreads;
    const uint32_t p0 = (uint32_t)tid * chunk, pe = min(np, p0 + chunk);
    if (p0 >= pe) return;
    uint32_t code = code_at(s, n, p0, k);
    const uint32_t msk = (1u << (2 * k)) - 1u;
    for (uint32_t pb = p0; pb < pe; pb += kU) {
      uint32_t ps[kU], cs[kU];
      bool on[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const uint32_t p = pb + u;
        on[u] = p < pe;
        ps[u] = p; cs[u] = code;
        if (p + 1 < pe) {
          const uint32_t q = p + k;                           // the letter that enters
          code = ((code << 2) | (q < (uint32_t)k ? map1(s[q]) : map2(s[q]))) & msk;
        }
      }
      fn(ps, cs, on);
    }
  };
  for_kmers(pr, ref.n, [&](const uint32_t (&p)[kU], const uint32_t (&c)[kU], const bool (&on)[kU]) { tr.add(c, p, on); });
  __threadfence();
  __syncthreads();
  SP_STAMP(1);
  for_kmers(p1, S1.n, [&](const uint32_t (&p)[kU], const uint32_t (&c)[kU], const bool (&on)[kU]) {
    int u1[kU];
    tr.unique_pos(c, on, u1);
    bool on1[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) on1[u] = on[u] && u1[u] >= 0;
    t1.add(c, p, on1);
  });
  __threadfence();
  __syncthreads();
  SP_STAMP(2);
  for_kmers(p2, S2.n, [&](const uint32_t (&p)[kU], const uint32_t (&c)[kU], const bool (&on)[kU]) {
    int u1[kU];
    t1.unique_pos(c, on, u1);
    bool on1[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) on1[u] = on[u] && u1[u] >= 0;
    t2.add(c, p, on1);
  });
  __threadfence();
  __syncthreads();
  SP_STAMP(3);
  // candidates per reference position: partner positions when the k-mer is unique in all three reads
  for_kmers(pr, ref.n, [&](const uint32_t (&p)[kU], const uint32_t (&c)[kU], const bool (&on)[kU]) {
    int b[kU], a[kU];
    t2.unique_pos(c, on, b);
    bool onb[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) onb[u] = on[u] && b[u] >= 0;
    t1.unique_pos(c, onb, a);
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (on[u]) { stg(g.ca + p[u], a[u]); stg(g.cb + p[u], b[u]); }
  });
  __threadfence();
  __syncthreads();
  SP_STAMP(4);
  }
  // anchors (:234-251), wavefront 0: position 0 without a distance test, then greedily every candidate more than
  // minSize loop steps after the last one taken (loop index j = position - 1, last_indexed starts at 0)
  const uint32_t np = n_kmers(ref.n, k);
  if (!on_chip && !long_chip) {
  if (tid < 64) {
    int n = 0;
    if (tid == 0) {
      const int a0 = ldg(g.ca), b0 = ldg(g.cb);
      if (b0 >= 0) { L.ar[0] = 0; L.aa[0] = a0; L.ab[0] = b0; n = 1; }
    }
    n = __shfl(n, 0);
    uint32_t last = 0;
    bool fail = false;
    constexpr int kB = 4;                                   // blocks of 64 positions per trip to memory
    for (uint32_t base0 = 1; base0 < np; base0 += 64 * kB) {
      int av[kB], bv[kB];
#pragma unroll
      for (int q = 0; q < kB; ++q) {
        const uint32_t p = base0 + 64 * q + tid;
        av[q] = p < np ? ldg(g.ca + p) : -1;
        bv[q] = p < np ? ldg(g.cb + p) : -1;
      }
#pragma unroll
      for (int q = 0; q < kB; ++q) {
        const uint32_t base = base0 + 64 * q;
        const int a = av[q], b = a >= 0 ? bv[q] : -1;
        unsigned long long m = __builtin_amdgcn_ballot_w64(a >= 0);
        // candidates at most minSize loop steps after the last anchor are out: drop them from the mask at once
        if (m && last + minSize + 1 >= base) {
          const uint32_t skip = last + minSize + 2 - base;    // first lane whose j = base + l - 1 exceeds last + minSize
          m = skip >= 64 ? 0ull : m & (~0ull << skip);
        }
        while (m) {
          const int l = __builtin_ctzll(m);
          const uint32_t j = base + l - 1;                    // loop index of this position
          const int sa = __builtin_amdgcn_readlane(a, l), sb = __builtin_amdgcn_readlane(b, l);
          if (n < L.cap) { if (tid == 0) { L.ar[n] = (typename LV::elem)(j + 1); L.aa[n] = (typename LV::elem)sa; L.ab[n] = (typename LV::elem)sb; } }
          else fail = true;
          ++n;
          last = j;
          const uint32_t skip = (uint32_t)l + minSize + 1;     // the next one must sit more than minSize steps further on
          m = skip >= 64 ? 0ull : m & (~0ull << skip);
        }
      }
    }
    if (tid == 0) { L.s->n = fail ? 0 : n; if (fail) L.s->fail = 1; }
  }
  __syncthreads();
  SP_STAMP(5);
  }
  // longest chain, back to front (:79-126): wavefront 0
  const int n = uniform(L.s->n);
  if (g.stamps && tid == 0) atomicAdd(g.stamps + 14, (unsigned long long)n);
  // The chain is one wavefront's work, a third of the pass, and the tables are dead by now.  When the caller knows
  // that the same three sequences come again with a smaller k (best_split's next round), the other wavefronts use
  // the time: tables and bitmap cleared, the first words of the packed sequences fitted to that k (by all, before
  // the chain starts), then the reference's k-mers entered -- the next pass (tables_lds) finds that done.  Should a
  // re-split come in between, or best_split stop, the work was for nothing and nobody relies on it.
  // (Worth it when the chain is long enough to hide the work: some 430 cycles per anchor against 35 k cycles for the
  // reference's table and 60 k more for the second.)
  const int kPrepOne = 60, kPrepTwo = 140;
  if (on_chip && next_k > 0 && n >= kPrepOne) {
    extern __shared__ int32_t dyn_lds_[];
    uint32_t *lds = reinterpret_cast<uint32_t *>(dyn_lds_) + g.lds_tab;
    uint32_t *wr = lds, *sr = wr + (kLdsCapRef + kLdsCapOther) / 2, *s1 = sr + kLdsSeqWords, *s2 = s1 + kLdsSeqWords;
    uint32_t *bm = s2 + kLdsSeqWords;
    int *flag = reinterpret_cast<int *>(bm + kLdsBitWords);
    for (uint32_t i = tid; i < kLdsBitWords; i += kSplitThreads) bm[i] = 0u;
    for (uint32_t i = tid; i < (kLdsCapRef + kLdsCapOther) / 2; i += kSplitThreads) wr[i] = 0xFFFFFFFFu;
    if (tid == 0) { flag[2] = 0; flag[3] = 0; }
    if (tid < 3) {
      const uint8_t *sq = tid == 0 ? pr : tid == 1 ? p1 : p2;
      const uint32_t nn = tid == 0 ? ref.n : tid == 1 ? S1.n : S2.n;
      uint32_t v = 0;
      for (uint32_t i = 0; i < 16; ++i)
        if (i < nn) v |= (i < (uint32_t)next_k ? map1(sq[i]) : map2(sq[i])) << (2 * i);
      (tid == 0 ? sr : tid == 1 ? s1 : s2)[0] = v;
    }
    __syncthreads();
    if (tid >= 64) {
      const uint32_t kmsk_n = (1u << (2 * next_k)) - 1u, me = (uint32_t)tid - 64u, others = (uint32_t)kSplitThreads - 64u;
      const LTab trn{wr, kLdsCapRef - 1, sr, kmsk_n}, t1n{wr + kLdsCapRef / 2, kLdsCapOther - 1, s1, kmsk_n};
      flat_add_by(trn, n_kmers(ref.n, next_k), me, others);
      if (tid == 64) { flag[1] = next_k; flag[3] = 1; }
      if (n >= kPrepTwo) {
      // ... and, the chain being as long as it is, the second table too.  Its look-ups need the reference's table
      // complete: the seven wavefronts meet at a counter in LDS (they are all resident: they can wait for each other
      // without the eighth).
      __threadfence_block();
      if ((tid & 63) == 0) atomicAdd(&flag[2], 1);
      while (__hip_atomic_load(&flag[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (kSplitThreads - 64) / 64) __builtin_amdgcn_s_sleep(2);
      __threadfence_block();
      if (tid == 64) flag[0] = 0;                            // (nobody counts before the second meeting)
      const uint32_t mine = lane_positions(n_kmers(S1.n, next_k), me, others);
      const uint32_t hits = flat_find(trn, s1, mine, kmsk_n, nullptr, me, others);
      __threadfence_block();
      if ((tid & 63) == 0) atomicAdd(&flag[2], 1);
      while (__hip_atomic_load(&flag[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 2 * ((kSplitThreads - 64) / 64)) __builtin_amdgcn_s_sleep(2);
      __threadfence_block();
      flat_add(t1n, hits, flag, (int)kLdsFill, me, others);
      __threadfence_block();
      if ((tid & 63) == 0) atomicAdd(&flag[2], 1);
      while (__hip_atomic_load(&flag[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 3 * ((kSplitThreads - 64) / 64)) __builtin_amdgcn_s_sleep(2);
      __threadfence_block();
      // (a second table that filled up is left half built: the next pass then starts from scratch)
      if (tid == 64) flag[3] = __hip_atomic_load(&flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= (int)kLdsFill ? 2 : -1;
      }
    }
  }
  if (tid < 64 && n > 0) {
    // The successors an anchor can chain to lie within 1000 reference bases: at most 48 anchors, which the
    // wavefront keeps in registers -- lane l holds anchor i + 1 + l (positions and chain length), shifted by one
    // lane per step with the anchor just finished entering at lane 0.  No LDS read sits on the loop's critical
    // path; the anchors themselves are fetched 64 at a time.
    int wr = 0, wa = 0, wb = 0, wc = 0;
    int fr = 0, fa = 0, fb = 0;
    if (minSize >= 15u && n <= 0xFFFF) {
      // The usual case (anchors at least 16 bases apart: fewer than 63 within 1000 bases) the other way round: not
      // "anchor i looks at its successors" -- a maximum over the wavefront per anchor, on the critical path -- but
      // "anchor i, once its chain length is known, offers itself to its predecessors".  Lane l keeps the one
      // anchor with index = l (mod 64) that is still waiting (indices i - 63 .. i) and the best offer it has had;
      // anchors finish in descending order, so when i's turn comes every successor in reach has made its offer:
      // one readlane instead of a reduction, and each lane updates its own maximum.  Later offers come from
      // earlier anchors: the key (length, 0xFFFF - index) keeps the longest and, among equals, the earliest.
      using E = typename LV::elem;
      const int lane = tid;
      auto fetch = [&](int b, int &r, int &a, int &bb) {
        const int j = b * 64 + lane;
        const bool v = b >= 0 && j < n;
        r = v ? (int)L.ar[j] : 0x7fffffff; a = v ? (int)L.aa[j] : 0; bb = v ? (int)L.ab[j] : 0;   // (no offer passes the test of an empty lane)
      };
      const int c0 = (n - 1) & 63, b0 = (n - 1) >> 6;
      int nr, na, nb, pr, pa, pb;
      fetch(b0 - 1, nr, na, nb);                              // what enters a lane when its anchor has finished: the block below
      fetch(b0, pr, pa, pb);
      if (lane > c0) { pr = nr; pa = na; pb = nb; }
      uint32_t offer = 0, rkey = 0;
      uint32_t back = 0xFFFFu - (uint32_t)(n - 1);                     // 0xFFFF - i
      for (int i = n - 1; i >= 0; --i, ++back) {
        const int c = i & 63;
        if (c == 63 && i != n - 1) fetch((i >> 6) - 1, nr, na, nb);     // the lanes now hold exactly block i / 64
        const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)offer, c);   // (1 + the successor's length) << 16 | 0xFFFF - successor, 0 without one
        const int sr = __builtin_amdgcn_readlane(pr, c), sa = __builtin_amdgcn_readlane(pa, c), sb = __builtin_amdgcn_readlane(pb, c);
        const bool here = lane == c;
        rkey = here ? key : rkey;
        pr = here ? nr : pr; pa = here ? na : pa; pb = here ? nb : pb;    // anchor i - 64 takes the lane
        offer = here ? 0u : offer;
        const uint32_t mine = ((key & 0xFFFF0000u) + 0x10000u) | back;
        // (ref :98-110) a successor lies less than 1000 further on in all three reads, and further on at all
        const uint32_t far = max(max((uint32_t)(sr - pr - 1), (uint32_t)(sa - pa - 1)), (uint32_t)(sb - pb - 1));
        offer = max(offer, far < 999u ? mine : 0u);
        if (c == 0) {                                                    // a block is complete: its results go to the arrays
          const int j = i + lane;
          if (j < n) { L.cl[j] = (E)(rkey >> 16); L.cn[j] = (E)(rkey ? 0xFFFF - (int)(rkey & 0xFFFFu) : -1); }
        }
      }
    } else
    for (int i = n - 1; i >= 0; --i) {
      if (i == n - 1 || (i & 63) == 63) {
        const int jj = (i & ~63) + tid;
        fr = jj < n ? (int)L.ar[jj] : 0; fa = jj < n ? (int)L.aa[jj] : 0; fb = jj < n ? (int)L.ab[jj] : 0;
      }
      const int ri = __builtin_amdgcn_readlane(fr, i & 63), ai = __builtin_amdgcn_readlane(fa, i & 63),
                bi = __builtin_amdgcn_readlane(fb, i & 63);
      const bool near = i + 1 + tid < n && wr - ri < 1000 && wr > ri;
      int best = -1, nxt = -1;
      if (__builtin_amdgcn_ballot_w64(tid == 63 && near) == 0) {
        const bool ok = near && wa - ai < 1000 && wa > ai && wb - bi < 1000 && wb > bi;
        // longest first, the earlier successor among equals: one key, one reduction over the wavefront
        const uint32_t top = wave_max(ok ? (((uint32_t)wc + 1u) << 6) | (uint32_t)(63 - tid) : 0u);
        if (top) { best = (int)(top >> 6) - 1; nxt = i + 1 + 63 - (int)(top & 63u); }
      } else {
        // more than 64 successors in reach (cannot happen with anchors more than 20 bases apart): the plain scan
        for (int j0 = i + 1; j0 < n; j0 += 64) {
          const int j = j0 + tid;
          const bool nr = j < n && L.ar[j] - ri < 1000 && L.ar[j] > ri;
          const bool ok = nr && L.aa[j] - ai < 1000 && L.aa[j] > ai && L.ab[j] - bi < 1000 && L.ab[j] > bi;
          const uint32_t top = wave_max(ok ? (((uint32_t)L.cl[j] + 1u) << 6) | (uint32_t)(63 - tid) : 0u);
          if (top) {
            const int v = (int)(top >> 6) - 1, vj = j0 + 63 - (int)(top & 63u);
            if (v > best) { best = v; nxt = vj; }            // strict: an earlier block's successor wins ties
          }
          // ref: the scan stops at the first anchor too far on the reference ("TOO FAR NOW", :98-101)
          if (__builtin_amdgcn_ballot_w64(j < n && !nr) != 0) break;
        }
      }
      if (tid == 0) { L.cl[i] = (typename LV::elem)(1 + best); L.cn[i] = (typename LV::elem)(best >= 0 ? nxt : -1); }
      // the window moves on: lane l takes lane l - 1, lane 0 the anchor just finished (DPP wave_shr:1)
      wr = __builtin_amdgcn_update_dpp(ri, wr, 0x138, 0xf, 0xf, false);
      wa = __builtin_amdgcn_update_dpp(ai, wa, 0x138, 0xf, 0xf, false);
      wb = __builtin_amdgcn_update_dpp(bi, wb, 0x138, 0xf, 0xf, false);
      wc = __builtin_amdgcn_update_dpp(1 + best, wc, 0x138, 0xf, 0xf, false);
    }
    __builtin_amdgcn_wave_barrier();
    // start: the longest, the earliest among equals
    int v = -1, vi = 0x7fffffff;
    for (int i = tid; i < n; i += 64) if (L.cl[i] > v) { v = L.cl[i]; vi = i; }
    for (int d = 32; d; d >>= 1) {
      const int ov = __shfl_xor(v, d), oi = __shfl_xor(vi, d);
      if (ov > v || (ov == v && oi < vi)) { v = ov; vi = oi; }
    }
    // the chain as a list of anchor indices, written over cl (no longer needed).  The chain only moves forward, so
    // the walk goes block by block: a block's 64 successors in a register, the steps inside the block by readlane
    // (no LDS read per step), the visited anchors as a mask whose lanes then write their list entries together.
    {
      int i = uniform(vi), c = 0;
      while (i != LV::kNoNext) {
        const int blk = i >> 6;
        const int j = blk * 64 + tid;
        const int nx = j < n ? (int)L.cn[j] : (int)LV::kNoNext;
        unsigned long long seen = 0;
        do {
          seen |= 1ull << (i & 63);
          i = __builtin_amdgcn_readlane(nx, i & 63);
        } while (i != LV::kNoNext && (i >> 6) == blk);
        if ((seen >> tid) & 1ull) L.cl[c + __builtin_popcountll(seen & ((1ull << tid) - 1ull))] = (typename LV::elem)j;
        c += __builtin_popcountll(seen);
      }
      if (tid == 0) { L.s->start = vi; L.s->nchain = c; }
    }
  }
  __syncthreads();
  SP_STAMP(6);
  if (g.stamps && threadIdx.x == 0) atomicAdd(g.stamps + 7, 1ull);
}

__device__ __forceinline__ DSeq dsub(DSeq s, uint32_t pos, uint32_t len)     // std::string::substr semantics
{
  if (pos > s.n) pos = s.n;
  return DSeq{s.base + pos, min(len, s.n - pos)};
}
__device__ __forceinline__ DSeq dsub(DSeq s, uint32_t pos) { return dsub(s, pos, 0xFFFFFFFFu); }

// one split() of the reference (:175-308) including the two re-splits of a missing start / end.  Workgroup-wide;
// the window list `out` (in HBM) is built by wavefront 0; returns its length through sh[0] (LDS).
template <int LONG, class LV>
__device__ void split_read(const WG &g, const LV &L0, const LV &L1, DSeq ref, DSeq S1, DSeq S2, int k, int32_t *out, int32_t *tmp,
                           int *sh /* LDS ints: [0] n out, [1] overflow, [2..] scratch */, int next_k = 0 /* the k of best_split's next round, if any */)
{
  (void)tmp;
  const int tid = threadIdx.x;
  split_core<LONG>(g, L0, ref, S1, S2, k, 20u, next_k);
  unsigned long long sp_t_ = g.stamps ? __builtin_readcyclecounter() : 0;
  const int64_t cap = g.maxwin;
  // the list's length and overflow flag: the same in all lanes of wavefront 0, which is the only one that uses them
  int on = 0;
  bool over = false;
  auto wstore = [&](int slot, uint32_t ro, uint32_t rl, uint32_t ao, uint32_t al, uint32_t bo, uint32_t bl, int nfill) {
    if (slot >= cap) return;                      // (counted all the same: the caller sees on > cap)
    // (through a pointer known to be global memory: FLAT stores would also count as LDS operations, and the next read
    // of the anchor arrays would wait for their trip to memory)
    int32_t *w = out + 8 * (int64_t)slot;
    stg_global(w, (int)ro); stg_global(w + 1, (int)rl); stg_global(w + 2, (int)ao); stg_global(w + 3, (int)al);
    stg_global(w + 4, (int)bo); stg_global(w + 5, (int)bl); stg_global(w + 6, nfill); stg_global(w + 7, 0);
  };
  // ---- the walk along a chain (:279-293, and the same rule inside the re-splits): anchors i0 .. n_end - 1 of L's chain
  // list, a window cut at every anchor that lies more than thr behind the end of the window before in all three
  // sequences with sizes within a factor 1.5.  Wavefront 0.  Which anchor cuts the next window depends only on the
  // anchor that cut the last one: every lane works out, for its own anchor, which later anchor of the block would
  // follow it (a few trials each, all lanes at once), the wavefront then hops along these links from the first
  // anchor the carried-in window end admits -- one readlane per window instead of a test per anchor -- and the
  // accepted lanes write their windows together: emit(slot, ends of the window before, the lane's anchor).
  // |size_S - size_R| < size_R * 0.5 is 2 |d| < size_R, exactly.
  auto walk = [&](const LV &L, int i0, int n_end, uint32_t thr, uint32_t &pr_, uint32_t &p1_, uint32_t &p2_, auto &&emit) {
    auto cuts = [&](uint32_t er, uint32_t e1, uint32_t e2, int r, int a, int b) {      // window ends before, anchor
      const int size_R = (int)((uint32_t)r - er), size_S1 = (int)((uint32_t)a - e1),
                size_S2 = (int)((uint32_t)b - e2);                    // ref: ints from unsigned arithmetic, compared as unsigned
      return (uint32_t)size_R > thr && (uint32_t)size_S1 > thr && (uint32_t)size_S2 > thr &&
             2ll * llabs((long long)size_S1 - size_R) < (long long)size_R && 2ll * llabs((long long)size_S2 - size_R) < (long long)size_R;
    };
    for (int base = i0; base < n_end; base += 64) {
      const int idx = base + tid;
      const bool in = idx < n_end;
      const int anl = in ? (int)L.cl[idx] : 0;
      const int vr = in ? (int)L.ar[anl] : 0, va = in ? (int)L.aa[anl] : 0, vb = in ? (int)L.ab[anl] : 0;
      const int cnt = min(64, n_end - base);
      const uint32_t pr0 = pr_, p10 = p1_, p20 = p2_;
      int nxt = 64;                                                    // none in this block
      for (int d = 1; __builtin_amdgcn_ballot_w64(nxt == 64 && tid + d < cnt) != 0; ++d) {
        const int src = min(tid + d, 63);
        const int cr = __shfl(vr, src), ca = __shfl(va, src), cb = __shfl(vb, src);
        if (nxt == 64 && tid + d < cnt && cuts((uint32_t)(vr + k), (uint32_t)(va + k), (uint32_t)(vb + k), cr, ca, cb)) nxt = tid + d;
      }
      unsigned long long acc = 0;
      const unsigned long long first = __builtin_amdgcn_ballot_w64(tid < cnt && cuts(pr_, p1_, p2_, vr, va, vb));
      if (first) {
        int l = __builtin_ctzll(first), last = l;
        while (l < 64) { acc |= 1ull << l; last = l; l = __builtin_amdgcn_readlane(nxt, l); }
        pr_ = (uint32_t)(__builtin_amdgcn_readlane(vr, last) + k); p1_ = (uint32_t)(__builtin_amdgcn_readlane(va, last) + k);
        p2_ = (uint32_t)(__builtin_amdgcn_readlane(vb, last) + k);
      }
      const unsigned long long below = acc & ((1ull << tid) - 1ull);
      const int prev = below ? 63 - __builtin_clzll(below) : 0;
      const int qr = __shfl(vr, prev), qa = __shfl(va, prev), qb = __shfl(vb, prev);
      if ((acc >> tid) & 1ull) {
        const uint32_t mr = below ? (uint32_t)(qr + k) : pr0, m1 = below ? (uint32_t)(qa + k) : p10, m2 = below ? (uint32_t)(qb + k) : p20;
        emit(on + __builtin_popcountll(below), mr, m1, m2, vr, va, vb);
      }
      on += __builtin_popcountll(acc);
    }
  };
  // a re-split's windows (:264-277, :294-306): reference against uncorrected by the plain rule with minSize ms; the
  // corrected side is the 'N' filler except in ONE window (`special`: the re-split's last window at the start of a
  // read, its first at the end), which takes what there is of the corrected read (co, cl_) -- when there is any
  auto resplit_windows = [&](uint32_t rn, uint32_t un, uint32_t roff, uint32_t uoff, uint32_t ms, bool special_last, uint32_t co, uint32_t cl_) {
    const DSeq rr{0, rn}, r1{0, un};
    const int nb1 = uniform(L1.s->nchain);
    const int on0 = on;
    uint32_t a_ = 0, b_ = 0, c_ = 0;
    auto put = [&](int slot, bool special, DSeq wr, DSeq w1) {
      if (special && cl_ > 0) wstore(slot, (uint32_t)wr.base + roff, wr.n, (uint32_t)w1.base + uoff, w1.n, co, cl_, 0);
      else wstore(slot, (uint32_t)wr.base + roff, wr.n, (uint32_t)w1.base + uoff, w1.n, 0, 1, 1);
    };
    walk(L1, 0, nb1 - 1, ms, a_, b_, c_, [&](int slot, uint32_t mr, uint32_t m1, uint32_t, int vr, int va, int) {
      put(slot, !special_last && slot == on0, dsub(rr, mr, (uint32_t)(vr - (int)mr + k)), dsub(r1, m1, (uint32_t)(va - (int)m1 + k)));
    });
    if (tid == 0) put(on, special_last || on == on0, dsub(rr, a_), dsub(r1, b_));       // what is left behind the last cut
    ++on;
    if (uniform(L1.s->fail)) over = true;
  };
  uint32_t pred_ref = 0, pred_S1 = 0, pred_S2 = 0;
  int i = 0;
  const int nbl = L0.s->nchain;
  if (tid == 0) { sh[0] = 0; sh[1] = 0; }
  __syncthreads();
  if (nbl == 0) {
    if (tid == 0) { wstore(0, 0, ref.n, 0, S1.n, 0, S2.n, 0); sh[0] = cap >= 1 ? 1 : 0; sh[1] = cap < 1 || L0.s->fail; }
    __syncthreads();
    return;
  }
  // ---- missing start (:264-277) ----
  const int a0 = L0.cl[0];
  const uint32_t sr = min(ref.n, (uint32_t)(L0.ar[a0] + k)), s1 = min(S1.n, (uint32_t)(L0.aa[a0] + k)), s2 = min(S2.n, (uint32_t)(L0.ab[a0] + k));
  const bool rec_start = (uint64_t)s2 * 2 < sr && sr - s2 > 200;
  if (rec_start) {
    SP_STAMP(15);
    split_core<LONG>(g, L1, DSeq{ref.base, sr}, DSeq{S1.base, s1}, DSeq{ref.base, sr}, k, (uint32_t)(1.2 * s2));
    if (g.stamps) sp_t_ = __builtin_readcyclecounter();
    if (tid < 64) resplit_windows(sr, s1, 0u, 0u, (uint32_t)(1.2 * s2), true, 0u, s2);
    pred_S1 = (uint32_t)(L0.aa[a0] + k); pred_ref = (uint32_t)(L0.ar[a0] + k); pred_S2 = (uint32_t)(L0.ab[a0] + k);
    i = 1;
  }
  SP_STAMP(19);
  if (tid < 64) {
    uint32_t pr_ = uniform(pred_ref), p1_ = uniform(pred_S1), p2_ = uniform(pred_S2);
    walk(L0, i, nbl - 1, 20u, pr_, p1_, p2_, [&](int slot, uint32_t mr, uint32_t m1, uint32_t m2, int vr, int va, int vb) {
      const DSeq wr = dsub(DSeq{0, ref.n}, mr, (uint32_t)(vr - (int)mr + k)),
                 w1 = dsub(DSeq{0, S1.n}, m1, (uint32_t)(va - (int)m1 + k)),
                 w2 = dsub(DSeq{0, S2.n}, m2, (uint32_t)(vb - (int)m2 + k));
      wstore(slot, (uint32_t)wr.base, wr.n, (uint32_t)w1.base, w1.n, (uint32_t)w2.base, w2.n, 0);
    });
    pred_ref = pr_; pred_S1 = p1_; pred_S2 = p2_;
    if (tid == 0) { sh[4] = (int)pred_ref; sh[5] = (int)pred_S1; sh[6] = (int)pred_S2; }
  }
  __syncthreads();
  SP_STAMP(20);
  pred_ref = (uint32_t)sh[4]; pred_S1 = (uint32_t)sh[5]; pred_S2 = (uint32_t)sh[6];
  // ---- the end (:294-306) ----
  const DSeq er = dsub(DSeq{0, ref.n}, pred_ref), e1 = dsub(DSeq{0, S1.n}, pred_S1), e2 = dsub(DSeq{0, S2.n}, pred_S2);
  const bool rec_end = (uint64_t)e2.n * 2 < er.n && er.n - e2.n > 200;
  __syncthreads();              // L1 is about to be reused
  if (rec_end) {
    const DSeq gr{ref.base + er.base, er.n}, g1{S1.base + e1.base, e1.n};
    SP_STAMP(15);
    split_core<LONG>(g, L1, gr, g1, gr, k, (uint32_t)(1.2 * e2.n));
    if (g.stamps) sp_t_ = __builtin_readcyclecounter();
    if (tid < 64) resplit_windows(er.n, e1.n, (uint32_t)er.base, (uint32_t)e1.base, (uint32_t)(1.2 * e2.n), false, (uint32_t)e2.base, e2.n);
  } else if (tid < 64) {
    if (tid == 0) wstore(on, (uint32_t)er.base, er.n, (uint32_t)e1.base, e1.n, (uint32_t)e2.base, e2.n, 0);
    ++on;
  }
  if (tid == 0) {
    if (on > cap) { over = true; on = (int)cap; }
    sh[0] = on; sh[1] = (over || L0.s->fail) ? 1 : 0;
  }
  __syncthreads();
  SP_STAMP(15);
}

// ref: largest_fragment (:158-169) measures every LINE of the "header\nseq\n" text, header lines included.
// Workgroup-wide: the window list sits in HBM (thread 0 wrote it), and one thread reading its few hundred lengths
// one after the other was a quarter of a read's time -- every thread takes a share, the maximum goes through LDS.
__device__ uint32_t largest_fragment(const int32_t *wl, int n, uint32_t hdr_len, int *slot)
{
  const int tid = threadIdx.x;
  if (tid == 0) *slot = 0;
  __syncthreads();
  uint32_t r = 0;
  for (int w = tid; w < n; w += kSplitThreads) r = max(r, (uint32_t)ldg(wl + 8 * (int64_t)w + 1) + 1);
  r = wave_max(r);
  if ((tid & 63) == 0 && r) atomicMax(slot, (int)r);
  __syncthreads();
  uint32_t res = (uint32_t)*slot;
  if (n >= 1) res = max(res, n >= 2 ? hdr_len + 1 : hdr_len);
  return res;
}

template <bool BIG, int LONG = 0>
__global__ void __launch_bounds__(kSplitThreads, (LONG == 2 ? kSplitThreads / 256 : kSplitThreads >= 1024 ? 8 : kSplitThreads >= 768 ? 6 : kSplitThreads >= 512 ? 4 : 2)) k_split(SplitArgs a)
{
  extern __shared__ int32_t s_anc[];              // !BIG: 2 levels x 5 arrays x a.maxanc entries (sized by the batch's longest read)
  __shared__ LvlState s_lvl[2];
  __shared__ int sh[8];
  const int tid = threadIdx.x;
  using LV = typename std::conditional<BIG, Lvl32, Lvl16>::type;
  using AT = typename LV::elem;
  LV L0, L1;
  {
    const int cap = (int)a.maxanc;
    if constexpr (BIG) {
      int32_t *base = a.anc + (int64_t)blockIdx.x * 2 * 5 * a.maxanc;
      L0.ar.p = base; L0.aa.p = base + cap; L0.ab.p = base + 2 * cap; L0.cl.p = base + 3 * cap; L0.cn.p = base + 4 * cap;
      base += 5 * (int64_t)cap;
      L1.ar.p = base; L1.aa.p = base + cap; L1.ab.p = base + 2 * cap; L1.cl.p = base + 3 * cap; L1.cn.p = base + 4 * cap;
    } else {
      const uint32_t c = (uint32_t)cap;
      L0.ar.off = 0; L0.aa.off = c; L0.ab.off = 2 * c; L0.cl.off = 3 * c; L0.cn.off = 4 * c;
      L1.ar.off = 5 * c; L1.aa.off = 6 * c; L1.ab.off = 7 * c; L1.cl.off = 8 * c; L1.cn.off = 9 * c;
    }
    L0.cap = L1.cap = cap;
    L0.s = &s_lvl[0]; L1.s = &s_lvl[1];
  }
  WG g;
  g.reads = a.reads;
  for (int t = 0; t < 3; ++t) {
    g.tab[t].ent = a.ent + ((int64_t)blockIdx.x * 3 + t) * a.tab_cap;
    g.tab[t].mask = 0;
  }
  g.ca = a.ca + (int64_t)blockIdx.x * (a.maxlen + 2);
  g.cb = a.cb + (int64_t)blockIdx.x * (a.maxlen + 2);
  g.wl = a.wl + (int64_t)blockIdx.x * 3 * a.maxwin * 8;
  g.tab_cap = a.tab_cap; g.maxwin = a.maxwin;
  g.stamps = a.stamps;
  g.lds_tab = (!BIG && a.lds_tables) ? (int)((2 * 5 * a.maxanc * sizeof(AT) + 3) / 4) : -1;
  g.lds_long = (BIG && a.lds_long) ? 0 : -1;                       // (the anchor arrays of such a batch are in HBM)
  g.long_b = a.lds_long >= 2;
  // A read's cost grows faster than its length and the lengths of a batch differ by an order of magnitude: the
  // workgroups draw the reads from a counter, longest first (a fixed stride left the chip a third idle on long reads).
  for (;;) {
    __syncthreads();
    if (tid == 0) sh[7] = atomicAdd(a.next, 1);
    __syncthreads();
    const int64_t turn = sh[7];
    if (turn >= a.n_reads) break;
    const int64_t r = a.order[turn];
    const DSeq ref{a.read_off[3 * r], (uint32_t)(a.read_off[3 * r + 1] - a.read_off[3 * r])};
    const DSeq S1{a.read_off[3 * r + 1], (uint32_t)(a.read_off[3 * r + 2] - a.read_off[3 * r + 1])};
    const DSeq S2{a.read_off[3 * r + 2], (uint32_t)(a.read_off[3 * r + 3] - a.read_off[3 * r + 2])};
    int kind;
    int nout = 0;
    unsigned long long sp_t_ = g.stamps ? __builtin_readcyclecounter() : 0;
    int32_t *dst = a.out_win + 8 * a.out_first[r];
    const int64_t dcap = a.out_first[r + 1] - a.out_first[r];
    __syncthreads();
    if (tid == 0) {
      L0.s->fail = 0; L1.s->fail = 0;
      if (g.lds_tab >= 0) { s_anc[g.lds_tab + (int)kLdsKeyWord + 2] = 0; s_anc[g.lds_tab + (int)kLdsKeyWord - 3] = 0; }   // no sequence is packed yet (tables_lds: a length of 0 matches none), nothing prepared
    }
    __syncthreads();
    if (ref.n <= 2) kind = -1;                                                   // :414
    else if ((double)S2.n / ref.n >= a.thr) {                                   // :415
      // best_split (:310-332): k = 15, then smaller k while the largest fragment shrinks
      int32_t *best = g.wl, *aux = g.wl + a.maxwin * 8, *tmp = g.wl + 2 * a.maxwin * 8;
      int kk = 15;
      split_read<LONG>(g, L0, L1, ref, S1, S2, kk, best, tmp, sh, kk - 2);
      int nbest = sh[0];
      bool over = sh[1] != 0;
      uint32_t largest = largest_fragment(best, nbest, (uint32_t)a.hdr_len[r], &sh[2]);
      for (;;) {
        kk -= 2;
        if (kk < 9 || over) break;
        __syncthreads();
        split_read<LONG>(g, L0, L1, ref, S1, S2, kk, aux, tmp, sh, kk - 2 >= 9 ? kk - 2 : 0);
        const int naux = sh[0];
        over = over || sh[1] != 0;
        const uint32_t la = largest_fragment(aux, naux, (uint32_t)a.hdr_len[r], &sh[3]);
        if (la < largest) { largest = la; int32_t *t = best; best = aux; aux = t; nbest = naux; }
        else break;
      }
      __syncthreads();
      if (over || nbest > dcap) kind = -2;                                        // the host code takes this read
      else if (nbest <= 1) kind = 2;                                             // :417-423
      else {
        kind = 0;
        nout = nbest;
        for (int64_t i = tid; i < 8 * (int64_t)nbest; i += kSplitThreads) dst[i] = ldg(best + i);
      }
    } else kind = 1;                                                             // :425-431
    if (tid == 0) { a.out_kind[r] = kind; a.out_cnt[r] = kind == 0 ? nout : (kind == 1 || kind == 2) ? 1 : 0; }
    SP_STAMP(8);                                   // whole read (the split_core phases included)
    if (g.stamps && tid == 0) atomicAdd(g.stamps + 9, 1ull);
  }
}

// ---- layout: windows -> the engine's input (bases of reference, corrected, uncorrected per window) ----
struct LayoutArgs {
  int64_t n_reads;
  const uint8_t *reads; const int64_t *read_off;
  const int32_t *out_win; const int64_t *out_first; const int32_t *out_cnt, *out_kind;
  const int64_t *win_first;      // per read: index of its first emitted window (exclusive scan of out_cnt)
  int64_t *wlen;                 // per emitted window: 3 lengths (reference, corrected, uncorrected)
  const int64_t *woff;           // exclusive scan of wlen: byte offsets
  uint8_t *bases;
};

__global__ void __launch_bounds__(256) k_split_lens(LayoutArgs a)
{
  for (int64_t r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
    const int kind = a.out_kind[r];
    const int64_t w0 = a.win_first[r];
    if (kind == 1 || kind == 2) { if (threadIdx.x < 3) a.wlen[3 * w0 + threadIdx.x] = 3; continue; }
    if (kind != 0) continue;
    const int32_t *w = a.out_win + 8 * a.out_first[r];
    for (int i = threadIdx.x; i < a.out_cnt[r]; i += 256) {
      a.wlen[3 * (w0 + i)] = w[8 * i + 1];            // reference
      a.wlen[3 * (w0 + i) + 1] = w[8 * i + 5];        // corrected (S2)
      a.wlen[3 * (w0 + i) + 2] = w[8 * i + 3];        // uncorrected (S1)
    }
  }
}

__global__ void __launch_bounds__(256) k_split_copy(LayoutArgs a)
{
  for (int64_t r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
    const int kind = a.out_kind[r];
    const int64_t w0 = a.win_first[r];
    if (kind == 1 || kind == 2) { if (threadIdx.x < 9) a.bases[a.woff[3 * w0] + threadIdx.x] = 'A'; continue; }
    if (kind != 0) continue;
    const int32_t *w = a.out_win + 8 * a.out_first[r];
    const uint8_t *pr = a.reads + a.read_off[3 * r], *p1 = a.reads + a.read_off[3 * r + 1], *p2 = a.reads + a.read_off[3 * r + 2];
    const int nw = a.out_cnt[r];
    // a wavefront per window, in turn
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = wave; i < nw; i += 4) {
      const int ro = w[8 * i], rl = w[8 * i + 1], ao = w[8 * i + 2], al = w[8 * i + 3], bo = w[8 * i + 4], bl = w[8 * i + 5], nf = w[8 * i + 6];
      uint8_t *d = a.bases + a.woff[3 * (w0 + i)];
      for (int j = lane; j < rl; j += 64) d[j] = pr[ro + j];
      d += rl;
      if (nf) { if (lane == 0) d[0] = 'N'; }
      else for (int j = lane; j < bl; j += 64) d[j] = p2[bo + j];
      d += bl;
      for (int j = lane; j < al; j += 64) d[j] = p1[ao + j];
    }
  }
}

}  // namespace elector

using namespace elector;

// ------------------------------------------------------------------ host ---

extern "C" int elector_split_reads_device(elector_ctx *c, int64_t n_in, const uint8_t *reads, const int64_t *read_off,
                                           const int32_t *hdr_len, double size_threshold, int nthreads,
                                           elector_windows_dev *out)
{
  if (!c || !out) return ELECTOR_E_INVAL;
  std::memset(out, 0, sizeof *out);
  if (n_in < 0 || (n_in > 0 && (!reads || !read_off || !hdr_len))) return elector_fail(c, ELECTOR_E_INVAL, "bad arguments");
  std::lock_guard<std::mutex> lock(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  int64_t maxlen = 1;
  for (int64_t r = 0; r < 3 * n_in; ++r) {
    const int64_t l = read_off[r + 1] - read_off[r];
    if (l < 0 || l > 0x7fffffff) return elector_fail(c, ELECTOR_E_INVAL, "read offsets");
    maxlen = std::max(maxlen, l);
  }
  const int64_t total = n_in ? read_off[3 * n_in] : 0;
  // per read: room for its windows (an anchor at most every 21 bases, plus the fillers of a re-split)
  std::vector<int64_t> first((size_t)n_in + 1, 0);
  int64_t maxwin = 16;
  for (int64_t r = 0; r < n_in; ++r) {
    const int64_t cap = (read_off[3 * r + 1] - read_off[3 * r]) / 16 + 16;
    first[(size_t)r + 1] = first[(size_t)r] + cap;
    maxwin = std::max(maxwin, cap);
  }
  // The batch goes to the device in up to three launches.  Reads within the small on-chip tables (12.5 kb) take the
  // kernel with the anchor arrays in LDS, two workgroups per CU; reads of up to 32,767 bases the kernel of the medium
  // partitioned tables (16-bit slots, 78 KB of LDS, two workgroups per CU, anchor arrays in HBM); longer ones the kernel of the large
  // partitioned tables (tables_long: one workgroup per CU, 153 KB of LDS), which also holds the HBM-table path for
  // what is longer still.  A PacBio batch whose lengths straddle 12.5 kb thus keeps all its reads on chip.
  // (ELECTOR_SPLIT_NO_LONG / ELECTOR_SPLIT_HBM_TABLES: one launch as before, HBM tables for the long reads;
  // ELECTOR_SPLIT_NO_MEDIUM: the medium reads with the long ones.)
  struct Launch {
    std::vector<int32_t> reads;                 // longest first: the order the workgroups take them in
    int64_t maxlen = 1, maxwin = 16, tab_cap = 64, maxanc = 8, per_block = 0;
    int blocks = 0;
    int lng = 0;                                // 0: small tables / HBM tables; 1: medium reads' kernel; 2: long reads' kernel
    bool big = false;
    size_t at_keys = 0, at_ca = 0, at_wl = 0, at_anc = 0;     // its share of the workspace (the launches run side by side)
  };
  Launch part[3];
  const bool no_long = std::getenv("ELECTOR_SPLIT_HBM_TABLES") || std::getenv("ELECTOR_SPLIT_NO_LONG");
  const bool no_medium = std::getenv("ELECTOR_SPLIT_NO_MEDIUM") != nullptr;       // (A/B: the medium reads with the long ones)
  for (int64_t r = 0; r < n_in; ++r) {
    const int64_t m = std::max(std::max(read_off[3 * r + 1] - read_off[3 * r], read_off[3 * r + 2] - read_off[3 * r + 1]),
                               read_off[3 * r + 3] - read_off[3 * r + 2]);
    Launch &L = part[no_long || m <= (int64_t)kLdsMaxN ? 0 : m <= (int64_t)LongS::kMaxN && !no_medium ? 1 : 2];
    L.reads.push_back((int32_t)r);
    L.maxlen = std::max(L.maxlen, m);
    L.maxwin = std::max(L.maxwin, (read_off[3 * r + 1] - read_off[3 * r]) / 16 + 16);
  }
  const int max_blocks = std::getenv("ELECTOR_SPLIT_BLOCKS") ? std::max(1, std::atoi(std::getenv("ELECTOR_SPLIT_BLOCKS"))) : 1024;
  int dev_cus = 256;
  (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, c->device);
  // per-block scratch (three hash tables of 64-bit slots, two candidate arrays) grows with the launch's LONGEST read: a
  // single 300 kb read would ask for 25 GB at 1024 blocks.  The blocks are capped by a byte budget (a block loops over
  // reads anyway), and a workspace that still cannot be had sends the batch to the host splitter (ELECTOR_E_LIMIT).
  const int64_t budget = (int64_t)8 << 30;
  size_t need_keys = 0, need_ca = 0, need_wl = 0, need_anc = 0;
  for (int k = 0; k < 3; ++k) {
    Launch &L = part[k];
    if (L.reads.empty()) continue;
    std::sort(L.reads.begin(), L.reads.end(), [&](int32_t x, int32_t y) {
      const int64_t lx = read_off[3 * (int64_t)x + 3] - read_off[3 * (int64_t)x], ly = read_off[3 * (int64_t)y + 3] - read_off[3 * (int64_t)y];
      return lx != ly ? lx > ly : x < y;
    });
    while (L.tab_cap < 2 * L.maxlen + 2) L.tab_cap <<= 1;
    // anchors are more than 20 bases apart on the reference (the re-split of a missing end can use less: it
    // overflows into the host path, like a window list that does not fit)
    L.maxanc = L.maxlen / 21 + 8;
    L.lng = k;
    L.big = L.lng != 0 || L.maxanc > kMaxAnchors;
    L.per_block = 3 * L.tab_cap * 8 + 2 * (L.maxlen + 2) * 4 + 3 * L.maxwin * 8 * 4;
    L.blocks = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((int64_t)L.reads.size(), L.lng == 2 ? std::max(1, dev_cus) : L.lng == 1 ? 2 * std::max(1, dev_cus) : max_blocks),
                                                             std::max<int64_t>(64, budget / L.per_block)));
    L.at_keys = need_keys; L.at_ca = need_ca; L.at_wl = need_wl; L.at_anc = need_anc;
    need_keys += (size_t)L.blocks * 3 * (size_t)L.tab_cap * 8;
    need_ca += ((size_t)L.blocks * (size_t)(L.maxlen + 2) * 4 + 63) & ~(size_t)63;
    need_wl += (size_t)L.blocks * 3 * (size_t)L.maxwin * 8 * 4;
    if (L.big) need_anc += ((size_t)L.blocks * 2 * 5 * (size_t)L.maxanc * 4 + 127) & ~(size_t)63;
  }
  hipStream_t st = c->stream;
  int rc = c->d_sp_reads.ensure((size_t)total + 64) | c->d_sp_off.ensure((size_t)(3 * n_in + 1) * 8 + 64) |
           c->d_sp_hdr.ensure((size_t)n_in * 4 + 64) | c->d_sp_keys.ensure(need_keys + 64) |
           c->d_sp_ca.ensure(need_ca + 64) | c->d_sp_cb.ensure(need_ca + 64) | c->d_sp_wl.ensure(need_wl + 64) |
           c->d_sp_win.ensure((size_t)first[(size_t)n_in] * 8 * 4 + 64) | c->d_sp_first.ensure((size_t)(n_in + 1) * 8) |
           c->d_sp_cnt.ensure((size_t)(n_in + 1) * 4 * 2 + 64) | c->d_sp_wfirst.ensure((size_t)(n_in + 2) * 8 * 2) |
           (need_anc ? c->d_sp_anc.ensure(need_anc) : 0);
  if (rc) return elector_fail(c, ELECTOR_E_LIMIT, "device splitter workspace does not fit device memory: the host splitter takes this batch");
  if (n_in == 0) return ELECTOR_OK;
  auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
  const bool dbg = std::getenv("ELECTOR_DEBUG_SPLIT") != nullptr;
  const double t0 = now_ms();
  // the reads to the device through pinned staging, `nthreads` host threads filling one half while the other half's
  // copy runs (a pageable source is otherwise staged by the runtime on one thread, several times slower)
  hipPointerAttribute_t src_attr;
  const bool src_pinned = hipPointerGetAttributes(&src_attr, reads) == hipSuccess && src_attr.type == hipMemoryTypeHost;
  if (!src_pinned) (void)hipGetLastError();
  if (src_pinned) {
    // the reader's buffers are registered with the runtime (io_host.cpp): one DMA, no staging
    HIPCHK(c, hipMemcpyAsync(c->d_sp_reads.p, reads, (size_t)total, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));
  } else {
    constexpr size_t kChunk = (size_t)32 << 20;
    if (c->h_rows.ensure(2 * kChunk)) return elector_fail(c, ELECTOR_E_NOMEM, "pinned staging");
    uint8_t *stage = c->h_rows.as<uint8_t>();
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int k = 0; k < 2; ++k) HIPCHK(c, hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    int half = 0;
    bool used[2] = {false, false};
    int hip_rc = 0;
    for (size_t at = 0; at < (size_t)total && !hip_rc; at += kChunk, half ^= 1) {
      const size_t len = std::min(kChunk, (size_t)total - at);
      if (used[half]) hip_rc |= hipEventSynchronize(ev[half]) != hipSuccess;
      uint8_t *dst = stage + (size_t)half * kChunk;
      const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, nthreads), std::min<size_t>(8, len >> 20)));
      if (nt <= 1) std::memcpy(dst, reads + at, len);
      else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) {
          const size_t a0 = len * t / nt, a1 = len * (t + 1) / nt;
          th.emplace_back([=]() { std::memcpy(dst + a0, reads + at + a0, a1 - a0); });
        }
        for (auto &x : th) x.join();
      }
      hip_rc |= hipMemcpyAsync(c->d_sp_reads.as<uint8_t>() + at, dst, len, hipMemcpyHostToDevice, st) != hipSuccess;
      hip_rc |= hipEventRecord(ev[half], st) != hipSuccess;
      used[half] = true;
    }
    for (int k = 0; k < 2; ++k) { if (used[k]) (void)hipEventSynchronize(ev[k]); (void)hipEventDestroy(ev[k]); }
    if (hip_rc) return elector_fail(c, ELECTOR_E_HIP, "reads to the device");
  }
  const double t1 = now_ms();
  HIPCHK(c, hipMemcpyAsync(c->d_sp_off.p, read_off, (size_t)(3 * n_in + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->d_sp_hdr.p, hdr_len, (size_t)n_in * 4, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->d_sp_first.p, first.data(), (size_t)(n_in + 1) * 8, hipMemcpyHostToDevice, st));
  SplitArgs a;
  a.reads = c->d_sp_reads.as<uint8_t>(); a.read_off = c->d_sp_off.as<int64_t>(); a.hdr_len = c->d_sp_hdr.as<int32_t>();
  a.thr = size_threshold;
  a.ent = c->d_sp_keys.as<unsigned long long>();
  a.ca = c->d_sp_ca.as<int32_t>(); a.cb = c->d_sp_cb.as<int32_t>(); a.wl = c->d_sp_wl.as<int32_t>();
  a.out_win = c->d_sp_win.as<int32_t>(); a.out_first = c->d_sp_first.as<int64_t>();
  a.out_cnt = c->d_sp_cnt.as<int32_t>(); a.out_kind = a.out_cnt + (n_in + 1);
  a.stamps = nullptr;
  if (dbg) {
    if (c->d_sp_scan.ensure(4096)) return elector_fail(c, ELECTOR_E_NOMEM, "stamps");
    a.stamps = c->d_sp_scan.as<unsigned long long>();
    HIPCHK(c, hipMemsetAsync(a.stamps, 0, 256, st));
  }
  {
    // the counters the workgroups draw their reads from, and the launches' reads behind them
    std::vector<int32_t> order((size_t)n_in + 8, 0);
    std::copy(part[0].reads.begin(), part[0].reads.end(), order.begin() + 8);
    std::copy(part[1].reads.begin(), part[1].reads.end(), order.begin() + 8 + (ptrdiff_t)part[0].reads.size());
    std::copy(part[2].reads.begin(), part[2].reads.end(), order.begin() + 8 + (ptrdiff_t)(part[0].reads.size() + part[1].reads.size()));
    HIPCHK(c, hipMemcpyAsync(c->d_sp_wfirst.p, order.data(), order.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));                               // (the vector goes out of scope)
  }
  auto launch = [&](const Launch &L, int k, int64_t order_at, hipStream_t st) -> int {      // (st: the stream of this launch)
    if (L.reads.empty()) return 0;
    a.n_reads = (int64_t)L.reads.size();
    a.ent = reinterpret_cast<unsigned long long *>(c->d_sp_keys.as<uint8_t>() + L.at_keys);
    a.ca = reinterpret_cast<int32_t *>(c->d_sp_ca.as<uint8_t>() + L.at_ca); a.cb = reinterpret_cast<int32_t *>(c->d_sp_cb.as<uint8_t>() + L.at_ca);
    a.wl = reinterpret_cast<int32_t *>(c->d_sp_wl.as<uint8_t>() + L.at_wl);
    a.next = c->d_sp_wfirst.as<int32_t>() + 2 * k;
    a.order = c->d_sp_wfirst.as<int32_t>() + 8 + order_at;
    a.tab_cap = L.tab_cap; a.maxlen = L.maxlen; a.maxwin = L.maxwin;
    a.anc = nullptr; a.maxanc = L.maxanc;
    a.lds_tables = 0;
    a.lds_long = L.lng == 2 ? (std::getenv("ELECTOR_SPLIT_LONG") ? std::atoi(std::getenv("ELECTOR_SPLIT_LONG")) : 2) : L.lng;
    if (L.big) {
      a.anc = reinterpret_cast<int32_t *>(c->d_sp_anc.as<uint8_t>() + L.at_anc);
      static DeviceOnce once_long, once_medium;
      if (L.lng == 2 && once_long.need()) {
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_split<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024 - 512));
        once_long.done();
      }
      if (L.lng == 1 && once_medium.need()) {
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_split<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024 - 512));
        once_medium.done();
      }
      if (L.lng == 2) hipLaunchKernelGGL((k_split<true, 2>), dim3((unsigned)L.blocks), dim3(kSplitThreads), kLongBytes, st, a);
      else if (L.lng == 1) hipLaunchKernelGGL((k_split<true, 1>), dim3((unsigned)L.blocks), dim3(kSplitThreads), kLongSBytes, st, a);
      else hipLaunchKernelGGL((k_split<true, 0>), dim3((unsigned)L.blocks), dim3(kSplitThreads), (size_t)16, st, a);
    } else {
      // the anchor arrays in LDS, as many entries as the launch's longest read can need
      static DeviceOnce once;
      if (once.need()) {
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_split<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024 - 512));
        once.done();
      }
      // on-chip tables when most of the launch's reads are short enough for them and they fit beside the anchors
      size_t lds = ((size_t)(2 * 5 * L.maxanc * 2) + 3) & ~(size_t)3;   // 16-bit anchor arrays
      int64_t fit = 0;
      for (const int32_t r : L.reads) {
        const int64_t m = std::max(std::max(read_off[3 * (int64_t)r + 1] - read_off[3 * (int64_t)r], read_off[3 * (int64_t)r + 2] - read_off[3 * (int64_t)r + 1]),
                                   read_off[3 * (int64_t)r + 3] - read_off[3 * (int64_t)r + 2]);
        fit += m <= (int64_t)kLdsMaxN;
      }
      a.lds_tables = !std::getenv("ELECTOR_SPLIT_HBM_TABLES") && 2 * fit >= (int64_t)L.reads.size() && lds + kLdsTabBytes <= (size_t)(160 * 1024 - 512);
      if (a.lds_tables) lds += kLdsTabBytes;
      hipLaunchKernelGGL(k_split<false>, dim3((unsigned)L.blocks), dim3(kSplitThreads), lds, st, a);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
  };
  // the launches side by side on streams of their own (a launch's last workgroups leave most of the chip idle: the
  // next launch's workgroups move in), the long reads first
  if (!c->aux_ready) {
    for (int k = 0; k < elector_ctx::kAux; ++k) {
      if (c->make_stream(&c->aux[k])) return elector_fail(c, ELECTOR_E_HIP, "stream");
      HIPCHK(c, hipEventCreateWithFlags(&c->aux_done[k], hipEventDisableTiming));
    }
    HIPCHK(c, hipEventCreateWithFlags(&c->fork, hipEventDisableTiming));
    c->aux_ready = true;
  }
  {
    const int64_t at[3] = {0, (int64_t)part[0].reads.size(), (int64_t)(part[0].reads.size() + part[1].reads.size())};
    for (int k = 2; k >= 0; --k) {
      if (part[k].reads.empty()) continue;
      if ((rc = launch(part[k], k, at[k], c->aux[k])) != 0) return rc;
      HIPCHK(c, hipEventRecord(c->aux_done[k], c->aux[k]));
      HIPCHK(c, hipStreamWaitEvent(st, c->aux_done[k], 0));
    }
  }
  // kinds and counts to the host: the reads the device could not take are split by the host code
  std::vector<int32_t> cnt((size_t)n_in), kind((size_t)n_in);
  auto fetch_kinds = [&]() -> int {
    HIPCHK(c, hipMemcpyAsync(cnt.data(), a.out_cnt, (size_t)n_in * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(kind.data(), a.out_kind, (size_t)n_in * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return 0;
  };
  if ((rc = fetch_kinds()) != 0) return rc;
  {
    // A read whose anchors did not fit the arrays sized for its launch (a re-split with a tiny minSize can take an
    // anchor every other base) gets a second try with room for an anchor per base, in HBM -- before the whole batch
    // is handed to the host splitter for its sake.
    Launch again;
    for (int64_t r = 0; r < n_in; ++r)
      if (kind[(size_t)r] == -2) {
        again.reads.push_back((int32_t)r);
        for (int q = 0; q < 3; ++q) again.maxlen = std::max(again.maxlen, read_off[3 * r + q + 1] - read_off[3 * r + q]);
        again.maxwin = std::max(again.maxwin, (read_off[3 * r + 1] - read_off[3 * r]) / 16 + 16);
      }
    if (!again.reads.empty() && !no_long && (int64_t)again.reads.size() <= 4096) {
      while (again.tab_cap < 2 * again.maxlen + 2) again.tab_cap <<= 1;
      again.maxanc = again.maxlen + 8;
      again.lng = 2; again.big = true;
      again.blocks = (int)std::min<int64_t>((int64_t)again.reads.size(), std::max(1, dev_cus));
      if (c->d_sp_keys.ensure((size_t)again.blocks * 3 * (size_t)again.tab_cap * 8 + 64) | c->d_sp_ca.ensure((size_t)again.blocks * (size_t)(again.maxlen + 2) * 4 + 64) |
          c->d_sp_cb.ensure((size_t)again.blocks * (size_t)(again.maxlen + 2) * 4 + 64) | c->d_sp_wl.ensure((size_t)again.blocks * 3 * (size_t)again.maxwin * 8 * 4 + 64) |
          c->d_sp_anc.ensure((size_t)again.blocks * 2 * 5 * (size_t)again.maxanc * 4 + 64))
        return elector_fail(c, ELECTOR_E_LIMIT, "device splitter workspace does not fit device memory: the host splitter takes this batch");
      a.ent = c->d_sp_keys.as<unsigned long long>();
      a.ca = c->d_sp_ca.as<int32_t>(); a.cb = c->d_sp_cb.as<int32_t>(); a.wl = c->d_sp_wl.as<int32_t>();
      std::vector<int32_t> order(again.reads.size() + 8, 0);
      std::copy(again.reads.begin(), again.reads.end(), order.begin() + 8);
      HIPCHK(c, hipMemcpyAsync(c->d_sp_wfirst.p, order.data(), order.size() * 4, hipMemcpyHostToDevice, st));
      HIPCHK(c, hipStreamSynchronize(st));
      if ((rc = launch(again, 0, 0, st)) != 0) return rc;
      if ((rc = fetch_kinds()) != 0) return rc;
    }
  }
  const double t2 = now_ms();
  if (dbg) {
    unsigned long long hs[24];
    (void)hipMemcpy(hs, a.stamps, sizeof hs, hipMemcpyDeviceToHost);
    const double nc = hs[7] ? (double)hs[7] : 1.0, nr = hs[9] ? (double)hs[9] : 1.0;
    std::fprintf(stderr, "[elector] k_split: %llu reads, %.2f split() passes per read; cycles per pass: reset %.0f, table ref %.0f, table unc %.0f, table cor %.0f, candidates %.0f, anchors %.0f, chain %.0f; whole read %.0f; passes on the LDS tables %llu of %llu; per pass: ref %.0f bases, unc %.0f, look-up turns of wavefront 0 in the unc phase %.1f, anchors %.1f; window lists (cycles per read) %.0f; inside anchors: exit tables %.0f, chaining %.0f, writing %.0f; window lists: before the walk %.0f, walk %.0f\n",
                 hs[9], nc / nr, hs[0] / nc, hs[1] / nc, hs[2] / nc, hs[3] / nc, hs[4] / nc, hs[5] / nc, hs[6] / nc, hs[8] / nr, hs[10], hs[7], hs[11] / nc, hs[12] / nc, hs[13] / nc, hs[14] / nc, hs[15] / nr, hs[16] / nc, hs[17] / nc, hs[18] / nc, hs[19] / nr, hs[20] / nr);
  }
  (void)nthreads;
  int64_t n_host = 0;
  for (int64_t r = 0; r < n_in; ++r) n_host += kind[(size_t)r] == -2;
  if (n_host)   // rare: a read beyond the on-chip limits (kMaxAnchors anchors, window list); the caller uses the host entry
    return elector_fail(c, ELECTOR_E_LIMIT, "device splitter: a read exceeds the on-chip limits");
  // emitted reads, window counts, first window per read
  std::vector<int64_t> wfirst((size_t)n_in + 1, 0);
  int64_t nreads = 0, nwin = 0, small = 0, wrong = 0;
  for (int64_t r = 0; r < n_in; ++r) {
    wfirst[(size_t)r] = nwin;
    if (kind[(size_t)r] < 0) continue;
    ++nreads;
    nwin += cnt[(size_t)r];
    small += kind[(size_t)r] == 1;
    wrong += kind[(size_t)r] == 2;
  }
  wfirst[(size_t)n_in] = nwin;
  out->read_first = (int64_t *)std::malloc((size_t)(nreads + 1) * 8);
  out->read_index = (int64_t *)std::malloc((size_t)(nreads + 1) * 8);
  out->off = (int64_t *)std::malloc((size_t)(3 * nwin + 1) * 8);
  if (!out->read_first || !out->read_index || !out->off) { elector_windows_dev_free(out); return elector_fail(c, ELECTOR_E_NOMEM, "host arrays"); }
  {
    int64_t ri = 0;
    for (int64_t r = 0; r < n_in; ++r)
      if (kind[(size_t)r] >= 0) { out->read_first[ri] = wfirst[(size_t)r]; out->read_index[ri] = r; ++ri; }
    out->read_first[ri] = nwin;
  }
  rc = c->d_sp_wlen.ensure((size_t)(3 * nwin + 2) * 8) | c->d_sp_woff.ensure((size_t)(3 * nwin + 2) * 8);
  if (rc) { elector_windows_dev_free(out); return elector_fail(c, ELECTOR_E_NOMEM, "device splitter layout"); }
  int64_t *d_wfirst = c->d_sp_wfirst.as<int64_t>();
  HIPCHK(c, hipMemcpyAsync(d_wfirst, wfirst.data(), (size_t)(n_in + 1) * 8, hipMemcpyHostToDevice, st));
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(n_in, 1024));     // of the layout kernels (a block per read, in turn)
  LayoutArgs la;
  la.n_reads = n_in; la.reads = a.reads; la.read_off = a.read_off; la.out_win = a.out_win; la.out_first = a.out_first;
  la.out_cnt = a.out_cnt; la.out_kind = a.out_kind; la.win_first = d_wfirst;
  la.wlen = c->d_sp_wlen.as<int64_t>(); la.woff = c->d_sp_woff.as<int64_t>(); la.bases = nullptr;
  HIPCHK(c, hipMemsetAsync(la.wlen, 0, (size_t)(3 * nwin + 1) * 8, st));
  hipLaunchKernelGGL(k_split_lens, dim3((unsigned)blocks), dim3(256), 0, st, la);
  size_t tmp_bytes = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, la.wlen, c->d_sp_woff.as<int64_t>(), (int)(3 * nwin + 1), st);
  if (c->d_sp_scan.ensure(tmp_bytes + 64)) { elector_windows_dev_free(out); return elector_fail(c, ELECTOR_E_NOMEM, "scan scratch"); }
  if (hipcub::DeviceScan::ExclusiveSum(c->d_sp_scan.p, tmp_bytes, la.wlen, c->d_sp_woff.as<int64_t>(), (int)(3 * nwin + 1), st) != hipSuccess) {
    elector_windows_dev_free(out);
    return elector_fail(c, ELECTOR_E_HIP, "scan");
  }
  HIPCHK(c, hipMemcpyAsync(out->off, c->d_sp_woff.p, (size_t)(3 * nwin + 1) * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  const int64_t nb = out->off[3 * nwin];
  if (c->d_sp_bases.ensure((size_t)nb + 64)) { elector_windows_dev_free(out); return elector_fail(c, ELECTOR_E_NOMEM, "window bases"); }
  la.bases = c->d_sp_bases.as<uint8_t>();
  hipLaunchKernelGGL(k_split_copy, dim3((unsigned)blocks), dim3(256), 0, st, la);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(st));
  out->n_reads = nreads; out->n_windows = nwin; out->d_bases = la.bases; out->small_reads = small; out->wrong_reads = wrong;
  out->d_off = c->d_sp_woff.as<int64_t>();
  if (dbg) std::fprintf(stderr, "[elector] device splitter, host view: reads to the device %.1f ms, k_split %.1f ms, layout + window bases %.1f ms\n", t1 - t0, t2 - t1, now_ms() - t2);
  return ELECTOR_OK;
}

extern "C" void elector_windows_dev_free(elector_windows_dev *w)
{
  if (!w) return;
  std::free(w->off); std::free(w->read_first); std::free(w->read_index);
  std::memset(w, 0, sizeof *w);
}

// window bases out of the splitter's workspace into a caller's device buffer (or a host buffer), after the
// work queued on the context
extern "C" int elector_ctx_copy(elector_ctx *c, const void *src, void *dst, int64_t bytes)
{
  if (!c || (bytes > 0 && (!src || !dst)) || bytes < 0) return ELECTOR_E_INVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDefault, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ELECTOR_OK;
}

// test / debugging helper: device memory of this context's device to a host buffer
extern "C" int elector_ctx_copy_to_host(elector_ctx *c, const void *d_src, void *h_dst, int64_t bytes)
{
  if (!c || (bytes > 0 && (!d_src || !h_dst)) || bytes < 0) return ELECTOR_E_INVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
  return ELECTOR_OK;
}
