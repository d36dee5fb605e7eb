// elector_amd/csrc/bundle.hip -- a12: heaviest-bundle consensus of every window
// (optional output; the reference compiles it but never calls it, main.c:345-347).
//
// Reference behaviour restated: heaviest_bundle (heaviest_bundle.c:16-78),
// assign_sequence_bundle_id (:80-110), generate_lpo_bundles (:144-172) on the
// graph that fuse_lpo leaves after both fusions (lpo.c:413-463,602-656).
//
// The POA kernels never materialise the graph of (ref + cor + unc): they emit the
// MSA columns directly.  k_bundle rebuilds exactly the part of it that the bundle
// search looks at, from what the batch left in HBM (the graph after fusion #1:
// xinfo, ring1; the x -> y map of alignment #2: map16; the uncorrected symbols):
//
//   node record (16 B, node space of the window, in final node order)
//     pos[3]   position of the node's letter in ref / cor / unc (0xFFFF = absent)
//     nxt[3]   node that holds the next letter of ref / cor / unc  (0xFFFF = last)
//     col      MSA column (lpo_format.c:346-371: a new column at every ring change)
//     letter, cons   symbol index; bit k set = node lies on consensus path k
//
// The right-link list of a node, in the order the reference's add_lpo_link calls
// leave it (lpo.c:227-241: x's links first, then y's if new), is nxt[0], nxt[1],
// nxt[2] with duplicates dropped, because the graph is built ref <- cor <- unc.
//
// One lane = one window: the search is a right-to-left dynamic program over a
// chain-like graph with order-dependent tie-breaks (first best link wins, highest
// node index wins among equal path scores); the windows are what is parallel.
// Integer only except the one float comparison of :96.
//
// Round 4, second form (k_bundle_lds): everything a window's search touches lives in LDS.  The first form kept node
// records, scores and best links in HBM (22 bytes per node, each read three to five times per pass: 66 GB through L2
// per E. coli batch, 9.5 ms).  A window of up to 208 nodes needs 8-bit node ids only, and the overlap rule of :35-48
// needs no positions: a right neighbour counts source q when it is the node the q-link points to, or -- for a source
// this node lacks -- when it holds q's first letter.  So a node is one dword (three links, has / first bits), its
// score and best link another, column and letter a halfword: 11 bytes per node and window with the consensus row,
// 64 windows per workgroup, two to four workgroups per CU by the size of their largest graph.  The inputs are
// staged in (packed to a dword per node of the graph after fusion #1 by k_bundle_inputs, which also counts the nodes
// fusion #2 will make: the workgroup's LDS class), the consensus rows are put together in LDS and written out a row
// per instruction.  Windows beyond 208 nodes keep the first form (k_bundle_hbm).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstring>
#include <vector>

#include "ctx.h"
#include "elector_poa.h"
#include "poa_device.h"

namespace elector {

constexpr int kMaxBundles = 3;     // three source sequences: each productive pass bundles >= 1 of them

struct BundleArgs {
  int64_t n;
  const int64_t *off;
  const uint8_t *sym;
  const int2 *xinfo;
  const uint16_t *ring1;
  const uint32_t *map16;      // x_to_y of alignment #2 per node of the graph after fusion #1 (kNone32 = unaligned)
  const int32_t *n1;
  const int32_t *ncol;        // from the POA kernels (cross-check)
  int32_t *status;
  const DevTables *tab;
  float min_fraction;
  int debug;                  // measurement: 1 = stop behind the graph's records, 2 = behind the first pass's dynamic program
  // Scratch of the search, private to a window: node records, path scores, best right links.  Round 4: laid out per
  // BLOCK of 64 windows with the lane as the fastest index -- record k of the block's lane l at (base + k * 64 + l) --
  // so that the lanes of a wavefront, which walk their windows' nodes in step, touch one or two kilobyte rows per
  // instruction instead of 64 lines in 64 different windows' regions.  The blocks take the windows in the order of the
  // class lists (similar sizes side by side); a block's room is 64 x its largest window's node bound.
  const uint32_t *order_a;    // window of (block, lane): the fused classes' lists ...
  const uint32_t *order_b;    // ... then the generic list
  int64_t n_a;
  const int64_t *blk_base;    // per block: first record slot
  uint4 *node;                // 16-byte records
  int32_t *score;
  uint16_t *path;             // best right link
  // the search's inputs in the same per-block layout (k_bundle_inputs): letter | flags of node k of the graph after
  // fusion #1, its ring id and x -> y entry, the uncorrected symbols
  uint32_t *in_xy, *in_map;
  uint16_t *in_ring;
  uint8_t *in_ys;
  uint8_t *cons;              // out: window w, bundle k: ncol bytes at cons + 3*off[3w] + k*ncol
  int32_t *info;              // out: 8 per window: nbundle, count[3], bundle id of ref/cor/unc, ncol
  // second form: a dword per node of the graph after fusion #1 (letter | flags << 8 | x -> y entry << 16 | starts a
  // ring << 24; blocks whose windows all have n1, Lu <= 254), the blocks' classes as lists (k_bundle_inputs fills them)
  uint32_t *in_px;
  int32_t *cls_count;         // [kBundleClasses]
  int32_t *cls_list;          // [kBundleClasses][nblocks]
  int64_t nblocks;
  const int32_t *blocks;      // the launch's blocks (a class list)
};

// LDS classes of the second form: nodes per window a workgroup of 64 windows has room for (four, three, two, one
// workgroup per CU); then the first form on the packed inputs, and on the wide ones (a window with n1 or Lu > 254)
constexpr int kBundleClasses = 6;
constexpr int kBundleCap[4] = {52, 69, 104, 208};
constexpr int kBundleHbmPacked = 4, kBundleHbmWide = 5;
// share of a class's blocks whose records live in HBM instead of LDS.  Alone on the chip 50 % is the fastest split (E. coli batch
// 4.8 ms; 4.9 with 70, 5.3 with 100, 5.7 with 30); beside the alignment kernels of a four-context pipeline -- whose window slots
// want the LDS too -- the HBM form is the better neighbour: the pipelined step with the search 9.9 / 9.45 / 9.5 ms with 50 / 70 /
// 100 (yeast -split: 15.9 / 15.1 / 14.9; profiles/r05_bundles_share.txt).  Per class: the workgroups of the two large classes
// hold 37-73 KB each (two or one per CU) and are the worst neighbours, the small ones' 19-25 KB are not -- 30 / 50 / 100 / 100
// against 70 throughout: 8.8 against 8.9-9.0 ms (E. coli), 14.2 against 14.4 (yeast -split), and no slower alone (4.7 / 7.2).
constexpr int kBundleGlobalPct[4] = {30, 50, 100, 100};
constexpr uint32_t kPxNone = 0xFFu;
constexpr int kLdsSt = 65;                 // dwords between a lane's consecutive entries: a row of 64 lanes + 1 (staging
                                           // writes a window's entries from 64 lanes: stride 65 spreads them over the banks)
__host__ __device__ constexpr int bundle_row_stride(int C) { return 4 * (((C + 3) / 4) | 1); }   // bytes, an odd number of dwords
__host__ __device__ constexpr size_t bundle_lds_bytes(int C)
{
  return (size_t)C * kLdsSt * 4 * 2 + (size_t)C * 66 * 2 + (size_t)64 * bundle_row_stride(C);
}

struct NodeRec {
  uint16_t pos[3], nxt[3], col;
  uint8_t letter, cons;
};
static_assert(sizeof(NodeRec) == 16, "node record is one 16-byte load");

__device__ __forceinline__ NodeRec load_node(const uint4 *p)
{
  const uint4 v = *p;
  NodeRec r;
  r.pos[0] = (uint16_t)(v.x & 0xFFFF); r.pos[1] = (uint16_t)(v.x >> 16); r.pos[2] = (uint16_t)(v.y & 0xFFFF);
  r.nxt[0] = (uint16_t)(v.y >> 16); r.nxt[1] = (uint16_t)(v.z & 0xFFFF); r.nxt[2] = (uint16_t)(v.z >> 16);
  r.col = (uint16_t)(v.w & 0xFFFF); r.letter = (uint8_t)((v.w >> 16) & 0xFF); r.cons = (uint8_t)(v.w >> 24);
  return r;
}

__device__ __forceinline__ void store_node(uint4 *p, const NodeRec &r)
{
  uint4 v;
  v.x = (uint32_t)r.pos[0] | ((uint32_t)r.pos[1] << 16);
  v.y = (uint32_t)r.pos[2] | ((uint32_t)r.nxt[0] << 16);
  v.z = (uint32_t)r.nxt[1] | ((uint32_t)r.nxt[2] << 16);
  v.w = (uint32_t)r.col | ((uint32_t)r.letter << 16) | ((uint32_t)r.cons << 24);
  *p = v;
}

__device__ __forceinline__ int64_t bundle_window(const BundleArgs &a, int64_t idx)
{
  if (idx >= a.n) return -1;
  return idx < a.n_a ? (int64_t)a.order_a[idx] : (int64_t)a.order_b[idx - a.n_a];
}

// the block a workgroup of the search works on: the launch's list, or its own index
__device__ __forceinline__ int64_t bundle_block(const BundleArgs &a) { return a.blocks ? (int64_t)a.blocks[blockIdx.x] : (int64_t)blockIdx.x; }

// room of every block: 64 x the largest node bound (|PO| after fusion #1 + Lu + 1) among its windows; blk[b] holds it, a scan turns
// the array into the blocks' first slots
__global__ void __launch_bounds__(64) k_bundle_plan(BundleArgs a, int64_t *blk, int64_t nblocks)
{
  const int64_t w = bundle_window(a, (int64_t)blockIdx.x * 64 + threadIdx.x);
  int bound = 0, big = 0;
  if (w >= 0 && a.status[w] == 0) {
    const int64_t tot = a.off[3 * w + 3] - a.off[3 * w];
    // nodes after fusion #2: every uncorrected letter adds one at most
    if (tot < 65535) {
      const int n1 = a.n1[w], lu = (int)(a.off[3 * w + 3] - a.off[3 * w + 2]);
      bound = n1 + lu + 1;
      big = max(n1, lu);
    }
  }
  for (int d = 1; d < 64; d <<= 1) { bound = max(bound, __shfl_xor(bound, d)); big = max(big, __shfl_xor(big, d)); }
  if (threadIdx.x == 0) {
    blk[blockIdx.x] = 64 * (int64_t)bound;
    if (blockIdx.x == 0) blk[nblocks] = 0;
    // the blocks with a window beyond the 8-bit forms: a list of their own, so that their (long-lived) search can start
    // before the other blocks' inputs are through
    if (big > 254) a.cls_list[(int64_t)kBundleHbmWide * a.nblocks + atomicAdd(a.cls_count + kBundleHbmWide, 1)] = (int32_t)blockIdx.x;
  }
}

// The graph the alignment kernels left in HBM lives in every window's own node space: a lane that walks its window's
// arrays touches lines no other lane of its wavefront shares.  This kernel turns the arrays the search reads into
// the per-block layout, 64 x 64 tiles through LDS: rows of one window in (coalesced), rows of one index out (coalesced).
// A block whose windows all have n1, Lu <= 254 gets ONE packed dword per node (in_px); the others the three arrays as
// they are.  On the way it counts, per window, the uncorrected letters fusion #2 will merge into a node of the graph
// (aligned to it, same letter: lpo.c:620-640) -- n1 + Lu less that count is the number of nodes the search will see,
// and the block's largest decides its class (cls_list).
// (a.blocks set: the launch of the wide blocks, whose list k_bundle_plan made; otherwise every block but those)
// WIDE: the launch of the wide blocks (three arrays as they are); the other launch packs a dword per node and needs the
// tiles of the x -> y map and of the ring ids for nothing: 21 KB of LDS per workgroup instead of 47 (seven workgroups
// per CU instead of three, and a lighter neighbour for the alignment kernels' window slots)
template <bool WIDE>
__global__ void __launch_bounds__(256) k_bundle_inputs(BundleArgs a, int all_hbm)
{
  const int64_t blk = bundle_block(a);
  __shared__ uint32_t t_xy[64][65], t_map[WIDE ? 64 : 1][65];           // (packed blocks: t_xy holds the packed dwords)
  __shared__ uint16_t t_ring[WIDE ? 64 : 1][66];
  __shared__ uint8_t t_ys[64][68];
  __shared__ int s_n1[64], s_lu[64], s_fused[64], s_max, s_wide;
  __shared__ int64_t s_nb[64], s_o2[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;        // four wavefronts share the tiles' rows
  if (wave == 0) {
    const int64_t w = bundle_window(a, blk * 64 + lane);
    int n1 = 0, lu = 0;
    int64_t nb = 0, o2 = 0;
    if (w >= 0 && a.status[w] == 0) {
      const int64_t o0 = a.off[3 * w], o3 = a.off[3 * w + 3];
      o2 = a.off[3 * w + 2];
      if (o3 - o0 < 65535) { n1 = a.n1[w]; lu = (int)(o3 - o2); nb = o0 + w; }
    }
    s_n1[lane] = n1; s_lu[lane] = lu; s_nb[lane] = nb; s_o2[lane] = o2; s_fused[lane] = 0;
    int nmax = max(n1, lu);
    for (int d = 1; d < 64; d <<= 1) nmax = max(nmax, __shfl_xor(nmax, d));
    if (lane == 0) { s_max = nmax; s_wide = nmax > 254; }
  }
  __syncthreads();
  const int nmax = s_max;
  if ((s_wide != 0) != WIDE) return;
  constexpr bool wide = WIDE;
  const int my_n1 = s_n1[lane], my_lu = s_lu[lane];
  const int64_t base = a.blk_base[blk];
  int fused = 0;                                              // of window `lane`, this wavefront's share of the indices
  const int64_t my_o2 = s_o2[lane];
  for (int k0 = 0; k0 < nmax; k0 += 64) {
    const int k = k0 + lane;
#pragma unroll
    for (int j = 0; j < 16; ++j) {                          // element k0 + lane of window wl (all sixteen windows' loads in flight)
      const int wl = wave * 16 + j;
      if (k < s_n1[wl]) {
        const int64_t at = s_nb[wl] + k;
        const uint32_t xy = (uint32_t)a.xinfo[at + 1].y, m = a.map16[at];
        const uint16_t r = a.ring1[at];
        if constexpr (wide) { t_xy[wl][lane] = xy; t_map[wl][lane] = m; t_ring[wl][lane] = r; }
        else {
          const bool newring = k == 0 || a.ring1[at - 1] != r;
          t_xy[wl][lane] = (xy & 0xFFFFu) | ((m == kNone32 ? kPxNone : (m & 0xFFu)) << 16) | (newring ? 1u << 24 : 0u);
        }
      }
      if (k < s_lu[wl]) t_ys[wl][lane] = a.sym[s_o2[wl] + k];
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {                          // index k0 + kk of window `lane`
      const int kk = wave * 16 + j, kx = k0 + kk;
      const int64_t at = base + (int64_t)kx * 64 + lane;
      if (kx < my_n1) {
        const uint32_t v = t_xy[lane][kk];
        uint32_t m;
        if constexpr (wide) { m = t_map[lane][kk]; a.in_xy[at] = v; a.in_map[at] = m; a.in_ring[at] = t_ring[lane][kk]; }
        else { m = (v >> 16) & 0xFFu; if (m == kPxNone) m = kNone32; a.in_px[at] = v; }
        // an uncorrected letter aligned to this node and equal to its letter will be fused into it: mostly a letter of
        // this tile (the alignment runs near the diagonal), else one load
        if (m != kNone32 && (int)m < my_lu) {
          const int d = (int)m - k0;
          const uint32_t y = d >= 0 && d < 64 ? (uint32_t)t_ys[lane][d] : (uint32_t)a.sym[my_o2 + m];
          fused += (v & 0xFFu) == y ? 1 : 0;
        }
      }
      if (wide && kx < my_lu) a.in_ys[at] = t_ys[lane][kk];
    }
    if (!wide)                                                // four uncorrected letters per dword (in_map's room)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = wave * 4 + j, kx = k0 + 4 * q;
        if (kx < my_lu) a.in_map[base + (int64_t)(kx >> 2) * 64 + lane] = *reinterpret_cast<const uint32_t *>(&t_ys[lane][4 * q]);
      }
    __syncthreads();
  }
  if (fused) atomicAdd(&s_fused[lane], fused);
  __syncthreads();
  if (wave == 0) {
    int n2 = my_n1 + my_lu - s_fused[lane];
    for (int d = 1; d < 64; d <<= 1) n2 = max(n2, __shfl_xor(n2, d));
    if (lane == 0 && !wide) {
      int cls = kBundleHbmPacked;
      if (!all_hbm)
        for (int q = 3; q >= 0; --q) if (n2 <= kBundleCap[q]) cls = q;
      const int at = atomicAdd(a.cls_count + cls, 1);
      a.cls_list[(int64_t)cls * a.nblocks + at] = (int32_t)blk;
    }
  }
}

// the graph after fusion #2 in the reference's node order (lpo.c:431-459): the nodes of the graph after fusion #1 with
// the uncorrected letters merged in -- fused into the node they are aligned to when the letters agree, a node of the
// same ring otherwise, a node (and ring) of their own when unaligned.  `in` reads the inputs: node(k) = letter |
// flags << 8 | x -> y entry << 16 (kPxWide: none) | "starts a ring" << 31 of node k of the graph after fusion #1, ys(i)
// = uncorrected letter i.  add(ring, letter, has_ref, has_cor, unc position or -1) takes the nodes in order and counts
// them in n.  Every node is read once (the look-ahead to the ring's aligned member re-reads only rings of several nodes).
constexpr uint32_t kPxWide = 0x7FFFu;      // "not aligned" in the 15-bit map field of node()
// One node per turn of ONE loop with one add() in it: the lanes of a wavefront are at different places of the rule
// (an unaligned letter here, a ring member there), and with a call per case every turn ran every case's code.
template <class In, class Add>
__device__ __forceinline__ void bundle_graph(int n1, int Lu, bool act, const In &in, Add &add, const int &n)
{
  int ix = 0, iy = 0, blk_new = -1, ay = -1;
  bool fresh = true, paired = false;       // fresh: node ix's ring has not been looked at; paired: its aligned letter has been dealt with
  uint32_t cur = n1 > 0 ? in.node(0) : 0u, nxt = n1 > 1 ? in.node(1) : 0x80000000u;
  int ycur = Lu > 0 ? in.ys(0) : 0;        // uncorrected letter iy
  for (;;) {
    bool emit = false, r = false, c = false, step_y = false;
    int ring = 0, letter = 0, upos = -1;
    if (ix < n1) {
      if (fresh) {
        if (cur >> 31) blk_new = -1;
        // the ring's first aligned member from here on: the uncorrected letters in front of its partner come first
        ay = -1;
        uint32_t m = (cur >> 16) & 0x7FFFu;
        if (m != kPxWide) ay = (int)m;
        else if (!(nxt >> 31)) {
          m = (nxt >> 16) & 0x7FFFu;
          if (m != kPxWide) ay = (int)m;
          else
            for (int k = ix + 2; k < n1; ++k) {
              const uint32_t v = in.node(k);
              if (v >> 31) break;
              m = (v >> 16) & 0x7FFFu;
              if (m != kPxWide) { ay = (int)m; break; }
            }
        }
        fresh = false; paired = false;
      }
      const bool aligned = ((cur >> 16) & 0x7FFFu) != kPxWide && !paired && iy < Lu;
      const int xl = (int)(cur & 0xFFu);
      emit = true;
      if (ay >= 0 && iy < ay && iy < Lu) { ring = n; letter = ycur; upos = iy; step_y = true; }                    // unaligned letter, a ring of its own
      else if (aligned && xl != ycur) {                                                                              // aligned, another letter: a node of the ring
        if (blk_new < 0) blk_new = n;
        ring = blk_new; letter = ycur; upos = iy; step_y = true; paired = true;
      } else {                                                                                                         // the node itself, the letter fused when it agrees
        if (aligned) { upos = iy; step_y = true; }
        if (blk_new < 0) blk_new = n;
        const int fl = (int)((cur >> 8) & 0xFFu);
        ring = blk_new; letter = xl; r = (fl & kFlagHasRef) != 0; c = (fl & kFlagHasCor) != 0;
        ++ix; cur = nxt; nxt = ix + 1 < n1 ? in.node(ix + 1) : 0x80000000u; fresh = true;
      }
    } else if (act && iy < Lu) { emit = true; ring = n; letter = ycur; upos = iy; step_y = true; }
    if (__ballot(emit) == 0) break;
    if (emit) add(ring, letter, r, c, upos);
    if (step_y) { ++iy; if (iy < Lu) ycur = in.ys(iy); }
  }
}

// inputs in the per-block layout in HBM (entry k of this lane's window at [k * 64]): packed or as three arrays
template <bool PACKED>
struct HbmIn {
  const uint32_t *px, *xy, *mp;
  const uint16_t *rg;
  const uint8_t *y;
  __device__ __forceinline__ uint32_t node(int k) const
  {
    if (PACKED) {
      const uint32_t v = px[(int64_t)k * 64], m = (v >> 16) & 0xFFu;
      return (v & 0xFFFFu) | ((m == kPxNone ? kPxWide : m) << 16) | ((v >> 24) << 31);
    }
    const uint32_t m = mp[(int64_t)k * 64];
    const bool newring = k == 0 || rg[(int64_t)k * 64] != rg[(int64_t)(k - 1) * 64];
    // (a window with an x -> y entry beyond 32,766 has 65,535 letters or more and is not searched)
    return (xy[(int64_t)k * 64] & 0xFFFFu) | ((m == kNone32 ? kPxWide : (m & 0x7FFFu)) << 16) | (newring ? 0x80000000u : 0u);
  }
  __device__ __forceinline__ int ys(int i) const
  {
    if (PACKED) return (int)((mp[(int64_t)(i >> 2) * 64] >> (8 * (i & 3))) & 0xFFu);     // (four letters per dword, in in_map's room)
    return (int)y[(int64_t)i * 64];
  }
};

// first form: node records, scores and best links in HBM -- the windows beyond the LDS classes
template <bool PACKED>
__global__ void __launch_bounds__(64) k_bundle_hbm(BundleArgs a)
{
  const int lane = threadIdx.x;
  const int64_t blk = bundle_block(a);
  const int64_t w = bundle_window(a, blk * 64 + lane);
  bool act = w >= 0;
  int32_t *info = a.info + 8 * (act ? w : 0);
  if (act) {
    for (int k = 0; k < 8; ++k) info[k] = k >= 4 && k <= 6 ? -1 : 0;
    if (a.status[w]) act = false;
  }
  const int64_t o0 = act ? a.off[3 * w] : 0, o1 = act ? a.off[3 * w + 1] : 0, o2 = act ? a.off[3 * w + 2] : 0, o3 = act ? a.off[3 * w + 3] : 0;
  if (o3 - o0 >= 65535) act = false;        // node records hold 16-bit node ids: no bundles for such windows
  const int Lr = (int)(o1 - o0), Lc = (int)(o2 - o1), Lu = (int)(o3 - o2), n1 = act ? a.n1[w] : 0;
  // this lane's column of the block's scratch and inputs: entry k at [k * 64]
  const int64_t base = a.blk_base[blk] + lane;
  HbmIn<PACKED> in;
  in.px = a.in_px + base; in.xy = a.in_xy + base; in.mp = a.in_map + base; in.rg = a.in_ring + base; in.y = a.in_ys + base;
  uint4 *node = a.node + base;
  uint16_t *nodeh = reinterpret_cast<uint16_t *>(node);
  int32_t *score = a.score + base;
  uint16_t *path = a.path + base;
  constexpr int ST = 64;                     // stride between a lane's consecutive records

  // ---- the graph after fusion #2 (lpo.c:431-459 node order), as node records ----
  int n = 0, col = 0, prev_ring = 0, posr = 0, posc = 0;
  int last[3] = {-1, -1, -1};
  auto add = [&](int ring, int letter, bool r, bool c, int upos) {
    if (ring != prev_ring) { ++col; prev_ring = ring; }
    NodeRec rec;
    rec.pos[0] = r ? (uint16_t)posr : (uint16_t)kNone16;
    rec.pos[1] = c ? (uint16_t)posc : (uint16_t)kNone16;
    rec.pos[2] = upos >= 0 ? (uint16_t)upos : (uint16_t)kNone16;
    rec.nxt[0] = rec.nxt[1] = rec.nxt[2] = (uint16_t)kNone16;
    rec.col = (uint16_t)col; rec.letter = (uint8_t)letter; rec.cons = 0;
    store_node(node + (int64_t)n * ST, rec);
    // the previous letter of each source now knows its right neighbour
    if (r) { if (last[0] >= 0) nodeh[8 * ((int64_t)last[0] * ST) + 3] = (uint16_t)n; last[0] = n; ++posr; }
    if (c) { if (last[1] >= 0) nodeh[8 * ((int64_t)last[1] * ST) + 4] = (uint16_t)n; last[1] = n; ++posc; }
    if (upos >= 0) { if (last[2] >= 0) nodeh[8 * ((int64_t)last[2] * ST) + 5] = (uint16_t)n; last[2] = n; }
    ++n;
  };
  bundle_graph(n1, act ? Lu : 0, act, in, add, n);
  const int n2 = n, ncol = col + 1;
  if (act) {
    info[7] = ncol;
    if (ncol != a.ncol[w] || posr != Lr || posc != Lc || n2 > Lr + Lc + Lu) { a.status[w] = 3; act = false; }
  }
  if (a.debug == 1) return;
  // the wavefront's lanes walk their nodes from the same index down: a lane whose window is shorter waits
  int nmax = act ? n2 : 0;
  for (int d = 1; d < 64; d <<= 1) nmax = max(nmax, __shfl_xor(nmax, d));

  // ---- generate_lpo_bundles (heaviest_bundle.c:144-172) ----
  int wt[3] = {1, 1, 1}, bid[3] = {-1, -1, -1};
  const int slen[3] = {Lr, Lc, Lu};
  int nbundled = 0, nseq = 3, ib = 0;
  uint8_t *cons = a.cons + 3 * o0;
  bool go = act;
  for (int pass = 0; pass < kMaxBundles; ++pass) {
    go = go && nbundled < nseq && ib < kMaxBundles;
    // every source bundled: with all weights zero every path scores 0, the best start is the last node (first in the
    // right-to-left scan, :62-66), its path is that one node and the loop ends on `path length < 10` (:152) -- the
    // reference runs the pass for that; here it is not run
    if (go && wt[0] + wt[1] + wt[2] == 0) go = false;
    if (__ballot(go) == 0) break;
    // heaviest_bundle (:16-78): right-to-left over the nodes
    int best = kNeg, ibest = -1;
    for (int i = nmax - 1; i >= 0; --i) {
      if (!go || i >= n2) continue;
      const NodeRec me = load_node(node + (int64_t)i * ST);
      int right_score = 0, right_overlap = 0, best_right = -1;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int rn = me.nxt[s];
        if (rn == (int)kNone16) continue;
        if ((s >= 1 && rn == me.nxt[0]) || (s == 2 && rn == me.nxt[1])) continue;   // add_lpo_link keeps one copy
        const NodeRec rt = load_node(node + (int64_t)rn * ST);
        int ov = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          // contains_pos[q] = pos + 1 for weighted sources of this node, 0 otherwise (:35-39); a right
          // node counts source q when contains_pos[q] == its position (:46-48) -- so a node holding the
          // FIRST letter of a source this node lacks also counts
          if (!wt[q] || rt.pos[q] == (uint16_t)kNone16) continue;
          const int cp = me.pos[q] != (uint16_t)kNone16 ? me.pos[q] + 1 : 0;
          if (cp == (int)rt.pos[q]) ov += wt[q];
        }
        const int sr = score[(int64_t)rn * ST];
        if (ov > right_overlap || (ov == right_overlap && sr > right_score)) {
          right_overlap = ov; right_score = sr; best_right = rn;
        }
      }
      path[(int64_t)i * ST] = (uint16_t)(best_right < 0 ? (int)kNone16 : best_right);
      const int sc = right_score + right_overlap;
      score[(int64_t)i * ST] = sc;
      if (sc > best) { best = sc; ibest = i; }
    }
    if (a.debug == 2) { if (act && best == -12345) info[0] = ibest; return; }
    if (go) {
      // the path, its length, how many letters of each source lie on it
      int plen = 0, cnt[3] = {0, 0, 0};
      for (int i = ibest; i >= 0;) {
        const NodeRec me = load_node(node + (int64_t)i * ST);
        ++plen;
#pragma unroll
        for (int q = 0; q < 3; ++q) cnt[q] += me.pos[q] != (uint16_t)kNone16;
        const int nx = path[(int64_t)i * ST];
        i = nx == (int)kNone16 ? -1 : nx;
      }
      if (plen < 10) go = false;                                              // :152
      else {
        int count = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q)
          if (bid[q] < 0 && (float)slen[q] * a.min_fraction <= (float)cnt[q]) { bid[q] = ib; wt[q] = 0; ++count; }   // :95-100
        // add_path_sequence: the consensus row of this bundle
        uint8_t *row = cons + (int64_t)ib * ncol;
        {                                                                     // '.' everywhere: dwords where aligned
          uint8_t *p = row;
          int left = ncol;
          while (left > 0 && (reinterpret_cast<uintptr_t>(p) & 3u)) { *p++ = '.'; --left; }
          for (; left >= 4; left -= 4, p += 4) *reinterpret_cast<uint32_t *>(p) = 0x2E2E2E2Eu;
          while (left-- > 0) *p++ = '.';
        }
        for (int i = ibest; i >= 0;) {
          const NodeRec me = load_node(node + (int64_t)i * ST);
          row[me.col] = a.tab->chr[me.letter & 31];
          const int nx = path[(int64_t)i * ST];
          i = nx == (int)kNone16 ? -1 : nx;
        }
        info[1 + ib] = count;
        ++ib; ++nseq;
        nbundled += count;
        if (count < 1) go = false;                                            // :166
      }
    }
  }
  if (act) {
    info[0] = ib;
    info[4] = bid[0]; info[5] = bid[1]; info[6] = bid[2];
  }
}

// second form: the whole search in LDS (see the head of the file).  C = nodes per window the workgroup has room for.
//   A[k]  node k: links to the nodes that hold the next letter of ref / cor / unc (bytes 0-2, 0xFF = none),
//         byte 3: bits 0-2 the node holds a letter of ref / cor / unc, bits 3-5 ... the FIRST letter of it
//   S[k]  build: the staged dword of node k of the graph after fusion #1; search: path score << 8 | best right link
//   B[k]  column | letter << 8 (halfword)
//   R     per window a row of bundle_row_stride(C) bytes: build: the uncorrected letters; search: the consensus row
// entry k of lane l at [k * 65 + l] (A, S), [k * 66 + l] (B): the lanes walk their windows in step and touch one row.
// GLOBAL: the same records in the block's scratch in HBM instead (entry k of lane l at [k * 64 + l]; no LDS, so seven
// wavefronts per SIMD where the LDS form has one): the two forms run side by side on the blocks of a class, one bound by
// the latency of its single wavefront per SIMD, the other by L2 traffic.
template <int C, bool GLOBAL>
__global__ void __launch_bounds__(64) k_bundle_lds(BundleArgs a)
{
  extern __shared__ __align__(16) uint32_t lds_b[];
  constexpr int ST = GLOBAL ? 64 : kLdsSt, STB = GLOBAL ? 64 : 66, RS = bundle_row_stride(C), RST = GLOBAL ? 64 : 1;
  const int lane = threadIdx.x;
  const int64_t blk = bundle_block(a);
  const int64_t gbase = a.blk_base[blk] + lane;
  // (GLOBAL: a packed block leaves its slots of in_xy and in_ys unused -- the node dwords and the rows go there)
  uint32_t *A = GLOBAL ? a.in_xy + gbase : lds_b + threadIdx.x;
  uint32_t *S = GLOBAL ? reinterpret_cast<uint32_t *>(a.score) + gbase : lds_b + C * kLdsSt + threadIdx.x;
  uint16_t *B = GLOBAL ? a.path + gbase : reinterpret_cast<uint16_t *>(lds_b + 2 * C * kLdsSt) + threadIdx.x;
  uint8_t *R = GLOBAL ? a.in_ys + gbase
                      : reinterpret_cast<uint8_t *>(reinterpret_cast<uint16_t *>(lds_b + 2 * C * kLdsSt) + C * 66) + (size_t)threadIdx.x * RS;
  const int64_t w = bundle_window(a, blk * 64 + lane);
  bool act = w >= 0;
  int32_t *info = a.info + 8 * (act ? w : 0);
  if (act) {
    for (int k = 0; k < 8; ++k) info[k] = k >= 4 && k <= 6 ? -1 : 0;
    if (a.status[w]) act = false;
  }
  const int64_t o0 = act ? a.off[3 * w] : 0, o1 = act ? a.off[3 * w + 1] : 0, o2 = act ? a.off[3 * w + 2] : 0, o3 = act ? a.off[3 * w + 3] : 0;
  if (o3 - o0 >= 65535) act = false;
  const int Lr = (int)(o1 - o0), Lc = (int)(o2 - o1), Lu = act ? (int)(o3 - o2) : 0, n1 = act ? a.n1[w] : 0;
  if (n1 > C || Lu > C) { if (act) a.status[w] = 3; act = false; }            // (the block's class says otherwise)
  // ---- inputs in: this lane's column of the block's arrays (entry k at [k * 64]: a row of 64 lanes per load) ----
  if (!GLOBAL) {
    const int64_t base = a.blk_base[blk] + lane;
    const uint32_t *px = a.in_px + base, *y4 = a.in_map + base;     // (packed blocks: four uncorrected letters per dword)
    const int n1e = act ? n1 : 0, lue = act ? (Lu + 3) >> 2 : 0;
    int m1 = n1e, mu = lue;
    for (int d = 1; d < 64; d <<= 1) { m1 = max(m1, __shfl_xor(m1, d)); mu = max(mu, __shfl_xor(mu, d)); }
    uint32_t *R4 = reinterpret_cast<uint32_t *>(R);
    constexpr int UN = 8;                                           // loads in flight per trip
    for (int k = 0; k < m1; k += UN) {
      uint32_t v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) v[u] = k + u < n1e ? px[(int64_t)(k + u) * 64] : 0u;
#pragma unroll
      for (int u = 0; u < UN; ++u) if (k + u < n1e) S[(k + u) * ST] = v[u];
    }
    for (int i = 0; i < mu; i += UN) {
      uint32_t v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) v[u] = i + u < lue ? y4[(int64_t)(i + u) * 64] : 0u;
#pragma unroll
      for (int u = 0; u < UN; ++u) if (i + u < lue) R4[i + u] = v[u];
    }
  }
  struct LdsIn {
    const uint32_t *S, *gpx, *gy4;
    const uint8_t *R;
    __device__ __forceinline__ uint32_t node(int k) const
    {
      const uint32_t v = GLOBAL ? gpx[(int64_t)k * 64] : S[k * kLdsSt], m = (v >> 16) & 0xFFu;
      return (v & 0xFFFFu) | ((m == kPxNone ? kPxWide : m) << 16) | ((v >> 24) << 31);
    }
    __device__ __forceinline__ int ys(int i) const
    {
      if (GLOBAL) return (int)((gy4[(int64_t)(i >> 2) * 64] >> (8 * (i & 3))) & 0xFFu);
      return (int)R[i];
    }
  } in{S, a.in_px + gbase, a.in_map + gbase, R};

  // ---- the graph after fusion #2, as LDS records ----
  int n = 0, col = 0, prev_ring = 0, posr = 0, posc = 0;
  int last0 = -1, last1 = -1, last2 = -1;
  bool over = false;
  auto add = [&](int ring, int letter, bool r, bool c, int upos) {
    // (plain arithmetic on the counters: with `if (r) ++posr` in two places the compiler kept posr / posc in scratch
    // memory behind a computed address, a trip to HBM per node)
    col += ring != prev_ring ? 1 : 0;
    prev_ring = ring;
    const bool ok = n < C, u = upos >= 0;
    over = over || !ok;
    const uint32_t flags = (r ? 1u : 0u) | (c ? 2u : 0u) | (u ? 4u : 0u) | (r && posr == 0 ? 8u : 0u) | (c && posc == 0 ? 16u : 0u) |
                           (upos == 0 ? 32u : 0u);
    uint8_t *Ab = reinterpret_cast<uint8_t *>(A);
    if (ok) {
      A[n * ST] = 0x00FFFFFFu | (flags << 24);
      B[n * STB] = (uint16_t)((uint32_t)col | ((uint32_t)letter << 8));
      if (r && last0 >= 0) Ab[4 * (last0 * ST) + 0] = (uint8_t)n;
      if (c && last1 >= 0) Ab[4 * (last1 * ST) + 1] = (uint8_t)n;
      if (u && last2 >= 0) Ab[4 * (last2 * ST) + 2] = (uint8_t)n;
    }
    last0 = r ? n : last0; last1 = c ? n : last1; last2 = u ? n : last2;
    posr += r ? 1 : 0; posc += c ? 1 : 0;
    n += 1;
  };
  bundle_graph(n1, Lu, act, in, add, n);
  const int n2 = n, ncol = col + 1;
  if (act) {
    info[7] = ncol;
    if (over || ncol != a.ncol[w] || posr != Lr || posc != Lc || n2 > Lr + Lc + Lu) { a.status[w] = 3; act = false; }
  }
  if (a.debug == 1) return;
  int nmax = act ? n2 : 0;
  for (int d = 1; d < 64; d <<= 1) nmax = max(nmax, __shfl_xor(nmax, d));

  // ---- generate_lpo_bundles (heaviest_bundle.c:144-172) ----
  int wt[3] = {1, 1, 1}, bid[3] = {-1, -1, -1};
  const int slen[3] = {Lr, Lc, Lu};
  int nbundled = 0, nseq = 3, ib = 0;
  uint8_t *cons = a.cons + 3 * o0;
  bool go = act;
  for (int pass = 0; pass < kMaxBundles; ++pass) {
    go = go && nbundled < nseq && ib < kMaxBundles;
    if (go && wt[0] + wt[1] + wt[2] == 0) go = false;          // (see k_bundle_hbm: the all-zero pass ends the loop either way)
    if (__ballot(go) == 0) break;
    if (go) {                                                  // '.' in the lane's row
      if (GLOBAL) for (int cc = 0; cc < ncol; ++cc) R[(int64_t)cc * RST] = '.';
      else {
        uint32_t *Rd = reinterpret_cast<uint32_t *>(R);
        for (int d = 0; 4 * d < ncol; ++d) Rd[d] = 0x2E2E2E2Eu;
      }
    }
    // heaviest_bundle (:16-78): right to left over the nodes; the weights as a mask over the has / first bits
    const uint32_t wmask = (wt[0] ? 1u : 0u) | (wt[1] ? 2u : 0u) | (wt[2] ? 4u : 0u);
    int best = kNeg, ibest = -1;
    for (int i = nmax - 1; i >= 0; --i) {
      const bool on = go && i < n2;
      const uint32_t me = on ? A[i * ST] : 0x00FFFFFFu;
      const uint32_t l0 = me & 0xFFu, l1 = (me >> 8) & 0xFFu, l2 = (me >> 16) & 0xFFu, has = (me >> 24) & 7u;
      // the link list in the reference's order (lpo.c:227-241): ref's, cor's, unc's right neighbour, one copy each
      const bool use0 = l0 != 0xFFu, use1 = l1 != 0xFFu && l1 != l0, use2 = l2 != 0xFFu && l2 != l0 && l2 != l1;
      int right_score = 0, right_overlap = 0, best_right = -1;
      auto offer = [&](bool use, uint32_t rn, uint32_t ra, uint32_t rs) {
        if (!use) return;
        // a source this node holds counts when rn is its link's target, one it lacks when rn holds its first letter (:35-48)
        const uint32_t mine = (l0 == rn ? 1u : 0u) | (l1 == rn ? 2u : 0u) | (l2 == rn ? 4u : 0u);
        const uint32_t first = (ra >> 27) & 7u;
        const int ov = (int)__popc(((mine & has) | (first & ~has)) & wmask);
        const int sr = (int)(rs >> 8);
        if (ov > right_overlap || (ov == right_overlap && sr > right_score)) { right_overlap = ov; right_score = sr; best_right = (int)rn; }
      };
      // most nodes of most windows have ONE right neighbour: the wavefront reads a second and third only where a lane has them
      const unsigned long long more = __ballot(use1 || use2);
      const uint32_t a0 = use0 ? A[l0 * ST] : 0u, s0 = use0 ? S[l0 * ST] : 0u;
      if (more == 0ull) offer(use0, l0, a0, s0);
      else {
        const uint32_t a1 = use1 ? A[l1 * ST] : 0u, a2 = use2 ? A[l2 * ST] : 0u;
        const uint32_t s1 = use1 ? S[l1 * ST] : 0u, s2 = use2 ? S[l2 * ST] : 0u;
        offer(use0, l0, a0, s0);
        offer(use1, l1, a1, s1);
        offer(use2, l2, a2, s2);
      }
      if (on) {
        const int sc = right_score + right_overlap;
        S[i * ST] = ((uint32_t)sc << 8) | (uint32_t)(best_right < 0 ? 0xFF : best_right);
        if (sc > best) { best = sc; ibest = i; }
      }
    }
    if (a.debug == 2) { if (act && best == -12345) info[0] = ibest; return; }
    // the path: its length, how many letters of each source lie on it, its letters into the window's row
    int plen = 0, cnt[3] = {0, 0, 0};
    {
      int i = go ? ibest : -1;
      while (__ballot(i >= 0) != 0) {
        if (i >= 0) {
          const uint32_t me = A[i * ST], sc = S[i * ST];
          const uint32_t cb = B[i * STB];
          ++plen;
          cnt[0] += (int)((me >> 24) & 1u); cnt[1] += (int)((me >> 25) & 1u); cnt[2] += (int)((me >> 26) & 1u);
          R[(int64_t)(cb & 0xFFu) * RST] = a.tab->chr[(cb >> 8) & 31u];
          const int nx = (int)(sc & 0xFFu);
          i = nx == 0xFF ? -1 : nx;
        }
      }
    }
    if (go) {
      if (plen < 10) go = false;                                              // :152
      else {
        int count = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q)
          if (bid[q] < 0 && (float)slen[q] * a.min_fraction <= (float)cnt[q]) { bid[q] = ib; wt[q] = 0; ++count; }   // :95-100
        // add_path_sequence's row out: dwords from the lane's row in LDS (the destination is byte-aligned: unaligned
        // dword stores), the last bytes one by one
        uint8_t *dst = cons + (int64_t)ib * ncol;
        const uint32_t *R4 = reinterpret_cast<const uint32_t *>(R);
        int c4 = 0;
        for (; 4 * c4 + 4 <= ncol; ++c4) {
          uint32_t v;
          if (GLOBAL) v = (uint32_t)R[(int64_t)(4 * c4) * RST] | ((uint32_t)R[(int64_t)(4 * c4 + 1) * RST] << 8) |
                          ((uint32_t)R[(int64_t)(4 * c4 + 2) * RST] << 16) | ((uint32_t)R[(int64_t)(4 * c4 + 3) * RST] << 24);
          else v = R4[c4];
          __builtin_memcpy(dst + 4 * c4, &v, 4);
        }
        for (int c1 = 4 * c4; c1 < ncol; ++c1) dst[c1] = R[(int64_t)c1 * RST];
        info[1 + ib] = count;
        ++ib; ++nseq;
        nbundled += count;
        if (count < 1) go = false;                                            // :166
      }
    }
  }
  if (act) {
    info[0] = ib;
    info[4] = bid[0]; info[5] = bid[1]; info[6] = bid[2];
  }
}

// consensus rows of all windows, packed: window w's nbundle[w] rows at out + cons_off[w]
__global__ void __launch_bounds__(256) k_cons_pack(const uint8_t *__restrict__ cons, const int64_t *__restrict__ off,
                                                    const int32_t *__restrict__ info, const int64_t *__restrict__ cons_off,
                                                    uint8_t *__restrict__ out, int64_t n)
{
  const int64_t w = blockIdx.x;
  if (w >= n) return;
  const int64_t bytes = (int64_t)info[8 * w] * info[8 * w + 7];
  const uint8_t *src = cons + 3 * off[3 * w];
  uint8_t *dst = out + cons_off[w];
  for (int64_t i = threadIdx.x; i < bytes; i += blockDim.x) dst[i] = src[i];
}

}  // namespace elector

using namespace elector;

extern "C" int elector_ctx_keep_graph(elector_ctx *c, int on)
{
  if (!c) return ELECTOR_E_INVAL;
  std::lock_guard<std::mutex> lock(c->mu);
  c->keep_graph = on != 0;
  return ELECTOR_OK;
}

template <int C>
static int launch_bundle_lds_t(const BundleArgs &a, int blocks, hipStream_t st)
{
  static DeviceOnce once;
  if (once.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_bundle_lds<C, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bundle_lds_bytes(C)) != hipSuccess)
      return -1;
    once.done();
  }
  hipLaunchKernelGGL((k_bundle_lds<C, false>), dim3((unsigned)blocks), dim3(64), bundle_lds_bytes(C), st, a);
  return 0;
}

static int launch_bundle_lds(const BundleArgs &a, int cls, int blocks, hipStream_t st)
{
  switch (cls) {
    case 0: return launch_bundle_lds_t<kBundleCap[0]>(a, blocks, st);
    case 1: return launch_bundle_lds_t<kBundleCap[1]>(a, blocks, st);
    case 2: return launch_bundle_lds_t<kBundleCap[2]>(a, blocks, st);
    case 3: return launch_bundle_lds_t<kBundleCap[3]>(a, blocks, st);
  }
  return -2;
}

// the bundle search of the last batch queued on the context's stream; results stay in the context's device buffers
static int bundles_enqueue(elector_ctx *c, int64_t n, float minimum_fraction)
{
  if (!c->keep_graph || !c->graph_valid || n != c->last_n)
    return elector_fail(c, ELECTOR_E_INVAL, "no graph kept: call elector_ctx_keep_graph(ctx, 1) before the POA batch");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = c->d_bcons.ensure((size_t)3 * c->last_total + 64) | c->d_binfo.ensure((size_t)n * 32) |
           c->d_rowoff.ensure((size_t)(n + 1) * 8);
  if (rc) return elector_fail(c, ELECTOR_E_NOMEM, "bundle workspace");
  hipStream_t st = c->stream;
  BundleArgs a;
  a.n = n;
  a.off = c->d_off.as<int64_t>();
  a.sym = c->d_sym.as<uint8_t>();
  a.xinfo = c->d_xinfo.as<int2>();
  a.ring1 = c->d_ring1.as<uint16_t>();
  a.map16 = c->d_map16.as<uint32_t>();
  a.n1 = c->d_n1.as<int32_t>();
  a.ncol = c->last_ncol;
  a.status = c->last_status;
  a.tab = c->d_tab.as<DevTables>();
  a.min_fraction = minimum_fraction;
  a.debug = std::getenv("ELECTOR_DEBUG_BUNDLE") ? std::atoi(std::getenv("ELECTOR_DEBUG_BUNDLE")) : 0;
  // the windows in the order of the batch's class lists (d_list: the fused classes, largest windows first within a
  // class; d_perm: the generic list), 64 per block
  a.order_a = c->d_list.as<uint32_t>();
  a.order_b = c->d_perm.as<uint32_t>();
  a.n_a = n - c->last_n_generic;
  const int64_t nblocks = (n + 63) / 64;
  size_t tmp_bytes = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (int64_t *)nullptr, (int64_t *)nullptr, (int)(nblocks + 1), st);
  if (c->d_bplan.ensure((size_t)(nblocks + 1) * 16 + tmp_bytes + 256)) return elector_fail(c, ELECTOR_E_NOMEM, "bundle plan");
  int64_t *blk_room = c->d_bplan.as<int64_t>(), *blk_base = blk_room + (nblocks + 1);
  a.blk_base = blk_base;
  a.node = nullptr; a.score = nullptr; a.path = nullptr; a.cons = nullptr; a.info = nullptr;
  a.in_xy = a.in_map = a.in_px = nullptr; a.in_ring = nullptr; a.in_ys = nullptr;
  a.cls_count = a.cls_list = nullptr; a.nblocks = nblocks; a.blocks = nullptr;
  if (c->d_bcls.ensure((size_t)kBundleClasses * (size_t)(nblocks + 1) * 4 + 64)) return elector_fail(c, ELECTOR_E_NOMEM, "bundle classes");
  a.cls_count = c->d_bcls.as<int32_t>();
  a.cls_list = a.cls_count + 16;
  timed_begin(c, 5, st);
  HIPCHK(c, hipMemsetAsync(a.cls_count, 0, 64, st));
  hipLaunchKernelGGL(k_bundle_plan, dim3((unsigned)nblocks), dim3(64), 0, st, a, blk_room, nblocks);
  if (hipcub::DeviceScan::ExclusiveSum(blk_base + (nblocks + 1), tmp_bytes, blk_room, blk_base, (int)(nblocks + 1), st) != hipSuccess)
    return elector_fail(c, ELECTOR_E_HIP, "bundle plan scan");
  int64_t slots = 0;
  int32_t cnt[kBundleClasses] = {};
  HIPCHK(c, hipMemcpyAsync(&slots, blk_base + nblocks, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(&cnt[kBundleHbmWide], a.cls_count + kBundleHbmWide, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  rc = c->d_bnode.ensure((size_t)(slots + 64) * 16) | c->d_bscore.ensure((size_t)(slots + 64) * 4) | c->d_bpath.ensure((size_t)(slots + 64) * 2) |
       c->d_bin.ensure((size_t)(slots + 64) * 15 + 64);
  if (rc) return elector_fail(c, ELECTOR_E_NOMEM, "bundle scratch");
  a.in_xy = c->d_bin.as<uint32_t>();
  a.in_map = a.in_xy + (slots + 64);
  a.in_px = a.in_map + (slots + 64);
  a.in_ring = reinterpret_cast<uint16_t *>(a.in_px + (slots + 64));
  a.in_ys = reinterpret_cast<uint8_t *>(a.in_ring + (slots + 64));
  a.node = c->d_bnode.as<uint4>();
  a.score = c->d_bscore.as<int32_t>();
  a.path = c->d_bpath.as<uint16_t>();
  a.cons = c->d_bcons.as<uint8_t>();
  a.info = c->d_binfo.as<int32_t>();
  a.nblocks = nblocks;
  a.blocks = nullptr;
  // ELECTOR_BUNDLE_HBM=1: every block through the first form (A/B, and the parity of the two forms)
  const bool all_hbm = std::getenv("ELECTOR_BUNDLE_HBM") && std::atoi(std::getenv("ELECTOR_BUNDLE_HBM")) != 0;
  // The launches side by side on the context's auxiliary streams.  The few blocks with a window beyond 254 letters (the
  // first form on the wide inputs) hold windows of thousands of nodes, one wavefront each that lives for milliseconds:
  // they get their inputs and start first, on a chain of their own.
  if (!c->aux_ready) {
    for (int k = 0; k < elector_ctx::kAux; ++k) {
      if (c->make_stream(&c->aux[k])) return elector_fail(c, ELECTOR_E_HIP, "stream");
      HIPCHK(c, hipEventCreateWithFlags(&c->aux_done[k], hipEventDisableTiming));
    }
    HIPCHK(c, hipEventCreateWithFlags(&c->fork, hipEventDisableTiming));
    c->aux_ready = true;
  }
  static_assert(elector_ctx::kAux >= 4, "four launch chains");
  HIPCHK(c, hipEventRecord(c->fork, st));
  bool used[elector_ctx::kAux] = {};
  auto chain = [&](int k) -> hipStream_t {
    if (!used[k]) { (void)hipStreamWaitEvent(c->aux[k], c->fork, 0); used[k] = true; }
    return c->aux[k];
  };
  if (cnt[kBundleHbmWide] > 0) {
    hipStream_t sx = chain(0);
    a.blocks = a.cls_list + (int64_t)kBundleHbmWide * nblocks;
    hipLaunchKernelGGL(k_bundle_inputs<true>, dim3((unsigned)cnt[kBundleHbmWide]), dim3(256), 0, sx, a, all_hbm ? 1 : 0);
    hipLaunchKernelGGL(k_bundle_hbm<false>, dim3((unsigned)cnt[kBundleHbmWide]), dim3(64), 0, sx, a);
    a.blocks = nullptr;
  }
  hipLaunchKernelGGL(k_bundle_inputs<false>, dim3((unsigned)nblocks), dim3(256), 0, st, a, all_hbm ? 1 : 0);
  HIPCHK(c, hipMemcpyAsync(cnt, a.cls_count, 4 * kBundleHbmWide, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  HIPCHK(c, hipEventRecord(c->fork, st));                      // (the chains below start behind the inputs)
  int64_t seen = cnt[kBundleHbmWide];
  if (std::getenv("ELECTOR_DEBUG_BINS"))
    std::fprintf(stderr, "[elector] bundle search: blocks of 64 windows per class (LDS 52 / 69 / 104 / 208 nodes, HBM packed, HBM wide): %d %d %d %d %d %d\n",
                 cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5]);
  for (int q = kBundleHbmPacked; q >= 0; --q) {              // the long-lived ones first
    seen += cnt[q];
    if (cnt[q] <= 0) continue;
    static const int stream_of[kBundleHbmPacked + 1] = {1, 2, 3, 2, 1};
    const int k = stream_of[q];
    hipStream_t sx = c->aux[k];
    (void)hipStreamWaitEvent(sx, c->fork, 0);
    used[k] = true;
    a.blocks = a.cls_list + (int64_t)q * nblocks;
    if (q == kBundleHbmPacked) hipLaunchKernelGGL(k_bundle_hbm<true>, dim3((unsigned)cnt[q]), dim3(64), 0, sx, a);
    else {
      // ELECTOR_BUNDLE_GLOBAL_PCT: share of an LDS class's blocks that go through the same code on records in HBM
      // (E. coli batch: 7.5 ms with 0, 5.8 / 5.2 / 5.6 with 30 / 50 / 70, 6.1 with 100)
      // (one number for every class, or four separated by commas: the classes of 52 / 69 / 104 / 208 nodes)
      int gpct = kBundleGlobalPct[q];
      if (const char *e = std::getenv("ELECTOR_BUNDLE_GLOBAL_PCT")) {
        int v[4] = {0, 0, 0, 0};
        const int got = std::sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]);
        if (got >= 1) gpct = std::max(0, std::min(100, got == 4 ? v[q] : v[0]));
      }
      const int ng = (int)((int64_t)cnt[q] * gpct / 100), nl = cnt[q] - ng;
      if (nl > 0 && (rc = launch_bundle_lds(a, q, nl, sx)) != 0) return elector_fail(c, ELECTOR_E_HIP, "k_bundle_lds attribute");
      if (ng > 0) {
        a.blocks += nl;
        hipLaunchKernelGGL((k_bundle_lds<254, true>), dim3((unsigned)ng), dim3(64), 0, st, a);      // (the context's own stream is free meanwhile)
      }
    }
  }
  for (int k = 0; k < elector_ctx::kAux; ++k)
    if (used[k]) { HIPCHK(c, hipEventRecord(c->aux_done[k], c->aux[k])); HIPCHK(c, hipStreamWaitEvent(st, c->aux_done[k], 0)); }
  timed_end(c, st);
  HIPCHK(c, hipGetLastError());
  if (seen != nblocks) return elector_fail(c, ELECTOR_E_HIP, "bundle blocks lost between the classes");
  return ELECTOR_OK;
}

int elector_bundles_flush(elector_ctx *c)
{
  if (!c->bundles_pending) return ELECTOR_OK;
  c->bundles_pending = false;
  return bundles_enqueue(c, c->bundles_n, c->bundles_fraction);
}

extern "C" int elector_poa_bundles_enqueue(elector_ctx *c, int64_t n, float minimum_fraction)
{
  if (!c) return ELECTOR_E_INVAL;
  if (n < 0) return elector_fail(c, ELECTOR_E_INVAL, "bad arguments");
  std::lock_guard<std::mutex> lock(c->mu);
  if (n == 0) return ELECTOR_OK;
  if (c->bundles_now) return bundles_enqueue(c, n, minimum_fraction);
  // noted; queued at the context's next call that waits for it anyway (ctx.h).  What can be refused is refused here.
  if (!c->keep_graph || !c->graph_valid || n != c->last_n)
    return elector_fail(c, ELECTOR_E_INVAL, "no graph kept: call elector_ctx_keep_graph(ctx, 1) before the POA batch");
  c->bundles_pending = true;
  c->bundles_n = n;
  c->bundles_fraction = minimum_fraction;
  return ELECTOR_OK;
}

extern "C" int elector_poa_bundles(elector_ctx *c, int64_t n, float minimum_fraction, uint8_t *cons_rows,
                                   int64_t cons_cap, int64_t *cons_off, int32_t *info)
{
  if (!c) return ELECTOR_E_INVAL;
  if (n < 0 || !cons_off || (n > 0 && !info)) return elector_fail(c, ELECTOR_E_INVAL, "bad arguments");
  std::lock_guard<std::mutex> lock(c->mu);
  cons_off[0] = 0;
  if (n == 0) return ELECTOR_OK;
  c->bundles_pending = false;                  // (a noted search of the same batch: this call runs it)
  int rc = bundles_enqueue(c, n, minimum_fraction);
  if (rc) return rc;
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemcpyAsync(info, c->d_binfo.p, (size_t)n * 32, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  for (int64_t w = 0; w < n; ++w) cons_off[w + 1] = cons_off[w] + (int64_t)info[8 * w] * info[8 * w + 7];
  if (cons_off[n] > cons_cap || (cons_off[n] > 0 && !cons_rows)) return elector_fail(c, ELECTOR_E_INVAL, "consensus buffer too small");
  if (cons_off[n] > 0) {
    rc = c->d_rows.ensure((size_t)cons_off[n] + 64);
    if (rc) return elector_fail(c, ELECTOR_E_NOMEM, "consensus rows");
    HIPCHK(c, hipMemcpyAsync(c->d_rowoff.p, cons_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_cons_pack, dim3((unsigned)n), dim3(256), 0, st, c->d_bcons.as<uint8_t>(), c->d_off.as<int64_t>(),
                       c->d_binfo.as<int32_t>(), c->d_rowoff.as<int64_t>(), c->d_rows.as<uint8_t>(), n);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(cons_rows, c->d_rows.p, (size_t)cons_off[n], hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
  }
  return ELECTOR_OK;
}
