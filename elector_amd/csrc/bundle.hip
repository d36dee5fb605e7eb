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
// Latency / L2 bound; integer only except the one float comparison of :96.
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
};

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

// room of every block: 64 x the largest node bound (|PO| after fusion #1 + Lu + 1) among its windows; blk[b] holds it, a scan turns
// the array into the blocks' first slots
__global__ void __launch_bounds__(64) k_bundle_plan(BundleArgs a, int64_t *blk, int64_t nblocks)
{
  const int64_t w = bundle_window(a, (int64_t)blockIdx.x * 64 + threadIdx.x);
  int bound = 0;
  if (w >= 0 && a.status[w] == 0) {
    const int64_t tot = a.off[3 * w + 3] - a.off[3 * w];
    // nodes after fusion #2: every uncorrected letter adds one at most
    if (tot < 65535) bound = a.n1[w] + (int)(a.off[3 * w + 3] - a.off[3 * w + 2]) + 1;
  }
  for (int d = 1; d < 64; d <<= 1) bound = max(bound, __shfl_xor(bound, d));
  if (threadIdx.x == 0) { blk[blockIdx.x] = 64 * (int64_t)bound; if (blockIdx.x == 0) blk[nblocks] = 0; }
}

// The graph the alignment kernels left in HBM lives in every window's own node space: a lane that walks its window's
// arrays touches lines no other lane of its wavefront shares.  This kernel turns the four arrays the search reads into
// the per-block layout, 64 x 64 tiles through LDS: rows of one window in (coalesced), rows of one index out (coalesced).
__global__ void __launch_bounds__(256) k_bundle_inputs(BundleArgs a)
{
  __shared__ uint32_t t_xy[64][65], t_map[64][65];
  __shared__ uint16_t t_ring[64][66];
  __shared__ uint8_t t_ys[64][68];
  __shared__ int s_n1[64], s_lu[64], s_max;
  __shared__ int64_t s_nb[64], s_o2[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;        // four wavefronts share the tiles' rows
  if (wave == 0) {
    const int64_t w = bundle_window(a, (int64_t)blockIdx.x * 64 + lane);
    int n1 = 0, lu = 0;
    int64_t nb = 0, o2 = 0;
    if (w >= 0 && a.status[w] == 0) {
      const int64_t o0 = a.off[3 * w], o3 = a.off[3 * w + 3];
      o2 = a.off[3 * w + 2];
      if (o3 - o0 < 65535) { n1 = a.n1[w]; lu = (int)(o3 - o2); nb = o0 + w; }
    }
    s_n1[lane] = n1; s_lu[lane] = lu; s_nb[lane] = nb; s_o2[lane] = o2;
    int nmax = max(n1, lu);
    for (int d = 1; d < 64; d <<= 1) nmax = max(nmax, __shfl_xor(nmax, d));
    if (lane == 0) s_max = nmax;
  }
  __syncthreads();
  const int nmax = s_max;
  const int my_n1 = s_n1[lane], my_lu = s_lu[lane];
  const int64_t base = a.blk_base[blockIdx.x];
  for (int k0 = 0; k0 < nmax; k0 += 64) {
    const int k = k0 + lane;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {                          // element k0 + lane of window wl
      const int wl = wave * 16 + j;
      if (k < s_n1[wl]) {
        const int64_t at = s_nb[wl] + k;
        t_xy[wl][lane] = (uint32_t)a.xinfo[at + 1].y;
        t_map[wl][lane] = a.map16[at];
        t_ring[wl][lane] = a.ring1[at];
      }
      if (k < s_lu[wl]) t_ys[wl][lane] = a.sym[s_o2[wl] + k];
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {                          // index k0 + kk of window `lane`
      const int kk = wave * 16 + j, kx = k0 + kk;
      const int64_t at = base + (int64_t)kx * 64 + lane;
      if (kx < my_n1) { a.in_xy[at] = t_xy[lane][kk]; a.in_map[at] = t_map[lane][kk]; a.in_ring[at] = t_ring[lane][kk]; }
      if (kx < my_lu) a.in_ys[at] = t_ys[lane][kk];
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(64) k_bundle(BundleArgs a)
{
  const int lane = threadIdx.x;
  const int64_t w = bundle_window(a, (int64_t)blockIdx.x * 64 + lane);
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
  const int64_t base = a.blk_base[blockIdx.x] + lane;
  const uint32_t *in_xy = a.in_xy + base, *in_map = a.in_map + base;
  const uint16_t *in_ring = a.in_ring + base;
  const uint8_t *in_ys = a.in_ys + base;
  auto ring1 = [&](int k) { return (int)in_ring[(int64_t)k * 64]; };
  auto x2y = [&](int k) { return in_map[(int64_t)k * 64]; };
  auto ys = [&](int i) { return (int)in_ys[(int64_t)i * 64]; };
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
  int iy = 0, blk_old = -1, blk_new = -1;
  for (int ix = 0; ix < n1; ++ix) {
    const int r0 = ring1(ix);
    if (r0 != blk_old) { blk_old = r0; blk_new = -1; }
    for (int k = ix; k < n1 && ring1(k) == r0; ++k) {
      const int ay = (int)x2y(k);
      if (ay != (int)kNone32) {
        while (iy < ay && iy < Lu) { add(n, ys(iy), false, false, iy); ++iy; }
        break;
      }
    }
    const int xi = (int)in_xy[(int64_t)ix * 64];
    const int letter = xi & 0xFF, fl = xi >> 8;
    int fused_pos = -1;
    if (x2y(ix) != kNone32 && iy < Lu) {
      if (letter == ys(iy)) fused_pos = iy;
      else {
        if (blk_new < 0) blk_new = n;
        add(blk_new, ys(iy), false, false, iy);
      }
      ++iy;
    }
    if (blk_new < 0) blk_new = n;
    add(blk_new, letter, (fl & kFlagHasRef) != 0, (fl & kFlagHasCor) != 0, fused_pos);
  }
  if (act) while (iy < Lu) { add(n, ys(iy), false, false, iy); ++iy; }
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
  hipLaunchKernelGGL(k_bundle_plan, dim3((unsigned)nblocks), dim3(64), 0, st, a, blk_room, nblocks);
  if (hipcub::DeviceScan::ExclusiveSum(blk_base + (nblocks + 1), tmp_bytes, blk_room, blk_base, (int)(nblocks + 1), st) != hipSuccess)
    return elector_fail(c, ELECTOR_E_HIP, "bundle plan scan");
  int64_t slots = 0;
  HIPCHK(c, hipMemcpyAsync(&slots, blk_base + nblocks, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  rc = c->d_bnode.ensure((size_t)(slots + 64) * 16) | c->d_bscore.ensure((size_t)(slots + 64) * 4) | c->d_bpath.ensure((size_t)(slots + 64) * 2) |
       c->d_bin.ensure((size_t)(slots + 64) * 11 + 64);
  if (rc) return elector_fail(c, ELECTOR_E_NOMEM, "bundle scratch");
  a.in_xy = c->d_bin.as<uint32_t>();
  a.in_map = a.in_xy + (slots + 64);
  a.in_ring = reinterpret_cast<uint16_t *>(a.in_map + (slots + 64));
  a.in_ys = reinterpret_cast<uint8_t *>(a.in_ring + (slots + 64));
  a.node = c->d_bnode.as<uint4>();
  a.score = c->d_bscore.as<int32_t>();
  a.path = c->d_bpath.as<uint16_t>();
  a.cons = c->d_bcons.as<uint8_t>();
  a.info = c->d_binfo.as<int32_t>();
  timed_begin(c, 5, st);
  hipLaunchKernelGGL(k_bundle_inputs, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_bundle, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, a);
  timed_end(c, st);
  HIPCHK(c, hipGetLastError());
  return ELECTOR_OK;
}

extern "C" int elector_poa_bundles_enqueue(elector_ctx *c, int64_t n, float minimum_fraction)
{
  if (!c) return ELECTOR_E_INVAL;
  if (n < 0) return elector_fail(c, ELECTOR_E_INVAL, "bad arguments");
  std::lock_guard<std::mutex> lock(c->mu);
  if (n == 0) return ELECTOR_OK;
  return bundles_enqueue(c, n, minimum_fraction);
}

extern "C" int elector_poa_bundles(elector_ctx *c, int64_t n, float minimum_fraction, uint8_t *cons_rows,
                                   int64_t cons_cap, int64_t *cons_off, int32_t *info)
{
  if (!c) return ELECTOR_E_INVAL;
  if (n < 0 || !cons_off || (n > 0 && !info)) return elector_fail(c, ELECTOR_E_INVAL, "bad arguments");
  std::lock_guard<std::mutex> lock(c->mu);
  cons_off[0] = 0;
  if (n == 0) return ELECTOR_OK;
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
