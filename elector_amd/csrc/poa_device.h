// elector_amd/csrc/poa_device.h -- shared device/host definitions of the POA engine.
//
// Data layout in HBM (all arrays are per batch, owned by the context workspace):
//
//   off[3n+1]     int64   byte offsets of (ref, cor, unc) per window in `bases`
//   sym[total]    u8      symbol indices (a2), same offsets as `bases`
//   "node space"  window w owns slots  nb = off[3w] + w  ..  nb + Lr+Lc+Lu  (inclusive)
//                 i.e. one more slot than it has bases, so that 1-based node
//                 indices jj = 1..|PO| and the virtual start node jj = 0 fit.
//       xinfo[nb+jj]  int2  DP view of node jj of the graph after fusion #1:
//                             .x = d1 | d2<<16     (DP predecessor list as distances back,
//                                                   node jj - d; d1 = 0: the virtual start,
//                                                   d2 = 0: no second predecessor)
//                             .y = letter | flags<<8
//       ring1[nb+j]   u16   ring id (column id) of node j, low 16 bits (only ever compared
//                           between neighbouring nodes)
//       map16[nb+j]   u32   x_to_y of the alignment being traced (0xFFFFFFFF = unaligned)
//       carry[nb+jj]  i32   packed (score, tag) of the last row of a DP strip
//   moves         u32     per window and alignment: [strip][t>>3][lane] dwords,
//                         8 steps x 4 bits per dword (see mv_index)
//   cols[3*off[3w] + 3c + r]  u8   the window's MSA, column-interleaved
//
// DP geometry: lanes are ROWS (the linear read y), time is the anti-diagonal:
// lane l of strip s owns row ii = 63 s + l (ii = 0 is the virtual row -1 of the
// reference's DP, align_lpo_po2.c:272-286); at step t it computes node
// jj = t - l.  Lane 0 does not compute: it replays the previous strip's last
// row (or the virtual row) so that lane 1 can read it like any other neighbour.
#pragma once
#include <atomic>
#include <stdint.h>

namespace elector {

constexpr int kStripRows = 63;
constexpr int kNeg = -999999;           // min_score, align_lpo_po2.c:198
constexpr uint32_t kNone16 = 0xFFFFu;
constexpr uint32_t kNone32 = 0xFFFFFFFFu;   // "not aligned" in the 32-bit maps of the generic path
constexpr int kTagBits = 6;             // packed cell = score << 6 | tag

// node flags (xinfo.y >> 8)
constexpr int kFlagFinal = 1, kFlagHasRef = 2, kFlagHasCor = 4, kFlagInitial = 8;

// move nibble = x-ordinal (0..2) | y-ordinal (0..1) << 2   (align_lpo_po2.c:12-16)
constexpr int kMoveX1 = 1, kMoveX2 = 2, kMoveY = 4;

struct DevTables {
  int32_t gpx[64];
  int32_t gpy[64];
  int32_t sub[32 * 32];
  uint8_t lut[256];     // raw byte -> symbol index (a2)
  uint8_t chr[32];      // symbol index -> output character (a10)
};

struct KParams {
  int M;                 // max_gap_length
  int open_x, ext_x, open_y, ext_y, match, mismatch;   // used when !GEN
};

// Largest |score| (plus the headroom the recurrences need around it) that a live cell of a DP over nx columns and ny
// rows can hold under simple scoring.  The 16-bit kernels take a window when this stays below 16000.
//
// The safe bound is maxpen x (nx + ny): no step of a path costs more than the largest penalty.  The tighter one --
// every cell is at least as good as the path made of one x-gap and one y-gap, 2 open + ext (nx + ny) -- is NOT a
// property of this recurrence in general: a cell keeps ONE gap state (align_lpo_po2.c:374-407) and a match wins only
// strictly (:384), so below a cell whose diagonal won by a hair the column pays a full opening again, and a column
// can fall by up to open - 1 per row instead of ext.  When open, ext and both substitution scores are multiples of
// ext (the shipped 0 / -10 / 10 / 5), "wins by a hair" is "wins by at least ext": whatever the diagonal gains on the
// gap path it loses again when the next gap opens, and the all-gap path bounds every live cell from below.  Only then
// is the tight bound used (it admits windows twice as long: the 700 x 700 un-anchored windows of trimmed / split reads
// stay in the 16-bit kernels); any other parameter set gets the safe bound.
__host__ __device__ inline int64_t score_span(const KParams &kp, int64_t nx, int64_t ny)
{
  auto ab = [](int64_t v) { return v < 0 ? -v : v; };
  const int64_t ext = ab(kp.ext_x) > ab(kp.ext_y) ? ab(kp.ext_x) : ab(kp.ext_y);
  const int64_t open = ab(kp.open_x) > ab(kp.open_y) ? ab(kp.open_x) : ab(kp.open_y);
  const int64_t sub = ab(kp.match) > ab(kp.mismatch) ? ab(kp.match) : ab(kp.mismatch);
  const int64_t pen = open > sub ? (open > ext ? open : ext) : (sub > ext ? sub : ext);
  const bool lattice = kp.ext_x > 0 && kp.ext_x == kp.ext_y && kp.open_x % kp.ext_x == 0 && kp.open_y % kp.ext_x == 0 &&
                       kp.match % kp.ext_x == 0 && kp.mismatch % kp.ext_x == 0;
  const int64_t down = lattice ? 2 * open + ext * (nx + ny) : pen * (nx + ny);
  const int64_t up = (kp.match > 0 ? kp.match : 0) * (nx < ny ? nx : ny);
  return (down > up ? down : up) + 3 * pen;
}

// dwords of move storage per strip for an alignment with Lx columns
__host__ __device__ inline int mv_tw(int Lx) { return (Lx + 71) >> 3; }
__host__ __device__ inline int n_strips(int Ly) { return (Ly + kStripRows - 1) / kStripRows; }
// dword index of the nibble of cell (ii, jj), ii >= 1 (1-based row), jj >= 1
__host__ __device__ inline int64_t mv_index(int tw, int ii, int jj, int *shift)
{
  int s = (ii - 1) / kStripRows, lane = (ii - 1) % kStripRows + 1, t = jj + lane;
  *shift = 4 * (t & 7);
  return ((int64_t)s * tw + (t >> 3)) * 64 + lane;
}

struct BatchArgs {
  int64_t n;                 // windows in the launch's list (all windows of the batch for k_left_b)
  const uint32_t *perm;      // the list: window ids in processing order
  const int32_t *count_ptr;  // when set: the list length lives on the device (list built by k_left_b)
  const int64_t *off;        // [3n+1]
  const uint8_t *bases;      // raw
  uint8_t *sym;
  int2 *xinfo;
  uint16_t *ring1;
  uint32_t *map16;           // x_to_y of the alignment being traced, 32 bits per node (kNone32 = unaligned)
  int32_t *carry;
  uint32_t *moves;
  const int64_t *mv1;        // dword offset of alignment #1 moves per window
  const int64_t *mv2;
  int32_t *n1;               // |PO| after fusion #1
  uint8_t *cls;              // ring class needed by alignment #2 (0,1,2; 3 = unsupported)
  int32_t *score1, *score2, *bx2;
  uint8_t *cols;
  int32_t *ncol;
  int32_t *status;
  const int32_t *linx, *liny;   // packed border cells: k gap steps from the origin
  const DevTables *tab;
  KParams kp;
  const uint8_t *skip_a;        // 1 = alignment #1 / fusion #1 already done by the fused kernel
  const uint8_t *skip_b;        // 1 = alignment #2 / fusion #2 already done by the fused kernel
  uint8_t *mark_b;              // when set, k_fuse2 marks the windows it finishes
  const uint8_t *tiled;         // per window: bit 0 = alignment #1, bit 1 = alignment #2 computed by the tiled kernels
};

// arguments of k_poa (poa_pack.hip): the whole window in one kernel
struct PackArgs {
  BatchArgs b;
  const uint32_t *list;     // window ids of this bin, in processing order; entries 2p, 2p+1 form pair p
  int64_t nlist;
  const uint4 *pdesc;       // k_gather's output for this bin: two uint4 per list entry (PackDesc)
  const uint32_t *psym;     // ... and the entry's three symbol strings back to back, pstride dwords per entry
  int pstride;
  int slot_bytes;           // LDS bytes per window slot
  uint8_t *done_a;
  uint8_t *done_b;
  const uint8_t *triv;      // per window: 0 = needs alignment #1, 1 = corrected == reference, 2 = one substitution
  uint32_t *mv_pool;        // moves scratch: [XCD][slot][mv_tw steps][64 lanes] words; a wave borrows a slot of its XCD
  int mv_tw;
  int32_t *mv_q;            // per XCD a queue of free slot ids: [head][tail][ids ...], kPoolStride ints apart
  int mv_slots;             // slots per XCD: at least the wavefronts an XCD can hold of any launch of the chain
  uint32_t *hand;           // windows handed back to the two-kernel path, appended at hand[atomicAdd(hand_count, 1)]
  int32_t *hand_count;
  uint32_t *far;            // windows whose graph has ONE edge from more than two nodes back and qualifies otherwise: the
  int32_t *far_count;       // group's k_poa<G, 8, true> launch takes them (far_cap entries; what does not fit goes to `hand`)
  int far_cap;
  const int32_t *nlist_dev; // when set: the list length lives on the device (a far list), nlist is its capacity
  int debug;
  unsigned long long *stamps;
  int keep_graph;           // a12: k_poa also leaves the graph after fusion #1 and the x -> y map of alignment #2 in HBM
                            // (BatchArgs::xinfo .y / ring1 / map16, as the two-kernel path does: what k_bundle_inputs reads)
};

// arguments of k_gather (poa_pack.hip): k_poa's inputs laid out in list order
struct GatherArgs {
  const uint32_t *list;
  int64_t nlist;
  const int64_t *off;
  const uint8_t *sym;
  const int32_t *status;
  const uint8_t *done_a, *done_b, *triv;
  uint4 *pdesc;
  uint32_t *psym;
  int pstride;               // dwords per entry
  const int32_t *nlist_dev;  // when set: the list length lives on the device (a far list), nlist is its capacity
};

// Long windows (the reference's whole-read fallback) are cut into tiles of one strip of 63 rows by
// kTileCols columns; the tiles of one anti-diagonal (strip + column block) are independent and run
// in one launch (k_dp1_tile / k_dp2_tile), so a window of many strips uses many wavefronts.
constexpr int kTileCols = 2048;          // multiple of 64
constexpr int kTileRing = 32;            // ring depth of the tiled alignment #2 (ring class 0 graphs)
// per strip and column-block parity: 64 boundary cells (+ for alignment #2: the ring, best score / column)
constexpr int kTileState1 = 64, kTileState2 = 64 + kTileRing * 64 + 64;

struct TileArgs {
  const uint32_t *wlist;        // the long windows of this pass
  const int64_t *st_off;        // per long window: its state area in tstate (ints)
  int32_t *tstate;
  uint8_t *tiled;
  int d;                        // anti-diagonal of this launch
};

// LDS bytes the fused kernels need for a window worked on by a group of G lanes with R rows
// per lane (a strip has G*R rows); must mirror the slot layouts in poa_fused.hip.
// moves: 2 bits per cell, one byte (R <= 4) or two (R <= 8) per lane and step.  They live in HBM
// (written once, coalesced: one word per lane and anti-diagonal step; read back by the traceback
// while still in L2), indexed [block][strip][step][lane].
__host__ __device__ inline int fused_mv_bytes(int R) { return R <= 4 ? 1 : 2; }

// Alignment #1 may keep its moves in LDS for the classes up to this many lanes per window.  Measured
// on the bench batch: 0 (all moves in HBM) wins -- the traceback of alignment #1 gets slower (its many
// resident waves' moves fall out of L2), but the LDS it frees goes to the concurrently running
// alignment #2 kernels, whose occupancy is LDS-bound.
#ifndef ELECTOR_A_LDS_MAXG
#define ELECTOR_A_LDS_MAXG 0
#endif
__host__ __device__ constexpr bool fused_a_moves_in_lds(int G) { return G <= ELECTOR_A_LDS_MAXG; }

// bytes of k_fused_a's node maps (u16 per ref letter, 2 x u16 per cor letter)
__host__ __device__ inline int fused_a_maps_bytes(int Lr, int Lc) { return 2 * ((Lr + 1) & ~1) + 4 * ((Lc + 1) & ~1) + 8; }

__host__ __device__ inline int fused_a_slot_need(int Lr, int Lc, int G, int R)
{
  const int ns = (Lc + R * G - 1) / (R * G);
  int o = 16 + ((Lr + Lc + 3) & ~3);
  o += (2 * Lr + 3) & ~3;
  o += (ns > 1 ? 4 * (Lr + 1) : 0);
  o = (o + 7) & ~7;
  const int mv = fused_a_moves_in_lds(G) ? ns * Lr * G * fused_mv_bytes(R) : 0, st = fused_a_maps_bytes(Lr, Lc);
  return o + (mv > st ? mv : st);
}

// bytes of k_fused_b's staged MSA columns (3 per column) + the column of every uncorrected letter
__host__ __device__ inline int fused_b_cols_bytes(int n1, int Lu) { return ((3 * (n1 + Lu) + 8 + 3) & ~3) + 2 * Lu + 4; }

__host__ __device__ inline int fused_b_slot_need(int n1, int Lu, int G, int R)
{
  const int ns = (Lu + R * G - 1) / (R * G);
  int o = 16 + ((Lu + 3) & ~3);
  o += 4 * (n1 + 1);
  o += (2 * n1 + 3) & ~3;
  o += (2 * n1 + 3) & ~3;
  o += (2 * (n1 + 1) + 3) & ~3;
  o += (ns > 1 ? 2 * (n1 + 1) : 0);
  o = (o + 3) & ~3;
  // ordinal bytes of the two-predecessor nodes (estimated: one node in six), overlaid by the staged columns
  const int ob = (4 + n1 / 6) * ns * G, st = fused_b_cols_bytes(n1, Lu);
  return o + (ob > st ? ob : st);
}

// bytes of one wave's score ring in k_fused_b: D time slots x 64 lanes x R 16-bit cells
__host__ __device__ inline int fused_ring_bytes(int R, int D) { return D * 64 * 4 * ((R + 1) / 2); }

// free-slot queue of k_poa's moves scratch, per XCD: [head, pad to 16 ints][tail, pad][slot ids]
constexpr int kPoolSlotsMax = 32 * 32 + 8;
constexpr int kPoolStride = 32 + ((kPoolSlotsMax + 31) & ~31);

// LDS slot of one window in k_poa (poa_pack.hip), see the layout there
__host__ __device__ inline int poa_xi_cap(int Lr, int Lc)
{
  // nodes after fusion #1: Lr + Lc minus the fused pairs; a well-corrected read fuses nearly all of min(Lr, Lc)
  const int lo = Lr < Lc ? Lr : Lc, hi = Lr < Lc ? Lc : Lr;
  return hi + lo / 8 + 8;
}

// bytes per entry of the alignment #1 / fusion #1 index maps of a geometry class
__host__ __device__ constexpr int poa_idx_bytes(int G) { return G == 8 ? 1 : 2; }

__host__ __device__ inline int poa_union_a(int Lr, int Lc, int G)
{
  return ((Lr + Lc + 3) & ~3) + ((poa_idx_bytes(G) * (2 * ((Lr + 1) & ~1) + 2 * ((Lc + 1) & ~1)) + 3) & ~3);
}

__host__ __device__ inline int poa_union_b(int n1, int Lu, int G)
{
  const int x2y = (2 * n1 + 3) & ~3;
  const int cols = 2 * Lu + 4;
  const int ord = (8 + n1 / 8) * G;
  return x2y + (cols > ord ? cols : ord);
}

__host__ __device__ inline int poa_slot_need(int Lr, int Lc, int Lu, int G)
{
  const int cap = poa_xi_cap(Lr, Lc);
  const int ua = poa_union_a(Lr, Lc, G), ub = poa_union_b(cap, Lu, G);
  return 16 + ((Lu + 3) & ~3) + 4 * (cap + 2) + (ua > ub ? ua : ub);
}

// ... of a window that skips alignment #1 (corrected sequence within one edit of its reference: trivial_graph builds the
// graph, at most Lr + 1 nodes): no index maps in union A, records for Lr + 2 nodes
__host__ __device__ inline int poa_xi_cap_triv(int Lr) { return Lr + 2; }

__host__ __device__ inline int poa_slot_need_triv(int Lr, int Lc, int Lu, int G)
{
  const int cap = poa_xi_cap_triv(Lr);
  const int ua = (Lr + Lc + 3) & ~3, ub = poa_union_b(cap, Lu, G);
  return 16 + ((Lu + 3) & ~3) + 4 * (cap + 2) + (ua > ub ? ua : ub);
}

// "has this kernel's dynamic-LDS limit been raised on the CURRENT device yet?" -- hipFuncSetAttribute is per device, a
// process may hold contexts on several (one bit per device id; setting the attribute twice from two threads is harmless)
struct DeviceOnce {
  std::atomic<unsigned long long> mask{0};
  static unsigned long long bit()
  {
    int dev = 0;                                       // (a local: several threads ask at once)
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return 1ull << (dev & 63);
  }
  bool need() { return (mask.load(std::memory_order_acquire) & bit()) == 0; }
  void done() { mask.fetch_or(bit(), std::memory_order_release); }
};

}  // namespace elector
