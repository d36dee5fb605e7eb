// elector_amd/csrc/poa_fused.hip -- LDS-resident, row-blocked kernels for the
// windows that fit on chip (the bulk: the splitter's windows are 27-400 bases).
//
// Geometry: one wavefront works on 64/G windows; each window owns G lanes
// (G = 8, 16, 32 or 64) and a slot of LDS.  A lane holds R consecutive rows of the
// linear read y in registers (R = 4..8, chosen with G so that one strip of G*R
// rows is just as tall as the window), so a strip costs Lx + G - 1 anti-diagonal
// steps: lane g computes column jj = t - g for its R rows top to bottom (the
// vertical dependence stays inside the lane).  The row above a lane's block
// arrives by DPP row_shr:1 (wave_shr:1 for G > 16); the group's first lane
// receives the strip's top border.
//
//   k_fused_a  alignment #1 (linear x linear, registers only) -> traceback ->
//              fusion #1 -> PO graph (xinfo, ring1) to HBM.  Replaces k_dp1 + k_fuse1.
//   k_fused_b  alignment #2 (PO x linear, time ring in LDS; ring-free for wavefronts
//              of chain graphs) -> traceback -> fusion #2 -> MSA columns.
//              Replaces k_dp2 + k_fuse2.
// The moves (2 bits per cell) stream to an HBM scratch, coalesced, one word per lane
// and step, and are read back by the same wavefront's traceback, G cells per round.
//
// Uniform-scoring parameters only (the shipped matrix); other parameter sets and
// windows that do not fit their LDS slot stay on the generic kernels of
// poa_kernels.hip (the `done` flags say which windows were handled here).
// Reference behaviour restated: align_lpo_po2.c:178-433 (DP, tie-breaks),
// :108-168 (traceback), lpo.c:413-463,602-656 (fusion), lpo_format.c:337-393 (rows).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "poa_device.h"

namespace elector {

struct FusedArgs {
  BatchArgs b;
  const uint32_t *list;     // window ids of this bin, in processing order
  int64_t nlist;
  int slot_bytes;           // LDS bytes per window slot
  uint8_t *done_a;          // 1 = alignment #1 + fusion #1 done by k_fused_a
  uint8_t *done_b;          // 1 = alignment #2 + fusion #2 done by k_fused_b
  int32_t *rowinit;         // node space: score of the virtual row -1 at node jj
  int debug;                // timing experiments only (bit0: skip DP, bit1: skip serial stage)
  int keep_map;             // k_fused_b also leaves the x -> y map of alignment #2 in HBM (a12 needs it)
  uint8_t *mv_pool;         // moves scratch of this launch: [block][mv_ns strips][mv_tw steps][64 lanes] words
  int mv_tw, mv_ns;
  const int32_t *nlist_dev; // k_fused_a: when set, only the first *nlist_dev windows of the list still need it
  const uint8_t *triv;      // k_fused_b: when set, triv[w] = the graph of window w is a plain chain (k_trivial)
  int64_t grid_blocks;      // > 0: launch at most this many blocks (they loop over the list)
};

__device__ __forceinline__ int row_shr1(int old, int v)
{
  // DPP row_shr:1 within each row of 16 lanes; lane 0 of a row keeps `old`.
  return __builtin_amdgcn_update_dpp(old, v, 0x111, 0xF, 0xF, false);
}

__device__ __forceinline__ int wave_shr1_old(int old, int v)
{
  // DPP wave_shr:1 over the whole wave; lane 0 keeps `old`.
  return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xF, 0xF, false);
}

// value of the lane above inside a group of G lanes; the group's first lane gets `border`
template <int G>
__device__ __forceinline__ int shift_in(int border, int v, int g)
{
  if (G == 16) return row_shr1(border, v);
  if (G == 8) { const int r = row_shr1(border, v); return g == 0 ? border : r; }
  const int r = wave_shr1_old(border, v);
  return (G < 64 && g == 0) ? border : r;
}

__device__ __forceinline__ int align_up(int x, int a) { return (x + a - 1) & ~(a - 1); }

// number of leading lanes of this lane's group of G (group index q) whose flag is set
template <int G>
__device__ __forceinline__ int diag_run(bool flag, int q)
{
  const unsigned long long bm = __builtin_amdgcn_ballot_w64(flag);
  if (G == 64) return ~bm == 0ull ? 64 : __builtin_ctzll(~bm);
  const unsigned long long gm = (bm >> (q * G)) & ((1ull << (G & 63)) - 1);
  return __builtin_ctzll(~gm | (1ull << (G & 63)));
}

// phase stamps for the timing-experiment build path (debug bit 2): cycles per phase
// summed over waves into a scratch array that nothing else reads
#define PHASE_STAMP(idx)                                                                          \
  do {                                                                                            \
    if ((a.debug & 4) && threadIdx.x == 0) {                                                      \
      const unsigned long long now_ = __builtin_readcyclecounter();                               \
      atomicAdd(reinterpret_cast<unsigned long long *>(a.rowinit) + (idx), now_ - stamp_);        \
      stamp_ = now_;                                                                              \
    }                                                                                             \
  } while (0)

// a lane's move pairs of one column: 2 bits per cell, R cells
template <int R> struct MvWord { using type = uint16_t; };
template <> struct MvWord<1> { using type = uint8_t; };
template <> struct MvWord<2> { using type = uint8_t; };
template <> struct MvWord<3> { using type = uint8_t; };
template <> struct MvWord<4> { using type = uint8_t; };

// one DP cell with uniform scoring: returns new score, sets match flag / move nibble
struct CellOut { int S; bool m; uint32_t nib; };

__device__ __forceinline__ CellOut cell_1pred(int diag, int insX, int insY, int subv)
{
  const int mat = diag + subv;
  const int mx = max(insX, insY);
  CellOut o;
  o.m = mat > mx;                                   // align_lpo_po2.c:384 (strict on both)
  o.S = max(mat, mx);
  o.nib = o.m ? (kMoveX1 | kMoveY) : (insX > insY ? kMoveX1 : kMoveY);   // :392 ties -> y
  return o;
}

// ---------------------------------------------------------------- k_fused_a ---

// one window of k_fused_a: ids, lengths and its LDS slot layout
// slot: [header 16 B, unused][ref+cor symbols][x2y u16][carry i32 (multi-strip only)][region]
// region = the node maps of the fusion (node of every ref / cor letter, cor -> ref map, u16 each).
// The moves (R cells x 2 bits per lane and step, in 1 or 2 bytes) go to the launch's HBM scratch.
struct WinA {
  bool valid;
  uint32_t w;
  int64_t o0;
  int Lr, Lc, ns, off_x2y, off_carry, off_region;
};

template <int G, int R>
__device__ __forceinline__ WinA load_win_a(const FusedArgs &a, int64_t li)
{
  constexpr int RS = R * G;
  WinA v;
  v.valid = li < (a.nlist_dev ? (int64_t)*a.nlist_dev : a.nlist);
  v.w = v.valid ? a.list[li] : 0;
  v.valid = v.valid && a.b.status[v.w] == 0 && a.done_a[v.w] == 0;
  v.o0 = 0; v.Lr = 0; v.Lc = 0;
  if (v.valid) {
    v.o0 = a.b.off[3 * (int64_t)v.w];
    v.Lr = (int)(a.b.off[3 * (int64_t)v.w + 1] - v.o0);
    v.Lc = (int)(a.b.off[3 * (int64_t)v.w + 2] - v.o0) - v.Lr;
  }
  v.ns = (v.Lc + RS - 1) / RS;
  v.off_x2y = 16 + align_up(v.Lr + v.Lc, 4);
  v.off_carry = v.off_x2y + align_up(2 * v.Lr, 4);
  v.off_region = align_up(v.off_carry + (v.ns > 1 ? 4 * (v.Lr + 1) : 0), 8);
  constexpr int MVB = R <= 4 ? 1 : 2;
  const int region_bytes = max(fused_a_moves_in_lds(G) ? v.ns * v.Lr * G * MVB : 0, fused_a_maps_bytes(v.Lr, v.Lc));
  v.valid = v.valid && (v.off_region + region_bytes <= a.slot_bytes) &&
            (fused_a_moves_in_lds(G) || (v.ns <= a.mv_ns && v.Lr + G <= a.mv_tw));
  return v;
}

// WV wavefronts per block, each owning 64/G windows from staging to the fused graph.
template <int G, int R, int WV>
__device__ __forceinline__ void fused_a_body(const FusedArgs &a, uint8_t *lds, const int64_t vblk)
{
  constexpr int NW = 64 / G, RS = R * G;     // windows per wave, rows per strip
  using mv_t = typename MvWord<R>::type;     // a lane's R move pairs of one column
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane / G, g = lane & (G - 1);
  const KParams kp = a.b.kp;
  const int sidx = wv * NW + q;               // this lane's window slot in the block
  if (a.nlist_dev && (int64_t)(NW * WV) * vblk >= (int64_t)*a.nlist_dev) return;   // the rest of the list is done already
  const WinA W = load_win_a<G, R>(a, (int64_t)(NW * WV) * vblk + sidx);
  const bool valid = W.valid;
  const int Lr = W.Lr, Lc = W.Lc, ns = W.ns;
  uint8_t *slot = lds + sidx * a.slot_bytes;
  uint8_t *xs = slot + 16, *ys = xs + Lr;
  int32_t *carry = reinterpret_cast<int32_t *>(slot + W.off_carry);
  // moves: LDS (region, [strip][column][lane of the group]) for the small classes, else the launch's
  // HBM scratch ([strip][step][lane of the wave])
  constexpr bool kMvLds = fused_a_moves_in_lds(G);
  mv_t *mvl = reinterpret_cast<mv_t *>(slot + W.off_region);
  mv_t *mvg = reinterpret_cast<mv_t *>(a.mv_pool) + ((int64_t)blockIdx.x * WV + wv) * a.mv_ns * a.mv_tw * 64;
  const int mvtw = a.mv_tw;

  unsigned long long stamp_ = (a.debug & 4) ? __builtin_readcyclecounter() : 0;
  if (valid) {
    const uint8_t *src = a.b.sym + W.o0;
    for (int i = g; i < Lr + Lc; i += G) xs[i] = src[i];
  }
  __builtin_amdgcn_wave_barrier();
  PHASE_STAMP(0);

  // wave-uniform loop bounds
  int tmax = valid ? Lr + G - 1 : 0, nsmax = valid ? ns : 0;
  for (int d = G; d < 64; d <<= 1) {
    tmax = max(tmax, __shfl_xor(tmax, d));
    nsmax = max(nsmax, __shfl_xor(nsmax, d));
  }
  tmax = __builtin_amdgcn_readfirstlane(tmax);
  nsmax = __builtin_amdgcn_readfirstlane(nsmax);

  int score = kNeg;
  const int gstar = ((Lc - 1) % RS) / R, kstar = ((Lc - 1) % RS) % R;
  if (a.debug & 1) nsmax = 0;
  for (int s = 0; s < nsmax; ++s) {
    const bool sv = valid && s < ns;
    const int ii0 = RS * s + R * g + 1;                    // first of this lane's R rows (1-based)
    int yl[R], S[R], Ex[R], Ey[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int ii = ii0 + k;
      yl[k] = (sv && ii <= Lc) ? ys[ii - 1] : 255;
      S[k] = -(kp.open_y + (ii - 1) * kp.ext_y);           // column -1: ii gap steps from the origin
      Ex[k] = S[k] - kp.ext_x;
      Ey[k] = S[k] - kp.ext_y;
    }
    int dg0 = (ii0 == 1) ? 0 : -(kp.open_y + (ii0 - 2) * kp.ext_y);   // cell (row above, column -1)
    const bool wr_carry = sv && (s + 1 < ns) && g == G - 1;
    int xl_next = (sv && g == 0 && Lr >= 1) ? xs[0] : 0;    // letter of column jj = t - g, fetched one step ahead
    for (int t = 1; t <= tmax; ++t) {
      // strip top border at column t (for lane 0 of the group), through the DPP `old` operand
      int bS, bEy;
      if (s == 0) { bS = -(kp.open_x + (t - 1) * kp.ext_x); bEy = bS - kp.ext_y; }
      else {
        const int c = (sv && t <= Lr) ? carry[t] : 0;
        bS = c >> 1; bEy = bS - ((c & 1) ? kp.open_y : kp.ext_y);
      }
      const int upS = shift_in<G>(bS, S[R - 1], g);
      const int upEy = shift_in<G>(bEy, Ey[R - 1], g);
      const int jj = t - g;
      const int xl = xl_next;
      xl_next = (sv && jj >= 0 && jj < Lr) ? xs[jj] : 0;
      if (sv && jj >= 1 && jj <= Lr) {
        int diag = dg0, insY = upEy;
        uint32_t mv8 = 0;
        bool mlast = false;
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const int oldS = S[k];
          const CellOut c = cell_1pred(diag, Ex[k], insY, xl == yl[k] ? kp.match : kp.mismatch);
          S[k] = c.S;
          Ex[k] = c.S - (c.m ? kp.open_x : kp.ext_x);
          Ey[k] = c.S - (c.m ? kp.open_y : kp.ext_y);
          mv8 |= ((c.nib & 1) | (c.nib >> 1)) << (2 * k);    // 2 bits: bit0 = x step, bit1 = y step
          diag = oldS; insY = Ey[k]; mlast = c.m;
        }
        dg0 = upS;
        if (kMvLds) mvl[(s * Lr + (jj - 1)) * G + g] = (mv_t)mv8;
        else mvg[(s * mvtw + t) * 64 + lane] = (mv_t)mv8;
        if (wr_carry) carry[jj] = (S[R - 1] << 1) | (mlast ? 1 : 0);
      }
    }
    if (sv && s == ns - 1 && g == gstar) {
#pragma unroll
      for (int k = 0; k < R; ++k) if (k == kstar) score = S[k];
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (valid && g == gstar) a.b.score1[W.w] = score;
  uint16_t *x2y = reinterpret_cast<uint16_t *>(slot + W.off_x2y);
  if (valid) for (int i = g; i < Lr; i += G) x2y[i] = (uint16_t)kNone16;
  __syncthreads();

  PHASE_STAMP(1);
  bool bad = false;
  // ---- traceback #1 (align_lpo_po2.c:108-168), G cells per round: lane g fetches the move of cell
  // (x - g, y - g); as long as those are matches the walk stays on this diagonal and the lanes record
  // their aligned pairs together; the first cell off the diagonal decides where the next round starts.
  // The moves come from HBM: one memory latency per round instead of one per cell. ----
  {
    int x = Lr - 1, y = Lc - 1, guard = Lr + Lc + 2;
    bool alive = valid && !(a.debug & 2);
    while (__builtin_amdgcn_ballot_w64(alive) != 0) {
      const int cx = x - g, cy = y - g;
      const bool inb = alive && cx >= 0 && cy >= 0;
      uint32_t two = 0;
      if (inb) {
        const int r = cy % RS, rl = r / R, rk = r - rl * R;
        const uint32_t word = kMvLds ? (uint32_t)mvl[((cy / RS) * Lr + cx) * G + rl]
                                     : (uint32_t)mvg[((cy / RS) * mvtw + (cx + 1 + rl)) * 64 + q * G + rl];
        two = (word >> (2 * rk)) & 3u;
      }
      const int xo = two & 1, yo = two >> 1;
      const int run = diag_run<G>(inb && xo && yo, q);
      if (g < run) x2y[cx] = (uint16_t)cy;
      // what this lane's cell would do to the walk if it is the first one off the diagonal
      const bool stop = !inb || (!xo && !yo);
      int nx = cx - xo, ny = cy - yo, fl = (stop ? 1 : 0) | ((inb && !xo && !yo) ? 2 : 0);
      const int src = min(run, G - 1);
      nx = __shfl(nx, src, G); ny = __shfl(ny, src, G); fl = __shfl(fl, src, G);
      if (run >= G) { nx = x - G; ny = y - G; fl = 0; }
      if (alive) {
        x = nx; y = ny;
        if (fl & 2) bad = true;
        if ((fl & 1) || --guard <= 0) alive = false;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  PHASE_STAMP(5);
  // ---- fusion #1 (lpo.c:602-668 fuse_lpo_remap on two linear sequences), spread over the window's
  // G lanes.  Node order of the fused graph: walking the reference, every pending corrected letter up
  // to the aligned one comes first (an aligned but different letter directly before its reference
  // letter, sharing its ring), identical aligned letters are one node.  So
  //   node(ref i) = i + (corrected letters consumed up to i) - (fused pairs up to i)
  //   node(cor j) = j - (fused pairs before j) + (reference letters before the next aligned pair)
  // and the predecessors are the nodes of ref i-1 / cor j-1.  Scans over the letters give the terms;
  // every lane then writes its nodes straight to the graph arrays in HBM. ----
  int n1 = 0, maxd = 1;
  {
    uint16_t *node_ref = reinterpret_cast<uint16_t *>(slot + W.off_region);
    uint16_t *node_cor = node_ref + ((Lr + 1) & ~1);
    uint16_t *y2x = node_cor + ((Lc + 1) & ~1);
    const int cx = (Lr + G - 1) / G, cy = (Lc + G - 1) / G;               // letters per lane
    int cxmax = valid ? cx : 0, cymax = valid ? cy : 0;
    for (int d = G; d < 64; d <<= 1) {
      cxmax = max(cxmax, __shfl_xor(cxmax, d));
      cymax = max(cymax, __shfl_xor(cymax, d));
    }
    cxmax = __builtin_amdgcn_readfirstlane(cxmax);
    cymax = __builtin_amdgcn_readfirstlane(cymax);
    if (a.debug & (2 | 16)) { cxmax = 0; cymax = 0; }
    if (valid) for (int i = g; i < Lc; i += G) y2x[i] = (uint16_t)kNone16;
    __builtin_amdgcn_wave_barrier();
    const int x0 = g * cx, x1 = valid ? min(Lr, x0 + cx) : 0;
    const int y0 = g * cy, y1 = valid ? min(Lc, y0 + cy) : 0;
    // reference letters: corrected letters consumed (P) and fused pairs (F) up to each of them
    int pmax = 0, fcnt = 0;
    for (int it = 0; it < cxmax; ++it) {
      const int ix = x0 + it;
      if (ix < x1) {
        const int ay = x2y[ix];
        if (ay != (int)kNone16) {
          if (ay < Lc) y2x[ay] = (uint16_t)ix; else bad = true;
          if (ay + 1 <= pmax) bad = true;                                // a path is monotone
          pmax = ay + 1;
          fcnt += (ay < Lc && xs[ix] == ys[ay]);
        }
      }
    }
    int sp = pmax, sf = fcnt;
    for (int d = 1; d < G; d <<= 1) {
      const int tp = __shfl_up(sp, d, G), tf = __shfl_up(sf, d, G);
      if (g >= d) { sp = max(sp, tp); sf += tf; }
    }
    const int fused_all = __shfl(sf, G - 1, G);
    int P = __shfl_up(sp, 1, G), F = sf - fcnt;
    if (g == 0) P = 0;
    if (pmax > 0 && x1 > x0) {
      // first aligned letter of this lane's chunk must lie beyond everything before it
      int first = 0;
      for (int ix = x0; ix < x1; ++ix) { const int ay = x2y[ix]; if (ay != (int)kNone16) { first = ay + 1; break; } }
      if (first <= P) bad = true;
    }
    for (int it = 0; it < cxmax; ++it) {
      const int ix = x0 + it;
      if (ix < x1) {
        const int ay = x2y[ix];
        if (ay != (int)kNone16) { P = ay + 1; F += (ay < Lc && xs[ix] == ys[ay]); }
        node_ref[ix] = (uint16_t)(ix + P - F);
      }
    }
    __builtin_amdgcn_wave_barrier();
    // corrected letters: fused pairs before each of them, reference letters before the next aligned pair
    constexpr int kUndef = -1;
    int fy = 0, klow = kUndef;
    for (int it = 0; it < cymax; ++it) {
      const int y = y1 - 1 - it;
      if (y >= y0) {
        const int x = y2x[y];
        if (x != (int)kNone16) { klow = x; fy += xs[x] == ys[y]; }
      }
    }
    int sfy = fy, sfx = klow;
    for (int d = 1; d < G; d <<= 1) {
      const int tf = __shfl_up(sfy, d, G), tk = __shfl_down(sfx, d, G);
      if (g >= d) sfy += tf;
      if (g + d < G && sfx == kUndef) sfx = tk;
    }
    int K = __shfl_down(sfx, 1, G), fy_run = sfy;                          // fused pairs below the end of this chunk
    if (g == G - 1 || K == kUndef) K = Lr;
    for (int it = 0; it < cymax; ++it) {
      const int y = y1 - 1 - it;
      if (y >= y0) {
        const int x = y2x[y];
        const bool al = x != (int)kNone16, fu = al && xs[x] == ys[y];
        if (fu) --fy_run;
        if (al) K = x;
        node_cor[y] = fu ? node_ref[x] : (uint16_t)(y - fy_run + K);
      }
    }
    __builtin_amdgcn_wave_barrier();
    n1 = Lr + Lc - fused_all;
    // emit: predecessor list, letter, flags and ring of every node (lpo.c:413-463 reindexing, :227-241 links)
    const int64_t nb = W.o0 + W.w;
    int2 *gx = a.b.xinfo + nb;
    uint16_t *gr = a.b.ring1 + nb;
    auto emit = [&](int n, int letter, int flags, int ring, int sa, int sb) {
      int d1, d2 = 0;                                      // distances back; d1 0 = virtual start, d2 0 = none
      const int jj = n + 1;
      if (sa < 0) d1 = 0;
      else if (flags & kFlagInitial) { d1 = 0; d2 = n - sa; if (sb >= 0) bad = true; }
      else { d1 = n - sa; if (sb >= 0) d2 = n - sb; }
      maxd = max(maxd, max(d1, d2));
      if (n >= 0 && n < Lr + Lc) {
        gx[jj] = make_int2(d1 | (d2 << 16), letter | (flags << 8));
        gr[n] = (uint16_t)ring;
      } else bad = true;
    };
    for (int it = 0; it < cxmax; ++it) {
      const int ix = x0 + it;
      if (ix < x1) {
        const int ay = x2y[ix], n = node_ref[ix];
        const bool al = ay != (int)kNone16 && ay < Lc, fu = al && xs[ix] == ys[ay];
        int fl = kFlagHasRef | (ix == 0 ? kFlagInitial : 0) | (ix == Lr - 1 ? kFlagFinal : 0);
        int sa = ix > 0 ? (int)node_ref[ix - 1] : -1, sb = -1, ring = n;
        if (fu) {
          fl |= kFlagHasCor | (ay == 0 ? kFlagInitial : 0) | (ay == Lc - 1 ? kFlagFinal : 0);
          const int lasty = ay > 0 ? (int)node_cor[ay - 1] : -1;
          if (lasty >= 0 && lasty != sa) { if (sa < 0) sa = lasty; else sb = lasty; }
        } else if (al) ring = n - 1;                                       // joins the ring of its corrected partner
        emit(n, xs[ix], fl, ring, sa, sb);
      }
    }
    for (int it = 0; it < cymax; ++it) {
      const int y = y0 + it;
      if (y < y1) {
        const int x = y2x[y];
        if (!(x != (int)kNone16 && xs[x] == ys[y])) {
          const int n = node_cor[y];
          emit(n, ys[y], kFlagHasCor | (y == 0 ? kFlagInitial : 0) | (y == Lc - 1 ? kFlagFinal : 0), n,
               y > 0 ? (int)node_cor[y - 1] : -1, -1);
        }
      }
    }
    for (int d = 1; d < G; d <<= 1) {
      maxd = max(maxd, __shfl_xor(maxd, d, G));
      bad = bad || __shfl_xor(bad ? 1 : 0, d, G) != 0;
    }
  }
  PHASE_STAMP(2);
  if (valid && g == 0) {
    a.b.n1[W.w] = n1;
    const int need = maxd + 2;
    if (a.debug & 32) atomicAdd(reinterpret_cast<unsigned long long *>(a.rowinit) + 16 + min(need, 15), 1ull);   // histogram (diagnostics)
    // bits 0-1: ring class of the generic k_dp2 (0: LDS ring, 1: with the HBM shadow); bit 6 / bit 7: too deep
    // for k_fused_b's 4- / 8-deep ring
    a.b.cls[W.w] = (uint8_t)((need <= 32 ? 0 : 1) | (need > 4 ? 0x40 : 0) | (need > 8 ? 0x80 : 0));
    if (bad) a.b.status[W.w] = 3;
    a.done_a[W.w] = 1;
  }
  PHASE_STAMP(3);
  if ((a.debug & 4) && threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long *>(a.rowinit) + 4, 1ull);
}

// The kernel: a block takes the list's chunks of NW * WV windows in a grid-stride loop (the launches behind
// k_poa work on short device-built lists with a grid much smaller than the list's capacity).
template <int G, int R, int WV>
__global__ void __launch_bounds__(64 * WV) k_fused_a(FusedArgs a)
{
  extern __shared__ __align__(16) uint8_t lds[];
  constexpr int NW = 64 / G;
  const int64_t cnt = a.nlist_dev ? (int64_t)*a.nlist_dev : a.nlist;
  for (int64_t vblk = blockIdx.x; vblk * (NW * WV) < cnt; vblk += gridDim.x) {
    fused_a_body<G, R, WV>(a, lds, vblk);
    __syncthreads();
  }
}

// ---------------------------------------------------------------- k_fused_b ---
// Alignment #2 on the PO graph of (ref + cor).  A node has <= 2 DP predecessors at
// any distance; the common case (one predecessor, the previous node) is served from
// registers, everything else from an 8-deep LDS ring indexed by time that holds each
// lane's R cells of the last 8 steps as 16-bit (score << 1 | came-from-match), laid out
// [slot][dword][lane] so that every access is bank-conflict free.

// The ring depth D (time slots, a power of two) is a template parameter: a node's predecessors must
// lie at most D - 2 nodes back.  Only D = 8 is instantiated: on well-corrected reads 99.8 % of the
// windows would do with D = 4, but the smaller ring did not pay (see poa_host.hip).
constexpr int kNeg16 = -16383;       // score of the "no predecessor" cells (below any 16-bit-eligible score)

__device__ __forceinline__ int cell16_S(int c) { return c >> 1; }

// the column layout rule in its plain serial form (one lane): the fallback of k_fused_b's parallel version
__device__ __noinline__ int fuse2_columns_serial(int n1, int Lu, const uint32_t *xinfo, const uint16_t *ring1,
                                                 const uint16_t *x2y, const uint8_t *ys, const uint8_t *chr,
                                                 uint8_t *cols_st)
{
  int col = 0, prev_ring = 0;
  uint8_t c0 = '.', c1 = '.', c2 = '.';
  auto flush = [&]() { cols_st[3 * col] = c0; cols_st[3 * col + 1] = c1; cols_st[3 * col + 2] = c2; };
  auto place = [&](int ring_id, int letter, bool r, bool c, bool u) {
    if (ring_id != prev_ring) { flush(); ++col; c0 = c1 = c2 = '.'; prev_ring = ring_id; }
    const uint8_t ch = chr[letter & 31];
    if (r) c0 = ch;
    if (c) c1 = ch;
    if (u) c2 = ch;
  };
  int n = 0, iy = 0, blk_old = -1, blk_new = -1;
  for (int ix = 0; ix < n1; ++ix) {
    const int r0 = ring1[ix];
    if (r0 != blk_old) { blk_old = r0; blk_new = -1; }
    for (int k = ix; k < n1 && ring1[k] == r0; ++k) {
      const int ay = x2y[k];
      if (ay != (int)kNone16) {
        while (iy < ay) { place(n, ys[iy], false, false, true); ++n; ++iy; }
        break;
      }
    }
    const uint32_t xv = xinfo[ix + 1];
    const int letter = (xv >> 8) & 0xFF, fl = (int)(xv >> 16);
    bool fused = false;
    if (x2y[ix] != (uint16_t)kNone16 && iy < Lu) {
      if (letter == ys[iy]) fused = true;
      else {
        if (blk_new < 0) blk_new = n;
        place(blk_new, ys[iy], false, false, true);
        ++n;
      }
      ++iy;
    }
    if (blk_new < 0) blk_new = n;
    place(blk_new, letter, (fl & kFlagHasRef) != 0, (fl & kFlagHasCor) != 0, fused);
    ++n;
  }
  while (iy < Lu) { place(n, ys[iy], false, false, true); ++n; ++iy; }
  flush();
  return col + 1;
}

// one window of k_fused_b and its LDS slot layout:
// [header 16 B: k2 / ncol, ok, best, bestx][unc symbols][node info u32[n1+1]][ring1 u16[n1]][x2y u16[n1]]
// [bnd0 i16[n1+1]][bnd1 i16[n1+1] if ns>1][region]
// node info = d1 (0 = virtual start) | d2 << 4 (0 = none, 15 = virtual) | letter << 8 | flags << 16 | k2 << 24
// region = predecessor-ordinal bytes of the K2 nodes that have two predecessors; reused after the
// traceback for the staged MSA columns.  The moves (R cells x 2 bits per lane and step, in 1 or 2
// bytes) go to the launch's HBM scratch.
struct WinB {
  bool valid;
  uint32_t w;
  int64_t o0, o2;
  int n1, Lu, ns, off_xi, off_r1, off_x2y, off_b0, off_b1, off_region;
};

template <int G, int R, int D>
__device__ __forceinline__ WinB load_win_b(const FusedArgs &a, int64_t li)
{
  constexpr int RS = R * G;
  const KParams kp = a.b.kp;
  WinB v;
  v.valid = li < (a.nlist_dev ? (int64_t)*a.nlist_dev : a.nlist);
  v.w = v.valid ? a.list[li] : 0;
  v.valid = v.valid && a.b.status[v.w] == 0 && a.done_a[v.w] != 0 && a.done_b[v.w] == 0 &&
            (a.b.cls[v.w] & (D >= 8 ? 0x83 : 0xC3)) == 0;
  v.o0 = 0; v.o2 = 0; v.n1 = 0; v.Lu = 0;
  if (v.valid) {
    v.o0 = a.b.off[3 * (int64_t)v.w];
    v.o2 = a.b.off[3 * (int64_t)v.w + 2];
    v.Lu = (int)(a.b.off[3 * (int64_t)v.w + 3] - v.o2);
    v.n1 = a.b.n1[v.w];
  }
  v.ns = (v.Lu + RS - 1) / RS;
  v.off_xi = 16 + align_up(v.Lu, 4);
  v.off_r1 = v.off_xi + 4 * (v.n1 + 1);
  v.off_x2y = v.off_r1 + align_up(2 * v.n1, 4);
  v.off_b0 = v.off_x2y + align_up(2 * v.n1, 4);
  v.off_b1 = v.off_b0 + align_up(2 * (v.n1 + 1), 4);
  v.off_region = align_up(v.off_b1 + (v.ns > 1 ? 2 * (v.n1 + 1) : 0), 4);
  const int region_min = fused_b_cols_bytes(v.n1, v.Lu);
  v.valid = v.valid && (v.off_region + region_min <= a.slot_bytes) && (score_span(kp, v.n1 + G, v.ns * RS) < 16000) &&
            v.ns <= a.mv_ns && v.n1 + G <= a.mv_tw;
  return v;
}

template <int G, int R, int D, int WV>
__device__ __forceinline__ void fused_b_body(const FusedArgs &a, uint8_t *lds, const int64_t vblk)
{
  constexpr int NW = 64 / G, RS = R * G;     // windows per wave, rows per strip
  constexpr int RW = (R + 1) / 2;                         // ring dwords per lane and slot
  constexpr int kRingDepth = D;
  constexpr int kRingBytes = D * 64 * 4 * RW;
  using mv_t = typename MvWord<R>::type;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane / G, g = lane & (G - 1);
  const KParams kp = a.b.kp;
  const int sidx = wv * NW + q;
  const WinB W = load_win_b<G, R, D>(a, (int64_t)(NW * WV) * vblk + sidx);
  bool valid = W.valid;
  if (WV == 1 && __builtin_amdgcn_ballot_w64(valid) == 0) return;       // nothing left for this wave
  const uint32_t w = W.w;
  const int64_t o0 = W.o0;
  const int n1 = W.n1, Lu = W.Lu, ns = W.ns;
  // LDS: [output characters 64 B][one score ring per wave][window slots]
  uint8_t *chr = lds;
  uint8_t *slot = lds + 64 + WV * kRingBytes + sidx * a.slot_bytes;
  uint32_t *ring = reinterpret_cast<uint32_t *>(lds + 64 + wv * kRingBytes);   // this wave's [D][RW][64] dwords = 2 cells of 16 bits
  const uint16_t *ring16 = reinterpret_cast<const uint16_t *>(ring);
  int32_t *hdr = reinterpret_cast<int32_t *>(slot);
  uint8_t *ys = slot + 16;
  uint32_t *xinfo = reinterpret_cast<uint32_t *>(slot + W.off_xi);
  uint16_t *ring1 = reinterpret_cast<uint16_t *>(slot + W.off_r1);
  int16_t *bnd0 = reinterpret_cast<int16_t *>(slot + W.off_b0);
  int16_t *bnd1 = reinterpret_cast<int16_t *>(slot + W.off_b1);
  mv_t *mv = reinterpret_cast<mv_t *>(a.mv_pool) + ((int64_t)blockIdx.x * WV + wv) * a.mv_ns * a.mv_tw * 64;
  const int mvtw = a.mv_tw;
  uint8_t *ordb = slot + W.off_region;                           // [K2][ns][G]

  unsigned long long stamp_ = (a.debug & 4) ? __builtin_readcyclecounter() : 0;
  if (threadIdx.x < 32) chr[threadIdx.x] = a.b.tab->chr[threadIdx.x];
  if (valid) {
    // staging: loads of four rounds are issued before their values are used
    const int64_t nb = o0 + w;
    const uint8_t *sy = a.b.sym + W.o2;
    const int2 *gx = a.b.xinfo + nb;
    const uint16_t *gr = a.b.ring1 + nb;
    for (int ib = 1 + g; ib <= n1; ib += 4 * G) {
      int2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = ib + u * G; v[u] = i <= n1 ? gx[i] : make_int2(0, 0); }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = ib + u * G;
        if (i <= n1) {
          const uint32_t d1 = (uint32_t)v[u].x & 0xFFFFu, d2 = (uint32_t)v[u].x >> 16;   // <= 6 (ring eligibility)
          xinfo[i] = d1 | (d2 << 4) | ((uint32_t)(v[u].y & 0xFF) << 8) | ((uint32_t)((v[u].y >> 8) & 0xFF) << 16);
        }
      }
    }
    for (int ib = g; ib < n1; ib += 4 * G) {
      uint16_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = ib + u * G; v[u] = i < n1 ? gr[i] : (uint16_t)0; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = ib + u * G; if (i < n1) ring1[i] = v[u]; }
    }
    for (int ib = g; ib < Lu; ib += 4 * G) {
      uint8_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = ib + u * G; v[u] = i < Lu ? sy[i] : (uint8_t)0; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = ib + u * G; if (i < Lu) ys[i] = v[u]; }
    }
    if (g == 0) { bnd0[0] = 1; hdr[2] = kNeg; hdr[3] = -1; }      // border origin: score 0, counts as "open"
  }
  __syncthreads();
  // nodes / letters per lane for the passes that walk them in contiguous chunks
  const int cn = (n1 + G - 1) / G, cy = (Lu + G - 1) / G;
  int cnmax = valid ? cn : 0, cymax = valid ? cy : 0;
  for (int d = G; d < 64; d <<= 1) {
    cnmax = max(cnmax, __shfl_xor(cnmax, d));
    cymax = max(cymax, __shfl_xor(cymax, d));
  }
  cnmax = __builtin_amdgcn_readfirstlane(cnmax);
  cymax = __builtin_amdgcn_readfirstlane(cymax);
  // ---- index of the nodes with two predecessors (they own a row of ordinal bytes), final fit check ----
  {
    const int j0 = 1 + g * cn, j1 = valid ? min(n1 + 1, j0 + cn) : 0;
    int cnt = 0;
    for (int it = 0; it < cnmax; ++it) {
      const int jj = j0 + it;
      if (jj < j1) cnt += ((xinfo[jj] >> 4) & 15u) != 0;
    }
    int sc = cnt;
    for (int d = 1; d < G; d <<= 1) {
      const int t = __shfl_up(sc, d, G);
      if (g >= d) sc += t;
    }
    const int k2n = __shfl(sc, G - 1, G);
    int k = sc - cnt;
    for (int it = 0; it < cnmax; ++it) {
      const int jj = j0 + it;
      if (jj < j1) {
        const uint32_t inf = xinfo[jj];
        if ((inf >> 4) & 15u) { xinfo[jj] = inf | ((uint32_t)min(k, 255) << 24); ++k; }
      }
    }
    valid = valid && k2n <= 255 &&
            (W.off_region + max(k2n * ns * G, fused_b_cols_bytes(n1, Lu)) <= a.slot_bytes);
  }
  __syncthreads();

  PHASE_STAMP(8);
  int tmax = valid ? n1 + G - 1 : 0, nsmax = valid ? ns : 0;
  for (int d = G; d < 64; d <<= 1) {
    tmax = max(tmax, __shfl_xor(tmax, d));
    nsmax = max(nsmax, __shfl_xor(nsmax, d));
  }
  tmax = __builtin_amdgcn_readfirstlane(tmax);
  nsmax = __builtin_amdgcn_readfirstlane(nsmax);

  int best = kNeg, bestx = -1;
  int dbg_one = 0, dbg_two = 0;                 // diagnostics (debug bit 5): steps on the one- / two-predecessor path
  const int gstar = ((Lu - 1) % RS) / R, kstar = ((Lu - 1) % RS) % R;
  if (a.debug & 1) nsmax = 0;
  // A wavefront whose windows all have a plain chain as their graph (k_trivial: corrected = reference;
  // the partition puts such windows side by side) needs no predecessor ring: every node's only
  // predecessor is the previous column, so the cells left of and above-left of a lane's block are its
  // own registers and the DPP value of the step before -- the recurrence of k_fused_a, at a third of
  // the per-step overhead.  Same moves, same borders, same end cell (the chain's last node is its only
  // FINAL one), so everything after the loop is shared.
  const bool chain_wave = a.triv != nullptr && nsmax == 1 && !(a.debug & 128) &&
                          __builtin_amdgcn_ballot_w64(valid && a.triv[w] == 0) == 0;
  if (chain_wave) {
    int yl[R], S[R], Ex[R], Ey[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int ii = R * g + 1 + k;
      yl[k] = (valid && ii <= Lu) ? ys[ii - 1] : 255;
      S[k] = -(kp.open_y + (ii - 1) * kp.ext_y);           // column -1: ii gap steps from the origin
      Ex[k] = S[k] - kp.ext_x;
      Ey[k] = S[k] - kp.ext_y;
    }
    int dg0 = (g == 0) ? 0 : -(kp.open_y + (R * g - 1) * kp.ext_y);   // cell (row above, column -1)
    int xl_next = (valid && g == 0 && n1 >= 1) ? (int)((xinfo[1] >> 8) & 0xFF) : 0;
    for (int t = 1; t <= tmax; ++t) {
      const int bS = -(kp.open_x + (t - 1) * kp.ext_x), bEy = bS - kp.ext_y;   // the virtual row over a chain
      const int upS = shift_in<G>(bS, S[R - 1], g);
      const int upEy = shift_in<G>(bEy, Ey[R - 1], g);
      const int jj = t - g;
      const int xl = xl_next;
      xl_next = (valid && jj >= 0 && jj < n1) ? (int)((xinfo[jj + 1] >> 8) & 0xFF) : 0;
      if (valid && jj >= 1 && jj <= n1) {
        int diag = dg0, insY = upEy;
        uint32_t mv8 = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const int oldS = S[k];
          const CellOut c = cell_1pred(diag, Ex[k], insY, xl == yl[k] ? kp.match : kp.mismatch);
          S[k] = c.S;
          Ex[k] = c.S - (c.m ? kp.open_x : kp.ext_x);
          Ey[k] = c.S - (c.m ? kp.open_y : kp.ext_y);
          mv8 |= ((c.nib & 1) | (c.nib >> 1)) << (2 * k);    // 2 bits: bit0 = x step, bit1 = y step
          diag = oldS; insY = Ey[k];
        }
        dg0 = upS;
        mv[t * 64 + lane] = (mv_t)mv8;
      }
    }
    if (valid && g == gstar) {
#pragma unroll
      for (int k = 0; k < R; ++k) if (k == kstar) best = S[k];
      bestx = n1 - 1;
    }
  } else
  for (int s = 0; s < nsmax; ++s) {
    const bool sv = valid && s < ns;
    const int16_t *bcur = (s & 1) ? bnd1 : bnd0;
    int16_t *bnext = (s & 1) ? bnd0 : bnd1;
    const int ii0 = RS * s + R * g + 1;
    int yl[R], S[R], M[R], Ey[R], colS[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int ii = ii0 + k;
      yl[k] = (sv && ii <= Lu) ? ys[ii - 1] : 255;
      colS[k] = -(kp.open_y + (ii - 1) * kp.ext_y);
      S[k] = colS[k]; M[k] = 0;
      Ey[k] = S[k] - kp.ext_y;
    }
    const bool wr_carry = sv && (s + 1 < ns) && g == G - 1;
    const bool last_strip_row = sv && s == ns - 1 && g == gstar;
    uint32_t xi_next = (sv && g == 0 && n1 >= 1) ? xinfo[1] : 0u;
    for (int t = 1; t <= tmax; ++t) {
      const int jj = t - g;
      const uint32_t xi = xi_next;
      xi_next = (sv && jj >= 0 && jj < n1) ? xinfo[jj + 1] : 0u;
      const bool act = sv && jj >= 1 && jj <= n1;
      const int d1i = xi & 15, d2i = (xi >> 4) & 15;
      const int pp1 = d1i ? jj - d1i : 0, pp2 = d2i == 15 ? 0 : jj - d2i;
      const bool has2 = act && d2i != 0;
      const int ppa = act ? pp1 : 0;
      const int b1 = bcur[sv ? ppa : 0], b2 = bcur[has2 ? pp2 : 0];
      int bc = (sv && t <= n1) ? bcur[t] : 0;
      if (s == 0 && g == 0 && act) {
        // the virtual row -1 over the graph (align_lpo_po2.c:275-286): column t, one step ahead of its use
        int r = (b1 >> 1) - ((b1 & 1) ? kp.open_x : kp.ext_x);
        if (has2) r = max(r, (b2 >> 1) - ((b2 & 1) ? kp.open_x : kp.ext_x));
        bc = r << 1;
        bnd0[t] = (int16_t)bc;
      }
      const int bS = bc >> 1, bEy = bS - ((bc & 1) ? kp.open_y : kp.ext_y);
      const int upEy = shift_in<G>(bEy, Ey[R - 1], g);
      const int lm1 = (lane - 1) & 63;
      // ---- predecessor cells.  Own R cells at column pp: this lane's ring slot of
      // d = jj - pp steps ago; the cell above them: lane-1's slot one step earlier (the strip
      // border array for the group's first lane).  The cells of the virtual start column are a
      // function of the row (patched in below, rare); a missing second predecessor gets very
      // negative cells: no value selects. ----
      const bool virt1 = act && ppa == 0, virt2 = has2 && pp2 == 0;
      const int xl = (xi >> 8) & 0xFF;
      uint32_t mv8 = 0, sec4 = 0;
      int nS[R], nM[R];
      // One step of the lane's R cells.  TWO = some lane of the wave is at a node with two predecessors
      // (about half of the steps on well-corrected reads); otherwise the second predecessor's loads,
      // unpacking and comparisons are not even issued.
      auto cells = [&](auto two_tag) {
        constexpr bool TWO = decltype(two_tag)::value;
        const int sa = (t - (jj - ppa)) & (kRingDepth - 1);
        const int sat = (t - (jj - ppa) - 1) & (kRingDepth - 1);
        const int sb = (t - (jj - pp2)) & (kRingDepth - 1);
        const int sbt = (t - (jj - pp2) - 1) & (kRingDepth - 1);
        uint32_t c1[RW], c2[RW];
#pragma unroll
        for (int d = 0; d < RW; ++d) {
          c1[d] = ring[(sa * RW + d) * 64 + lane];
          c2[d] = TWO ? ring[(sb * RW + d) * 64 + lane] : 0u;
        }
        // the last cell of the lane above: dword (R-1)/2, half (R-1)&1 of its slot one step earlier
        int r1 = (int16_t)ring16[((sat * RW + ((R - 1) >> 1)) * 64 + lm1) * 2 + ((R - 1) & 1)];
        int r2 = TWO ? (int)(int16_t)ring16[((sbt * RW + ((R - 1) >> 1)) * 64 + lm1) * 2 + ((R - 1) & 1)] : 0;
        if (__builtin_amdgcn_ballot_w64(virt1 || (TWO && virt2)) != 0) {
          // column -1: row ii holds -(open_y + (ii - 1) ext_y), not a match; the row above this lane's first is ii0 - 1
          const int above = (colS[0] + kp.ext_y) * 2;
#pragma unroll
          for (int d = 0; d < RW; ++d) {
            const uint32_t lo = (uint32_t)((colS[2 * d] * 2) & 0xFFFF);
            const uint32_t hi = (2 * d + 1 < R) ? (uint32_t)((colS[(2 * d + 1 < R) ? 2 * d + 1 : 0] * 2) & 0xFFFF) : 0u;
            if (virt1) c1[d] = lo | (hi << 16);
            if (TWO && virt2) c2[d] = lo | (hi << 16);
          }
          if (virt1) r1 = above;
          if (TWO && virt2) r2 = above;
        }
        if (act) {
          const int d1top = ((g == 0) ? b1 : r1) >> 1;
          const int d2top = (TWO && has2) ? (((g == 0) ? b2 : r2) >> 1) : kNeg16;
          int o1S[R], o1M[R], o2S[R], o2M[R];
#pragma unroll
          for (int k = 0; k < R; ++k) {
            const int e1 = (k & 1) ? (int)((int32_t)c1[k >> 1] >> 16) : (int)(int16_t)(c1[k >> 1] & 0xFFFF);
            o1S[k] = e1 >> 1; o1M[k] = e1 & 1;
            if (TWO) {
              const int e2 = (k & 1) ? (int)((int32_t)c2[k >> 1] >> 16) : (int)(int16_t)(c2[k >> 1] & 0xFFFF);
              o2S[k] = has2 ? e2 >> 1 : kNeg16; o2M[k] = has2 ? e2 & 1 : 0;
            } else { o2S[k] = kNeg16; o2M[k] = 0; }
          }
          // ---- the lane's R cells, top to bottom ----
          int insY = upEy, dt1 = d1top, dt2 = d2top;
#pragma unroll
          for (int k = 0; k < R; ++k) {
            const int cx1 = o1S[k] - (o1M[k] ? kp.open_x : kp.ext_x);
            const int cx2 = TWO ? o2S[k] - (o2M[k] ? kp.open_x : kp.ext_x) : kNeg16;
            const bool px2 = TWO && cx2 > cx1;                  // first maximum wins (:361-371)
            const int insX = TWO ? max(cx1, cx2) : cx1;
            const bool pm2 = TWO && dt2 > dt1;                  // (:348-357)
            const int mat = (TWO ? max(dt1, dt2) : dt1) + (xl == yl[k] ? kp.match : kp.mismatch);
            const int mx = max(insX, insY);
            const bool m = mat > mx;
            const bool xw = insX > insY;
            const int Sn = max(mat, mx);
            // 2 bits per cell: bit 0 = step along x, bit 1 = step along y; which of two predecessors
            // was taken goes to the node's ordinal byte
            mv8 |= (m ? 3u : (xw ? 1u : 2u)) << (2 * k);
            if (TWO) sec4 |= ((m ? pm2 : px2) ? 1u : 0u) << k;
            nS[k] = Sn; nM[k] = m ? 1 : 0;
            insY = Sn - (m ? kp.open_y : kp.ext_y);
            dt1 = o1S[k]; dt2 = o2S[k];
            Ey[k] = insY;
          }
        }
      };
      if (__builtin_amdgcn_ballot_w64(has2) != 0) { cells(std::true_type{}); if ((a.debug & 32) && lane == 0) ++dbg_two; }
      else { cells(std::false_type{}); if ((a.debug & 32) && lane == 0) ++dbg_one; }
      if (act) {
#pragma unroll
        for (int k = 0; k < R; ++k) { S[k] = nS[k]; M[k] = nM[k]; }
#pragma unroll
        for (int d = 0; d < RW; ++d) {
          const uint32_t lo = (uint32_t)(((S[2 * d] << 1) | M[2 * d]) & 0xFFFF);
          const uint32_t hi = (2 * d + 1 < R) ? (uint32_t)((S[(2 * d + 1 < R) ? 2 * d + 1 : 0] << 1) | M[(2 * d + 1 < R) ? 2 * d + 1 : 0]) : 0u;
          ring[((t & (kRingDepth - 1)) * RW + d) * 64 + lane] = lo | (hi << 16);
        }
        mv[(s * mvtw + t) * 64 + lane] = (mv_t)mv8;
        if (has2) ordb[((xi >> 24) * ns + s) * G + g] = (uint8_t)sec4;
        if (wr_carry) bnext[jj] = (int16_t)((S[R - 1] << 1) | M[R - 1]);
        if (last_strip_row && ((xi >> 16) & kFlagFinal)) {
          int sv2 = S[0];
#pragma unroll
          for (int k = 1; k < R; ++k) if (k == kstar) sv2 = S[k];
          if (sv2 > best) { best = sv2; bestx = jj - 1; }      // ties keep the smaller column (:410-417)
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (wr_carry) bnext[0] = (int16_t)((-(kp.open_y + (RS * (s + 1) - 1) * kp.ext_y)) * 2);   // column -1 of the carried row
    __builtin_amdgcn_wave_barrier();
  }

  if ((a.debug & 32) && lane == 0) {
    atomicAdd(reinterpret_cast<unsigned long long *>(a.rowinit) + 14, (unsigned long long)dbg_one);
    atomicAdd(reinterpret_cast<unsigned long long *>(a.rowinit) + 15, (unsigned long long)dbg_two);
  }
  if (valid && g == gstar) { hdr[2] = best; hdr[3] = bestx; }
  if (valid) {
    uint16_t *x2y = reinterpret_cast<uint16_t *>(slot + W.off_x2y);
    for (int i = g; i < n1; i += G) x2y[i] = (uint16_t)kNone16;
  }
  __syncthreads();

  PHASE_STAMP(9);
  uint16_t *x2y = reinterpret_cast<uint16_t *>(slot + W.off_x2y);
  uint8_t *cols_st = slot + W.off_region;                               // overlays the moves after the traceback
  uint16_t *col_y = reinterpret_cast<uint16_t *>(slot + W.off_region + align_up(3 * (n1 + Lu) + 8, 4));
  bool bad = false;
  // ---- traceback #2 (align_lpo_po2.c:108-168), G cells per round as in k_fused_a: a cell keeps the walk
  // on its diagonal when it is a match whose chosen predecessor is the node right before it ----
  {
    int x = valid ? hdr[3] : -1, y = Lu - 1, guard = n1 + Lu + 2;
    bool alive = valid && !(a.debug & 2);
    while (__builtin_amdgcn_ballot_w64(alive) != 0) {
      const int cx = x - g, cy = y - g;
      const bool inb = alive && cx >= 0 && cy >= 0;
      int xo = 0, yo = 0, px = cx;
      if (inb) {
        const int r = cy % RS, rl = r / R, rk = r - rl * R;
        const uint32_t inf = xinfo[cx + 1];
        const uint32_t two = ((uint32_t)mv[((cy / RS) * mvtw + (cx + 1 + rl)) * 64 + q * G + rl] >> (2 * rk)) & 3u;
        xo = two & 1; yo = two >> 1;
        if (xo) {
          const int d2v = (inf >> 4) & 15;
          const int sec = d2v ? (ordb[((inf >> 24) * ns + (cy / RS)) * G + rl] >> rk) & 1 : 0;
          const int dd = sec ? d2v : (int)(inf & 15);
          px = (dd == 0 || dd == 15) ? -1 : cx - dd;
        }
      }
      const int run = diag_run<G>(inb && xo && yo && px == cx - 1, q);
      if (inb && xo && yo && g <= run) x2y[cx] = (uint16_t)cy;      // the run's pairs, and the breaker's if it is a match
      const bool stop = !inb || (!xo && !yo);
      int nx = px, ny = cy - yo, fl = (stop ? 1 : 0) | ((inb && !xo && !yo) ? 2 : 0);
      const int src = min(run, G - 1);
      nx = __shfl(nx, src, G); ny = __shfl(ny, src, G); fl = __shfl(fl, src, G);
      if (run >= G) { nx = x - G; ny = y - G; fl = 0; }
      if (alive) {
        x = nx; y = ny;
        if (fl & 2) bad = true;
        if ((fl & 1) || --guard <= 0) alive = false;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (a.keep_map && valid) {
    uint32_t *gm = a.b.map16 + o0 + w;
    for (int i = g; i < n1; i += G) gm[i] = x2y[i] == (uint16_t)kNone16 ? kNone32 : (uint32_t)x2y[i];
  }
  PHASE_STAMP(13);
  // ---- fusion #2 and the MSA columns (lpo.c:602-668 column layout rule, lpo_format.c:337-393), spread
  // over the window's G lanes.  Only the columns are needed, not the fused graph: every ring of the
  // (ref + cor) graph is one column, an uncorrected letter aligned to one of the ring's nodes joins it,
  // every other uncorrected letter gets a column of its own just before the next ring that holds an
  // aligned letter (at the end when there is none).  Column of ring k = k + (aligned letter of the last
  // aligned ring <= k) - (aligned rings before that one); scans over nodes give those terms.
  int ncol = 0;
  {
    if (a.debug & 2) { cnmax = 0; cymax = 0; }
    if (valid) for (int i = g; i < Lu; i += G) col_y[i] = (uint16_t)kNone16;
    const int i0 = g * cn, i1 = valid ? min(n1, i0 + cn) : 0;
    // the aligned letter of the ring that starts at node ix (-1: none); cnt > 1 cannot come from a path
    auto ring_aligned = [&](int ix, int r0, bool *odd) {
      int ay = -1, cnt = 0;
      for (int k = ix; k < n1 && ring1[k] == r0; ++k) {
        const int v = x2y[k];
        if (v != (int)kNone16) { if (!cnt) ay = v; ++cnt; }
      }
      if (cnt > 1) *odd = true;
      return ay;
    };
    bool odd = false;
    int ngs = 0, nal = 0, last_ay = -1;
    for (int it = 0; it < cnmax; ++it) {
      const int ix = i0 + it;
      if (ix < i1) {
        const int r0 = ring1[ix];
        if (ix == 0 || ring1[ix - 1] != r0) {
          ++ngs;
          const int ay = ring_aligned(ix, r0, &odd);
          if (ay >= 0) { ++nal; if (ay <= last_ay) odd = true; last_ay = ay; }
        }
      }
    }
    int sg = ngs, sa = nal, sy = last_ay;                                  // inclusive scans over the group's lanes
    for (int d = 1; d < G; d <<= 1) {
      const int tg = __shfl_up(sg, d, G), ta = __shfl_up(sa, d, G), ty = __shfl_up(sy, d, G);
      if (g >= d) { sg += tg; sa += ta; sy = max(sy, ty); }
    }
    const int prev_ay = __shfl_up(sy, 1, G);
    int gcount = sg - ngs, alc = sa - nal, A = g > 0 ? prev_ay : -1;
    const int ngroups = __shfl(sg, G - 1, G), nal_all = __shfl(sa, G - 1, G);
    ncol = ngroups + Lu - nal_all;
    __builtin_amdgcn_wave_barrier();
    for (int it = 0; it < cnmax; ++it) {
      const int ix = i0 + it;
      if (ix < i1) {
        const int r0 = ring1[ix];
        if (ix == 0 || ring1[ix - 1] != r0) {
          int ay = -1;
          uint8_t c0 = '.', c1 = '.';
          for (int k = ix; k < n1 && ring1[k] == r0; ++k) {
            const uint32_t xv = xinfo[k + 1];
            const uint8_t ch = chr[(xv >> 8) & 31];
            if ((xv >> 16) & kFlagHasRef) c0 = ch;
            if ((xv >> 16) & kFlagHasCor) c1 = ch;
            const int v = x2y[k];
            if (v != (int)kNone16 && ay < 0) ay = v;
          }
          if (ay >= 0) { if (ay <= A) odd = true; ++alc; A = ay; }
          const int col = gcount + (alc >= 1 ? A - (alc - 1) : 0);
          if (col >= 0 && col < n1 + Lu) {
            cols_st[3 * col] = c0;
            cols_st[3 * col + 1] = c1;
            cols_st[3 * col + 2] = ay >= 0 ? chr[ys[ay] & 31] : (uint8_t)'.';
          } else odd = true;
          if (ay >= 0) col_y[ay] = (uint16_t)col;
          ++gcount;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // letters without a partner: column = K + y with K = (column - letter) of the next aligned letter
    constexpr int kUndef = -0x40000000;
    const int y0 = g * cy, y1 = valid ? min(Lu, y0 + cy) : 0;
    int klow = kUndef;
    for (int it = 0; it < cymax; ++it) {
      const int y = y1 - 1 - it;
      if (y >= y0) { const int c = col_y[y]; if (c != (int)kNone16) klow = c - y; }
    }
    int sfx = klow;                                                        // nearest defined value at or after this lane
    for (int d = 1; d < G; d <<= 1) {
      const int t = __shfl_down(sfx, d, G);
      if (g + d < G && sfx == kUndef) sfx = t;
    }
    int K = __shfl_down(sfx, 1, G);
    if (g == G - 1 || K == kUndef) K = ncol - Lu;
    for (int it = 0; it < cymax; ++it) {
      const int y = y1 - 1 - it;
      if (y >= y0) {
        const int c = col_y[y];
        if (c != (int)kNone16) K = c - y;
        else {
          const int col = K + y;
          if (col >= 0 && col < n1 + Lu) {
            cols_st[3 * col] = '.'; cols_st[3 * col + 1] = '.'; cols_st[3 * col + 2] = chr[ys[y] & 31];
          } else odd = true;
        }
      }
    }
    for (int d = 1; d < G; d <<= 1) odd = odd || __shfl_xor(odd ? 1 : 0, d, G) != 0;
    __builtin_amdgcn_wave_barrier();
    // an alignment that is not a monotone path through the rings cannot happen; if it ever does,
    // the plain serial form of the rule decides
    if (valid && odd && g == 0 && !bad) ncol = fuse2_columns_serial(n1, Lu, xinfo, ring1, x2y, ys, chr, cols_st);
    ncol = __shfl(ncol, 0, G);
  }
  __builtin_amdgcn_wave_barrier();
  PHASE_STAMP(10);
  if (valid) {
    uint8_t *gc = a.b.cols + 3 * o0;
    for (int i = g; i < 3 * ncol; i += G) gc[i] = cols_st[i];
    if (g == 0) {
      a.b.ncol[w] = ncol;
      a.b.score2[w] = hdr[2];
      a.b.bx2[w] = hdr[3];
      if (bad) a.b.status[w] = 3;
      a.done_b[w] = 1;
    }
  }
  PHASE_STAMP(11);
  if ((a.debug & 4) && threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long *>(a.rowinit) + 12, 1ull);
}

template <int G, int R, int D, int WV>
__global__ void __launch_bounds__(64 * WV) k_fused_b(FusedArgs a)
{
  extern __shared__ __align__(16) uint8_t lds[];
  constexpr int NW = 64 / G;
  const int64_t cnt = a.nlist_dev ? (int64_t)*a.nlist_dev : a.nlist;
  for (int64_t vblk = blockIdx.x; vblk * (NW * WV) < cnt; vblk += gridDim.x) {
    fused_b_body<G, R, D, WV>(a, lds, vblk);
    __syncthreads();
  }
}

// ---------------------------------------------------------------- launcher ---

// blocks of a launch: one per NB windows of the list, or a.grid_blocks when the caller caps the grid (the
// kernels loop over the list)
static int64_t fused_grid(const FusedArgs &a, int NB)
{
  const int64_t full = (a.nlist + NB - 1) / NB;
  return a.grid_blocks > 0 && a.grid_blocks < full ? a.grid_blocks : full;
}

template <int G, int R>
static int launch_a_t(const FusedArgs &a, hipStream_t st)
{
  constexpr int NB = 64 / G;     // windows per block (one wave)
  static DeviceOnce once;
  if (once.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_fused_a<G, R, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024 - 256) != hipSuccess)
      return -1;
    once.done();
  }
  hipLaunchKernelGGL((k_fused_a<G, R, 1>), dim3((unsigned)fused_grid(a, NB)), dim3(64), NB * a.slot_bytes, st, a);
  return 0;
}

template <int G, int R, int D>
static int launch_b_t(const FusedArgs &a, hipStream_t st)
{
  constexpr int NB = 64 / G;
  static DeviceOnce once;
  if (once.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_fused_b<G, R, D, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024 - 256) != hipSuccess)
      return -1;
    once.done();
  }
  hipLaunchKernelGGL((k_fused_b<G, R, D, 1>), dim3((unsigned)fused_grid(a, NB)), dim3(64),
                     64 + NB * a.slot_bytes + fused_ring_bytes(R, D), st, a);
  return 0;
}

// the geometry classes the host may ask for (poa_host.hip kFusedClasses)
#define ELECTOR_FUSED_CLASSES(X) \
  X(8, 4) X(8, 5) X(8, 6) X(8, 7) X(8, 8) X(16, 5) X(16, 6) X(16, 7) X(16, 8) \
  X(32, 5) X(32, 6) X(32, 7) X(32, 8) X(64, 5) X(64, 6) X(64, 7) X(64, 8)

int launch_fused_a(const FusedArgs &a, int G, int R, hipStream_t st)
{
  if (a.nlist <= 0) return 0;
#define X(g, r) if (G == g && R == r) return launch_a_t<g, r>(a, st);
  ELECTOR_FUSED_CLASSES(X)
#undef X
  return -2;
}

int launch_fused_b(const FusedArgs &a, int G, int R, int D, hipStream_t st)
{
  if (a.nlist <= 0) return 0;
#define X(g, r) if (G == g && R == r && D == 8) return launch_b_t<g, r, 8>(a, st);
  ELECTOR_FUSED_CLASSES(X)
#undef X
  return -2;
}

}  // namespace elector
