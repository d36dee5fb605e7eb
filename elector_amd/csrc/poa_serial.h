// elector_amd/csrc/poa_serial.h -- the per-window serial stages shared by the
// generic kernels (poa_kernels.hip):
// traceback (align_lpo_po2.c:108-168), fusion (lpo.c:413-463,602-656) and MSA
// column emission (lpo_format.c:337-393).  One lane = one window; the caller
// supplies a reader for the move nibble of DP cell (row ii, column jj), both 1-based.
#pragma once
#include <hip/hip_runtime.h>
#include "poa_device.h"

namespace elector {

// moves as the strip kernels k_dp1/k_dp2 store them
struct StripMoves {
  const uint32_t *mv;
  int tw;
  __device__ __forceinline__ uint32_t operator()(int ii, int jj) const
  {
    int sh;
    const int64_t idx = mv_index(tw, ii, jj, &sh);
    return (mv[idx] >> sh) & 15u;
  }
};

struct NodeWriter {
  int2 *xinfo;
  uint16_t *ring1;
  int maxd;
  bool bad;
  bool far = false;          // a predecessor more than 65,535 nodes back: ELECTOR_W_TOOLONG
  // sa, sb: stored predecessors (new indices) in stored order, -1 = none
  __device__ __forceinline__ void emit(int n, int letter, int flags, int ring, int sa, int sb)
  {
    int d1, d2 = 0;                                            // distances back; d1 0 = virtual start, d2 0 = none
    if (sa < 0) { d1 = 0; }                                    // no stored link: [-1]
    else if (flags & kFlagInitial) { d1 = 0; d2 = n - sa; if (sb >= 0) bad = true; }   // virtual -1 first (:69-79)
    else { d1 = n - sa; if (sb >= 0) d2 = n - sb; }
    maxd = max(maxd, max(d1, d2));
    if (d1 > 0xFFFF || d2 > 0xFFFF) { far = true; d1 &= 0xFFFF; d2 &= 0xFFFF; }   // the node record holds 16-bit distances
    const int jj = n + 1;
    xinfo[jj] = make_int2(d1 | (d2 << 16), letter | (flags << 8));
    ring1[n] = (uint16_t)ring;
  }
};

template <class MV>
__device__ void fuse1_window(const BatchArgs &a, const uint32_t w, const MV getmv)
{
  if (a.status[w] || (a.skip_a && a.skip_a[w])) return;
  const int64_t o0 = a.off[3 * (int64_t)w], o1 = a.off[3 * (int64_t)w + 1], o2 = a.off[3 * (int64_t)w + 2];
  const int Lr = (int)(o1 - o0), Lc = (int)(o2 - o1);
  const uint8_t *xs = a.sym + o0, *ys = a.sym + o1;
  const int64_t nb = o0 + w;
  uint32_t *x2y = a.map16 + nb;
  NodeWriter nw{a.xinfo + nb, a.ring1 + nb, 1, false};

  for (int j = 0; j < Lr; ++j) x2y[j] = kNone32;
  {
    int x = Lr - 1, y = Lc - 1, guard = Lr + Lc + 2;
    while (x >= 0 && y >= 0 && guard-- > 0) {
      const uint32_t nib = getmv(y + 1, x + 1);
      const int xo = nib & 3, yo = nib >> 2;
      if (xo && yo) x2y[x] = (uint32_t)y;
      if (!xo && !yo) { nw.bad = true; break; }
      if (xo) --x;
      if (yo) --y;
    }
  }

  int n = 0, iy = 0, lastx = -1, lasty = -1;
  for (int ix = 0; ix < Lr; ++ix) {
    const int ay = (int)x2y[ix];
    const bool al = ay != (int)kNone32;
    if (al)
      while (iy < ay) {                                         // pending y-only letters go first (lpo.c:432-438)
        const int fl = kFlagHasCor | (iy == 0 ? kFlagInitial : 0) | (iy == Lc - 1 ? kFlagFinal : 0);
        nw.emit(n, ys[iy], fl, n, lasty, -1);
        lasty = n; ++n; ++iy;
      }
    int fl = kFlagHasRef | (ix == 0 ? kFlagInitial : 0) | (ix == Lr - 1 ? kFlagFinal : 0);
    int sa = lastx, sb = -1, ring = n;
    if (al && iy < Lc) {
      if (xs[ix] == ys[iy]) {                                   // identical letters fuse (lpo.c:379-382,447-448)
        fl |= kFlagHasCor | (iy == 0 ? kFlagInitial : 0) | (iy == Lc - 1 ? kFlagFinal : 0);
        if (lasty >= 0 && lasty != lastx) { if (sa < 0) sa = lasty; else sb = lasty; }
        nw.emit(n, xs[ix], fl, n, sa, sb);
        lastx = lasty = n; ++n; ++iy;
        continue;
      }
      // mismatch: y gets its own node immediately before x and shares x's ring (lpo.c:449-450,647-649)
      const int fy = kFlagHasCor | (iy == 0 ? kFlagInitial : 0) | (iy == Lc - 1 ? kFlagFinal : 0);
      nw.emit(n, ys[iy], fy, n, lasty, -1);
      ring = n; lasty = n; ++n; ++iy;
    }
    nw.emit(n, xs[ix], fl, ring, sa, sb);
    lastx = n; ++n;
  }
  while (iy < Lc) {                                             // tail of y (lpo.c:457-459)
    const int fl = kFlagHasCor | (iy == 0 ? kFlagInitial : 0) | (iy == Lc - 1 ? kFlagFinal : 0);
    nw.emit(n, ys[iy], fl, n, lasty, -1);
    lasty = n; ++n; ++iy;
  }
  a.n1[w] = n;
  // ring depth class for k_dp2: D must cover max predecessor distance + 2
  const int need = nw.maxd + 2;
  a.cls[w] = (uint8_t)((need <= 32 ? 0 : 1) | (need > 4 ? 0x40 : 0) | (need > 8 ? 0x80 : 0));
  if (nw.far) a.status[w] = 2;
  if (nw.bad) a.status[w] = 3;
}


// ---------------------------------------------------------------- k_fuse2 ---
// One lane per window: trace alignment #2 back through the PO predecessor
// lists, fuse the uncorrected read (lpo.c:413-463 with non-trivial x rings) and
// emit the MSA directly as columns (lpo_format.c:346-371: a new column whenever
// the ring id changes).  The fused graph itself is never materialised.

template <class MV>
__device__ void fuse2_window(const BatchArgs &a, const uint32_t w, const MV getmv)
{
  if (a.skip_b && a.skip_b[w]) return;
  if (a.status[w]) { a.ncol[w] = 0; return; }
  const int64_t o0 = a.off[3 * (int64_t)w], o2 = a.off[3 * (int64_t)w + 2], o3 = a.off[3 * (int64_t)w + 3];
  const int n1 = a.n1[w], Lu = (int)(o3 - o2);
  const uint8_t *ys = a.sym + o2;
  const int64_t nb = o0 + w;
  const int2 *xinfo = a.xinfo + nb;
  const uint16_t *ring1 = a.ring1 + nb;
  uint32_t *x2y = a.map16 + nb;
  uint8_t *cols = a.cols + 3 * o0;
  const uint8_t *chr = a.tab->chr;
  bool bad = false;

  for (int j = 0; j < n1; ++j) x2y[j] = kNone32;
  {
    int x = a.bx2[w], y = Lu - 1, guard = n1 + Lu + 2;
    while (x >= 0 && y >= 0 && guard-- > 0) {
      const uint32_t nib = getmv(y + 1, x + 1);
      const int xo = nib & 3, yo = nib >> 2;
      if (xo && yo) x2y[x] = (uint32_t)y;
      if (!xo && !yo) { bad = true; break; }
      if (xo) {
        const uint32_t pl = (uint32_t)xinfo[x + 1].x;
        const int d = (xo == 1) ? (int)(pl & 0xFFFF) : (int)(pl >> 16);
        x = d ? x - d : -1;                                       // 0 = the virtual start
      }
      if (yo) --y;
    }
  }

  // column writer
  int col = 0, prev_ring = 0;
  uint8_t c0 = '.', c1 = '.', c2 = '.';
  auto flush = [&]() { cols[3 * col] = c0; cols[3 * col + 1] = c1; cols[3 * col + 2] = c2; };
  auto place = [&](int ring, int letter, bool r, bool c, bool u) {
    if (ring != prev_ring) { flush(); ++col; c0 = c1 = c2 = '.'; prev_ring = ring; }
    const uint8_t ch = chr[letter];
    if (r) c0 = ch;
    if (c) c1 = ch;
    if (u) c2 = ch;
  };

  int n = 0, iy = 0, blk_old = -1, blk_new = -1;
  for (int ix = 0; ix < n1; ++ix) {
    const int r0 = ring1[ix];
    if (r0 != blk_old) { blk_old = r0; blk_new = -1; }
    // if any later member of this ring block is aligned, its pending y letters come first (lpo.c:432-438)
    for (int k = ix; k < n1 && ring1[k] == r0; ++k) {
      const int ay = (int)x2y[k];
      if (ay != (int)kNone32) {
        while (iy < ay) { place(n, ys[iy], false, false, true); ++n; ++iy; }
        break;
      }
    }
    const int xi = xinfo[ix + 1].y;
    const int letter = xi & 0xFF, fl = xi >> 8;
    bool fused = false;
    if (x2y[ix] != kNone32 && iy < Lu) {
      if (letter == ys[iy]) fused = true;
      else {
        if (blk_new < 0) blk_new = n;                           // y becomes the ring's smallest index
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
  a.ncol[w] = col + 1;
  if (bad) a.status[w] = 3;
  if (a.mark_b) a.mark_b[w] = 1;
}


}  // namespace elector
