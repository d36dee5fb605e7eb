// elector_amd/csrc/poa_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the
// triplet-MSA engine.  One wavefront = one window for the two dynamic programs;
// one lane = one window for the light serial stages (traceback, fusion, rows).
//
//   k_symbolize   a2   raw bytes -> symbol indices           (seq_util.c:37-52,253-263; create_seq.c:129-132)
//   k_dp1         a5   linear(ref) x linear(cor) DP          (align_lpo_po2.c:178-433)
//   k_fuse1       a7+a8 traceback #1 + fusion #1 -> PO graph (align_lpo_po2.c:108-168; lpo.c:413-463,602-656)
//   k_dp2         a4+a5 PO(ref+cor) x linear(unc) DP         (align_lpo_po2.c:29-105,178-433)
//   k_fuse2       a7+a8+a10 traceback #2 + fusion #2 + MSA columns (lpo_format.c:337-393)
//   k_rows        column-interleaved MSA -> three rows (host-buffer API only)
//
// No MFMA: the recurrence is an integer max-plus chain, not a contraction.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "poa_device.h"
#include "poa_serial.h"

namespace elector {

// ---------------------------------------------------------------- helpers ---

__device__ __forceinline__ int wave_shr1(int v)
{
  // DPP wave_shr:1 -- lane l receives lane l-1's value, lane 0 keeps its own.
  return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xF, 0xF, false);
}

__device__ __forceinline__ int pack_cell(int s, int g) { return (s << kTagBits) | g; }
__device__ __forceinline__ int cell_score(int p) { return p >> kTagBits; }
__device__ __forceinline__ int cell_tag(int p) { return p & ((1 << kTagBits) - 1); }

// carry rows are handed from one strip to the next through memory by the same
// wave; sc1 accesses keep the per-CU L1 out of the picture.
__device__ __forceinline__ int ld_carry(const int32_t *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_carry(int32_t *p, int v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool GEN>
struct Scoring {
  const int *gpx, *gpy, *sub;   // LDS tables when GEN
  KParams kp;
  __device__ __forceinline__ int pen_x(int g) const { return GEN ? gpx[g] : (g ? kp.ext_x : kp.open_x); }
  __device__ __forceinline__ int pen_y(int g) const { return GEN ? gpy[g] : (g ? kp.ext_y : kp.open_y); }
  __device__ __forceinline__ int subst(int a, int b) const
  {
    return GEN ? sub[a * 32 + b] : (a == b ? kp.match : kp.mismatch);
  }
  // gap-tag transition (align_lpo_po2.c:230-249); the initial tag M+1 behaves
  // exactly like 0 in global mode, so borders use 0.
  __device__ __forceinline__ int next_tag(int g) const { return min(g + 1, kp.M); }
};

template <bool GEN>
__device__ __forceinline__ void load_tables(int *lds, const DevTables *tab, int lane)
{
  if (GEN) {
    for (int i = lane; i < 64; i += 64) { lds[i] = tab->gpx[i]; lds[64 + i] = tab->gpy[i]; }
    for (int i = lane; i < 1024; i += 64) lds[128 + i] = tab->sub[i];
    __syncthreads();
  }
}

// ------------------------------------------------------------ k_symbolize ---

__global__ void __launch_bounds__(256) k_symbolize(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                    int64_t nbytes, const DevTables *__restrict__ tab)
{
  __shared__ uint8_t lut[256];
  lut[threadIdx.x] = tab->lut[threadIdx.x];
  __syncthreads();
  const int64_t nvec = nbytes >> 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    uint4 v = reinterpret_cast<const uint4 *>(in)[i];
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t x = w[k];
      w[k] = (uint32_t)lut[x & 255] | ((uint32_t)lut[(x >> 8) & 255] << 8) |
             ((uint32_t)lut[(x >> 16) & 255] << 16) | ((uint32_t)lut[x >> 24] << 24);
    }
    reinterpret_cast<uint4 *>(out)[i] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  // tail bytes
  for (int64_t i = (nvec << 4) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += stride)
    out[i] = lut[in[i]];
}

// ------------------------------------------------------------------ k_dp1 ---
// Alignment #1: x = reference (linear), y = corrected (linear).  Every column
// has the single predecessor jj-1, so the three neighbours of a cell come from
// registers: left = own previous step, up = lane-1 previous step (DPP), diag =
// the `up` received one step earlier.

// NW wavefronts per window: the strips side by side, each a good 128 steps behind the one above (see dp2_window).
template <bool GEN, int NW>
__device__ void dp1_window(const BatchArgs &a, const Scoring<GEN> &sc, const uint32_t w, const int lane, const int wv,
                           volatile int *prog)
{
  if (a.status[w] || (a.skip_a && a.skip_a[w]) || (a.tiled && (a.tiled[w] & 1))) return;
  const int64_t o0 = a.off[3 * (int64_t)w], o1 = a.off[3 * (int64_t)w + 1], o2 = a.off[3 * (int64_t)w + 2];
  const int Lx = (int)(o1 - o0), Ly = (int)(o2 - o1);
  const uint8_t *xs = a.sym + o0, *ys = a.sym + o1;
  uint32_t *mv = a.moves + a.mv1[w];
  int32_t *carry = a.carry + (o0 + w);
  const int tw = mv_tw(Lx), ns = n_strips(Ly);

  for (int s = wv; s < ns; s += NW) {
    const int ii = s * kStripRows + lane;
    const bool rowok = lane >= 1 && ii <= Ly;
    const int yl = rowok ? ys[ii - 1] : 0;
    const int colp = a.liny[min(ii, Ly)];
    const int32_t *src0 = (s == 0) ? a.linx : carry;
    const bool wr_carry = (s + 1 < ns);
    const int nl = min(kStripRows, Ly - s * kStripRows);
    const int T = Lx + nl;
    int S1 = 0, g1 = 0, sdiag = 0, xl = 0, xblk = 0, c0blk = 0;
    uint32_t mvacc = 0;
    for (int t = 0; t <= T; ++t) {
      if ((t & 63) == 0) {
        const int j = t + lane;
        xblk = (j >= 1 && j <= Lx) ? xs[j - 1] : 0;
        if (NW > 1 && s > 0) {
          const int need = ((s - 1) << 20) | (t + 127);
          while (prog[(wv + NW - 1) % NW] < need) __builtin_amdgcn_s_sleep(4);
          asm volatile("" ::: "memory");
        }
        c0blk = (j <= Lx) ? ld_carry(src0 + j) : 0;
      }
      const int x0 = __builtin_amdgcn_readlane(xblk, t & 63);
      const int c0 = __builtin_amdgcn_readlane(c0blk, t & 63);
      const int s_up = wave_shr1(S1), g_up = wave_shr1(g1);
      xl = wave_shr1(xl);
      if (lane == 0) xl = x0;
      const int jj = t - lane;
      const bool cell = rowok && jj >= 1 && jj <= Lx;

      const int insY = s_up - sc.pen_y(g_up);
      const int mat = sdiag + sc.subst(xl, yl);
      const int insX = S1 - sc.pen_x(g1);
      const bool m = (mat > insY) && (mat > insX);          // align_lpo_po2.c:384
      const bool xw = !m && (insX > insY);                  // :392
      int S = m ? mat : (xw ? insX : insY);
      int g = m ? 0 : sc.next_tag(xw ? g1 : g_up);
      uint32_t nib = m ? (kMoveX1 | kMoveY) : (xw ? kMoveX1 : kMoveY);
      if (!cell) { S = S1; g = g1; nib = 0; }
      if (jj == 0) { S = cell_score(colp); g = cell_tag(colp); }
      if (lane == 0) { S = cell_score(c0); g = cell_tag(c0); }
      sdiag = s_up;
      S1 = S; g1 = g;

      mvacc |= nib << (4 * (t & 7));
      if ((t & 7) == 7 || t == T) { mv[((int64_t)s * tw + (t >> 3)) * 64 + lane] = mvacc; mvacc = 0; }
      if (wr_carry && lane == 63 && jj >= 0 && jj <= Lx) st_carry(carry + jj, pack_cell(S, g));
      if (cell && ii == Ly && jj == Lx) a.score1[w] = S;    // the only FINAL x FINAL cell
      if (NW > 1 && wr_carry && (t & 63) == 63) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) prog[wv] = (s << 20) | (t + 1);
      }
    }
    if (wr_carry) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (NW > 1 && lane == 0) prog[wv] = (s + 1) << 20;
    }
  }
}

// number of windows in the launch's list: host value, or a device counter for
// lists built on the device
__device__ __forceinline__ int64_t list_count(const BatchArgs &a)
{
  return a.count_ptr ? (int64_t)*a.count_ptr : a.n;
}

template <bool GEN, int NW>
__global__ void __launch_bounds__(64 * NW) k_dp1(BatchArgs a)
{
  __shared__ int lds_tab[GEN ? (128 + 1024) : 1];
  __shared__ int prog_lds[NW];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  load_tables<GEN>(lds_tab, a.tab, threadIdx.x);
  Scoring<GEN> sc{lds_tab, lds_tab + 64, lds_tab + 128, a.kp};
  volatile int *prog = prog_lds;
  const int64_t cnt = list_count(a);
  for (int64_t i = blockIdx.x; i < cnt; i += gridDim.x) {
    if (NW > 1) {
      __syncthreads();
      if (threadIdx.x < NW) prog[threadIdx.x] = 0;
      __syncthreads();
    }
    dp1_window<GEN, NW>(a, sc, a.perm[i], lane, wv, prog);
  }
}

// ------------------------------------------------------------------ k_dp2 ---
// Alignment #2: x = PO graph of (ref + cor), y = uncorrected (linear).  A node
// has at most two DP predecessors (one per source read, or the virtual start
// plus one).  Predecessors can be any number of nodes back, so every lane keeps
// its last D results in an LDS ring indexed by time: cell (ii, pp) was produced
// by this lane d = jj - pp steps ago, and (ii-1, pp) by lane-1 one step before
// that.  Bank = lane, so ring accesses never conflict.

// DEEP: predecessors farther back than the LDS ring reaches (a corrected piece whose long unaligned
// tail sits between two reference letters, Master_Splitter.cpp:295-301, puts hundreds of nodes between
// a node and its predecessor) come from a shadow of the ring in HBM that holds every step of the strip:
// gring[t][lane], written once per step (fire and forget), read only by the rare far accesses.
//
// NW wavefronts per window: wavefront v takes the strips v, v + NW, ... and runs strip s a good 128 steps behind strip
// s - 1 (whose last lane hands down the carry row, column jj at step jj + 63; a strip asks for 64 columns of it at a
// time).  A wavefront alone issues one instruction in 2 ns whatever it is, and a step of this recurrence is some 250
// instructions: a 500-row window (9 strips x 1,100 steps) is 5 ms of one wavefront, which is what a batch's last
// launches -- the few hundred windows the on-chip kernels hand back -- held the chip for; side by side the strips take
// T + 128 (ns - 1) steps instead of ns T.  prog[v] = (strip << 20) | steps of it that are through (carry stores landed),
// (strip + 1) << 20 when the strip is: monotone per wavefront, read by the wavefront of the next strip.
template <bool GEN, int D, bool DEEP, int NW>
__device__ void dp2_window(const BatchArgs &a, const Scoring<GEN> &sc, int *ring, const int cls, const uint32_t w,
                           const int lane, int32_t *gring, const int wv, volatile int *prog)
{
  if (a.status[w] || (cls >= 0 && (a.cls[w] & 3) != cls) || (a.skip_b && a.skip_b[w]) || (a.tiled && (a.tiled[w] & 2))) return;
  const int64_t o0 = a.off[3 * (int64_t)w], o2 = a.off[3 * (int64_t)w + 2], o3 = a.off[3 * (int64_t)w + 3];
  const int Lx = a.n1[w], Ly = (int)(o3 - o2);
  const uint8_t *ys = a.sym + o2;
  const int2 *xinfo = a.xinfo + (o0 + w);
  uint32_t *mv = a.moves + a.mv2[w];
  int32_t *carry = a.carry + (o0 + w);
  const int tw = mv_tw(Lx), ns = n_strips(Ly);
  constexpr int MASK = D - 1;
  constexpr int kNoPred = 0;                      // d1 = 0 (virtual start), d2 = 0 (none)

  int best = kNeg, bestx = -1;
  for (int s = wv; s < ns; s += NW) {
    const int ii = s * kStripRows + lane;
    const bool rowok = lane >= 1 && ii <= Ly;
    const bool rowvirt = (s == 0 && lane == 0);
    const int yl = rowok ? ys[ii - 1] : 0;
    const int col_own = a.liny[min(ii, Ly)];
    const int col_dg = a.liny[min(max(ii - 1, 0), Ly)];
    const bool wr_carry = (s + 1 < ns);
    const int nl = min(kStripRows, Ly - s * kStripRows);
    const int T = Lx + nl;
    int S1 = 0, g1 = 0, xlo = kNoPred, xhi = 0, xb_lo = kNoPred, xb_hi = 0, c0blk = 0;
    uint32_t mvacc = 0;
    for (int t = 0; t <= T; ++t) {
      if ((t & 63) == 0) {
        const int j = t + lane;
        int2 xi = (j >= 1 && j <= Lx) ? xinfo[j] : make_int2(kNoPred, 0);
        xb_lo = xi.x; xb_hi = xi.y;
        if (NW > 1 && s > 0) {
          // the strip above is through step t + 126: its carry of the columns t .. t + 63 has landed
          const int need = ((s - 1) << 20) | (t + 127);
          while (prog[(wv + NW - 1) % NW] < need) __builtin_amdgcn_s_sleep(4);
          asm volatile("" ::: "memory");
        }
        c0blk = (s > 0 && j <= Lx) ? ld_carry(carry + j) : 0;
      }
      const int x0lo = __builtin_amdgcn_readlane(xb_lo, t & 63);
      const int x0hi = __builtin_amdgcn_readlane(xb_hi, t & 63);
      const int c0 = __builtin_amdgcn_readlane(c0blk, t & 63);
      const int s_up = wave_shr1(S1), g_up = wave_shr1(g1);
      xlo = wave_shr1(xlo); xhi = wave_shr1(xhi);
      if (lane == 0) { xlo = x0lo; xhi = x0hi; }
      const int jj = t - lane;
      const bool incol = jj >= 1 && jj <= Lx;
      const int d1 = xlo & 0xFFFF, d2 = (int)((uint32_t)xlo >> 16);   // distances back to the predecessors
      const int xl = xhi & 0xFF;
      const bool has2 = incol && d2 != 0;
      const bool virt1 = !incol || d1 == 0;
      const int lm1 = max(lane - 1, 0);

      // predecessor cells: own row from this lane's ring, row above from lane-1's
      int own1 = col_own, dg1 = col_dg, own2 = 0, dg2 = 0;
      const bool far1 = DEEP && !virt1 && d1 > D - 2, far2 = DEEP && has2 && d2 > D - 2;
      if (DEEP && __builtin_amdgcn_ballot_w64(far1 || far2) != 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the shadow stores of the steps read below have landed
      if (!virt1) {
        const int d = d1;
        if (far1) {
          own1 = ld_carry(gring + (int64_t)(t - d) * 64 + lane);
          dg1 = ld_carry(gring + (int64_t)(t - d - 1) * 64 + lm1);
        } else {
          own1 = ring[((t - d) & MASK) * 64 + lane];
          dg1 = ring[((t - d - 1) & MASK) * 64 + lm1];
        }
      }
      if (has2) {
        const int d = d2;
        if (far2) {
          own2 = ld_carry(gring + (int64_t)(t - d) * 64 + lane);
          dg2 = ld_carry(gring + (int64_t)(t - d - 1) * 64 + lm1);
        } else {
          own2 = ring[((t - d) & MASK) * 64 + lane];
          dg2 = ring[((t - d - 1) & MASK) * 64 + lm1];
        }
      }
      // X-insertion: first maximum over the predecessor list wins (:361-371)
      const int gx1 = cell_tag(own1), gx2 = cell_tag(own2);
      const int cx1 = cell_score(own1) - sc.pen_x(gx1);
      const int cx2 = has2 ? cell_score(own2) - sc.pen_x(gx2) : kNeg;
      const bool px2 = cx2 > cx1;
      const int insX = px2 ? cx2 : cx1, gX = px2 ? gx2 : gx1;
      // match: first maximum over (y-pred outer, x-pred inner) (:348-357,374-376)
      const int m1 = cell_score(dg1), m2 = has2 ? cell_score(dg2) : kNeg;
      const bool pm2 = m2 > m1;
      int mat = (pm2 ? m2 : m1) + sc.subst(xl, yl);
      // Y-insertion: y is linear, one predecessor (:334-345)
      int insY = s_up - sc.pen_y(g_up);
      if (rowvirt) { mat = kNeg; insY = kNeg; }              // row -1: gaps along x only (:275-286)

      const bool m = (mat > insY) && (mat > insX);
      const bool xw = !m && (insX > insY);
      int S = m ? mat : (xw ? insX : insY);
      int g = m ? 0 : sc.next_tag(xw ? gX : g_up);
      uint32_t nib = m ? ((pm2 ? kMoveX2 : kMoveX1) | kMoveY) : (xw ? (px2 ? kMoveX2 : kMoveX1) : kMoveY);
      const bool cell = rowok && incol;
      if (!(cell || (rowvirt && incol))) { S = S1; g = g1; }
      if (!cell) nib = 0;
      if (jj == 0) { S = cell_score(col_own); g = cell_tag(col_own); }
      if (lane == 0 && s > 0) { S = cell_score(c0); g = cell_tag(c0); }
      S1 = S; g1 = g;

      ring[(t & MASK) * 64 + lane] = pack_cell(S, g);
      if (DEEP) st_carry(gring + (int64_t)t * 64 + lane, pack_cell(S, g));
      __builtin_amdgcn_wave_barrier();

      mvacc |= nib << (4 * (t & 7));
      if ((t & 7) == 7 || t == T) { mv[((int64_t)s * tw + (t >> 3)) * 64 + lane] = mvacc; mvacc = 0; }
      if (wr_carry && lane == 63 && jj >= 0 && jj <= Lx) st_carry(carry + jj, pack_cell(S, g));
      // end cell: FINAL x node on the last row; ties keep the smaller column (:410-417)
      if (cell && ii == Ly && ((xhi >> 8) & kFlagFinal) && S > best) { best = S; bestx = jj - 1; }
      if (NW > 1 && wr_carry && (t & 63) == 63) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) prog[wv] = (s << 20) | (t + 1);
      }
    }
    if (wr_carry) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (NW > 1 && lane == 0) prog[wv] = (s + 1) << 20;
    }
  }
  if (wv == (ns - 1) % NW && lane == (Ly - 1) % kStripRows + 1) { a.score2[w] = best; a.bx2[w] = bestx; }
}

template <bool GEN, int D, bool DEEP, int NW>
__global__ void __launch_bounds__(64 * NW) k_dp2(BatchArgs a, int cls, int32_t *gring, int64_t gring_block)
{
  extern __shared__ int dp2_lds[];                // NW rings of D * 64 cells, then prog[NW]
  __shared__ int lds_tab[GEN ? (128 + 1024) : 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  load_tables<GEN>(lds_tab, a.tab, threadIdx.x);
  Scoring<GEN> sc{lds_tab, lds_tab + 64, lds_tab + 128, a.kp};
  int *ring = dp2_lds + wv * (D * 64);
  volatile int *prog = dp2_lds + NW * (D * 64);
  const int64_t cnt = list_count(a);
  for (int64_t i = blockIdx.x; i < cnt; i += gridDim.x) {
    if (NW > 1) {
      if (threadIdx.x < NW) prog[threadIdx.x] = 0;
      __syncthreads();
    }
    dp2_window<GEN, D, DEEP, NW>(a, sc, ring, cls, a.perm[i], lane,
                                 DEEP ? gring + ((int64_t)blockIdx.x * NW + wv) * gring_block : nullptr, wv, prog);
    __syncthreads();
  }
}

// --------------------------------------------------------------- tiled DP ---
// The same two recurrences for LONG windows, one tile (strip s, column block c) per wavefront and one
// launch per anti-diagonal d = s + c (the grid's y index picks the window).  A tile needs from its left
// neighbour (s, c-1) the cell of every row at the block's last column -- and, for alignment #2, that
// wavefront's predecessor ring -- and from the tile above (s-1, c) the carry row over its own columns;
// both ran on the anti-diagonal before.  The state lives per strip and block parity in `tstate`.
// Moves are OR-ed into the (zeroed) scratch: a lane's dword of eight steps may straddle two tiles.

template <bool GEN>
__global__ void __launch_bounds__(64) k_dp1_tile(BatchArgs a, TileArgs ta)
{
  __shared__ int lds_tab[GEN ? (128 + 1024) : 1];
  const int lane = threadIdx.x;
  load_tables<GEN>(lds_tab, a.tab, lane);
  Scoring<GEN> sc{lds_tab, lds_tab + 64, lds_tab + 128, a.kp};
  const uint32_t w = ta.wlist[blockIdx.y];
  if (a.status[w]) return;
  const int64_t o0 = a.off[3 * (int64_t)w], o1 = a.off[3 * (int64_t)w + 1], o2 = a.off[3 * (int64_t)w + 2];
  const int Lx = (int)(o1 - o0), Ly = (int)(o2 - o1);
  const int tw = mv_tw(Lx), ns = n_strips(Ly), ncb = (Lx + kTileCols - 1) / kTileCols;
  const int s = (int)blockIdx.x + max(0, ta.d - (ncb - 1)), c = ta.d - s;
  if (s >= ns || c < 0) return;
  const uint8_t *xs = a.sym + o0, *ys = a.sym + o1;
  uint32_t *mv = a.moves + a.mv1[w];
  int32_t *carry = a.carry + (o0 + w);
  int32_t *stw = ta.tstate + ta.st_off[blockIdx.y];                     // [ns][2][kTileState1]
  const int c_lo = c * kTileCols, c_hi = min(c_lo + kTileCols, Lx);       // this tile: columns (c_lo, c_hi]

  const int ii = s * kStripRows + lane;
  const bool rowok = lane >= 1 && ii <= Ly;
  const int yl = rowok ? ys[ii - 1] : 0;
  const int colp = a.liny[min(ii, Ly)];
  const int32_t *src0 = (s == 0) ? a.linx : carry;
  const bool wr_carry = (s + 1 < ns);
  int S1 = 0, g1 = 0, sdiag = 0, xl = 0, xblk = 0, c0blk = 0;
  if (c > 0) { const int p = stw[(s * 2 + ((c - 1) & 1)) * kTileState1 + lane]; S1 = cell_score(p); g1 = cell_tag(p); }
  uint32_t mvacc = 0;
  const int t_end = c_hi + 63;
  for (int t = c_lo; t <= t_end; ++t) {
    if ((t & 63) == 0) {
      const int j = t + lane;
      xblk = (j >= 1 && j <= Lx) ? xs[j - 1] : 0;
      c0blk = (j <= c_hi) ? ld_carry(src0 + j) : 0;                        // the row above, this block's columns only
    }
    const int x0 = __builtin_amdgcn_readlane(xblk, t & 63);
    const int c0 = __builtin_amdgcn_readlane(c0blk, t & 63);
    const int s_up = wave_shr1(S1), g_up = wave_shr1(g1);
    xl = wave_shr1(xl);
    if (lane == 0) xl = x0;
    const int jj = t - lane;
    const bool incol = jj > c_lo && jj <= c_hi;
    const bool first = c == 0 && jj == 0;                                  // column -1 of the window
    const bool cell = rowok && incol;

    const int insY = s_up - sc.pen_y(g_up);
    const int mat = sdiag + sc.subst(xl, yl);
    const int insX = S1 - sc.pen_x(g1);
    const bool m = (mat > insY) && (mat > insX);
    const bool xw = !m && (insX > insY);
    int S = m ? mat : (xw ? insX : insY);
    int g = m ? 0 : sc.next_tag(xw ? g1 : g_up);
    uint32_t nib = m ? (kMoveX1 | kMoveY) : (xw ? kMoveX1 : kMoveY);
    if (!cell) { S = S1; g = g1; nib = 0; }
    if (first) { S = cell_score(colp); g = cell_tag(colp); }
    if (lane == 0 && (incol || first)) { S = cell_score(c0); g = cell_tag(c0); }
    sdiag = s_up;
    S1 = S; g1 = g;

    mvacc |= nib << (4 * (t & 7));
    if ((t & 7) == 7 || t == t_end) {
      if (mvacc) mv[((int64_t)s * tw + (t >> 3)) * 64 + lane] |= mvacc;
      mvacc = 0;
    }
    if (wr_carry && lane == 63 && (incol || first)) st_carry(carry + jj, pack_cell(S, g));
    if (cell && ii == Ly && jj == Lx) a.score1[w] = S;
  }
  stw[(s * 2 + (c & 1)) * kTileState1 + lane] = pack_cell(S1, g1);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}

template <bool GEN>
__global__ void __launch_bounds__(64) k_dp2_tile(BatchArgs a, TileArgs ta)
{
  constexpr int D = kTileRing, MASK = D - 1;
  __shared__ int ring[D * 64];
  __shared__ int lds_tab[GEN ? (128 + 1024) : 1];
  const int lane = threadIdx.x;
  load_tables<GEN>(lds_tab, a.tab, lane);
  Scoring<GEN> sc{lds_tab, lds_tab + 64, lds_tab + 128, a.kp};
  const uint32_t w = ta.wlist[blockIdx.y];
  if (a.status[w] || (a.cls[w] & 3) != 0) return;                         // deeper graphs: the one-wavefront kernel
  const int64_t o0 = a.off[3 * (int64_t)w], o2 = a.off[3 * (int64_t)w + 2], o3 = a.off[3 * (int64_t)w + 3];
  const int Lx = a.n1[w], Ly = (int)(o3 - o2);
  const int tw = mv_tw(Lx), ns = n_strips(Ly), ncb = (Lx + kTileCols - 1) / kTileCols;
  const int s = (int)blockIdx.x + max(0, ta.d - (ncb - 1)), c = ta.d - s;
  if (s >= ns || c < 0 || c >= ncb) return;
  const uint8_t *ys = a.sym + o2;
  const int2 *xinfo = a.xinfo + (o0 + w);
  uint32_t *mv = a.moves + a.mv2[w];
  int32_t *carry = a.carry + (o0 + w);
  int32_t *stw = ta.tstate + ta.st_off[blockIdx.y];                     // [ns][2][kTileState2]
  const int32_t *st_in = stw + (int64_t)(s * 2 + ((c - 1) & 1)) * kTileState2;
  int32_t *st_out = stw + (int64_t)(s * 2 + (c & 1)) * kTileState2;
  const int c_lo = c * kTileCols, c_hi = min(c_lo + kTileCols, Lx);
  constexpr int kNoPred = 0;

  const int ii = s * kStripRows + lane;
  const bool rowok = lane >= 1 && ii <= Ly;
  const bool rowvirt = (s == 0 && lane == 0);
  const int yl = rowok ? ys[ii - 1] : 0;
  const int col_own = a.liny[min(ii, Ly)];
  const int col_dg = a.liny[min(max(ii - 1, 0), Ly)];
  const bool wr_carry = (s + 1 < ns);
  int S1 = 0, g1 = 0, xlo = kNoPred, xhi = 0, xb_lo = kNoPred, xb_hi = 0, c0blk = 0;
  int best = kNeg, bestx = -1;
  if (c > 0) {
    const int p = st_in[lane];
    S1 = cell_score(p); g1 = cell_tag(p);
    for (int k = 0; k < D; ++k) ring[k * 64 + lane] = st_in[64 + k * 64 + lane];
    best = st_in[64 + D * 64]; bestx = st_in[64 + D * 64 + 1];
  }
  __syncthreads();
  uint32_t mvacc = 0;
  const int t_end = c_hi + 63;
  for (int t = c_lo; t <= t_end; ++t) {
    if ((t & 63) == 0) {
      const int j = t + lane;
      int2 xi = (j >= 1 && j <= Lx) ? xinfo[j] : make_int2(kNoPred, 0);
      xb_lo = xi.x; xb_hi = xi.y;
      c0blk = (s > 0 && j <= c_hi) ? ld_carry(carry + j) : 0;
    }
    const int x0lo = __builtin_amdgcn_readlane(xb_lo, t & 63);
    const int x0hi = __builtin_amdgcn_readlane(xb_hi, t & 63);
    const int c0 = __builtin_amdgcn_readlane(c0blk, t & 63);
    const int s_up = wave_shr1(S1), g_up = wave_shr1(g1);
    xlo = wave_shr1(xlo); xhi = wave_shr1(xhi);
    if (lane == 0) { xlo = x0lo; xhi = x0hi; }
    const int jj = t - lane;
    const bool incol = jj > c_lo && jj <= c_hi;
    const bool first = c == 0 && jj == 0;
    const int d1 = xlo & 0xFFFF, d2 = (int)((uint32_t)xlo >> 16);
    const int xl = xhi & 0xFF;
    const bool has2 = incol && d2 != 0;
    const bool virt1 = !incol || d1 == 0;
    const int lm1 = max(lane - 1, 0);

    int own1 = col_own, dg1 = col_dg, own2 = 0, dg2 = 0;
    if (!virt1) {
      own1 = ring[((t - d1) & MASK) * 64 + lane];
      dg1 = ring[((t - d1 - 1) & MASK) * 64 + lm1];
    }
    if (has2) {
      own2 = ring[((t - d2) & MASK) * 64 + lane];
      dg2 = ring[((t - d2 - 1) & MASK) * 64 + lm1];
    }
    const int gx1 = cell_tag(own1), gx2 = cell_tag(own2);
    const int cx1 = cell_score(own1) - sc.pen_x(gx1);
    const int cx2 = has2 ? cell_score(own2) - sc.pen_x(gx2) : kNeg;
    const bool px2 = cx2 > cx1;
    const int insX = px2 ? cx2 : cx1, gX = px2 ? gx2 : gx1;
    const int m1 = cell_score(dg1), m2 = has2 ? cell_score(dg2) : kNeg;
    const bool pm2 = m2 > m1;
    int mat = (pm2 ? m2 : m1) + sc.subst(xl, yl);
    int insY = s_up - sc.pen_y(g_up);
    if (rowvirt) { mat = kNeg; insY = kNeg; }

    const bool m = (mat > insY) && (mat > insX);
    const bool xw = !m && (insX > insY);
    int S = m ? mat : (xw ? insX : insY);
    int g = m ? 0 : sc.next_tag(xw ? gX : g_up);
    uint32_t nib = m ? ((pm2 ? kMoveX2 : kMoveX1) | kMoveY) : (xw ? (px2 ? kMoveX2 : kMoveX1) : kMoveY);
    const bool cell = rowok && incol;
    if (!(cell || (rowvirt && incol))) { S = S1; g = g1; }
    if (!cell) nib = 0;
    if (first) { S = cell_score(col_own); g = cell_tag(col_own); }
    if (lane == 0 && s > 0 && (incol || first)) { S = cell_score(c0); g = cell_tag(c0); }
    S1 = S; g1 = g;

    // only live cells enter the ring: a lane's last kTileRing columns must survive the idle steps at the tile's end
    if (incol || first) ring[(t & MASK) * 64 + lane] = pack_cell(S, g);
    __builtin_amdgcn_wave_barrier();

    mvacc |= nib << (4 * (t & 7));
    if ((t & 7) == 7 || t == t_end) {
      if (mvacc) mv[((int64_t)s * tw + (t >> 3)) * 64 + lane] |= mvacc;
      mvacc = 0;
    }
    if (wr_carry && lane == 63 && (incol || first)) st_carry(carry + jj, pack_cell(S, g));
    if (cell && ii == Ly && ((xhi >> 8) & kFlagFinal) && S > best) { best = S; bestx = jj - 1; }
  }
  __syncthreads();
  st_out[lane] = pack_cell(S1, g1);
  for (int k = 0; k < D; ++k) st_out[64 + k * 64 + lane] = ring[k * 64 + lane];
  if (lane == (Ly - 1) % kStripRows + 1) {
    st_out[64 + D * 64] = best; st_out[64 + D * 64 + 1] = bestx;
    if (s == ns - 1 && c == ncb - 1) { a.score2[w] = best; a.bx2[w] = bestx; ta.tiled[w] |= 2; }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}

// ---------------------------------------------------------------- k_fuse1 ---
// One lane per window: trace alignment #1 back (align_lpo_po2.c:108-168) and
// fuse the corrected read into the reference chain (lpo.c:413-463, 602-656).
// Both inputs are linear, so rings are single nodes and the walk needs only the
// running "previous node of each read" to build predecessor lists.

__global__ void __launch_bounds__(64) k_fuse1(BatchArgs a)
{
  const int64_t cnt = list_count(a), stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
    const uint32_t w = a.perm[i];
    const int Lr = (int)(a.off[3 * (int64_t)w + 1] - a.off[3 * (int64_t)w]);
    fuse1_window(a, w, StripMoves{a.moves + a.mv1[w], mv_tw(Lr)});
  }
}

__global__ void __launch_bounds__(64) k_fuse2(BatchArgs a)
{
  const int64_t cnt = list_count(a), stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
    const uint32_t w = a.perm[i];
    fuse2_window(a, w, StripMoves{a.moves + a.mv2[w], mv_tw(a.n1[w])});
  }
}

// ---------------------------------------------------------------- k_left_b ---
// After the fused kernels: collect the windows alignment #2 still has to be done
// for (not handled on chip: graph deeper than the on-chip ring, scores outside the
// 16-bit ring cells, slot estimate exceeded).  Windows the host already routed to
// the generic path have their moves offset; the others get one from a bump
// allocator over a fixed scratch budget.
__global__ void __launch_bounds__(256) k_left_b(BatchArgs a, uint32_t *list, int32_t *count, const uint8_t *done_b,
                                               int64_t *mv2, unsigned long long *bump, unsigned long long bump_base,
                                               unsigned long long bump_cap, int32_t *overflow, int round, int last_round)
{
  // overflow[r] = windows that found no room in round r; a round that follows one without overflow has nothing to do
  if (round > 0 && overflow[round - 1] == 0) return;
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= a.n) return;
  if (a.status[w]) { if (!done_b[w]) a.ncol[w] = 0; return; }
  if (done_b[w]) return;
  if (mv2[w] < 0) {
    const int64_t o2 = a.off[3 * w + 2], o3 = a.off[3 * w + 3];
    const unsigned long long need = (unsigned long long)n_strips((int)(o3 - o2)) * mv_tw(a.n1[w]) * 64;
    const unsigned long long at = atomicAdd(bump, need);
    if (at + need > bump_cap) {
      atomicAdd(overflow + round, 1);
      if (last_round) { a.status[w] = 2; a.ncol[w] = 0; }     // more scratch than kLeftRoundsMax passes provide
      return;                                                  // otherwise: the next round takes it
    }
    mv2[w] = (int64_t)(bump_base + at);
  }
  list[atomicAdd(count, 1)] = (uint32_t)w;
}

// -------------------------------------------------------------- k_trivial ---
// Alignment #1 of a window whose corrected sequence EQUALS its reference (most windows of
// well-corrected reads) needs no dynamic program: with uniform scoring, match >= 0, mismatch <=
// match and positive gap penalties the diagonal is the strictly best predecessor of every diagonal
// cell (a cell left of or above it has at least one gap step less than it has letters), so the
// traceback (align_lpo_po2.c:108-168) pairs letter i with letter i and the fusion (lpo.c:602-656)
// gives the plain chain with every node holding both letters.  8 lanes per window compare the two
// symbol strings and, when equal, write that chain; the window is then marked done for alignment #1.
// The windows that do need alignment #1 get a sort key: the 8-column bucket of the first letter at which
// corrected and reference differ, i.e. where their graph will have its first two-predecessor node.
// Wavefronts of k_fused_b whose windows reach such nodes at the same steps take its two-predecessor
// path together instead of one after the other.
constexpr int kPartBuckets = 32;        // 0..14: first difference in columns 8k..8k+7 (14: beyond); 16..30: the same for the
                                        // one-substitution / one-indel windows k_poa settles without alignment #1 (their wavefronts
                                        // then skip it altogether; 30 also takes the one-letter fillers); 31: corrected equals reference

// first index i < n at which p[i] != q[i] (n: none), by the eight lanes of a window's group together: sixteen bytes of
// each string per lane and round from wherever they start (the hardware takes unaligned addresses; d_sym is padded by 64
// bytes), so a window of up to 128 letters is ONE trip to memory -- the byte-wise loop it replaces made a trip per
// eight letters, each waiting for the one before (0.37 ms per yeast -split batch, most of it waiting)
__device__ __forceinline__ int first_diff8(const uint8_t *p, const uint8_t *q, int n, int g)
{
  int fd = n;
  for (int base = 0; base < n; base += 128) {
    const int i0 = base + 16 * g;
    if (i0 < n) {
      uint4 x, y;
      __builtin_memcpy(&x, p + i0, 16);
      __builtin_memcpy(&y, q + i0, 16);
      const uint32_t t0 = x.x ^ y.x, t1 = x.y ^ y.y, t2 = x.z ^ y.z, t3 = x.w ^ y.w;
      int d = 16;
      if (t3) d = 12 + (__builtin_ctz(t3) >> 3);
      if (t2) d = 8 + (__builtin_ctz(t2) >> 3);
      if (t1) d = 4 + (__builtin_ctz(t1) >> 3);
      if (t0) d = __builtin_ctz(t0) >> 3;
      if (d < 16 && i0 + d < n) fd = i0 + d;
    }
    for (int s = 1; s < 8; s <<= 1) fd = min(fd, __shfl_xor(fd, s, 8));
    if (fd < n) break;
  }
  return fd;
}

// does letter c occur in p[0 .. n)?  (the same way)
__device__ __forceinline__ bool has_letter8(const uint8_t *p, int n, uint8_t c, int g)
{
  const uint32_t cc = 0x01010101u * c;
  bool any = false;
  for (int base = 0; base < n; base += 128) {
    const int i0 = base + 16 * g;
    if (i0 < n) {
      uint4 x;
      __builtin_memcpy(&x, p + i0, 16);
      const uint32_t v[4] = {x.x ^ cc, x.y ^ cc, x.z ^ cc, x.w ^ cc};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t z = ~(((v[k] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v[k]) & 0x80808080u;      // bit 7 of every byte that is zero
        if (z && i0 + 4 * k + (__builtin_ctz(z) >> 3) < n) any = true;
      }
    }
  }
  for (int s = 1; s < 8; s <<= 1) any = __shfl_xor(any ? 1 : 0, s, 8) != 0 || any;
  return any;
}

__global__ void __launch_bounds__(64) k_trivial(BatchArgs a, uint8_t *done_a, uint8_t *triv, uint8_t *pkey, int one_sub_ok)
{
  const int64_t w = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 3);
  const int g = threadIdx.x & 7;
  if (w >= a.n) return;
  const int64_t o0 = a.off[3 * w], o1 = a.off[3 * w + 1], o2 = a.off[3 * w + 2];
  const int Lr = (int)(o1 - o0), Lc = (int)(o2 - o1);
  const uint8_t *xs = a.sym + o0, *ys = a.sym + o1;
  const int nmin = min(Lr, Lc);
  const int fd = first_diff8(xs, ys, nmin, g);              // first index at which the two strings differ
  const bool eq = a.status[w] == 0 && Lr == Lc && fd == nmin;
  if (g == 0) { triv[w] = eq ? 1 : 0; pkey[w] = (uint8_t)(eq ? kPartBuckets - 1 : min(fd >> 3, 14)); }
  const bool flags_only = (one_sub_ok & 2) != 0;      // k_poa builds these graphs itself, in LDS
  const bool indel_ok = (one_sub_ok & 4) != 0;
  const bool filler_ok = (one_sub_ok & 8) != 0;
  one_sub_ok &= 1;
  if (flags_only && eq) return;
  if (flags_only && filler_ok && Lc == 1 && Lr >= 2 && a.status[w] == 0) {
    // The splitter's filler: a corrected window of ONE letter (`N`) that occurs nowhere in the reference window
    // (Master_Splitter.cpp:139-154,268-277).  With the shipped scores (mismatch -10, opening 10, extension 5) cell
    // 0 of the single row is a mismatch (-10), and in every later cell j the three offers tie at -(15 + 5 j): the
    // gap state is shared between the two directions (align_lpo_po2.c:374-407), so the x-gap behind the mismatch
    // opens (-10 - 10) exactly where the diagonal (-(10 + 5 (j - 1)) - 10) and the y-gap from the border row
    // (-(10 + 5 j) - 5) stand; no strict winner means "y-insertion" (:384-407), the traceback from the last cell
    // leaves the matrix at once, the letter stays unaligned and the fusion appends it behind the reference's
    // chain (lpo.c:602-668).  k_poa writes that graph (trivial_graph); no dynamic program of Lr steps for one row.
    const bool any = has_letter8(xs, Lr, ys[0], g);
    if (!any && g == 0) { triv[w] = 5; pkey[w] = (uint8_t)(16 + 14); }
    if (!any) return;
  }
  if (!eq) {
    // One substitution and nothing else (same length, the strings agree after position fd): the diagonal
    // with its one mismatch beats every alignment with gaps (two gap openings at least) and is the
    // strictly best predecessor of every diagonal cell as long as |mismatch| < open + extension, so the
    // pairs are again letter i with letter i; the fusion puts the corrected letter in a node of its own
    // right before the reference letter's, both in one ring (lpo.c:449-450,647-649).  No dynamic program.
    if (!one_sub_ok || a.status[w] != 0) return;
    if (Lr != Lc) {
      // One inserted or one deleted letter and nothing else (the strings agree up to fd and, shifted by one,
      // after it): one gap of length 1 and no mismatch beats every other alignment (mismatch < match, and any
      // other gap layout opens at least two more gap positions); the gap may sit anywhere in the run of equal
      // letters that ends at fd, and the traceback's tie rule (a gap step wins against the diagonal on the way
      // back from the end, align_lpo_po2.c:108-168) puts it at the run's last letter, fd itself.  k_poa builds
      // the graph (poa_pack.hip trivial_graph); the other paths run the dynamic program.
      if (!flags_only || !indel_ok || nmin < 1 || (Lc != Lr - 1 && Lc != Lr + 1)) return;
      const bool del = Lc == Lr - 1;
      const bool rest = del ? first_diff8(xs + fd + 1, ys + fd, Lc - fd, g) == Lc - fd
                            : first_diff8(xs + fd, ys + fd + 1, Lr - fd, g) == Lr - fd;
      if (rest && g == 0) { triv[w] = del ? 3 : 4; pkey[w] = (uint8_t)(16 + min(fd >> 3, 14)); }
      return;
    }
    if (first_diff8(xs + fd + 1, ys + fd + 1, Lr - fd - 1, g) != Lr - fd - 1) return;
    if (flags_only) { if (g == 0) { triv[w] = 2; pkey[w] = (uint8_t)(16 + min(fd >> 3, 14)); } return; }
    const int e = fd, L = Lr;
    const int64_t nb = o0 + w;
    for (int i = g; i < L; i += 8) {
      const int fl_pos = (i == 0 ? kFlagInitial : 0) | (i == L - 1 ? kFlagFinal : 0);
      if (i != e) {
        const int n = i < e ? i : i + 1;
        const int d1 = i == 0 ? 0 : 1, d2 = i == e + 1 ? 2 : 0;          // after the bubble: the reference letter's node, then the corrected letter's
        a.xinfo[nb + n + 1] = make_int2(d1 | (d2 << 16), xs[i] | ((kFlagHasRef | kFlagHasCor | fl_pos) << 8));
        a.ring1[nb + n] = (uint16_t)n;
      } else {
        a.xinfo[nb + e + 1] = make_int2(e == 0 ? 0 : 1, ys[e] | ((kFlagHasCor | fl_pos) << 8));
        a.ring1[nb + e] = (uint16_t)e;
        a.xinfo[nb + e + 2] = make_int2(e == 0 ? 0 : 2, xs[e] | ((kFlagHasRef | fl_pos) << 8));
        a.ring1[nb + e + 1] = (uint16_t)e;
      }
    }
    if (g == 0) {
      a.n1[w] = L + 1;
      a.cls[w] = 0;                     // max predecessor distance 2: ring need 4
      a.score1[w] = (L - 1) * a.kp.match + a.kp.mismatch;
      done_a[w] = 1;
    }
    return;
  }
  const int64_t nb = o0 + w;
  for (int n = g; n < Lr; n += 8) {
    const int fl = kFlagHasRef | kFlagHasCor | (n == 0 ? kFlagInitial : 0) | (n == Lr - 1 ? kFlagFinal : 0);
    a.xinfo[nb + n + 1] = make_int2(n == 0 ? 0 : 1, xs[n] | (fl << 8));     // d1 = 1 (0: virtual start), no second predecessor
    a.ring1[nb + n] = (uint16_t)n;
  }
  if (g == 0) {
    a.n1[w] = Lr;
    a.cls[w] = 0;                       // max predecessor distance 1
    a.score1[w] = Lr * a.kp.match;
    done_a[w] = 1;
  }
}

// Stable bucket sort of every launch bin's window list by pkey (so: windows that still need alignment
// #1 first, ordered by where their first bubble will be; the trivial ones after them); count[bin] =
// how many need alignment #1.  Three small launches over chunks of kPartChunk list entries: per-chunk
// bucket counts, a scan over each bin's (bucket, chunk) pairs, the scatter.
// chunk table (host-built): first list index, length, bin slot
struct PartChunk { int64_t first; int32_t len, bin; };

__global__ void __launch_bounds__(256) k_part_count(const uint32_t *__restrict__ in, const PartChunk *__restrict__ chunks,
                                                    const uint8_t *__restrict__ pkey, int32_t *__restrict__ chunk_cnt)
{
  __shared__ int s_cnt[kPartBuckets];
  const PartChunk ch = chunks[blockIdx.x];
  if (threadIdx.x < kPartBuckets) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < ch.len; i += 256) atomicAdd(&s_cnt[pkey[in[ch.first + i]]], 1);
  __syncthreads();
  if (threadIdx.x < kPartBuckets) chunk_cnt[blockIdx.x * kPartBuckets + threadIdx.x] = s_cnt[threadIdx.x];
}

// one wave per bin: exclusive scan over its (bucket-major, chunk-minor) counts, in place;
// the windows in front of the last bucket are the ones that need alignment #1
__global__ void __launch_bounds__(64) k_part_scan(const int32_t *__restrict__ bin_chunks /* first chunk, #chunks per bin */,
                                                  int32_t *__restrict__ chunk_cnt, int32_t *__restrict__ count,
                                                  int32_t *__restrict__ count_a1)
{
  const int c0 = bin_chunks[2 * blockIdx.x], nc = bin_chunks[2 * blockIdx.x + 1], lane = threadIdx.x;
  int run = 0;
  for (int k = 0; k < kPartBuckets; ++k) {
    if (k == kPartBuckets - 1 && lane == 0) count[blockIdx.x] = run;
    if (k == 16 && lane == 0) count_a1[blockIdx.x] = run;      // buckets 0..15: no shortcut graph, alignment #1 runs
    for (int base = 0; base < nc; base += 64) {
      const int i = base + lane;
      const int v = i < nc ? chunk_cnt[(c0 + i) * kPartBuckets + k] : 0;
      int inc = v;
      for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
      if (i < nc) chunk_cnt[(c0 + i) * kPartBuckets + k] = run + inc - v;
      run += __shfl(inc, 63);
    }
  }
}

__global__ void __launch_bounds__(256) k_part_scatter(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                      const PartChunk *__restrict__ chunks, const int64_t *__restrict__ bins,
                                                      const uint8_t *__restrict__ pkey, const int32_t *__restrict__ chunk_base)
{
  __shared__ int s_wave[4][kPartBuckets], s_run[kPartBuckets];
  const PartChunk ch = chunks[blockIdx.x];
  const int64_t bin_first = bins[2 * ch.bin];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid < kPartBuckets) s_run[tid] = chunk_base[blockIdx.x * kPartBuckets + tid];
  __syncthreads();
  for (int i0 = 0; i0 < ch.len; i0 += 256) {
    const int i = i0 + tid;
    const bool in_range = i < ch.len;
    const uint32_t w = in_range ? in[ch.first + i] : 0u;
    const int key = in_range ? (int)pkey[w] : -1;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    int rank = 0;
    for (int k = 0; k < kPartBuckets; ++k) {
      const unsigned long long bm = __builtin_amdgcn_ballot_w64(key == k);
      if (lane == 0) s_wave[wave][k] = __builtin_popcountll(bm);
      if (key == k) rank = __builtin_popcountll(bm & below);
    }
    __syncthreads();
    if (in_range) {
      int pos = s_run[key] + rank;
      for (int q = 0; q < wave; ++q) pos += s_wave[q][key];
      out[bin_first + pos] = w;
    }
    __syncthreads();
    if (tid < kPartBuckets) s_run[tid] += s_wave[0][tid] + s_wave[1][tid] + s_wave[2][tid] + s_wave[3][tid];
    __syncthreads();
  }
}

// ----------------------------------------------------------------- k_rows ---
// column-interleaved MSA -> three contiguous rows per window (host-buffer API)

__global__ void __launch_bounds__(256) k_rows(const uint8_t *__restrict__ cols, const int64_t *__restrict__ off,
                                               const int32_t *__restrict__ ncol, const int64_t *__restrict__ row_off,
                                               uint8_t *__restrict__ rows, int64_t n)
{
  const int64_t w = blockIdx.x;
  if (w >= n) return;
  const int nc = ncol[w];
  const uint8_t *src = cols + 3 * off[3 * w];
  uint8_t *dst = rows + row_off[w];
  for (int i = threadIdx.x; i < 3 * nc; i += blockDim.x) {
    const int r = i / nc, c = i - r * nc;
    dst[i] = src[3 * c + r];
  }
}

// ---------------------------------------------------------------- launchers ---

void launch_symbolize(const uint8_t *in, uint8_t *out, int64_t nbytes, const DevTables *tab, hipStream_t st)
{
  if (nbytes <= 0) return;
  int64_t blocks = (nbytes / 16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_symbolize, dim3((unsigned)blocks), dim3(256), 0, st, in, out, nbytes, tab);
}

// grid for a list launch: the exact count when the host knows it, a fixed grid of
// looping blocks when the count lives on the device
static unsigned list_grid(const BatchArgs &a, int64_t per_block, unsigned device_grid)
{
  if (a.count_ptr) return device_grid;
  int64_t g = (a.n + per_block - 1) / per_block;
  return (unsigned)(g < 1 ? 1 : g);
}

// nw > 1: kDp2Waves wavefronts per window (a short list of long windows: the longest one sets the time)
void launch_dp1(const BatchArgs &a, bool gen, hipStream_t st, int nw)
{
  const unsigned g = list_grid(a, 1, 4096);
  if (gen) hipLaunchKernelGGL((k_dp1<true, 1>), dim3(g), dim3(64), 0, st, a);
  else if (nw >= 16) hipLaunchKernelGGL((k_dp1<false, 16>), dim3(g), dim3(64 * 16), 0, st, a);
  else if (nw > 1) hipLaunchKernelGGL((k_dp1<false, 8>), dim3(g), dim3(64 * 8), 0, st, a);
  else hipLaunchKernelGGL((k_dp1<false, 1>), dim3(g), dim3(64), 0, st, a);
}

// one launch per anti-diagonal of tiles; ntiles = upper bound of the tiles on it, nw = long windows
void launch_dp1_tile(const BatchArgs &a, const TileArgs &ta, bool gen, int ntiles, int nw, hipStream_t st)
{
  if (gen) hipLaunchKernelGGL(k_dp1_tile<true>, dim3((unsigned)ntiles, (unsigned)nw), dim3(64), 0, st, a, ta);
  else hipLaunchKernelGGL(k_dp1_tile<false>, dim3((unsigned)ntiles, (unsigned)nw), dim3(64), 0, st, a, ta);
}

void launch_dp2_tile(const BatchArgs &a, const TileArgs &ta, bool gen, int ntiles, int nw, hipStream_t st)
{
  if (gen) hipLaunchKernelGGL(k_dp2_tile<true>, dim3((unsigned)ntiles, (unsigned)nw), dim3(64), 0, st, a, ta);
  else hipLaunchKernelGGL(k_dp2_tile<false>, dim3((unsigned)ntiles, (unsigned)nw), dim3(64), 0, st, a, ta);
}

void launch_fuse1(const BatchArgs &a, hipStream_t st)
{
  hipLaunchKernelGGL(k_fuse1, dim3(list_grid(a, 64, 256)), dim3(64), 0, st, a);
}

void launch_left_b(const BatchArgs &a, uint32_t *list, int32_t *count, const uint8_t *done_b, int64_t *mv2,
                   unsigned long long *bump, unsigned long long bump_base, unsigned long long bump_cap, int round,
                   int last_round, hipStream_t st)
{
  if (a.n <= 0) return;
  hipLaunchKernelGGL(k_left_b, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, st, a, list, count, done_b, mv2, bump,
                     bump_base, bump_cap, count + 8, round, last_round);
}

// cls 0: every predecessor within 30 nodes (LDS ring of 32 steps); cls 1: deeper graphs, the DEEP variant
// with its HBM shadow ring -- gring holds `blocks` regions of gring_block ints, one per block of the launch
// nw > 1: kDp2Waves wavefronts per window (the strips side by side), for the short lists whose longest window sets the time
constexpr int kDp2Waves = 8;
void launch_dp2(const BatchArgs &a, bool gen, int cls, hipStream_t st, int32_t *gring, int64_t gring_block, int blocks, int nw)
{
  constexpr int NW = kDp2Waves;
  const size_t lds1 = 32 * 256, ldsN = (size_t)NW * 32 * 256 + NW * 4;
  if (cls == 0) {
    const dim3 g(list_grid(a, 1, 4096));
    if (gen) hipLaunchKernelGGL((k_dp2<true, 32, false, 1>), g, dim3(64), lds1, st, a, cls, nullptr, (int64_t)0);
    else if (nw > 1) hipLaunchKernelGGL((k_dp2<false, 32, false, NW>), g, dim3(64 * NW), ldsN, st, a, cls, nullptr, (int64_t)0);
    else hipLaunchKernelGGL((k_dp2<false, 32, false, 1>), g, dim3(64), lds1, st, a, cls, nullptr, (int64_t)0);
    return;
  }
  if (!gring || blocks <= 0) return;                  // no window of the batch can be that deep
  // (`blocks` regions of gring_block ints: a block of NW wavefronts takes NW of them)
  if (gen) hipLaunchKernelGGL((k_dp2<true, 32, true, 1>), dim3((unsigned)blocks), dim3(64), lds1, st, a, cls, gring, gring_block);
  else if (nw > 1 && blocks >= NW)
    hipLaunchKernelGGL((k_dp2<false, 32, true, NW>), dim3((unsigned)(blocks / NW)), dim3(64 * NW), ldsN, st, a, cls, gring, gring_block);
  else hipLaunchKernelGGL((k_dp2<false, 32, true, 1>), dim3((unsigned)blocks), dim3(64), lds1, st, a, cls, gring, gring_block);
}

// Both classes of a short list in ONE launch of the DEEP instance (cls -1: it takes the near predecessors from the LDS ring
// like the plain one; its shadow stores are fire and forget): the launch lasts as long as its longest window, not as the
// longest of each class one after the other.  Where that instance cannot run, the two launches.
void launch_dp2_list(const BatchArgs &a, bool gen, hipStream_t st, int32_t *gring, int64_t gring_block, int blocks, int nw)
{
  constexpr int NW = kDp2Waves;
  if (!gen && gring && nw >= 16 && blocks >= 16) {
    // (the host-routed windows: longer than any on-chip class takes, a dozen strips and more)
    static bool attr_set = false;
    if (!attr_set) {
      attr_set = true;
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dp2<false, 32, true, 16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                16 * 32 * 256 + 64);
    }
    hipLaunchKernelGGL((k_dp2<false, 32, true, 16>), dim3((unsigned)(blocks / 16)), dim3(64 * 16), (size_t)16 * 32 * 256 + 64, st,
                       a, -1, gring, gring_block);
    return;
  }
  if (!gen && gring && blocks >= NW) {
    hipLaunchKernelGGL((k_dp2<false, 32, true, NW>), dim3((unsigned)(blocks / NW)), dim3(64 * NW), (size_t)NW * 32 * 256 + NW * 4, st,
                       a, -1, gring, gring_block);
    return;
  }
  for (int cls = 0; cls < 2; ++cls) launch_dp2(a, gen, cls, st, gring, gring_block, blocks, NW);
}

void launch_fuse2(const BatchArgs &a, hipStream_t st)
{
  hipLaunchKernelGGL(k_fuse2, dim3(list_grid(a, 64, 256)), dim3(64), 0, st, a);
}

void launch_trivial(const BatchArgs &a, uint8_t *done_a, uint8_t *triv, uint8_t *pkey, bool flags_only, hipStream_t st)
{
  if (a.n <= 0) return;
  // the one-substitution shortcut needs |mismatch - match| < gap opening + extension in both directions
  const int one_sub_ok = !std::getenv("ELECTOR_NO_ONESUB") &&
                         (a.kp.match - a.kp.mismatch) < (a.kp.open_x < a.kp.open_y ? a.kp.open_x : a.kp.open_y) +
                                                         (a.kp.ext_x < a.kp.ext_y ? a.kp.ext_x : a.kp.ext_y);
  // the one-insertion / one-deletion shortcut: a mismatch must cost something, gap penalties alike in both directions
  const int indel_ok = !std::getenv("ELECTOR_NO_ONEINDEL") && a.kp.mismatch < a.kp.match && a.kp.open_x == a.kp.open_y &&
                       a.kp.ext_x == a.kp.ext_y;
  // the one-letter filler shortcut is derived for the shipped numbers only
  const int filler_ok = !std::getenv("ELECTOR_NO_FILLER") && a.kp.match == 0 && a.kp.mismatch == -10 && a.kp.open_x == 10 &&
                        a.kp.open_y == 10 && a.kp.ext_x == 5 && a.kp.ext_y == 5;
  hipLaunchKernelGGL(k_trivial, dim3((unsigned)((a.n + 7) / 8)), dim3(64), 0, st, a, done_a, triv, pkey,
                     one_sub_ok | (flags_only ? 2 : 0) | (indel_ok ? 4 : 0) | (filler_ok ? 8 : 0));
}

int partition_buckets() { return kPartBuckets; }

void launch_partition(const uint32_t *in, uint32_t *out, const int64_t *bins, int nbins, const void *chunks, int nchunks,
                      const int32_t *bin_chunks, const uint8_t *pkey, int32_t *chunk_cnt, int32_t *count, int32_t *count_a1,
                      hipStream_t st)
{
  if (nbins <= 0 || nchunks <= 0) return;
  const PartChunk *ch = reinterpret_cast<const PartChunk *>(chunks);
  hipLaunchKernelGGL(k_part_count, dim3((unsigned)nchunks), dim3(256), 0, st, in, ch, pkey, chunk_cnt);
  hipLaunchKernelGGL(k_part_scan, dim3((unsigned)nbins), dim3(64), 0, st, bin_chunks, chunk_cnt, count, count_a1);
  hipLaunchKernelGGL(k_part_scatter, dim3((unsigned)nchunks), dim3(256), 0, st, in, out, ch, bins, pkey, chunk_cnt);
}

void launch_rows(const uint8_t *cols, const int64_t *off, const int32_t *ncol, const int64_t *row_off,
                 uint8_t *rows, int64_t n, hipStream_t st)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_rows, dim3((unsigned)n), dim3(256), 0, st, cols, off, ncol, row_off, rows, n);
}

}  // namespace elector
