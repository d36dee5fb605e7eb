// elector_amd/csrc/poa_lane.hip -- lane-per-window kernels for the bulk of the
// windows (the splitter's windows are mostly 30-130 bases).
//
// One lane = one window, 64 windows per wavefront, windows of a wave sorted to
// near-equal size.  Every lane runs the plain row-by-row DP of the reference
// (align_lpo_po2.c:309-418) on its own window; what makes it fast is the layout:
//
//   * the DP row lives in LDS as R[column][lane] (32-bit cells: score << 1 |
//     came-from-match): bank = lane, so all 64 lanes read/write their own cell of
//     the same column index without conflicts, whatever column each lane is at;
//   * the row is updated in place; the cell it overwrites (row above, same column)
//     is pushed into a 16-deep per-lane history H[column & 15][lane], which is what
//     a PO predecessor up to 14 nodes back needs for the "match" term;
//   * moves are packed 8 columns x 4 bits per lane and stored as one coalesced
//     256-byte row per 8 columns to a per-wave scratch tile in HBM (L2-resident);
//   * no anti-diagonal skew, no idle lanes: traceback, fusion and column emission
//     (poa_serial.h) run with all 64 lanes busy, each on its own window.
//
//   k_lane_a  alignment #1 (reference x corrected) + traceback + fusion #1
//   k_lane_b  alignment #2 (PO graph x uncorrected) + traceback + fusion #2 + columns
//
// Uniform-scoring parameters only; windows whose graph does not fit the launch's
// LDS capacity or whose predecessors reach back more than 14 nodes are left to the
// other kernel families (done flags).
#include <hip/hip_runtime.h>
#include "poa_device.h"
#include "poa_serial.h"

namespace elector {

struct LaneArgs {
  BatchArgs b;
  const uint32_t *list;      // window ids, 64 consecutive entries per wave
  int64_t nlist;
  const int64_t *mv_off;     // per wave: dword offset of its moves tile in b.moves
  const int32_t *dims;       // per wave: {max Lr, max Lc, max Lu, max (Lr + Lc)}
  int cap;                   // LDS capacity of this launch, in columns / nodes
  uint8_t *done_a;
  uint8_t *done_b;
};

// moves of a lane-per-window tile: dword ((row-1) * tw + (col-1)/8) * 64 + lane
struct LaneMoves {
  const uint32_t *mv;
  int tw, lane;
  __device__ __forceinline__ uint32_t operator()(int ii, int jj) const
  {
    return (mv[((int64_t)(ii - 1) * tw + ((jj - 1) >> 3)) * 64 + lane] >> (4 * ((jj - 1) & 7))) & 15u;
  }
};

__device__ __forceinline__ int lcell(int s, bool m) { return (s << 1) | (m ? 1 : 0); }

// ----------------------------------------------------------------- k_lane_a ---

__global__ void __launch_bounds__(64) k_lane_a(LaneArgs a)
{
  extern __shared__ __align__(16) uint32_t lds[];
  const int lane = threadIdx.x;
  const KParams kp = a.b.kp;
  const int64_t li = 64 * (int64_t)blockIdx.x + lane;
  bool valid = li < a.nlist;
  const uint32_t w = valid ? a.list[li] : 0;
  valid = valid && a.b.status[w] == 0;
  int64_t o0 = 0;
  int Lr = 0, Lc = 0;
  if (valid) {
    o0 = a.b.off[3 * (int64_t)w];
    Lr = (int)(a.b.off[3 * (int64_t)w + 1] - o0);
    Lc = (int)(a.b.off[3 * (int64_t)w + 2] - o0) - Lr;
  }
  valid = valid && Lr <= a.cap;
  if (!valid) { Lr = 0; Lc = 0; }
  const int LrM = a.dims[4 * blockIdx.x], LcM = a.dims[4 * blockIdx.x + 1];
  const int cap = a.cap;
  uint32_t *R = lds;                                   // [(cap + 1)][64]
  uint8_t *xs = reinterpret_cast<uint8_t *>(lds + (cap + 1) * 64);   // [cap][64]
  const uint8_t *sym = a.b.sym + o0;
  uint32_t *mv = a.b.moves + a.mv_off[blockIdx.x];
  const int tw = (LrM + 7) >> 3;

  for (int j = 0; j < LrM && j < cap; ++j) xs[j * 64 + lane] = (j < Lr) ? sym[j] : 0;
  // virtual row -1: jj gap steps along x from the origin (align_lpo_po2.c:272-286)
  R[lane] = lcell(0, true);
  for (int jj = 1; jj <= LrM && jj <= cap; ++jj) R[jj * 64 + lane] = lcell(-(kp.open_x + (jj - 1) * kp.ext_x), false);
  __syncthreads();

  int score = kNeg;
  int ynext = (Lc >= 1) ? sym[Lr] : 255;
  for (int ii = 1; ii <= LcM; ++ii) {
    const bool rowact = ii <= Lc;
    const int yl = ynext;
    ynext = (ii < Lc) ? sym[Lr + ii] : 255;
    // column -1 of this row and of the row above (:290-302)
    int leftS = -(kp.open_y + (ii - 1) * kp.ext_y);
    bool leftM = false;
    int diagS = (ii == 1) ? 0 : -(kp.open_y + (ii - 2) * kp.ext_y);
    uint32_t mvacc = 0;
    for (int jj = 1; jj <= LrM; ++jj) {
      const bool act = rowact && jj <= Lr;
      const int old = (int)R[jj * 64 + lane];
      const int xl = xs[(jj - 1) * 64 + lane];
      const int upS = old >> 1;
      const int insY = upS - ((old & 1) ? kp.open_y : kp.ext_y);
      const int insX = leftS - (leftM ? kp.open_x : kp.ext_x);
      const int mat = diagS + (xl == yl ? kp.match : kp.mismatch);
      const int mx = max(insX, insY);
      const bool m = mat > mx;                                         // :384
      const int S = max(mat, mx);
      const uint32_t nib = m ? (kMoveX1 | kMoveY) : (insX > insY ? kMoveX1 : kMoveY);   // :392
      diagS = upS;
      if (act) {
        R[jj * 64 + lane] = (uint32_t)lcell(S, m);
        leftS = S; leftM = m;
        mvacc |= nib << (4 * ((jj - 1) & 7));
      }
      if ((jj & 7) == 0 || jj == LrM) { mv[((int64_t)(ii - 1) * tw + ((jj - 1) >> 3)) * 64 + lane] = mvacc; mvacc = 0; }
    }
    if (rowact && ii == Lc) score = leftS;                             // the only FINAL x FINAL cell
  }
  // make this wave's moves visible to its own scattered reads below
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (valid) {
    a.b.score1[w] = score;
    fuse1_window(a.b, w, LaneMoves{mv, tw, lane});
    a.done_a[w] = 1;
  }
}

// ----------------------------------------------------------------- k_lane_b ---

__global__ void __launch_bounds__(64) k_lane_b(LaneArgs a)
{
  extern __shared__ __align__(16) uint32_t lds[];
  const int lane = threadIdx.x;
  const KParams kp = a.b.kp;
  const int64_t li = 64 * (int64_t)blockIdx.x + lane;
  bool valid = li < a.nlist;
  const uint32_t w = valid ? a.list[li] : 0;
  valid = valid && a.b.status[w] == 0 && a.done_a[w] != 0 && (a.b.cls[w] & 0x83) == 0;   // cls bit 7: deeper than the ring
  int64_t o0 = 0, o2 = 0;
  int n1 = 0, Lu = 0;
  if (valid) {
    o0 = a.b.off[3 * (int64_t)w];
    o2 = a.b.off[3 * (int64_t)w + 2];
    Lu = (int)(a.b.off[3 * (int64_t)w + 3] - o2);
    n1 = a.b.n1[w];
  }
  const int cap = a.cap;
  const int tw = (a.dims[4 * blockIdx.x + 3] + 7) >> 3;          // tile width from the host's bound on |PO|
  valid = valid && n1 <= cap && n1 <= 8 * tw;
  if (!valid) { n1 = 0; Lu = 0; }
  int NM = n1, LuM = Lu;
  for (int d = 1; d < 64; d <<= 1) { NM = max(NM, __shfl_xor(NM, d)); LuM = max(LuM, __shfl_xor(LuM, d)); }
  NM = __builtin_amdgcn_readfirstlane(NM);
  LuM = __builtin_amdgcn_readfirstlane(LuM);
  uint32_t *I = lds;                                   // [(cap + 1)][64] node info
  uint32_t *R = lds + (cap + 1) * 64;                  // [(cap + 1)][64] DP row
  uint32_t *H = lds + 2 * (cap + 1) * 64;              // [16][64] cells of the row above, by column & 15
  const uint8_t *ys = a.b.sym + o2;
  const int2 *xinfo = a.b.xinfo + (o0 + w);
  uint32_t *mv = a.b.moves + a.mv_off[blockIdx.x];

  // node info packed to one dword: d1 (0 = virtual start) | d2 << 8 (0 = none, 255 = virtual)
  // | letter << 16 | final flag << 24; virtual row -1 over the graph (align_lpo_po2.c:275-286)
  R[lane] = lcell(0, true);
  for (int jj = 1; jj <= NM; ++jj) {
    uint32_t inf = 0;
    int r = 0;
    if (jj <= n1) {
      const int2 xi = xinfo[jj];
      const int d1 = xi.x & 0xFFFF, d2 = (int)((uint32_t)xi.x >> 16);
      const int pp1 = d1 ? jj - d1 : 0, pp2 = d2 ? jj - d2 : 0;
      inf = (uint32_t)d1 | ((uint32_t)d2 << 8) | ((uint32_t)(xi.y & 0xFF) << 16) |
            ((uint32_t)((xi.y >> 8) & kFlagFinal) << 24);
      r = ((int)R[pp1 * 64 + lane] >> 1) - (pp1 == 0 ? kp.open_x : kp.ext_x);
      if (d2) {
        const int p2 = d2 == 255 ? 0 : pp2;
        r = max(r, ((int)R[p2 * 64 + lane] >> 1) - (p2 == 0 ? kp.open_x : kp.ext_x));
      }
    }
    I[jj * 64 + lane] = inf;
    R[jj * 64 + lane] = (uint32_t)lcell(r, false);
  }
  __syncthreads();

  int best = kNeg, bestx = -1;
  int ynext = (Lu >= 1) ? ys[0] : 255;
  for (int ii = 1; ii <= LuM; ++ii) {
    const bool rowact = ii <= Lu;
    const int yl = ynext;
    ynext = (ii < Lu) ? ys[ii] : 255;
    const int colCur = -(kp.open_y + (ii - 1) * kp.ext_y);
    const int colPrev = (ii == 1) ? 0 : -(kp.open_y + (ii - 2) * kp.ext_y);
    int leftS = colCur, prevOld = colPrev;
    bool leftM = false;
    uint32_t mvacc = 0;
    for (int jj = 1; jj <= NM; ++jj) {
      const bool act = rowact && jj <= n1;
      const uint32_t inf = I[jj * 64 + lane];
      const int old = (int)R[jj * 64 + lane];
      const int d1 = inf & 255, d2 = (inf >> 8) & 255, xl = (inf >> 16) & 255;
      const int upS = old >> 1;
      // predecessor 1: the previous node by default (registers); virtual start or a node further back otherwise
      int o1S = leftS, dg1 = prevOld;
      bool o1M = leftM;
      if (act && d1 != 1) {
        if (d1 == 0) { o1S = colCur; o1M = false; dg1 = colPrev; }
        else {
          const int p = jj - d1;
          const int c = (int)R[p * 64 + lane];
          o1S = c >> 1; o1M = (c & 1) != 0;
          dg1 = (int)H[(p & 15) * 64 + lane] >> 1;
        }
      }
      int o2S = kNeg, dg2 = kNeg;
      bool o2M = false;
      if (act && d2 != 0) {
        if (d2 == 255) { o2S = colCur; dg2 = colPrev; }
        else {
          const int p = jj - d2;
          const int c = (int)R[p * 64 + lane];
          o2S = c >> 1; o2M = (c & 1) != 0;
          dg2 = (int)H[(p & 15) * 64 + lane] >> 1;
        }
      }
      const int cx1 = o1S - (o1M ? kp.open_x : kp.ext_x);
      const int cx2 = o2S - (o2M ? kp.open_x : kp.ext_x);
      const bool px2 = cx2 > cx1;                                       // first maximum wins (:361-371)
      const int insX = max(cx1, cx2);
      const bool pm2 = dg2 > dg1;                                       // (:348-357)
      const int mat = max(dg1, dg2) + (xl == yl ? kp.match : kp.mismatch);
      const int insY = upS - ((old & 1) ? kp.open_y : kp.ext_y);
      const int mx = max(insX, insY);
      const bool m = mat > mx;
      const int S = max(mat, mx);
      const uint32_t second = (m ? pm2 : px2) ? 1u : 0u;
      const uint32_t nib = m ? (kMoveX1 | kMoveY) + second : (insX > insY ? kMoveX1 + second : (uint32_t)kMoveY);
      prevOld = upS;
      if (act) {
        H[(jj & 15) * 64 + lane] = (uint32_t)old;
        R[jj * 64 + lane] = (uint32_t)lcell(S, m);
        leftS = S; leftM = m;
        mvacc |= nib << (4 * ((jj - 1) & 7));
        if (ii == Lu && (inf >> 24) && S > best) { best = S; bestx = jj - 1; }   // ties keep the smaller column
      }
      if ((jj & 7) == 0 || jj == NM) { mv[((int64_t)(ii - 1) * tw + ((jj - 1) >> 3)) * 64 + lane] = mvacc; mvacc = 0; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (valid) {
    a.b.score2[w] = best;
    a.b.bx2[w] = bestx;
    fuse2_window(a.b, w, LaneMoves{mv, tw, lane});
    a.done_b[w] = 1;
  }
}

// ---------------------------------------------------------------- launchers ---

static int lane_lds_a(int cap) { return (cap + 1) * 256 + cap * 64; }
static int lane_lds_b(int cap) { return (2 * (cap + 1) + 16) * 256; }

int launch_lane_a(const LaneArgs &a, hipStream_t st)
{
  if (a.nlist <= 0) return 0;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_lane_a), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024 - 256) != hipSuccess)
      return -1;
    attr = true;
  }
  hipLaunchKernelGGL(k_lane_a, dim3((unsigned)((a.nlist + 63) / 64)), dim3(64), lane_lds_a(a.cap), st, a);
  return 0;
}

int launch_lane_b(const LaneArgs &a, hipStream_t st)
{
  if (a.nlist <= 0) return 0;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_lane_b), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024 - 256) != hipSuccess)
      return -1;
    attr = true;
  }
  hipLaunchKernelGGL(k_lane_b, dim3((unsigned)((a.nlist + 63) / 64)), dim3(64), lane_lds_b(a.cap), st, a);
  return 0;
}

}  // namespace elector
