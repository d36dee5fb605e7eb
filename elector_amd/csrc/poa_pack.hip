// elector_amd/csrc/poa_pack.hip -- the whole window in ONE kernel, two windows per lane group, packed
// 16-bit arithmetic: k_poa<G,R>.
//
//   symbols -> [alignment #1 -> traceback -> fusion #1] -> alignment #2 -> traceback -> fusion #2 -> MSA columns
//
// What it changes against the k_fused_a / k_fused_b pair of poa_fused.hip (which stays as the path for the
// windows this kernel hands back):
//   * the graph after fusion #1 never leaves LDS (no 10 B/node round trip through HBM, one launch per
//     geometry class instead of two);
//   * a group of G lanes works on TWO windows at once: every score register holds window A's value in its low
//     half and window B's in its high half, and the recurrence runs on v_pk_max_i16 / v_pk_add_i16 /
//     v_pk_sub_i16 / v_pk_mad_i16 / v_pk_min_u16 -- one instruction, two cells.  Scores fit 16 bits for the
//     windows taken here (score_span() of poa_device.h < 16000);
//   * alignment #2 needs no score ring in LDS: a window is taken when every predecessor of its graph lies at
//     most two nodes back (99.8 % of the windows of well-corrected reads), so the predecessor columns are the
//     two columns the lane computed last, held in registers; which of the two, per window, is a bit-field
//     select (v_bfi_b32) under a mask built from the node's record;
//   * the windows whose corrected sequence equals the reference, or differs by one substitution, get their
//     graph written straight into LDS (k_trivial only classifies and sorts now).
// Windows that do not qualify (deeper graphs, more rows than one strip holds, scores beyond 16 bits, general
// or asymmetric scoring parameters, LDS slot too small) are appended to a device list and go through
// k_fused_a / k_fused_b / the generic kernels afterwards.
//
// Geometry as in poa_fused.hip: lane g of a group holds R consecutive rows of the linear read y and computes
// column jj = t - g at anti-diagonal step t; the row above a lane's block arrives by DPP.  One strip only.
// Reference behaviour restated: align_lpo_po2.c:178-433 (DP, tie-breaks), :108-168 (traceback),
// lpo.c:413-463,602-656 (fusion), lpo_format.c:337-393 (rows).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "poa_device.h"

namespace elector {


// ------------------------------------------------------------ packed helpers ---
// two 16-bit lanes per register: window A low, window B high.  Inline asm where hipcc would otherwise
// unpack the halves (it turns min(x, 1) and multiply-add on short2 into per-half compares and selects).
__device__ __forceinline__ uint32_t pk2(int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }
__device__ __forceinline__ uint32_t pk1(int v) { return pk2(v, v); }
__device__ __forceinline__ int pk_half(uint32_t v, int h) { return h ? (int)((int32_t)v >> 16) : (int)(int16_t)(v & 0xFFFFu); }

__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b)
{
  uint32_t d;
  asm("v_pk_max_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)
{
  uint32_t d;
  asm("v_pk_add_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b)
{
  uint32_t d;
  asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// a * K + c per half, K wave-uniform (an SGPR operand: no copy into a VGPR)
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t K, uint32_t c)
{
  uint32_t d;
  asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(K), "v"(c));
  return d;
}
// a - K per half, K wave-uniform
__device__ __forceinline__ uint32_t pk_subk(uint32_t a, uint32_t K)
{
  uint32_t d;
  asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(K));
  return d;
}
// min(x, 1) per half, x >= 0: the "differs" / "is greater" bit
__device__ __forceinline__ uint32_t pk_bit(uint32_t x, uint32_t ones)
{
  uint32_t d;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(x), "s"(ones));
  return d;
}
// (m & a) | (~m & b)
__device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t a, uint32_t b)
{
  uint32_t d;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(m), "v"(a), "v"(b));
  return d;
}
// keep an accumulator materialised here (the compiler otherwise postpones the OR chain and holds every
// pair's bits in a register of its own)
__device__ __forceinline__ void pin(uint32_t &x) { asm volatile("" : "+v"(x)); }

// One row (two cells, one per half) of the affine-gap recurrence in ONE block.  The compiler has to assume that
// an inline-asm result may be a partial-register write (the dst_sel forwarding hazard of this chip) and puts a
// wait state in front of every instruction that reads one: with one statement per instruction that was six
// `s_nop 0` per row.  In: the letters' comparison operands, the x-gap / y-gap offers, the best diagonal
// predecessor.  Out: score Sn, what the cell offers a y-gap below it (En), the "match wins" bit, and the row's
// two move bits OR-ed into mvw at bit SH.
template <int SH>
__device__ __forceinline__ void pk_row(uint32_t xlp, uint32_t ylp, uint32_t insX, uint32_t insY, uint32_t dmax, uint32_t ONES,
                                       uint32_t KSUB, uint32_t KEXT, uint32_t KDELTA, uint32_t &Sn, uint32_t &En,
                                       uint32_t &mbit, uint32_t &mvw)
{
  uint32_t t0, mx;
  asm("v_xor_b32 %[t0], %[xl], %[yl]\n\t"
      "v_pk_max_i16 %[mx], %[ix], %[iy]\n\t"
      "v_pk_min_u16 %[t0], %[t0], %[one]\n\t"
      "v_pk_mad_i16 %[t0], %[t0], %[ksub], %[dm]\n\t"
      "v_pk_max_i16 %[sn], %[t0], %[mx]\n\t"
      "v_pk_sub_i16 %[t0], %[sn], %[mx]\n\t"
      "v_pk_sub_i16 %[mx], %[mx], %[iy]\n\t"
      "v_pk_sub_i16 %[en], %[sn], %[kext]\n\t"
      "v_pk_min_u16 %[mb], %[t0], %[one]\n\t"
      "v_pk_min_u16 %[mx], %[mx], %[one]\n\t"
      "v_pk_mad_i16 %[en], %[mb], %[kdelta], %[en]\n\t"
      "v_lshl_or_b32 %[mx], %[mb], 1, %[mx]\n\t"
      "v_lshl_or_b32 %[mv], %[mx], %[sh], %[mv]"
      : [sn] "=&v"(Sn), [en] "=&v"(En), [mb] "=&v"(mbit), [mv] "+v"(mvw), [t0] "=&v"(t0), [mx] "=&v"(mx)
      : [xl] "v"(xlp), [yl] "v"(ylp), [ix] "v"(insX), [iy] "v"(insY), [dm] "v"(dmax), [one] "s"(ONES), [ksub] "s"(KSUB),
        [kext] "s"(KEXT), [kdelta] "s"(KDELTA), [sh] "n"(SH));
}

// The same row IN PLACE: Sb / Eb hold the cell two columns back on entry (a form that can name that column has read it)
// and the new cell on exit, so that a column's values never change register from step to step -- with fresh outputs the
// compiler rotated the arrays through copies (three v_mov_b64 per step and a dozen v_mov in alignment #1's loop).  The
// first row of a step (SH = 0) starts the step's word of moves.
template <int SH>
__device__ __forceinline__ void pk_row_ip(uint32_t xlp, uint32_t ylp, uint32_t insX, uint32_t insY, uint32_t dmax, uint32_t ONES,
                                          uint32_t KSUB, uint32_t KEXT, uint32_t KDELTA, uint32_t &Sb, uint32_t &Eb, uint32_t &mbit,
                                          uint32_t &mvw)
{
  uint32_t mx;
  if constexpr (SH == 0) {
    asm("v_xor_b32 %[mb], %[xl], %[yl]\n\t"
        "v_pk_max_i16 %[mx], %[ix], %[iy]\n\t"
        "v_pk_min_u16 %[mb], %[mb], %[one]\n\t"
        "v_pk_mad_i16 %[mb], %[mb], %[ksub], %[dm]\n\t"
        "v_pk_max_i16 %[sn], %[mb], %[mx]\n\t"
        "v_pk_sub_i16 %[mb], %[sn], %[mx]\n\t"
        "v_pk_sub_i16 %[mx], %[mx], %[iy]\n\t"
        "v_pk_sub_i16 %[en], %[sn], %[kext]\n\t"
        "v_pk_min_u16 %[mb], %[mb], %[one]\n\t"
        "v_pk_min_u16 %[mx], %[mx], %[one]\n\t"
        "v_pk_mad_i16 %[en], %[mb], %[kdelta], %[en]\n\t"
        "v_lshl_or_b32 %[mv], %[mb], 1, %[mx]"
        : [sn] "+v"(Sb), [en] "+v"(Eb), [mb] "=&v"(mbit), [mv] "=&v"(mvw), [mx] "=&v"(mx)
        : [xl] "v"(xlp), [yl] "v"(ylp), [ix] "v"(insX), [iy] "v"(insY), [dm] "v"(dmax), [one] "s"(ONES), [ksub] "s"(KSUB),
          [kext] "s"(KEXT), [kdelta] "s"(KDELTA));
  } else {
    asm("v_xor_b32 %[mb], %[xl], %[yl]\n\t"
        "v_pk_max_i16 %[mx], %[ix], %[iy]\n\t"
        "v_pk_min_u16 %[mb], %[mb], %[one]\n\t"
        "v_pk_mad_i16 %[mb], %[mb], %[ksub], %[dm]\n\t"
        "v_pk_max_i16 %[sn], %[mb], %[mx]\n\t"
        "v_pk_sub_i16 %[mb], %[sn], %[mx]\n\t"
        "v_pk_sub_i16 %[mx], %[mx], %[iy]\n\t"
        "v_pk_sub_i16 %[en], %[sn], %[kext]\n\t"
        "v_pk_min_u16 %[mb], %[mb], %[one]\n\t"
        "v_pk_min_u16 %[mx], %[mx], %[one]\n\t"
        "v_pk_mad_i16 %[en], %[mb], %[kdelta], %[en]\n\t"
        "v_lshl_or_b32 %[mx], %[mb], 1, %[mx]\n\t"
        "v_lshl_or_b32 %[mv], %[mx], %[sh], %[mv]"
        : [sn] "+v"(Sb), [en] "+v"(Eb), [mb] "=&v"(mbit), [mv] "+v"(mvw), [mx] "=&v"(mx)
        : [xl] "v"(xlp), [yl] "v"(ylp), [ix] "v"(insX), [iy] "v"(insY), [dm] "v"(dmax), [one] "s"(ONES), [ksub] "s"(KSUB),
          [kext] "s"(KEXT), [kdelta] "s"(KDELTA), [sh] "n"(SH));
  }
}

// A row of a step in which some lane of the wavefront sits at a node with TWO predecessors (one and two columns back, in
// either order: masks M1 / M2 say "two back" per half for the first / second; a node without a second repeats the first).
// In: what the row above hands down -- dt1, the first predecessor's cell on the diagonal, and dmax, the better of the two
// (the first wins ties, align_lpo_po2.c:348-357) -- and insY.  Out: the same for the row below, the cell in place, the
// move bits, and in sec bit K "the second predecessor gave the move that won" (strictly better on the diagonal if the
// cell is a match, for the x-insertion otherwise, :361-371).  25 instructions, one block.
template <int K>
__device__ __forceinline__ void pk_row_two(uint32_t xlp, uint32_t ylp, uint32_t M1, uint32_t M2, uint32_t Sa, uint32_t Ea, uint32_t insY,
                                           uint32_t ONES, uint32_t KSUB, uint32_t KEXT, uint32_t KDELTA, uint32_t &dt1, uint32_t &dmax,
                                           uint32_t &Sb, uint32_t &Eb, uint32_t &mvw, uint32_t &sec)
{
  uint32_t t0, mx, e1, d1;
  asm("v_xor_b32 %[t0], %[xl], %[yl]\n\t"
      "v_pk_min_u16 %[t0], %[t0], %[one]\n\t"
      "v_pk_sub_i16 %[d1], %[dm], %[dt]\n\t"
      "v_pk_mad_i16 %[t0], %[t0], %[ksub], %[dm]\n\t"
      "v_bfi_b32 %[dt], %[m1], %[sb], %[sa]\n\t"
      "v_bfi_b32 %[dm], %[m2], %[sb], %[sa]\n\t"
      "v_pk_max_i16 %[dm], %[dt], %[dm]\n\t"
      "v_bfi_b32 %[e1], %[m1], %[eb], %[ea]\n\t"
      "v_bfi_b32 %[mx], %[m2], %[eb], %[ea]\n\t"
      "v_pk_max_i16 %[mx], %[e1], %[mx]\n\t"
      "v_pk_sub_i16 %[e1], %[mx], %[e1]\n\t"
      "v_pk_max_i16 %[mx], %[mx], %[iy]\n\t"
      "v_pk_max_i16 %[sb], %[t0], %[mx]\n\t"
      "v_pk_sub_i16 %[t0], %[sb], %[mx]\n\t"
      "v_pk_sub_i16 %[mx], %[mx], %[iy]\n\t"
      "v_pk_sub_i16 %[eb], %[sb], %[kext]\n\t"
      "v_pk_min_u16 %[t0], %[t0], %[one]\n\t"
      "v_pk_min_u16 %[mx], %[mx], %[one]\n\t"
      "v_pk_mad_i16 %[eb], %[t0], %[kdelta], %[eb]\n\t"
      "v_lshl_or_b32 %[mx], %[t0], 1, %[mx]\n\t"
      "v_lshl_or_b32 %[mv], %[mx], %[sh], %[mv]\n\t"
      "v_pk_sub_i16 %[t0], 0, %[t0]\n\t"
      "v_bfi_b32 %[e1], %[t0], %[d1], %[e1]\n\t"
      "v_pk_min_u16 %[e1], %[e1], %[one]\n\t"
      "v_lshl_or_b32 %[sec], %[e1], %[k], %[sec]"
      : [sb] "+v"(Sb), [eb] "+v"(Eb), [dt] "+v"(dt1), [dm] "+v"(dmax), [mv] "+v"(mvw), [sec] "+v"(sec), [t0] "=&v"(t0),
        [mx] "=&v"(mx), [e1] "=&v"(e1), [d1] "=&v"(d1)
      : [xl] "v"(xlp), [yl] "v"(ylp), [m1] "v"(M1), [m2] "v"(M2), [sa] "v"(Sa), [ea] "v"(Ea), [iy] "v"(insY), [one] "s"(ONES),
        [ksub] "s"(KSUB), [kext] "s"(KEXT), [kdelta] "s"(KDELTA), [sh] "n"(2 * K), [k] "n"(K));
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant expression
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F &&f)
{
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

// clamp(x, lo, hi) in one instruction (lo <= hi)
__device__ __forceinline__ int med3(int x, int lo, int hi)
{
  int d;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(lo), "v"(hi));
  return d;
}

// ------------------------------------------------------------------ lane groups ---
// A pair of windows is worked on by a group of G lanes.  G = 16 .. 64: lanes q G .. q G + G - 1.  G = 8: the two groups
// of a DPP row of 16 lanes are INTERLEAVED -- lane 16 r + 2 g + p is lane g of group 2 r + p -- so that "the lane
// above" is row_shr:2, which leaves the row's first two lanes (the two groups' first lanes) with the border value they
// hold: one instruction per shift, as for G = 16 (row_shr:1) and G = 64 (wave_shr:1).  With contiguous groups of 8 the
// second group's first lane received its neighbour's value and every shift needed a select behind it.
template <int G>
struct LG {
  static __device__ __forceinline__ int q(int lane) { return G == 8 ? ((lane >> 4) << 1) | (lane & 1) : lane / G; }
  static __device__ __forceinline__ int g(int lane) { return G == 8 ? (lane >> 1) & 7 : lane & (G - 1); }
  // lane gg of the group `lane` belongs to
  static __device__ __forceinline__ int lane_of(int lane, int gg) { return G == 8 ? (lane & ~0xE) | (gg << 1) : (lane & ~(G - 1)) | gg; }
};

// DPP move: lanes without a source (or outside row mask RM) keep `old`
template <int CTRL, int RM = 0xF>
__device__ __forceinline__ int dpp_mov(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, RM, 0xF, false); }
constexpr int kRowShr = 0x110, kRowShl = 0x100, kWaveShr1 = 0x138, kRowBcast15 = 0x142, kRowBcast31 = 0x143;

// value of lane g - D of the group; a lane with g < D keeps its own (the scans' __shfl_up): one DPP move for G = 8, 16
template <int G, int D>
__device__ __forceinline__ int grp_up(int v)
{
  if constexpr (G == 8) return dpp_mov<kRowShr + 2 * D>(v, v);
  else if constexpr (G == 16) return dpp_mov<kRowShr + D>(v, v);
  else return __shfl_up(v, D, G);
}
// value of lane g + D of the group; own value where g + D >= G
template <int G, int D>
__device__ __forceinline__ int grp_down(int v)
{
  if constexpr (G == 8) return dpp_mov<kRowShl + 2 * D>(v, v);
  else if constexpr (G == 16) return dpp_mov<kRowShl + D>(v, v);
  else return __shfl_down(v, D, G);
}
// value of lane src of the group
template <int G>
__device__ __forceinline__ int grp_get(int v, int src, int lane) { return __shfl(v, LG<G>::lane_of(lane, src)); }
// value of lane g ^ d of the group
template <int G>
__device__ __forceinline__ int grp_xor(int v, int d) { return G == 8 ? __shfl_xor(v, 2 * d) : __shfl_xor(v, d, G); }

// f(integral_constant<int, 1>), f(<2>), f(<4>) ... below G: the distances of a scan over the group's lanes
template <int G, int D = 1, class F>
__device__ __forceinline__ void for_pow2_below(F &&f)
{
  if constexpr (D < G) {
    f(std::integral_constant<int, D>{});
    for_pow2_below<G, 2 * D>(f);
  }
}

// maximum over the wavefront, uniform (six DPP moves and a readlane instead of ds_bpermute round trips)
__device__ __forceinline__ int wave_max(int v)
{
  v = max(v, dpp_mov<kRowShr + 1>(v, v));
  v = max(v, dpp_mov<kRowShr + 2>(v, v));
  v = max(v, dpp_mov<kRowShr + 4>(v, v));
  v = max(v, dpp_mov<kRowShr + 8>(v, v));
  v = max(v, dpp_mov<kRowBcast15, 0xA>(v, v));
  v = max(v, dpp_mov<kRowBcast31, 0xC>(v, v));
  return __builtin_amdgcn_readlane(v, 63);
}

// value of the lane above inside the group; the group's first lane gets `border`
template <int G>
__device__ __forceinline__ uint32_t pk_shift_in(uint32_t border, uint32_t v, int g)
{
  if (G == 16) return (uint32_t)dpp_mov<kRowShr + 1>((int)border, (int)v);
  if (G == 8) return (uint32_t)dpp_mov<kRowShr + 2>((int)border, (int)v);
  const uint32_t r = (uint32_t)dpp_mov<kWaveShr1>((int)border, (int)v);
  return (G < 64 && g == 0) ? border : r;
}

// the traceback reads the moves its wave stored a moment ago: served by L2 (sc1), never by a line another
// wave's earlier use of the same scratch slot left in this CU's L1
__device__ __forceinline__ uint32_t ld_moves(const uint32_t *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int pk_align_up(int x, int a) { return (x + a - 1) & ~(a - 1); }

// Moves layout: [step t][64 lanes], one coalesced 256-byte row per DP step.  (A layout that keeps four steps of a
// lane group in one 128-byte line -- [group][t / 4][lane][t % 4], written from a four-step register buffer -- halves
// the traceback's L2 requests but costs four v_cndmask per step; the kernel is bound by VALU issue, and it was 3 %
// slower on the bench batch.)  l: lane of the group that holds the cell's row.
template <int G>
__device__ __forceinline__ int mv_word(int lane, int t, int l) { return t * 64 + LG<G>::lane_of(lane, l); }

// leading lanes of the group (from g = 0) whose flag is set
template <int G>
__device__ __forceinline__ int pk_diag_run(bool flag, int lane)
{
  const unsigned long long bm = __builtin_amdgcn_ballot_w64(flag);
  if (G == 64) return ~bm == 0ull ? 64 : __builtin_ctzll(~bm);
  if (G == 8) {
    const uint32_t gm = (uint32_t)(bm >> (lane & 0x31)) & 0x5555u;     // the group's eight bits, two apart
    return __builtin_ctz((~gm & 0x5555u) | 0x10000u) >> 1;
  }
  const unsigned long long gm = (bm >> (lane & ~(G - 1))) & ((1ull << (G & 63)) - 1);
  return __builtin_ctzll(~gm | (1ull << (G & 63)));
}

// ------------------------------------------------------------------- node record ---
// one dword per node of the graph after fusion #1, as alignment #2 reads it (LDS only):
//   bit 0   first predecessor two nodes back (else one node back)
//   bit 1   second predecessor two nodes back (a node without one repeats bit 0: both candidates are then the
//           same cell and the first wins every comparison)
//   bit 2   has a second predecessor       bit 3   first predecessor is the virtual start, more than two columns
//   back (a virtual start one or two columns back is column 0 in the registers: bits 0 / 1 as for any node)
//   bit 4   unused                         bit 5   node opens a new MSA column (ring)
//   bits 8-12 letter    bits 16-19 flags (kFlag*)    bits 24-31 ordinal among the two-predecessor nodes
//   bit 6   (k_poa<.., true> only) the first predecessor is the window's FAR node (WinP::fnode: more than two nodes back;
//           bits 0 / 3 then say "one back" and are not used)        bit 7   ... the second predecessor is
constexpr uint32_t kN_Far1 = 1u, kN_Far2 = 2u, kN_Has2 = 4u, kN_Virt1 = 8u, kN_Virt2 = 16u, kN_NewCol = 32u;
constexpr uint32_t kN_FarA = 64u, kN_FarB = 128u;

#include "poa_engine_gen.h"

// LDS slot of one window (bytes); must mirror poa_slot_need() below.
//   [hdr 16][unc symbols][node records u32 (xi_cap + 2: records 1 .. n1 between two zero guards)][union]
//   union, alignment #1 .. fusion #1: [ref + cor symbols][x2y Lr][node_ref Lr][node_cor Lc][y2x Lc]  (one byte per
//          entry in the 8-lane classes, whose windows have fewer than 255 nodes; two otherwise)
//   union, alignment #2 .. output:    [x2y u16 n1][ordinal bytes k2 * G | col_y u16 Lu]
struct WinP {
  bool valid;
  uint32_t w;
  int64_t o0;
  int Lr, Lc, Lu, n1, triv, xi_cap;
  uint8_t *slot;
  int off_xi, off_u;
  int score1, k2n;
  int fnode;                // node the one far edge of the graph comes from (-1: none)
};

// Across a dynamic program nothing of a window is wanted but what the loop itself holds; the rest -- lengths, flags, what the
// phases so far have found -- waits in the 16-byte header of the window's LDS slot (free once the symbols are staged) and is
// read back behind the loop, the window's id and output offset from its descriptor in memory.  The compiler parked the same
// values in scratch memory (up to 33 registers spilled per instance); spilled registers are what its fault of DESIGN.md 4.1
// needs, and the build accepts none in k_poa (tools/check_spills.py).
//   [0] Lr | Lc << 16   [1] Lu | triv << 16 | valid << 20 | k2n << 24   [2] n1 | (fnode + 1) << 16   [3] score1
__device__ __forceinline__ void park_win(const WinP &W, int g)
{
  if (g == 0) {
    uint4 hd;
    hd.x = (uint32_t)W.Lr | ((uint32_t)W.Lc << 16);
    hd.y = (uint32_t)W.Lu | ((uint32_t)W.triv << 16) | (W.valid ? 1u << 20 : 0u) | ((uint32_t)W.k2n << 24);
    hd.z = (uint32_t)W.n1 | ((uint32_t)(W.fnode + 1) << 16);
    hd.w = (uint32_t)W.score1;
    *reinterpret_cast<uint4 *>(W.slot) = hd;
  }
}
// desc: the window's two descriptor words in memory (k_gather's PackDesc)
__device__ __forceinline__ void unpark_win(WinP &W, uint8_t *slot, const uint4 *desc)
{
  const uint4 d0 = desc[0], d1 = desc[1];
  const uint4 hd = *reinterpret_cast<const uint4 *>(slot);
  W.slot = slot;
  W.Lr = (int)(hd.x & 0xFFFFu); W.Lc = (int)(hd.x >> 16);
  W.Lu = (int)(hd.y & 0xFFFFu); W.triv = (int)((hd.y >> 16) & 7u); W.valid = ((hd.y >> 20) & 1u) != 0u; W.k2n = (int)(hd.y >> 24);
  W.n1 = (int)(hd.z & 0xFFFFu); W.fnode = (int)(hd.z >> 16) - 1;
  W.score1 = (int)hd.w;
  W.xi_cap = poa_xi_cap(W.Lr, W.Lc);
  W.off_xi = 16 + pk_align_up(W.Lu, 4);
  W.off_u = W.off_xi + 4 * (W.xi_cap + 2);
  W.w = d0.x;
  W.o0 = W.valid ? (int64_t)(((uint64_t)d1.y << 32) | d1.x) : 0;
}

// phase stamps (debug bit 2): cycles per phase summed over waves
// The debug facilities of k_poa (PackArgs::debug: phase stamps, why windows leave, phases dropped for instruction counts,
// round 2's traceback) are compiled in with -DELECTOR_POA_DEBUG=1 only (ELECTOR_HIPCC_FLAGS): their branches -- each an
// EXEC-narrowing region around an atomic -- cost scalar registers and, with ROCm 7.2's compiler, gave the register
// allocator places to put a VGPR spill store IN FRONT of the `s_or_b64 exec` that closes the region (the store then runs
// under an empty mask and the reload reads garbage: tools/check_spills.py, run by the build).
#ifndef ELECTOR_POA_DEBUG
#define ELECTOR_POA_DEBUG 0
#endif
constexpr bool kPoaDebug = ELECTOR_POA_DEBUG != 0;
// (bisection switches of the generated loops: -DELECTOR_NO_ENG_DP1 / -DELECTOR_NO_ENG_FIRST leave a loop to the C++ steps)
#ifdef ELECTOR_NO_ENG_DP1
constexpr bool kUseEngDp1 = false;
#else
constexpr bool kUseEngDp1 = true;
#endif
#ifdef ELECTOR_NO_ENG_FIRST
constexpr bool kUseEngFirst = false;
#else
constexpr bool kUseEngFirst = true;
#endif
#define PK_STAMP(idx)                                                                             \
  do {                                                                                            \
    if ((dbg & 4) && threadIdx.x == 0) {                                                      \
      const unsigned long long now_ = __builtin_readcyclecounter();                               \
      atomicAdd(a.stamps + (idx), now_ - stamp_);                                                 \
      stamp_ = now_;                                                                              \
    }                                                                                             \
  } while (0)

// ------------------------------------------------------------------ per-window phases ---

// traceback of alignment #1 (align_lpo_po2.c:108-168), G cells per round (see poa_fused.hip).  Moves: bit 1 =
// match (diagonal), bit 0 = x-insertion beats y-insertion; window half h of the word at [step][lane].  Both
// windows of the lane group walk in lockstep: a round is one trip to the moves scratch (L2) for the two of them.
template <int G, int R, typename IT>
__device__ __forceinline__ void traceback_a(const WinP (&W)[2], const bool (&need)[2], const uint32_t *mv, int lane, int g,
                                            IT *(&x2y)[2])
{
  int x[2], y[2], guard[2];
  bool alive[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) { x[h] = W[h].Lr - 1; y[h] = W[h].Lc - 1; guard[h] = W[h].Lr + W[h].Lc + 2; alive[h] = need[h]; }
  while (__builtin_amdgcn_ballot_w64(alive[0] || alive[1]) != 0) {
    int cx[2], cy[2], rk[2];
    bool inb[2];
    uint32_t word[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      cx[h] = x[h] - g; cy[h] = y[h] - g;
      inb[h] = alive[h] && cx[h] >= 0 && cy[h] >= 0;
      const int rl = inb[h] ? cy[h] / R : 0;
      rk[h] = cy[h] - rl * R;
      word[h] = ld_moves(mv + (inb[h] ? mv_word<G>(lane, cx[h] + 1 + rl, rl) : 0));  // unconditional: both in flight
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int xo = 0, yo = 0;
      if (inb[h]) {
        const uint32_t two = (word[h] >> (16 * h + 2 * rk[h])) & 3u;
        const int m = two >> 1, xw = two & 1;
        xo = m | xw; yo = m | (xw ^ 1);
      }
      const int run = pk_diag_run<G>(inb[h] && xo && yo, lane);
      if (g < run) x2y[h][cx[h]] = (IT)cy[h];
      // where the walk goes on, from the lane at the end of the run: one shuffle for (x, y, stop)
      int nx = cx[h] - xo, ny = cy[h] - yo, fl = inb[h] ? 0 : 1;
      const int src = min(run, G - 1);
      const int nxt = grp_get<G>(((nx + 2) & 0xFFF) | (((ny + 2) & 0xFFF) << 12) | (fl << 24), src, lane);
      nx = (nxt & 0xFFF) - 2; ny = ((nxt >> 12) & 0xFFF) - 2; fl = nxt >> 24;
      if (run >= G) { nx = x[h] - G; ny = y[h] - G; fl = 0; }
      if (alive[h]) {
        x[h] = nx; y[h] = ny;
        if ((fl & 1) || --guard[h] <= 0) alive[h] = false;
      }
    }
  }
}

// fusion #1 (lpo.c:602-668 on two linear sequences) spread over the window's G lanes, as in k_fused_a; the
// node records go to LDS.  Returns false when the graph does not qualify for this kernel (a predecessor more
// than two nodes back, more nodes than the slot holds) -- the window is then handed back.
// FAR: the kernel instance that holds ONE far edge per window (the column of its source node is kept aside by the
// dynamic program); the others report such a graph through *far_out and do not take it.
template <int G, typename IT, bool FAR>
__device__ __forceinline__ bool fusion_1(WinP &W, int lane, int g, const uint8_t *xs, const uint8_t *ys, IT *x2y,
                                         uint32_t *xinfo, bool *bad_out, bool *far_out, int *why, int *nfar_out)
{
  constexpr int kNoneI = (int)(IT)~(IT)0;                               // "not aligned" in the index type of this class
  const int Lr = W.Lr, Lc = W.Lc;
  const bool on = W.valid && W.triv == 0;
  bool bad = false;
  IT *node_ref = x2y + ((Lr + 1) & ~1);
  IT *node_cor = node_ref + ((Lr + 1) & ~1);
  IT *y2x = node_cor + ((Lc + 1) & ~1);
  const int cx = (Lr + G - 1) / G, cy = (Lc + G - 1) / G;               // letters per lane
  const int cxmax = wave_max(on ? cx : 0), cymax = wave_max(on ? cy : 0);
  if (on) for (int i = g; i < Lc; i += G) y2x[i] = (IT)kNoneI;
  __builtin_amdgcn_wave_barrier();
  const int x0 = g * cx, x1 = on ? min(Lr, x0 + cx) : 0;
  const int y0 = g * cy, y1 = on ? min(Lc, y0 + cy) : 0;
  int pmax = 0, fcnt = 0;
  for (int it = 0; it < cxmax; ++it) {
    const int ix = x0 + it;
    if (ix < x1) {
      const int ay = x2y[ix];
      if (ay != kNoneI) {
        if (ay < Lc) y2x[ay] = (IT)ix; else bad = true;
        if (ay + 1 <= pmax) bad = true;                                // a path is monotone
        pmax = ay + 1;
        fcnt += (ay < Lc && xs[ix] == ys[ay]);
      }
    }
  }
  int sp = pmax, sf = fcnt;
  for_pow2_below<G>([&](auto dc) {
    constexpr int d = decltype(dc)::value;
    const int tp = grp_up<G, d>(sp), tf = grp_up<G, d>(sf);
    if (g >= d) { sp = max(sp, tp); sf += tf; }
  });
  const int fused_all = grp_get<G>(sf, G - 1, lane);
  int P = grp_up<G, 1>(sp), F = sf - fcnt;
  if (g == 0) P = 0;
  if (pmax > 0 && x1 > x0) {
    int first = 0;
    for (int ix = x0; ix < x1; ++ix) { const int ay = x2y[ix]; if (ay != kNoneI) { first = ay + 1; break; } }
    if (first <= P) bad = true;
  }
  for (int it = 0; it < cxmax; ++it) {
    const int ix = x0 + it;
    if (ix < x1) {
      const int ay = x2y[ix];
      if (ay != kNoneI) { P = ay + 1; F += (ay < Lc && xs[ix] == ys[ay]); }
      node_ref[ix] = (IT)(ix + P - F);
    }
  }
  __builtin_amdgcn_wave_barrier();
  constexpr int kUndef = -1;
  int fy = 0, klow = kUndef;
  for (int it = 0; it < cymax; ++it) {
    const int y = y1 - 1 - it;
    if (y >= y0) {
      const int x = y2x[y];
      if (x != kNoneI) { klow = x; fy += xs[x] == ys[y]; }
    }
  }
  int sfy = fy, sfx = klow;
  for_pow2_below<G>([&](auto dc) {
    constexpr int d = decltype(dc)::value;
    const int tf = grp_up<G, d>(sfy), tk = grp_down<G, d>(sfx);
    if (g >= d) sfy += tf;
    if (g + d < G && sfx == kUndef) sfx = tk;
  });
  int K = grp_down<G, 1>(sfx), fy_run = sfy;
  if (g == G - 1 || K == kUndef) K = Lr;
  for (int it = 0; it < cymax; ++it) {
    const int y = y1 - 1 - it;
    if (y >= y0) {
      const int x = y2x[y];
      const bool al = x != kNoneI, fu = al && xs[x] == ys[y];
      if (fu) --fy_run;
      if (al) K = x;
      node_cor[y] = fu ? node_ref[x] : (IT)(y - fy_run + K);
    }
  }
  __builtin_amdgcn_wave_barrier();
  const int n1 = Lr + Lc - fused_all;
  bool fits = n1 <= W.xi_cap && n1 >= 1;
  // every lane of the group must agree before any node is written
  int maxd = 1, k2 = 0, nfar = 0, fsrc = -1;
  if (on && fits) {
    auto emit = [&](int n, int letter, int flags, int ring, int sa, int sb) {
      // sa / sb: predecessor nodes (-1 none); an INITIAL node has the virtual start first (align_lpo_po2.c:69-79)
      uint32_t rec = ((uint32_t)letter << 8) | ((uint32_t)flags << 16) | (ring == n ? kN_NewCol : 0u);
      int d1, d2 = 0;
      bool v1 = false, v2 = false, has2 = false;
      // the virtual start is column 0: from node n (column n + 1) it lies n + 1 columns back
      if (sa < 0) { v1 = true; d1 = n + 1; }
      else if (flags & kFlagInitial) { v1 = true; d1 = n + 1; has2 = true; d2 = n - sa; if (sb >= 0) bad = true; }
      else { d1 = n - sa; if (sb >= 0) { has2 = true; d2 = n - sb; } }
      // a virtual start more than two columns back cannot come from the two register columns: the DP patches
      // in the cells of column 0, which depend on the row only
      if (v1 && d1 > 2) { rec |= kN_Virt1; d1 = 1; }
      // an edge from more than two nodes back (a corrected piece that aligns at both ends of the window, an indel of
      // two or more letters in the corrected read): one per window is kept aside by k_poa<.., true>
      if (!v1 && d1 > 2) { ++nfar; fsrc = sa; d1 = 1; rec |= kN_FarA | (has2 ? 0u : kN_FarB); }
      if (has2 && d2 > 2) { ++nfar; fsrc = sb; d2 = 1; rec |= kN_FarB; }
      maxd = max(maxd, max(d1, d2));
      if (d1 == 2) rec |= kN_Far1;
      if (has2) { rec |= kN_Has2; if (d2 == 2) rec |= kN_Far2; ++k2; }
      else if (d1 == 2) rec |= kN_Far2;
      (void)v2;
      if (n >= 0 && n < n1) xinfo[n + 1] = rec; else bad = true;
    };
    for (int it = 0; it < cxmax; ++it) {
      const int ix = x0 + it;
      if (ix < x1) {
        const int ay = x2y[ix], n = node_ref[ix];
        const bool al = ay != kNoneI && ay < Lc, fu = al && xs[ix] == ys[ay];
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
        if (!(x != kNoneI && xs[x] == ys[y])) {
          const int n = node_cor[y];
          emit(n, ys[y], kFlagHasCor | (y == 0 ? kFlagInitial : 0) | (y == Lc - 1 ? kFlagFinal : 0), n,
               y > 0 ? (int)node_cor[y - 1] : -1, -1);
        }
      }
    }
  }
  for (int d = 1; d < G; d <<= 1) {
    maxd = max(maxd, grp_xor<G>(maxd, d));
    bad = bad || grp_xor<G>(bad ? 1 : 0, d) != 0;
    nfar += grp_xor<G>(nfar, d);
    fsrc = max(fsrc, grp_xor<G>(fsrc, d));
  }
  if (on) W.n1 = n1;
  *bad_out = bad;
  const bool near_ok = !on || (fits && maxd <= 2 && !bad);
  *why = !on ? 0 : !fits ? 1 : bad ? 2 : nfar == 1 ? 3 : nfar > 1 ? 4 : 0;      // (debug: why a window is not kept)
  *nfar_out = nfar;
  if (FAR) {
    if (on && nfar == 1) W.fnode = fsrc;
    return near_ok && (!on || nfar <= 1);
  }
  *far_out = on && near_ok && nfar == 1;
  return near_ok && (!on || nfar == 0);
}

// the graph of a window whose corrected sequence equals its reference (chain, every node holds both letters)
// or differs from it by one substitution at position e (the corrected letter gets a node of its own right
// before the reference letter's, both in one ring: lpo.c:449-450,647-649) -- see k_trivial in poa_kernels.hip
template <int G>
__device__ __forceinline__ void trivial_graph(WinP &W, int g, const uint8_t *xs, const uint8_t *ys, uint32_t *xinfo)
{
  if (!W.valid || W.triv == 0) return;
  const int L = W.Lr;
  if (W.triv == 1) {
    for (int n = g; n < L; n += G) {
      const int fl = kFlagHasRef | kFlagHasCor | (n == 0 ? kFlagInitial : 0) | (n == L - 1 ? kFlagFinal : 0);
      xinfo[n + 1] = ((uint32_t)xs[n] << 8) | ((uint32_t)fl << 16) | kN_NewCol;   // node 0: the virtual start is column 0, one back
    }
    W.n1 = L;
    W.k2n = 0;
    return;
  }
  if (W.triv == 5) {
    // the one-letter filler that occurs nowhere in the reference (k_trivial): the reference's chain, then the
    // letter, unaligned, as a node of its own whose only predecessor is the virtual start -- far behind it
    for (int n = g; n < L; n += G) {
      const int fl = kFlagHasRef | (n == 0 ? kFlagInitial : 0) | (n == L - 1 ? kFlagFinal : 0);
      xinfo[n + 1] = ((uint32_t)xs[n] << 8) | ((uint32_t)fl << 16) | kN_NewCol;
    }
    if (g == 0)
      xinfo[L + 1] = ((uint32_t)ys[0] << 8) | ((uint32_t)(kFlagHasCor | kFlagInitial | kFlagFinal) << 16) | kN_NewCol | kN_Virt1;
    W.n1 = L + 1;
    W.k2n = 0;
    return;
  }
  const int Lc = W.Lc, nm = min(L, Lc);
  int e = nm;
  for (int i = g; i < nm && i < e; i += G) if (xs[i] != ys[i]) e = i;
  for (int d = 1; d < G; d <<= 1) e = min(e, grp_xor<G>(e, d));
  if (W.triv == 3) {
    // one deleted letter, reference letter e (k_trivial): the chain of the reference with letter e on its own;
    // the node after it has the corrected read's edge from two nodes back as its second predecessor -- or, when
    // e = 0, the virtual start (two columns back) first and node 0 second (what fusion_1's emit makes of it)
    for (int i = g; i < L; i += G) {
      int fl = kFlagHasRef | (i == 0 ? kFlagInitial : 0) | (i == L - 1 ? kFlagFinal : 0);
      if (i != e) { const int y = i < e ? i : i - 1; fl |= kFlagHasCor | (y == 0 ? kFlagInitial : 0) | (y == Lc - 1 ? kFlagFinal : 0); }
      uint32_t rec = ((uint32_t)xs[i] << 8) | ((uint32_t)fl << 16) | kN_NewCol;
      if (i == e + 1) rec |= e == 0 ? (kN_Far1 | kN_Has2) : (kN_Has2 | kN_Far2);
      xinfo[i + 1] = rec;
    }
    W.n1 = L;
    W.k2n = e + 1 < L ? 1 : 0;                       // the one two-predecessor node, ordinal 0 (the record's ordinal bits are 0)
    return;
  }
  if (W.triv == 4) {
    // one inserted letter, corrected letter e: a node of its own at index e; the reference letter after it
    // (node e + 1) has the node before the insertion (two back; the virtual start when e = 0) first and the
    // inserted letter's node second
    for (int n = g; n < L + 1; n += G) {
      uint32_t rec;
      if (n == e) {
        const int fl = kFlagHasCor | (e == 0 ? kFlagInitial : 0) | (e == Lc - 1 ? kFlagFinal : 0);
        rec = ((uint32_t)ys[e] << 8) | ((uint32_t)fl << 16) | kN_NewCol;
      } else {
        const int ix = n < e ? n : n - 1;                               // fused with corrected letter n
        const int fl = kFlagHasRef | kFlagHasCor | ((ix == 0 || n == 0) ? kFlagInitial : 0) | ((ix == L - 1 || n == Lc - 1) ? kFlagFinal : 0);
        rec = ((uint32_t)xs[ix] << 8) | ((uint32_t)fl << 16) | kN_NewCol;
        if (n == e + 1) rec |= kN_Far1 | kN_Has2;
      }
      xinfo[n + 1] = rec;
    }
    W.n1 = L + 1;
    W.k2n = e + 1 <= L ? 1 : 0;
    return;
  }
  for (int i = g; i < L; i += G) {
    const int fl_pos = (i == 0 ? kFlagInitial : 0) | (i == L - 1 ? kFlagFinal : 0);
    if (i != e) {
      const int n = i < e ? i : i + 1;
      // after the bubble: first predecessor the reference letter's node (1 back), second the corrected letter's (2 back)
      uint32_t rec = ((uint32_t)xs[i] << 8) | ((uint32_t)(kFlagHasRef | kFlagHasCor | fl_pos) << 16) | kN_NewCol;
      if (i == e + 1) rec |= kN_Has2 | kN_Far2;
      xinfo[n + 1] = rec;
    } else {
      // corrected letter: predecessor one back (virtual at the window's start); opens the column
      uint32_t rc = ((uint32_t)ys[e] << 8) | ((uint32_t)(kFlagHasCor | fl_pos) << 16) | kN_NewCol;
      // reference letter: predecessor two back (the node before the bubble), same column
      uint32_t rr = ((uint32_t)xs[e] << 8) | ((uint32_t)(kFlagHasRef | fl_pos) << 16) | kN_Far1 | kN_Far2;
      xinfo[e + 1] = rc;
      xinfo[e + 2] = rr;
    }
  }
  W.n1 = L + 1;
  W.k2n = e + 1 < L ? 1 : 0;
}

// the column layout rule in its plain serial form (one lane): the fallback of the parallel version in k_poa
// (fuse2_columns_serial of poa_fused.hip with the ring ids read off the column-start bits)
__device__ __noinline__ int pk_columns_serial(int n1, int Lu, const uint32_t *xinfo, const uint16_t *x2y, const uint8_t *ys,
                                              const uint8_t *chr, uint8_t *cols_st)
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
  int n = 0, iy = 0, blk_new = -1;
  for (int ix = 0; ix < n1; ++ix) {
    if (xinfo[ix + 1] & kN_NewCol) blk_new = -1;
    for (int k = ix; k < n1 && (k == ix || !(xinfo[k + 1] & kN_NewCol)); ++k) {
      const int ay = x2y[k];
      if (ay != (int)kNone16) {
        while (iy < ay) { place(n, ys[iy], false, false, true); ++n; ++iy; }
        break;
      }
    }
    const uint32_t xv = xinfo[ix + 1];
    const int letter = (xv >> 8) & 0xFF, fl = (int)((xv >> 16) & 0xFF);
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


// traceback of alignment #2 (align_lpo_po2.c:108-168), G cells per round: a cell keeps the walk on its
// diagonal when it is a match whose chosen predecessor is the node right before it.  Both windows of the lane
// group in lockstep, as in traceback_a.
// The per-lane work of a round is straight code: everything is computed for every lane on clamped addresses and merged
// with selects; the one rarely needed look-up (which of two predecessors a cell took: one node per window has two) sits
// behind a wave-wide test.  (With nested conditions a round was some 250 instructions, most of them exec-mask
// bookkeeping.)
template <int G, int R, bool FAR>
__device__ __forceinline__ void traceback_b2(const WinP (&W)[2], const uint32_t *mv, int lane, int g, uint32_t *(&xinfo)[2],
                                             uint8_t *(&ordb)[2], uint16_t *(&x2y)[2], const int (&bestx)[2],
                                             bool (&bad)[2], int &rounds)
{
  int x[2], y[2], guard[2];
  bool alive[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    x[h] = W[h].valid ? bestx[h] : -1; y[h] = W[h].Lu - 1; guard[h] = W[h].n1 + W[h].Lu + 2; alive[h] = W[h].valid;
    bad[h] = false;
  }
  while (__builtin_amdgcn_ballot_w64(alive[0] || alive[1]) != 0) {
    ++rounds;
    int cx[2], cy[2], rk[2], rlv[2];
    bool inb[2];
    uint32_t word[2], rec[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      cx[h] = x[h] - g; cy[h] = y[h] - g;
      inb[h] = alive[h] && (cx[h] | cy[h]) >= 0;
      const int cyc = inb[h] ? cy[h] : 0, cxc = inb[h] ? cx[h] : -1;
      const int rl = cyc / R;
      rlv[h] = rl; rk[h] = cyc - rl * R;
      word[h] = ld_moves(mv + (inb[h] ? mv_word<G>(lane, cxc + 1 + rl, rl) : 0));
      rec[h] = xinfo[h][cxc + 1];                                        // record 0 is the zero guard
    }
    // which of two predecessors: only cells at the (rare) two-predecessor nodes that step along x ask
    int sec[2] = {0, 0};
    {
      bool ask[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) ask[h] = inb[h] && (rec[h] & kN_Has2) != 0u;
      if (__builtin_amdgcn_ballot_w64(ask[0] || ask[1]) != 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (ask[h]) sec[h] = (ordb[h][(rec[h] >> 24) * G + rlv[h]] >> rk[h]) & 1;
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t two = inb[h] ? (word[h] >> (16 * h + 2 * rk[h])) & 3u : 0u;
      const int m = (int)(two >> 1), xw = (int)(two & 1u);
      const int xo = inb[h] ? (m | xw) : 0, yo = inb[h] ? (m | (xw ^ 1)) : 0;
      const int far = (int)((rec[h] >> sec[h]) & 1u);                   // bit 0: first predecessor two back, bit 1: second
      int px = cx[h] - 1 - far;                                        // < 0: the virtual start
      bool virt = px < -1 || (sec[h] == 0 && (rec[h] & kN_Virt1) != 0u);
      if (FAR) {                                                       // the chosen predecessor is the window's far node
        const bool fx = (rec[h] & (sec[h] ? kN_FarB : kN_FarA)) != 0u;
        px = fx ? W[h].fnode : px;
        virt = virt && !fx;
      }
      px = xo ? (virt ? -1 : px) : cx[h];
      const bool diag = xo && yo;                                      // (inb is in both)
      const int run = pk_diag_run<G>(diag && px == cx[h] - 1, lane);
      if (diag && g <= run) x2y[h][cx[h]] = (uint16_t)cy[h];           // the run's pairs, and the breaker's if it is a match
      const int nxp = ((px + 2) & 0xFFF) | (((cy[h] - yo + 2) & 0xFFF) << 12) | (inb[h] ? 0 : 1 << 24);
      const int nxt = grp_get<G>(nxp, min(run, G - 1), lane);
      const bool whole = run >= G;
      const int nx = whole ? x[h] - G : (nxt & 0xFFF) - 2, ny = whole ? y[h] - G : ((nxt >> 12) & 0xFFF) - 2;
      const bool stop = !whole && (nxt >> 24) != 0;
      if (alive[h]) {
        x[h] = nx; y[h] = ny;
        if (stop) alive[h] = false;
        else if (--guard[h] <= 0) { bad[h] = true; alive[h] = false; }
      }
    }
  }
}

// fusion #2 and the MSA columns (lpo.c:602-668 column layout rule, lpo_format.c:337-393) over the window's G
// lanes, as in k_fused_b: every ring of the (ref + cor) graph is one column, an uncorrected letter aligned to
// one of the ring's nodes joins it, every other uncorrected letter gets a column of its own just before the
// next ring that holds an aligned letter.  Returns the number of columns staged in cols_st.
template <int G>
__device__ __forceinline__ int columns_2(const WinP &W, int lane, int g, const uint32_t *xinfo, const uint16_t *x2y, const uint8_t *ys,
                                         const uint8_t *chr, uint8_t *cols_st, uint16_t *col_y, bool bad)
{
  const bool valid = W.valid;
  const int n1 = W.n1, Lu = W.Lu;
  const int cn = (n1 + G - 1) / G, cy = (Lu + G - 1) / G;
  const int cnmax = wave_max(valid ? cn : 0), cymax = wave_max(valid ? cy : 0);
  if (valid) for (int i = g; i < Lu; i += G) col_y[i] = (uint16_t)kNone16;
  const int i0 = g * cn, i1 = valid ? min(n1, i0 + cn) : 0;
  auto starts = [&](int ix) { return (xinfo[ix + 1] & kN_NewCol) != 0; };
  // the aligned letter of the ring that starts at node ix (-1: none); cnt > 1 cannot come from a path
  auto ring_aligned = [&](int ix, bool *odd) {
    int ay = -1, cnt = 0;
    for (int k = ix; k < n1 && (k == ix || !starts(k)); ++k) {
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
    if (ix < i1 && starts(ix)) {
      ++ngs;
      const int ay = ring_aligned(ix, &odd);
      if (ay >= 0) { ++nal; if (ay <= last_ay) odd = true; last_ay = ay; }
    }
  }
  int sg = ngs, sa = nal, sy = last_ay;                                  // inclusive scans over the group's lanes
  for_pow2_below<G>([&](auto dc) {
    constexpr int d = decltype(dc)::value;
    const int tg = grp_up<G, d>(sg), ta = grp_up<G, d>(sa), ty = grp_up<G, d>(sy);
    if (g >= d) { sg += tg; sa += ta; sy = max(sy, ty); }
  });
  const int prev_ay = grp_up<G, 1>(sy);
  int gcount = sg - ngs, alc = sa - nal, A = g > 0 ? prev_ay : -1;
  const int ngroups = grp_get<G>(sg, G - 1, lane), nal_all = grp_get<G>(sa, G - 1, lane);
  int ncol = ngroups + Lu - nal_all;
  __builtin_amdgcn_wave_barrier();
  for (int it = 0; it < cnmax; ++it) {
    const int ix = i0 + it;
    if (ix < i1 && starts(ix)) {
      int ay = -1;
      uint8_t c0 = '.', c1 = '.';
      for (int k = ix; k < n1 && (k == ix || !starts(k)); ++k) {
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
  for_pow2_below<G>([&](auto dc) {
    constexpr int d = decltype(dc)::value;
    const int t = grp_down<G, d>(sfx);
    if (g + d < G && sfx == kUndef) sfx = t;
  });
  int K = grp_down<G, 1>(sfx);
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
  for (int d = 1; d < G; d <<= 1) odd = odd || grp_xor<G>(odd ? 1 : 0, d) != 0;
  __builtin_amdgcn_wave_barrier();
  // an alignment that is not a monotone path through the rings cannot happen; if it ever does, the plain
  // serial form of the rule decides
  if (valid && odd && g == 0 && !bad) ncol = pk_columns_serial(n1, Lu, xinfo, x2y, ys, chr, cols_st);
  ncol = grp_get<G>(ncol, 0, lane);
  return ncol;
}

// ------------------------------------------------------------------------ k_gather ---
// What a wavefront of k_poa needs of its 16 .. 2 windows, laid out in LIST order: a 32-byte descriptor per list
// entry and the entry's three symbol strings (reference, corrected, uncorrected back to back) at a fixed stride.
// k_poa then stages a wavefront's windows with loads whose addresses depend on nothing but the block index -- one
// trip to memory for descriptors and symbols together, whole cache lines -- instead of three dependent trips
// (list entry -> offsets / flags scattered over the batch's arrays -> symbols) to some eighty different lines.
// The random accesses move here, into a kernel that has nothing to wait for but memory and hides it with
// occupancy.  8 lanes per list entry.
struct PackDesc { uint32_t w, Lr, Lc, Lu; uint32_t o0_lo, o0_hi, triv_ok, pad; };   // triv | ok << 8


// one entry's work: p = its index in list / pdesc, psym / pstride = its list's
__device__ __forceinline__ void gather_entry(const GatherArgs &a, int64_t p, int g, uint32_t *psym, int pstride);

__global__ void __launch_bounds__(256) k_gather(GatherArgs a)
{
  const int64_t p = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
  const int g = threadIdx.x & 7;
  if (p >= a.nlist) return;
  if (a.nlist_dev && p >= (int64_t)*a.nlist_dev) return;                  // (k_poa reads the same count)
  gather_entry(a, p, g, a.psym + p * (int64_t)a.pstride, a.pstride);
}

__device__ __forceinline__ void gather_entry(const GatherArgs &a, int64_t p, int g, uint32_t *dst, int pstride)
{
  const uint32_t w = a.list[p];
  const int64_t o0 = a.off[3 * (int64_t)w], o1 = a.off[3 * (int64_t)w + 1], o2 = a.off[3 * (int64_t)w + 2], o3 = a.off[3 * (int64_t)w + 3];
  const int64_t total = o3 - o0;
  const bool ok = a.status[w] == 0 && a.done_a[w] == 0 && a.done_b[w] == 0 && total <= 4 * (int64_t)pstride;
  if (g == 0) {
    a.pdesc[2 * p] = make_uint4(w, (uint32_t)(o1 - o0), (uint32_t)(o2 - o1), (uint32_t)(o3 - o2));
    a.pdesc[2 * p + 1] = make_uint4((uint32_t)((uint64_t)o0 & 0xFFFFFFFFu), (uint32_t)((uint64_t)o0 >> 32),
                                    (uint32_t)a.triv[w] | (ok ? 0x100u : 0u), 0u);
  }
  if (!ok) return;
  // Sixteen bytes per load from wherever the window starts (the hardware takes unaligned addresses), two loads of a lane
  // asked for before either is stored: a window of up to 256 letters -- nearly all -- is ONE trip to memory for its eight
  // lanes.  (Round 4: aligned dword pairs and a funnel shift per dword, a trip per 32 letters.)  d_sym is padded by 64
  // bytes and an entry's stride is a multiple of 16 that holds the window: the sixteen bytes of the last piece stay inside.
  const uint8_t *src = a.sym + o0;
  const int nq = (int)((total + 15) >> 4);
  for (int k0 = 0; k0 < nq; k0 += 16) {
    const int ka = k0 + g, kb = k0 + 8 + g;
    uint4 va, vb;
    if (ka < nq) __builtin_memcpy(&va, src + 16 * (int64_t)ka, 16);
    if (kb < nq) __builtin_memcpy(&vb, src + 16 * (int64_t)kb, 16);
    if (ka < nq) __builtin_memcpy(dst + 4 * ka, &va, 16);
    if (kb < nq) __builtin_memcpy(dst + 4 * kb, &vb, 16);
  }
}

void launch_gather(const GatherArgs &a, hipStream_t st)
{
  if (a.nlist <= 0) return;
  hipLaunchKernelGGL(k_gather, dim3((unsigned)((a.nlist + 31) / 32)), dim3(256), 0, st, a);
}

// ------------------------------------------------------------------------ k_poa ---

// window descriptor -> WinP; the loads were issued by the caller (all of both windows in flight together)
template <int G, int R>
__device__ __forceinline__ void fit_win(WinP &W, const PackArgs &a, bool listed, uint32_t w, int64_t o0, int Lr, int Lc, int Lu,
                                        int triv, uint8_t *slot)
{
  constexpr int RS = R * G;
  const KParams kp = a.b.kp;
  W.valid = listed;
  W.w = w;
  W.o0 = listed ? o0 : 0;
  W.Lr = listed ? Lr : 0; W.Lc = listed ? Lc : 0; W.Lu = listed ? Lu : 0;
  W.triv = listed ? triv : 0;
  W.n1 = 0; W.score1 = kNeg; W.k2n = 0; W.fnode = -1;
  W.slot = slot;
  W.xi_cap = poa_xi_cap(W.Lr, W.Lc);
  W.off_xi = 16 + pk_align_up(W.Lu, 4);
  W.off_u = W.off_xi + 4 * (W.xi_cap + 2);                     // records 1 .. n1 between two zero guards
  const int ua = poa_union_a(W.Lr, W.Lc, G), ub = poa_union_b(W.xi_cap, W.Lu, G);
  W.valid = W.valid && W.off_u + max(ua, ub) <= a.slot_bytes && W.Lc <= RS && W.Lu <= RS &&
            (poa_idx_bytes(G) > 1 || W.Lr + W.Lc <= 254) &&
            max(W.Lr, W.xi_cap) + G + 2 <= a.mv_tw && score_span(kp, max(W.Lr, W.xi_cap) + G, RS) < 16000;
}

// Three wavefronts per SIMD (168 registers) for every class.  Up to round 4 the classes of up to six rows per lane ran four
// (128 registers, a dozen or two of them spilled): +6 % then, when a wavefront's life was mostly waiting.  With the
// dynamic programs as generated loops a wavefront issues close to what a SIMD can take; three of them measured 1-2.5 %
// FASTER than four (un-overlapped k_poa 5.38 against 5.44 ms on the E. coli batch, 8.41 against 8.62 on yeast -split) --
// and the four-wave build, at 54 spilled registers beside the loops' pinned ones, ran into the compiler's
// spill-under-a-narrowed-mask fault again: windows differing from call to call.
#ifndef ELECTOR_POA_WAVES
#define ELECTOR_POA_WAVES 3
#endif
template <int G, int R, bool FAR>
__global__ void __launch_bounds__(64, ELECTOR_POA_WAVES) k_poa(PackArgs a)
{
  extern __shared__ __align__(16) uint8_t lds[];
  constexpr int NP = 64 / G;                       // pairs of windows per wave
  // index type of the alignment #1 / fusion #1 maps: one byte in the classes whose windows are short
  using IT = typename std::conditional<poa_idx_bytes(G) == 1, uint8_t, uint16_t>::type;
  const int lane = threadIdx.x, q = LG<G>::q(lane), g = LG<G>::g(lane);
  const int dbg = kPoaDebug ? a.debug : 0;         // (a build without ELECTOR_POA_DEBUG holds none of the debug branches)
  // a far list is launched at its capacity: the blocks outside leave before they touch anything
  const int64_t nlist = a.nlist_dev ? min(a.nlist, (int64_t)*a.nlist_dev) : a.nlist;
  if ((int64_t)blockIdx.x * (2 * NP) >= nlist) return;
  const KParams kp = a.b.kp;
  uint8_t *chr = lds;
  unsigned long long stamp_ = (dbg & 4) ? __builtin_readcyclecounter() : 0;

  // Moves scratch.  The moves of a wave are dead when it ends, so the scratch is a pool of slots as large as the
  // number of waves that can be resident, not one region per block: the same few dozen megabytes are written and
  // read back over and over and stay in L2 / Infinity Cache instead of streaming through HBM.  The per-XCD L2s
  // are not coherent with each other, so a slot is only ever used from one XCD: the pool is split by XCC id, and
  // every XCD has a queue of its free slot ids -- a wave takes the id at its ticket (head), gives it back at the
  // end (tail).  There are more slots than an XCD can hold waves, so a ticket's entry is filled by the time it
  // is drawn or shortly after.
  uint32_t xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  int32_t *mvq = a.mv_q + (size_t)xcc * kPoolStride;
  uint32_t ticket = 0;

  // ---- descriptors and symbols of both windows in ONE trip: k_gather laid them out in list order, so every address
  // below follows from the block index.  The symbol loads of the first round are issued before the descriptors
  // have arrived (a fixed number of dwords per entry, the class's stride); what lies behind a window's end is read
  // and dropped. ----
  WinP W[2];
  const int64_t pi = (int64_t)blockIdx.x * NP + q;
  uint8_t *us[2], *U[2];
  uint32_t *xinfo[2];
  bool any_valid;
  // the windows' values back from their slot headers (park_win), the pointers into the slots with them
  auto unpark = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t e = 2 * pi + h < nlist ? 2 * pi + h : 0;
      unpark_win(W[h], lds + 64 + (size_t)(2 * LG<G>::q((int)threadIdx.x) + h) * a.slot_bytes, a.pdesc + 2 * e);
      us[h] = W[h].slot + 16;
      xinfo[h] = reinterpret_cast<uint32_t *>(W[h].slot + W[h].off_xi);
      U[h] = W[h].slot + W[h].off_u;
    }
  };
  {
    constexpr int UB = G == 8 ? 8 : 4;                          // dwords per lane, window and round
    bool inl[2];
    const uint32_t *src4[2];
    uint4 d0[2], d1[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      inl[h] = 2 * pi + h < nlist;
      const int64_t e = inl[h] ? 2 * pi + h : 0;
      d0[h] = a.pdesc[2 * e];
      d1[h] = a.pdesc[2 * e + 1];
      src4[h] = a.psym + e * (int64_t)a.pstride;
    }
    const uint8_t chr_l = a.b.tab->chr[lane & 31];
    const int nd = a.pstride;
    uint32_t v[2][UB];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int k = g + u * G;
        v[h][u] = k < nd ? src4[h][k] : 0u;
      }
    // the ticket is drawn behind the loads: memory operations return in order, so the loads do not wait for the
    // (slower) atomic
    if (lane == 0) ticket = (uint32_t)atomicAdd(mvq, 1);
    if (lane < 32) chr[lane] = chr_l;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool listed = inl[h] && ((d1[h].z >> 8) & 1u) != 0u;
      const int64_t o0 = (int64_t)(((uint64_t)d1[h].y << 32) | d1[h].x);
      fit_win<G, R>(W[h], a, listed, d0[h].x, o0, (int)d0[h].y, (int)d0[h].z, (int)d0[h].w, (int)(d1[h].z & 0xFFu),
                    lds + 64 + (size_t)(2 * q + h) * a.slot_bytes);
      // a listed window this kernel cannot take (slot, rows, score range) goes to the two-kernel path at once
      if (listed && !W[h].valid && g == 0) {
        a.hand[atomicAdd(a.hand_count, 1)] = W[h].w;
        if (dbg & 8) atomicAdd(a.stamps + 6, 1ull);                // (debug: refused at the door: slot, rows, score range)
      }
    }
    any_valid = __builtin_amdgcn_ballot_w64(W[0].valid || W[1].valid) != 0;
    PK_STAMP(8);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      us[h] = W[h].slot + 16;
      xinfo[h] = reinterpret_cast<uint32_t *>(W[h].slot + W[h].off_xi);
      U[h] = W[h].slot + W[h].off_u;
    }
    if (any_valid) {
      int nrc[2], ntot[2], ndw = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        nrc[h] = W[h].Lr + W[h].Lc;
        ntot[h] = W[h].valid ? nrc[h] + W[h].Lu : 0;
        ndw = max(ndw, (ntot[h] + 3) >> 2);
      }
      ndw = wave_max(ndw);
      for (int kb = 0; kb < ndw; kb += UB * G) {
        if (kb > 0) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int u = 0; u < UB; ++u) {
              const int k = kb + g + u * G;
              v[h][u] = k < nd ? src4[h][k] : 0u;
            }
        }
        // Dword k holds the string's bytes 4k .. 4k + 3: reference + corrected live in U, the uncorrected string in
        // us.  Whole (unaligned) dwords are stored; the one that straddles the border goes to both places.  Bytes
        // that fall outside a region land in its slack -- the slot's 16-byte header in front of us, the node
        // records' guards either side, the map area behind the strings in U -- all of which is written later.
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            const int i0 = 4 * (kb + g + u * G);
            if (i0 < ntot[h]) {
              if (i0 < nrc[h]) __builtin_memcpy(U[h] + i0, &v[h][u], 4);
              if (i0 + 3 >= nrc[h]) __builtin_memcpy(us[h] + (i0 - nrc[h]), &v[h][u], 4);
            }
          }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  PK_STAMP(9);
  int mslot = 0;
  if (lane == 0) {
    int32_t *e = mvq + 32 + ticket % (uint32_t)a.mv_slots;
    while ((mslot = atomicExch(e, -1)) < 0) __builtin_amdgcn_s_sleep(2);
  }
  mslot = __builtin_amdgcn_readfirstlane(mslot);
  PK_STAMP(10);
  uint32_t *mv = a.mv_pool + ((size_t)xcc * a.mv_slots + mslot) * (size_t)a.mv_tw * 64;
  if (!any_valid) {                                                    // nothing for this wave: the slot goes straight back
    if (lane == 0) {
      const uint32_t t = (uint32_t)atomicAdd(mvq + 16, 1);
      int32_t *e = mvq + 32 + t % (uint32_t)a.mv_slots;
      while (atomicCAS(e, -1, mslot) != -1) __builtin_amdgcn_s_sleep(2);
    }
    return;
  }
  __syncthreads();
  PK_STAMP(0);
  // instruction-count experiments (debug bits 32 .. 256): drop the windows after a phase, the rest of the kernel idles
  if (dbg & 32) { W[0].valid = W[1].valid = false; }

  // packed constants (both halves alike)
  const uint32_t ONES = 0x00010001u;
  const uint32_t KSUB = pk1(kp.mismatch), KEXT = pk1(kp.ext_x), KDELTA = pk1(-(kp.open_x - kp.ext_x));   // match == 0 (host)
  // Every score of the two dynamic programs is kept ONE BELOW its value (kBias: the borders start at -1, the scores reported
  // add it back): the recurrences only compare scores and add penalties, so nothing else changes -- but every 16-bit half is
  // then negative, negative halves are ordered as unsigned numbers the way the scores are, and a subtraction whose result
  // cannot be negative (a maximum less one of its arguments, a score less a small constant) needs no borrow from the upper
  // half: the generated loops do it as ONE 32-bit v_sub_u32 (2 cycles) instead of v_pk_sub_i16 (4).  Nothing wraps: every
  // cell a lane ever holds -- rows below a window's last, columns behind its last node included -- is a cell of some
  // alignment of at most the wavefront's longest window, and that window passed score_span() < 16000.
  constexpr int kBias = -1;

  // ================= alignment #1 (linear x linear), the windows that need it =================
  bool needA[2] = {W[0].valid && W[0].triv == 0, W[1].valid && W[1].triv == 0};
  if (__builtin_amdgcn_ballot_w64(needA[0] || needA[1]) != 0) {
    // (the reference letters by their LDS offset: no pointer lives through the loop)
    const uint32_t oxA = (uint32_t)(64 + (2 * q) * a.slot_bytes + W[0].off_u), oxB = (uint32_t)(64 + (2 * q + 1) * a.slot_bytes + W[1].off_u);
    // (register tuples: the generated loop, poa_engine_gen.h, takes them pinned to fixed registers; S2 is its second column)
    typedef uint32_t VR __attribute__((ext_vector_type(R)));
    VR ylp, S, E, S2;
    static_for<R>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const int ii = R * g + 1 + k;
      const int ya = (needA[0] && ii <= W[0].Lc) ? U[0][W[0].Lr + ii - 1] : 255;
      const int yb = (needA[1] && ii <= W[1].Lc) ? U[1][W[1].Lr + ii - 1] : 255;
      ylp[k] = pk2(ya, yb);
      const int v = -(kp.open_y + (ii - 1) * kp.ext_y);       // column -1: ii gap steps from the origin
      S[k] = pk1(v + kBias);
      E[k] = pk1(v - kp.ext_x + kBias);
      S2[k] = 0u;
    });
    uint32_t dg0 = pk1((g == 0 ? 0 : -(kp.open_y + (R * g - 1) * kp.ext_y)) + kBias);   // cell (row above, column -1)
    // the last step some lane needs: the lane of a window's last row at its last column (a class's strip is taller than most of
    // its windows: the lanes below a window's last row never mattered, and the steps that only they took are not run)
    const int tmax = wave_max(max(needA[0] ? W[0].Lr + (W[0].Lc - 1) / R : 0, needA[1] ? W[1].Lr + (W[1].Lc - 1) / R : 0));
    int xa_next = (needA[0] && g == 0 && W[0].Lr >= 1) ? lds[oxA] : 0, xb_next = (needA[1] && g == 0 && W[1].Lr >= 1) ? lds[oxB] : 0;
    const int gstar0 = (W[0].Lc - 1) / R, kstar0 = (W[0].Lc - 1) % R, gstar1 = (W[1].Lc - 1) / R, kstar1 = (W[1].Lc - 1) % R;
    uint32_t bS = pk1(-kp.open_x + kBias);                     // row -1 at column t: -(open_x + (t - 1) ext_x)
    const int lastA = max(W[0].Lr - 1, 0), lastB = max(W[1].Lr - 1, 0);
    const int capA = (needA[0] && g == gstar0) ? W[0].Lr : -1, capB = (needA[1] && g == gstar1) ? W[1].Lr : -1;   // column to watch, or none
    int sc1a = kNeg, sc1b = kNeg;                              // the two scores, where they appear
    // FIRST: the steps in which some lane has not reached its first column yet (t < G)
    auto stepA = [&](auto first_tag, int t) {
      constexpr bool FIRST = decltype(first_tag)::value;
      const uint32_t upS = pk_shift_in<G>(bS, S[R - 1], g);
      const uint32_t upE = pk_shift_in<G>(pk_subk(bS, KEXT), E[R - 1], g);
      bS = pk_subk(bS, KEXT);
      const int jj = t - g;
      const uint32_t xlp = (uint32_t)xa_next | ((uint32_t)xb_next << 16);
      // (clamped, no test: past the window's last column a lane computes cells nobody reads)
      xa_next = lds[oxA + (uint32_t)med3(jj, 0, lastA)];
      xb_next = lds[oxB + (uint32_t)med3(jj, 0, lastB)];
      if (!FIRST || jj >= 1) {                                     // (past its window's last column a lane computes on: nobody reads it)
        uint32_t diag = dg0, insY = upE, mvw = 0;
        static_for<R>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          const uint32_t oldS = S[k];
          // move bits: match strictly best (align_lpo_po2.c:384), x-insertion strictly above y-insertion (:392);
          // En = Sn - (match ? open : ext)
          uint32_t Sn, En, mbit;
          pk_row<2 * k>(xlp, ylp[k], E[k], insY, diag, ONES, KSUB, KEXT, KDELTA, Sn, En, mbit, mvw);
          S[k] = Sn;
          E[k] = En;
          diag = oldS; insY = En;
        });
        dg0 = upS;
        mv[t * 64 + lane] = mvw;
        // the alignment's score: last column, last row -- the lane that holds that row watches for its column
        const bool endA = jj == capA, endB = jj == capB;
        if (endA || endB) {
          static_for<R>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if (endA && k == kstar0) sc1a = pk_half(S[k], 0) - kBias;
            if (endB && k == kstar1) sc1b = pk_half(S[k], 1) - kBias;
          });
        }
      }
    };
    {
      int t = 1;
      // (what the loop below needs of the windows, taken before they go to their slot headers)
      int tcap = 0x3fffffff, inside = 1;
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (needA[h]) {
          tcap = min(tcap, W[h].Lr + (W[h].Lc - 1) / R);
          if (W[h].off_u + tmax + 2 > a.slot_bytes) inside = 0;
        }
      park_win(W[0], g);
      park_win(W[1], g);
      for (; t <= tmax && t < G; ++t) stepA(std::true_type{}, t);
      if constexpr (kUseEngDp1 && Dp1Engine<G, R>::kHave) {
        // the steps from the first in which every lane computes up to the one in which a window's score appears (its last
        // row at its last column): the generated loop, in pairs.  It fetches the reference letters without the clamp of
        // the steps here: every address stays inside the window's slot.
        tcap = -wave_max(-tcap);
        inside = -wave_max(-inside);
        const int t1 = min(tcap, tmax + 1);
        if (inside && t == G && t + 2 <= t1) {
          const int tend = t + ((t1 - t) & ~1);
          EngState st;
          st.a[0] = (uint32_t)xa_next; st.a[1] = (uint32_t)xb_next; st.a[2] = 0u; st.a[3] = 0u;
          // (the letters in hand are those of column t - g: index t - 1 - g, never negative from here on, clamped at the end)
          st.a[4] = oxA + (uint32_t)min(t - 1 - g, lastA);
          st.a[5] = oxB + (uint32_t)min(t - 1 - g, lastB);
          st.a[6] = dg0; st.a[7] = KEXT;
          st.b[0] = bS; st.b[1] = 0u;
          EngConsts cst;
          cst.one = ONES; cst.ksub = KSUB; cst.kext = KEXT; cst.kdelta = KDELTA; cst.kopen = 0u; cst.k16 = 0xFFFFu; cst.psel = 0u;
          uint32_t loff = (uint32_t)(t * 256 + lane * 4);
          Dp1Engine<G, R>::run(ylp, S, E, S2, st, cst, t, tend, loff, mv, __builtin_amdgcn_ballot_w64(g == 0));
          xa_next = (int)st.a[0]; xb_next = (int)st.a[1]; dg0 = st.a[6]; bS = st.b[0];
        }
      }
      for (; t <= tmax; ++t) stepA(std::false_type{}, t);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (the generated loop's stores of moves)
    unpark();
    if (needA[0]) W[0].score1 = sc1a;
    if (needA[1]) W[1].score1 = sc1b;
  }
  // the score sits with the lane that holds the corrected read's last row
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int sc = grp_get<G>(W[h].score1, needA[h] ? (W[h].Lc - 1) / R : 0, lane);
    if (needA[h]) W[h].score1 = sc;
  }
  PK_STAMP(1);
  // ---- traceback #1, fusion #1 (per window), trivial graphs ----
  bool bad[2] = {false, false}, keep[2], farw[2] = {false, false};
  int why[2] = {0, 0}, nfar_dbg[2] = {0, 0};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    IT *x2y = reinterpret_cast<IT *>(U[h] + pk_align_up(W[h].Lr + W[h].Lc, 4));
    if (needA[h]) for (int i = g; i < W[h].Lr; i += G) x2y[i] = (IT)~(IT)0;
  }
  __syncthreads();
  {
    IT *x2ya[2] = {reinterpret_cast<IT *>(U[0] + pk_align_up(W[0].Lr + W[0].Lc, 4)),
                         reinterpret_cast<IT *>(U[1] + pk_align_up(W[1].Lr + W[1].Lc, 4))};
    traceback_a<G, R, IT>(W, needA, mv, lane, g, x2ya);
  }
  __builtin_amdgcn_wave_barrier();
  PK_STAMP(2);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    IT *x2y = reinterpret_cast<IT *>(U[h] + pk_align_up(W[h].Lr + W[h].Lc, 4));
    keep[h] = true;
    if (__builtin_amdgcn_ballot_w64(needA[h]) != 0)
      keep[h] = fusion_1<G, IT, FAR>(W[h], lane, g, U[h], U[h] + W[h].Lr, x2y, xinfo[h], &bad[h], &farw[h], &why[h], &nfar_dbg[h]);
    trivial_graph<G>(W[h], g, U[h], U[h] + W[h].Lr, xinfo[h]);
    if (W[h].valid && W[h].triv == 1) W[h].score1 = W[h].Lr * kp.match;
    if (W[h].valid && W[h].triv == 2) W[h].score1 = (W[h].Lr - 1) * kp.match + kp.mismatch;
    if (W[h].valid && (W[h].triv == 3 || W[h].triv == 4)) W[h].score1 = min(W[h].Lr, W[h].Lc) * kp.match - kp.open_x;
    if (W[h].valid && W[h].triv == 5) W[h].score1 = -(kp.open_x + kp.ext_x * W[h].Lr);
  }
  __syncthreads();
  PK_STAMP(3);
  if (dbg & 64) { W[0].valid = W[1].valid = false; }

  // ---- ordinals of the two-predecessor nodes (they own a row of ordinal bytes); final fit check ----
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const bool on = W[h].valid && keep[h];
    const bool cnt_on = on && W[h].triv == 0;           // the directly written graphs have at most one such node, ordinal 0
    const int n1 = W[h].n1;
    const int cn = (n1 + G - 1) / G;
    const int cnmax = wave_max(cnt_on ? cn : 0);
    const int j0 = 1 + g * cn, j1 = cnt_on ? min(n1 + 1, j0 + cn) : 0;
    int cnt = 0;
    for (int it = 0; it < cnmax; ++it) { const int jj = j0 + it; if (jj < j1) cnt += (xinfo[h][jj] & kN_Has2) != 0; }
    int sc = cnt, scl = 0;
    if (cnmax > 0) {                                   // (wave-uniform: no window of this wavefront came out of fusion #1)
      for_pow2_below<G>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        const int t = grp_up<G, d>(sc);
        if (g >= d) sc += t;
      });
      scl = grp_get<G>(sc, G - 1, lane);
    }
    const int k2n = cnt_on ? scl : W[h].k2n;
    int k = sc - cnt;
    for (int it = 0; it < cnmax; ++it) {
      const int jj = j0 + it;
      if (jj < j1) {
        const uint32_t inf = xinfo[h][jj];
        if (inf & kN_Has2) { xinfo[h][jj] = inf | ((uint32_t)min(k, 255) << 24); ++k; }
      }
    }
    W[h].k2n = k2n;
    const int region = pk_align_up(2 * n1, 4);
    const int cols_need = 2 * W[h].Lu + 4;
    if (on && !(k2n <= 255 && W[h].off_u + region + max(k2n * G, cols_need) <= a.slot_bytes && n1 + G + 2 <= a.mv_tw)) keep[h] = false;
    if (W[h].valid && (!keep[h] || bad[h])) {
      // not for this kernel: a graph with ONE far edge goes to the group's k_poa<G, 8, true> launch (while its list
      // has room), anything else to the two-kernel path, which takes the window from scratch
      if (g == 0) {
        bool placed = false;
        // debug bit 8: why windows leave this kernel (1 nodes beyond the slot's records, 2 broken path, 3 one far edge,
        // 4 several far edges, 5 ordinal rows / moves steps beyond the slot), counted per bin
        if (dbg & 8) atomicAdd(a.stamps + (keep[h] && !bad[h] ? 5 : why[h]), 1ull);
        if ((dbg & 8) && why[h] == 4) atomicAdd(a.stamps + 8 + min(nfar_dbg[h], 7), 1ull);   // (debug: how many far edges: 10 .. 15)
        if (!FAR && farw[h] && !bad[h] && a.far != nullptr) {
          const int at = atomicAdd(a.far_count, 1);
          if (at < a.far_cap) { a.far[at] = W[h].w; placed = true; }
        }
        if (!placed) a.hand[atomicAdd(a.hand_count, 1)] = W[h].w;
      }
      W[h].valid = false;
    }
  }
  // zero guards either side of the node records: alignment #2 reads record clamp(jj + 1, 0, n1 + 1) without a test
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (g == 0) { xinfo[h][0] = 0u; xinfo[h][(W[h].valid ? W[h].n1 : 0) + 1] = 0u; }
  __syncthreads();
  PK_STAMP(4);


  // ================= alignment #2 (graph x linear): the two predecessor columns in registers =================
  uint16_t *x2yb[2];
  uint8_t *ordb[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    x2yb[h] = reinterpret_cast<uint16_t *>(U[h]);
    ordb[h] = U[h] + pk_align_up(2 * W[h].n1, 4);
  }
  int best[2] = {kNeg, kNeg}, bestx[2] = {-1, -1};
  {
    // (the last step some lane needs: the lane of a window's last row at its last node; an even number of steps)
    const int tmax = (wave_max(max(W[0].valid ? W[0].n1 + (W[0].Lu - 1) / R : 0, W[1].valid ? W[1].n1 + (W[1].Lu - 1) / R : 0)) + 1) & ~1;
    // the letters of the lane's rows and the two columns it holds, each a tuple of R registers: the generated loop
    // (poa_engine_gen.h) takes them pinned to fixed registers
    typedef uint32_t VR __attribute__((ext_vector_type(R)));
    VR ylp, S1, E1, S2, E2;
    static_for<R>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const int ii = R * g + 1 + k;
      const int ya = (W[0].valid && ii <= W[0].Lu) ? us[0][ii - 1] : 255;
      const int yb = (W[1].valid && ii <= W[1].Lu) ? us[1][ii - 1] : 255;
      ylp[k] = pk2(ya, yb);
      const int v = -(kp.open_y + (ii - 1) * kp.ext_y);        // column 0 (the virtual start): ii gap steps, not a match
      S1[k] = S2[k] = pk1(v + kBias);
      E1[k] = E2[k] = pk1(v - kp.ext_x + kBias);
    });
    // the row above at column jj - 2 is what the shift delivered as "column jj - 1" one step earlier: one DPP
    // shift per step instead of two (before the first step: column 0 on both sides)
    uint32_t prev_up1 = pk_shift_in<G>(pk1(kBias), S1[R - 1], g);
    const uint32_t colS0 = pk1(-(kp.open_y + (R * g) * kp.ext_y) + kBias);       // column 0 at this lane's first row
    const uint32_t colAbove = pk1((g == 0 ? 0 : -(kp.open_y + (R * g - 1) * kp.ext_y)) + kBias);   // ... at the row above it (the origin for lane 0)
    const uint32_t KOPENNEG = pk1(-kp.open_x + kBias);
    // row -1 over the graph (align_lpo_po2.c:275-286): its score at the column before this one, and what the cells of the
    // two columns before offer a gap (score less the extension; the origin counts as "open").  Only the group's first
    // lane reads them, as the border of the shifts.
    uint32_t BR1 = pk1(kBias), BE1 = pk1(-kp.open_x + kBias), BE2 = pk1(-kp.open_x + kBias);
    const int gstar0 = (W[0].Lu - 1) / R, kstar0 = (W[0].Lu - 1) % R, gstar1 = (W[1].Lu - 1) / R, kstar1 = (W[1].Lu - 1) % R;
    const int n1c0 = (W[0].valid ? W[0].n1 : 0) + 1, n1c1 = (W[1].valid ? W[1].n1 : 0) + 1;   // index of the upper guard
    // node records by their byte offset in LDS: record i of window h at xb_h + 4 i, the upper zero guard at xe_h
    const uint32_t xbA = (uint32_t)(64 + (2 * q) * a.slot_bytes + W[0].off_xi), xbB = (uint32_t)(64 + (2 * q + 1) * a.slot_bytes + W[1].off_xi);
    const uint32_t xeA = xbA + 4u * (uint32_t)n1c0, xeB = xbB + 4u * (uint32_t)n1c1;
    auto rec_at = [&](uint32_t o) { return *reinterpret_cast<const uint32_t *>(lds + o); };
    uint32_t oA = xbA + 4u * (uint32_t)min(max(1 - g, 0), n1c0), oB = xbB + 4u * (uint32_t)min(max(1 - g, 0), n1c1);
    uint32_t xiA_next = rec_at(oA), xiB_next = rec_at(oB);
    // the lane that holds a window's last row watches for nodes that can end the alignment
    const uint32_t finA_bit = (W[0].valid && g == gstar0) ? ((uint32_t)kFlagFinal << 16) : 0u;
    const uint32_t finB_bit = (W[1].valid && g == gstar1) ? ((uint32_t)kFlagFinal << 16) : 0u;
    // ... from the step on at which it can meet the first of them: a final node holds the last letter of the reference or
    // of the corrected sequence, and the node of a sequence's letter i has index i at least (a filler's lone letter comes
    // behind the whole reference).  Up to that step the loop runs without the test.
    int tfin;
    {
      int nf = 0x3fffffff;
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (W[h].valid) nf = min(nf, (W[h].triv == 5 ? W[h].Lr : min(W[h].Lr, W[h].Lc)) + (W[h].Lu - 1) / R);   // index + 1 + lane
      tfin = -wave_max(-nf);
    }

    // k_poa<.., true>: the column of the window's far node (scores, what they offer a gap, the row above it, row -1
    // over the graph) is kept aside when the lane passes it and stands in for a predecessor column at the one node
    // whose record names it (kN_FarA / kN_FarB)
    uint32_t FS[FAR ? R : 1], FE[FAR ? R : 1], Fup = 0, FBE = 0;
    const int fcol0 = (FAR && W[0].valid && W[0].fnode >= 0) ? W[0].fnode + 1 : -100;
    const int fcol1 = (FAR && W[1].valid && W[1].fnode >= 0) ? W[1].fnode + 1 : -100;
    if (FAR) {
#pragma unroll
      for (int k = 0; k < R; ++k) FS[k] = FE[k] = 0;
    }
    int n_steps = 0, n_two = 0, n_virt = 0;                        // debug: steps per code path
    // the two windows' ordinal bytes (this lane's column of them) by LDS offset
    const uint32_t ordA = (uint32_t)(64 + (2 * q) * a.slot_bytes + W[0].off_u + pk_align_up(2 * W[0].n1, 4) + g);
    const uint32_t ordB = (uint32_t)(64 + (2 * q + 1) * a.slot_bytes + W[1].off_u + pk_align_up(2 * W[1].n1, 4) + g);
    const uint32_t kFormBits = kN_Far1 | kN_Far2 | kN_Has2 | kN_Virt1 | (FAR ? kN_FarA | kN_FarB : 0u);
    const uint32_t kFarBits = FAR ? kN_FarA | kN_FarB : 0u;

    // One step = one anti-diagonal: lane g computes column jj = t - g of its R rows, reading column jj - 1 from (Sa, Ea)
    // and writing column jj over column jj - 2 in (Sb, Eb): the rows work IN PLACE (the forms that can name column jj - 2
    // read it first), so no value changes register between steps.  SLOW: the steps in which some lane has not reached its
    // first column yet (t <= G: such a lane keeps column 0) and the steps in which a window's last row can meet a final
    // node (t >= tfin); the steps in between run without either test and fetch their node records through a running
    // offset.  Five forms of the step, each straight code of its own, chosen by wave-wide tests on the records' low bits:
    // NEAR -- every lane at a node whose only predecessor is the node before it (a stretch of plain chain in all windows
    // of the wave): nothing to select; PLAIN -- some lane's predecessor lies two columns back; TWO -- some lane at a node
    // with a second predecessor; VIRT1 -- some lane's predecessor is the virtual start more than two columns back (the
    // lone letter of a filler window), no second predecessors; and everything together (with the far column, FAR).
    auto step = [&](auto slow_tag, int t, VR &Sa, VR &Ea, VR &Sb, VR &Eb) {
      constexpr bool SLOW = decltype(slow_tag)::value;
      const int jj = t - g;
      const uint32_t xiA = xiA_next, xiB = xiB_next;
      if constexpr (SLOW) {
        oA = xbA + 4u * (uint32_t)med3(jj + 1, 0, n1c0);
        oB = xbB + 4u * (uint32_t)med3(jj + 1, 0, n1c1);
      } else {
        oA = min(oA + 4u, xeA);                                      // (LDS offsets reach 160 K: no 16-bit minimum)
        oB = min(oB + 4u, xeB);
      }
      xiA_next = rec_at(oA);
      xiB_next = rec_at(oB);
      // the two letters (byte 1 of each record) to the low bytes of the two halves: one v_perm_b32
      const uint32_t xlp = __builtin_amdgcn_perm(xiB, xiA, 0x0c050c01u);
      const uint32_t xor_ = xiA | xiB;
      const bool active = !SLOW || jj >= 1;                          // before its first column a lane keeps column 0
      uint32_t BRj, BEj, mvw = 0, up1 = 0;
      // the row above: at column jj - 1 (up1), at column jj - 2 (up2, last step's up1), and what it offers a y-gap at column jj
      auto shifts = [&](uint32_t &up2, uint32_t &upE) {
        BEj = pk_subk(BRj, KEXT);
        up1 = pk_shift_in<G>(BR1, Sb[R - 1], g);
        up2 = prev_up1;
        prev_up1 = up1;
        upE = pk_shift_in<G>(BEj, Ea[R - 1], g);
      };
      auto mask16 = [&](int bit) { return bfi(0xFFFFu, 0u - ((xiA >> bit) & 1u), 0u - ((xiB >> bit) & 1u)); };
      // one row, the cell in place (an element of a register tuple cannot be bound to a reference: through locals)
      auto row = [&](auto kc, uint32_t insX, uint32_t iy, uint32_t dm, uint32_t &mb) {
        constexpr int k = decltype(kc)::value;
        uint32_t sn = Sb[k], en = Eb[k];
        pk_row_ip<2 * k>(xlp, ylp[k], insX, iy, dm, ONES, KSUB, KEXT, KDELTA, sn, en, mb, mvw);
        Sb[k] = sn; Eb[k] = en;
      };
      if (__builtin_amdgcn_ballot_w64((xor_ & kFormBits) != 0u) == 0) {
        // ---- NEAR ----
        uint32_t up2, upE;
        BRj = BE1;
        shifts(up2, upE);
        if (active) {
          uint32_t dm = up1, iy = upE, mb;
          static_for<R>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            row(kc, Ea[k], iy, dm, mb);
            dm = Sa[k]; iy = Eb[k];
          });
        }
      } else if (__builtin_amdgcn_ballot_w64((xor_ & (kN_Virt1 | kN_Has2 | kFarBits)) != 0u) == 0) {
        // ---- PLAIN: per half, the predecessor one or two columns back ----
        const uint32_t M1 = mask16(0);
        uint32_t up2, upE;
        BRj = bfi(M1, BE2, BE1);
        shifts(up2, upE);
        if (active) {
          uint32_t dm = bfi(M1, up2, up1), iy = upE, mb;
          static_for<R>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const uint32_t c1S = bfi(M1, Sb[k], Sa[k]), c1E = bfi(M1, Eb[k], Ea[k]);
            row(kc, c1E, iy, dm, mb);
            dm = c1S; iy = Eb[k];
          });
        }
      } else if (__builtin_amdgcn_ballot_w64((xor_ & (kN_Virt1 | kFarBits)) != 0u) == 0) {
        // ---- TWO: some lane at a node with a second predecessor (a node without one repeats the first: bit 1 = bit 0) ----
        const uint32_t M1 = mask16(0), M2 = mask16(1);
        uint32_t up2, upE;
        BRj = pk_max(bfi(M1, BE2, BE1), bfi(M2, BE2, BE1));
        shifts(up2, upE);
        if (active) {
          uint32_t dt1 = bfi(M1, up2, up1), dm = pk_max(dt1, bfi(M2, up2, up1)), iy = upE, secw = 0;
          static_for<R>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            uint32_t sn = Sb[k], en = Eb[k];
            pk_row_two<k>(xlp, ylp[k], M1, M2, Sa[k], Ea[k], iy, ONES, KSUB, KEXT, KDELTA, dt1, dm, sn, en, mvw, secw);
            Sb[k] = sn; Eb[k] = en;
            iy = en;
          });
          if (xiA & kN_Has2) lds[ordA + (xiA >> 24) * G] = (uint8_t)secw;
          if (xiB & kN_Has2) lds[ordB + (xiB >> 24) * G] = (uint8_t)(secw >> 16);
          if (dbg & 4) ++n_two;
        }
      } else if (__builtin_amdgcn_ballot_w64((xor_ & (kN_Has2 | kFarBits)) != 0u) == 0) {
        // ---- VIRT1: some lane's predecessor is the virtual start, more than two columns back: column 0, which depends on
        // the row only ----
        const uint32_t M1 = mask16(0), V1 = mask16(3);
        uint32_t up2, upE;
        BRj = bfi(V1, KOPENNEG, bfi(M1, BE2, BE1));
        shifts(up2, upE);
        if (active) {
          uint32_t dm = bfi(V1, colAbove, bfi(M1, up2, up1)), iy = upE, mb, vcS = colS0;
          static_for<R>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const uint32_t vcE = pk_subk(vcS, KEXT);                  // (= column 0 one row down)
            const uint32_t c1S = bfi(V1, vcS, bfi(M1, Sb[k], Sa[k])), c1E = bfi(V1, vcE, bfi(M1, Eb[k], Ea[k]));
            row(kc, c1E, iy, dm, mb);
            dm = c1S; iy = Eb[k]; vcS = vcE;
          });
          if (dbg & 4) ++n_virt;
        }
      } else {
        // ---- everything: second predecessors, far virtual starts and (FAR) the column kept aside ----
        const bool farv = FAR && __builtin_amdgcn_ballot_w64((xor_ & (kN_FarA | kN_FarB)) != 0u) != 0;
        const uint32_t M1 = mask16(0), M2 = mask16(1), V1 = mask16(3);
        const uint32_t V2 = V1 & ~mask16(2);                         // the repeat of a virtual first predecessor is virtual too
        const uint32_t MA = farv ? mask16(6) : 0u, MB = farv ? mask16(7) : 0u;
        uint32_t up2, upE;
        {
          uint32_t bb1 = bfi(V1, KOPENNEG, bfi(M1, BE2, BE1)), bb2 = bfi(V2, KOPENNEG, bfi(M2, BE2, BE1));
          if (FAR) { bb1 = bfi(MA, FBE, bb1); bb2 = bfi(MB, FBE, bb2); }
          BRj = pk_max(bb1, bb2);
        }
        shifts(up2, upE);
        if (active) {
          uint32_t dt1 = bfi(V1, colAbove, bfi(M1, up2, up1)), dt2 = bfi(V2, colAbove, bfi(M2, up2, up1));
          if (FAR) { dt1 = bfi(MA, Fup, dt1); dt2 = bfi(MB, Fup, dt2); }
          uint32_t iy = upE, secw = 0, vcS = colS0;
          static_for<R>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const uint32_t vcE = pk_subk(vcS, KEXT);
            uint32_t c1S = bfi(V1, vcS, bfi(M1, Sb[k], Sa[k])), c1E = bfi(V1, vcE, bfi(M1, Eb[k], Ea[k]));
            uint32_t c2S = bfi(V2, vcS, bfi(M2, Sb[k], Sa[k])), c2E = bfi(V2, vcE, bfi(M2, Eb[k], Ea[k]));
            if (FAR) {
              c1S = bfi(MA, FS[FAR ? k : 0], c1S); c1E = bfi(MA, FE[FAR ? k : 0], c1E);
              c2S = bfi(MB, FS[FAR ? k : 0], c2S); c2E = bfi(MB, FE[FAR ? k : 0], c2E);
            }
            const uint32_t insX = pk_max(c1E, c2E);                  // first maximum wins (:361-371)
            const uint32_t dmax = pk_max(dt1, dt2);                  // (:348-357)
            uint32_t mb;
            row(kc, insX, iy, dmax, mb);
            const uint32_t pm = pk_bit(pk_sub(dmax, dt1), ONES);     // second predecessor strictly better on the diagonal
            const uint32_t px = pk_bit(pk_sub(insX, c1E), ONES);     // ... for the x-insertion
            secw |= bfi(pk_sub(0u, mb), pm, px) << k;
            pin(secw);
            dt1 = c1S; dt2 = c2S; iy = Eb[k]; vcS = vcE;
          });
          if (xiA & kN_Has2) lds[ordA + (xiA >> 24) * G] = (uint8_t)secw;
          if (xiB & kN_Has2) lds[ordB + (xiB >> 24) * G] = (uint8_t)(secw >> 16);
          if (dbg & 4) ++n_virt;
        }
      }
      if (dbg & 4) ++n_steps;
      if (FAR) {
        // the far node's column, just computed, is kept aside; one step later the row above it arrives
        const bool sA = jj == fcol0, sB = jj == fcol1, uA = jj == fcol0 + 1, uB = jj == fcol1 + 1;
        if (__builtin_amdgcn_ballot_w64(sA || sB || uA || uB) != 0) {
          const uint32_t ms = (sA ? 0xFFFFu : 0u) | (sB ? 0xFFFF0000u : 0u), mu = (uA ? 0xFFFFu : 0u) | (uB ? 0xFFFF0000u : 0u);
          static_for<FAR ? R : 1>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            FS[k] = bfi(ms, Sb[k], FS[k]); FE[k] = bfi(ms, Eb[k], FE[k]);
          });
          FBE = bfi(ms, BEj, FBE);
          Fup = bfi(mu, up1, Fup);
        }
      }
      if (active) {
        BR1 = BRj; BE2 = BE1; BE1 = BEj;
        // every lane stores: a lane past its window's end writes a word nobody reads.  The row's address is uniform.
        typedef __attribute__((address_space(1))) uint32_t gu32;
        gu32 *mrow = (gu32 *)(mv + (size_t)t * 64);
        asm volatile("" : "+s"(mrow));
        mrow[lane] = mvw;
        if constexpr (SLOW) {
          if (((xiA & finA_bit) | (xiB & finB_bit)) != 0u) {
            const bool finA = (xiA & finA_bit) != 0u, finB = (xiB & finB_bit) != 0u;
            static_for<R>([&](auto kc) {
              constexpr int k = decltype(kc)::value;
              if (finA && k == kstar0) { const int v = pk_half(Sb[k], 0) - kBias; if (v > best[0]) { best[0] = v; bestx[0] = jj - 1; } }
              if (finB && k == kstar1) { const int v = pk_half(Sb[k], 1) - kBias; if (v > best[1]) { best[1] = v; bestx[1] = jj - 1; } }
            });
          }
        }
      }
    };
    {
      // pairs of steps (t, t + 1), t odd: (S1, E1) and (S2, E2) swap roles.  G is even: the pairs line up with the phases.
      const int tF = min(tmax + 1, max(G + 1, (tfin - 1) | 1));      // first step of the closing slow phase
      int t = 1;
      // nothing of the windows is wanted from here to the end of the loops: to their slot headers
      park_win(W[0], g);
      park_win(W[1], g);
      // the generated loops (poa_engine_gen.h) and what they take and hand back
      constexpr bool kEngine = !FAR && Dp2Engine<G, R>::kHave;           // (a debug build's step counters see the C++ steps only)
      EngState est;
      EngLane ecn;
      EngConsts ecs;
      const unsigned long long g0mask = __builtin_amdgcn_ballot_w64(g == 0);
      if constexpr (kEngine) {
        ecn.v[0] = xeA; ecn.v[1] = xeB; ecn.v[2] = colS0; ecn.v[3] = colAbove;
        ecn.v[4] = ordA;
        ecn.v[5] = ordB;
        ecn.v[6] = (uint32_t)g; ecn.v[7] = KEXT;
        ecs.one = ONES; ecs.ksub = KSUB; ecs.kext = KEXT; ecs.kdelta = KDELTA; ecs.kopen = KOPENNEG; ecs.k16 = 0xFFFFu;
        ecs.psel = 0x0c050c01u;
      }
      auto eng_in = [&]() {
        est.a[0] = xiA_next; est.a[1] = xiB_next; est.a[2] = 0u; est.a[3] = 0u; est.a[4] = oA; est.a[5] = oB; est.a[6] = BR1; est.a[7] = prev_up1;
        est.b[0] = BE1; est.b[1] = BE2;
      };
      auto eng_out = [&]() {
        xiA_next = est.a[0]; xiB_next = est.a[1]; BR1 = est.a[6]; prev_up1 = est.a[7]; BE1 = est.b[0]; BE2 = est.b[1];
      };
      // behind the steps in which some lane has no column yet -- or, sooner, at the step from which final nodes are watched
      // for (short windows): those steps run here, with the test
      const int tG = min(min(tmax, G) + 1, (tfin - 1) | 1);
      if constexpr (kEngine && kUseEngFirst) {
        while (t < tG) {
          // (the record offsets: record max(t - g, 0) of the step to run; the statement advances them lane by lane)
          oA = xbA + 4u * (uint32_t)min(max(t - g, 0), n1c0);
          oB = xbB + 4u * (uint32_t)min(max(t - g, 0), n1c1);
          eng_in();
          uint32_t loff = (uint32_t)(t * 256 + lane * 4);
          Dp2Engine<G, R>::run_first(ylp, S1, E1, S2, E2, est, ecn, ecs, t, tG, loff, mv, g0mask);
          eng_out();
          if (t < tG) {
            if (t & 1) {
              step(std::true_type{}, t, S1, E1, S2, E2);
              step(std::true_type{}, t + 1, S2, E2, S1, E1);
              t += 2;
            } else {
              step(std::true_type{}, t, S2, E2, S1, E1);
              t += 1;
            }
          }
        }
      }
      for (; t <= tmax && t <= G; t += 2) {
        step(std::true_type{}, t, S1, E1, S2, E2);
        step(std::true_type{}, t + 1, S2, E2, S1, E1);
      }
      // the steps in between: the generated loop (the classes that have one), which leaves at a step it has no form for --
      // that step, or the pair it opens, is then run here -- or the same steps as C++
      auto offsets_at = [&](int tt) {                                // the running record offsets: record min(jj, n1 + 1) of step tt
        oA = xbA + 4u * (uint32_t)min(tt - g, n1c0);
        oB = xbB + 4u * (uint32_t)min(tt - g, n1c1);
      };
      if constexpr (kEngine) {
        while (t < tF) {
          offsets_at(t);
          eng_in();
          uint32_t loff = (uint32_t)(t * 256 + lane * 4);
          Dp2Engine<G, R>::run(ylp, S1, E1, S2, E2, est, ecn, ecs, t, tF, loff, mv, g0mask);
          eng_out();
          if (t < tF) {
            offsets_at(t);
            if (t & 1) {
              step(std::false_type{}, t, S1, E1, S2, E2);
              step(std::false_type{}, t + 1, S2, E2, S1, E1);
              t += 2;
            } else {
              step(std::false_type{}, t, S2, E2, S1, E1);
              t += 1;
            }
          }
        }
      } else if (t < tF) {
        offsets_at(t);
        for (; t < tF; t += 2) {
          step(std::false_type{}, t, S1, E1, S2, E2);
          step(std::false_type{}, t + 1, S2, E2, S1, E1);
        }
      }
      for (; t <= tmax; t += 2) {
        step(std::true_type{}, t, S1, E1, S2, E2);
        step(std::true_type{}, t + 1, S2, E2, S1, E1);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // (the generated loop's stores of moves)
    unpark();
    if ((dbg & 4) && threadIdx.x == 0) {
      atomicAdd(a.stamps + 12, (unsigned long long)n_steps);
      atomicAdd(a.stamps + 13, (unsigned long long)n_two);
      atomicAdd(a.stamps + 14, (unsigned long long)n_virt);
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    x2yb[h] = reinterpret_cast<uint16_t *>(U[h]);
    ordb[h] = U[h] + pk_align_up(2 * W[h].n1, 4);
  }
  PK_STAMP(5);
  if (dbg & 128) { W[0].valid = W[1].valid = false; }
  // best end cell to every lane of the group
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int gs = (W[h].Lu - 1) / R;
    best[h] = grp_get<G>(best[h], W[h].valid ? gs : 0, lane);
    bestx[h] = grp_get<G>(bestx[h], W[h].valid ? gs : 0, lane);
    if (W[h].valid) for (int i = g; i < W[h].n1; i += G) x2yb[h][i] = (uint16_t)kNone16;
  }
  __syncthreads();
  bool badb[2] = {false, false};
  int tb_rounds = 0;
  traceback_b2<G, R, FAR>(W, mv, lane, g, xinfo, ordb, x2yb, bestx, badb, tb_rounds);
  if ((dbg & 4) && threadIdx.x == 0) atomicAdd(a.stamps + 11, (unsigned long long)tb_rounds);
  __builtin_amdgcn_wave_barrier();
  PK_STAMP(6);
  if (dbg & 256) { W[0].valid = W[1].valid = false; }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (__builtin_amdgcn_ballot_w64(W[h].valid) == 0) continue;
    // the columns go straight to the window's place in the output (3 (Lr + Lc + Lu) bytes of room, at most
    // n1 + Lu columns); the column of every uncorrected letter lives where the ordinal bytes were
    uint8_t *cols_st = a.b.cols + 3 * W[h].o0;
    uint16_t *col_y = reinterpret_cast<uint16_t *>(ordb[h]);
    for (int d = 1; d < G; d <<= 1) badb[h] = badb[h] || grp_xor<G>(badb[h] ? 1 : 0, d) != 0;
    if (a.keep_graph && W[h].valid) {
      // a12 (heaviest_bundle.c works on the graph after both fusions): what the search rebuilds that graph from, in the
      // window's node space as k_fused_a / k_fused_b leave it -- letter and flags of node i of the graph after fusion #1
      // (xinfo[i + 1].y; the predecessor word .x is not written: only the alignment kernels of the two-kernel path read
      // it, and they do not run on this window), its ring id (the ring's first node: a ring of two sequences' graph holds
      // two nodes at most, so a node that opens no column follows the one that did) and its x -> y entry
      const int64_t nb = W[h].o0 + (int64_t)W[h].w;
      int2 *gx = a.b.xinfo + nb + 1;
      uint16_t *gr = a.b.ring1 + nb;
      uint32_t *gm = a.b.map16 + nb;
      for (int i = g; i < W[h].n1; i += G) {
        const uint32_t rec = xinfo[h][i + 1], m = x2yb[h][i];
        gx[i].y = (int)(((rec >> 8) & 0x1Fu) | (((rec >> 16) & 0xFu) << 8));
        gr[i] = (uint16_t)((rec & kN_NewCol) ? i : i - 1);
        gm[i] = m == kNone16 ? kNone32 : m;
      }
    }
    const int ncol = columns_2<G>(W[h], lane, g, xinfo[h], x2yb[h], us[h], chr, cols_st, col_y, badb[h]);
    __builtin_amdgcn_wave_barrier();
    if (W[h].valid) {
      if (g == 0) {
        const uint32_t w = W[h].w;
        a.b.ncol[w] = ncol;
        a.b.n1[w] = W[h].n1;
        a.b.score1[w] = W[h].score1;
        a.b.score2[w] = best[h];
        a.b.bx2[w] = bestx[h];
        if (badb[h]) a.b.status[w] = 3;
        a.done_a[w] = 1;
        a.done_b[w] = 1;
      }
    }
  }
  PK_STAMP(7);
  if ((dbg & 4) && threadIdx.x == 0) atomicAdd(a.stamps + 15, 1ull);
  // give the moves slot back: every access of this wave to it has completed
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) {
    const uint32_t t = (uint32_t)atomicAdd(mvq + 16, 1);
    int32_t *e = mvq + 32 + t % (uint32_t)a.mv_slots;
    while (atomicCAS(e, -1, mslot) != -1) __builtin_amdgcn_s_sleep(2);
  }
}

// ---------------------------------------------------------------- launcher ---

// free-slot queues of the moves scratch: every XCD's queue holds the ids 0 .. slots - 1, head 0, tail = slots
__global__ void __launch_bounds__(256) k_poa_pool_init(int32_t *q, int nq, int slots)
{
  for (int x = blockIdx.x; x < nq; x += gridDim.x) {
    int32_t *mvq = q + (size_t)x * kPoolStride;
    if (threadIdx.x == 0) { mvq[0] = 0; mvq[16] = slots; }
    for (int i = threadIdx.x; i < kPoolStride - 32; i += 256) mvq[32 + i] = i < slots ? i : -1;
  }
}

void launch_poa_pool_init(int32_t *q, int nq, int slots, hipStream_t st)
{
  hipLaunchKernelGGL(k_poa_pool_init, dim3((unsigned)nq), dim3(256), 0, st, q, nq, slots);
}

template <int G, int R, bool FAR>
static int launch_poa_t(const PackArgs &a, hipStream_t st)
{
  constexpr int NW = 2 * (64 / G);     // windows per block (one wave)
  static DeviceOnce once;
  if (once.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_poa<G, R, FAR>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024 - 256) != hipSuccess)
      return -1;
    once.done();
  }
  hipLaunchKernelGGL((k_poa<G, R, FAR>), dim3((unsigned)((a.nlist + NW - 1) / NW)), dim3(64), 64 + NW * a.slot_bytes, st, a);
  return 0;
}

#define ELECTOR_PACK_CLASSES(X) \
  X(8, 4) X(8, 5) X(8, 6) X(8, 7) X(8, 8) X(16, 5) X(16, 6) X(16, 7) X(16, 8) \
  X(32, 5) X(32, 6) X(32, 7) X(32, 8) X(64, 5) X(64, 6) X(64, 7) X(64, 8)

int launch_poa(const PackArgs &a, int G, int R, hipStream_t st)
{
  if (a.nlist <= 0) return 0;
#define X(g, r) if (G == g && R == r) return launch_poa_t<g, r, false>(a, st);
  ELECTOR_PACK_CLASSES(X)
#undef X
  return -2;
}

// were k_poa's debug facilities compiled in (-DELECTOR_POA_DEBUG=1)?
bool poa_debug_built() { return kPoaDebug; }

// the far-edge instance of a lane-group size: 8 rows per lane hold every window of the group's classes
int launch_poa_far(const PackArgs &a, int G, hipStream_t st)
{
  if (a.nlist <= 0) return 0;
  if (G == 8) return launch_poa_t<8, 8, true>(a, st);
  if (G == 16) return launch_poa_t<16, 8, true>(a, st);
  if (G == 32) return launch_poa_t<32, 8, true>(a, st);
  if (G == 64) return launch_poa_t<64, 8, true>(a, st);
  return -2;
}

}  // namespace elector
