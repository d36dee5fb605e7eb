// elector_amd/csrc/poa_classes.h -- launch classes of the fused kernels, shared by the host (poa_host.hip) and the
// device-side classification (poa_classify.hip): which geometry class and LDS slot tier a window of lengths
// (Lr, Lc, Lu) goes to.  Everything here is integer arithmetic on the three lengths and the scoring constants.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "poa_device.h"

namespace elector {

// fused-kernel geometry classes: G lanes per window x R rows per lane, ordered by the rows one strip holds (must match
// ELECTOR_FUSED_CLASSES in poa_fused.hip and ELECTOR_PACK_CLASSES in poa_pack.hip).  A window goes to the first class
// whose strip is at least as tall as its longer read, so few lanes idle.
//   class  0..4 : G = 8,  R = 4..8        class  5..8 : G = 16, R = 5..8
//   class  9..12: G = 32, R = 5..8        class 13..16: G = 64, R = 5..8
constexpr int kNC = 17;
__host__ __device__ constexpr int cls_G(int ci) { return ci < 5 ? 8 : 16 << ((ci - 5) >> 2); }
__host__ __device__ constexpr int cls_R(int ci) { return ci < 5 ? 4 + ci : 5 + ((ci - 5) & 3); }

// LDS slot sizes (bytes per window): a launch bin is (geometry class, slot tier).  512 .. 2048 in steps of 128, then
// eight tiers per octave: (2048 << o) + 128 + i * (256 << o), o = 0 .. 5, i = 0 .. 7 (the last is 123008).
constexpr int kNT = 61;
__host__ __device__ constexpr int tier_bytes(int t)
{
  return t <= 12 ? 512 + 128 * t : (2048 << ((t - 13) >> 3)) + 128 + (256 << ((t - 13) >> 3)) * ((t - 13) & 7);
}
// first tier whose slot holds `need` bytes (kNT: none)
__host__ __device__ inline int tier_of(int64_t need)
{
  if (need <= 512) return 0;
  if (need <= 2048) return (int)((need - 512 + 127) >> 7);
  if (need > tier_bytes(kNT - 1)) return kNT;
  int o = 0;                                           // octave whose last tier holds the need
  while ((((int64_t)3840) << o) + 128 < need) ++o;
  const int64_t start = ((int64_t)2048 << o) + 128, step = (int64_t)256 << o;
  return 13 + 8 * o + (need <= start ? 0 : (int)((need - start + step - 1) / step));
}
constexpr int kBins = kNC * kNT;
constexpr int kKeys = 256;              // size keys of the list sort (largest window first)

// largest slot a class can give each of its 64/G windows (k_fused_b: 64 B table + one score ring per wave)
__host__ __device__ inline int class_max_slot(int ci)
{
  const int nw = 64 / cls_G(ci);
  return ((160 * 1024 - 256 - 64 - fused_ring_bytes(cls_R(ci), 8)) / nw) & ~127;
}

// Window status by its lengths (include/elector_poa.h: ELECTOR_W_*): 0 ok, 1 empty, 2 too long
__host__ __device__ inline int window_status(int64_t lr, int64_t lc, int64_t lu, int pen_abs_max, int64_t max_seq,
                                             int64_t window_moves_max_dwords)
{
  if (lr == 0 || lc == 0 || lu == 0) return 1;
  if (lr > max_seq || lc > max_seq || lu > max_seq || (int64_t)pen_abs_max * (lr + lc + lu + 4) >= ((int64_t)1 << 24) ||
      (int64_t)n_strips((int)lc) * mv_tw((int)lr) * 64 + (int64_t)n_strips((int)lu) * mv_tw((int)(lr + lc)) * 64 >
          window_moves_max_dwords)
    return 2;
  return 0;
}

// Size key of the lists (largest first).  Inside a geometry class the rows per lane are fixed and a wavefront runs for
// as many steps as its longest window has columns: the exact reference length as the key makes the wavefronts of a list
// homogeneous in steps (the device's stable partition by the trivial-window key keeps this order inside each of its
// buckets).  Long windows share keys 16 apart; `coarse`: max(Lr, Lu) / 8 as up to round 2 (A/B).
__host__ __device__ inline int window_size_key(int64_t lr, int64_t lu, bool coarse)
{
  int k = coarse ? (int)((lr > lu ? lr : lu) >> 3) : lr < 192 ? (int)lr : 192 + (int)(((lr - 192) >> 4) < 63 ? ((lr - 192) >> 4) : 63);
  if (k >= kKeys) k = kKeys - 1;
  return kKeys - 1 - k;
}

struct WindowClass { int bin, need_a, need_pack, need_triv; };   // bin < 0: no fused class takes the window (generic path)

// One class for both fused kernels and k_poa; |PO| is not known yet: typical growth estimate, windows whose graph turns
// out larger are handed back by the device.  force_cls >= 0 (testing): that geometry class or none.
__host__ __device__ inline WindowClass window_class(const KParams &kp, int64_t lr, int64_t lc, int64_t lu, int force_cls)
{
  WindowClass r{-1, 0, 0, 0};
  const int rows = (int)(lc > lu ? lc : lu);
  int c0 = 0;
  while (c0 < kNC - 1 && cls_G(c0) * cls_R(c0) < rows) ++c0;
  if (force_cls >= 0 && force_cls < kNC) c0 = force_cls;
  for (int ci = c0; ci < kNC; ++ci) {
    const int G = cls_G(ci), R = cls_R(ci);
    // k_fused_b's 16-bit ring cells hold scores up to about +-16000 (it hands larger windows back)
    // (32-bit arithmetic: the lengths are below ELECTOR_MAX_SEQ here, and a 64-bit division costs a hundred instructions)
    if (score_span(kp, (int)lr + (int)lr / 16 + 6 + G, (((int)lu + G * R - 1) / (G * R)) * (G * R)) >= 16000) continue;
    const int need_a = fused_a_slot_need((int)lr, (int)lc, G, R);
    const int nb = fused_b_slot_need((int)(lr + lr / 16 + 6), (int)lu, G, R);
    const int need = need_a > nb ? need_a : nb;
    const int cmax = class_max_slot(ci);
    if (need > cmax) {
      if (force_cls >= 0) break;
      continue;                                                    // a class with fewer windows per wave has larger slots
    }
    const int t = tier_of(need);
    if (t >= kNT) break;
    if (tier_bytes(t) > cmax) {
      if (force_cls >= 0) break;
      continue;
    }
    r.bin = ci * kNT + t;
    r.need_a = need_a;
    r.need_pack = poa_slot_need((int)lr, (int)lc, (int)lu, G);
    r.need_triv = poa_slot_need_triv((int)lr, (int)lc, (int)lu, G);
    break;
  }
  return r;
}

// ---- device-side classification and list sort (poa_classify.hip) ----
constexpr int kAccRows = 8;             // per bin: count, then the maxima of need_a, Lr, Lc, Lu, Lr + Lc, need_pack, need_triv
constexpr int kSortDestMax = 40;        // lists a batch may have (launch bins + the generic list)

struct ClassifyArgs {
  int64_t n, total;
  const int64_t *off;
  KParams kp;
  int pen_abs_max, use_fused, force_cls, coarse;
  int64_t window_moves_max;
  int32_t *status;
  int16_t *bin;
  uint8_t *wkey;
  int32_t *acc;                         // [kAccRows][kBins]
  unsigned long long *glob;             // generic windows, moves if every fused window were handed back, max Lr + Lc, bad offsets
};

struct SortArgs {
  int64_t n;
  const int16_t *bin;
  const uint8_t *wkey;
  const int16_t *dest_of;               // [kBins + 1]: list of a bin (entry kBins: the generic list)
  int ndest;                            // lists, the generic one included (last)
  uint32_t *hist;                       // [ndest][kKeys] counts, then cursors
  const int64_t *dest_first;            // [ndest] first entry of a list (the generic list: 0, it lives in an array of its own)
  uint32_t *lists, *generic;
};

void launch_classify(const ClassifyArgs &a, hipStream_t st);
// bytes (4-byte aligned both sides) from device memory into page-locked host memory by a kernel's stores; -1 when the
// host address is not device-addressable
int launch_words_to_host(void *host_dst, const void *src, size_t bytes, hipStream_t st);
int launch_sort(const SortArgs &a, hipStream_t st);
void launch_generic_info(const uint32_t *glist, int64_t ng, const int64_t *off, const int32_t *status, int32_t *info, hipStream_t st);
void launch_generic_moves(const uint32_t *glist, int64_t ng, const int64_t *gmv, int64_t *mv1, int64_t *mv2, hipStream_t st);

}  // namespace elector
