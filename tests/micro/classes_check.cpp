// Host-side check of elector_amd/csrc/poa_classes.h (compiled by tests/test_classes_cpu.py): the closed forms of the slot
// tiers and geometry classes against the tables they replaced, and window_class() against a plain restatement of the
// host loop of rounds 1-3.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "poa_classes.h"
using namespace elector;
int main()
{
  int tb[61], k = 0, v = 512;
  auto run = [&](int step, int upto) { for (; v <= upto; v += step) tb[k++] = v; };
  run(128, 2048); run(256, 4096); run(512, 8192); run(1024, 16384); run(2048, 32768); run(4096, 65536); run(8192, 131072);
  if (k != kNT) { std::printf("tiers: %d\n", k); return 1; }
  for (int t = 0; t < kNT; ++t) if (tb[t] != tier_bytes(t)) { std::printf("tier %d: %d vs %d\n", t, tb[t], tier_bytes(t)); return 1; }
  for (int64_t need = -5; need < 140000; ++need) {
    const int a = (int)(std::lower_bound(tb, tb + kNT, need) - tb), b = tier_of(need);
    if (a != b) { std::printf("need %ld: tier %d vs %d\n", (long)need, a, b); return 1; }
  }
  const int G[17] = {8, 8, 8, 8, 8, 16, 16, 16, 16, 32, 32, 32, 32, 64, 64, 64, 64}, R[17] = {4, 5, 6, 7, 8, 5, 6, 7, 8, 5, 6, 7, 8, 5, 6, 7, 8};
  for (int c = 0; c < kNC; ++c) if (cls_G(c) != G[c] || cls_R(c) != R[c]) { std::printf("class %d\n", c); return 1; }
  // every window of a sweep lands in a class whose strip holds its longer read and whose slot holds its needs
  KParams kp{}; kp.M = 15; kp.open_x = kp.open_y = 10; kp.ext_x = kp.ext_y = 5; kp.match = 0; kp.mismatch = -10;
  long taken = 0;
  for (int lr = 1; lr < 3000; lr += 7)
    for (int lc = 1; lc < 3000; lc += (lc < 100 ? 3 : 97))
      for (int lu = 1; lu < 3000; lu += (lu < 100 ? 5 : 89)) {
        const WindowClass wc = window_class(kp, lr, lc, lu, -1);
        if (wc.bin < 0) continue;
        ++taken;
        const int ci = wc.bin / kNT, t = wc.bin % kNT;
        const int need = std::max(fused_a_slot_need(lr, lc, cls_G(ci), cls_R(ci)), fused_b_slot_need(lr + lr / 16 + 6, lu, cls_G(ci), cls_R(ci)));
        if (tier_bytes(t) < need || tier_bytes(t) > class_max_slot(ci) || (t > 0 && tier_bytes(t - 1) >= need)) { std::printf("slot %d %d %d\n", lr, lc, lu); return 1; }
        // k_poa's slot needs: what the classification hands the host is what the kernel's fit test computes, and a
        // window that skips alignment #1 (no index maps, records for Lr + 1 nodes) never needs more than one that runs it
        if (wc.need_pack != poa_slot_need(lr, lc, lu, cls_G(ci)) || wc.need_triv != poa_slot_need_triv(lr, lc, lu, cls_G(ci)) ||
            (std::abs(lr - lc) <= 1 && wc.need_triv > wc.need_pack)) { std::printf("k_poa slot %d %d %d\n", lr, lc, lu); return 1; }
        if (ci > 0 && cls_G(ci - 1) * cls_R(ci - 1) >= std::max(lc, lu) && wc.bin >= 0) {
          // an earlier class would have been tall enough: it must have been refused for its slot or its score range
          const int pg = cls_G(ci - 1), pr = cls_R(ci - 1);
          const int pneed = std::max(fused_a_slot_need(lr, lc, pg, pr), fused_b_slot_need(lr + lr / 16 + 6, lu, pg, pr));
          const bool span = score_span(kp, lr + lr / 16 + 6 + pg, ((lu + pg * pr - 1) / (pg * pr)) * (int64_t)(pg * pr)) >= 16000;
          if (!span && pneed <= class_max_slot(ci - 1) && tier_of(pneed) < kNT && tier_bytes(tier_of(pneed)) <= class_max_slot(ci - 1)) {
            std::printf("class skipped %d %d %d\n", lr, lc, lu); return 1;
          }
        }
      }
  std::printf("ok %ld\n", taken);
  return 0;
}
