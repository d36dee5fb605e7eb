# What the stages behind the alignment cost INSIDE the four-context pipeline (not what they last alone): ms per step of
# three rotated batches with the alignment kernels only / + merge and counters / + the rows to page-locked host memory.
# Usage (GPU box): [ELECTOR_LIB=path/to/other/build.so] [MODES=2,1,0,2,1,0] python tools/pipeline_cost.py [profile]
# (round 5: merge + counters cost the step about what they last un-overlapped -- any wavefront of theirs on a SIMD takes
# the place of one of k_poa's three -- so they are A/B-ed here, alternating two builds on one box)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from elector_amd.poa import PoaEngine
prof = sys.argv[1] if len(sys.argv) > 1 else "yeast50x_nanosim_consent_split"
steps = int(os.environ.get("STEPS", "40"))
bs = [bench.prepare_batch((prof, 10001, 1000 + k, bench.cpu_share(), None)) for k in range(3)]
dev = torch.device("cuda", 0)
for b in bs:
    b.d_bases = torch.from_numpy(b.win.bases).to(dev)
    b.d_off = torch.from_numpy(np.ascontiguousarray(b.win.off, dtype=np.int64)).to(dev)
mt = max(b.total for b in bs); mn = max(b.n for b in bs)
NE = 4
eng = [PoaEngine(0) for _ in range(NE)]
outs = [(torch.empty(3 * mt + 64, dtype=torch.uint8, device=dev), torch.empty(mn, dtype=torch.int32, device=dev),
         torch.empty(mn, dtype=torch.int32, device=dev)) for _ in range(NE)]
cap = 3 * mt + 64
pinned = [[torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(2)] for _ in range(NE)]
turn = [0] * NE
def run(mode, steps):
    pend = []
    t0 = time.perf_counter()
    for s in range(steps):
        e = s % NE; b = bs[s % 3]
        if len(pend) >= NE:
            pe, pn = pend.pop(0)
            if mode: eng[pe].msa_stats_collect(pn)
            else: eng[pe].sync()
        dc, dn, ds = outs[e]
        eng[e].align_device_offsets(b.d_bases, b.d_off, b.n, b.total, dc, dn, ds)
        npc = 0
        if mode == 1: npc = eng[e].msa_stats_enqueue(b.n, dc, dn, ds, b.piece_first, b.read_first)
        if mode == 2:
            turn[e] ^= 1
            npc = eng[e].msa_stats_enqueue(b.n, dc, dn, ds, b.piece_first, b.read_first, rows_out=pinned[e][turn[e]].data_ptr(), rows_cap=cap)
        pend.append((e, npc))
    for pe, pn in pend:
        if mode: eng[pe].msa_stats_collect(pn)
        else: eng[pe].sync()
    if mode == 2:
        for g in eng: g.msa_rows_wait()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
out = []
for mode in [int(x) for x in os.environ.get("MODES", "2,1,0,2,1,0").split(",")]:
    run(mode, 8)
    out.append("%s %.3f" % (("align", "+counters", "+rows")[mode], run(mode, steps)))
print(prof, os.path.basename(os.environ.get("ELECTOR_LIB", "cur")), " | ".join(out), flush=True)
