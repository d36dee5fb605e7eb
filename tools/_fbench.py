import os, sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from elector_amd import split, synthetic
from elector_amd.poa import PoaEngine
reads = synthetic.read_triples("ecoli30x_simlord_lordec", 2000, seed=1000)
win = split.split_reads(reads, 0.1, None, nthreads=64)
dev = torch.device("cuda", 0)
d_bases = torch.from_numpy(win.bases).to(dev); n = win.n_windows
d_cols = torch.empty(3*int(win.off[-1])+64, dtype=torch.uint8, device=dev)
d_ncol = torch.empty(n, dtype=torch.int32, device=dev); d_status = torch.empty(n, dtype=torch.int32, device=dev)
for dbg in [int(x) for x in os.environ.get("FB_DEBUGS","0,1,2,3").split(",")]:
    os.environ["ELECTOR_DEBUG_FUSED"] = str(dbg)
    eng = PoaEngine(0)
    for _ in range(2): eng.align_device(d_bases, win.off, d_cols, d_ncol, d_status)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(5): eng.align_device(d_bases, win.off, d_cols, d_ncol, d_status)
    eng.sync()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    eng.timing_enable(True); eng.timing_reset()
    for _ in range(3): eng.align_device(d_bases, win.off, d_cols, d_ncol, d_status)
    eng.sync()
    print("debug", dbg, "wall %.3f ms/step" % wall, "k_a %.3f ms/step" % (eng.timing_read(0)[0]/3), "dp2 %.3f" % (eng.timing_read(1)[0]/3), "other %.3f" % (eng.timing_read(2)[0]/3), flush=True)
    eng.close()
