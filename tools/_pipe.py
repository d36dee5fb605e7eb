"""Where does the host spend a bench step?  Per-call wall times of the pipelined loop of bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elector_amd import split, synthetic
from elector_amd.poa import PoaEngine
reads = synthetic.read_triples("ecoli30x_simlord_lordec", 4000, seed=1000)
win = split.split_reads(reads, 0.1, None, nthreads=64)
dev = torch.device("cuda", 0)
d_bases = torch.from_numpy(win.bases).to(dev); n = win.n_windows; off = win.off
d_cols = torch.empty(3*int(off[-1])+64, dtype=torch.uint8, device=dev)
d_ncol = torch.empty(n, dtype=torch.int32, device=dev); d_status = torch.empty(n, dtype=torch.int32, device=dev)
eng = PoaEngine(0)
if os.environ.get("TIMING"): eng.timing_enable(True)
piece_first = win.read_first; read_first = np.arange(win.n_reads + 1, dtype=np.int64)
pending = []
rows = []
t00 = time.perf_counter()
for i in range(10):
    t0 = time.perf_counter(); eng.align_device(d_bases, off, d_cols, d_ncol, d_status)
    t1 = time.perf_counter(); pending.append(eng.msa_stats_enqueue(n, d_cols, d_ncol, d_status, piece_first, read_first))
    t2 = time.perf_counter()
    if len(pending) > 1: eng.msa_stats_collect(pending.pop(0))
    t3 = time.perf_counter()
    rows.append(((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))
while pending: eng.msa_stats_collect(pending.pop(0))
eng.sync(); t9 = time.perf_counter()
for r in rows: print("align_device %.2f  stats_enqueue %.2f  collect %.2f ms" % r)
print("total %.2f ms per step" % ((t9 - t00) / 10 * 1e3))
