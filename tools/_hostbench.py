import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elector_amd import split, synthetic
from elector_amd.poa import PoaEngine
reads = synthetic.read_triples("ecoli30x_simlord_lordec", 4000, seed=1000)
win = split.split_reads(reads, 0.1, None, nthreads=64)
dev = torch.device("cuda", 0)
d_bases = torch.from_numpy(win.bases).to(dev); n = win.n_windows
d_cols = torch.empty(3*int(win.off[-1])+64, dtype=torch.uint8, device=dev)
d_ncol = torch.empty(n, dtype=torch.int32, device=dev); d_status = torch.empty(n, dtype=torch.int32, device=dev)
eng = PoaEngine(0)
for _ in range(2): eng.align_device(d_bases, win.off, d_cols, d_ncol, d_status)
eng.sync()
te=[]; tt=[]
for _ in range(5):
    t0=time.perf_counter(); eng.align_device(d_bases, win.off, d_cols, d_ncol, d_status); t1=time.perf_counter(); eng.sync(); t2=time.perf_counter()
    te.append((t1-t0)*1e3); tt.append((t2-t0)*1e3)
print("enqueue ms", np.round(te,2), "total ms", np.round(tt,2))
