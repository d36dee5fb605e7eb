"""GPU-box helper: hand-back counts, ring-depth histogram and per-kernel event times of one workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elector_amd import split, synthetic
from elector_amd.poa import PoaEngine
prof = sys.argv[1] if len(sys.argv) > 1 else "yeast50x_nanosim_consent_split"
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
tr, hd, ro = synthetic.read_pieces(prof, nreads, 1000)
win = split.split_reads(tr, 0.1, hd, nthreads=16)
dev = torch.device("cuda", 0)
d_bases = torch.from_numpy(win.bases).to(dev); n = win.n_windows
d_cols = torch.empty(3*int(win.off[-1])+64, dtype=torch.uint8, device=dev)
d_ncol = torch.empty(n, dtype=torch.int32, device=dev); d_status = torch.empty(n, dtype=torch.int32, device=dev)
os.environ["ELECTOR_DEBUG_FUSED"] = os.environ.get("DBG", "36")
os.environ["ELECTOR_DEBUG_BINS"] = "1"
eng = PoaEngine(0)
eng.option("chains", 1)
for _ in range(2): eng.align_device(d_bases, win.off, d_cols, d_ncol, d_status)
eng.sync()
print("windows", n, flush=True)
