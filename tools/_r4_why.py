"""Why windows leave k_poa: one batch of a profile with ELECTOR_DEBUG_FUSED=8 / ELECTOR_DEBUG_BINS.  Usage: python tools/_r4_why.py [profile] [reads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ELECTOR_DEBUG_FUSED"] = "8"
os.environ["ELECTOR_DEBUG_BINS"] = "1"
import numpy as np, torch
from elector_amd import split, synthetic
from elector_amd.poa import PoaEngine
prof = sys.argv[1] if len(sys.argv) > 1 else "yeast50x_nanosim_consent_split"
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 10001
triples, headers, read_of = synthetic.read_pieces(prof, reads, seed=1000)
win = split.split_reads(triples, 0.1, headers, nthreads=16)
dev = torch.device("cuda", 0)
eng = PoaEngine(0)
eng.option("chains", 1)
d_bases = torch.from_numpy(win.bases).to(dev)
n = win.n_windows
d_cols = torch.zeros(3 * int(win.off[-1]) + 64, dtype=torch.uint8, device=dev)
d_ncol = torch.empty(n, dtype=torch.int32, device=dev); d_status = torch.empty(n, dtype=torch.int32, device=dev)
eng.align_device(d_bases, win.off, d_cols, d_ncol, d_status)
eng.sync()
