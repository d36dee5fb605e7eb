"""PCIe-inclusive rate of the host-buffer entry (elector_poa_batch): bases in host memory in,
compact MSA rows in host memory out.  Printed for DESIGN.md section 6; never bench.py's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from elector_amd import split, synthetic
from elector_amd.poa import PoaEngine
reads = synthetic.read_triples("ecoli30x_simlord_lordec", 4000, seed=1000)
nb = sum(len(r[0]) for r in reads)
win = split.split_reads(reads, 0.1, None, nthreads=64)
eng = PoaEngine(0)
for _ in range(2):
    eng.align_packed(win.bases, win.off)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); rows, row_off, ncol, status, _ = eng.align_packed(win.bases, win.off); ts.append(time.perf_counter() - t0)
t = min(ts)
print("elector_poa_batch (host buffers): %.2f ms per %d reads (%d windows, %.1f MB in, %.1f MB rows out) = %.0f Mbases/s PCIe-inclusive"
      % (t * 1e3, len(reads), win.n_windows, win.off[-1] / 1e6, row_off[-1] / 1e6, nb / t / 1e6))
