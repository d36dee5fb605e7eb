"""GPU-box helper: device splitter time (k_split alone, from the library's own host view) and parity with the host
splitter on the bench batch.  Usage: python tools/_r3_split.py [profile] [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ELECTOR_DEBUG_SPLIT"] = "1"
import numpy as np
from elector_amd import split, synthetic
from elector_amd.poa import PoaEngine
prof = sys.argv[1] if len(sys.argv) > 1 else "ecoli30x_simlord_lordec"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10001
eng = PoaEngine(0)
tr, hd, ro = synthetic.read_pieces(prof, n, 1000)
nb = sum(len(t[0]) for t in tr)
buf, off, hl = split.pack_reads(tr, hd)
split.split_packed_device(eng, buf[: off[3 * 200]], off[: 3 * 200 + 1], hl[:200], 0.1)
for _ in range(3):
    t0 = time.perf_counter(); d = split.split_packed_device(eng, buf, off, hl, 0.1); t1 = time.perf_counter()
    print(prof, "reads", len(tr), "Mbases %.1f" % (nb / 1e6), "device call %.1f ms" % ((t1 - t0) * 1e3), flush=True)
h = split.split_packed(buf, off, hl, 0.1, nthreads=16)
same = (d.n_windows == h.n_windows and np.array_equal(d.off, h.off) and np.array_equal(d.read_first, h.read_first) and
        np.array_equal(d.read_index, h.read_index) and d.small_reads == h.small_reads and d.wrong_reads == h.wrong_reads)
if same and isinstance(d.d_bases, split.DevBases):
    same = np.array_equal(d.d_bases.numpy(), h.bases)
print("windows", d.n_windows, "parity with the host splitter:", "OK" if same else "DIFFERENT", flush=True)
sys.exit(0 if same else 1)
