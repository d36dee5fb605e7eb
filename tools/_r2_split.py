"""GPU-box helper: device splitter throughput against the host splitter."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from elector_amd import split, synthetic
from elector_amd.poa import PoaEngine
eng = PoaEngine(0)
for prof, n in (("ecoli30x_simlord_lordec", 10001), ("yeast50x_nanosim_consent_split", 10001), ("chr1_20x_ont_50kb", 1500)):
    tr, hd, ro = synthetic.read_pieces(prof, n, 1000)
    nb = sum(len(t[0]) for t in tr)
    split.split_reads_device(eng, tr[:200], 0.1, hd[:200])
    t0 = time.perf_counter(); d = split.split_reads_device(eng, tr, 0.1, hd); t1 = time.perf_counter()
    buf, off, hl = split.pack_reads(tr, hd)
    t2 = time.perf_counter(); d2 = split.split_reads_device(eng, tr, 0.1, hd); t3 = time.perf_counter()
    h = split.split_reads(tr, 0.1, hd, nthreads=16); t4 = time.perf_counter()
    print(prof, "reads", len(tr), "Mbases", nb/1e6, "device %.3f s (%.0f Mbases/s; incl. python packing)" % (t3-t2, nb/(t3-t2)/1e6),
          "host16 %.3f s (%.0f Mbases/s)" % (t4-t3, nb/(t4-t3)/1e6), "windows", d.n_windows, h.n_windows,
          "devpath", isinstance(d.d_bases, split.DevBases), flush=True)
