"""GPU-box helper: device splitter against the host splitter over several seeds and profiles (more reads than the
test suite's soak).  Usage: python tools/_r3_split_soak.py [reads per case]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from elector_amd import split, synthetic
from elector_amd.poa import PoaEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
eng = PoaEngine(0)
bad = 0
for prof, scale in (("ecoli30x_simlord_lordec", 1.0), ("yeast50x_nanosim_consent", 1.0), ("yeast50x_nanosim_consent_split", 1.0),
                    ("celegans30x_simlord_mixed", 1.0), ("chr1_20x_ont_50kb", 0.1)):
    for seed in (11, 12, 13, 14):
        tr, hd, _ = synthetic.read_pieces(prof, max(50, int(n * scale)), seed)
        buf, off, hl = split.pack_reads(tr, hd)
        d = split.split_packed_device(eng, buf, off, hl, 0.1)
        h = split.split_packed(buf, off, hl, 0.1, nthreads=32)
        same = (d.n_windows == h.n_windows and np.array_equal(d.off, h.off) and np.array_equal(d.read_first, h.read_first) and
                np.array_equal(d.read_index, h.read_index) and d.small_reads == h.small_reads and d.wrong_reads == h.wrong_reads)
        if same and isinstance(d.d_bases, split.DevBases):
            same = np.array_equal(d.d_bases.numpy(), h.bases)
        print(prof, seed, "reads", len(tr), "windows", d.n_windows, "device path" if isinstance(d.d_bases, split.DevBases) else "HOST FALLBACK", "OK" if same else "DIFFERENT", flush=True)
        bad += not same
sys.exit(1 if bad else 0)
