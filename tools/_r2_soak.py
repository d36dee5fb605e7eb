"""GPU-box helper: random windows (many with one-indel / one-substitution corrected sequences) through the C ABI
against the oracle, bit-exact rows and scores.  Usage: python tools/_r2_soak.py [seeds...]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle_lib, synth
from elector_amd import poa

eng = poa.PoaEngine(0)
seeds = [int(x) for x in sys.argv[1:]] or [101, 102, 103]
total = bad = 0
for seed in seeds:
    for (n, lo, hi, eu, ec) in ((30000, 20, 90, 0.15, 0.01), (20000, 30, 140, 0.12, 0.02), (6000, 100, 400, 0.15, 0.015), (20000, 5, 40, 0.2, 0.03)):
        triples = synth.window_triples(seed, n, lo, hi, err_unc=eu, err_cor=ec)
        bases, off = synth.pack_windows(triples)
        t0 = time.time()
        exp_rows, exp_ncol, exp_scores, _ = oracle_lib.batch(np.frombuffer(bases, dtype=np.uint8), off)
        t1 = time.time()
        got, scores = eng.align(triples, want_scores=True)
        nb = sum(1 for w in range(len(triples)) if got[w] != exp_rows[w]) + int((scores != exp_scores).any(axis=1).sum())
        total += len(triples); bad += nb
        print("seed", seed, "windows", len(triples), "len", lo, hi, "err", eu, ec, "differing", nb, "oracle %.1fs gpu %.1fs" % (t1 - t0, time.time() - t1), flush=True)
print("TOTAL", total, "differing", bad)
sys.exit(1 if bad else 0)
