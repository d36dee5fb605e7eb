import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import numpy as np
import synth, oracle_lib
from elector_amd.poa import PoaEngine
eng = PoaEngine(0)
tot = 0
for seed in (211,):
    for (n, lo, hi, eu, ec) in ((30000, 20, 90, 0.15, 0.01), (20000, 30, 140, 0.12, 0.02)):
        triples = synth.window_triples(seed, n, lo, hi, err_unc=eu, err_cor=ec)
        bases, off = synth.pack_windows(triples)
        exp_rows, exp_ncol, exp_scores, _ = oracle_lib.batch(np.frombuffer(bases, dtype=np.uint8), off)
        for rep in range(2):
            got, scores = eng.align(triples, want_scores=True)
            bad = [w for w in range(len(triples)) if got[w] != exp_rows[w]]
            eq = sum(1 for w in bad if triples[w][0] == triples[w][1])
            L = [len(triples[w][0]) for w in bad]
            print("n", n, "rep", rep, "bad", len(bad), "cor==ref among bad", eq, "Lr min/max", (min(L), max(L)) if L else None, "first", bad[:8])
            if bad:
                import collections
                print("  Lr histogram of bad (by 8):", sorted(collections.Counter(l // 8 * 8 for l in L).items()))
