import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, synth, oracle_lib
from elector_amd.poa import PoaEngine
rng = np.random.default_rng(71)
long_ref = synth.random_seq(rng, 66100)
short = long_ref[30000:31500]
trip = [(long_ref, synth.mutate(rng, short, 0.02), synth.mutate(rng, short, 0.12)),
        (short, synth.mutate(rng, long_ref, 0.02), synth.mutate(rng, short, 0.12)),
        (short, synth.mutate(rng, short, 0.02), synth.mutate(rng, long_ref, 0.12))]
eng = PoaEngine(0)
bases, off = synth.pack_windows(trip)
rows, row_off, ncol, status, scores = eng.align_packed(np.frombuffer(bases, dtype=np.uint8), off, want_scores=True, strict=False)
print("status", status, "ncol", ncol, "scores", scores.tolist())
exp_rows, exp_ncol, exp_scores, _ = oracle_lib.batch(np.frombuffer(bases, dtype=np.uint8), off)
print("exp ncol", exp_ncol, exp_scores.tolist())
