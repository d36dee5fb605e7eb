"""GPU-box helper : parity soak of the final build through the C ABI against the oracle -- the random window mixes of
rounds 2-3 plus far-edge windows (tests/test_poa_gpu.py: far_edge_triples) and noisier corrected sequences (4-6 % error:
several far edges, deep graphs), rows and both scores bit-exact.  Usage: python tools/parity_soak.py [--bundles] [seeds...]
--bundles: the batches run with elector_ctx_keep_graph and the heaviest-bundle search (a12) on every window is compared too
(consensus rows, counts, bundle ids against the oracle's heaviest_bundle restatement) -- k_poa's graph output."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle_lib, synth
from elector_amd import poa
src = open(os.path.join(ROOT, "tests", "test_poa_gpu.py")).read()
exec(src[src.index("def far_edge_triples"):src.index("def test_far_edge_windows(engine)")])

eng = poa.PoaEngine(0)
BUNDLES = "--bundles" in sys.argv[1:]
seeds = [int(x) for x in sys.argv[1:] if x != "--bundles"] or [201, 202]
total = bad = 0
def run(tag, triples):
    global total, bad
    bases, off = synth.pack_windows(triples)
    t0 = time.time()
    exp_rows, exp_ncol, exp_scores, _ = oracle_lib.batch(np.frombuffer(bases, dtype=np.uint8), off)
    t1 = time.time()
    if BUNDLES:
        exp_b = oracle_lib.batch_bundles(np.frombuffer(bases, dtype=np.uint8), off, 0.9)
        t1 = time.time()
        got, got_b = eng.align_with_bundles(triples, 0.9)
        nb = sum(1 for w in range(len(triples)) if got[w] != exp_rows[w] or got_b[w] != exp_b[w])
    else:
        got, scores = eng.align(triples, want_scores=True)
        nb = sum(1 for w in range(len(triples)) if got[w] != exp_rows[w]) + int((scores != exp_scores).any(axis=1).sum())
    total += len(triples); bad += nb
    print(tag, "windows", len(triples), "differing", nb, "oracle %.1fs gpu %.1fs" % (t1 - t0, time.time() - t1), flush=True)
for seed in seeds:
    for (n, lo, hi, eu, ec) in ((30000, 20, 90, 0.15, 0.01), (20000, 30, 140, 0.12, 0.02), (6000, 100, 400, 0.15, 0.015), (20000, 5, 40, 0.2, 0.03),
                                (12000, 30, 200, 0.12, 0.05), (4000, 150, 500, 0.15, 0.04)):
        run("seed %d len %d-%d err %.2f/%.3f" % (seed, lo, hi, eu, ec), synth.window_triples(seed, n, lo, hi, err_unc=eu, err_cor=ec))
    run("seed %d far edges" % seed, far_edge_triples(seed, 12000))
print("TOTAL", total, "differing", bad)
sys.exit(1 if bad else 0)
