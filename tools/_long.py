"""Timing of the long-window path (no parity check here: tests/test_poa_gpu.py does that)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import synth
from elector_amd.poa import PoaEngine
L = int(sys.argv[1]) if len(sys.argv) > 1 else 17000
t = synth.window_triples(51, 1, L, L, err_unc=0.12, err_cor=0.02)
eng = PoaEngine(0)
eng.align(t)
t0 = time.perf_counter(); rows = eng.align(t); dt = time.perf_counter() - t0
print("L=%d tile_cells=%s: %.3f s per window (%d columns)" % (L, os.environ.get("ELECTOR_TILE_CELLS", "default"), dt, len(rows[0][0])))
