"""debug: which windows does the new k_poa lose?  reference = the two-kernel path in the same process"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elector_amd import split, synthetic
from elector_amd.poa import PoaEngine
eng = PoaEngine(0)
reads = synthetic.read_triples("ecoli30x_simlord_lordec", 300, seed=5)
win = split.split_reads(reads, 0.1, None, nthreads=8)
rows, row_off, ncol, status, _ = eng.align_packed(win.bases, win.off)
rows = rows.copy(); ncol = ncol.copy(); row_off = row_off.copy()
os.environ["ELECTOR_NO_PACK"] = "1"
rows0, row_off0, ncol0, status0, _ = eng.align_packed(win.bases, win.off)
del os.environ["ELECTOR_NO_PACK"]
off = win.off
print("windows", win.n_windows, "ncol differs", int((ncol != ncol0).sum()), "row_off differs", int((row_off != row_off0).sum()))
r, r0 = rows.tobytes(), rows0.tobytes()
bad = [w for w in range(win.n_windows) if r[int(row_off[w]):int(row_off[w]) + 3 * int(ncol[w])] != r0[int(row_off0[w]):int(row_off0[w]) + 3 * int(ncol0[w])]]
print("bad", len(bad))
L = np.diff(off).reshape(-1, 3)
b = win.bases.tobytes()
kinds = {}
for w in bad:
    a, nc = int(row_off[w]), int(ncol[w])
    seg = r[a:a + 3 * nc]
    kind = "zero" if not any(seg) else "other"
    lr, lc, lu = (int(x) for x in L[w])
    eq = b[off[3 * w]:off[3 * w + 1]] == b[off[3 * w + 1]:off[3 * w + 2]]
    key = (kind, "rows<=64" if max(lc, lu) <= 64 else "rows<=80" if max(lc, lu) <= 80 else "rows<=96" if max(lc, lu) <= 96 else "more", "equal" if eq else "differs", "ncol same" if ncol[w] == ncol0[w] else "ncol differs")
    kinds[key] = kinds.get(key, 0) + 1
for k, v in sorted(kinds.items()):
    print(v, k)
# how many of the class's windows are fine?
cls = [(max(int(L[w][1]), int(L[w][2])) > 80 and max(int(L[w][1]), int(L[w][2])) <= 96) for w in range(win.n_windows)]
print("windows with 80 < rows <= 96:", sum(cls), "bad among them:", sum(1 for w in bad if cls[w]))
for w in bad[:12]:
    a, nc = int(row_off[w]), int(ncol[w])
    print(w, L[w], "ncol", nc, int(ncol0[w]), r[a:a + 16], r0[int(row_off0[w]):int(row_off0[w]) + 16])
