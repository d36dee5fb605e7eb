"""A process's FIRST call through k_poa against the two-kernel path of the same process (ELECTOR_NO_PACK=1), on the
windows of 300 E. coli-like reads: rows and column counts bit-exact.  Run as a program by test_poa_gpu.py.

Why a process of its own: a VGPR spill reload that reads lanes its store never wrote ("a compiler fault
the build now looks for") returns whatever the scratch memory holds -- zeros in a fresh process, the right values of the
launch before in a warm one, which is why a session-wide engine fixture does not see it."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elector_amd import split, synthetic  # noqa: E402
from elector_amd.poa import PoaEngine  # noqa: E402


def main():
    eng = PoaEngine(0)
    reads = synthetic.read_triples("ecoli30x_simlord_lordec", 300, seed=5)
    win = split.split_reads(reads, 0.1, None, nthreads=8)
    rows, row_off, ncol, status, _ = eng.align_packed(win.bases, win.off)
    rows, row_off, ncol = rows.copy(), row_off.copy(), ncol.copy()
    os.environ["ELECTOR_NO_PACK"] = "1"
    rows0, row_off0, ncol0, status0, _ = eng.align_packed(win.bases, win.off)
    del os.environ["ELECTOR_NO_PACK"]
    assert not status.any() and not status0.any()
    r, r0 = rows.tobytes(), rows0.tobytes()
    bad = [w for w in range(win.n_windows)
           if ncol[w] != ncol0[w] or r[int(row_off[w]):int(row_off[w]) + 3 * int(ncol[w])] != r0[int(row_off0[w]):int(row_off0[w]) + 3 * int(ncol0[w])]]
    print("windows %d differing %d" % (win.n_windows, len(bad)))
    if bad:
        L = np.diff(win.off).reshape(-1, 3)
        for w in bad[:10]:
            print("window", w, "lengths", L[w].tolist(), "ncol", int(ncol[w]), int(ncol0[w]))
    eng.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
