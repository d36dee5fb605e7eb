"""bench.py's every-window comparison of the device MSA with the reference `poa`'s output (the
`parity_vs_reference` object of the bench line): the comparison itself, checked here without a GPU -- the device's
column-interleaved layout is filled from the CPU oracle's rows, the reference side is the real binary (oracle/_ref)
driven exactly as bench.cpu_baseline drives it."""
import importlib
import os

import numpy as np
import pytest

import oracle_lib
import synth

pytestmark = pytest.mark.skipif(not oracle_lib.have_reference_binaries(), reason="oracle/_ref not built (no /root/reference)")


class _Win:
    pass


def _device_layout(rows, ncol, off):
    cols = np.zeros(3 * int(off[-1]) + 64, dtype=np.uint8)
    for w, (r, c, u) in enumerate(rows):
        a, nc = 3 * int(off[3 * w]), int(ncol[w])
        blk = np.stack([np.frombuffer(x, dtype=np.uint8) for x in (r, c, u)], axis=1).reshape(-1)
        cols[a:a + 3 * nc] = blk
    return cols


def test_every_window_comparison(monkeypatch):
    bench = importlib.import_module("bench")
    triples = synth.window_triples(5, 700, 5, 120) + synth.adversarial_triples(6, 120)
    bases, off = synth.pack_windows(triples)
    win = _Win()
    win.bases, win.off, win.n_windows = np.frombuffer(bases, dtype=np.uint8), off, len(triples)
    rows, ncol, _, _ = oracle_lib.batch(win.bases, off)
    cols = _device_layout(rows, ncol, off)
    lr = (off[1::3] - off[0:-1:3]).astype(np.int64)
    monkeypatch.setattr(os, "cpu_count", lambda: 4)
    base, parity = bench.cpu_baseline(win, lr, 2.0, (cols, ncol))
    assert base["kind"] == "reference" and parity == {"windows": len(triples), "differing": 0, "against": parity["against"]}
    # one wrong letter, one wrong column count, one wrong gap: three windows differ
    bad = cols.copy()
    bad[3 * int(off[3 * 10]) + 4] ^= 1
    bad[3 * int(off[3 * 500]) + 2] = ord(".") if bad[3 * int(off[3 * 500]) + 2] != ord(".") else ord("a")
    n2 = ncol.copy()
    n2[300] += 1
    _, parity = bench.cpu_baseline(win, lr, 2.0, (bad, n2))
    assert parity["windows"] == len(triples) and parity["differing"] == 3
