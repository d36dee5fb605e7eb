"""a12 parity: heaviest-bundle consensus from the HIP kernel (k_bundle, through the C ABI) against
(1) the records the real reference printed (tests/golden/bundles.tsv, made by oracle/hb_driver.c
linked to the reference objects) and (2) the CPU oracle on seeded windows.  Bit-exact: consensus
rows, "containing N seqs" counts, bundle ids of the three sequences."""
import numpy as np
import pytest

import golden_io
import oracle_lib
import synth

pytestmark = pytest.mark.gpu


def check(engine, triples, frac=0.9):
    bases, off = synth.pack_windows(triples)
    exp = oracle_lib.batch_bundles(np.frombuffer(bases, dtype=np.uint8), off, frac)
    rows, got = engine.align_with_bundles(triples, frac)
    exp_rows = oracle_lib.batch(np.frombuffer(bases, dtype=np.uint8), off)[0]
    assert rows == exp_rows
    bad = [w for w in range(len(triples)) if got[w] != exp[w]]
    assert not bad, "first differing window %d of %d: %r\n got %r\n exp %r" % (
        bad[0], len(bad), triples[bad[0]], got[bad[0]], exp[bad[0]])
    return got


def test_golden_bundles(engine):
    gold = golden_io.bundles()
    rows, got = engine.align_with_bundles([g[0] for g in gold])
    for w, (t, recs) in enumerate(gold):
        exp_rows, exp_counts = golden_io.bundle_expectation(recs)
        assert rows[w] == tuple(r for _, r in recs[:3]), t
        assert got[w][0] == exp_rows and got[w][1] == exp_counts, (w, t, got[w])


def test_bundles_typical(engine):
    got = check(engine, synth.window_triples(21, 3000, 7, 160))
    assert len(got) == 3000


def test_bundles_adversarial(engine):
    check(engine, synth.adversarial_triples(22, 2400))


def test_bundles_noisy_and_thresholds(engine):
    t = synth.window_triples(23, 800, 10, 200, err_unc=0.3, err_cor=0.25)
    check(engine, t)                  # several bundles per window
    check(engine, t, frac=0.5)
    check(engine, t, frac=1.0)


def test_bundles_multi_strip_and_long(engine):
    check(engine, synth.window_triples(24, 300, 100, 420))
    check(engine, synth.window_triples(25, 6, 1500, 3000))


def test_bundles_need_keep_graph(engine):
    from elector_amd._capi import ElectorError
    t = synth.window_triples(26, 10, 30, 60)
    engine.align(t)
    with pytest.raises(ElectorError):
        engine.bundles(len(t), 10000)


def test_bundles_first_form(engine, monkeypatch):
    """ELECTOR_BUNDLE_HBM=1: every block through the first form of the search (node records in HBM) -- the path of
    windows beyond 208 nodes -- on windows the LDS form takes otherwise."""
    monkeypatch.setenv("ELECTOR_BUNDLE_HBM", "1")
    check(engine, synth.window_triples(27, 1500, 7, 160))
    check(engine, synth.adversarial_triples(28, 1200))


def test_bundles_lds_class_borders(engine):
    """windows whose graphs have about 52 / 69 / 104 / 208 nodes: the borders of the LDS classes and of the first form"""
    t = []
    for k, lo in enumerate((40, 56, 88, 180)):
        t += synth.window_triples(30 + k, 700, lo, lo + 30, err_unc=0.12, err_cor=0.02)
    check(engine, t)


@pytest.mark.parametrize("pct", ["0", "100"])
def test_bundles_lds_and_global_forms(engine, monkeypatch, pct):
    """the second form of the search with its records in LDS (ELECTOR_BUNDLE_GLOBAL_PCT=0) and in HBM (100); the default
    sends half of every class's blocks each way"""
    monkeypatch.setenv("ELECTOR_BUNDLE_GLOBAL_PCT", pct)
    check(engine, synth.window_triples(41, 2500, 7, 200))
    check(engine, synth.adversarial_triples(42, 1200))
    t = synth.window_triples(43, 600, 10, 200, err_unc=0.3, err_cor=0.25)
    check(engine, t, frac=0.5)
