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


def test_bundles_on_k_poa_graphs(engine):
    """Round 5: with elector_ctx_keep_graph the batch stays on k_poa, which writes what the search reads (letters and flags
    of the graph after fusion #1, ring ids, the x -> y map) from its LDS records.  Every way k_poa comes by a graph:
    shortcut graphs (corrected = reference, one substitution, one indel, the one-letter filler), the dynamic program,
    the far-edge instance, and the windows it hands back to the two-kernel path."""
    import test_poa_gpu as tp
    rng = np.random.default_rng(51)
    t = tp.one_indel_triples(52)[:1500] + tp.far_edge_triples(53, 900) + synth.window_triples(54, 1500, 5, 300)
    for L in (9, 40, 130, 260):                       # corrected = reference, one substitution, a filler letter
        for _ in range(40):
            ref = synth.random_seq(rng, L)
            unc = synth.mutate(rng, ref, 0.12) or b"A"
            p = int(rng.integers(0, L))
            sub = ref[:p] + bytes([b"ACGT"[(b"ACGT".index(ref[p:p + 1]) + 1) % 4]]) + ref[p + 1:]
            t += [(ref, ref, unc), (ref, sub, unc), (ref, b"N", unc)]
    check(engine, t)


def test_bundles_k_poa_equals_two_kernel_path(engine, monkeypatch):
    """the same batch with k_poa switched off (ELECTOR_NO_PACK: the graph from k_fused_a / k_fused_b, round 4's only way)"""
    import test_poa_gpu as tp
    t = synth.window_triples(55, 2000, 7, 200) + synth.adversarial_triples(56, 600) + tp.far_edge_triples(57, 300)
    rows, got = engine.align_with_bundles(t)
    monkeypatch.setenv("ELECTOR_NO_PACK", "1")
    rows2, got2 = engine.align_with_bundles(t)
    assert rows == rows2 and got == got2
    check(engine, t)


@pytest.mark.parametrize("cls", [0, 4, 7, 11, 16])
def test_bundles_geometry_classes(engine, cls, monkeypatch):
    """the graph output of several k_poa instances (lanes per window x rows per lane) on windows of every size"""
    monkeypatch.setenv("ELECTOR_FORCE_CLASS", str(cls))
    check(engine, synth.window_triples(300 + cls, 400, 5, 420) + synth.adversarial_triples(320 + cls, 200))


def test_bundles_enqueue_is_noted_and_queued_later(engine):
    """elector_poa_bundles_enqueue notes the search; the context queues it at its next call that waits anyway -- in front
    of the next batch (whose offsets and graph it must not see), at a sync, at the fetch.  Wrong sizes are refused at once."""
    from elector_amd._capi import ElectorError
    a = synth.window_triples(61, 1500, 7, 200) + synth.adversarial_triples(62, 300)
    b = synth.window_triples(63, 900, 20, 260)
    bases_b, off_b = synth.pack_windows(b)
    exp_b = oracle_lib.batch_bundles(np.frombuffer(bases_b, dtype=np.uint8), off_b, 0.9)
    engine.keep_graph(True)
    try:
        for now in (0, 1):
            engine.option("bundles_now", now)
            engine.align(a, strict=False)
            with pytest.raises(ElectorError):
                engine.bundles_enqueue(len(a) + 1)
            engine.bundles_enqueue(len(a))
            rows_b = engine.align(b, strict=False)               # the noted search of `a` runs before `b` takes its arrays
            engine.bundles_enqueue(len(b))
            engine.sync()                                        # ... and this one here
            assert engine.bundles(len(b), int(off_b[-1])) == exp_b
            assert rows_b == oracle_lib.batch(np.frombuffer(bases_b, dtype=np.uint8), off_b)[0]
    finally:
        engine.option("bundles_now", 0)
        engine.keep_graph(False)
