"""A wider randomised parity sweep of the HIP path against the oracle: error regimes from perfectly
corrected to badly corrected reads (deep predecessor distances, many two-predecessor nodes, windows
the fused kernels hand back to the generic ones), low-complexity alphabets (ties), and batch shapes
that stress the host side (one window, all windows identical, all windows trivial / none trivial,
sizes straddling every geometry class boundary)."""
import numpy as np
import pytest

import oracle_lib
import synth

pytestmark = pytest.mark.gpu


def check(engine, triples):
    bases, off = synth.pack_windows(triples)
    exp_rows, _, exp_scores, _ = oracle_lib.batch(np.frombuffer(bases, dtype=np.uint8), off)
    got, scores = engine.align(triples, want_scores=True)
    bad = [w for w in range(len(triples)) if got[w] != exp_rows[w]]
    assert not bad, "first differing window %d of %d: %r\n got %r\n exp %r" % (
        bad[0], len(bad), triples[bad[0]], got[bad[0]], exp_rows[bad[0]])
    assert np.array_equal(scores, exp_scores)


@pytest.mark.parametrize("err_unc,err_cor", [(0.15, 0.0), (0.15, 0.003), (0.12, 0.02), (0.2, 0.08), (0.35, 0.3), (0.02, 0.15)])
def test_error_regimes(engine, err_unc, err_cor):
    seed = int(1000 * err_unc + 100000 * err_cor)
    check(engine, synth.window_triples(seed, 5000, 5, 140, err_unc=err_unc, err_cor=err_cor))


def test_class_boundaries(engine):
    """uncorrected lengths on both sides of every strip height G x R"""
    rng = np.random.default_rng(77)
    caps = [32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384, 448, 512]
    triples = []
    for cap in caps:
        for lu in (cap - 1, cap, cap + 1):
            for _ in range(6):
                unc = synth.random_seq(rng, lu)
                ref = synth.mutate(rng, unc, 0.13)
                triples.append((ref, synth.mutate(rng, ref, 0.01), unc))
    check(engine, triples)


def test_low_complexity(engine):
    rng = np.random.default_rng(78)
    two = np.frombuffer(b"AC", dtype=np.uint8)
    triples = []
    for _ in range(1500):
        L = int(rng.integers(8, 100))
        ref = synth.random_seq(rng, L, two)
        triples.append((ref, synth.mutate(rng, ref, 0.05, alphabet=two), synth.mutate(rng, ref, 0.2, alphabet=two)))
    check(engine, triples)


def test_batch_shapes(engine):
    rng = np.random.default_rng(79)
    ref = synth.random_seq(rng, 61)
    unc = synth.mutate(rng, ref, 0.15)
    check(engine, [(ref, ref, unc)])                                        # one window, trivial alignment #1
    check(engine, [(ref, synth.mutate(rng, ref, 0.05), unc)])                # one window
    check(engine, [(ref, ref, unc)] * 3000)                                  # all identical, all trivial
    t = synth.window_triples(80, 3000, 20, 90, err_cor=0.0)                  # all trivial, mixed sizes
    check(engine, t)
    check(engine, [(r, synth.mutate(rng, r, 0.2) + b"A", u) for r, _, u in t])   # none trivial


def test_degenerate_batches(engine):
    from elector_amd._capi import W_EMPTY
    assert engine.align([]) == []
    got = engine.align([(b"", b"ACGT", b"ACGT"), (b"ACGT", b"", b"ACGT"), (b"ACGT", b"ACGT", b""), (b"A", b"A", b"A")], strict=False)
    assert got[:3] == [None, None, None] and got[3] == (b"a", b"a", b"a")
    rows, row_off, ncol, status, _ = engine.align_packed(*[np.asarray(x) for x in (np.frombuffer(b"ACGTACGT", dtype=np.uint8),
                                                                                   np.array([0, 0, 4, 8], dtype=np.int64))], strict=False)
    assert status[0] == W_EMPTY and ncol[0] == 0
    with pytest.raises(Exception):
        engine.align([(b"", b"ACGT", b"ACGT")])          # strict: a failed window raises
