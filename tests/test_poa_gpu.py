"""Parity tests proper: the HIP path, called through the C ABI, against the CPU
oracle on the same seeded inputs.  Bit-exact: MSA rows (bytes), column counts
and the best scores of both alignments (integer DP)."""
import numpy as np
import pytest

import oracle_lib
import synth

pytestmark = pytest.mark.gpu


def check(engine, triples):
    bases, off = synth.pack_windows(triples)
    exp_rows, exp_ncol, exp_scores, _ = oracle_lib.batch(np.frombuffer(bases, dtype=np.uint8), off)
    got, scores = engine.align(triples, want_scores=True)
    bad = [w for w in range(len(triples)) if got[w] != exp_rows[w]]
    assert not bad, "first differing window %d of %d: %r\n got %r\n exp %r" % (
        bad[0], len(bad), triples[bad[0]], got[bad[0]], exp_rows[bad[0]])
    assert np.array_equal(scores, exp_scores)


def test_small_windows(engine):
    check(engine, synth.window_triples(11, 2000, 1, 62))


def test_typical_windows(engine):
    check(engine, synth.window_triples(12, 3000, 7, 120))


def test_multi_strip_windows(engine):
    check(engine, synth.window_triples(13, 600, 100, 420))


def test_adversarial_windows(engine):
    check(engine, synth.adversarial_triples(14, 3600))


def test_noisy_corrected(engine):
    # heavy corrected-read errors -> big bubbles, long predecessor distances
    check(engine, synth.window_triples(15, 800, 20, 200, err_unc=0.3, err_cor=0.25))


def test_long_windows(engine):
    check(engine, synth.window_triples(16, 12, 1500, 3000))
