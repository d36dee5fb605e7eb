"""The CPU oracle against the golden vectors produced by the real reference
binaries (tests/golden, generator oracle/make_golden.py).  Runs everywhere."""
import ctypes
import os

import numpy as np
import pytest

import golden_io
import oracle_lib
import synth


@pytest.mark.parametrize("name", ["windows_example.tsv", "windows_synth.tsv", "windows_adversarial.tsv"])
def test_window_rows(name):
    gold = golden_io.windows(name)
    triples = [g[0] for g in gold]
    bases, off = synth.pack_windows(triples)
    rows, ncol, scores, cells = oracle_lib.batch(np.frombuffer(bases, dtype=np.uint8), off)
    for w, (t, exp) in enumerate(gold):
        assert rows[w] == exp, (w, t)
        assert ncol[w] == len(exp[0])
    assert cells > 0


def test_heaviest_bundle_rows(tmp_path):
    gold = golden_io.bundles()
    triples = [g[0] for g in gold]
    n1, n2, n3 = synth.write_fasta_triples(triples, str(tmp_path / "in"))
    mat = oracle_lib.write_matrix(str(tmp_path / "p.mat"))
    out = str(tmp_path / "out")
    assert oracle_lib.run_files(mat, n1, n3, n2, out, with_bundles=True) == len(triples)
    lines = open(out, "rb").read().split(b"\n")
    k = 0
    for t, recs in gold:
        for h, r in recs:
            assert lines[k] == h and lines[k + 1] == r, (t, h)
            k += 2
    assert lines[k:] == [b""]


def test_heaviest_bundle_batch_entry():
    """the in-memory batch entry the GPU parity tests compare against gives the golden rows too"""
    gold = golden_io.bundles()
    bases, off = synth.pack_windows([g[0] for g in gold])
    got = oracle_lib.batch_bundles(np.frombuffer(bases, dtype=np.uint8), off)
    for (t, recs), (rows, counts, ids) in zip(gold, got):
        exp_rows, exp_counts = golden_io.bundle_expectation(recs)
        assert rows == exp_rows and counts == exp_counts, t
        assert sum(1 for i in ids if i >= 0) == sum(counts)


def test_matrix_parameters(tmp_path):
    g = golden_io.params()

    class P(ctypes.Structure):
        _fields_ = [("nsymbol", ctypes.c_int), ("symbol", ctypes.c_char * 129), ("_pad", ctypes.c_char * 3),
                    ("score", (ctypes.c_int * 128) * 128), ("gap_set", (ctypes.c_int * 3) * 2),
                    ("trunc", ctypes.c_int), ("decay", ctypes.c_int), ("M", ctypes.c_int),
                    ("gpx", ctypes.c_int * 256), ("gpy", ctypes.c_int * 256)]
    # defaults == what the reference file parsed to; our writer round-trips
    for buf in (oracle_lib.default_params(),
                oracle_lib.read_params(oracle_lib.write_matrix(str(tmp_path / "w.mat")))):
        q = P.from_buffer(buf)
        assert q.nsymbol == g["nsymbol"] and q.symbol.decode() == g["symbol"] and q.M == g["max_gap_length"]
        assert list(q.gpx[: q.M + 2]) == g["gap_penalty_x"] and list(q.gpy[: q.M + 2]) == g["gap_penalty_y"]
        assert [[q.score[i][j] for j in range(q.nsymbol)] for i in range(q.nsymbol)] == g["score"]


def test_file_level_driver_matches_batch(tmp_path):
    triples = synth.window_triples(5, 60, 3, 90)
    n1, n2, n3 = synth.write_fasta_triples(triples, str(tmp_path / "in"))
    mat = oracle_lib.write_matrix(str(tmp_path / "p.mat"))
    out = str(tmp_path / "out")
    assert oracle_lib.run_files(mat, n1, n3, n2, out) == len(triples)
    lines = open(out, "rb").read().split(b"\n")
    bases, off = synth.pack_windows(triples)
    rows, _, _, _ = oracle_lib.batch(np.frombuffer(bases, dtype=np.uint8), off)
    for w in range(len(triples)):
        assert lines[6 * w] == b">w%d untitled" % w
        assert (lines[6 * w + 1], lines[6 * w + 3], lines[6 * w + 5]) == rows[w]
