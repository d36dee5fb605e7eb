"""Statistics kernel on the GPU (through the C ABI) against the Python oracle and
the golden vectors of the real reference module.  Integers: bit-exact."""
import io
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import stats_oracle  # noqa: E402

import msa_gen  # noqa: E402
from test_stats_cpu import GOLD, oracle_counter_array  # noqa: E402
from elector_amd import computeStats as cs  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,n,L", [(21, 40, 900), (22, 24, 2500), (23, 60, 400)])
def test_counters_equal_oracle(tmp_path, engine, seed, n, L):
    reads = msa_gen.make_reads(seed, n, L)
    txt, _, _ = msa_gen.msa_text(reads)
    path = tmp_path / "msa.fa"
    path.write_text(txt)
    pieces = cs.parse_msa(str(path), cs.getSplit(str(path)))
    res, oracle_pieces = stats_oracle.compute_metrics(txt, 5)
    exp = oracle_counter_array(pieces, oracle_pieces)
    got, last_mask = cs.stats_counters(pieces, None, engine)
    proc = got[:, cs.ES_PROCESSED] == 1
    assert np.array_equal(proc, exp[:, cs.ES_PROCESSED] == 1)
    assert np.array_equal(got[proc], exp[proc]), np.argwhere(got[proc] != exp[proc])[:5]
    assert cs.homopolymer_ratios(pieces, last_mask, 5) == res["lastReadRatios"]


def test_full_report_equals_reference_golden(tmp_path, engine, capsys):
    cs._engine = engine
    for i, case in enumerate(json.load(open(GOLD))):
        d = tmp_path / ("c%d" % i)
        d.mkdir()
        (d / "msa.fa").write_text(case["msa"])
        (d / "cor.fa").write_text(">x\nACGT\n")
        log = io.StringIO()
        tup = cs.outputRecallPrecision(str(d / "cor.fa"), str(d), log, case["small"], case["wrong"], 5, 0.1,
                                       "sizes.txt", {k: tuple(v) for k, v in case["clips"].items()})
        assert json.loads(json.dumps(tup)) == case["tuple"]
        assert capsys.readouterr().out == case["stdout"]
        assert log.getvalue() == case["log"]
        assert (d / "per_read_metrics.txt").read_text() == case["per_read"]


def test_stats_edge_cases(tmp_path, engine):
    """Short pieces (<= 10 columns, skipped), all-gap ends, a read whose last piece is short."""
    rows = [("r1", "acgtacgtacgtacgtacgtacgtacgtacgt", "acgtacgtacgtacgaacgtacgtacgtacgt", "acgtacgtacg.acgtacgtacgtacgtacgt"),
            ("r2", "aaa", "aaa", "aaa"),
            ("r3", "." * 30 + "acgtacgtacgtacgtacgtacgtacgt" * 3, "ggg" + "." * 27 + "acgtacgtacgtacgtacgtacgtacgt" * 3,
             "." * 30 + "acgtacgtacgtacgtacgtacgtacgt" * 3),
            ("r4", "acgtacgtacgtacgtacgtacgtacgtacgtacgt" * 3, "." * 60 + "acgtacgtacgtacgtacgtacgtacgtacgtacgtacgtacgtacgt",
             "acgtacgtacgtacgtacgtacgtacgtacgtacgt" * 3),
            ("r4", "acgtac", "acgtac", "acgtac")]
    txt = "".join(">%s \n%s\n>%s \n%s\n>%s \n%s\n" % (h, a, h, b, h, c) for h, a, b, c in rows)
    path = tmp_path / "msa.fa"
    path.write_text(txt)
    pieces = cs.parse_msa(str(path), cs.getSplit(str(path)))
    res, oracle_pieces = stats_oracle.compute_metrics(txt, 5)
    exp = oracle_counter_array(pieces, oracle_pieces)
    got, _ = cs.stats_counters(pieces, None, engine)
    proc = exp[:, cs.ES_PROCESSED] == 1
    assert np.array_equal(got[:, cs.ES_PROCESSED], exp[:, cs.ES_PROCESSED])
    assert np.array_equal(got[proc], exp[proc])
