"""Statistics path, CPU side: the Python oracle against golden vectors from the
real reference module, the product's host logic (msa parsing, read grouping,
float aggregation, report text, homopolymer state machine) against the oracle.
The per-column GPU kernel itself is covered by tests/test_stats_gpu.py."""
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
from elector_amd import computeStats as cs  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "stats_golden.json")


def oracle_counter_array(pieces, oracle_pieces):
    """Oracle per-piece dicts -> the C-ABI counter layout (processed pieces in order)."""
    out = np.zeros((len(pieces.cols), cs.ES_NCOUNTERS), dtype=np.int64)
    out[:, cs.ES_EXT_LEFT] = out[:, cs.ES_EXT_RIGHT] = out[:, cs.ES_MISSING_LAST] = -1
    it = iter(oracle_pieces)
    for p in range(len(pieces.cols)):
        if pieces.cols[p] <= 10:
            continue
        k = next(it)
        for name, idx in (("TP", cs.ES_TP), ("FP", cs.ES_FP), ("FN", cs.ES_FN), ("cor", cs.ES_COR), ("unc", cs.ES_UNC),
                          ("ucor", cs.ES_UCOR), ("uunc", cs.ES_UUNC), ("gc_ref", cs.ES_GC_REF), ("gc_cor", cs.ES_GC_COR),
                          ("insU", cs.ES_INS_U), ("delU", cs.ES_DEL_U), ("subU", cs.ES_SUB_U), ("insC", cs.ES_INS_C),
                          ("delC", cs.ES_DEL_C), ("subC", cs.ES_SUB_C), ("len_ref", cs.ES_LEN_REF),
                          ("len_cor", cs.ES_LEN_COR), ("len_unc", cs.ES_LEN_UNC), ("gaps_left", cs.ES_GAPS_LEFT),
                          ("gaps_right", cs.ES_GAPS_RIGHT), ("ext_left", cs.ES_EXT_LEFT), ("ext_right", cs.ES_EXT_RIGHT),
                          ("missing_after", cs.ES_MISSING), ("missing_last", cs.ES_MISSING_LAST)):
            out[p, idx] = k[name]
        out[p, cs.ES_PROCESSED] = 1
    return out


def test_oracle_against_reference_golden():
    for case in json.load(open(GOLD)):
        tup, stdout, log, per_read, sizes = stats_oracle.output_recall_precision(
            case["msa"], case["small"], case["wrong"], 5, 0.1, {k: tuple(v) for k, v in case["clips"].items()})
        assert json.loads(json.dumps(tup)) == case["tuple"]
        assert "None\n" + stdout == case["stdout"]
        assert log == case["log"]
        assert "score metric\n" + "".join(per_read) == case["per_read"]


@pytest.mark.parametrize("seed", [3, 4])
def test_host_aggregation_and_report(tmp_path, seed, capsys):
    """parse_msa + aggregate + report text of the product, fed with ORACLE counters:
    must reproduce the oracle's (= the reference's) tuple, stdout, log and files."""
    reads = msa_gen.make_reads(seed * 100, 30, 900)
    txt, small, wrong = msa_gen.msa_text(reads)
    (tmp_path / "msa.fa").write_text(txt)
    cor = tmp_path / "corrected.fa"
    cor.write_text("".join(">c%d\n%s\n" % (i, r[2].decode()) for i, r in enumerate(reads)))
    res, oracle_pieces = stats_oracle.compute_metrics(txt, 5)
    exp_tuple, exp_out, exp_log, exp_per, exp_sizes = stats_oracle.output_recall_precision(txt, small, wrong, 5, 0.1)

    def fake_counters(pieces, clipsNb=None, engine=None):
        masks = [np.array(k["mask"], dtype=np.uint8) for k in oracle_pieces]
        it = iter(masks)
        last = []
        for p in range(len(pieces.cols)):
            m = next(it) if pieces.cols[p] > 10 else np.zeros(pieces.cols[p], dtype=np.uint8)
            if p >= pieces.read_first[-2]:
                last.append(m)
        return oracle_counter_array(pieces, oracle_pieces), np.concatenate(last)
    orig = cs.stats_counters
    cs.stats_counters = fake_counters
    try:
        log = io.StringIO()
        tup = cs.outputRecallPrecision(str(cor), str(tmp_path), log, small, wrong, 5, 0.1, "sizes.txt", {})
    finally:
        cs.stats_counters = orig
    assert tup == exp_tuple
    assert capsys.readouterr().out == "None\n" + exp_out
    assert log.getvalue() == exp_log
    assert (tmp_path / "per_read_metrics.txt").read_text() == "score metric\n" + "".join(exp_per)
    sizes = (tmp_path / "sizes.txt").read_text().split("\n")
    assert sizes[0] == "size type"
    assert [s + "\n" for s in sizes[1:1 + len(exp_sizes)]] == exp_sizes


@pytest.mark.parametrize("seed", [1, 2, 5])
def test_homopolymer_pairs_host_function(seed):
    """The library's O(1)-state homopolymer walk against the oracle's list version."""
    reads = msa_gen.make_reads(seed * 7, 16, 900)
    txt, _, _ = msa_gen.msa_text(reads)
    res, oracle_pieces = stats_oracle.compute_metrics(txt, 5)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "msa.fa")
        open(path, "w").write(txt)
        pieces = cs.parse_msa(path, cs.getSplit(path))
    # masks of every processed piece, then check read by read using the oracle ratios of each read
    lines = txt.split("\n")
    it = iter(oracle_pieces)
    masks = [np.array(next(it)["mask"], dtype=np.uint8) if pieces.cols[p] > 10 else np.zeros(pieces.cols[p], np.uint8)
             for p in range(len(pieces.cols))]
    last = np.concatenate(masks[int(pieces.read_first[-2]):])
    assert cs.homopolymer_ratios(pieces, last, 5) == res["lastReadRatios"]


def test_get_split_matches_uniq_semantics(tmp_path):
    p = tmp_path / "m.fa"
    p.write_text(">a \nAC\n>a \nAC\n>a \nAC\n>a \nAC\n>a \nAC\n>a \nAC\n>b \nAC\n>b \nAC\n>b \nAC\n")
    assert cs.getSplit(str(p)) == {"a": 2, "b": 1}
    assert stats_oracle.split_counts(p.read_text().split("\n")) == {"a": 2, "b": 1}
