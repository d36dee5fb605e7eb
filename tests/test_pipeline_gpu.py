"""Call site #1 end to end on the GPU: elector_amd.alignment.getPOA must write the
same msa.fa bytes (and return the same counters) as the reference chain
masterSplitter -> poa -> Donatello (golden vectors made by oracle/make_golden.py
with the real binaries), and feeding that file to the computeStats mirror must
agree with the statistics oracle."""
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import stats_oracle  # noqa: E402

from elector_amd import alignment, computeStats  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden", "pipeline_golden.json")


@pytest.mark.parametrize("idx", [0, 1])
def test_getpoa_writes_reference_msa(tmp_path, engine, capsys, idx):
    case = json.load(open(GOLD))[idx]
    for k in ("ref", "cor", "unc"):
        (tmp_path / (k + ".fa")).write_text(case[k])
    small, wrong = alignment.getPOA(str(tmp_path / "cor.fa"), str(tmp_path / "ref.fa"), str(tmp_path / "unc.fa"),
                                    4, str(tmp_path), 0.1, engine=engine)
    capsys.readouterr()
    assert (small, wrong) == (case["small"], case["wrong"])
    assert (tmp_path / "msa.fa").read_text() == case["msa"]
    # soft name variant + append semantics
    alignment.getPOA(str(tmp_path / "cor.fa"), str(tmp_path / "ref.fa"), str(tmp_path / "unc.fa"),
                     1, str(tmp_path), 0.1, soft="lordec", engine=engine)
    alignment.getPOA(str(tmp_path / "cor.fa"), str(tmp_path / "ref.fa"), str(tmp_path / "unc.fa"),
                     1, str(tmp_path), 0.1, soft="lordec", engine=engine)
    capsys.readouterr()
    assert (tmp_path / "msa_lordec.fa").read_text() == case["msa"] * 2
    # call site #2 on the file just written
    computeStats._engine = engine
    log = io.StringIO()
    try:
        exp_tuple, exp_out, exp_log, _, _ = stats_oracle.output_recall_precision(case["msa"], small, wrong, 5, 0.1)
    except KeyError:
        # a header with a title breaks the reference's own header bookkeeping
        # (computeStats.py:548 KeyError); the mirror must fail the same way
        with pytest.raises(KeyError):
            computeStats.outputRecallPrecision(str(tmp_path / "cor.fa"), str(tmp_path), log, small, wrong, 5, 0.1,
                                               "sizes.txt", {})
        return
    tup = computeStats.outputRecallPrecision(str(tmp_path / "cor.fa"), str(tmp_path), log, small, wrong, 5, 0.1,
                                             "sizes.txt", {})
    out = capsys.readouterr().out
    assert tup == exp_tuple and out == "None\n" + exp_out and log.getvalue() == exp_log
