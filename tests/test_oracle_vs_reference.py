"""The CPU oracle against the REAL reference binaries (oracle/_ref, built from
/root/reference by oracle/Makefile) on fresh seeded inputs.  Skipped where the
binaries are absent; the committed golden vectors cover that case."""
import os
import subprocess

import pytest

import oracle_lib
import synth

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not oracle_lib.have_reference_binaries(), reason="oracle/_ref not built")]
REF = oracle_lib.REF_DIR


def run_ref_poa(mat, n1, n2, n3, out, hb=False):
    if hb:
        cmd = [os.path.join(REF, "poa_hb"), mat, n1, n3, n2, out]
    else:
        cmd = [os.path.join(REF, "poa"), "-pir", out, "-preserve_seqorder", "-corrected_reads_fasta", n3,
               "-reference_reads_fasta", n1, "-uncorrected_reads_fasta", n2, "-preserve_seqorder", "-threads", "1",
               "-pathMatrix", mat]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out, "rb").read()


@pytest.mark.parametrize("case", ["typical", "long", "adversarial", "noisy"])
@pytest.mark.parametrize("hb", [False, True])
def test_oracle_equals_reference(tmp_path, case, hb):
    triples = {"typical": lambda: synth.window_triples(201, 400, 1, 130),
               "long": lambda: synth.window_triples(202, 40, 150, 420),
               "adversarial": lambda: synth.adversarial_triples(203, 600),
               "noisy": lambda: synth.window_triples(204, 150, 10, 150, 0.35, 0.3)}[case]()
    n1, n2, n3 = synth.write_fasta_triples(triples, str(tmp_path / "in"))
    mat = oracle_lib.write_matrix(str(tmp_path / "p.mat"))
    exp = run_ref_poa(mat, n1, n2, n3, str(tmp_path / "ref.out"), hb)
    assert oracle_lib.run_files(mat, n1, n3, n2, str(tmp_path / "or.out"), with_bundles=hb) == len(triples)
    assert open(str(tmp_path / "or.out"), "rb").read() == exp


def test_general_matrix(tmp_path):
    """A matrix with non-uniform scores and decaying gap penalties (exercises the
    gap-tag table the shipped matrix collapses)."""
    mat = str(tmp_path / "g.mat")
    with open(mat, "w") as f:
        f.write("GAP-TRUNCATION-LENGTH=3\nGAP-DECAY-LENGTH=4\nGAP-PENALTIES=9 4 1\n  A a c g t n\n")
        rows = {"A": [1, -3, -3, -3, -3, -1], "a": [-3, 4, -5, -2, -5, -1], "c": [-3, -5, 4, -5, -2, -1],
                "g": [-3, -2, -5, 4, -5, -1], "t": [-3, -5, -2, -5, 4, -1], "n": [-1, -1, -1, -1, -1, 0]}
        for k, v in rows.items():
            f.write(k + " " + " ".join(str(x) for x in v) + "\n")
    triples = synth.window_triples(205, 300, 1, 140) + synth.adversarial_triples(206, 240)
    n1, n2, n3 = synth.write_fasta_triples(triples, str(tmp_path / "in"))
    exp = run_ref_poa(mat, n1, n2, n3, str(tmp_path / "ref.out"))
    assert oracle_lib.run_files(mat, n1, n3, n2, str(tmp_path / "or.out")) == len(triples)
    assert open(str(tmp_path / "or.out"), "rb").read() == exp
