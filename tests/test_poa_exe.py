"""The `poa`-compatible executable (elector_amd/bin/poa): ELECTOR's own command line
(elector/alignment.py:60) in, the file `bin/poa -pir` writes out -- against the golden rows of the real binary and,
where oracle/_ref travelled, against the real binary's whole output on the same three files."""
import os
import subprocess

import pytest

import golden_io
import oracle_lib
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "elector_amd", "bin", "poa")


def command(exe, d, names, out):
    return [exe, "-pir", out, "-preserve_seqorder", "-corrected_reads_fasta", names[2], "-reference_reads_fasta", names[0],
            "-uncorrected_reads_fasta", names[1], "-preserve_seqorder", "-threads", "1", "-pathMatrix",
            oracle_lib.write_matrix(os.path.join(d, "params.mat"))]


def test_executable_refuses_to_run_without_a_gpu(tmp_path):
    """no CPU path behind the executable either: exit code 2 and the library's message"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert os.path.exists(EXE), "python -m elector_amd.build builds it"
    names = synth.write_fasta_triples([(b"ACGT", b"ACGT", b"ACGT")], str(tmp_path / "in"))
    r = subprocess.run(command(EXE, str(tmp_path), names, str(tmp_path / "out")), capture_output=True, text=True)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["windows_example.tsv", "windows_adversarial.tsv"])
def test_executable_writes_what_poa_writes(tmp_path, name):
    gold = golden_io.windows(name)
    triples = [t for t, _ in gold]
    d = str(tmp_path)
    names = synth.write_fasta_triples(triples, os.path.join(d, "in"))
    out = os.path.join(d, "smsa")
    r = subprocess.run(command(EXE, d, names, out), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == "0 1 2 \n"
    lines = open(out, "rb").read().split(b"\n")
    assert len(lines) == 6 * len(triples) + 1 and lines[-1] == b""
    for w, (_, rows) in enumerate(gold):
        for k in range(3):
            assert lines[6 * w + 2 * k] == b">w%d untitled" % w
            assert lines[6 * w + 2 * k + 1] == rows[k], (w, k)
    ref = os.path.join(oracle_lib.REF_DIR, "poa")
    if os.path.exists(ref):
        out2 = os.path.join(d, "smsa_ref")
        subprocess.run(command(ref, d, names, out2), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert open(out, "rb").read() == open(out2, "rb").read()


@pytest.mark.gpu
def test_executable_reads_fasta_like_the_reference(tmp_path):
    """titles, several sequence lines per record, blanks inside them, '*' lines"""
    d = str(tmp_path)
    recs = [(b">r1 some title here", b"ACGTAC\nGTAC GT\n"), (b">r2", b"acgtnn\n*ignored\nACGT\n"), (b">r3\tx", b"ACGTACGTAA\n")]
    paths = []
    for k, mut in enumerate((lambda s: s, lambda s: s.replace(b"GTAC", b"GAAC"), lambda s: s[:-3] + b"T\n")):
        p = os.path.join(d, "f%d.fa" % k)
        with open(p, "wb") as f:
            for h, s in recs:
                f.write(h + b"\n" + mut(s))
        paths.append(p)
    out = os.path.join(d, "o")
    r = subprocess.run(command(EXE, d, [paths[0], paths[2], paths[1]], out), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ref = os.path.join(oracle_lib.REF_DIR, "poa")
    if os.path.exists(ref):
        out2 = os.path.join(d, "o_ref")
        subprocess.run(command(ref, d, [paths[0], paths[2], paths[1]], out2), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert open(out, "rb").read() == open(out2, "rb").read()
    else:
        lines = open(out, "rb").read().split(b"\n")
        assert lines[0] == b">r1 some title here" and lines[6] == b">r2 untitled" and lines[12] == b">r3 x"
