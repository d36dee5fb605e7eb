"""Read-set preparation: elector_amd.readAndSortFiles against the oracle
restatement and against outputs derived by hand from the reference's rules (readAndSortFiles.py:150-191).
Parity unpinned -- see oracle/sort_oracle.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sort_oracle  # noqa: E402

from elector_amd import readAndSortFiles as rs  # noqa: E402


def test_sort_by_hand(tmp_path):
    src = tmp_path / "cor.fa"
    # junk in front of the first record, a multi-line sequence with blanks and a Windows line end, equal headers
    # (stable order), a header with trailing blanks and inner words, an empty sequence
    src.write_bytes(b"; comment\nACGT\n>read2 extra words  \nAC GT\r\nTT\n\n>read10\nGGGG\n>read2 extra words\nCCCC\n>read1\n>read10\nAAAA\n")
    out = tmp_path / "sorted.fa"
    occ = rs.readAndSortFasta(str(src), str(out))
    assert out.read_text() == (">read1\n\n>read10\nGGGG\n>read10\nAAAA\n>read2 extra words\nACGTTT\n>read2 extra words\nCCCC\n")
    assert occ == {"read1": 1, "read10": 2, "read2 extra words": 2}


def test_duplicate_by_hand(tmp_path):
    ref = tmp_path / "ref.fa"
    unc = tmp_path / "unc.fa"
    ref.write_text(">a\nAAAA\n>b\nCCCC\n>c\nGGGG\n")
    unc.write_text(">a\nAAAT\n>b\nCCCT\n>c\nGGGT\n")
    nr, nu = rs.duplicateRefReads(str(ref), str(unc), {"a": 2, "c": 1}, 3, str(tmp_path / "nu.fa"), str(tmp_path / "nr.fa"))
    assert open(nr).read() == ">a_0\nAAAA\n>a_1\nAAAA\n>c_0\nGGGG\n"
    assert open(nu).read() == ">a_0\nAAAT\n>a_1\nAAAT\n>c_0\nGGGT\n"
    # a read set without trimmed or split reads is rewritten as well (the reference's dict-against-list test)
    nr, nu = rs.duplicateRefReads(str(ref), str(unc), {"a": 1, "b": 1, "c": 1}, 3, str(tmp_path / "nu2.fa"), str(tmp_path / "nr2.fa"))
    assert open(nr).read() == ">a_0\nAAAA\n>b_0\nCCCC\n>c_0\nGGGG\n"


def _random_fasta(rng, path, names, multi_line):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(path, "wb") as f:
        for name in names:
            seq = acgt[rng.integers(0, 4, int(rng.integers(0, 300)))].tobytes()
            f.write(b">" + name + (b" \t" if rng.random() < 0.2 else b"") + (b"\r\n" if rng.random() < 0.1 else b"\n"))
            if multi_line:
                w = int(rng.integers(20, 90))
                for k in range(0, len(seq), w):
                    f.write(seq[k:k + w] + (b" \n" if rng.random() < 0.1 else b"\n"))
                if rng.random() < 0.2:
                    f.write(b"\n")
            else:
                f.write(seq + b"\n")


def test_product_equals_oracle(tmp_path):
    rng = np.random.default_rng(3)
    base = [b"read%d" % i for i in range(60)]
    for trial in range(4):
        perm = [base[i] for i in rng.permutation(len(base))]
        cor_names = [n for n in perm if rng.random() < 0.8 for _ in range(int(rng.integers(1, 4)))]
        cor_names = [cor_names[i] for i in rng.permutation(len(cor_names))]
        files = {}
        for tag, names in (("ref", perm), ("unc", [base[i] for i in rng.permutation(len(base))]), ("cor", cor_names)):
            files[tag] = str(tmp_path / ("%s%d.fa" % (tag, trial)))
            _random_fasta(rng, files[tag], names, multi_line=trial % 2 == 1)
        d1, d2 = tmp_path / ("p%d" % trial), tmp_path / ("o%d" % trial)
        d1.mkdir(); d2.mkdir()
        cor, ref, unc = rs.sortAndDuplicate(None if trial < 2 else "lordec", files["ref"], files["unc"], files["cor"], len(cor_names), str(d1))
        tag = "" if trial < 2 else "_lordec"
        by = "" if trial < 2 else "_by_lordec"
        assert os.path.basename(cor) == "corrected_sorted%s.fa" % by
        assert os.path.basename(ref) == "reference_sorted_duplicated%s.fa" % tag
        assert os.path.basename(unc) == "uncorrected_sorted_duplicated%s.fa" % tag
        sort_oracle.read_and_sort_fasta(files["unc"], str(d2 / "us.fa"))
        sort_oracle.read_and_sort_fasta(files["ref"], str(d2 / "rs.fa"))
        occ = sort_oracle.read_and_sort_fasta(files["cor"], str(d2 / "cs.fa"))
        sort_oracle.duplicate_ref_reads(str(d2 / "rs.fa"), str(d2 / "us.fa"), occ, len(cor_names), str(d2 / "ud.fa"), str(d2 / "rd.fa"))
        assert open(cor, "rb").read() == (d2 / "cs.fa").read_bytes()
        assert open(ref, "rb").read() == (d2 / "rd.fa").read_bytes()
        assert open(unc, "rb").read() == (d2 / "ud.fa").read_bytes()
        # the prepared files are in lock step: one reference and one uncorrected record per corrected record
        nrec = open(cor, "rb").read().count(b">")
        assert nrec == len(cor_names) == open(ref, "rb").read().count(b">") == open(unc, "rb").read().count(b">")
