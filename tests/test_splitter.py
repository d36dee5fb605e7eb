"""Window splitter and merger (host C++ behind the C ABI) against golden vectors
from the reference's masterSplitter / Donatello, and against the binaries
themselves where oracle/_ref exists."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import golden_io
import oracle_lib
import synth
from elector_amd import split


def test_splitter_golden():
    reads, wins, small, wrong = golden_io.splitter()
    got = split.split_reads([r[1] for r in reads], 0.1, [r[0] for r in reads], nthreads=3)
    assert got.triples() == [w[1] for w in wins]
    assert (got.small_reads, got.wrong_reads) == (small, wrong)
    # window -> read bookkeeping reproduces the headers masterSplitter wrote
    hdr = []
    for r in range(got.n_reads):
        hdr += [reads[int(got.read_index[r])][0]] * int(got.read_first[r + 1] - got.read_first[r])
    assert hdr == [w[0] for w in wins]


def test_splitter_thread_count_independent():
    reads = synth.read_triples(31, 12, 1500)
    a = split.split_reads(reads, 0.1, None, nthreads=1)
    b = split.split_reads(reads, 0.1, None, nthreads=5)
    assert a.triples() == b.triples() and a.read_first.tolist() == b.read_first.tolist()


def test_splitter_empty_and_tiny():
    w = split.split_reads([], 0.1)
    assert w.n_windows == 0 and w.n_reads == 0
    w = split.split_reads([(b"AC", b"AC", b"AC"), (b"ACG", b"ACG", b"ACG")], 0.1)
    # reference length <= 2 is skipped entirely; a 3-base read has no anchor chain -> AAA dummy
    assert w.n_reads == 1 and w.triples() == [(b"AAA", b"AAA", b"AAA")] and w.wrong_reads == 1


def test_merger_golden():
    s, m = golden_io.merger()
    n = len(s) // 6
    heads = [s[6 * w + 4] for w in range(n)]           # Donatello keys on the third header (Donatello.cpp:59,68)
    rows = [(s[6 * w + 1], s[6 * w + 3], s[6 * w + 5]) for w in range(n)]
    first = [0]
    for w in range(1, n):
        if heads[w] != heads[w - 1]:
            first.append(w)
    first.append(n)
    ncol = np.array([len(r[0]) for r in rows], dtype=np.int32)
    row_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(3 * ncol.astype(np.int64), out=row_off[1:])
    flat = np.frombuffer(b"".join(b"".join(r) for r in rows), dtype=np.uint8)
    out, ro, cols = split.merge_windows(np.array(first, dtype=np.int64), flat, row_off, ncol)
    buf = out.tobytes()
    got = []
    for r in range(len(first) - 1):
        h = heads[first[r]]
        hdr = h[: len(h) - 11] + b" "                  # Donatello.cpp:71-73
        a, nc = int(ro[r]), int(cols[r])
        got += [hdr, buf[a:a + nc], hdr, buf[a + nc:a + 2 * nc], hdr, buf[a + 2 * nc:a + 3 * nc]]
    assert got == m


@pytest.mark.reference
@pytest.mark.skipif(not oracle_lib.have_reference_binaries(), reason="oracle/_ref not built")
@pytest.mark.parametrize("case", ["plain", "trimmed", "noisy"])
def test_splitter_equals_reference_binary(tmp_path, case):
    rng = np.random.default_rng(77)
    if case == "plain":
        reads = synth.read_triples(41, 25, 4000)
    elif case == "noisy":
        reads = synth.read_triples(42, 25, 2500, err_unc=0.25, err_cor=0.1)
    else:
        reads = []
        for (r, c, u) in synth.read_triples(43, 25, 4000):
            k = int(rng.integers(0, 4))
            c = c[len(c) // 3:] if k == 0 else c[: len(c) // 2] if k == 1 else c[len(c) // 4: 3 * len(c) // 4] if k == 2 else c
            reads.append((r, c, u))
    headers = [b">rd%d_0" % i for i in range(len(reads))]
    d = str(tmp_path)
    with open(d + "/r.fa", "wb") as fr, open(d + "/u.fa", "wb") as fu, open(d + "/c.fa", "wb") as fc:
        for (r, c, u), h in zip(reads, headers):
            fr.write(h + b"\n" + r + b"\n")
            fc.write(h + b"\n" + c + b"\n")
            fu.write(h + b"\n" + u + b"\n")
    subprocess.run([os.path.join(oracle_lib.REF_DIR, "masterSplitter"), d + "/r.fa", d + "/u.fa", d + "/c.fa",
                    d + "/out1", d + "/out2", d + "/out3", "7", "200", "10000", "0.1", d], stdout=subprocess.DEVNULL)

    def cat(p):
        out = []
        for i in range(200):
            ls = open(d + "/" + p + str(i), "rb").read().split(b"\n")
            out += [ls[k + 1] for k in range(0, len(ls) - 1, 2)]
        return out
    exp = list(zip(cat("out1"), cat("out3"), cat("out2")))
    got = split.split_reads(reads, 0.1, headers, nthreads=4)
    assert got.triples() == exp
    assert got.small_reads == int(open(d + "/small_reads.txt").read())
    assert got.wrong_reads == int(open(d + "/wrongly_cor_reads.txt").read())
