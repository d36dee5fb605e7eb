"""The native file ends of call site #1 (elector_amd/csrc/io_host.cpp) against their Python restatements:
the record loop + batch rule of elector_amd.alignment (_batches), and Donatello's record layout."""
import os
import random

import numpy as np
import pytest

from elector_amd import alignment, split


def _write(tmp_path, records, trailing_newline=(True, True, True), drop_last=(0, 0, 0)):
    paths = []
    for which, name in enumerate(("ref.fa", "unc.fa", "cor.fa")):
        p = tmp_path / name
        recs = records[: len(records) - drop_last[which]]
        txt = b"".join(h + b"\n" + seqs[which] + b"\n" for h, seqs in recs)
        if not trailing_newline[which] and txt.endswith(b"\n"):
            txt = txt[:-1]
        p.write_bytes(txt)
        paths.append(str(p))
    return paths


def _records(n, seed, titled=False):
    rng = random.Random(seed)
    out = []
    read = 0
    while len(out) < n:
        pieces = rng.choice((1, 1, 1, 2, 3))
        for _ in range(pieces):
            name = b">read%d_%s" % (read, b"x" * rng.randint(0, 14))
            if titled and rng.random() < 0.3:
                name += rng.choice((b" ", b"\t", b"  ")) + b"title %d" % rng.randint(0, 9)
            lr = rng.choice((0, 1, 2, 3, 5, 40, 200))
            seqs = tuple(bytes(rng.choice(b"ACGTacgtN") for _ in range(max(0, lr + rng.randint(-2, 2)))) for _ in range(3))
            seqs = (bytes(rng.choice(b"ACGT") for _ in range(lr)),) + seqs[1:]
            out.append((name, seqs))
        read += 1
    return out[:n]


def _python_batches(paths, min_records, start, stop, monkeypatch):
    monkeypatch.setattr(alignment, "READS_PER_BATCH", min_records)
    return [(first, reads, hdrs) for first, reads, hdrs in alignment._batches(paths[0], paths[1], paths[2], start, stop)]


def _native_batches(paths, min_records, start, stop):
    rf = split.ReadsFile(paths[0], paths[1], paths[2])
    out = []
    while True:
        b = rf.next(min_records, start, stop)
        if b is None:
            break
        reads, hdrs = [], []
        o = b.seq_off
        raw = b.seq.tobytes()
        for i in range(b.n):
            ref, unc, cor = (raw[o[3 * i + j]:o[3 * i + j + 1]] for j in range(3))
            reads.append((ref, cor, unc))
            hdrs.append(b.header(i))
        assert list(b.hdr_len) == [len(h) for h in hdrs]
        out.append((b.first_index, reads, hdrs))
    rf.close()
    return out


@pytest.mark.parametrize("titled", [False, True])
@pytest.mark.parametrize("min_records", [1, 7, 64, 1000])
def test_reader_matches_python(tmp_path, monkeypatch, titled, min_records):
    paths = _write(tmp_path, _records(300, 5 + min_records, titled))
    for start, stop in ((0, None), (0, 50), (13, 140), (299, None), (400, None)):
        want = _python_batches(paths, min_records, start, stop, monkeypatch)
        got = _native_batches(paths, min_records, start, stop)
        assert got == want, (start, stop)


def test_reader_ragged_files_and_missing_newlines(tmp_path, monkeypatch):
    recs = _records(120, 77, True)
    for k, (tn, dl) in enumerate((((False, True, True), (0, 0, 0)), ((True, False, False), (0, 0, 0)),
                                  ((True, True, True), (0, 9, 0)), ((True, True, True), (0, 0, 31)),
                                  ((False, False, False), (5, 0, 2)))):
        d = tmp_path / ("c%d" % k)
        d.mkdir()
        paths = _write(d, recs, tn, dl)
        assert _native_batches(paths, 16, 0, None) == _python_batches(paths, 16, 0, None, monkeypatch)


def test_reader_empty_and_missing(tmp_path):
    paths = _write(tmp_path, [])
    assert _native_batches(paths, 10, 0, None) == []
    with pytest.raises(Exception):
        split.ReadsFile(str(tmp_path / "nope.fa"), paths[1], paths[2])


def test_msa_format_matches_python():
    rng = np.random.default_rng(3)
    n = 257
    cols = rng.integers(0, 300, n).astype(np.int64)
    cols[5] = 0
    rows = rng.integers(97, 123, int(3 * cols.sum()), dtype=np.uint8)
    hdrs = [b">r%d " % i * (1 + i % 3) for i in range(n)]
    drop = (rng.random(n) < 0.1).astype(np.uint8)
    raw = rows.tobytes()

    def python(dropv):
        at, out = 0, []
        for p in range(n):
            nc = int(cols[p])
            if dropv is None or not dropv[p]:
                h = hdrs[p]
                out.append(h + b"\n" + raw[at:at + nc] + b"\n" + h + b"\n" + raw[at + nc:at + 2 * nc] + b"\n" +
                           h + b"\n" + raw[at + 2 * nc:at + 3 * nc] + b"\n")
            at += 3 * nc
        return b"".join(out)

    for threads in (1, 4):
        assert split.msa_format(rows, cols, hdrs, None, threads) == python(None)
        assert split.msa_format(rows, cols, hdrs, drop, threads) == python(drop)
    assert split.msa_format(np.zeros(0, np.uint8), np.zeros(0, np.int64), [], None, 2) == b""


def _python_scan(paths):
    lr, lu, lc, fresh, last = [], [], [], [], None
    for k, href, ref, cor, unc in alignment._triples(paths[0], paths[1], paths[2]):
        key = alignment._donatello_header(alignment._poa_header(href))
        lr.append(len(ref)); lu.append(len(unc)); lc.append(len(cor))
        fresh.append(key != last)
        last = key
    return lr, lu, lc, fresh


def test_scan_matches_python(tmp_path):
    """elector_reads_scan (the one native pass behind the multi-GPU shard bounds) against the Python record loop:
    same kept records, same lengths, same read boundaries -- also on ragged files and missing final newlines."""
    recs = _records(300, 91, True)
    for k, (tn, dl) in enumerate((((True, True, True), (0, 0, 0)), ((False, False, False), (0, 0, 0)),
                                  ((True, True, True), (0, 9, 0)), ((True, False, True), (0, 0, 31)),
                                  ((True, True, True), (300, 0, 0)))):
        d = tmp_path / ("s%d" % k)
        d.mkdir()
        paths = _write(d, recs, tn, dl)
        lr, lu, lc, fresh = split.scan_reads(paths[0], paths[1], paths[2])
        want = _python_scan(paths)
        assert (lr.tolist(), lu.tolist(), lc.tolist(), fresh.tolist()) == (want[0], want[1], want[2], want[3]), k
    with pytest.raises(Exception):
        split.scan_reads(str(tmp_path / "nope.fa"), paths[1], paths[2])


def test_shard_bounds_cut_at_read_boundaries(tmp_path):
    """alignment._shard: contiguous record ranges, never inside a run of records with one header line, the empty
    shard of a world larger than the number of reads, a read that outweighs all others"""
    recs = [(b">big_0", (b"ACGT" * 5000, b"ACGT" * 5000, b"ACGT" * 5000))]
    for i in range(6):
        for j in range(1 + i % 3):
            recs.append((b">r%d_%d" % (i, j), (b"ACGTA" * 40, b"ACGTA" * 40, b"ACGTA" * 40)))
    paths = _write(tmp_path, recs)
    _, _, _, fresh = split.scan_reads(paths[0], paths[1], paths[2])
    read_starts = set(np.nonzero(fresh)[0].tolist()) | {len(recs)}
    for world in (1, 2, 4, 8, 16):
        b = alignment._shard(paths[0], paths[1], paths[2], world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == len(recs) and all(x <= y for x, y in zip(b, b[1:]))
        assert set(b) <= read_starts
        if world > 1:
            assert (0, 1) in set(zip(b, b[1:]))                 # the dominant read is a shard of its own
        if world == 16:
            assert any(x == y for x, y in zip(b, b[1:]))        # more ranks than reads: some shard is empty


def test_reader_batches_stay_valid_for_the_buffer_sets(tmp_path):
    """include/elector_split.h: a batch's buffers stay valid until ELECTOR_READ_SETS - 1 more calls have been made
    (getPOA's parser reads that far ahead of its splitter threads)."""
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "elector_split.h")).read()
    assert int(re.search(r"#define ELECTOR_READ_SETS (\d+)", hdr).group(1)) == split.READ_SETS >= 3
    paths = _write(tmp_path, _records(400, 91, False))
    rf = split.ReadsFile(paths[0], paths[1], paths[2])
    held = []                       # (batch, copy of its bytes at the time it was handed out)
    while True:
        b = rf.next(9, 0, None)
        if b is None:
            break
        held.append((b, b.seq.tobytes(), bytes(b.hdr)))
        for old, seq, hd in held[-split.READ_SETS:]:
            assert old.seq.tobytes() == seq and bytes(old.hdr) == hd
    assert len(held) > 2 * split.READ_SETS
    rf.close()
