"""Build small msa.fa texts through the CPU oracle pipeline (splitter output ->
oracle POA -> merge), for the statistics tests.  Test infrastructure."""
import numpy as np

import oracle_lib
import synth
from elector_amd import split


def make_reads(seed, n, mean_len=1500):
    """-> [(header, ref, cor, unc)] with plain, trimmed, split (2-3 pieces) and
    extended corrected reads mixed, headers as ELECTOR's duplicateRefReads names them."""
    rng = np.random.default_rng(seed)
    out = []
    base = synth.read_triples(seed + 1, n, mean_len, min_len=300)
    for i, (r, c, u) in enumerate(base):
        kind = int(rng.integers(0, 8))
        name = b">read%d" % i
        if kind == 0:      # trimmed left
            out.append((name + b"_0", r, c[len(c) // 3:], u))
        elif kind == 1:    # trimmed right
            out.append((name + b"_0", r, c[: int(len(c) * 0.6)], u))
        elif kind == 2:    # split in two pieces
            a = len(c) // 2
            out.append((name + b"_0", r, c[: a - 60], u))
            out.append((name + b"_1", r, c[a + 60:], u))
        elif kind == 3:    # split in three
            a, b = len(c) // 3, 2 * len(c) // 3
            out.append((name + b"_0", r, c[: a - 40], u))
            out.append((name + b"_1", r, c[a + 40: b - 40], u))
            out.append((name + b"_2", r, c[b + 40:], u))
        elif kind == 4:    # extended corrected read
            ext = synth.random_seq(rng, int(rng.integers(25, 80)))
            out.append((name + b"_0", r, ext + c + synth.random_seq(rng, int(rng.integers(0, 60))), u))
        elif kind == 5:    # homopolymer-rich
            hp = (b"A" * 7 + b"C" + b"T" * 6 + b"GG")
            r2 = r[:200] + hp + r[200:400] + hp + r[400:]
            out.append((name + b"_0", r2, synth.mutate(rng, r2, 0.01), synth.mutate(rng, r2, 0.15)))
        else:
            out.append((name + b"_0", r, c, u))
    return out


def msa_text(reads, size_threshold=0.1):
    """-> (msa.fa text, small_reads, wrong_reads) exactly as ELECTOR's
    masterSplitter -> poa -> Donatello chain writes it."""
    hdrs = [h for (h, _, _, _) in reads]
    win = split.split_reads([(r, c, u) for (_, r, c, u) in reads], size_threshold, hdrs, nthreads=2)
    rows, ncol, _, _ = oracle_lib.batch(win.bases, win.off)
    flat = np.frombuffer(b"".join(b"".join(r) for r in rows), dtype=np.uint8)
    row_off = np.zeros(win.n_windows + 1, dtype=np.int64)
    np.cumsum(3 * ncol.astype(np.int64), out=row_off[1:])
    mr, mo, mc = split.merge_windows(win.read_first, flat, row_off, ncol)
    buf = mr.tobytes()
    out = []
    for k in range(win.n_reads):
        h = hdrs[int(win.read_index[k])] + b" untitled"          # poa prints ">name untitled" (fasta_format.c:35-37)
        hd = h[: len(h) - 11] + b" "                              # Donatello.cpp:71-73
        a, nc = int(mo[k]), int(mc[k])
        for r in range(3):
            out.append(hd)
            out.append(buf[a + r * nc: a + (r + 1) * nc])
    return (b"\n".join(out) + b"\n").decode(), win.small_reads, win.wrong_reads


def edge_reads_slots(seed=51):
    """a13 edge: more reads than one slot file holds (51), with runs of IDENTICAL header lines placed inside a
    slot (Donatello concatenates their windows into one record, Donatello.cpp:61-84) and across the slot
    boundaries 50|51 and 101|102 (different slot files: separate records)."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(130):
        ref = synth.random_seq(rng, int(rng.integers(260, 420)))
        cor = synth.mutate(rng, ref, 0.02)
        unc = synth.mutate(rng, ref, 0.12)
        name = b">r%03d_0" % i
        if i in (10, 11, 12):            # three in a row inside slot 0
            name = b">same_a_0"
        if i in (49, 50, 51, 52):        # 49, 50 in slot 0; 51, 52 in slot 1
            name = b">same_b_0"
        if i in (101, 102):              # 101 in slot 1; 102 in slot 2
            name = b">same_c_0"
        if i == 77:                      # a titled header: poa prints the title instead of "untitled"
            name = b">r077_0 some title"
        out.append((name, ref, cor, unc))
    return out


def edge_reads_batchcut(seed=52, n=10060):
    """a13 edge: more reads than one masterSplitter batch (10,001: Master_Splitter.cpp:397-399).  Nearly all
    reads are too short to anchor (they become `AAA` dummies, :417-423), which keeps the case cheap; the
    reads around the cut are real, and identical header lines sit on both sides of it (records 9999|10000
    share a slot, 10000|10001 are in different batches, 10001|10002 share slot 0 of the second batch)."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        real = 9990 <= i <= 10012 or i % 997 == 0
        if real:
            ref = synth.random_seq(rng, int(rng.integers(180, 260)))
            cor, unc = synth.mutate(rng, ref, 0.02), synth.mutate(rng, ref, 0.12)
        else:
            ref = synth.random_seq(rng, int(rng.integers(12, 20)))
            cor, unc = ref, ref
        name = b">q%05d_0" % i
        if i in (9999, 10000, 10001, 10002):
            name = b">cut_0"
        if i == 5000:                     # a record masterSplitter skips without counting (:414)
            ref, cor, unc = b"AC", b"AC", b"AC"
        out.append((name, ref, cor, unc))
    return out
