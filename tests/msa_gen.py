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
