"""Drop-in mirror of the reference's elector/alignment.py for call site #1 of the
hot path (elector/__main__.py:140):

    getPOA(corrected, reference, uncorrected, threads, outDir,
           SIZE_CORRECTED_READ_THRESHOLD, soft=None) -> (small_reads, wrongly_cor_reads)

Same signature, same return value, same `outDir/msa.fa` (or `msa_<soft>.fa`)
bytes.  Where the reference spawns `masterSplitter`, 200 `poa` processes and 200
`Donatello` processes per batch and moves every window through files
(alignment.py:98-129), this keeps a batch in memory: the splitter and the merger
are host functions of the library (include/elector_split.h), the triplet MSAs
are computed on the GPU (include/elector_poa.h).  There is no CPU fallback.

Batch protocol kept from the reference: 10,001 reads per batch
(Master_Splitter.cpp:362,397-399: `i > max_nuc_amount`), reads with a reference
shorter than 3 bases are skipped without counting (:414), windows of a read are
merged in input order, the output file is opened in append mode like
Donatello's (Donatello.cpp:48).
"""
import os
import sys

import numpy as np

from . import split
from .poa import PoaEngine, read_params, default_params

READS_PER_BATCH = 10001          # alignment.py:82 amount_read = 10000, splitter stops after i > amount
READS_PER_SLOT = 10000 // 200 + 1   # Master_Splitter.cpp:366-369: slot = i / (max/nb_file + 1)

_engine = None


def _get_engine(matrix_path=None):
    global _engine
    if _engine is None:
        params = read_params(matrix_path) if matrix_path else default_params()
        _engine = PoaEngine(int(os.environ.get("LOCAL_RANK", "0")), params)
    return _engine


def _records(path):
    """Header line / sequence line pairs as masterSplitter's getline pairs read them
    (Master_Splitter.cpp:407-412)."""
    with open(path, "rb") as f:
        while True:
            h = f.readline()
            if not h:
                return
            s = f.readline()
            yield h.rstrip(b"\n"), s.rstrip(b"\n")


def _poa_header(href):
    """What `poa` prints for a window whose FASTA header line is href: the reader
    takes the first blank-delimited token after '>' as the name and the rest of
    the line as the title, "untitled" when there is none (fasta_format.c:33-37),
    and the writer prints '>name title' (lpo_format.c:410)."""
    body = href[1:].lstrip()
    i = 0
    while i < len(body) and not body[i:i + 1].isspace():
        i += 1
    name, rest = body[:i], body[i:].lstrip()
    return b">" + name + b" " + (rest if rest else b"untitled")


def _donatello_header(h):
    """Donatello prints header.substr(0, header.size() - 11) + " " (Donatello.cpp:71-73); the
    subtraction is unsigned, so a header line shorter than 11 bytes wraps around and is kept whole."""
    return (h if len(h) < 11 else h[: len(h) - 11]) + b" "


def align_batch(engine, reads, headers, size_threshold, threads):
    """One batch: [(reference, corrected, uncorrected)] + header lines ->
    (list of (header_out, ref_row, cor_row, unc_row) per output record, small, wrong)."""
    win = split.split_reads(reads, size_threshold, headers, nthreads=max(1, int(threads)))
    if win.n_windows == 0:
        return [], win.small_reads, win.wrong_reads
    rows, row_off, ncol, status, _ = engine.align_packed(win.bases, win.off)
    # Donatello concatenates consecutive windows with the same header inside one
    # slot file (Donatello.cpp:61-84); reads keep distinct headers in ELECTOR, so
    # this is one record per read unless two neighbours share a header line.
    hdr = [_poa_header(headers[int(i)]) for i in win.read_index]
    groups = [0]
    for r in range(1, win.n_reads):
        same_slot = (r // READS_PER_SLOT) == ((r - 1) // READS_PER_SLOT)
        if not (same_slot and hdr[r] == hdr[r - 1]):
            groups.append(r)
    groups.append(win.n_reads)
    first = np.array([win.read_first[g] for g in groups], dtype=np.int64)
    mrows, moff, mcols = split.merge_windows(first, rows, row_off, ncol)
    buf = mrows.tobytes()
    out = []
    for k in range(len(groups) - 1):
        h = hdr[groups[k]]
        a, nc = int(moff[k]), int(mcols[k])
        out.append((_donatello_header(h), buf[a:a + nc], buf[a + nc:a + 2 * nc], buf[a + 2 * nc:a + 3 * nc]))
    return out, win.small_reads, win.wrong_reads


def getPOA(corrected, reference, uncorrected, threads, outDir, SIZE_CORRECTED_READ_THRESHOLD, soft=None,
           engine=None, matrix=None):
    """elector/alignment.py:67-131"""
    amount_read = 1000 * 10
    print("- Means that a large amount of reads has been handled: " + str(amount_read))
    engine = engine or _get_engine(matrix)
    if soft is not None:
        mergeOut = outDir + "/msa_" + soft + ".fa"
    else:
        mergeOut = outDir + "/msa.fa"
    small_reads = 0
    wrongly_cor_reads = 0
    it_ref, it_unc, it_cor = _records(reference), _records(uncorrected), _records(corrected)
    done = False
    with open(mergeOut, "ab") as out:
        while not done:
            reads, headers = [], []
            while len(reads) < READS_PER_BATCH:
                try:
                    href, ref = next(it_ref)
                    _, unc = next(it_unc)
                    _, cor = next(it_cor)
                except StopIteration:
                    done = True
                    break
                if len(ref) > 2:                      # Master_Splitter.cpp:414
                    reads.append((ref, cor, unc))
                    headers.append(href)
            records, small, wrong = align_batch(engine, reads, headers, SIZE_CORRECTED_READ_THRESHOLD, threads)
            small_reads += small
            wrongly_cor_reads += wrong
            for h, r0, r1, r2 in records:
                out.write(h + b"\n" + r0 + b"\n" + h + b"\n" + r1 + b"\n" + h + b"\n" + r2 + b"\n")
            with open(outDir + "/small_reads.txt", "w") as f:
                f.write(str(small) + "\n")
            with open(outDir + "/wrongly_cor_reads.txt", "w") as f:
                f.write(str(wrong) + "\n")
            sys.stdout.write('-' * 200)
            sys.stdout.flush()
    return small_reads, wrongly_cor_reads
