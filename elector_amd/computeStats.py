"""Drop-in mirror of the reference's elector/computeStats.py for call site #2 of
the hot path (elector/__main__.py:141):

    outputRecallPrecision(correctedFileName, outDir, logFile, smallReadNumber,
                          wronglyCorrectedReadsNumber, reportedHomopolThreshold,
                          SIZE_CORRECTED_READ_THRESHOLD, fileSizeName, clipsNb,
                          beg=0, end=0, soft=None)  -> 19-tuple

Same signature, same return tuple (including the two str members), same stdout
block, log text and side files (`*per_read_metrics.txt`, the read-size file).
The per-column work (computeStats.py:61-189, 291-328, 371-440, 472-498, 712-752)
runs on the GPU behind the C ABI (include/elector_stats.h) and returns integers;
every ratio, mean and round() below is computed on the host in read order, so the
numbers do not depend on how reads are sharded over GPUs.

There is no CPU fallback: without the HIP library / a gfx950 device this raises.
"""
import ctypes as C
import os
import sys
import statistics

import numpy as np

from . import _capi
from ._capi import ElectorError

THRESH = 5     # computeStats.py:40
THRESH2 = 20   # computeStats.py:41

# counter indices (include/elector_stats.h)
(ES_TP, ES_FP, ES_FN, ES_COR, ES_UNC, ES_UCOR, ES_UUNC, ES_GC_REF, ES_GC_COR, ES_INS_U, ES_DEL_U, ES_SUB_U,
 ES_INS_C, ES_DEL_C, ES_SUB_C, ES_LEN_REF, ES_LEN_COR, ES_LEN_UNC, ES_GAPS_LEFT, ES_GAPS_RIGHT, ES_EXT_LEFT,
 ES_EXT_RIGHT, ES_MISSING, ES_MISSING_LAST, ES_PROCESSED, ES_NCOUNTERS) = range(26)

_engine = None


def _default_engine():
    """One context on this process's GPU (LOCAL_RANK when launched by torchrun)."""
    global _engine
    if _engine is None:
        from .poa import PoaEngine
        _engine = PoaEngine(int(os.environ.get("LOCAL_RANK", "0")))
    return _engine


def _bind(L):
    if not getattr(L, "_stats_bound", False):
        vp, i64 = C.c_void_p, C.c_int64
        L.elector_stats_batch.argtypes = [vp, i64, vp, i64, vp, vp, vp, vp, vp, vp]
        L.elector_homopolymer_pairs.restype = i64
        L.elector_homopolymer_pairs.argtypes = [i64, vp, vp, vp, vp, C.c_int32, vp, i64]
        L._stats_bound = True
    return L


class Pieces:
    """msa.fa parsed into the C-ABI layout."""
    __slots__ = ("headers", "header_nos", "rows", "row_off", "cols", "read_first", "n_split_reads_flags")


def getSplit(fileName):
    """computeStats.py:45-56 without the `grep | uniq -c` subprocess: run lengths of
    identical header lines // 3, keyed by the header with blanks removed."""
    readToSplit = dict()
    prev, cnt = None, 0

    def flush():
        if prev is not None:
            readToSplit[prev.split(">")[1].replace(" ", "").replace("\t", "")] = int(cnt / 3)
    with open(fileName) as f:
        for line in f:
            if ">" in line:
                line = line.rstrip("\n")
                if line == prev:
                    cnt += 1
                else:
                    flush()
                    prev, cnt = line, 1
    flush()
    return readToSplit


def parse_msa(fileName, readsToSplit):
    """Walk the file the way computeMetrics does (computeStats.py:546-658): six
    lines per piece, readsToSplit[header] pieces per read."""
    with open(fileName) as f:
        lines = f.readlines()
    headers, header_nos, chunks, cols, read_first = [], [], [], [], [0]
    nb = 0
    header_no = header = None
    while nb < len(lines):
        if ">" not in lines[nb]:
            nfrag = readsToSplit[header_no]
            for _ in range(nfrag if nfrag > 1 else 1):
                if nb >= len(lines):
                    break
                ref = lines[nb].rstrip(); nb += 2
                cor = lines[nb].rstrip() if nb < len(lines) else ""; nb += 2
                unc = lines[nb].rstrip() if nb < len(lines) else ""; nb += 1
                n = len(ref)
                if len(cor) != n or len(unc) != n:
                    raise ValueError("msa rows of unequal length for read %r" % header)
                headers.append(header)
                header_nos.append(header_no)
                chunks.append(ref + cor + unc)
                cols.append(n)
                if nb < len(lines):
                    header_no = lines[nb].split(">")[1].split(" ")[0]
                    header = lines[nb].split(">")[1].rstrip()
                    nb += 1
            read_first.append(len(cols))
        else:
            header_no = lines[nb].split(">")[1].split(" ")[0]
            header = lines[nb].split(">")[1].rstrip()
            nb += 1
    p = Pieces()
    p.headers, p.header_nos = headers, header_nos
    p.cols = np.asarray(cols, dtype=np.int64)
    p.row_off = np.zeros(len(cols) + 1, dtype=np.int64)
    np.cumsum(3 * p.cols, out=p.row_off[1:])
    p.rows = np.frombuffer("".join(chunks).encode("latin-1"), dtype=np.uint8)
    p.read_first = np.asarray(read_first, dtype=np.int64)
    return p


def stats_counters(pieces, clipsNb=None, engine=None):
    """-> (counters int64[n_pieces, ES_NCOUNTERS], last_mask uint8[...]) from the GPU."""
    engine = engine or _default_engine()
    L = _bind(_capi.lib())
    n_pieces = len(pieces.cols)
    n_reads = len(pieces.read_first) - 1
    counters = np.zeros((n_pieces, ES_NCOUNTERS), dtype=np.int64)
    clips = None
    if clipsNb:
        clips = np.zeros((n_pieces, 2), dtype=np.int32)
        for i, h in enumerate(pieces.headers):
            if h in clipsNb:
                clips[i, 0], clips[i, 1] = clipsNb[h][0], clipsNb[h][1]
    last_cols = int(pieces.cols[pieces.read_first[-2]:].sum()) if n_reads else 0
    last_mask = np.zeros(last_cols + 1, dtype=np.uint8)
    rc = L.elector_stats_batch(engine._h, n_reads, pieces.read_first.ctypes.data, n_pieces,
                               pieces.rows.ctypes.data, pieces.row_off.ctypes.data, pieces.cols.ctypes.data,
                               clips.ctypes.data if clips is not None else None, counters.ctypes.data,
                               last_mask.ctypes.data)
    if rc:
        raise ElectorError(rc, L.elector_ctx_last_error(engine._h).decode())
    return counters, last_mask[:last_cols]


def homopolymer_ratios(pieces, last_mask, threshold):
    """Ratios of the LAST read (the only ones the reference reports,
    computeStats.py:560,671-674): round(corrected_run / reference_run, 2) per homopolymer."""
    L = _bind(_capi.lib())
    n_reads = len(pieces.read_first) - 1
    if n_reads == 0:
        return []
    p0 = int(pieces.read_first[-2])
    npieces = len(pieces.cols) - p0
    cols = np.ascontiguousarray(pieces.cols[p0:])
    row_off = np.ascontiguousarray(pieces.row_off[p0:] - pieces.row_off[p0])
    rows = np.ascontiguousarray(pieces.rows[pieces.row_off[p0]:])
    cap = int(cols.sum()) + 1
    pairs = np.zeros((cap, 2), dtype=np.int32)
    mask = np.ascontiguousarray(last_mask, dtype=np.uint8)
    n = L.elector_homopolymer_pairs(npieces, rows.ctypes.data, row_off.ctypes.data, cols.ctypes.data,
                                    mask.ctypes.data, int(threshold), pairs.ctypes.data, cap)
    if n < 0:
        raise ElectorError(int(n))
    return [round(int(c) * 1.0 / int(r), 2) for c, r in pairs[:n]]


class ElectorReport(C.Structure):
    _fields_ = [("nb_reads", C.c_int64), ("throughput", C.c_int64), ("uncor_throughput", C.c_int64),
                ("precision", C.c_double), ("recall", C.c_double), ("cor_bases_rate", C.c_double), ("error_rate", C.c_double),
                ("uncor_cor_bases_rate", C.c_double), ("uncor_error_rate", C.c_double), ("gc_ref", C.c_double), ("gc_cor", C.c_double),
                ("indelsubs_unc", C.c_int64 * 3), ("indelsubs_cor", C.c_int64 * 3),
                ("count_split", C.c_int64), ("count_trimmed", C.c_int64), ("count_extended", C.c_int64),
                ("n_missing", C.c_int64), ("n_len_cor", C.c_int64), ("n_extended", C.c_int64),
                ("missing_size", C.POINTER(C.c_int64)), ("len_corrected", C.POINTER(C.c_int64)),
                ("extended_bases", C.POINTER(C.c_int64)), ("per_read_text", C.POINTER(C.c_char)),
                ("per_read_bytes", C.c_int64), ("flags", C.c_int32), ("pad", C.c_int32)]


def aggregate(pieces, counters, ratios, outPerReadMetrics):
    """The host half of computeMetrics (computeStats.py:519-675): per-read ratios in read order from the integer
    counters -- in the library (elector_report_aggregate, report_host.cpp); same return tuple as the reference's
    computeMetrics, the lines of per_read_metrics.txt written to outPerReadMetrics."""
    L = _capi.lib()
    if not getattr(L, "_report_bound", False):
        L.elector_report_aggregate.argtypes = [C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(ElectorReport)]
        L.elector_report_free.argtypes = [C.POINTER(ElectorReport)]
        L.elector_report_free.restype = None
        L.elector_read_size_lines.argtypes = [C.c_char_p, C.c_int]
        L.elector_read_size_lines.restype = C.c_int64
        L._report_bound = True
    read_first = np.ascontiguousarray(pieces.read_first, dtype=np.int64)
    counters = np.ascontiguousarray(counters, dtype=np.int64)
    n_reads = len(read_first) - 1
    n_pieces = counters.shape[0] if counters.ndim == 2 else 0
    rep = ElectorReport()
    rc = L.elector_report_aggregate(n_reads, read_first.ctypes.data, n_pieces, counters.ctypes.data, C.byref(rep))
    if rc:
        raise ElectorError(rc)
    try:
        if rep.flags & 3:
            # what the reference's own arithmetic does with no assessed read or a piece without letters
            raise ZeroDivisionError("division by zero")
        outPerReadMetrics.write(C.string_at(rep.per_read_text, rep.per_read_bytes).decode("ascii"))

        def lst(ptr, n):
            return np.ctypeslib.as_array(ptr, shape=(n,)).tolist() if n else []

        zero = bool(rep.flags & 4)
        meanRatioHomopolymers = statistics.mean(ratios) if len(ratios) > 1 else 1
        return (rep.nb_reads, rep.throughput, rep.uncor_throughput, 0 if zero else rep.precision, 0 if zero else rep.recall,
                0 if zero else rep.cor_bases_rate, rep.error_rate, 0 if zero else rep.uncor_cor_bases_rate,
                rep.uncor_error_rate, lst(rep.missing_size, rep.n_missing), rep.gc_ref, rep.gc_cor,
                list(rep.indelsubs_unc), list(rep.indelsubs_cor), meanRatioHomopolymers,
                lst(rep.len_corrected, rep.n_len_cor), rep.count_split, rep.count_trimmed, rep.count_extended,
                lst(rep.extended_bases, rep.n_extended))
    finally:
        L.elector_report_free(C.byref(rep))


def cached_pieces(fileName, clipsNb):
    """What elector_amd.alignment.getPOA left behind for this msa file, if the file is still exactly what it
    wrote: (Pieces with read_first / cols only, counters, Pieces of the last read with its rows, its mask).
    The counters were computed on the device while the MSAs were there (elector_msa_stats_enqueue), so the
    text file does not have to be parsed or uploaded again.  None when there is nothing usable: another
    file, a file appended to, soft clips to apply (computeStats.py:718-741 needs them per header)."""
    from . import alignment
    ent = alignment.MSA_CACHE.get(os.path.abspath(fileName))
    if ent is not None and ent.get("no_file") and not os.path.exists(fileName):
        # getPOA(write_msa=False): the run left no file, its counters are all there is
        if clipsNb:
            raise ValueError("soft clips are applied per header line of msa.fa (computeStats.py:718-741): run getPOA with "
                             "write_msa=True")
    else:
        if clipsNb or ent is None or ent.get("no_file"):
            return None
        try:
            st = os.stat(fileName)
        except OSError:
            return None
        if (st.st_size, st.st_mtime_ns) != ent["sig"]:
            return None
    # The reference finds a read's pieces through two different views of its header line (getSplit's key: the
    # line without blanks, computeStats.py:52; the walk's key: its first blank-delimited token, :548,612).  They
    # agree unless the header has a title; then the reference's own bookkeeping derails (KeyError, or pieces
    # counted under another read) and only the parsing path reproduces that.
    if not ent.get("plain_headers", False):
        return None
    p = Pieces()
    p.headers = p.header_nos = None
    p.cols = ent["cols"]
    p.read_first = ent["read_first"]
    p.rows = p.row_off = None
    last = Pieces()
    p0 = int(ent["last_first_piece"])
    last.cols = np.ascontiguousarray(ent["cols"][p0:])
    last.row_off = np.zeros(len(last.cols) + 1, dtype=np.int64)
    np.cumsum(3 * last.cols, out=last.row_off[1:])
    last.rows = np.ascontiguousarray(ent["last_rows"])
    last.read_first = np.asarray([0, len(last.cols)], dtype=np.int64)
    return p, ent["counters"], last, np.ascontiguousarray(ent["last_mask"])


def computeMetrics(fileName, outPerReadMetrics, correctedFileName, reportedThreshold, clipsNb, readsToSplit=None,
                   engine=None):
    """Same return tuple as the reference's computeMetrics (computeStats.py:519-675)."""
    hit = cached_pieces(fileName, clipsNb)
    if hit is not None:
        pieces, counters, last, last_mask = hit
        ratios = homopolymer_ratios(last, last_mask, reportedThreshold)
        return aggregate(pieces, counters, ratios, outPerReadMetrics)
    if readsToSplit is None:
        readsToSplit = getSplit(fileName)
    pieces = parse_msa(fileName, readsToSplit)
    counters, last_mask = stats_counters(pieces, clipsNb, engine)
    ratios = homopolymer_ratios(pieces, last_mask, reportedThreshold)
    return aggregate(pieces, counters, ratios, outPerReadMetrics)


def outputReadSizeDistribution(correctedFileName, outFileName, outDir, trimmedOrSplit, lenAllReads):
    """computeStats.py:273-286"""
    with open(outDir + "/" + outFileName, 'w') as out:
        out.write("size type\n")
        L = _capi.lib()
        arr = np.ascontiguousarray(np.asarray(lenAllReads, dtype=np.int64).reshape(-1))
        out.flush()
        L.elector_write_count_lines.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_int]
        L.elector_write_count_lines.restype = C.c_int64
        n = L.elector_write_count_lines(arr.ctypes.data, len(arr), b" reads", out.fileno())
        if n < 0:
            raise ElectorError(int(n))
        if trimmedOrSplit != 0:
            # the reference reads a header line, then a sequence line whose last character it drops, until a header
            # read comes back empty: one pass over the file in the library (elector_read_size_lines)
            L.elector_read_size_lines.argtypes = [C.c_char_p, C.c_int]
            L.elector_read_size_lines.restype = C.c_int64
            n = L.elector_read_size_lines(os.fsencode(correctedFileName), out.fileno())
            if n < 0:
                raise FileNotFoundError(correctedFileName) if n == _capi.E_IO else ElectorError(int(n))


def outputRecallPrecision(correctedFileName, outDir, logFile, smallReadNumber, wronglyCorrectedReadsNumber,
                          reportedHomopolThreshold, SIZE_CORRECTED_READ_THRESHOLD, fileSizeName, clipsNb,
                          beg=0, end=0, soft=None):
    """computeStats.py:196-263"""
    print(soft)
    if soft is not None:
        outMetrics = open(outDir + "/" + soft + "_per_read_metrics.txt", 'w')
        msa = outDir + "/msa_" + soft + ".fa"
    else:
        outMetrics = open(outDir + "/per_read_metrics.txt", 'w')
        msa = outDir + "/msa.fa"
    outMetrics.write("score metric\n")
    import time
    t_report = time.perf_counter()
    (nbReads, throughput, uncorThroughput, precision, recall, corBasesRate, errorRate, uncorCorBasesRate,
     uncorErrorRate, missingSize, GCRateRef, GCRateCorr, indelsubsUncorr, indelsubsCorr, ratioHomopolymers,
     lenAllCorrectedReads, countReadSplit, countReadTrimmed, countReadExtended, extendedBasesCount) = \
        computeMetrics(msa, outMetrics, correctedFileName, reportedHomopolThreshold, clipsNb)

    t_metrics = time.perf_counter()
    outputReadSizeDistribution(correctedFileName, fileSizeName, outDir, countReadSplit + countReadTrimmed,
                               lenAllCorrectedReads)
    outMetrics.close()
    if os.environ.get("ELECTOR_DEBUG_HOST"):
        sys.stderr.write("[elector] report: metrics %.1f ms, read size distribution %.1f ms\n"
                         % (1e3 * (t_metrics - t_report), 1e3 * (time.perf_counter() - t_metrics)))
    meanMissingSize = 0
    if countReadSplit + countReadTrimmed > 0:
        meanMissingSize = round(sum(missingSize) / (countReadSplit + countReadTrimmed), 1)
    meanExtendedBases = 0
    if countReadExtended > 0:
        meanExtendedBases = round(sum(extendedBasesCount) / countReadExtended, 1)
    if soft is not None:
        print(soft)

    recall = round(recall, 7)
    precision = round(precision, 7)
    corBasesRate = round(corBasesRate, 7)
    errorRate = round(errorRate, 7)
    GCRateRef = round(GCRateRef * 100, 7)
    GCRateCorr = round(GCRateCorr * 100, 7)

    # (stdout prefix, log prefix, value): one row per reported quantity.  print(a, b)
    # joins with one blank, so stdout prefixes carry the blank(s) the reference's
    # print calls produce (computeStats.py:232-262).
    pct = SIZE_CORRECTED_READ_THRESHOLD * 100
    report = [
        ("Assessed reads:  ", "Assessed reads: ", nbReads),
        ("Throughput (uncorrected) ", "\nThroughput (uncorrected): ", uncorThroughput),
        ("Throughput (corrected):  ", "\nThroughput (corrected): ", throughput),
        ("Recall: ", "\nRecall (computed only on corrected bases):", recall),
        ("Precision: ", "\nPrecision (computed only on corrected bases):", precision),
        ("Average correct bases rate (uncorrected):  ", "\nAverage correct bases rate (uncorrected):", uncorCorBasesRate),
        ("Error rate (uncorrected): ", "\nError rate (uncorrected): ", 1 - uncorCorBasesRate),
        ("Average correct bases rate (corrected):  ", "\nAverage correct bases rate (corrected):", corBasesRate),
        ("Error rate (corrected): ", "\nError rate (corrected): ", 1 - corBasesRate),
        ("Number of trimmed/split reads: ", "\nNumber of trimmed/split reads:", countReadSplit + countReadTrimmed),
        ("Mean missing size in trimmed/split reads: ", "\nMean missing size in trimmed/split reads:", meanMissingSize),
        ("Number of over-corrected reads by extention:  ", "\nNumber of over-corrected reads by extention: ", countReadExtended),
        ("Mean extension size in over-corrected reads:  ", "\nMean extension size in over-corrected reads: ", meanExtendedBases),
        ("%GC in reference reads:  ", "\n%GC in reference reads: ", GCRateRef),
        ("%GC in corrected reads:  ", "\n%GC in corrected reads: ", GCRateCorr),
        ("Number of corrected reads which length is < " + str(pct) + " % of the original read: ",
         "\nNumber of corrected reads which length is <" + str(pct) + "% of the original read:", smallReadNumber),
        ("Number of very low quality corrected reads:  ", "\nNumber of very low quality corrected reads: ", wronglyCorrectedReadsNumber),
        ("Number of insertions in uncorrected:  ", "\nNumber of insertions in uncorrected: ", indelsubsUncorr[0]),
        ("Number of insertions in corrected:  ", "\nNumber of insertions in corrected: ", indelsubsCorr[0]),
        ("Number of deletions in uncorrected:  ", "\nNumber of deletions in uncorrected: ", indelsubsUncorr[1]),
        ("Number of deletions in corrected:  ", "\nNumber of deletions in corrected: ", indelsubsCorr[1]),
        ("Number of substitutions in uncorrected:  ", "\nNumber of substitutions in uncorrected: ", indelsubsUncorr[2]),
        ("Number of substitutions in corrected:  ", "\nNumber of substitutions in corrected: ", indelsubsCorr[2]),
        ("Ratio of homopolymer sizes in corrected vs reference: ", "\nRatio of homopolymer sizes in corrected vs reference: ", ratioHomopolymers),
    ]
    banner = "*********** SUMMARY ***********"
    print(banner)
    for out_label, _, value in report:
        print(out_label + str(value))
    logFile.write(banner + "\n" + "".join(log_label + str(value) for _, log_label, value in report) + "\n")
    return (nbReads, throughput, precision, recall, corBasesRate, 1 - corBasesRate, smallReadNumber,
            wronglyCorrectedReadsNumber, GCRateRef, GCRateCorr, str(countReadSplit + countReadTrimmed),
            meanMissingSize, str(countReadExtended), meanExtendedBases, SIZE_CORRECTED_READ_THRESHOLD,
            indelsubsUncorr, indelsubsCorr, countReadSplit + countReadTrimmed, ratioHomopolymers)
