"""Host-side binding of the window splitter and window merger
(include/elector_split.h): the stages either side of the POA engine.

Reference: src/split/Master_Splitter.cpp (masterSplitter) and
src/split/Donatello.cpp, spawned from elector/alignment.py:99-101,120-122.
"""
import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import ElectorError

READ_SETS = 4            # ELECTOR_READ_SETS of include/elector_split.h: buffer sets a ReadsFile hands out in turn


class ElectorWindows(C.Structure):
    _fields_ = [
        ("n_reads", C.c_int64), ("n_windows", C.c_int64),
        ("bases", C.POINTER(C.c_uint8)), ("off", C.POINTER(C.c_int64)),
        ("read_first", C.POINTER(C.c_int64)), ("read_index", C.POINTER(C.c_int64)),
        ("small_reads", C.c_int64), ("wrong_reads", C.c_int64),
    ]


class ElectorWindowsDev(C.Structure):
    _fields_ = [
        ("n_reads", C.c_int64), ("n_windows", C.c_int64), ("d_bases", C.c_void_p),
        ("off", C.POINTER(C.c_int64)), ("read_first", C.POINTER(C.c_int64)), ("read_index", C.POINTER(C.c_int64)),
        ("small_reads", C.c_int64), ("wrong_reads", C.c_int64), ("d_off", C.c_void_p),
    ]


class ElectorReads(C.Structure):
    _fields_ = [("n", C.c_int64), ("first_index", C.c_int64),
                ("seq", C.POINTER(C.c_uint8)), ("seq_off", C.POINTER(C.c_int64)),
                ("hdr", C.POINTER(C.c_uint8)), ("hdr_off", C.POINTER(C.c_int64))]


class ElectorReadsIndex(C.Structure):
    _fields_ = [("n", C.c_int64), ("len", C.POINTER(C.c_int64)), ("new_read", C.POINTER(C.c_uint8))]


class ElectorMsa(C.Structure):
    _fields_ = [
        ("n_reads", C.c_int64), ("rows", C.POINTER(C.c_uint8)),
        ("row_off", C.POINTER(C.c_int64)), ("cols", C.POINTER(C.c_int64)),
    ]


def _lib():
    L = _capi.lib()
    if not getattr(L, "_split_bound", False):
        L.elector_split_reads.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                          C.POINTER(ElectorWindows)]
        L.elector_windows_free.argtypes = [C.POINTER(ElectorWindows)]
        L.elector_windows_free.restype = None
        L.elector_merge_windows.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.POINTER(ElectorMsa)]
        L.elector_msa_free.argtypes = [C.POINTER(ElectorMsa)]
        L.elector_msa_free.restype = None
        L.elector_split_reads_device.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                                 C.POINTER(ElectorWindowsDev)]
        L.elector_windows_dev_free.argtypes = [C.POINTER(ElectorWindowsDev)]
        L.elector_windows_dev_free.restype = None
        L.elector_ctx_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.elector_ctx_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.elector_reads_open.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.elector_reads_next.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(ElectorReads)]
        L.elector_reads_close.argtypes = [C.c_void_p]
        L.elector_reads_close.restype = None
        L.elector_reads_scan.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(ElectorReadsIndex)]
        L.elector_reads_index_free.argtypes = [C.POINTER(ElectorReadsIndex)]
        L.elector_reads_index_free.restype = None
        L.elector_msa_format.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int64, C.c_int]
        L.elector_msa_format.restype = C.c_int64
        L.elector_msa_records_write.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int, C.c_int]
        L.elector_msa_records_write.restype = C.c_int64
        L.elector_msa_records_pwrite.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_int, C.c_int64, C.c_int]
        L.elector_msa_records_pwrite.restype = C.c_int64
        L._split_bound = True
    return L


def _np(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class Windows:
    """Result of split_reads: numpy copies of the library-owned buffers."""
    __slots__ = ("n_reads", "n_windows", "bases", "off", "read_first", "read_index", "small_reads", "wrong_reads")

    def triples(self):
        b = self.bases.tobytes()
        o = self.off
        return [(b[o[3 * w]:o[3 * w + 1]], b[o[3 * w + 1]:o[3 * w + 2]], b[o[3 * w + 2]:o[3 * w + 3]])
                for w in range(self.n_windows)]


def split_reads(reads, size_threshold=0.1, headers=None, nthreads=1):
    """reads: [(reference, corrected, uncorrected)] full-length read triples (bytes).
    headers: per-read header lines (incl. '>'); only their length matters.
    Returns Windows in the engine's layout (reference, corrected, uncorrected per window)."""
    n = len(reads)
    off = np.zeros(3 * n + 1, dtype=np.int64)
    # masterSplitter's own order: reference, uncorrected, corrected
    parts = [s for (r, c, u) in reads for s in (r, u, c)]
    lens = np.fromiter((len(s) for s in parts), dtype=np.int64, count=3 * n)
    np.cumsum(lens, out=off[1:])
    buf = np.frombuffer(b"".join(parts), dtype=np.uint8) if n else np.zeros(0, dtype=np.uint8)
    if headers is None:
        hl = np.full(n, 2, dtype=np.int32)
    else:
        hl = np.fromiter((len(h) for h in headers), dtype=np.int32, count=n)
    return split_packed(buf, off, hl, size_threshold, nthreads)


class DevBases:
    """device pointer to the window bases the device splitter left in its context (valid until that
    context's next splitter call); quacks like a torch tensor where PoaEngine.align_device needs it"""

    def __init__(self, ptr, nbytes, engine):
        self._ptr, self.nbytes, self._engine = int(ptr or 0), int(nbytes), engine

    def data_ptr(self):
        return self._ptr

    def to_tensor(self):
        """the bases in a torch tensor of their own (device to device)"""
        import torch
        t = torch.empty(self.nbytes + 64, dtype=torch.uint8, device=torch.device("cuda", self._engine.device))
        if self.nbytes:
            rc = _lib().elector_ctx_copy(self._engine._h, self._ptr, t.data_ptr(), self.nbytes)
            if rc:
                raise ElectorError(rc)
        return t

    def numpy(self):
        out = np.zeros(self.nbytes, dtype=np.uint8)
        if self.nbytes:
            rc = _lib().elector_ctx_copy_to_host(self._engine._h, self._ptr, out.ctypes.data, self.nbytes)
            if rc:
                raise ElectorError(rc)
        return out


class DevWindows(Windows):
    """Windows whose bases live in device memory (`d_bases`), and -- when the device splitter made them -- their
    offsets too (`d_off`: what PoaEngine.align_device_offsets takes); `.host_bases` fetches the bases (tests)."""
    __slots__ = ("d_bases", "d_off")

    @property
    def host_bases(self):
        return self.d_bases.numpy()


def pack_reads(reads, headers=None):
    """[(reference, corrected, uncorrected)] -> (buf, off, hdr_len) in masterSplitter's order
    (reference, uncorrected, corrected per read)"""
    n = len(reads)
    off = np.zeros(3 * n + 1, dtype=np.int64)
    parts = [s for (r, c, u) in reads for s in (r, u, c)]
    lens = np.fromiter((len(s) for s in parts), dtype=np.int64, count=3 * n)
    np.cumsum(lens, out=off[1:])
    buf = np.frombuffer(b"".join(parts), dtype=np.uint8) if n else np.zeros(0, dtype=np.uint8)
    if headers is None:
        hl = np.full(n, 2, dtype=np.int32)
    else:
        hl = np.fromiter((len(h) for h in headers), dtype=np.int32, count=n)
    return buf, off, hl


def split_reads_device(engine, reads, size_threshold=0.1, headers=None, nthreads=None):
    """The splitter on the GPU of `engine` (include/elector_split.h: elector_split_reads_device): same windows as
    split_reads, their bases stay in device memory.  A batch with a read beyond the kernel's on-chip limits is
    split by the host code and uploaded, so the result is always a DevWindows."""
    buf, off, hl = pack_reads(reads, headers)
    return split_packed_device(engine, buf, off, hl, size_threshold, nthreads)


def split_packed_device(engine, buf, off, hl, size_threshold=0.1, nthreads=None):
    """split_reads_device on reads already packed (pack_reads / ReadsFile.next)"""
    import os
    import torch
    L = _lib()
    nthreads = nthreads or os.cpu_count() or 1
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    hl = np.ascontiguousarray(hl, dtype=np.int32)
    n = (len(off) - 1) // 3
    w = ElectorWindowsDev()
    import time
    t_call = time.perf_counter()
    rc = L.elector_split_reads_device(engine._h, n, buf.ctypes.data, off.ctypes.data, hl.ctypes.data,
                                      float(size_threshold), int(nthreads), C.byref(w))
    t_done = time.perf_counter()
    out = DevWindows()
    if rc == _capi.E_LIMIT:
        hw = split_packed(buf, off, hl, size_threshold, nthreads)
        for k in ("n_reads", "n_windows", "off", "read_first", "read_index", "small_reads", "wrong_reads"):
            setattr(out, k, getattr(hw, k))
        t = torch.from_numpy(hw.bases).to(torch.device("cuda", engine.device))
        out.bases = None
        out.d_bases = t                      # a torch tensor has data_ptr() too
        out.d_off = torch.from_numpy(np.ascontiguousarray(hw.off, dtype=np.int64)).to(torch.device("cuda", engine.device))
        return out
    if rc:
        raise ElectorError(rc, L.elector_ctx_last_error(engine._h).decode())
    try:
        out.n_reads, out.n_windows = int(w.n_reads), int(w.n_windows)
        out.off = _np(w.off, 3 * out.n_windows + 1, np.int64) if out.n_windows else np.zeros(1, dtype=np.int64)
        out.read_first = _np(w.read_first, out.n_reads + 1, np.int64) if out.n_reads else np.zeros(1, dtype=np.int64)
        out.read_index = _np(w.read_index, out.n_reads, np.int64)
        out.small_reads, out.wrong_reads = int(w.small_reads), int(w.wrong_reads)
        out.bases = None
        out.d_bases = DevBases(w.d_bases, int(out.off[-1]), engine)
        out.d_off = DevBases(w.d_off, 8 * (3 * out.n_windows + 1), engine)
        if os.environ.get("ELECTOR_DEBUG_HOST"):
            import sys
            sys.stderr.write("[elector] split_packed_device: library call %.1f ms, arrays to numpy %.1f ms\n"
                             % (1e3 * (t_done - t_call), 1e3 * (time.perf_counter() - t_done)))
        return out
    finally:
        L.elector_windows_dev_free(C.byref(w))


class ReadBatch:
    """reads of one processing batch as the library's reader hands them out"""
    __slots__ = ("n", "first_index", "seq", "seq_off", "hdr", "hdr_off")

    def header(self, i):
        return self.hdr[int(self.hdr_off[i]):int(self.hdr_off[i + 1])]

    @property
    def hdr_len(self):
        return np.diff(self.hdr_off).astype(np.int32)


class ReadsFile:
    """The three sorted FASTA files read as masterSplitter reads them, cut into processing batches
    (include/elector_split.h: elector_reads_open / _next)."""

    def __init__(self, reference, uncorrected, corrected, device=None):
        self._h = C.c_void_p()
        if device is not None:
            _lib().elector_reads_set_device.argtypes = [C.c_int]
            _lib().elector_reads_set_device.restype = None
            _lib().elector_reads_set_device(int(device))      # its batch buffers: page-locked for that device's runtime
        rc = _lib().elector_reads_open(os.fsencode(reference), os.fsencode(uncorrected), os.fsencode(corrected),
                                       C.byref(self._h))
        if rc:
            raise ElectorError(rc, "cannot open the read files")

    def next(self, min_records, start=0, stop=None):
        """-> ReadBatch, or None at the end.  The batch's arrays are views of the reader's buffers: valid until
        the next call."""
        L = _lib()
        r = ElectorReads()
        rc = L.elector_reads_next(self._h, int(min_records), int(start), -1 if stop is None else int(stop), C.byref(r))
        if rc:
            raise ElectorError(rc)
        if r.n == 0:
            return None
        b = ReadBatch()
        b.n, b.first_index = int(r.n), int(r.first_index)
        b.seq_off = np.ctypeslib.as_array(r.seq_off, shape=(3 * b.n + 1,))
        total = int(b.seq_off[-1])
        b.seq = np.ctypeslib.as_array(r.seq, shape=(max(1, total),))[:total]
        b.hdr_off = np.ctypeslib.as_array(r.hdr_off, shape=(b.n + 1,)).copy()
        b.hdr = C.string_at(r.hdr, int(b.hdr_off[-1]))
        return b

    def close(self):
        if self._h:
            _lib().elector_reads_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass


def scan_reads(reference, uncorrected, corrected):
    """One native pass over the three files (include/elector_split.h: elector_reads_scan) -> (lr, lu, lc, new_read):
    per kept record its three sequence lengths and whether it opens a new read (its msa.fa header line differs
    from the previous record's)."""
    L = _lib()
    ix = ElectorReadsIndex()
    rc = L.elector_reads_scan(os.fsencode(reference), os.fsencode(uncorrected), os.fsencode(corrected), C.byref(ix))
    if rc:
        raise ElectorError(rc, "cannot scan the read files")
    try:
        n = int(ix.n)
        lens = _np(ix.len, 3 * n, np.int64).reshape(n, 3)
        return lens[:, 0], lens[:, 1], lens[:, 2], _np(ix.new_read, n, np.uint8).astype(bool)
    finally:
        L.elector_reads_index_free(C.byref(ix))


def msa_format(rows, piece_cols, headers, drop=None, nthreads=1):
    """Donatello's records of the pieces (rows: per piece its three rows back to back) -> bytes"""
    L = _lib()
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    piece_cols = np.ascontiguousarray(piece_cols, dtype=np.int64)
    hdr, hdr_off = pack_headers(headers)
    d = None if drop is None else np.ascontiguousarray(drop, dtype=np.uint8)
    need = L.elector_msa_format(len(piece_cols), rows.ctypes.data, piece_cols.ctypes.data, hdr.ctypes.data, hdr_off.ctypes.data,
                                d.ctypes.data if d is not None else None, None, 0, 1)
    if need < 0:
        raise ElectorError(int(need))
    out = np.empty(max(1, int(need)), dtype=np.uint8)
    got = L.elector_msa_format(len(piece_cols), rows.ctypes.data, piece_cols.ctypes.data, hdr.ctypes.data, hdr_off.ctypes.data,
                               d.ctypes.data if d is not None else None, out.ctypes.data, int(need), int(nthreads))
    if got != need:
        raise ElectorError(int(got) if got < 0 else -1)
    return out[:int(need)].tobytes()


def pack_headers(headers):
    """[bytes] -> (uint8 blob, int64 offsets)"""
    off = np.zeros(len(headers) + 1, dtype=np.int64)
    np.cumsum(np.fromiter((len(h) for h in headers), dtype=np.int64, count=len(headers)), out=off[1:])
    blob = np.frombuffer(b"".join(headers) + b"\0", dtype=np.uint8)
    return blob, off


def msa_records_write(engine, piece_cols, headers, drop, fd, nthreads=1):
    """The merged records of the engine's last collected statistics job appended to the file behind `fd`
    (include/elector_split.h: elector_msa_records_write) -> bytes written"""
    L = _lib()
    piece_cols = np.ascontiguousarray(piece_cols, dtype=np.int64)
    hdr, hdr_off = pack_headers(headers)
    d = None if drop is None else np.ascontiguousarray(drop, dtype=np.uint8)
    got = L.elector_msa_records_write(engine._h, len(piece_cols), piece_cols.ctypes.data, hdr.ctypes.data, hdr_off.ctypes.data,
                                      d.ctypes.data if d is not None else None, int(fd), int(nthreads))
    if got < 0:
        raise ElectorError(int(got), L.elector_ctx_last_error(engine._h).decode())
    return int(got)


def msa_records_pwrite(engine, piece_cols, headers, drop, fd, offset, nthreads=1):
    """msa_records_write at byte `offset` of a descriptor opened without O_APPEND, the page-cache copy on `nthreads`
    threads (include/elector_split.h: elector_msa_records_pwrite) -> bytes written"""
    L = _lib()
    piece_cols = np.ascontiguousarray(piece_cols, dtype=np.int64)
    hdr, hdr_off = pack_headers(headers)
    d = None if drop is None else np.ascontiguousarray(drop, dtype=np.uint8)
    got = L.elector_msa_records_pwrite(engine._h, len(piece_cols), piece_cols.ctypes.data, hdr.ctypes.data, hdr_off.ctypes.data,
                                       d.ctypes.data if d is not None else None, int(fd), int(offset), int(nthreads))
    if got < 0:
        raise ElectorError(int(got), L.elector_ctx_last_error(engine._h).decode())
    return int(got)


def split_packed(buf, off, hdr_len, size_threshold=0.1, nthreads=1):
    L = _lib()
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    hdr_len = np.ascontiguousarray(hdr_len, dtype=np.int32)
    n = (len(off) - 1) // 3
    w = ElectorWindows()
    rc = L.elector_split_reads(n, buf.ctypes.data, off.ctypes.data, hdr_len.ctypes.data,
                               float(size_threshold), int(nthreads), C.byref(w))
    if rc:
        raise ElectorError(rc)
    try:
        out = Windows()
        out.n_reads, out.n_windows = int(w.n_reads), int(w.n_windows)
        out.off = _np(w.off, 3 * out.n_windows + 1, np.int64)
        out.bases = _np(w.bases, int(out.off[-1]) if out.n_windows else 0, np.uint8)
        out.read_first = _np(w.read_first, out.n_reads + 1, np.int64)
        out.read_index = _np(w.read_index, out.n_reads, np.int64)
        out.small_reads, out.wrong_reads = int(w.small_reads), int(w.wrong_reads)
        return out
    finally:
        L.elector_windows_free(C.byref(w))


def merge_windows(read_first, rows, row_off, ncol):
    """Per-read concatenation of window rows with the 'n' columns of the
    corrected row dropped.  -> (rows uint8, row_off int64[n+1], cols int64[n])"""
    L = _lib()
    read_first = np.ascontiguousarray(read_first, dtype=np.int64)
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    row_off = np.ascontiguousarray(row_off, dtype=np.int64)
    ncol = np.ascontiguousarray(ncol, dtype=np.int32)
    n = len(read_first) - 1
    m = ElectorMsa()
    rc = L.elector_merge_windows(n, read_first.ctypes.data, rows.ctypes.data, row_off.ctypes.data,
                                 ncol.ctypes.data, C.byref(m))
    if rc:
        raise ElectorError(rc)
    try:
        ro = _np(m.row_off, n + 1, np.int64)
        return _np(m.rows, int(ro[-1]) if n else 0, np.uint8), ro, _np(m.cols, n, np.int64)
    finally:
        L.elector_msa_free(C.byref(m))
