"""PoaEngine -- host-side handle on one GPU's triplet-MSA context.

Mirrors what one `poa` process does for ELECTOR (reference:
elector/alignment.py:59-63 -> src/poa-graph/main.c:241-287): a batch of
(reference, corrected, uncorrected) windows in, three MSA rows per window out.
All computation happens in the HIP library behind the C ABI.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import ElectorError, ElectorParams, ES_NCOUNTERS


def default_params():
    p = ElectorParams()
    _capi.lib().elector_params_default(C.byref(p))
    return p


def read_params(matrix_path):
    """a1: parse a poaV2 scoring-matrix file (what `-pathMatrix` names)."""
    p = ElectorParams()
    rc = _capi.lib().elector_params_read(str(matrix_path).encode(), C.byref(p))
    if rc:
        raise ElectorError(rc, str(matrix_path))
    return p


def pack_windows(triples):
    """[(ref, cor, unc) bytes] -> (bases uint8[total], off int64[3n+1])"""
    n = len(triples)
    off = np.zeros(3 * n + 1, dtype=np.int64)
    lens = np.fromiter((len(s) for t in triples for s in t), dtype=np.int64, count=3 * n)
    np.cumsum(lens, out=off[1:])
    bases = np.frombuffer(b"".join(s for t in triples for s in t), dtype=np.uint8)
    return bases, off


class PoaEngine:
    def __init__(self, device=0, params=None):
        self._lib = _capi.lib()
        self.params = params if params is not None else default_params()
        h = C.c_void_p()
        rc = self._lib.elector_ctx_create(int(device), C.byref(self.params), C.byref(h))
        if rc:
            raise ElectorError(rc)
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.elector_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow_window=False):
        if rc and not (allow_window and rc == _capi.E_WINDOW):
            raise ElectorError(rc, self._lib.elector_ctx_last_error(self._h).decode())

    # ---- host-buffer path (PCIe inclusive) --------------------------------
    def align_packed(self, bases, off, want_scores=False, strict=True):
        """bases uint8[total], off int64[3n+1] -> (rows uint8, row_off int64[n+1],
        ncol int32[n], status int32[n], scores int32[n,2] | None)"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.int64)
        n = (len(off) - 1) // 3
        cap = 3 * int(off[-1]) + 16
        rows = np.empty(cap, dtype=np.uint8)
        row_off = np.zeros(n + 1, dtype=np.int64)
        ncol = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        scores = np.zeros((n, 2), dtype=np.int32) if want_scores else None
        rc = self._lib.elector_poa_batch(
            self._h, n, bases.ctypes.data, off.ctypes.data, rows.ctypes.data, cap,
            row_off.ctypes.data, ncol.ctypes.data, status.ctypes.data,
            scores.ctypes.data if want_scores else None)
        self._check(rc, allow_window=not strict)
        return rows[: row_off[-1]], row_off, ncol, status, scores

    def align(self, triples, want_scores=False, strict=True):
        """[(ref, cor, unc)] -> list of (ref_row, cor_row, unc_row) bytes (None for failed windows)"""
        bases, off = pack_windows(triples)
        rows, row_off, ncol, status, scores = self.align_packed(bases, off, want_scores, strict)
        out = []
        buf = rows.tobytes()
        for w in range(len(triples)):
            if status[w]:
                out.append(None)
                continue
            a, nc = int(row_off[w]), int(ncol[w])
            out.append((buf[a:a + nc], buf[a + nc:a + 2 * nc], buf[a + 2 * nc:a + 3 * nc]))
        return (out, scores) if want_scores else out

    # ---- device-resident path (what bench.py times) -----------------------
    def align_device(self, d_bases, off, d_cols, d_ncol, d_status, d_scores=None):
        """torch CUDA tensors for the bulk data, numpy int64 host offsets.
        Enqueues on the engine's stream; call sync() before reading results."""
        off = np.ascontiguousarray(off, dtype=np.int64)
        n = (len(off) - 1) // 3
        rc = self._lib.elector_poa_batch_device(
            self._h, n, d_bases.data_ptr(), off.ctypes.data, d_cols.data_ptr(),
            d_ncol.data_ptr(), d_status.data_ptr(),
            d_scores.data_ptr() if d_scores is not None else None)
        self._check(rc)

    def align_device_offsets(self, d_bases, d_off, n, total, d_cols, d_ncol, d_status, d_scores=None):
        """align_device with the 3n + 1 window offsets resident in device memory as well (int64; what the device
        splitter leaves behind): no per-window work on the host at all.  d_bases / d_off: torch CUDA tensors or
        objects with data_ptr()."""
        rc = self._lib.elector_poa_batch_device_offsets(
            self._h, int(n), d_bases.data_ptr(), d_off.data_ptr(), int(total), d_cols.data_ptr(),
            d_ncol.data_ptr(), d_status.data_ptr(),
            d_scores.data_ptr() if d_scores is not None else None)
        self._check(rc)

    def msa_stats_device(self, n_windows, d_cols, d_ncol, d_status, piece_first, read_first, clips=None,
                         last_cap=0):
        """Second half of the MSA stage on the windows of the last align_device call:
        merge them into one record per piece and count (include/elector_stats.h).
        -> (counters int64[n_pieces, ES_NCOUNTERS], piece_cols int64[n_pieces],
            last_rows, last_mask)   -- the last two are None unless last_cap > 0."""
        piece_first = np.ascontiguousarray(piece_first, dtype=np.int64)
        read_first = np.ascontiguousarray(read_first, dtype=np.int64)
        n_pieces, n_reads = len(piece_first) - 1, len(read_first) - 1
        counters = np.zeros((n_pieces, ES_NCOUNTERS), dtype=np.int64)
        piece_cols = np.zeros(n_pieces, dtype=np.int64)
        if clips is not None:
            clips = np.ascontiguousarray(clips, dtype=np.int32)
        last_rows = np.zeros(3 * last_cap + 1, dtype=np.uint8) if last_cap > 0 else None
        last_mask = np.zeros(last_cap + 1, dtype=np.uint8) if last_cap > 0 else None
        rc = self._lib.elector_msa_stats_device(
            self._h, n_windows, d_cols.data_ptr(), d_ncol.data_ptr(), d_status.data_ptr(),
            n_pieces, piece_first.ctypes.data, n_reads, read_first.ctypes.data,
            clips.ctypes.data if clips is not None else None, counters.ctypes.data, piece_cols.ctypes.data,
            last_rows.ctypes.data if last_rows is not None else None,
            last_mask.ctypes.data if last_mask is not None else None, last_cap)
        self._check(rc)
        if last_cap > 0 and n_reads > 0:
            nl = int(piece_cols[int(read_first[-2]):].sum())
            last_rows, last_mask = last_rows[:3 * nl], last_mask[:nl]
        return counters, piece_cols, last_rows, last_mask

    def msa_stats_enqueue(self, n_windows, d_cols, d_ncol, d_status, piece_first, read_first, clips=None,
                          rows_out=None, rows_cap=0):
        """First half of msa_stats_device: queue merge + counters behind the POA kernels and
        return at once (at most two jobs in flight).  -> n_pieces, to pass to msa_stats_collect.
        rows_out (an address: page-locked host memory or device memory of rows_cap bytes, at least 3 per base of the
        batch): the merged rows are delivered there without a call in between (elector_msa_stats_enqueue_rows): device
        memory is complete when the job is collected, host memory after msa_rows_wait()."""
        piece_first = np.ascontiguousarray(piece_first, dtype=np.int64)
        read_first = np.ascontiguousarray(read_first, dtype=np.int64)
        if clips is not None:
            clips = np.ascontiguousarray(clips, dtype=np.int32)
        args = (self._h, n_windows, d_cols.data_ptr(), d_ncol.data_ptr(), d_status.data_ptr(),
                len(piece_first) - 1, piece_first.ctypes.data, len(read_first) - 1, read_first.ctypes.data,
                clips.ctypes.data if clips is not None else None)
        if rows_out is None:
            self._check(self._lib.elector_msa_stats_enqueue(*args))
        else:
            self._check(self._lib.elector_msa_stats_enqueue_rows(*args, C.c_void_p(int(rows_out)), int(rows_cap)))
        return len(piece_first) - 1

    def msa_stats_collect(self, n_pieces, last_cap=0):
        """Wait for the oldest queued job -> (counters int64[n_pieces, ES_NCOUNTERS], piece_cols); with
        last_cap > 0 (room in columns) also the rows and the statistics mask of the job's LAST read:
        -> (counters, piece_cols, last_rows uint8[3 * last_cap], last_mask uint8[last_cap]), per piece
        3 * cols row bytes resp. cols mask bytes back to back (the caller knows the read's pieces)."""
        counters = np.zeros((n_pieces, ES_NCOUNTERS), dtype=np.int64)
        piece_cols = np.zeros(n_pieces, dtype=np.int64)
        last_rows = np.zeros(3 * last_cap + 1, dtype=np.uint8) if last_cap > 0 else None
        last_mask = np.zeros(last_cap + 1, dtype=np.uint8) if last_cap > 0 else None
        self._check(self._lib.elector_msa_stats_collect(
            self._h, n_pieces, counters.ctypes.data, piece_cols.ctypes.data,
            last_rows.ctypes.data if last_rows is not None else None,
            last_mask.ctypes.data if last_mask is not None else None, last_cap))
        if last_cap > 0:
            return counters, piece_cols, last_rows, last_mask
        return counters, piece_cols

    def msa_rows_wait(self):
        """wait until the rows of every collected job that named a host `rows_out` have arrived there"""
        self._check(self._lib.elector_msa_rows_wait(self._h))

    def msa_rows_fetch(self, piece_cols):
        """Merged records of the last msa_stats_device call: uint8[3 * sum(piece_cols)],
        per piece the reference, corrected and uncorrected row back to back."""
        piece_cols = np.ascontiguousarray(piece_cols, dtype=np.int64)
        rows = np.zeros(3 * int(piece_cols.sum()) + 1, dtype=np.uint8)
        self._check(self._lib.elector_msa_rows_fetch(self._h, len(piece_cols), piece_cols.ctypes.data,
                                                     rows.ctypes.data))
        return rows[:-1]

    def msa_rows_fetch_into(self, piece_cols, host_ptr, capacity):
        """msa_rows_fetch into caller-owned host memory (pinned memory makes it one asynchronous DMA): `host_ptr`
        an address with room for `capacity` bytes.  -> bytes written."""
        piece_cols = np.ascontiguousarray(piece_cols, dtype=np.int64)
        need = 3 * int(piece_cols.sum())
        if need > int(capacity):
            raise ValueError("msa_rows_fetch_into: %d bytes needed, %d offered" % (need, capacity))
        self._check(self._lib.elector_msa_rows_fetch(self._h, len(piece_cols), piece_cols.ctypes.data, C.c_void_p(int(host_ptr))))
        return need

    # ---- a12: heaviest-bundle consensus (optional output) -------------------
    def keep_graph(self, on=True):
        """Make the following batches keep the graph data the bundle search needs."""
        self._check(self._lib.elector_ctx_keep_graph(self._h, 1 if on else 0))

    def bundles_enqueue(self, n, minimum_fraction=0.9):
        """the bundle search of the last batch, noted: the context queues it at its next call that waits for it anyway
        (statistics collect, sync, bundles(), the next batch); option("bundles_now", 1) queues it inside this call.
        Results stay on the device (timing: kind 5)"""
        self._check(self._lib.elector_poa_bundles_enqueue(self._h, int(n), float(minimum_fraction)))

    def bundles(self, n, total_bases, minimum_fraction=0.9):
        """generate_lpo_bundles(lpo, minimum_fraction) (heaviest_bundle.c:144-172) on every window of
        the last batch -> list of (consensus rows [bytes], counts [int], bundle ids (ref, cor, unc))."""
        cap = 3 * int(total_bases) + 16
        cons = np.zeros(cap, dtype=np.uint8)
        cons_off = np.zeros(n + 1, dtype=np.int64)
        info = np.zeros((n, 8), dtype=np.int32)
        self._check(self._lib.elector_poa_bundles(self._h, n, float(minimum_fraction), cons.ctypes.data, cap,
                                                  cons_off.ctypes.data, info.ctypes.data))
        buf = cons.tobytes()
        out = []
        for w in range(n):
            k, nc, a = int(info[w, 0]), int(info[w, 7]), int(cons_off[w])
            out.append(([buf[a + i * nc:a + (i + 1) * nc] for i in range(k)],
                        [int(x) for x in info[w, 1:1 + k]], tuple(int(x) for x in info[w, 4:7])))
        return out

    def align_with_bundles(self, triples, minimum_fraction=0.9):
        """[(ref, cor, unc)] -> (rows as align(), bundles as bundles())"""
        self.keep_graph(True)
        try:
            bases, off = pack_windows(triples)
            rows = self.align(triples, strict=False)
            return rows, self.bundles(len(triples), int(off[-1]), minimum_fraction)
        finally:
            self.keep_graph(False)

    def sync(self):
        self._check(self._lib.elector_ctx_sync(self._h))

    def timing_enable(self, on=True):
        self._check(self._lib.elector_ctx_timing_enable(self._h, 1 if on else 0))

    def timing_reset(self):
        self._check(self._lib.elector_ctx_timing_reset(self._h))

    def timing_read(self, kernel):
        ms, k = C.c_double(), C.c_int64()
        self._check(self._lib.elector_ctx_timing_read(self._h, kernel, C.byref(ms), C.byref(k)))
        return ms.value, k.value

    def option(self, name, value):
        """elector_ctx_option: "chains" = concurrent launch chains of the fused classes (0 = default); "priority" = -1 / 0 / +1,
        the context's streams at the device's highest / default / lowest priority; "cus" = lo * 1000 + hi, its streams on the
        compute units lo .. hi - 1 of the queue mask (both before the context's first call); "bundles_now" = 1: bundles_enqueue
        queues the search inside the call."""
        self._check(self._lib.elector_ctx_option(self._h, name.encode(), int(value)))

    def last_po_sizes(self, n):
        out = np.zeros(n, dtype=np.int32)
        self._check(self._lib.elector_ctx_last_po_sizes(self._h, n, out.ctypes.data))
        return out


class EnginePool:
    """Several engine contexts on one GPU that take consecutive batches in turn, so that several
    batches are in flight: the serial head and tail of one batch (symbolize, trivial pass, list
    sort; merge, statistics) run beside the alignment kernels of another (bench.py: three contexts
    are 15 % faster than one on 10,001-read batches).  Every context needs its own output buffers."""

    def __init__(self, device=0, n=2, params=None):
        self.engines = [PoaEngine(device, params) for _ in range(max(1, int(n)))]
        self._turn = 0

    def __len__(self):
        return len(self.engines)

    def next(self):
        """-> (index, engine) of the context that takes the next batch"""
        i = self._turn % len(self.engines)
        self._turn += 1
        return i, self.engines[i]

    def sync(self):
        for e in self.engines:
            e.sync()

    def close(self):
        for e in self.engines:
            e.close()

