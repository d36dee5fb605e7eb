"""Drop-in mirror of the reference's elector/alignment.py for call site #1 of the
hot path (elector/__main__.py:140):

    getPOA(corrected, reference, uncorrected, threads, outDir,
           SIZE_CORRECTED_READ_THRESHOLD, soft=None) -> (small_reads, wrongly_cor_reads)

Same signature, same return value, same `outDir/msa.fa` (or `msa_<soft>.fa`)
bytes.  Where the reference spawns `masterSplitter`, 200 `poa` processes and 200
`Donatello` processes per batch and moves every window through files
(alignment.py:98-129), a batch here never leaves memory, and the bulk of it never
leaves the GPU:

    parser thread   three FASTA files -> batches of reads (elector_reads_next, native; four buffer sets in turn)
    splitter thread batches -> HBM -> windows (elector_split_reads_device: one workgroup per read on a splitter
                    context of its own; ELECTOR_HOST_SPLIT=1 keeps the host threads of elector_split_reads instead)
    main thread     windows (already in HBM) -> triplet MSAs (elector_poa_batch_device) -> one record per piece
                    (k_merge) -> per-piece integer counters (k_stats) -> merged rows back and into msa.fa
                    (elector_msa_records_write)

Two engine contexts take the batches in turn, so the host work of one batch
(parsing, splitting, writing msa.fa) overlaps the kernels of the other.  The
per-piece counters the device computed on the way are kept for call site #2
(elector_amd.computeStats.outputRecallPrecision), which then does not have to
parse msa.fa again.  There is no CPU fallback.

Batch protocol kept from the reference, because it decides bytes of msa.fa:
10,001 reads per masterSplitter batch (Master_Splitter.cpp:362,397-399:
`i > max_nuc_amount`), reads with a reference shorter than 3 bases are skipped
without counting (:414), a read lands in slot file `k // 51` of its batch
(:366-369,435) and Donatello concatenates neighbouring windows with the same
header line inside one slot file (Donatello.cpp:61-84), the output file is opened
in append mode like Donatello's (Donatello.cpp:48).  A processing batch here is a
reference batch extended to the end of its last read (all pieces of a read are
counted together), which changes nothing in the file: the slot of a record is
computed from its position in the reference's own batches.

Multi-GPU (one process per GPU, torch.distributed initialised by the caller or
by torchrun): every rank takes a contiguous range of reads balanced by DP cells,
writes its part of msa.fa, rank 0 concatenates the parts in rank order and
receives every rank's counters in one gather (RCCL over xGMI when the backend is
nccl); no collective touches the alignment data.
"""
import os
import queue
import sys
import threading

import numpy as np

from . import split
from ._capi import ElectorError, ES_NCOUNTERS
from .poa import EnginePool, PoaEngine, read_params, default_params

READS_PER_BATCH = 10001          # alignment.py:82 amount_read = 10000, splitter stops after i > amount
READS_PER_SLOT = 10000 // 200 + 1   # Master_Splitter.cpp:366-369: slot = i / (max/nb_file + 1)

_pool = None
_splitters = {}
# seconds spent per stage since the last reset, summed over batches (bench_e2e.py reads them; the reader
# thread's stages overlap the main thread's)
STAGE_SECONDS = {}


STAGE_TRACE = [] if os.environ.get("ELECTOR_STAGE_TRACE") else None      # (stage, thread, start, end) of every stage instance


def _tick(stage, t0):
    import time
    t1 = time.perf_counter()
    STAGE_SECONDS[stage] = STAGE_SECONDS.get(stage, 0.0) + (t1 - t0)
    if STAGE_TRACE is not None:
        STAGE_TRACE.append((stage, threading.current_thread().name, t0, t1))


# msa path -> what outputRecallPrecision needs instead of the text file (see computeStats.cached_pieces)
MSA_CACHE = {}


def _get_pool(matrix_path=None, n=None):
    global _pool
    if _pool is None:
        params = read_params(matrix_path) if matrix_path else default_params()
        if n is None:
            # a context is busy from a batch's first kernel until its records are in the page cache (its output
            # buffers, its pinned rows and its text buffer): with two, the writers' ~70 ms per batch paced the whole
            # pipeline at one batch per 35 ms
            n = int(os.environ.get("ELECTOR_ENGINES", "3"))
        _pool = EnginePool(int(os.environ.get("LOCAL_RANK", "0")), n, params)
        # ELECTOR_ENGINE_CUS=lo:hi: the alignment contexts' streams on the compute units lo .. hi - 1 of the queue mask
        cus = os.environ.get("ELECTOR_ENGINE_CUS", "")
        if cus:
            lo, hi = (int(x) for x in cus.split(":"))
            for g in _pool.engines:
                g.option("cus", lo * 1000 + hi)
    return _pool


def _get_splitter(device, k=0):
    """a splitter thread's engine context: the device splitter works on its own streams and workspace"""
    if os.environ.get("ELECTOR_HOST_SPLIT", "0") not in ("", "0"):
        return None
    if (device, k) not in _splitters:
        _splitters[(device, k)] = PoaEngine(device)
        # ELECTOR_SPLIT_PRIORITY=-1|0|1: the splitter's streams at the device's highest / default / lowest priority (its
        # long-lived workgroups and the alignment kernels' wavefronts share the chip)
        prio = int(os.environ.get("ELECTOR_SPLIT_PRIORITY", "0") or 0)
        if prio:
            _splitters[(device, k)].option("priority", prio)
        # ELECTOR_SPLIT_CUS=lo:hi: the splitter's streams on the compute units lo .. hi - 1 of the queue mask (of 256)
        cus = os.environ.get("ELECTOR_SPLIT_CUS", "")
        if cus:
            lo, hi = (int(x) for x in cus.split(":"))
            _splitters[(device, k)].option("cus", lo * 1000 + hi)
    return _splitters[(device, k)]


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


class _Batch:
    """One processing batch: reads, their msa.fa header lines, and where its first read stands in the
    reference's own batch protocol."""
    __slots__ = ("first_index", "win", "piece_first", "read_first", "rec_hdr", "small", "wrong", "last", "d_bases", "d_off", "plain")


def _triples(reference, uncorrected, corrected, start=0, stop=None):
    """(header line of the reference record, reference, corrected, uncorrected) of the records
    [start, stop) that masterSplitter does not skip, with their index among the kept records."""
    it_ref, it_unc, it_cor = _records(reference), _records(uncorrected), _records(corrected)
    k = 0
    while True:
        try:
            href, ref = next(it_ref)
            _, unc = next(it_unc)
            _, cor = next(it_cor)
        except StopIteration:
            return
        if len(ref) > 2:                              # Master_Splitter.cpp:414
            if stop is not None and k >= stop:
                return
            if k >= start:
                yield k, href, ref, cor, unc
            k += 1


def _batches(reference, uncorrected, corrected, start=0, stop=None):
    """Processing batches: at least READS_PER_BATCH records, extended to the end of the last read
    (= run of records with one msa.fa header line, computeStats.py:45-56)."""
    cur, cur_hdr, first = [], [], None
    last_key = None
    for k, href, ref, cor, unc in _triples(reference, uncorrected, corrected, start, stop):
        key = _donatello_header(_poa_header(href))
        if len(cur) >= READS_PER_BATCH and key != last_key:
            yield first, cur, cur_hdr
            cur, cur_hdr, first = [], [], None
        if first is None:
            first = k
        cur.append((ref, cor, unc))
        cur_hdr.append(href)
        last_key = key
    if cur:
        yield first, cur, cur_hdr


def _prepare(rb, size_threshold, threads, splitter=None):
    """Reader half of a batch (rb: split.ReadBatch): windows, record boundaries (Donatello's same-header rule),
    read boundaries."""
    import time
    b = _Batch()
    b.first_index = first_index = rb.first_index
    t0 = time.perf_counter()
    b.d_bases = b.d_off = None
    if splitter is not None:
        win = split.split_packed_device(splitter, rb.seq, rb.seq_off, rb.hdr_len, size_threshold, nthreads=max(1, int(threads)))
        # out of the splitter's workspace, which its next call reuses
        t1 = time.perf_counter()
        b.d_bases = win.d_bases.to_tensor() if isinstance(win.d_bases, split.DevBases) else win.d_bases
        # ... and the offsets with them: the engine then takes the windows without a per-window pass on the host
        d_off = getattr(win, "d_off", None)
        b.d_off = d_off.to_tensor() if isinstance(d_off, split.DevBases) else d_off
        win.d_bases = win.d_off = None
        if os.environ.get("ELECTOR_DEBUG_HOST"):
            sys.stderr.write("[elector] split stage: call + arrays %.1f ms, bases to a tensor of their own %.1f ms\n"
                             % (1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1)))
        _tick("split (device, incl. reads H2D)", t0)
    else:
        win = split.split_packed(rb.seq, rb.seq_off, rb.hdr_len, size_threshold, nthreads=max(1, int(threads)))
        _tick("split (host threads)", t0)
    t0 = time.perf_counter()
    b.win, b.small, b.wrong = win, win.small_reads, win.wrong_reads
    ridx = np.asarray(win.read_index, dtype=np.int64)
    ho = np.asarray(rb.hdr_off).tolist()
    raw = [rb.hdr[ho[i]:ho[i + 1]] for i in ridx.tolist()]
    # what `poa` prints for the header line; without a blank or tab in the batch's header lines that is the line plus
    # " untitled" for every read, and two lines are equal exactly when the raw lines are
    plain = not any(ws in rb.hdr for ws in (b" ", b"\t", b"\r", b"\v", b"\f"))
    hdr = None if plain else [_poa_header(h) for h in raw]
    key = raw if plain else hdr
    n_r = win.n_reads
    # Donatello concatenates consecutive windows with the same header inside one slot file
    # (Donatello.cpp:61-84); the slot is the read's position in its reference batch // 51
    # (Master_Splitter.cpp:366-369,435).  win.read_index = position in this processing batch.
    if n_r > 1:
        k = first_index + ridx
        same_slot = (k[1:] // READS_PER_BATCH == k[:-1] // READS_PER_BATCH) & \
                    ((k[1:] % READS_PER_BATCH) // READS_PER_SLOT == (k[:-1] % READS_PER_BATCH) // READS_PER_SLOT)
        same_hdr = np.fromiter((key[r] == key[r - 1] for r in range(1, n_r)), dtype=bool, count=n_r - 1)
        groups = np.concatenate([[0], np.nonzero(~(same_slot & same_hdr))[0] + 1, [n_r]]).astype(np.int64)
    else:
        groups = np.asarray([0, n_r], dtype=np.int64) if n_r else np.zeros(1, dtype=np.int64)
    b.piece_first = np.asarray(win.read_first, dtype=np.int64)[groups]
    if plain:
        # (">name untitled")[:-11] + " " = the raw line less its last two bytes; a line of one byte is kept whole
        b.rec_hdr = [(raw[g][:-2] if len(raw[g]) >= 2 else raw[g] + b" untitled") + b" " for g in groups[:-1].tolist()]
    else:
        b.rec_hdr = [_donatello_header(hdr[g]) for g in groups[:-1].tolist()]
    # reads for the statistics: runs of records with one header line (computeStats.py:45-56)
    rh = b.rec_hdr
    n_rec = len(rh)
    if n_rec > 1:
        diff = np.fromiter((rh[p] != rh[p - 1] for p in range(1, n_rec)), dtype=bool, count=n_rec - 1)
        b.read_first = np.concatenate([[0], np.nonzero(diff)[0] + 1, [n_rec]]).astype(np.int64)
    else:
        b.read_first = np.asarray([0, n_rec], dtype=np.int64) if n_rec else np.zeros(1, dtype=np.int64)
    b.last = False
    b.plain = plain
    _tick("record / read boundaries (host)", t0)
    return b


class _Buffers:
    """Device buffers of one engine context, grown on demand."""

    def __init__(self, dev):
        self.dev, self.cap_bases, self.cap_win = dev, 0, 0
        self.bases = self.cols = self.ncol = self.status = self.held = None

    def fit(self, total, n, own_bases=True):
        import torch
        if total > self.cap_bases:
            self.cap_bases = int(total * 1.25) + 4096
            self.bases = torch.empty(self.cap_bases, dtype=torch.uint8, device=self.dev) if own_bases else None
            self.cols = torch.empty(3 * self.cap_bases + 64, dtype=torch.uint8, device=self.dev)
        elif own_bases and self.bases is None:
            self.bases = torch.empty(self.cap_bases, dtype=torch.uint8, device=self.dev)
        if n > self.cap_win:
            self.cap_win = int(n * 1.25) + 1024
            self.ncol = torch.empty(self.cap_win, dtype=torch.int32, device=self.dev)
            self.status = torch.empty(self.cap_win, dtype=torch.int32, device=self.dev)


def _shard(reference, uncorrected, corrected, world, rank=0, dist=None, device=None):
    """Contiguous ranges of kept records per rank, cut at read boundaries and balanced by the DP cells the
    reads will cost (elector_amd.distributed.read_cell_estimate).  ONE pass over the three files, in native code
    (elector_reads_scan), on rank 0; the world + 1 bounds are broadcast."""
    import torch
    from .distributed import read_cell_estimate, shard_bounds
    bounds = None
    if rank == 0:
        lr, lu, lc, fresh = split.scan_reads(reference, uncorrected, corrected)
        n = len(lr)
        starts = np.concatenate([np.nonzero(fresh)[0], [n]]).astype(np.int64) if n else np.zeros(1, dtype=np.int64)
        wr = np.add.reduceat(read_cell_estimate(lr, lc, lu), starts[:-1]) if n else np.zeros(0)
        bounds = starts[shard_bounds(wr, world)]
    if dist is None or world == 1:
        return [int(x) for x in bounds]
    t = torch.zeros(world + 1, dtype=torch.int64, device=device)
    if rank == 0:
        t.copy_(torch.from_numpy(np.ascontiguousarray(bounds, dtype=np.int64)))
    dist.broadcast(t, src=0)
    return [int(x) for x in t.cpu().tolist()]


def _append_file(out, path):
    """the bytes of `path` appended to the open file `out` inside the kernel (sendfile) where the platform offers it.
    `out` must NOT be open in append mode (Linux refuses sendfile onto an O_APPEND descriptor with EINVAL): it is
    positioned at its end here and written from there."""
    import errno
    out.flush()
    size = os.path.getsize(path)
    fd = out.fileno()
    os.lseek(fd, 0, os.SEEK_END)
    with open(path, "rb") as f:
        done = 0
        try:
            while done < size:
                k = os.sendfile(fd, f.fileno(), done, min(size - done, 1 << 30))
                if k <= 0:
                    break
                done += k
        except OSError as e:
            if e.errno not in (errno.EINVAL, errno.ENOSYS):      # "not offered here": fall back; anything else is an I/O error
                raise
        if done < size:
            f.seek(done)
            while True:
                chunk = f.read(1 << 24)
                if not chunk:
                    break
                out.write(chunk)
            out.flush()


def getPOA(corrected, reference, uncorrected, threads, outDir, SIZE_CORRECTED_READ_THRESHOLD, soft=None,
           engine=None, matrix=None, parity="raise", write_msa=None):
    """elector/alignment.py:67-131.

    write_msa=False (or ELECTOR_NO_MSA=1 when the argument is left None): msa[_soft].fa is NOT written.  The per-piece
    counters, computed on the device while the MSAs are there, are all call site #2 needs
    (computeStats.outputRecallPrecision takes them from this process's MSA_CACHE: same tuple, stdout, log and side
    files), and the 3 x 3.3 bytes per base of records are the largest item of a run ("keep a flag to
    still write it" -- the default).  Needs header lines without a title and no soft clips, which both make the
    reference's statistics depend on the text of the file.

    parity="raise" (default): a window the device cannot align (ELECTOR_W_*: an empty sequence, a
    sequence beyond ELECTOR_MAX_SEQ) raises.  parity="skip": its record is left out of msa.fa, as a
    `poa` process that dies on such a window leaves its slot's reads out in the reference
    (alignment.py:62); the number of records left out is printed."""
    import torch
    amount_read = 1000 * 10
    print("- Means that a large amount of reads has been handled: " + str(amount_read))
    if write_msa is None:
        write_msa = os.environ.get("ELECTOR_NO_MSA", "0") in ("", "0")
    if engine is not None:
        pool = engine if isinstance(engine, EnginePool) else None
        engines = pool.engines if pool else [engine]
    else:
        engines = _get_pool(matrix).engines
    if soft is not None:
        mergeOut = outDir + "/msa_" + soft + ".fa"
    else:
        mergeOut = outDir + "/msa.fa"
    rank, world = 0, 1
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
    except ImportError:
        dist = None
    dev = torch.device("cuda", engines[0].device)
    cdev = dev if world > 1 and dist.get_backend() == "nccl" else torch.device("cpu")     # where collectives live
    start, stop = 0, None
    if world > 1:
        bounds = _shard(reference, uncorrected, corrected, world, rank, dist, cdev)
        start, stop = bounds[rank], bounds[rank + 1]
    # a rank's part of the output is a file of its own, written from scratch (a part left behind by an aborted run
    # must not be appended to); rank 0 appends the parts to msa.fa in rank order, as Donatello appends
    part_path = mergeOut if world == 1 else mergeOut + ".part%d" % rank

    n_split = max(1, int(os.environ.get("ELECTOR_SPLITTERS", "2")))
    splitters = [_get_splitter(engines[0].device, k) for k in range(n_split)]
    bufs = [_Buffers(dev) for _ in engines]
    work = queue.Queue(maxsize=2 * n_split)
    failure = []

    # Threads in front of the main thread: one cuts the files into batches (native code), two split them, each with a
    # splitter context of its own -- the host's share of a split (reads into pinned staging, waiting for the kernel,
    # window offsets back, record boundaries) is as long as the kernel, and two batches in flight hide it.  The
    # batches carry sequence numbers; the main thread takes them in order.  The reader hands out views of its
    # split.READ_SETS buffer sets in turn, so batch i may only be read when batch i - READ_SETS has been split (with two
    # sets a splitter waited a whole parse for its next batch: the parser could not start it before that splitter's
    # previous batch was through).
    parsed = queue.Queue(maxsize=2)
    split_done = {}                                # sequence number -> Event: the batch's reads are no longer needed
    stop_all = threading.Event()                   # set when the main thread gives up: the threads wind down

    def put(q, item):
        """q.put that gives up when the run is being abandoned"""
        while not stop_all.is_set():
            try:
                q.put(item, timeout=0.2)
                return True
            except queue.Full:
                pass
        return False

    def wait_for(ev):
        while not stop_all.is_set():
            if ev.wait(timeout=0.2):
                return True
        return False

    readers_left = [n_split]
    readers_lock = threading.Lock()
    splitter_done = threading.Event()

    def parser():
        import time
        try:
            rf = split.ReadsFile(reference, uncorrected, corrected, device=engines[0].device)
            try:
                seq = 0
                while not stop_all.is_set():
                    if seq >= split.READ_SETS and not wait_for(split_done[seq - split.READ_SETS]):
                        break
                    t0 = time.perf_counter()
                    rb = rf.next(READS_PER_BATCH, start, stop)
                    _tick("parse FASTA (parser thread, native)", t0)
                    if rb is None:
                        break
                    split_done[seq] = threading.Event()
                    if not put(parsed, (seq, rb)):
                        break
                    seq += 1
                for _ in range(n_split):
                    put(parsed, None)
                # the last batches may still be in use: the file stays open until the splitters are through
                splitter_done.wait()
            finally:
                rf.close()
        except BaseException as e:           # noqa: BLE001 -- handed to the main thread
            failure.append(e)
            for _ in range(n_split):
                put(parsed, None)

    def reader(k):
        try:
            while not stop_all.is_set():
                try:
                    item = parsed.get(timeout=0.2)
                except queue.Empty:
                    continue
                if item is None or failure:
                    break
                seq, rb = item
                b = _prepare(rb, SIZE_CORRECTED_READ_THRESHOLD, threads, splitters[k])
                split_done[seq].set()
                if not put(work, (seq, b)):
                    break
        except BaseException as e:           # noqa: BLE001 -- handed to the main thread
            failure.append(e)
            stop_all.set()
        with readers_lock:
            readers_left[0] -= 1
            if readers_left[0] == 0:
                splitter_done.set()
        put(work, None)

    tp = threading.Thread(target=parser, daemon=True)
    tp.start()
    ths = [threading.Thread(target=reader, args=(k,), daemon=True) for k in range(n_split)]
    for t_ in ths:
        t_.start()

    small_reads = wrongly_cor_reads = 0
    skipped = 0
    # records already in the output file (append semantics, Donatello.cpp:48) are not this run's: call site #2
    # then parses the whole file instead of taking this run's device counters
    existed = os.path.exists(mergeOut) and os.path.getsize(mergeOut) > 0
    if existed and not write_msa:
        raise ValueError("getPOA(write_msa=False): %s already holds records; the counters of a run cannot stand for a "
                         "file it only appended to" % mergeOut)
    all_hdr, all_cols, all_counters, all_read_first = [], [], [], [0]
    last_rows = last_mask = None
    cache_ok = not existed
    # A batch's second half -- wait for its kernels, merged rows to the host, the records into the file -- belongs
    # to a writer thread: the main thread only classifies and enqueues, so the next batch's kernels are queued while
    # the previous batch's 265 MB of records are still on their way to the page cache.  One writer, jobs in batch
    # order: the file is written in order.  A context takes its next batch when its previous job is through
    # (its output buffers and its statistics slot are busy until then).
    jobs = queue.Queue()
    job_done = [None] * len(engines)                # per context: Event of its job in flight
    write_at = [0]                                   # end of the output file (the descriptor is not in append mode)

    write_turn = [0]                                 # job number whose place in the file is assigned next
    write_cv = threading.Condition()

    def finish(fd, jno, e, b, npieces, last_cap):
        nonlocal skipped, last_rows, last_mask, cache_ok
        import time
        eng = engines[e]
        t0 = time.perf_counter()
        counters, piece_cols, lrows, lmask = eng.msa_stats_collect(npieces, last_cap)
        status = bufs[e].status[: b.win.n_windows].cpu().numpy()
        _tick("wait for the GPU (writer threads)", t0)
        drop = np.zeros(npieces, dtype=bool)
        bad = np.nonzero(status)[0] if status.any() else None
        if bad is not None and parity == "skip":
            drop[np.unique(np.searchsorted(b.piece_first, bad, side="right") - 1)] = True
        # the records' place in the file and the run's bookkeeping, in job order; the bytes themselves -- rows to the
        # host, formatting, the copy into the page cache -- need no order and overlap with the next job's
        hl = np.fromiter((len(h) for h in b.rec_hdr), dtype=np.int64, count=npieces)
        nbytes = int((3 * (hl + piece_cols + 2))[~drop].sum())
        with write_cv:
            while write_turn[0] != jno and not stop_all.is_set():
                write_cv.wait(timeout=0.2)
            try:
                if bad is not None and parity != "skip":
                    raise ElectorError(-7, "window %d of the batch starting at read %d: status %d"
                                       % (int(bad[0]), b.first_index, int(status[bad[0]])))
                at = write_at[0]
                write_at[0] += nbytes
                if drop.any():
                    skipped += int(drop.sum())
                    cache_ok = False               # the counters of a read's other pieces saw the dropped one
                if not write_msa and not b.plain:
                    raise ValueError("getPOA(write_msa=False): header lines with a title make the reference's statistics "
                                     "depend on the text of msa.fa (computeStats.py:52,548); write the file")
                all_hdr.extend(b.rec_hdr)
                all_cols.append(piece_cols)
                all_counters.append(counters)
                base = all_read_first[-1]
                all_read_first.extend((base + b.read_first[1:]).tolist())
                if lrows is not None and npieces:
                    nl = int(piece_cols[int(b.read_first[-2]):].sum())          # columns of the batch's last read
                    last_rows, last_mask = lrows[:3 * nl].copy(), lmask[:nl].copy()
                with open(outDir + "/small_reads.txt", "w") as f:
                    f.write(str(b.small) + "\n")
                with open(outDir + "/wrongly_cor_reads.txt", "w") as f:
                    f.write(str(b.wrong) + "\n")
            finally:
                write_turn[0] = jno + 1
                write_cv.notify_all()
        t0 = time.perf_counter()
        if write_msa:
            # merged rows -> pinned host memory -> Donatello's records -> the file, in the library
            got = split.msa_records_pwrite(eng, piece_cols, b.rec_hdr, drop if drop.any() else None, fd, at,
                                           nthreads=max(1, min(16, int(threads))))
            if got != nbytes:
                raise ElectorError(-1, "msa.fa: %d bytes written where %d were reserved" % (got, nbytes))
            _tick("merged rows D2H + write msa.fa (writer threads, native)", t0)
        sys.stdout.write('-' * 200)
        sys.stdout.flush()

    def writer(fd):
        while True:
            job = jobs.get()
            if job is None:
                return
            done = job[-1]
            try:
                if not failure:
                    finish(fd, *job[:-1])
                else:
                    with write_cv:                # keep the turn moving for the jobs behind
                        write_turn[0] = max(write_turn[0], job[0] + 1)
                        write_cv.notify_all()
            except BaseException as e:       # noqa: BLE001 -- handed to the main thread
                failure.append(e)
                stop_all.set()
            finally:
                done.set()

    # Donatello appends (Donatello.cpp:48); a rank's part of a multi-rank run starts from scratch
    fd = None
    if write_msa:
        fd = os.open(part_path, os.O_WRONLY | os.O_CREAT | (os.O_TRUNC if world > 1 else 0), 0o666)
        write_at[0] = os.fstat(fd).st_size
    n_writers = max(1, min(len(engines), int(os.environ.get("ELECTOR_WRITERS", str(len(engines))))))
    tws = [threading.Thread(target=writer, args=(fd,), daemon=True) for _ in range(n_writers)]
    for t_ in tws:
        t_.start()
    n_jobs = 0
    turn = 0
    ready, next_seq, ended = {}, 0, 0

    def batches_in_order():
        """the split batches as the files hold them, whichever splitter thread finished first"""
        nonlocal next_seq, ended
        while True:
            while next_seq in ready:
                yield ready.pop(next_seq)
                next_seq += 1
            if ended == n_split:
                if ready:
                    raise RuntimeError("getPOA: a batch went missing between the splitter threads")
                return
            try:
                item = work.get(timeout=0.2)
            except queue.Empty:
                if failure:
                    raise failure[0]
                continue
            if failure:
                raise failure[0]
            if item is None:
                ended += 1
            else:
                ready[item[0]] = item[1]

    try:
        for b in batches_in_order():
            if failure:
                raise failure[0]
            small_reads += b.small
            wrongly_cor_reads += b.wrong
            win = b.win
            if win.n_windows == 0:
                continue
            e = turn % len(engines)
            turn += 1
            # one job per context at a time: its buffers and its statistics slot are busy until the writer is through
            if job_done[e] is not None:
                import time
                t0 = time.perf_counter()
                job_done[e].wait()
                _tick("wait for the writer (main thread)", t0)
                if failure:
                    raise failure[0]
            import time
            buf = bufs[e]
            total = int(win.off[-1])
            t0 = time.perf_counter()
            buf.fit(total, win.n_windows, own_bases=b.d_bases is None)
            if b.d_bases is None:
                buf.bases[:total].copy_(torch.from_numpy(win.bases), non_blocking=False)
                _tick("windows H2D", t0)
            bases = buf.bases if b.d_bases is None else b.d_bases
            buf.held = (bases, b.d_off)            # the batch's own tensors live until the context's next batch
            t0 = time.perf_counter()
            if b.d_off is not None:
                engines[e].align_device_offsets(bases, b.d_off, win.n_windows, total, buf.cols, buf.ncol, buf.status)
            else:
                engines[e].align_device(bases, win.off, buf.cols, buf.ncol, buf.status)
            npieces = engines[e].msa_stats_enqueue(win.n_windows, buf.cols, buf.ncol, buf.status, b.piece_first,
                                                   b.read_first)
            _tick("classify + enqueue kernels (host)", t0)
            # rows + mask of the batch's last read, for the homopolymer ratio of the run's last read
            p0 = int(b.read_first[-2])
            w0 = int(b.piece_first[p0])
            last_cap = int(win.off[-1] - win.off[3 * w0]) + 16
            job_done[e] = threading.Event()
            jobs.put((n_jobs, e, b, npieces, last_cap, job_done[e]))
            n_jobs += 1
    finally:
        # whatever happened, the threads end and give their buffers, files and device memory back
        for _ in tws:
            jobs.put(None)
        for t_ in tws:
            t_.join()
        if fd is not None:
            os.close(fd)
        stop_all.set()
        splitter_done.set()
        for t_ in ths:
            t_.join()
        tp.join()
        for g in engines:
            try:
                g.sync()
            except ElectorError:
                pass
    if failure:
        raise failure[0]
    if skipped:
        print("\n- " + str(skipped) + " records left out of " + mergeOut + " (windows the device refused, parity=\"skip\")")

    counters = np.concatenate(all_counters) if all_counters else np.zeros((0, ES_NCOUNTERS), dtype=np.int64)
    piece_cols = np.concatenate(all_cols) if all_cols else np.zeros(0, dtype=np.int64)
    if world > 1:
        # one gather of integer counters to rank 0; msa.fa = what it held + the ranks' parts in rank order
        from .distributed import gather_rows
        meta = np.concatenate([counters, piece_cols[:, None]], axis=1) if len(piece_cols) else \
            np.zeros((0, ES_NCOUNTERS + 1), dtype=np.int64)
        meta = gather_rows(meta)
        tot = torch.tensor([small_reads, wrongly_cor_reads], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot)
        small_reads, wrongly_cor_reads = int(tot[0]), int(tot[1])
        mine = (all_hdr, all_read_first, cache_ok,
                None if last_rows is None else (last_rows.tobytes(), last_mask.tobytes()))
        hdr_all = [None] * world if rank == 0 else None
        dist.gather_object(mine, hdr_all, dst=0)          # every part is on disk before rank 0 goes on
        if rank == 0:
            if write_msa:
                # (not "ab": sendfile refuses an O_APPEND descriptor; _append_file positions the file at its end itself)
                with open(mergeOut, "r+b" if os.path.exists(mergeOut) else "wb") as out:
                    for r in range(world):
                        part = mergeOut + ".part%d" % r
                        _append_file(out, part)
                        os.remove(part)
            counters, piece_cols = meta[:, :ES_NCOUNTERS], meta[:, ES_NCOUNTERS]
            all_hdr, all_read_first = [], [0]
            for hs, rf, ok, lr in hdr_all:
                base = all_read_first[-1]
                all_hdr.extend(hs)
                all_read_first.extend(base + x for x in rf[1:])
                cache_ok = cache_ok and ok
                if lr is not None:
                    last_rows = np.frombuffer(lr[0], dtype=np.uint8)
                    last_mask = np.frombuffer(lr[1], dtype=np.uint8)
        dist.barrier()
        if rank != 0:
            return small_reads, wrongly_cor_reads

    MSA_CACHE.pop(os.path.abspath(mergeOut), None)
    if not write_msa and not (cache_ok and len(piece_cols)):
        raise ValueError("getPOA(write_msa=False): this run's counters cannot stand for the file (records left out or no "
                         "record at all); write the file")
    if cache_ok and len(piece_cols):
        st = os.stat(mergeOut) if write_msa else None
        p0 = all_read_first[-2]
        plain = all(b" " not in h[1:-1] and b"\t" not in h for h in all_hdr)
        MSA_CACHE[os.path.abspath(mergeOut)] = dict(
            sig=(st.st_size, st.st_mtime_ns) if st else None, no_file=not write_msa, headers=all_hdr, plain_headers=plain, cols=piece_cols,
            read_first=np.asarray(all_read_first, dtype=np.int64), counters=counters,
            last_first_piece=p0, last_rows=last_rows, last_mask=last_mask)
    return small_reads, wrongly_cor_reads
