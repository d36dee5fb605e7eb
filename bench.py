#!/usr/bin/env python3
"""bench.py -- triplet-MSA throughput of the HIP hot path on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W [--profile P] [--reads R] [--serial]

N > 1: one rank per GPU over RCCL.  The driver launches the ranks with torch.distributed.run; when
WORLD_SIZE is not set and N > 1 this script starts them itself (fresh child processes, before
anything in this process touches a GPU) and relays rank 0's JSON line.  A WORLD_SIZE that differs
from --gpus is an error (exit 2).

A *step* is one pass of the hot path (symbolize -> alignment #1 -> fusion -> alignment #2 ->
fusion + MSA columns -> merge of each piece's windows -> per-piece integer counters back on the
host) over one batch of window triples that is already resident in HBM.  The batch is what
ELECTOR's own batch protocol hands to its POA engine: the windows of `--reads` synthetic long
reads (default 10,001, profile = BASELINE.json configs[1]: E. coli 30X SimLord-like PacBio reads,
15 % error, LoRDEC-like 1 % corrected), cut by this repository's reference-compatible splitter
on the host before the timed region.  Weak scaling: every rank processes its own shard of reads
(independent triples, no data-path collective); rank 0 gathers the per-piece integer counters
over RCCL once per step.  Several engine contexts per GPU take the steps in turn, so several
batches are in flight (the serial head and tail of one batch overlap the alignment kernels of
the others).

Prints ONE JSON line (rank 0).  `value` = reference-read bases of all ranks' triples per second
of the slowest rank.  `roofline` prices the dominant kernel against HBM peak using the
algorithmic bytes of DESIGN.md and that kernel's UN-OVERLAPPED launch durations: HIP events on
the launch stream during a serial pass (one context, one launch chain, every kernel alone on
the chip) that follows the timed region -- with `--serial` the timed region itself runs that
way, which is the command profiles/*_serial_kernel_stats.csv was taken from.  `roofline_valu`
is the ceiling that actually binds (VALU issue); `cpu_baseline` times the reference poaV2 binary
(oracle/_ref/poa, when it travelled with the snapshot) or the C oracle port on this host's cores
over a bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# hardware queues for the contexts' launch chains (see elector_amd/__init__.py); must precede the first HIP call
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# VALU issue peaks, G wave64-instructions per second for the chip: 256 CUs x 4 SIMDs x 2.4 GHz.  A SIMD takes one
# plain two-operand 32-bit instruction (v_add_u32 ...) every 2 cycles when two or more waves share it, but the
# instructions k_poa is made of -- packed 16-bit (v_pk_max_i16, v_pk_sub_i16, v_pk_mad_i16, v_pk_min_u16),
# three-operand (v_bfi_b32, v_lshl_or_b32) and DPP moves -- one every 4 cycles, however many waves there are
# (tests/micro/valu_rate.hip on this GPU, profiles/r02_valu_rate.txt: 950 G/s for v_add_u32, 565-590 G/s for each
# of the others at 4-8 waves per SIMD).  The kernel is priced against the 4-cycle peak.
VALU_PEAK_GINSTS_SIMPLE = 256 * 4 * 2.4 / 2
VALU_PEAK_GINSTS = 256 * 4 * 2.4 / 4
WORKLOADS = {   # BASELINE.json configs restated as synthetic profiles (elector_amd/synthetic.py)
    "ecoli10x_c1": "E. coli ~10X example restated (configs[0]): 459 reads of ~9.5 kb, uncorrected 10.3% err (1:1:1), corrected 0.6%, "
                   "8% trimmed / split, a few extended and a few stubs (run it with --reads 459; the whole real reference chain "
                   "on these reads is pinned in tests/golden/c1_chain.json)",
    "ecoli30x_simlord_lordec": "E. coli 30X SimLord-like PacBio (15% err), LoRDEC-like corrected (1% err), ~8 kb reads",
    "yeast50x_nanosim_consent": "S. cerevisiae 50X NanoSim-like ONT (12% err), CONSENT-like corrected (2% err), ~8 kb reads, whole corrected reads",
    "yeast50x_nanosim_consent_split": "S. cerevisiae 50X NanoSim-like ONT (12% err), CONSENT-like corrected (2% err) with -split: 33% of the reads in 2-3 pieces, 10% trimmed, ~8 kb reads",
    "celegans30x_simlord_mixed": "C. elegans 30X SimLord-like PacBio (15% err), corrected 1.5% err, mixed: 30% trimmed, 25% split, 5% extended, ~8 kb reads",
    "chr1_20x_ont_50kb": "Human chr1 20X NanoSim-like ONT (12% err), corrected (2% err), log-normal read lengths, mean 50 kb",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("ELECTOR_BENCH_READS", "10001")),
                    help="synthetic long reads per rank and step (10,001 = one batch of ELECTOR's own protocol, "
                         "elector/alignment.py:82, Master_Splitter.cpp:397-399)")
    ap.add_argument("--profile", default="ecoli30x_simlord_lordec", choices=sorted(WORKLOADS))
    ap.add_argument("--serial", action="store_true",
                    help="one engine context, one launch chain: every kernel runs alone on the chip (per-kernel times add up to the step)")
    ap.add_argument("--serial-steps", type=int, default=10, help="steps of the serial pass behind the timed region")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="N > 1: steps whose counters travel to rank 0 in one RCCL gather")
    ap.add_argument("--no-rows-to-host", action="store_true",
                    help="skip the second timed loop (the same steps plus the merged MSA rows copied to pinned host memory)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline leg")
    ap.add_argument("--end-to-end", action="store_true",
                    help="three FASTA files -> getPOA -> outputRecallPrecision with a stage table (see bench_e2e.py)")
    if "--end-to-end" in sys.argv[1:]:              # bench_e2e.py has options of its own (--reference-sample, --no-reference)
        return ap.parse_known_args()[0]
    return ap.parse_args()


def self_launch(args):
    """--gpus N without a launcher: start the N ranks as fresh processes (this process has not
    touched a GPU and never will) and pass rank 0's line through.  --standalone lets the launcher pick and hold a
    free rendezvous port itself (a port chosen here could be taken between the choice and its use)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def compare_with_reference(path, lo, hi, off, cols, ncol):
    """The file one reference `poa` process wrote for the windows [lo, hi) (lpo_format.c:398-426: per window
    `>name title` + row, three times, reference / corrected / uncorrected) against the device's column-interleaved
    MSA of the same windows.  -> number of windows whose header lines, row lengths or row bytes differ."""
    import numpy as np
    try:
        data = np.fromfile(path, dtype=np.uint8)
    except OSError:
        return hi - lo
    nl = np.flatnonzero(data == 10)
    nw = hi - lo
    if len(nl) != 6 * nw:
        return nw
    starts = np.concatenate([[0], nl[:-1] + 1]).reshape(nw, 6)
    ends = nl.reshape(nw, 6)
    nc = ncol[lo:hi].astype(np.int64)
    bad = np.zeros(nw, dtype=bool)
    for r in range(3):
        bad |= (ends[:, 2 * r + 1] - starts[:, 2 * r + 1]) != nc
    # header lines: `>w<index> untitled` three times (fasta_format.c:33-37)
    hdr = b"".join(b">w%d untitled" % w for w in range(lo, hi))
    hlen = np.fromiter((len(b">w%d untitled" % w) for w in range(lo, hi)), dtype=np.int64, count=nw)
    hoff = np.cumsum(hlen) - hlen
    hb = np.frombuffer(hdr, dtype=np.uint8)
    for r in range(3):
        bad |= (ends[:, 2 * r] - starts[:, 2 * r]) != hlen
    ok = np.flatnonzero(~bad)
    if len(ok):
        n_ok = nc[ok]
        tot = int(n_ok.sum())
        first = np.cumsum(n_ok) - n_ok
        c = np.arange(tot, dtype=np.int64) - np.repeat(first, n_ok)          # column within its window
        wrep = np.repeat(np.arange(len(ok)), n_ok)
        dev_at = np.repeat(3 * off[3 * (lo + ok)], n_ok) + 3 * c
        differs = np.zeros(len(ok), dtype=np.int64)
        for r in range(3):
            ref_at = np.repeat(starts[ok, 2 * r + 1], n_ok) + c
            np.add.at(differs, wrep[data[ref_at] != cols[dev_at + r]], 1)
        hl = hlen[ok]
        htot = int(hl.sum())
        hfirst = np.cumsum(hl) - hl
        hc = np.arange(htot, dtype=np.int64) - np.repeat(hfirst, hl)
        hrep = np.repeat(np.arange(len(ok)), hl)
        want = hb[np.repeat(hoff[ok], hl) + hc]
        for r in range(3):
            np.add.at(differs, hrep[data[np.repeat(starts[ok, 2 * r], hl) + hc] != want], 1)
        bad[ok[differs > 0]] = True
    return int(bad.sum())


def cpu_baseline(windows, ref_bases_per_window, seconds, device_msa=None):
    """Reference poaV2 (or the oracle port) on this host's cores over a bounded
    sample of the same window stream.  Test infrastructure: uses oracle/.

    device_msa = (cols uint8, ncol): the column-interleaved MSA the timed kernels left in HBM for this very batch,
    brought to the host.  Every window the reference binary aligned is then compared with it byte by byte
    -> second return value {"windows": compared, "differing": d} (None without the reference binary)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    ncores = os.cpu_count() or 1
    off = windows.off
    nwin = windows.n_windows
    ref_poa = os.path.join(oracle_lib.REF_DIR, "poa")
    if os.path.exists(ref_poa):
        # ~0.3 Mbases/s/core for the reference binary -> sample size
        target_bases = 0.3e6 * ncores * seconds
        cum = np.cumsum(ref_bases_per_window)
        ns = int(min(nwin, max(ncores, np.searchsorted(cum, target_bases) + 1)))
        b = windows.bases.tobytes()
        with tempfile.TemporaryDirectory() as d:
            mat = oracle_lib.write_matrix(os.path.join(d, "params.mat"))
            per = (ns + ncores - 1) // ncores
            cmds = []
            for p in range(ncores):
                lo, hi = p * per, min(ns, (p + 1) * per)
                if lo >= hi:
                    break
                names = [os.path.join(d, "out%d_%d" % (k, p)) for k in (1, 2, 3)]
                with open(names[0], "wb") as fr, open(names[1], "wb") as fu, open(names[2], "wb") as fc:
                    for w in range(lo, hi):
                        h = b">w%d\n" % w
                        fr.write(h + b[off[3 * w]:off[3 * w + 1]] + b"\n")
                        fc.write(h + b[off[3 * w + 1]:off[3 * w + 2]] + b"\n")
                        fu.write(h + b[off[3 * w + 2]:off[3 * w + 3]] + b"\n")
                # same command line ELECTOR issues (elector/alignment.py:60)
                cmds.append([ref_poa, "-pir", os.path.join(d, "smsa%d" % p), "-preserve_seqorder",
                             "-corrected_reads_fasta", names[2], "-reference_reads_fasta", names[0],
                             "-uncorrected_reads_fasta", names[1], "-preserve_seqorder", "-threads", "1",
                             "-pathMatrix", mat])
            t0 = time.perf_counter()
            procs = [subprocess.Popen(c, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for c in cmds]
            for p in procs:
                p.wait()
            dt = time.perf_counter() - t0
            parity = None
            if device_msa is not None:
                # after the clock: what the reference just wrote against what the timed kernels wrote
                from concurrent.futures import ThreadPoolExecutor
                cols, ncol = device_msa
                spans = [(os.path.join(d, "smsa%d" % p), p * per, min(ns, (p + 1) * per)) for p in range(len(cmds))]
                with ThreadPoolExecutor(max_workers=min(32, ncores)) as ex:
                    bad = list(ex.map(lambda a: compare_with_reference(a[0], a[1], a[2], off, cols, ncol), spans))
                parity = {"windows": ns, "differing": int(sum(bad)),
                          "against": "reference poa (oracle/_ref/poa built from the reference's sources), every window of "
                                     "the %s: header lines and the three rows, byte by byte"
                                     % ("timed batch" if ns == nwin else "sample")}
        nb = float(cum[ns - 1])
        return {"value": round(nb / dt / 1e6, 4), "unit": "Mbases/s", "cores": len(cmds), "kind": "reference",
                "sample": "%d windows (%d reference bases) of the step's window stream, one reference poa "
                          "process per core as elector/alignment.py's Pool does, wall %.2f s" % (ns, int(nb), dt)}, parity
    # port: single-threaded C oracle
    target_bases = 0.4e6 * seconds
    cum = np.cumsum(ref_bases_per_window)
    ns = int(min(nwin, np.searchsorted(cum, target_bases) + 1))
    t0 = time.perf_counter()
    oracle_lib.batch(windows.bases[: off[3 * ns]], off[: 3 * ns + 1])
    dt = time.perf_counter() - t0
    nb = float(cum[ns - 1])
    return {"value": round(nb / dt / 1e6, 4), "unit": "Mbases/s", "cores": 1, "kind": "port",
            "sample": "%d windows (%d reference bases), oracle/poa_oracle.c single thread, wall %.2f s" % (ns, int(nb), dt)}, None


def pmc_file(profile, reads):
    """The committed PMC passes over `bench.py --serial` on this workload (profiles/pmc_traffic_<profile>.json,
    written by tools/_pmc_traffic.py from tools/_r3_pmc.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes,
    gfx950 correction applied, SQ_INSTS_VALU) -> (dict, provenance text); (None, None) when there is none for this
    workload and batch size.  Counters cannot be read from inside the process being measured: the bench line
    REPLAYS them and says so."""
    for name in ("pmc_traffic_%s.json" % profile, "pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                t = json.load(f)
            if int(t["reads_per_gpu"]) != int(reads) or t.get("profile", "ecoli30x_simlord_lordec") != profile:
                continue
            return t, "replayed from profiles/%s (%s; counters of `%s`), not measured in this run" % (
                name, t.get("collected", "round 2"), t.get("command", "rocprofv3 --pmc"))
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def skipped_alignment1(win, lr, lc):
    """Windows whose alignment #1 the device skips (k_trivial, poa_kernels.hip): corrected equals the
    reference, differs from it by exactly one substitution, one inserted or one deleted letter, or is the one-letter
    filler of a stretch the corrected read does not cover.
    Host-side count for the `gcups` split."""
    import numpy as np
    out = np.zeros(len(lr), dtype=bool)
    off = win.off
    same = np.nonzero((lr == lc) & (lr > 0))[0]
    if len(same):
        L = lr[same]
        ends = np.cumsum(L)
        idx = np.repeat(off[3 * same] - (ends - L), L) + np.arange(int(ends[-1]), dtype=np.int64)
        mism = win.bases[idx] != win.bases[idx + np.repeat(L, L)]
        nmis = np.add.reduceat(mism.astype(np.int64), ends - L)
        out[same[nmis <= 1]] = True
    # the splitter's one-letter filler that occurs nowhere in the reference window (graph written directly)
    fill = np.nonzero((lc == 1) & (lr >= 2))[0]
    if len(fill):
        L = lr[fill]
        ends = np.cumsum(L)
        idx = np.repeat(off[3 * fill] - (ends - L), L) + np.arange(int(ends[-1]), dtype=np.int64)
        hit = win.bases[idx] == np.repeat(win.bases[off[3 * fill + 1]], L)
        out[fill[np.add.reduceat(hit.astype(np.int64), ends - L) == 0]] = True
    # one indel: the strings agree up to the first difference fd and, shifted by one, from fd on
    for d in (-1, 1):                                  # lc = lr + d
        sel = np.nonzero((lc == lr + d) & (np.minimum(lr, lc) >= 1))[0]
        if len(sel) == 0:
            continue
        m = np.minimum(lr[sel], lc[sel])
        ends = np.cumsum(m)
        starts = ends - m
        pos = np.arange(int(ends[-1]), dtype=np.int64) - np.repeat(starts, m)
        xi = np.repeat(off[3 * sel], m) + pos
        yi = np.repeat(off[3 * sel + 1], m) + pos
        direct = win.bases[xi] != win.bases[yi]
        shifted = (win.bases[xi + 1] != win.bases[yi]) if d < 0 else (win.bases[xi] != win.bases[yi + 1])
        big = np.int64(1 << 40)
        first_direct = np.minimum.reduceat(np.where(direct, pos, big), starts)
        last_shift = np.maximum.reduceat(np.where(shifted, pos, -1), starts)
        out[sel[last_shift < first_direct]] = True
    return out


def main():
    args = parse()
    if args.end_to_end:
        import bench_e2e
        return bench_e2e.main(args)
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU "
                             "(python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ...)\n"
                             % (args.gpus, world, args.gpus, args.gpus))
        sys.exit(2)
    import numpy as np
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # ELECTOR_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (the ranks
    # then share devices; the driver's runs use RCCL = "nccl", one GPU per rank)
    backend = os.environ.get("ELECTOR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    # ELECTOR_BENCH_FORCE_DIST=1 (under a launcher with one rank): the process group and every collective of the N > 1
    # path run with a world of one -- RCCL's own code paths on a box with a single GPU
    dist_on = world > 1 or (os.environ.get("ELECTOR_BENCH_FORCE_DIST", "0") not in ("", "0") and world_env is not None)
    if dist_on:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from elector_amd import split, synthetic
    from elector_amd.poa import PoaEngine

    # ---- untimed setup: synthetic reads -> windows (host), upload --------
    triples, headers, read_of = synthetic.read_pieces(args.profile, args.reads, seed=1000 + rank)
    piece_bases = int(sum(len(r[0]) for r in triples))
    nthreads = max(1, (os.cpu_count() or 1) // max(1, world))
    win = split.split_reads(triples, 0.1, headers, nthreads=nthreads)
    n_pieces_in = len(triples)
    del triples
    off = win.off
    n = win.n_windows
    lr = off[1::3] - off[0:-1:3]
    lc = off[2::3] - off[1:-1:3]
    lu = off[3::3] - off[2:-1:3]
    dev = torch.device("cuda", local)
    d_bases = torch.from_numpy(win.bases).to(dev)
    # the window offsets are resident in HBM like the bases (the device splitter leaves both there): the timed entry
    # is elector_poa_batch_device_offsets, which does no per-window work on the host.  ELECTOR_BENCH_HOST_OFFSETS=1
    # times the entry that takes them from a host array instead (A/B)
    d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.int64)).to(dev)
    total_bases = int(off[-1])
    host_offsets = os.environ.get("ELECTOR_BENCH_HOST_OFFSETS", "0") not in ("", "0")

    def align(engine, dc, dn, ds):
        if host_offsets:
            engine.align_device(d_bases, off, dc, dn, ds)
        else:
            engine.align_device_offsets(d_bases, d_off, n, total_bases, dc, dn, ds)
    # E engine contexts (four by default) take the steps in turn (several batches in flight: the serial head and tail
    # of one batch -- symbolize / trivial pass / list sort, merge / statistics -- run beside the
    # alignment kernels of the other).  Every context has its own output buffers.
    n_eng = 1 if args.serial else max(1, int(os.environ.get("ELECTOR_BENCH_ENGINES", "4")))
    engines = [PoaEngine(local) for _ in range(n_eng)]
    if args.serial:
        engines[0].option("chains", 1)
    outs = [(torch.empty(3 * int(off[-1]) + 64, dtype=torch.uint8, device=dev),
             torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
            for _ in range(n_eng)]
    eng = engines[0]
    d_cols, d_ncol, d_status = outs[0]
    # one msa.fa record per corrected piece (= per emitted read of the splitter); the pieces of one
    # read are one read again for the statistics (computeStats.py:45-56)
    piece_first = win.read_first
    read_first = synthetic.piece_groups(read_of, win.read_index)
    from elector_amd import distributed as edist
    from elector_amd._capi import ES_NCOUNTERS

    # every step gathers the same number of counter rows per rank (one per piece of the rank's batch): the ranks
    # tell each other once, the steps then need one collective each
    gather_sizes = edist.gather_sizes(len(piece_first) - 1) if dist_on else None
    pending = []
    gathers = []
    held = []
    gather_every = max(1, args.gather_every)
    gather_cap = gather_every * max(gather_sizes) if gather_sizes else 0       # rows of the largest block a gather carries
    gpool = None
    if dist_on:
        from concurrent.futures import ThreadPoolExecutor as _TPE
        # (a new thread's current device is 0 whatever this thread set: the helper binds itself to the rank's GPU, and the
        # pipe is told the device as well)
        gpool = _TPE(max_workers=1, initializer=torch.cuda.set_device, initargs=(local,))
    turn = [0]
    host_s = [0.0]                               # host time of classifying and enqueueing (the GPU work is asynchronous)
    gather_s = [0.0]                             # host time of starting a step's gather and taking in an earlier one

    def collect():
        """per-piece counters of the oldest queued step on the host (rank 0 receives every rank's rows)"""
        e, npieces = pending.pop(0)
        counters, _ = engines[e].msa_stats_collect(npieces)
        if not dist_on:
            return counters
        # Fewer, larger collectives: the counters of `gather_every` steps travel together (every step's counters
        # still reach rank 0 inside the timed region).  An RCCL collective holds compute units while it runs and its
        # small copies queue up behind whole launch chains of the alignment kernels: one gather per step cost 19 % of
        # the step rate in a one-rank rehearsal on RCCL, one per eight steps costs nothing measurable.  The staging,
        # the collective and taking its result in run on a helper thread; up to three are in flight.
        tg = time.perf_counter()
        held.append(counters)
        res = flush_gather(False)
        gather_s[0] += time.perf_counter() - tg
        return res

    def flush_gather(force):
        """-> the last step's rows of the oldest finished gather (rank 0), or None"""
        res = None
        if held and (force or len(held) >= gather_every):
            k = len(held)
            block = np.concatenate(held) if k > 1 else held[0]
            held.clear()
            sizes = [x * k for x in gather_sizes]
            gathers.append((k, gpool.submit(lambda c=block, z=sizes: edist.gather_rows_async(c, z, cap=gather_cap, device=local if backend == "nccl" else None).wait())))
        while gathers and (force or len(gathers) > 3):
            k, fut = gathers.pop(0)
            got = fut.result()
            if got is not None:
                # rows arrive rank by rank, each rank's k steps back to back: the last step of every rank
                parts, at = [], 0
                for x in gather_sizes:
                    parts.append(got[at + (k - 1) * x: at + k * x])
                    at += k * x
                res = np.concatenate(parts) if len(parts) > 1 else parts[0]
        return res

    def step():
        """Queue one step (windows in HBM -> POA kernels -> merge -> counters -> pinned host memory),
        then hand out the counters of the oldest step in flight: the host prepares step i+1 while the
        GPU still works on step i, as a run over many 10,001-read batches would."""
        e = turn[0] % n_eng
        turn[0] += 1
        dc, dn, ds = outs[e]
        th = time.perf_counter()
        align(engines[e], dc, dn, ds)
        pending.append((e, engines[e].msa_stats_enqueue(n, dc, dn, ds, piece_first, read_first)))
        host_s[0] += time.perf_counter() - th
        return collect() if len(pending) > n_eng else None

    # untimed setup, continued: grow every workspace (both halves of the double-buffered upload staging
    # and of the statistics slots) and let the HIP runtime size its queues for overlapped batches -- the
    # first batch that is enqueued while another still runs pays a one-time ~14 ms inside the runtime
    def drain():
        while pending:
            collect()
        flush_gather(True)

    for _ in range(3 * n_eng):
        step()
    drain()
    for _ in range(args.warmup):
        step()
    drain()
    for g in engines:
        g.sync()
        g.timing_enable(args.serial)
        g.timing_reset()

    # ---- timed region ----------------------------------------------------
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    for g in engines:
        g.sync()
    host_s[0] = 0.0
    gather_s[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    while pending:
        counters = collect()                     # every step's counters are on the host before the clock stops
    if dist_on:
        last = flush_gather(True)                # ... and, with several ranks, on rank 0
        counters = last if last is not None else counters
    for g in engines:
        g.sync()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    host_ms_per_step = host_s[0] / args.steps * 1e3
    gather_ms_per_step = gather_s[0] / args.steps * 1e3

    # the MSA the timed kernels wrote (context 0's last timed step), for the every-window comparison with the
    # reference binary behind the clock
    cols_timed = ncol_timed = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cols_timed = outs[0][0].cpu().numpy()
        ncol_timed = outs[0][1].cpu().numpy()
    exit_code = 0
    status = d_status.cpu().numpy()
    ncol = d_ncol.cpu().numpy().astype(np.int64)
    if status.any():
        raise SystemExit("bench: %d windows failed on device" % int((status != 0).sum()))
    po = eng.last_po_sizes(n).astype(np.int64)

    # ---- second timed loop: the same steps, and every step's merged MSA rows (the body of msa.fa, Donatello.cpp:86-91)
    # copied to pinned host memory as well -- SURVEY.md 8(d) words the metric "MSA rows + per-read counters back on
    # host"; `value` keeps the rows in HBM (call site #2 needs the counters only, 8(f2)), this figure brings them over
    # PCIe.  One helper thread per context collects the counters and fetches the rows (the library releases the GIL),
    # so the copies of one context run beside the kernels of the others.
    dt_rows = None
    rows_bytes = 0
    if not args.no_rows_to_host and not args.serial:
        from concurrent.futures import ThreadPoolExecutor
        cap = 3 * int(off[-1]) + 64
        pinned = [torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(n_eng)]
        pool = ThreadPoolExecutor(max_workers=n_eng)
        futs = [None] * n_eng
        fetched = [0]

        def finish_rows(e, npieces):
            c, pc = engines[e].msa_stats_collect(npieces)
            fetched[0] = engines[e].msa_rows_fetch_into(pc, pinned[e].data_ptr(), cap)
            return c

        def step_rows():
            e = turn[0] % n_eng
            turn[0] += 1
            if futs[e] is not None:
                futs[e].result()                     # this context's previous step is on the host, rows included
            dc, dn, ds = outs[e]
            align(engines[e], dc, dn, ds)
            futs[e] = pool.submit(finish_rows, e, engines[e].msa_stats_enqueue(n, dc, dn, ds, piece_first, read_first))

        def drain_rows():
            for e in range(n_eng):
                if futs[e] is not None:
                    futs[e].result()
                    futs[e] = None

        for _ in range(n_eng + 1):
            step_rows()
        drain_rows()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_rows()
        drain_rows()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        dt_rows = time.perf_counter() - t0
        rows_bytes = int(fetched[0])
        pool.shutdown()
        del pinned

    # ---- after the clock: serial pass for the per-kernel roofline -------------------------------------------
    # A context of its own with ONE launch chain from its first step on, after the timed contexts are gone: exactly
    # what `bench.py --serial` times and what profiles/*_serial_kernel_stats.csv (rocprofv3 over that command) shows,
    # so the figures below can be recomputed from the committed profile.  (Up to round 2 this pass reused a timed
    # context switched to one chain: its moves pool kept the two-chain size, twice the slots in rotation and a lower
    # hit rate in L2 / Infinity Cache -- k_poa came out 13 % slower than in the profile.)
    serial_steps = args.steps if args.serial else max(1, args.serial_steps)
    serial_wall = dt / args.steps
    if not args.serial:
        for g in engines[1:]:
            g.close()
        engines[0].close()
        eng = PoaEngine(local)
        engines = [eng]
        eng.option("chains", 1)

        def serial_step():
            align(eng, d_cols, d_ncol, d_status)
            eng.msa_stats_collect(eng.msa_stats_enqueue(n, d_cols, d_ncol, d_status, piece_first, read_first))
        for _ in range(4):                       # grow the workspace, settle the clocks
            serial_step()
        eng.sync()
        eng.timing_enable(True)
        eng.timing_reset()
        ts = time.perf_counter()
        for _ in range(serial_steps):
            serial_step()
        eng.sync()
        serial_wall = (time.perf_counter() - ts) / serial_steps
    skipped = skipped_alignment1(win, lr, lc)
    cells1, cells2 = int((lr * lc).sum()), int((po * lu).sum())
    cells1_computed = int((lr * lc)[~skipped].sum())
    t_dp1, k_dp1 = eng.timing_read(0)
    t_dp2, k_dp2 = eng.timing_read(1)
    t_oth = eng.timing_read(2)[0]
    t_st = eng.timing_read(3)[0]
    t_poa, k_poa = eng.timing_read(4)
    rdev = dev if backend == "nccl" else torch.device("cpu")
    tmax = torch.tensor([dt, dt_rows or 0.0], dtype=torch.float64, device=rdev)
    tot = torch.tensor([piece_bases, n, cells1 + cells2, cells1_computed + cells2], dtype=torch.int64, device=rdev)
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt_max, dt_rows_max = float(tmax[0].item()), float(tmax[1].item())
    # what the process group actually spans: one entry per rank, gathered over it (RCCL when the backend is nccl)
    props = torch.cuda.get_device_properties(local)
    me = {"rank": rank, "local_rank": local, "device": props.name, "uuid": str(getattr(props, "uuid", "")),
          "pci_bus_id": int(getattr(props, "pci_bus_id", -1)), "host": socket.gethostname()}
    ranks = [me]
    if dist_on:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    bases_all, windows_all, cells_all, cells_comp_all = (int(x) for x in tot.tolist())

    if rank == 0:
        value = bases_all * args.steps / dt_max / 1e6
        # roofline of the dominant kernel: algorithmic bytes = 8-bit inputs + 8-bit MSA out + descriptors
        alg_bytes = int((lr + lc + lu).sum() + 3 * ncol.sum() + 28 * n)
        # kernel classes: 4 = k_poa (the whole window in one kernel: both alignments, tracebacks, fusions); 0 / 1 =
        # alignment #1 / #2 stage of the two-kernel path (k_fused_a / k_fused_b; behind k_poa only the windows it
        # handed back); measured un-overlapped (serial pass)
        dom = max((("k_poa", t_poa, k_poa), ("k_fused_b", t_dp2, k_dp2), ("k_fused_a", t_dp1, k_dp1)), key=lambda x: x[1])
        launches = max(1, dom[2])
        avg_ms = dom[1] / launches
        bytes_per_launch = alg_bytes * serial_steps / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        pmc, pmc_prov = pmc_file(args.profile, args.reads)
        traffic = valu = None
        if pmc:
            try:
                traffic = int(pmc["kernels"][dom[0]]["traffic_bytes_per_launch"])
                valu = int(pmc["valu_wave_insts_per_step"])
            except (KeyError, ValueError):
                pass
        step_s = dt_max / args.steps
        out = {
            "metric": "triplet-MSA Mbases/s", "value": round(value, 3), "unit": "Mbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": "%s: %d reads per GPU per step, cut into windows by the ELECTOR splitter rules"
                                   % (WORKLOADS[args.profile], args.reads),
                       "profile": args.profile, "reads_per_gpu": args.reads, "triples_per_gpu": n_pieces_in,
                       "windows_per_gpu": n, "ref_bases_per_gpu": piece_bases,
                       "filler_windows_per_gpu": int((lc == 1).sum()),
                       "parallelism": "shard-by-read x%d" % world,
                       "batches_in_flight_per_gpu": n_eng, "serial": bool(args.serial),
                       # `value`: per-read counters back on the host, the merged MSA rows stay in HBM (SURVEY.md 8(f2):
                       # call site #2 is served from the device counters); `value_rows_to_host` adds the rows
                       "rows_to_host": False},
            "value_rows_to_host": None if not dt_rows_max else round(bases_all * args.steps / dt_rows_max / 1e6, 3),
            "rows_to_host": None if not dt_rows_max else {
                "ms_per_step": round(dt_rows_max / args.steps * 1e3, 3), "bytes_per_step_per_gpu": rows_bytes,
                "pcie_gbs_per_gpu": round(rows_bytes * args.steps / dt_rows_max / 1e9, 2),
                "note": "the same %d steps timed again with every step's merged rows (3 x columns bytes per piece) copied "
                        "to pinned host memory by a helper thread per context" % args.steps},
            "ranks": {"world": world, "backend": backend if dist_on else None,
                      "distinct_devices": len({(r["host"], r["uuid"] or r["pci_bus_id"]) for r in ranks}),
                      "devices": ranks},
            # dtype: k_poa's recurrences run on 16-bit scores, two windows per 32-bit lane (the fall-back kernels, 0.2 %
            # of the windows, on 32-bit ones)
            # DP cells per second.  effective: every cell the reference computes (Lr*Lc + |PO|*Lu per window);
            # computed: without alignment #1 of the windows whose corrected sequence equals the reference or
            # differs by one substitution or one indel, which the device settles without a dynamic program (k_trivial)
            "gcups_effective": round(cells_all * args.steps / dt_max / 1e9, 3),
            "gcups_computed": round(cells_comp_all * args.steps / dt_max / 1e9, 3),
            "alignment1_skipped_windows_frac": round(float(skipped.mean()), 4),
            "kernel_ms_per_step": {"k_poa": round(t_poa / serial_steps, 3),
                                   "alignment1_stage": round(t_dp1 / serial_steps, 3),
                                   "alignment2_stage": round(t_dp2 / serial_steps, 3),
                                   "other": round(t_oth / serial_steps, 3),
                                   "merge_and_counters": round(t_st / serial_steps, 3),
                                   "serial_step_wall": round(serial_wall * 1e3, 3),
                                   "host_classify_and_enqueue": round(host_ms_per_step, 3),
                                   "host_counters_gather": round(gather_ms_per_step, 3),
                                   "note": "HIP-event time per launch, summed per step, from %d un-overlapped steps "
                                           "(one context, one launch chain)%s"
                                           % (serial_steps, "" if args.serial else
                                              " in a fresh context after the timed region, = what `bench.py --serial` times")},
            "roofline": {"bound": "hbm", "kernel": dom[0], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "traffic_provenance": pmc_prov if traffic is not None else None,
                         "launches": int(launches), "avg_launch_ms": round(avg_ms, 4),
                         "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "whole_step_frac": round(alg_bytes / step_s / 1e9 / HBM_PEAK_GBS, 6)},
            # what actually binds: VALU issue + latency.  Per GPU: wave-instructions per
            # step from the committed PMC passes of this command, time measured live
            "roofline_valu": None if valu is None else {
                "bound": "valu-issue", "wave_insts_per_step": valu, "provenance": pmc_prov, "peak": round(VALU_PEAK_GINSTS, 1),
                "peak_two_operand_32bit": round(VALU_PEAK_GINSTS_SIMPLE, 1),
                "peak_note": "packed 16-bit, three-operand and DPP instructions issue once per 4 cycles per SIMD "
                             "(measured: tests/micro/valu_rate.hip, profiles/r02_valu_rate.txt)",
                "unit": "G wave-insts/s per GPU",
                "achieved": round(valu / step_s / 1e9, 1),
                "frac": round(valu / step_s / 1e9 / VALU_PEAK_GINSTS, 4)},
            "pieces_gathered": int(counters.shape[0]),
            "counters_checksum": int(counters[:, :ES_NCOUNTERS - 1].sum()),
        }
        parity = None
        if world == 1 and not args.no_cpu_baseline:
            # the MSA columns of the batch the timed region worked on, as the last timed step of context 0 left them
            out["cpu_baseline"], parity = cpu_baseline(win, lr, args.cpu_seconds, (cols_timed, ncol_timed))
            if parity is not None:
                out["parity_vs_reference"] = parity
        print(json.dumps(out), flush=True)
        if parity is not None and parity["differing"]:
            sys.stderr.write("bench.py: %d of %d windows differ from the reference poa's output\n"
                             % (parity["differing"], parity["windows"]))
            exit_code = 1
    for g in engines:
        g.close()
    if dist_on:
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
