#!/usr/bin/env python3
"""bench.py -- triplet-MSA throughput of the HIP hot path on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W [--profile P] [--reads R] [--serial]

N > 1: one rank per GPU over RCCL.  The driver launches the ranks with torch.distributed.run; when
WORLD_SIZE is not set and N > 1 this script starts them itself (fresh child processes, before
anything in this process touches a GPU) and relays rank 0's JSON line.  A WORLD_SIZE that differs
from --gpus is an error (exit 2).

A *step* is one pass of the hot path (window classification -> symbolize -> alignment #1 -> fusion ->
alignment #2 -> fusion + MSA columns -> merge of each piece's windows -> per-piece integer counters AND
the merged MSA rows back on the host: SURVEY.md 8(d)) over one batch of window triples that is already
resident in HBM, bases and offsets.  The batch is what ELECTOR's own batch protocol hands to its POA
engine: the windows of `--reads` synthetic long reads (default 10,001), cut by this repository's
reference-compatible splitter on the host before the timed region.  The steps rotate over `--batches`
(default 3) differently seeded batches per rank, ~0.9 GB of input: more than the Infinity Cache holds.
The default profile follows BASELINE.json's configs by GPU count: 1 or 2 -> configs[2] (yeast 50X ONT-like,
CONSENT `-split`: the larger of the two 1-GPU configs), 4 -> configs[3] (C. elegans mixed), 8 -> configs[4]
(chr1 50 kb reads).  At N = 1 the line also carries a `configs` array: the other single-GPU profiles of
BASELINE.json through the same code at one batch each, with their own reference-parity sample.
Weak scaling (default): every rank processes its own reads (independent triples, no data-path
collective); rank 0 gathers the per-piece integer counters over RCCL.  `--scaling strong`: ONE read set
(`--strong-units` x `--reads` reads, whatever N) is cut into per-rank ranges by the partitioner of
DESIGN.md section 5 (distributed.shard_bounds over distributed.read_cell_estimate), the line reports the
DP cells per rank and their imbalance.  Several engine contexts per GPU take the steps in turn, so several
batches are in flight.

Prints ONE JSON line (rank 0).  `value` = reference-read bases of all ranks' triples per second
of the slowest rank, rows and counters on the host; `value_rows_in_hbm` = the same steps with the merged
rows left in HBM (call site #2 needs the counters only).  `roofline` prices the dominant kernel against HBM peak using the
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


DEFAULT_PROFILE_BY_GPUS = {1: "yeast50x_nanosim_consent_split", 2: "yeast50x_nanosim_consent_split",
                           4: "celegans30x_simlord_mixed", 8: "chr1_20x_ont_50kb"}
# the single-GPU profiles the `configs` array runs beside the headline one: (profile, reads per batch)
CONFIGS_ARRAY = [("ecoli30x_simlord_lordec", 10001), ("yeast50x_nanosim_consent_split", 10001),
                 ("celegans30x_simlord_mixed", 10001), ("chr1_20x_ont_50kb", 2000)]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("ELECTOR_BENCH_READS", "10001")),
                    help="synthetic long reads per rank and step (10,001 = one batch of ELECTOR's own protocol, "
                         "elector/alignment.py:82, Master_Splitter.cpp:397-399)")
    ap.add_argument("--profile", default=None, choices=sorted(WORKLOADS),
                    help="default by GPU count: 1, 2 -> yeast50x_nanosim_consent_split, 4 -> celegans30x_simlord_mixed, "
                         "8 -> chr1_20x_ont_50kb (BASELINE.json configs[2..4])")
    ap.add_argument("--batches", type=int, default=3, help="differently seeded batches per rank the steps rotate over")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
    ap.add_argument("--strong-units", type=int, default=8, help="--scaling strong: the read set is this many x --reads reads, whatever N")
    ap.add_argument("--no-configs", action="store_true", help="N = 1: skip the `configs` array (the other single-GPU profiles)")
    ap.add_argument("--configs-steps", type=int, default=20, help="timed steps per entry of the `configs` array")
    ap.add_argument("--serial", action="store_true",
                    help="one engine context, one launch chain: every kernel runs alone on the chip (per-kernel times add up to the step)")
    ap.add_argument("--serial-steps", type=int, default=10, help="steps of the serial pass behind the timed region")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="N > 1: steps whose counters travel to rank 0 in one RCCL gather")
    ap.add_argument("--no-rows-in-hbm", "--no-rows-to-host", dest="no_second_loop", action="store_true",
                    help="skip the second timed loop (the same steps with the merged MSA rows left in HBM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline leg")
    ap.add_argument("--bundles", action="store_true",
                    help="a12: time the heaviest-bundle consensus search (k_bundle) of one batch per step, with its roofline")
    ap.add_argument("--end-to-end", action="store_true",
                    help="three FASTA files -> getPOA -> outputRecallPrecision with a stage table (see bench_e2e.py)")
    if "--end-to-end" in sys.argv[1:]:              # bench_e2e.py has options of its own (--reference-sample, --no-reference)
        args = ap.parse_known_args()[0]
    else:
        args = ap.parse_args()
    if args.profile is None:
        args.profile = DEFAULT_PROFILE_BY_GPUS.get(args.gpus, "yeast50x_nanosim_consent_split")
        args.profile_defaulted = True
    else:
        args.profile_defaulted = False
    # (--serial rotates over the same batches as the timed region: the per-kernel tables and PMC passes under profiles/ are
    # averages over the rotation, like `value`; --batches 1 is the single batch of rounds 2-4)
    if args.profile == "chr1_20x_ont_50kb" and "--reads" not in " ".join(sys.argv[1:]) and "ELECTOR_BENCH_READS" not in os.environ:
        args.reads = 2000                               # 50 kb reads: 2,000 of them are a batch of the usual size in bases
    return args


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
        # cores: what the processes could actually run on (affinity mask / cgroup quota of this box), not the host's count
        return {"value": round(nb / dt / 1e6, 4), "unit": "Mbases/s", "cores": min(len(cmds), cpu_share()), "kind": "reference",
                "processes": len(cmds), "host_cores": ncores,
                "sample": "%d windows (%d reference bases) of the step's window stream, one reference poa process per host core "
                          "as elector/alignment.py's Pool does (%d processes on the %d cores this box may use), wall %.2f s"
                          % (ns, int(nb), len(cmds), min(len(cmds), cpu_share()), dt)}, parity
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
    written by tools/pmc_traffic.py from tools/gpu_pmc.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes,
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


def cpu_share():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota (a GPU box hands a
    container 16 cores' worth of a 256-core host)"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    per = int(f.read().split()[0])
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


class Batch(object):
    """one batch of window triples: host side (the splitter's Windows + grouping) and, after upload(), the device side"""
    __slots__ = ("profile", "seed", "win", "n", "total", "lr", "lc", "lu", "piece_first", "read_first", "n_pieces_in",
                 "piece_bases", "n_reads", "d_bases", "d_off", "po", "ncol_sum")


def prepare_batch(job):
    """Worker (a forked process that never touches a GPU): synthetic reads -> windows on the host.
    job = (profile, reads, seed, nthreads, read range or None).  The read range (--scaling strong) keeps the pieces of
    the reads [lo, hi) of the seeded unit only."""
    import numpy as np
    from elector_amd import split, synthetic
    profile, reads, seed, nthreads, rng = job
    triples, headers, read_of = synthetic.read_pieces(profile, reads, seed=seed)
    if rng is not None:
        lo, hi = rng
        keep = [i for i, r in enumerate(read_of) if lo <= r < hi]
        triples = [triples[i] for i in keep]
        headers = [headers[i] for i in keep]
        read_of = [read_of[i] for i in keep]
    b = Batch()
    b.profile, b.seed = profile, seed
    b.n_pieces_in = len(triples)
    b.piece_bases = int(sum(len(r[0]) for r in triples))
    b.n_reads = len(set(read_of))
    b.win = split.split_reads(triples, 0.1, headers, nthreads=nthreads)
    off = b.win.off
    b.n, b.total = b.win.n_windows, int(off[-1])
    b.lr, b.lc, b.lu = off[1::3] - off[0:-1:3], off[2::3] - off[1:-1:3], off[3::3] - off[2:-1:3]
    b.piece_first = b.win.read_first
    b.read_first = synthetic.piece_groups(read_of, b.win.read_index)
    b.d_bases = b.d_off = b.po = None
    b.ncol_sum = 0
    return b


def unit_read_lengths(job):
    """Worker: per read of a seeded unit the lengths (reference, corrected pieces together, uncorrected) -- what the
    partitioner weighs (distributed.read_cell_estimate)"""
    import numpy as np
    from elector_amd import synthetic
    profile, reads, seed = job
    triples, _, read_of = synthetic.read_pieces(profile, reads, seed=seed)
    ids = np.asarray(read_of, dtype=np.int64)
    nr = int(ids.max()) + 1 if len(ids) else 0
    lr, lc, lu = np.zeros(nr, dtype=np.int64), np.zeros(nr, dtype=np.int64), np.zeros(nr, dtype=np.int64)
    for (r, c, u), i in zip(triples, ids):
        lr[i] = len(r)
        lu[i] = len(u)
        lc[i] += len(c)
    return lr, lc, lu


def bundles_mode(args):
    """a12 (heaviest_bundle.c:16-172; optional output, the reference never calls it): per step one batch through the
    alignment kernels that leave the graph in HBM (elector_ctx_keep_graph) and the bundle search on every window.
    Prints ONE JSON line: ms per step of the search (HIP events on its stream), windows per second, its HBM roofline."""
    import numpy as np
    import torch
    from elector_amd.poa import PoaEngine
    profile = args.profile if not args.profile_defaulted else "ecoli30x_simlord_lordec"
    b = prepare_batch((profile, args.reads, 1000, cpu_share(), None))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    d_bases = torch.from_numpy(b.win.bases).to(dev)
    d_off = torch.from_numpy(np.ascontiguousarray(b.win.off, dtype=np.int64)).to(dev)
    d_cols = torch.empty(3 * b.total + 64, dtype=torch.uint8, device=dev)
    d_ncol = torch.empty(b.n, dtype=torch.int32, device=dev)
    d_status = torch.empty(b.n, dtype=torch.int32, device=dev)
    eng = PoaEngine(0)
    eng.keep_graph(True)

    def step():
        eng.align_device_offsets(d_bases, d_off, b.n, b.total, d_cols, d_ncol, d_status)
        eng.bundles_enqueue(b.n)
    for _ in range(max(2, args.warmup)):
        step()
    eng.sync()
    eng.timing_enable(True)
    eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync()
    dt = time.perf_counter() - t0
    ms, launches = eng.timing_read(5)
    ncol = d_ncol.cpu().numpy().astype(np.int64)
    po = eng.last_po_sizes(b.n).astype(np.int64)
    # algorithmic bytes of the search: the graph after fusion #1 as the alignment kernels left it (8 + 2 + 4 bytes per node),
    # the uncorrected symbols, the column counts in; up to three consensus rows and 32 bytes of bundle bookkeeping out
    alg = int(14 * po.sum() + b.lu.sum() + 4 * b.n + 3 * ncol.sum() + 32 * b.n)
    per_launch_ms = ms / max(1, launches)
    eng.timing_enable(False)

    # What the search costs INSIDE the pipeline the headline runs: E contexts take the batch in turn (alignment, merge and
    # counters as bench.py's step, rows left in HBM), first without the search, then with it queued behind every
    # alignment; the difference of the two step times is its price there (the serial figure above hides nothing: one
    # context, the search alone on the chip).
    n_eng = max(1, int(os.environ.get("ELECTOR_BENCH_ENGINES", "4")))
    pool = [eng] + [PoaEngine(0) for _ in range(n_eng - 1)]
    outs = [(d_cols, d_ncol, d_status)] + [(torch.empty_like(d_cols), torch.empty_like(d_ncol), torch.empty_like(d_status))
                                           for _ in range(n_eng - 1)]

    from concurrent.futures import ThreadPoolExecutor
    helpers = ThreadPoolExecutor(max_workers=n_eng, initializer=torch.cuda.set_device, initargs=(0,))

    def pipelined(keep, with_search, threaded=False, deferred=False, now=False):
        """The search waits twice for its context's stream (the scratch size, the class counts); queued inside
        elector_poa_bundles_enqueue (`now`: option "bundles_now", the library's behaviour until round 5) those waits hold up
        the thread that feeds every context.  By default the library only notes the search and queues it at the context's
        next call that waits for it anyway (here elector_msa_stats_collect).  threaded / deferred: what a caller can do
        about it with "bundles_now" -- a helper thread per context, or the call put in front of the collect."""
        for g in pool:
            g.keep_graph(keep)
            g.option("bundles_now", 1 if (now or threaded or deferred) else 0)
        pending = []

        def take():
            pe, np_, fut = pending.pop(0)
            if fut is not None:
                fut.result()
            if with_search and deferred:
                # (deferred: the search of a context's batch is queued when the context comes round again -- its alignment
                # is through by then and the two waits last what the search's own first stages last)
                pool[pe].bundles_enqueue(b.n)
            pool[pe].msa_stats_collect(np_)

        def one(i):
            e = i % n_eng
            if len(pending) >= n_eng:
                take()
            dc, dn, ds = outs[e]
            pool[e].align_device_offsets(d_bases, d_off, b.n, b.total, dc, dn, ds)
            fut = None
            if with_search and not threaded and not deferred:
                pool[e].bundles_enqueue(b.n)
            npieces = pool[e].msa_stats_enqueue(b.n, dc, dn, ds, b.piece_first, b.read_first)
            if with_search and threaded:
                fut = helpers.submit(pool[e].bundles_enqueue, b.n)
            pending.append((e, npieces, fut))

        def drain():
            while pending:
                take()
            for g in pool:
                g.sync()
        for i in range(2 * n_eng + 1):
            one(i)
        drain()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(args.steps):
            one(i)
        drain()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / args.steps * 1e3
    pipe_plain = pipelined(False, False)        # the headline's alignment (k_poa; it leaves no graph behind)
    pipe_without = pipelined(True, False)       # the alignment kernels that keep the graph in HBM, no search
    pipe_with = pipelined(True, True)
    pipe_with_now = pipelined(True, True, now=True)
    pipe_with_thr = pipelined(True, True, threaded=True)
    pipe_with_def = pipelined(True, True, deferred=True)
    helpers.shutdown()
    for g in pool:
        g.option("bundles_now", 0)
    for g in pool[1:]:
        g.close()
    out = {"metric": "heaviest-bundle consensus (a12) ms per step", "value": round(ms / args.steps, 3), "unit": "ms", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "higher_is_better": False, "dtype": "int32", "data": "synthetic",
           "config": {"workload": "%s: %d reads per step" % (WORKLOADS[profile], args.reads), "profile": profile, "windows": b.n},
           "windows_per_s": round(b.n / (per_launch_ms * 1e-3), 1) if per_launch_ms > 0 else None,
           "step_ms_with_alignment": round(dt / args.steps * 1e3, 3),
           "pipelined": {"contexts": n_eng, "step_ms_without_graph": round(pipe_plain, 3), "step_ms_without_search": round(pipe_without, 3), "step_ms_with_search": round(pipe_with, 3), "step_ms_with_search_queued_inside_the_call": round(pipe_with_now, 3),
                         "step_ms_with_search_from_helper_threads": round(pipe_with_thr, 3),
                         "step_ms_with_search_queued_a_turn_later": round(pipe_with_def, 3),
                         "search_ms_inside_pipeline": round(min(pipe_with, pipe_with_now, pipe_with_thr, pipe_with_def) - pipe_without, 3),
                         "note": "alignment + merge + counters per step, rows left in HBM, the contexts taking the batch in turn; "
                                 "without_graph = the headline's kernels (k_poa keeps no graph), without_search = the graph-keeping "
                                 "alignment kernels alone, with_search = elector_poa_bundles_enqueue behind every alignment (the library notes the search and queues it when "
                                 "the context is collected); the other three with option bundles_now"},
           "roofline": {"bound": "hbm", "kernel": "k_bundle_inputs + k_bundle_lds<C, LDS | HBM> + k_bundle_hbm (side by side: the time is the search's, plan to last row)", "achieved": round(alg / (per_launch_ms * 1e-3) / 1e9, 3) if per_launch_ms > 0 else 0.0,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if per_launch_ms > 0 else 0.0,
                        "traffic": None, "launches": int(launches), "avg_launch_ms": round(per_launch_ms, 4),
                        "algorithmic_bytes_per_launch": alg}}
    print(json.dumps(out), flush=True)
    eng.close()


def main():
    args = parse()
    if args.end_to_end:
        import bench_e2e
        return bench_e2e.main(args)
    if args.bundles:
        return bundles_mode(args)
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
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    strong = args.scaling == "strong"
    n_batches = max(1, args.batches)
    with_configs = world == 1 and not strong and not args.no_configs and not args.serial and args.profile_defaulted
    ncores = max(1, cpu_share() // max(1, world))

    # ---- untimed setup, part 1 (before this process touches a GPU): synthetic reads -> windows on the host, the
    # batches side by side in forked workers ----
    jobs = []            # (kind, profile, reads, seed, range)
    if strong:
        units = max(1, args.strong_units)
        if os.environ.get("ELECTOR_BENCH_NO_FORK", "0") in ("", "0"):
            with ProcessPoolExecutor(max_workers=min(units, ncores), mp_context=mp.get_context("fork")) as ex:
                lens = list(ex.map(unit_read_lengths, [(args.profile, args.reads, 1000 + 7919 * u) for u in range(units)]))
        else:
            lens = [unit_read_lengths((args.profile, args.reads, 1000 + 7919 * u)) for u in range(units)]
        from elector_amd import distributed as edist0
        lr_all = np.concatenate([x[0] for x in lens]); lc_all = np.concatenate([x[1] for x in lens]); lu_all = np.concatenate([x[2] for x in lens])
        weights = edist0.read_cell_estimate(lr_all, lc_all, lu_all)
        bounds = edist0.shard_bounds(weights, world)
        unit_first = np.concatenate([[0], np.cumsum([len(x[0]) for x in lens])])
        r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
        for u in range(units):
            lo, hi = max(r0, int(unit_first[u])), min(r1, int(unit_first[u + 1]))
            if lo < hi:
                jobs.append((args.profile, args.reads, 1000 + 7919 * u, (lo - int(unit_first[u]), hi - int(unit_first[u]))))
        est_per_rank = [float(weights[int(bounds[r]):int(bounds[r + 1])].sum()) for r in range(world)]
        n_head = len(jobs)
    else:
        for b in range(n_batches):
            jobs.append((args.profile, args.reads, 1000 + rank + 7919 * b, None))
        n_head = len(jobs)
        if with_configs:
            for prof, reads in CONFIGS_ARRAY:
                if prof != args.profile:
                    # (a caller that asks for small batches -- the tests -- gets small entries too)
                    jobs.append((prof, reads if args.reads >= 10001 else min(reads, args.reads), 1000, None))
    nproc = max(1, min(len(jobs), ncores // 4 if ncores >= 8 else 1, 8))
    nthreads = max(1, ncores // nproc)
    t_setup = time.perf_counter()
    # (one job, or ELECTOR_BENCH_NO_FORK=1 -- runs under a profiler, whose preloaded library has initialised the GPU
    # before this program starts: in this process)
    if len(jobs) > 1 and os.environ.get("ELECTOR_BENCH_NO_FORK", "0") in ("", "0"):
        with ProcessPoolExecutor(max_workers=nproc, mp_context=mp.get_context("fork")) as ex:
            prepared = list(ex.map(prepare_batch, [(j[0], j[1], j[2], nthreads, j[3]) for j in jobs]))
    else:
        prepared = [prepare_batch((j[0], j[1], j[2], ncores, j[3])) for j in jobs]
    t_setup = time.perf_counter() - t_setup
    head_batches, other_batches = prepared[:n_head], prepared[n_head:]

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

    from elector_amd import distributed as edist
    from elector_amd.poa import PoaEngine
    from elector_amd._capi import ES_NCOUNTERS
    dev = torch.device("cuda", local)
    host_offsets = os.environ.get("ELECTOR_BENCH_HOST_OFFSETS", "0") not in ("", "0")

    def upload(b):
        """bases AND offsets resident in HBM (the device splitter leaves both there)"""
        b.d_bases = torch.from_numpy(b.win.bases).to(dev) if b.n else torch.zeros(64, dtype=torch.uint8, device=dev)
        b.d_off = torch.from_numpy(np.ascontiguousarray(b.win.off, dtype=np.int64)).to(dev)

    for b in head_batches:
        upload(b)
    all_batches = head_batches + other_batches
    max_total = max([b.total for b in all_batches] + [64])
    max_n = max([b.n for b in all_batches] + [1])
    # E engine contexts (four by default) take the batches in turn (several in flight: the serial head and tail
    # of one batch -- classification / symbolize / trivial pass / list sort, merge / statistics -- run beside the
    # alignment kernels of the others).  Every context has its own output buffers.
    n_eng = 1 if args.serial else max(1, int(os.environ.get("ELECTOR_BENCH_ENGINES", "4")))
    engines = [PoaEngine(local) for _ in range(n_eng)]
    if args.serial:
        engines[0].option("chains", 1)
    outs = [(torch.empty(3 * max_total + 64, dtype=torch.uint8, device=dev),
             torch.empty(max_n, dtype=torch.int32, device=dev), torch.empty(max_n, dtype=torch.int32, device=dev))
            for _ in range(n_eng)]
    rows_cap = 3 * max_total + 64
    # two pinned destinations per context, used in turn: a job's rows leave for the host when the job is collected and
    # travel while the context already works on its next batch
    pinned = [[torch.empty(rows_cap, dtype=torch.uint8).pin_memory() for _ in range(2)] for _ in range(n_eng)]
    pin_turn = [0] * n_eng

    def align(engine, b, dc, dn, ds):
        if host_offsets:
            engine.align_device(b.d_bases, b.win.off, dc, dn, ds)
        else:
            engine.align_device_offsets(b.d_bases, b.d_off, b.n, b.total, dc, dn, ds)

    class Runner(object):
        """the step loop over a list of units (a unit = the batches of one step)"""

        def __init__(self, units, gather=False):
            self.units = units
            self.pending = []                 # (engine, step id, batch, pieces) oldest first
            self.turn = 0
            self.step_id = 0
            self.parts = {}                   # step id -> [unit index, counters of its batches so far]
            self.host_enqueue_s = self.host_wait_s = self.gather_s = 0.0
            self.last_counters = None
            self.last_batch_of_engine = [None] * n_eng
            self.rows_bytes = 0
            self.done_at = {}                 # step id -> when its counters were on the host
            self.gather = gather and dist_on
            self.held, self.gathers = [], []
            if self.gather:
                # every rank tells the others once how many counter rows each unit brings
                self.sizes_u = [edist.gather_sizes(sum(len(b.piece_first) - 1 for b in u)) for u in units]
                self.gather_cap = max(1, args.gather_every) * max(max(s) for s in self.sizes_u)

        def collect_oldest(self):
            e, sid, b, npieces = self.pending.pop(0)
            t = time.perf_counter()
            counters, pc = engines[e].msa_stats_collect(npieces)
            self.host_wait_s += time.perf_counter() - t
            self.rows_bytes = 3 * int(pc.sum())
            ui, got = self.parts[sid]
            got.append(counters)
            if len(got) == len(self.units[ui]):
                del self.parts[sid]
                self.done_at[sid] = time.perf_counter()
                c = np.concatenate(got) if len(got) > 1 else got[0]
                if self.gather:
                    tg = time.perf_counter()
                    self.held.append((ui, c))
                    res = self.flush_gather(False)
                    if res is not None:
                        self.last_counters = res
                    self.gather_s += time.perf_counter() - tg
                else:
                    self.last_counters = c

        def flush_gather(self, force):
            """Fewer, larger collectives: the counters of `gather_every` steps travel together (every step's counters
            still reach rank 0 inside the timed region).  An RCCL collective holds compute units while it runs and its
            small copies queue up behind whole launch chains: one gather per step cost 19 % of the step rate in a one-rank
            rehearsal on RCCL, one per eight steps nothing measurable.  Staging, collective and taking the result in run
            on a helper thread; up to three are in flight.  -> the last step's rows of the oldest finished gather (rank 0)"""
            res = None
            if self.held and (force or len(self.held) >= max(1, args.gather_every)):
                held, self.held = self.held, []
                block = np.concatenate([c for _, c in held]) if len(held) > 1 else held[0][1]
                sizes = [sum(self.sizes_u[ui][r] for ui, _ in held) for r in range(len(self.sizes_u[0]))]
                last = [(sum(self.sizes_u[ui][r] for ui, _ in held[:-1]), self.sizes_u[held[-1][0]][r]) for r in range(len(sizes))]
                self.gathers.append((sizes, last, gpool.submit(
                    lambda c=block, z=sizes: edist.gather_rows_async(c, z, cap=self.gather_cap,
                                                                     device=local if backend == "nccl" else None).wait())))
            while self.gathers and (force or len(self.gathers) > 3):
                sizes, last, fut = self.gathers.pop(0)
                got = fut.result()
                if got is not None:
                    parts, at = [], 0
                    for r, z in enumerate(sizes):
                        parts.append(got[at + last[r][0]: at + last[r][0] + last[r][1]])
                        at += z
                    res = np.concatenate(parts) if len(parts) > 1 else parts[0]
            return res

        def step(self, rows):
            """Queue one step: per batch of the unit, windows in HBM -> POA kernels -> merge -> counters (and with `rows`
            the merged rows) -> pinned host memory.  A context's previous job is taken in right before the context is
            used again: the host prepares step i+1 while the GPU still works on the batches before it, as a run over
            many 10,001-read batches would."""
            ui = self.step_id % len(self.units)
            sid = self.step_id
            self.step_id += 1
            self.parts[sid] = [ui, []]
            for b in self.units[ui]:
                e = self.turn % n_eng
                self.turn += 1
                while len(self.pending) >= n_eng:
                    self.collect_oldest()
                dc, dn, ds = outs[e]
                t = time.perf_counter()
                align(engines[e], b, dc, dn, ds)
                pin_turn[e] ^= 1
                npieces = engines[e].msa_stats_enqueue(b.n, dc, dn, ds, b.piece_first, b.read_first,
                                                       rows_out=pinned[e][pin_turn[e]].data_ptr() if rows else None, rows_cap=rows_cap)
                self.host_enqueue_s += time.perf_counter() - t
                self.pending.append((e, sid, b, npieces))
                self.last_batch_of_engine[e] = b

        def drain(self):
            while self.pending:
                self.collect_oldest()
            for g in engines:
                g.msa_rows_wait()                 # ... and every step's rows have arrived
            if self.gather:
                res = self.flush_gather(True)
                if res is not None:
                    self.last_counters = res

        def timed(self, steps, rows):
            if dist_on:
                dist.barrier()
            torch.cuda.synchronize()
            for g in engines:
                g.sync()
            self.host_enqueue_s = self.host_wait_s = self.gather_s = 0.0
            self.first_timed = self.step_id
            t0 = self.t0 = time.perf_counter()
            for _ in range(steps):
                self.step(rows)
            self.drain()                         # every step's counters (and rows) are on the host before the clock stops
            for g in engines:
                g.sync()
            torch.cuda.synchronize()
            if dist_on:
                dist.barrier()
            return time.perf_counter() - t0

    gpool = None
    if dist_on:
        from concurrent.futures import ThreadPoolExecutor as _TPE
        # (a new thread's current device is 0 whatever this thread set: the helper binds itself to the rank's GPU, and the
        # pipe is told the device as well)
        gpool = _TPE(max_workers=1, initializer=torch.cuda.set_device, initargs=(local,))

    def measure(batches, steps, warmup, units=None, gather=False, second_loop=True):
        """warm up, then the two timed loops over `batches` -> dict of raw results"""
        units = units if units is not None else [[b] for b in batches]
        run = Runner(units, gather)
        # untimed: grow every workspace (both statistics slots of every context) and let the HIP runtime size its queues
        # for overlapped batches -- the first batch enqueued while another still runs pays a one-time ~14 ms inside it
        for _ in range(max(3 * n_eng // max(1, len(units[0])), len(units)) + 1):
            run.step(True)
        run.drain()
        # |PO| per window and the column total of every batch (DP cells, algorithmic bytes) from an untimed pass
        for b in batches:
            if b.po is None:
                dc, dn, ds = outs[0]
                align(engines[0], b, dc, dn, ds)
                # (the merge / statistics kernels too: every pass of this process then has the same launches, which the
                # per-step arithmetic of tools/pmc_traffic.py relies on)
                engines[0].msa_stats_collect(engines[0].msa_stats_enqueue(b.n, dc, dn, ds, b.piece_first, b.read_first))
                engines[0].sync()
                st = ds[:b.n].cpu().numpy()
                if st.any():
                    raise SystemExit("bench: %d windows failed on device" % int((st != 0).sum()))
                b.po = engines[0].last_po_sizes(b.n).astype(np.int64)
                b.ncol_sum = int(dn[:b.n].cpu().numpy().astype(np.int64).sum())
        for _ in range(warmup):
            run.step(True)
        run.drain()
        for g in engines:
            g.sync()
            g.timing_enable(args.serial)
            g.timing_reset()
        dt = run.timed(steps, True)
        # between the completions of the sixth and the last timed step: the region without its fill (the first batches find an
        # empty pipeline) -- the drain at the end stays in
        steady = None
        f0 = run.first_timed
        if steps >= 8 and f0 + 5 in run.done_at and f0 + steps - 1 in run.done_at:
            steady = (run.done_at[f0 + steps - 1] - run.done_at[f0 + 5]) / (steps - 6)
        done_ms = [round((run.done_at[f0 + k] - run.t0) * 1e3, 2) for k in range(steps) if f0 + k in run.done_at]
        res = {"dt": dt, "steady_s_per_step": steady, "done_ms": done_ms, "host_enqueue_ms": run.host_enqueue_s / steps * 1e3, "host_wait_ms": run.host_wait_s / steps * 1e3,
               "gather_ms": run.gather_s / steps * 1e3, "counters": run.last_counters, "rows_bytes": run.rows_bytes,
               "last_batch_of_engine": list(run.last_batch_of_engine), "dt_hbm": None}
        # the MSA the timed kernels wrote (context 0's last timed batch), for the comparison with the reference binary
        # behind the clock
        # (one entry per DISTINCT batch: the last timed steps went round the contexts, so between them the contexts hold every
        # batch of the rotation)
        res["msa0"] = None
        res["msa_all"] = []
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            seen = set()
            for e in range(n_eng):
                be = run.last_batch_of_engine[e]
                if be is None or id(be) in seen:
                    continue
                seen.add(id(be))
                res["msa_all"].append((be, outs[e][0][:3 * be.total + 64].cpu().numpy(), outs[e][1][:be.n].cpu().numpy()))
            if res["msa_all"]:
                res["msa0"] = res["msa_all"][0]
        if second_loop:
            run2 = Runner(units, gather)
            for _ in range(n_eng + 1):
                run2.step(False)
            run2.drain()
            res["dt_hbm"] = run2.timed(steps, False)
        return res

    # ---- headline: the timed region ----------------------------------------------------
    head_units = [list(head_batches)] if strong else None
    steps = args.steps
    R = measure(head_batches, steps, args.warmup, units=head_units, gather=True, second_loop=not args.no_second_loop and not args.serial)
    dt, dt_hbm = R["dt"], R["dt_hbm"]
    counters = R["counters"]
    b0 = head_batches[0] if head_batches else None

    # ---- the `configs` array: the other single-GPU profiles, one batch each ----
    configs = []
    if with_configs:
        for ob in other_batches:
            upload(ob)
            Ro = measure([ob], max(1, args.configs_steps), 2, second_loop=True)
            entry = {"profile": ob.profile, "workload": WORKLOADS[ob.profile], "reads_per_step": ob.n_reads, "windows_per_step": ob.n,
                     "ref_bases_per_step": ob.piece_bases, "steps": max(1, args.configs_steps), "batches_rotated": 1,
                     "value": round(ob.piece_bases * max(1, args.configs_steps) / Ro["dt"] / 1e6, 3),
                     "ms_per_step": round(Ro["dt"] / max(1, args.configs_steps) * 1e3, 3),
                     "value_rows_in_hbm": round(ob.piece_bases * max(1, args.configs_steps) / Ro["dt_hbm"] / 1e6, 3),
                     "counters_checksum": int(Ro["counters"][:, :ES_NCOUNTERS - 1].sum())}
            if Ro["msa0"] is not None:
                mb, mcols, mncol = Ro["msa0"]
                base, par = cpu_baseline(mb.win, mb.lr, min(4.0, args.cpu_seconds), (mcols, mncol))
                entry["cpu_baseline"] = base
                if par is not None:
                    entry["parity_vs_reference"] = par
            configs.append(entry)
            ob.d_bases = ob.d_off = None

    # ---- after the clock: serial pass for the per-kernel roofline -------------------------------------------
    # A context of its own with ONE launch chain from its first step on, after the timed contexts are gone: exactly
    # what `bench.py --serial --batches 1` times and what profiles/*_serial_kernel_stats.csv (rocprofv3 over that
    # command) shows, so the figures below can be recomputed from the committed profile.  Batch 0 only.
    serial_steps = args.steps if args.serial else max(1, args.serial_steps)
    serial_wall = dt / args.steps
    d_cols, d_ncol, d_status = outs[0]
    if not args.serial and b0 is not None:
        for g in engines[1:]:
            g.close()
        engines[0].close()
        eng = PoaEngine(local)
        engines = [eng]
        eng.option("chains", 1)

        # (the serial pass goes round the batches of the rotation like the timed region: a step per batch and round)
        ser_batches = [b0] if strong else list(head_batches)
        serial_steps = len(ser_batches) * max(1, (serial_steps + len(ser_batches) - 1) // len(ser_batches))

        def serial_step(i):
            bs = ser_batches[i % len(ser_batches)]
            align(eng, bs, d_cols, d_ncol, d_status)
            eng.msa_stats_collect(eng.msa_stats_enqueue(bs.n, d_cols, d_ncol, d_status, bs.piece_first, bs.read_first))
        for i in range(max(4, len(ser_batches))):  # grow the workspace, settle the clocks
            serial_step(i)
        eng.sync()
        eng.timing_enable(True)
        eng.timing_reset()
        ts = time.perf_counter()
        for i in range(serial_steps):
            serial_step(i)
        eng.sync()
        serial_wall = (time.perf_counter() - ts) / serial_steps
    eng = engines[0]
    t_dp1, k_dp1 = eng.timing_read(0)
    t_dp2, k_dp2 = eng.timing_read(1)
    t_oth = eng.timing_read(2)[0]
    t_st = eng.timing_read(3)[0]
    t_poa, k_poa = eng.timing_read(4)
    t_far = eng.timing_read(6)[0]

    # per step (averaged over the rotation): bases, windows, DP cells
    def per_step(f):
        if strong:
            return float(sum(f(b) for b in head_batches))
        return float(sum(f(b) for b in head_batches)) / max(1, len(head_batches))
    skipped = {id(b): skipped_alignment1(b.win, b.lr, b.lc) for b in head_batches}
    bases_step = per_step(lambda b: b.piece_bases)
    windows_step = per_step(lambda b: b.n)
    cells_step = per_step(lambda b: int((b.lr * b.lc).sum()) + int((b.po * b.lu).sum()))
    cells_comp_step = per_step(lambda b: int((b.lr * b.lc)[~skipped[id(b)]].sum()) + int((b.po * b.lu).sum()))
    my_cells = float(sum(int((b.lr * b.lc).sum()) + int((b.po * b.lu).sum()) for b in head_batches)) if strong else cells_step
    rdev = dev if backend == "nccl" else torch.device("cpu")
    tmax = torch.tensor([dt, dt_hbm or 0.0], dtype=torch.float64, device=rdev)
    tot = torch.tensor([bases_step, windows_step, cells_step, cells_comp_step], dtype=torch.float64, device=rdev)
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt_max, dt_hbm_max = float(tmax[0].item()), float(tmax[1].item())
    # what the process group actually spans: one entry per rank, gathered over it (RCCL when the backend is nccl)
    props = torch.cuda.get_device_properties(local)
    me = {"rank": rank, "local_rank": local, "device": props.name, "uuid": str(getattr(props, "uuid", "")),
          "pci_bus_id": int(getattr(props, "pci_bus_id", -1)), "host": socket.gethostname(),
          "reads": int(sum(b.n_reads for b in head_batches)) if strong else int(b0.n_reads if b0 else 0),
          "dp_cells_per_step": my_cells,
          # this rank's host side of the timed region (ms per step) and its setup: what eight ranks on one node's cores cost
          "host_classify_and_enqueue_ms": round(R["host_enqueue_ms"], 3), "host_wait_for_results_ms": round(R["host_wait_ms"], 3),
          "ms_per_step": round(dt / args.steps * 1e3, 3), "setup_s": round(t_setup, 1),
          "counters_checksum": int(counters[:, :ES_NCOUNTERS - 1].sum()) if (counters is not None and rank == 0) else None}
    ranks = [me]
    if dist_on:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    bases_all, windows_all, cells_all, cells_comp_all = (float(x) for x in tot.tolist())

    exit_code = 0
    if rank == 0:
        value = bases_all * args.steps / dt_max / 1e6
        step_s = dt_max / args.steps
        # roofline of the dominant kernel: algorithmic bytes = 8-bit inputs + 8-bit MSA out + descriptors (a step's average
        # over the rotated batches, which the serial pass goes round too)
        ab = [b0] if (strong or args.serial and len(head_batches) == 1) else list(head_batches)
        alg_bytes = int(sum(int((b.lr + b.lc + b.lu).sum()) + 3 * b.ncol_sum + 28 * b.n for b in ab) / max(1, len(ab))) if b0 is not None else 0
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
        cells_ranks = [r["dp_cells_per_step"] for r in ranks]
        out = {
            "metric": "triplet-MSA Mbases/s", "value": round(value, 3), "unit": "Mbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            # the timed region without its fill: from the completion (counters on the host) of its sixth step to that of its last
            "steady_state": None if not R.get("steady_s_per_step") else {
                "ms_per_step": round(R["steady_s_per_step"] * 1e3, 3), "value": round(bases_all / R["steady_s_per_step"] / 1e6, 3),
                "step_completions_ms": R.get("done_ms"),
                "note": "steps 6 .. %d of the timed region between completion events on rank 0; `value` and `ms_per_step` above are "
                        "the whole region (fill and drain included)" % args.steps},
            "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": "%s: %d reads per GPU per step, cut into windows by the ELECTOR splitter rules"
                                   % (WORKLOADS[args.profile], args.reads) if not strong else
                                   "%s: ONE read set of %d x %d reads per step whatever the GPU count, cut into per-rank ranges by "
                                   "DP-cell estimate (distributed.shard_bounds)" % (WORKLOADS[args.profile], args.strong_units, args.reads),
                       "profile": args.profile, "reads_per_gpu": args.reads, "triples_per_gpu": int(b0.n_pieces_in if b0 else 0),
                       "windows_per_gpu": int(round(windows_all / world)), "ref_bases_per_gpu": int(round(bases_all / world)),
                       "filler_windows_per_gpu": int((b0.lc == 1).sum()) if b0 is not None else 0,
                       "parallelism": "shard-by-read x%d" % world,
                       "batches_in_flight_per_gpu": n_eng, "serial": bool(args.serial),
                       "batches_rotated": len(head_batches), "input_bytes_rotated_per_gpu": int(sum(b.total + 8 * (3 * b.n + 1) for b in head_batches)),
                       # `value`: per-read counters AND the merged MSA rows back on the host; a kernel packs
                       # the rows, one copy per batch takes them to pinned host memory on the context's copy stream
                       "rows_to_host": True, "offsets_resident_in_hbm": not host_offsets},
            "rows_to_host": {"bytes_per_step_per_gpu": int(R["rows_bytes"]),
                             "pcie_gbs_per_gpu": round(R["rows_bytes"] * (len(head_batches) if strong else 1) / step_s / 1e9, 2)},
            "value_rows_in_hbm": None if not dt_hbm_max else round(bases_all * args.steps / dt_hbm_max / 1e6, 3),
            "rows_in_hbm": None if not dt_hbm_max else {
                "ms_per_step": round(dt_hbm_max / args.steps * 1e3, 3),
                "note": "the same %d steps timed again with the merged rows left in HBM: call site #2 is served from the "
                        "device counters (SURVEY.md 8(f2)), only they cross PCIe" % args.steps},
            "ranks": {"world": world, "backend": backend if dist_on else None,
                      "distinct_devices": len({(r["host"], r["uuid"] or r["pci_bus_id"]) for r in ranks}),
                      "dp_cells_imbalance_max_over_mean": round(max(cells_ranks) / (sum(cells_ranks) / len(cells_ranks)), 4) if sum(cells_ranks) > 0 else None,
                      "devices": ranks},
            # DP cells per second.  effective: every cell the reference computes (Lr*Lc + |PO|*Lu per window);
            # computed: without alignment #1 of the windows whose corrected sequence equals the reference or
            # differs by one substitution or one indel, which the device settles without a dynamic program (k_trivial)
            "gcups_effective": round(cells_all * args.steps / dt_max / 1e9, 3),
            "gcups_computed": round(cells_comp_all * args.steps / dt_max / 1e9, 3),
            "alignment1_skipped_windows_frac": round(float(np.mean(np.concatenate([skipped[id(b)] for b in head_batches]))), 4) if head_batches else None,
            "kernel_ms_per_step": {"k_poa": round(t_poa / serial_steps, 3),
                                   # k_poa<G, 8, true>: the windows with one far edge, one launch per lane-group size (a few
                                   # hundred wavefronts each: as long as their longest window when alone on the chip)
                                   "k_poa_far_instance": round(t_far / serial_steps, 3),
                                   "alignment1_stage": round(t_dp1 / serial_steps, 3),
                                   "alignment2_stage": round(t_dp2 / serial_steps, 3),
                                   "other": round(t_oth / serial_steps, 3),
                                   "merge_and_counters": round(t_st / serial_steps, 3),
                                   "serial_step_wall": round(serial_wall * 1e3, 3),
                                   # host time of the timed region per step: inside the two calls that queue a batch (they
                                   # include one short wait for the classification kernel's totals), and waiting for results
                                   "host_classify_and_enqueue": round(R["host_enqueue_ms"], 3),
                                   "host_wait_for_results": round(R["host_wait_ms"], 3),
                                   "host_counters_gather": round(R["gather_ms"], 3),
                                   "note": "HIP-event time per launch, summed per step, from %d un-overlapped steps over the %d rotated "
                                           "batches (one context, one launch chain)%s"
                                           % (serial_steps, len(head_batches), "" if args.serial else
                                              " in a fresh context after the timed region, = what `bench.py --serial` times")},
            "roofline": {"bound": "hbm", "kernel": dom[0], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "traffic_provenance": pmc_prov if traffic is not None else None,
                         "launches": int(launches), "avg_launch_ms": round(avg_ms, 4),
                         "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "whole_step_frac": round(alg_bytes / step_s / 1e9 / HBM_PEAK_GBS, 6)},
            # what actually binds: VALU issue + latency.  Per GPU: wave-instructions per
            # step from the committed PMC passes of `bench.py --serial` (averaged over the rotated batches), time measured live
            "roofline_valu": None if valu is None else {
                "bound": "valu-issue", "wave_insts_per_step": valu, "provenance": pmc_prov, "peak": round(VALU_PEAK_GINSTS, 1),
                "peak_two_operand_32bit": round(VALU_PEAK_GINSTS_SIMPLE, 1),
                "peak_note": "packed 16-bit, three-operand and DPP instructions issue once per 4 cycles per SIMD "
                             "(measured: tests/micro/valu_rate.hip, profiles/r02_valu_rate.txt)",
                "unit": "G wave-insts/s per GPU",
                # of `value`'s own step time; the same with the rows left in HBM beside it
                "achieved": round(valu / step_s / 1e9, 1),
                "frac": round(valu / step_s / 1e9 / VALU_PEAK_GINSTS, 4),
                "achieved_rows_in_hbm": None if not dt_hbm_max else round(valu / (dt_hbm_max / args.steps) / 1e9, 1),
                "frac_rows_in_hbm": None if not dt_hbm_max else round(valu / (dt_hbm_max / args.steps) / 1e9 / VALU_PEAK_GINSTS, 4)},
            "pieces_gathered": int(counters.shape[0]) if counters is not None else 0,
            "counters_checksum": int(counters[:, :ES_NCOUNTERS - 1].sum()) if counters is not None else 0,
            "setup_s": round(t_setup, 1),
        }
        if strong:
            out["strong"] = {"units": args.strong_units, "reads_total": int(sum(r["reads"] for r in ranks)),
                             "estimated_cells_per_rank": est_per_rank,
                             "estimate_imbalance_max_over_mean": round(max(est_per_rank) / (sum(est_per_rank) / len(est_per_rank)), 4)}
        parity = None
        if world == 1 and not args.no_cpu_baseline and R["msa0"] is not None:
            # the MSA columns of a batch the timed region worked on, as the last timed batch of context 0 left them
            mb, mcols, mncol = R["msa0"]
            out["cpu_baseline"], parity = cpu_baseline(mb.win, mb.lr, args.cpu_seconds, (mcols, mncol))
            # ... and the other batches of the rotation, each as a context left it (their CPU time is not the baseline's)
            for (ob_, ocols, oncol) in R["msa_all"][1:]:
                _, par2 = cpu_baseline(ob_.win, ob_.lr, args.cpu_seconds, (ocols, oncol))
                if parity is not None and par2 is not None:
                    parity["windows"] += par2["windows"]
                    parity["differing"] += par2["differing"]
            if parity is not None:
                parity["batches_compared"] = len(R["msa_all"])
                parity["batches_rotated"] = len(head_batches)
                out["parity_vs_reference"] = parity
        if configs:
            out["configs"] = configs
        print(json.dumps(out), flush=True)
        bad = [("headline", parity)] + [(c["profile"], c.get("parity_vs_reference")) for c in configs]
        for name, par in bad:
            if par is not None and par["differing"]:
                sys.stderr.write("bench.py: %s: %d of %d windows differ from the reference poa's output\n"
                                 % (name, par["differing"], par["windows"]))
                exit_code = 1
    for g in engines:
        g.close()
    if dist_on:
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
