#!/usr/bin/env python3
"""bench.py -- triplet-MSA throughput of the HIP hot path on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

Three engine contexts per GPU take the steps in turn, so three batches are in flight
(the serial head and tail of one batch overlap the alignment kernels of the others).

A *step* is one pass of the hot path (symbolize -> alignment #1 -> fusion ->
alignment #2 -> fusion + MSA columns -> merge of each read's windows -> per-read
integer counters back on the host) over one batch of window triples that is
already resident in HBM.  The batch is what ELECTOR's own batch protocol hands
to its POA engine: the windows of `--reads` synthetic long reads
(BASELINE.json configs[1] profile: E. coli 30X SimLord-like PacBio reads, 15 %
error, LoRDEC-like 1 % corrected), cut by this repository's reference-compatible
splitter on the host before the timed region.  Weak scaling: every rank
processes its own shard of reads (independent triples, no data-path
collective); rank 0 gathers the per-read integer counters over RCCL once per step.

Prints ONE JSON line (rank 0).  `value` = reference-read bases of all ranks per
second of the slowest rank.  `roofline` prices the dominant kernel against HBM
peak using the algorithmic bytes of DESIGN.md; `cpu_baseline` times the
reference poaV2 binary (oracle/_ref/poa, when it travelled with the snapshot)
or the C oracle port on this host's cores over a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
WORKLOADS = {   # BASELINE.json configs restated as synthetic profiles (elector_amd/synthetic.py)
    "ecoli30x_simlord_lordec": "E. coli 30X SimLord-like PacBio (15% err), LoRDEC-like corrected (1% err), ~8 kb reads",
    "yeast50x_nanosim_consent": "S. cerevisiae 50X NanoSim-like ONT (12% err), CONSENT-like corrected (2% err), ~8 kb reads",
    "chr1_20x_ont_50kb": "Human chr1 20X NanoSim-like ONT (12% err), corrected (2% err), ~50 kb reads",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("ELECTOR_BENCH_READS", "10001")),
                    help="synthetic long reads per rank and step (10,001 = one batch of ELECTOR's own protocol, "
                         "elector/alignment.py:82, Master_Splitter.cpp:397-399)")
    ap.add_argument("--profile", default="ecoli30x_simlord_lordec")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline leg")
    return ap.parse_args()


def cpu_baseline(windows, ref_bases_per_window, seconds):
    """Reference poaV2 (or the oracle port) on this host's cores over a bounded
    sample of the same window stream.  Test infrastructure: uses oracle/."""
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
        nb = float(cum[ns - 1])
        return {"value": round(nb / dt / 1e6, 4), "unit": "Mbases/s", "cores": len(cmds), "kind": "reference",
                "sample": "%d windows (%d reference bases) of the step's window stream, one reference poa "
                          "process per core as elector/alignment.py's Pool does, wall %.2f s" % (ns, int(nb), dt)}
    # port: single-threaded C oracle
    target_bases = 0.4e6 * seconds
    cum = np.cumsum(ref_bases_per_window)
    ns = int(min(nwin, np.searchsorted(cum, target_bases) + 1))
    t0 = time.perf_counter()
    oracle_lib.batch(windows.bases[: off[3 * ns]], off[: 3 * ns + 1])
    dt = time.perf_counter() - t0
    nb = float(cum[ns - 1])
    return {"value": round(nb / dt / 1e6, 4), "unit": "Mbases/s", "cores": 1, "kind": "port",
            "sample": "%d windows (%d reference bases), oracle/poa_oracle.c single thread, wall %.2f s" % (ns, int(nb), dt)}


def pmc_traffic(kernel, reads):
    """HBM bytes per launch of `kernel` from the committed PMC passes over this very command
    (profiles/pmc_traffic.json, written from tests/_pmc_bench.sh: FETCH_SIZE and WRITE_SIZE in
    separate rocprofv3 passes, gfx950 correction applied); None when the workload differs."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        if int(t["reads_per_gpu"]) != int(reads):
            return None
        return int(t["kernels"][kernel]["traffic_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def pmc_valu(reads):
    """Wavefront-VALU instructions per step of the two fused kernel families (SQ_INSTS_VALU, same
    committed PMC passes); None when the workload differs."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        return int(t["valu_wave_insts_per_step"]) if int(t["reads_per_gpu"]) == int(reads) else None
    except (OSError, KeyError, ValueError):
        return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from elector_amd import split, synthetic
    from elector_amd.poa import PoaEngine

    # ---- untimed setup: synthetic reads -> windows (host), upload --------
    reads = synthetic.read_triples(args.profile, args.reads, seed=1000 + rank)
    read_bases = int(sum(len(r[0]) for r in reads))
    nthreads = max(1, (os.cpu_count() or 1) // max(1, world))
    win = split.split_reads(reads, 0.1, None, nthreads=nthreads)
    del reads
    off = win.off
    n = win.n_windows
    lr = off[1::3] - off[0:-1:3]
    lc = off[2::3] - off[1:-1:3]
    lu = off[3::3] - off[2:-1:3]
    dev = torch.device("cuda", local)
    d_bases = torch.from_numpy(win.bases).to(dev)
    # E engine contexts take the steps in turn (several batches in flight: the serial head and tail
    # of one batch -- symbolize / trivial pass / list sort, merge / statistics -- run beside the
    # alignment kernels of the other).  Every context has its own output buffers.
    # Measured on the 10,001-read step: 16.0 ms with one context, 14.3 with two, 13.6 with three, 14.8 with four.
    n_eng = max(1, int(os.environ.get("ELECTOR_BENCH_ENGINES", "3")))
    engines = [PoaEngine(local) for _ in range(n_eng)]
    outs = [(torch.empty(3 * int(off[-1]) + 64, dtype=torch.uint8, device=dev),
             torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
            for _ in range(n_eng)]
    eng = engines[0]
    d_cols, d_ncol, d_status = outs[0]
    # one msa.fa record (piece) per read, one piece per read: the synthetic corrected reads are not split
    piece_first = win.read_first
    read_first = np.arange(win.n_reads + 1, dtype=np.int64)
    from elector_amd import distributed as edist
    from elector_amd._capi import ES_NCOUNTERS

    pending = []
    turn = [0]

    def collect():
        """per-read counters of the oldest queued step on the host (rank 0 receives every rank's rows)"""
        e, npieces = pending.pop(0)
        counters, _ = engines[e].msa_stats_collect(npieces)
        return edist.gather_rows(counters) if world > 1 else counters

    def step():
        """Queue one step (windows in HBM -> POA kernels -> merge -> counters -> pinned host memory),
        then hand out the counters of the oldest step in flight: the host prepares step i+1 while the
        GPU still works on step i, as a run over many 10,001-read batches would."""
        e = turn[0] % n_eng
        turn[0] += 1
        dc, dn, ds = outs[e]
        engines[e].align_device(d_bases, off, dc, dn, ds)
        pending.append((e, engines[e].msa_stats_enqueue(n, dc, dn, ds, piece_first, read_first)))
        return collect() if len(pending) > n_eng else None

    # untimed setup, continued: grow every workspace (both halves of the double-buffered upload staging
    # and of the statistics slots) and let the HIP runtime size its queues for overlapped batches -- the
    # first batch that is enqueued while another still runs pays a one-time ~14 ms inside the runtime
    for _ in range(3 * n_eng):
        step()
    while pending:
        collect()
    for _ in range(args.warmup):
        step()
    while pending:
        collect()
    for g in engines:
        g.sync()
        g.timing_enable(True)
        g.timing_reset()

    # ---- timed region ----------------------------------------------------
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    for g in engines:
        g.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    while pending:
        counters = collect()                     # every step's counters are on the host before the clock stops
    for g in engines:
        g.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0

    # ---- after the clock: checks, counters, gather -------------------------
    status = d_status.cpu().numpy()
    ncol = d_ncol.cpu().numpy().astype(np.int64)
    if status.any():
        raise SystemExit("bench: %d windows failed on device" % int((status != 0).sum()))
    po = eng.last_po_sizes(n).astype(np.int64)
    cells1, cells2 = int((lr * lc).sum()), int((po * lu).sum())
    t_dp1, k_dp1, t_dp2, k_dp2, t_oth, t_st = 0.0, 0, 0.0, 0, 0.0, 0.0
    for g in engines:
        x, y = g.timing_read(0); t_dp1 += x; k_dp1 += y
        x, y = g.timing_read(1); t_dp2 += x; k_dp2 += y
        t_oth += g.timing_read(2)[0]
        t_st += g.timing_read(3)[0]
    rdev = dev if backend == "nccl" else torch.device("cpu")
    tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
    tot = torch.tensor([read_bases, n, cells1 + cells2], dtype=torch.int64, device=rdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt_max = float(tmax.item())
    bases_all, windows_all, cells_all = (int(x) for x in tot.tolist())

    if rank == 0:
        value = bases_all * args.steps / dt_max / 1e6
        # roofline of the dominant kernel: algorithmic bytes = 8-bit inputs + 8-bit MSA out + descriptors
        alg_bytes = int((lr + lc + lu).sum() + 3 * ncol.sum() + 28 * n)
        # kernel classes: 0 = alignment #1 stage (k_fused_a<G>, or k_dp1 on the generic path),
        # 1 = alignment #2 stage (k_fused_b<G> / k_dp2); launches of the geometry classes run
        # concurrently on separate streams, so their event times overlap in wall time
        dom = ("k_fused_b", t_dp2, k_dp2) if t_dp2 >= t_dp1 else ("k_fused_a", t_dp1, k_dp1)
        launches = max(1, dom[2])
        avg_ms = dom[1] / launches
        bytes_per_launch = alg_bytes * args.steps / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = pmc_traffic(dom[0], args.reads)
        out = {
            "metric": "triplet-MSA Mbases/s", "value": round(value, 3), "unit": "Mbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s: %d reads per GPU per step, cut into windows by the ELECTOR splitter rules"
                                   % (WORKLOADS.get(args.profile, args.profile), args.reads),
                       "profile": args.profile, "reads_per_gpu": args.reads, "windows_per_gpu": n,
                       "ref_bases_per_gpu": read_bases, "parallelism": "shard-by-read x%d" % world,
                       "batches_in_flight_per_gpu": n_eng},
            "gcups": round(cells_all * args.steps / dt_max / 1e9, 3),
            "kernel_ms_per_step": {"alignment1_stage": round(t_dp1 / args.steps, 3),
                                   "alignment2_stage": round(t_dp2 / args.steps, 3),
                                   "other": round(t_oth / args.steps, 3),
                                   "merge_and_counters": round(t_st / args.steps, 3),
                                   "note": "sum of per-launch HIP-event times; the two launch chains overlap"},
            "roofline": {"bound": "hbm", "kernel": dom[0], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": traffic,
                         "launches": int(launches), "avg_launch_ms": round(avg_ms, 4),
                         "algorithmic_bytes_per_launch": int(bytes_per_launch)},
            # what actually binds: VALU issue.  peak = 256 CUs x 4 SIMDs x one wavefront
            # instruction per 4 cycles at 2.4 GHz; insts from the committed PMC passes, time measured live
            "roofline_valu": (lambda v: None if v is None else {
                "bound": "valu-issue", "wave_insts_per_step": v, "peak": 614.4, "unit": "G wave-insts/s",
                "achieved": round(v * world / (dt_max / args.steps) / 1e9, 1),
                "frac": round(v / (dt_max / args.steps) / 614.4e9, 4)})(pmc_valu(args.reads)),
            "reads_gathered": int(counters.shape[0]),
            "counters_checksum": int(counters[:, :ES_NCOUNTERS - 1].sum()),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(win, lr, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    for g in engines:
        g.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
