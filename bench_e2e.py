#!/usr/bin/env python3
"""bench_e2e.py -- the two call sites end to end: three FASTA files -> getPOA -> outputRecallPrecision.

    python bench.py --end-to-end [--profile P] [--reads R]        (or python bench_e2e.py ...)

What elector/__main__.py:140-141 does for one corrector: the sorted reference / uncorrected / corrected
FASTA files go through elector_amd.alignment.getPOA (reader thread: parse, reads to HBM, split on the GPU;
main thread: triplet MSAs, merge and per-piece counters on the GPU, msa.fa written from the merged
records; ELECTOR_HOST_SPLIT=1 splits on the host cores instead) and elector_amd.computeStats.outputRecallPrecision (the 19-tuple, report, side
files -- from the counters the device left behind, no second pass over msa.fa).

Prints ONE JSON line: end-to-end Mbases/s (reference-read bases of all triples / wall), a stage table
(seconds of host or wait time per stage, summed over batches; the reader thread's stages overlap the main
thread's) and, timed on this host's cores over a bounded sample of the same files, the reference chain
(oracle/_ref: masterSplitter -> poa per slot file under a Pool of all cores -> Donatello, exactly
elector/alignment.py:98-122) followed by the statistics port (oracle/stats_oracle.py; the reference module
itself is not on the GPU box).  This is a reported baseline, not the metric of BASELINE.json (bench.py's
default mode measures that).  Test infrastructure: uses oracle/.
"""
import argparse
import io
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def write_fasta(d, triples, headers, repeat=1):
    """the three sorted files; repeat > 1 writes the reads that many times over under distinct names (longer runs
    without generating more synthetic reads)"""
    paths = [os.path.join(d, n) for n in ("ref.fa", "cor.fa", "unc.fa")]
    with open(paths[0], "wb") as fr, open(paths[1], "wb") as fc, open(paths[2], "wb") as fu:
        for k in range(repeat):
            tag = b">" if repeat == 1 else b">x%d" % k
            for (r, c, u), h in zip(triples, headers):
                hk = tag + h[1:]
                fr.write(hk + b"\n" + r + b"\n")
                fc.write(hk + b"\n" + c + b"\n")
                fu.write(hk + b"\n" + u + b"\n")
    return paths


def _poa_slot(args):
    ref_dir, d, i, mat = args
    if os.stat(d + "/out3%d" % i).st_size != 0:
        subprocess.run([os.path.join(ref_dir, "poa"), "-pir", d + "/smsa%d" % i, "-preserve_seqorder",
                        "-corrected_reads_fasta", d + "/out3%d" % i, "-reference_reads_fasta", d + "/out1%d" % i,
                        "-uncorrected_reads_fasta", d + "/out2%d" % i, "-preserve_seqorder", "-threads", "1",
                        "-pathMatrix", mat], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return i


def reference_chain(ref_dir, paths, n_sample, cores):
    """masterSplitter -> 200 x poa under Pool(cores) -> 200 x Donatello on the first n_sample records of the
    files (elector/alignment.py:98-122), then the statistics port on the msa.fa it wrote."""
    from multiprocessing import Pool
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib
    import stats_oracle
    out = {"cores": cores, "records": n_sample}
    with tempfile.TemporaryDirectory() as d:
        sub = []
        bases = 0
        for p in paths:
            q = os.path.join(d, os.path.basename(p))
            with open(p, "rb") as f, open(q, "wb") as g:
                for k in range(2 * n_sample):
                    ln = f.readline()
                    if not ln:
                        break
                    g.write(ln)
                    if p is paths[0] and (k & 1):
                        bases += len(ln) - 1
            sub.append(q)
        mat = oracle_lib.write_matrix(os.path.join(d, "params.mat"))
        t_split = t_poa = t_merge = 0.0
        rc = 1
        small = wrong = 0
        while rc != 0:
            t0 = time.perf_counter()
            rc = subprocess.run([os.path.join(ref_dir, "masterSplitter"), sub[0], sub[2], sub[1], d + "/out1", d + "/out2",
                                 d + "/out3", "7", "200", "10000", "0.1", d], stdout=subprocess.DEVNULL).returncode
            t_split += time.perf_counter() - t0
            small += int(open(d + "/small_reads.txt").readline())
            wrong += int(open(d + "/wrongly_cor_reads.txt").readline())
            t0 = time.perf_counter()
            with Pool(processes=cores) as pool:
                for _ in pool.imap_unordered(_poa_slot, [(ref_dir, d, i, mat) for i in range(200)]):
                    pass
            t_poa += time.perf_counter() - t0
            t0 = time.perf_counter()
            for i in range(200):
                subprocess.run([os.path.join(ref_dir, "Donatello"), d + "/smsa%d" % i, d + "/msa.fa"])
            t_merge += time.perf_counter() - t0
            for f in os.listdir(d):
                if f.startswith(("out1", "out2", "out3", "smsa")):
                    os.remove(os.path.join(d, f))
        t0 = time.perf_counter()
        txt = open(d + "/msa.fa").read()
        with redirect_stdout(io.StringIO()):
            stats_oracle.output_recall_precision(txt, small, wrong, 5, 0.1)
        t_stats = time.perf_counter() - t0
    out.update({"ref_bases": bases, "seconds": {"masterSplitter": round(t_split, 3), "poa x200 (Pool)": round(t_poa, 3),
                                                  "Donatello x200": round(t_merge, 3),
                                                  "statistics (oracle/stats_oracle.py, the port)": round(t_stats, 3)},
                "Mbases_per_s": round(bases / (t_split + t_poa + t_merge + t_stats) / 1e6, 4)})
    return out


def main(args=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--profile", default="ecoli30x_simlord_lordec")
    ap.add_argument("--reads", type=int, default=40004, help="reads in the three files (four batches of ELECTOR's protocol)")
    ap.add_argument("--reference-sample", type=int, default=1500, help="records the reference chain is timed on")
    ap.add_argument("--no-reference", action="store_true")
    ap.add_argument("--end-to-end", action="store_true")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--workdir", default=None, help="directory for the input files and msa.fa (default: the system's temporary directory)")
    ap.add_argument("--no-nomsa", action="store_true", help="skip the second run (getPOA(write_msa=False): no msa.fa)")
    ap.add_argument("--repeat", type=int, default=1, help="write the synthetic reads this many times over (distinct names): "
                                                          "--reads 40004 --repeat 5 is twenty batches")
    a, _ = ap.parse_known_args()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench_e2e.py needs a GPU: the HIP path has no CPU fallback")
    from elector_amd import alignment, computeStats, synthetic
    triples, headers, read_of = synthetic.read_pieces(a.profile, a.reads, seed=2000)
    bases = int(sum(len(t[0]) for t in triples)) * max(1, a.repeat)
    # the three input files and msa.fa (3.3 x the input).  Measured on the GPU box: the default temporary directory
    # (page cache of the container's overlay file system) takes the records at ~10 GB/s until write-back throttling sets
    # in; /dev/shm (tmpfs) at ~2.5 GB/s -- shared-memory pages are allocated and zeroed under one lock
    work = tempfile.mkdtemp(prefix="elector_e2e_", dir=a.workdir)
    try:
        paths = write_fasta(work, triples, headers, max(1, a.repeat))
        del triples
        # warm-up on the first batches, one per engine context and splitter thread: context creation, workspace
        # growth to the batch size, pinned buffers, first-launch costs
        n_warm = max(int(os.environ.get("ELECTOR_ENGINES", "3")), int(os.environ.get("ELECTOR_SPLITTERS", "2"))) + 1
        wdir = os.path.join(work, "warm")
        os.mkdir(wdir)
        small_paths = []
        for p in paths:
            q = os.path.join(wdir, os.path.basename(p))
            with open(p, "rb") as f, open(q, "wb") as g:
                for _ in range(2 * 10001 * n_warm):
                    g.write(f.readline())
            small_paths.append(q)
        with redirect_stdout(io.StringIO()):
            alignment.getPOA(small_paths[1], small_paths[0], small_paths[2], a.threads, wdir, 0.1)
        alignment.STAGE_SECONDS.clear()
        # the input files have only just been written and the warm-up left records of its own: their dirty pages
        # would count against the run's write-back budget (the last batches' records were throttled to a third of the
        # page cache's rate) -- a user's reads are on disk before ELECTOR starts
        shutil.rmtree(wdir, ignore_errors=True)
        os.sync()

        outdir = os.path.join(work, "out")
        os.mkdir(outdir)
        buf = io.StringIO()
        t0 = time.perf_counter()
        with redirect_stdout(buf):
            small, wrong = alignment.getPOA(paths[1], paths[0], paths[2], a.threads, outdir, 0.1)
        t1 = time.perf_counter()
        hit = computeStats.cached_pieces(outdir + "/msa.fa", {}) is not None
        log = io.StringIO()
        prof = None
        if os.environ.get("ELECTOR_E2E_PROFILE"):
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        with redirect_stdout(buf):
            tup = computeStats.outputRecallPrecision(paths[1], outdir, log, small, wrong, 5, 0.1, "sizes.txt", {})
        t2 = time.perf_counter()
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(18)
        stages = {k: round(v, 3) for k, v in alignment.STAGE_SECONDS.items()}

        def gantt(t_from, t_to, title):
            """a text Gantt chart of getPOA: one row per thread, one column per 10 ms"""
            if alignment.STAGE_TRACE is None:
                return
            tr = [x for x in alignment.STAGE_TRACE if t_from <= x[2] < t_to]
            names = sorted({x[1] for x in tr})
            width = int((t_to - t_from) / 0.01) + 1
            sys.stderr.write(title + "\n")
            for nm in names:
                row = [" "] * width
                for stage, th_, a_, b_ in tr:
                    if th_ != nm:
                        continue
                    ch = stage[0].upper() if not stage.startswith("wait") else "."
                    for c in range(int((a_ - t_from) / 0.01), min(width, int((b_ - t_from) / 0.01) + 1)):
                        row[c] = ch
                sys.stderr.write("%-12s|%s|\n" % (nm[:12], "".join(row)))

        gantt(t0, t1, "getPOA with msa.fa")
        # the same report from the text file (what call site #2 cost before the counters were handed over); left out
        # on long runs (the text of twenty batches is 5 GB)
        # ... and the run again without msa.fa (getPOA(write_msa=False)): same report from the device
        # counters, no records formatted or written
        nomsa = None
        if not a.no_nomsa:
            outdir2 = os.path.join(work, "out_nomsa")
            os.mkdir(outdir2)
            alignment.STAGE_SECONDS.clear()
            os.sync()
            tn0 = time.perf_counter()
            with redirect_stdout(buf):
                small2, wrong2 = alignment.getPOA(paths[1], paths[0], paths[2], a.threads, outdir2, 0.1, write_msa=False)
            tn1 = time.perf_counter()
            gantt(tn0, tn1, "getPOA(write_msa=False)")
            with redirect_stdout(buf):
                tupn = computeStats.outputRecallPrecision(paths[1], outdir2, io.StringIO(), small2, wrong2, 5, 0.1, "sizes.txt", {})
            tn2 = time.perf_counter()
            assert tupn == tup and not os.path.exists(outdir2 + "/msa.fa"), "the report without msa.fa differs"
            nomsa = {"value": round(bases / (tn2 - tn0) / 1e6, 3), "unit": "Mbases/s",
                     "seconds": {"getPOA (wall)": round(tn1 - tn0, 3), "outputRecallPrecision (wall)": round(tn2 - tn1, 3)},
                     "getPOA_stage_seconds": {k: round(v, 3) for k, v in alignment.STAGE_SECONDS.items()},
                     "report_equal_to_the_run_with_the_file": True}
        t3 = None
        if a.repeat <= 1 and a.reads <= 50000:
            alignment.MSA_CACHE.clear()
            with redirect_stdout(buf):
                tup2 = computeStats.outputRecallPrecision(paths[1], outdir, io.StringIO(), small, wrong, 5, 0.1, "sizes.txt", {})
            t3 = time.perf_counter()
            assert tup == tup2, "device counters and parsed msa.fa disagree"
        out = {
            "metric": "end-to-end Mbases/s: three FASTA files -> getPOA -> outputRecallPrecision (19-tuple)",
            "value": round(bases / (t2 - t0) / 1e6, 3), "unit": "Mbases/s", "n_gpus": 1,
            "config": {"workload": a.profile, "reads": a.reads * max(1, a.repeat), "triples": len(headers) * max(1, a.repeat), "ref_bases": bases,
                       "batches": (len(headers) * max(1, a.repeat) + 10000) // 10001,
                       "host_threads": a.threads, "msa_fa_bytes": os.path.getsize(outdir + "/msa.fa"),
                       "workdir": os.path.dirname(work)},
            "seconds": {"getPOA (wall)": round(t1 - t0, 3), "outputRecallPrecision (wall, device counters)": round(t2 - t1, 3),
                        "outputRecallPrecision from the text file instead": None if t3 is None else round(t3 - t2, 3)},
            "getPOA_stage_seconds": stages,
            "device_counters_used": hit,
            "without_msa_fa": nomsa,
            "recall": tup[3], "precision": tup[2], "assessed_reads": tup[0],
        }
        ref_dir = os.path.join(ROOT, "oracle", "_ref")
        if not a.no_reference and os.path.exists(os.path.join(ref_dir, "masterSplitter")):
            out["reference_chain"] = reference_chain(ref_dir, paths, a.reference_sample, os.cpu_count() or 1)
        print(json.dumps(out), flush=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
