#!/bin/bash
# GPU-box helper: the timeline of ONE batch alone on the chip (one engine context, its two launch chains): which kernels make
# up a batch's latency -- what the fill and the drain of the timed region are made of.  rocprofv3 --kernel-trace
# --memory-copy-trace over bench.py with ELECTOR_BENCH_ENGINES=1.
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-timeline}; P=${2:-yeast50x_nanosim_consent_split}; mkdir -p $O; R=$PWD; export TMPDIR=/tmp ELECTOR_BENCH_NO_FORK=1 ELECTOR_BENCH_ENGINES=1
( cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/prof -o one -- python3 $R/bench.py --profile $P --batches 1 --steps 4 --warmup 2 --no-cpu-baseline --no-configs --serial-steps 1 --no-rows-in-hbm > $R/$O/bench.json 2> $R/$O/bench.err ) || { tail -5 $O/bench.err; exit 1; }
python3 - $O <<'PY'
import csv, glob, sys, re, json
o = sys.argv[1]
j = json.load(open(o + "/bench.json"))
print("ms/step (one context)", j["ms_per_step"])
kt = list(csv.DictReader(open(glob.glob(o + "/prof/**/*kernel_trace.csv", recursive=True)[0])))
mc = list(csv.DictReader(open(glob.glob(o + "/prof/**/*memory_copy_trace.csv", recursive=True)[0])))
ev = []
for r in kt:
    n = r["Kernel_Name"].replace("void ", "").replace("elector::", "").split("(")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, "q" + r["Queue_Id"]))
for r in mc:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", "copy "), "dma"))
ev.sort()
# the last batch: from the last k_classify before the end
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_classify")]
i0 = starts[-2] if len(starts) >= 2 else starts[-1]       # the last TIMED batch (the serial pass follows it)
i1 = starts[-1] if len(starts) >= 2 else len(ev)
t0 = ev[i0][0]
rows = ev[i0:i1]
tend = max(e[1] for e in rows)
print("batch: %d launches, first start to last end %.3f ms" % (len(rows), (tend - t0) / 1e6))
for s, e, n, q in rows:
    d = (e - s) / 1e6
    if d >= 0.15:
        print("  %8.3f .. %8.3f  %7.3f ms  %-4s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, d, q, n[:70]))
PY
find $O -name "*_trace.csv" -size +1M -delete
