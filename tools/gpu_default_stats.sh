#!/bin/bash
# GPU-box helper: rocprofv3 --kernel-trace --stats over the driver's command (profiles/rNN_default_kernel_stats.csv)
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-defstats}; mkdir -p $O; R=$PWD; export TMPDIR=/tmp ELECTOR_BENCH_NO_FORK=1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o default -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $R/$O/bench_under_rocprof.json 2> $R/$O/bench.err ) || { tail -5 $O/bench.err; exit 1; }
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/default_kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -delete
python3 - $O <<'PY'
import csv, json, sys
o = sys.argv[1]
j = json.load(open(o + "/bench_under_rocprof.json"))
rows = list(csv.DictReader(open(o + "/default_kernel_stats.csv")))
kp = [r for r in rows if "k_poa<" in r["Name"]]
calls = sum(int(r["Calls"]) for r in kp); tot = sum(float(r["TotalDurationNs"]) for r in kp)
print("under rocprofv3: value", j["value"], "ms/step", j["ms_per_step"], "| k_poa launches", calls, "avg launch ms %.4f" % (tot / calls / 1e6), "| bench's own:", j["roofline"]["avg_launch_ms"], "over", j["roofline"]["launches"])
PY
