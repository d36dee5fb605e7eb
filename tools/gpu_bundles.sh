#!/bin/bash
# GPU-box helper: a12's kernels -- ms per step whole / up to the graph records / up to the first pass's DP, and the kernel table
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-bundles}; mkdir -p $O
export TMPDIR=/tmp ELECTOR_BENCH_NO_FORK=1
R=$PWD
for D in 0 1 2; do
  ELECTOR_DEBUG_BUNDLE=$D timeout -k 10 300 python bench.py --bundles --steps 10 > $O/b$D.json 2> $O/b$D.err || { echo FAILED $D; tail -3 $O/b$D.err; exit 1; }
  python3 -c "
import json; j=json.load(open('$O/b$D.json')); print('debug $D', j['value'], 'ms/step', j['roofline']['launches'], 'launches')"
done
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o b -- python3 $R/bench.py --bundles --steps 10 > $R/$O/prof.json 2> $R/$O/prof.err ) || exit 3
python3 - $O <<'PY'
import csv, glob, sys
o = sys.argv[1]
f = glob.glob(o + "/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "bundle" in r["Name"] or "cons" in r["Name"]:
        print("  %-60s calls %5s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $O -name "*kernel_trace.csv" -delete
