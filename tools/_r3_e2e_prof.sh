#!/bin/bash
# GPU-box helper: kernel statistics of the end-to-end run (which kernels fill the GPU's time when both call sites run)
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r3e2eprof}
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o e2e -- python3 $R/bench.py --end-to-end --no-reference --repeat ${REPEAT:-3} > $R/$O/e2e.json 2> $R/$O/e2e.err ) || exit 3
python3 - $O <<'PY'
import csv, glob, json, sys
o = sys.argv[1]
f = glob.glob(o + "/prof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
agg = {}
for r in rows:
    k = r["Name"].replace("void ", "").replace("elector::", "").split("(")[0].split("<")[0]
    a = agg.setdefault(k, [0, 0.0]); a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
j = json.load(open(o + "/e2e.json"))
print("e2e", j["value"], j["seconds"])
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:24]:
    print("  %-28s calls %6d total %9.1f ms  %5.1f%%" % (k, c, t / 1e6, 100 * t / tot))
print("  all kernels %.1f ms" % (tot / 1e6))
PY
find $O -name "*kernel_trace.csv" -delete
