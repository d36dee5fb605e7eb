#!/bin/bash
# GPU-box helper: un-overlapped kernel table of one profile (rocprofv3 --kernel-trace --stats over bench.py --serial)
: ${GRAFT_REPO_ROOT:?}
P=${1:?profile}; O=gpurun_out/${2:-ks_$P}
EXTRA=""; [ "$P" = chr1_20x_ont_50kb ] && EXTRA="--reads 2000"
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o serial -- python3 $R/bench.py --serial --steps ${STEPS:-10} --no-cpu-baseline --profile $P $EXTRA > $R/$O/bench.json 2> $R/$O/bench.err ) || exit 3
python3 - $O <<'PY'
import csv, glob, json, sys
o = sys.argv[1]
f = glob.glob(o + "/prof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
j = json.load(open(o + "/bench.json"))
steps = j["steps"] + j["warmup"] + 3
tot = sum(float(r["TotalDurationNs"]) for r in rows)
agg = {}
for r in rows:
    k = r["Name"].replace("void ", "").replace("elector::", "").split("(")[0]
    k = k.split("<")[0] if k.startswith(("k_poa", "k_fused")) else k[:40]
    a = agg.setdefault(k, [0, 0.0]); a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
print(j["config"]["profile"], "value", j["value"], j["kernel_ms_per_step"])
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:16]:
    print("  %-42s calls/step %6.1f  ms/step %7.3f  %5.1f%%" % (k, c / steps, t / 1e6 / steps, 100 * t / tot))
PY
find $O -name "*kernel_trace.csv" -delete
