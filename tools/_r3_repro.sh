#!/bin/bash
# GPU-box helper: the default bench line, then rocprofv3 --kernel-trace --stats over `bench.py --serial` -- the
# line's per-kernel figures (its post-clock serial pass) must agree with the profile's averages
: ${GRAFT_REPO_ROOT:?}
set -o pipefail
O=gpurun_out/${1:-r3repro}
mkdir -p $O
python bench.py --steps ${STEPS:-20} --warmup 5 > $O/bench.json 2> $O/bench.err || exit 1
python bench.py --serial --steps 20 --no-cpu-baseline > $O/bench_serial.json 2> $O/bench_serial.err || exit 2
export TMPDIR=/tmp
R=$PWD
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_serial -o serial -- python3 $R/bench.py --serial --steps 20 --no-cpu-baseline > $R/$O/prof_serial.json 2> $R/$O/prof_serial.err ) || exit 3
python3 - $O <<'PY'
import csv, glob, json, sys
o = sys.argv[1]
line = json.load(open(o + "/bench.json"))
ser = json.load(open(o + "/bench_serial.json"))
f = glob.glob(o + "/prof_serial/**/*kernel_stats.csv", recursive=True)[0]
tot = {}
for r in csv.DictReader(open(f)):
    k = r["Name"].replace("void ", "").replace("elector::", "").split("(")[0].split("<")[0]
    t = tot.setdefault(k, [0, 0.0]); t[0] += int(r["Calls"]); t[1] += float(r["TotalDurationNs"])
kp = tot["k_poa"]
print("value", line["value"], "rows_to_host", line.get("value_rows_to_host"), line.get("rows_to_host"))
print("parity", line.get("parity_vs_reference"))
print("line   k_poa avg launch ms %.4f  per step %.3f" % (line["roofline"]["avg_launch_ms"], line["kernel_ms_per_step"]["k_poa"]))
print("serial k_poa avg launch ms %.4f  per step %.3f" % (ser["roofline"]["avg_launch_ms"], ser["kernel_ms_per_step"]["k_poa"]))
print("rocprof k_poa avg launch ms %.4f  calls %d" % (kp[1] / kp[0] / 1e6, kp[0]))
print("line kernel_ms", line["kernel_ms_per_step"])
print("serial kernel_ms", ser["kernel_ms_per_step"])
PY
find $O -name "*kernel_trace.csv" -delete
