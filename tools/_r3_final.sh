#!/bin/bash
# GPU-box helper: the round's bench lines (kept under profiles/ by hand): default command, every workload profile,
# the un-overlapped pass with rocprofv3 kernel statistics beside it
: ${GRAFT_REPO_ROOT:?}
set -o pipefail
O=gpurun_out/${1:-r3final}
mkdir -p $O
python bench.py > $O/bench_ecoli.json 2> $O/bench_ecoli.err || { tail -5 $O/bench_ecoli.err; exit 1; }
python3 -c "
import json; j=json.load(open('$O/bench_ecoli.json')); print('default', j['value'], j['ms_per_step'], j['value_rows_to_host'], j['roofline']['frac'], (j['roofline_valu'] or {}).get('frac'), j['cpu_baseline']['value'], j['parity_vs_reference'])"
for P in yeast50x_nanosim_consent yeast50x_nanosim_consent_split celegans30x_simlord_mixed chr1_20x_ont_50kb ecoli10x_c1; do
  EXTRA=""; [ "$P" = chr1_20x_ont_50kb ] && EXTRA="--reads 2000"; [ "$P" = ecoli10x_c1 ] && EXTRA="--reads 459"
  python bench.py --profile $P --steps 40 --no-cpu-baseline $EXTRA > $O/bench_$P.json 2> $O/bench_$P.err || { echo FAILED $P; tail -3 $O/bench_$P.err; exit 2; }
  python3 -c "
import json; j=json.load(open('$O/bench_$P.json')); print('$P', j['value'], j['ms_per_step'], j['value_rows_to_host'], {k:v for k,v in j['kernel_ms_per_step'].items() if k!='note'})"
done
python bench.py --serial --steps 20 --no-cpu-baseline > $O/bench_serial.json 2> $O/bench_serial.err || exit 5
export TMPDIR=/tmp
R=$PWD
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_serial -o serial -- python3 $R/bench.py --serial --steps 20 --no-cpu-baseline > $R/$O/prof_serial.json 2> $R/$O/prof_serial.err ) || exit 6
find $O -name "*kernel_trace.csv" -delete
python3 - $O <<'PY'
import csv, glob, json, sys
o = sys.argv[1]
line = json.load(open(o + "/bench_ecoli.json")); ser = json.load(open(o + "/bench_serial.json"))
f = glob.glob(o + "/prof_serial/**/*kernel_stats.csv", recursive=True)[0]
tot = {}
for r in csv.DictReader(open(f)):
    k = r["Name"].replace("void ", "").replace("elector::", "").split("(")[0].split("<")[0]
    t = tot.setdefault(k, [0, 0.0]); t[0] += int(r["Calls"]); t[1] += float(r["TotalDurationNs"])
kp = tot["k_poa"]
print("k_poa avg launch ms: line %.4f  --serial %.4f  rocprof %.4f" % (line["roofline"]["avg_launch_ms"], ser["roofline"]["avg_launch_ms"], kp[1] / kp[0] / 1e6))
PY
