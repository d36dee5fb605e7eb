#!/bin/bash
# GPU-box helper: which engine moves the merged rows to the host inside the timed region of the default
# command?  rocprofv3 --kernel-trace --memory-copy-trace over the default profile, the rows loop only.
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r5copy}; mkdir -p $O; R=$PWD; export TMPDIR=/tmp ELECTOR_BENCH_NO_FORK=1
( cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/prof -o rows -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --serial-steps 1 --no-rows-in-hbm > $R/$O/bench.json 2> $R/$O/bench.err ) || { tail -5 $O/bench.err; exit 1; }
python3 - $O <<'PY'
import csv, glob, sys, collections, json
o = sys.argv[1]
j = json.load(open(o + "/bench.json"))
print("value", j["value"], "ms/step", j["ms_per_step"], "rows bytes/step", j["rows_to_host"]["bytes_per_step_per_gpu"], "pcie GB/s", j["rows_to_host"]["pcie_gbs_per_gpu"])
f = glob.glob(o + "/prof/**/*memory_copy_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    by = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        k = r.get("Direction", "?")
        by[k][0] += 1; by[k][1] += d; by[k][2] = max(by[k][2], d)
    for k, (n, t, m) in sorted(by.items(), key=lambda x: -x[1][1]):
        print("copy-trace %-28s calls %6d total ms %9.2f max ms %7.3f" % (k, n, t, m))
else:
    print("no memory copy trace")
f = glob.glob(o + "/prof/**/*kernel_stats.csv", recursive=True)[0]
ks = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in ks)
for r in sorted(ks, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("kernel %-60s calls %6s total ms %9.2f avg us %9.1f %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
find $O -name "*_trace.csv" -size +1M -delete
