#!/bin/bash
# GPU-box helper: rocprofv3 --kernel-trace --memory-copy-trace --stats over the end-to-end run: kernel table per batch, copies by direction
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-e2estats}; P=${2:-ecoli30x_simlord_lordec}; mkdir -p $O; R=$PWD; export TMPDIR=/tmp ELECTOR_BENCH_NO_FORK=1
( cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/prof -o e2e -- python3 $R/bench.py --end-to-end --profile $P --repeat ${REPEAT:-5} --no-reference > $R/$O/e2e.json 2> $R/$O/e2e.err ) || { tail -5 $O/e2e.err; exit 1; }
python3 - $O <<'PY'
import csv, glob, json, sys
o = sys.argv[1]
j = json.load(open(o + "/e2e.json"))
print("under rocprofv3: with file", j["value"], "without", j["without_msa_fa"]["value"])
rows = list(csv.DictReader(open(glob.glob(o + "/prof/**/*kernel_stats.csv", recursive=True)[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
nb = max(int(r["Calls"]) for r in rows if "k_classify" in r["Name"])
print("batches (k_classify calls)", nb, "kernel ms per batch %.1f" % (tot / 1e6 / nb))
agg = {}
for r in rows:
    k = r["Name"].replace("void ", "").replace("elector::", "").split("(")[0]
    k = k.split("<")[0] if k.startswith(("k_poa", "k_fused", "k_split", "k_dp", "k_bundle")) else k[:40]
    a = agg.setdefault(k, [0, 0.0]); a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
    print("  %-42s calls/batch %6.1f  ms/batch %7.3f  %5.1f%%" % (k, c / nb, t / 1e6 / nb, 100 * t / tot))
# the blit kernels by size class (grid size x workgroup size ~ bytes moved) and duration
kt = csv.DictReader(open(glob.glob(o + "/prof/**/*kernel_trace.csv", recursive=True)[0]))
cls = {}
for r in kt:
    if "copyBuffer" not in r["Kernel_Name"]: continue
    g = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    key = "grid %9d" % g if g < 1 << 14 else "grid >= 16k (%d)" % (g >> 14 << 14)
    a = cls.setdefault("< 0.05 ms" if d < 0.05 else "0.05-0.5 ms" if d < 0.5 else "0.5-2 ms" if d < 2 else "2-5 ms" if d < 5 else ">= 5 ms", [0, 0.0, set()])
    a[0] += 1; a[1] += d; a[2].add(g)
for k, (c, t, gs) in sorted(cls.items()):
    print("  copyBuffer %-12s calls/batch %5.1f ms/batch %6.2f  grids %s" % (k, c / nb, t / nb, sorted(gs)[:6]))
mc = list(csv.DictReader(open(glob.glob(o + "/prof/**/*memory_copy_trace.csv", recursive=True)[0])))
d = {}
for r in mc:
    a = d.setdefault(r["Direction"], [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, (c, t) in d.items(): print("  copy engine", k, "calls/batch %.1f ms/batch %.2f" % (c / nb, t / nb))
PY
find $O/prof -name "*_trace.csv" -size +1M -delete
