#!/bin/bash
# GPU-box helper: durations of the memory copies while the rows loop runs (rocprofv3 --memory-copy-trace)
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4copy}; mkdir -p $O; R=$PWD; export TMPDIR=/tmp ELECTOR_BENCH_NO_FORK=1
( cd /tmp && rocprofv3 --memory-copy-trace --stats --output-format csv -d $R/$O/prof -o rows -- python3 $R/bench.py --profile yeast50x_nanosim_consent_split --batches 1 --steps 30 --no-cpu-baseline --serial-steps 1 --no-rows-in-hbm > $R/$O/bench.json 2> $R/$O/bench.err ) || { tail -5 $O/bench.err; exit 1; }
python3 - $O <<'PY'
import csv, glob, sys, collections
o=sys.argv[1]
f=glob.glob(o+"/prof/**/*memory_copy_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print(rows[0].keys())
big=[r for r in rows if "Bytes" in r or True]
sz=collections.Counter()
durs=[]
for r in rows:
    try:
        d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    except Exception: continue
    durs.append((d, r.get("Direction"), r))
durs.sort(key=lambda x:-x[0])
for d,dr,r in durs[:12]: print(round(d,3), dr, {k:r[k] for k in r if k not in ("Start_Timestamp","End_Timestamp")})
import statistics
long=[d for d,dr,r in durs if d>2.0]
print("copies > 2 ms:", len(long), "median ms", statistics.median(long) if long else None)
PY
find $O -name "*memory_copy_trace.csv" -size +2M -delete
