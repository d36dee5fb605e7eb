#!/bin/bash
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
# GPU-box helper: per-kernel totals and the longest dispatches of one serial bench run
P=${1:-yeast50x_nanosim_consent_split}; RD=${2:-10001}; O=gpurun_out/${3:-r2t}
mkdir -p $O; export TMPDIR=/tmp; R=$PWD
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o t -- python3 $R/bench.py --serial --profile $P --reads $RD --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/bench.json 2> $R/$O/bench.err ) || exit 1
python3 - $O <<'PY'
import csv,sys,re,collections,glob
o=sys.argv[1]
f=glob.glob(o+'/prof/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda:[0,0.0,0.0])
for r in rows:
    name=re.sub(r'\(.*','',r['Kernel_Name'].replace('void elector::','').replace('elector::',''))
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    a=agg[name]; a[0]+=1; a[1]+=d; a[2]=max(a[2],d)
tot=sum(v[1] for v in agg.values())
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:18]:
    print("%-40s calls %5d total %9.1f us %5.1f%% max %8.1f us"%(k[:40],v[0],v[1],100*v[1]/tot,v[2]))
PY
rm -rf $O/prof
