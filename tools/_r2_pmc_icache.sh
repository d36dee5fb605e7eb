#!/bin/bash
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
# GPU-box helper: instruction-cache and branch counters of the alignment kernels, pipelined and serial bench
TAG=${1:-r2ic}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
for MODE in "" "--serial"; do
  D=gpurun_out/$TAG/p${MODE#--}
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $D -- python3 bench.py $MODE --no-cpu-baseline --steps 3 --warmup 1 > $D.log 2>&1 || echo "pass $MODE failed" >> gpurun_out/$TAG/fail.txt
  python3 - $D "$MODE" <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_poa<" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
print("mode", sys.argv[2] or "pipelined", {k: "%.4g" % v for k, v in sorted(agg.items())})
PY
done
find gpurun_out/$TAG -name "*.csv" -size +5M -delete
