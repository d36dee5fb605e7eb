#!/bin/bash
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
# GPU-box helper: address-translation counters of k_poa (serial bench)
TAG=${1:-r2tlb}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum --output-format csv -d gpurun_out/$TAG/p1 -- python3 bench.py --serial --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/$TAG/p1.log 2>&1 || echo failed >> gpurun_out/$TAG/fail.txt
python3 - gpurun_out/$TAG/p1 <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_poa<" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
print({k: "%.4g" % v for k, v in sorted(agg.items())})
PY
find gpurun_out/$TAG -name "*.csv" -size +5M -delete
