#!/bin/bash
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
# GPU-box helper: VALU / SALU / LDS instruction counts of k_poa with the windows dropped after a phase
# (ELECTOR_DEBUG_FUSED bits 32 .. 256): the differences are the phases' instruction counts
TAG=${1:-r2insts}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
for D in 32 64 128 256 0; do
  DBG=$D timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/$TAG/d$D -- python3 tools/_r2_dbg.py ecoli30x_simlord_lordec 10001 > gpurun_out/$TAG/d$D.log 2>&1 || echo "run $D failed" >> gpurun_out/$TAG/fail.txt
  python3 - gpurun_out/$TAG/d$D $D <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_poa" in r["Kernel_Name"] and "pool_init" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
print("debug", sys.argv[2], {k: "%.4g" % v for k, v in sorted(agg.items())})
PY
done
find gpurun_out/$TAG -name "*.csv" -size +5M -delete
