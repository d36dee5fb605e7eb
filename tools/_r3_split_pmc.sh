#!/bin/bash
# GPU-box helper: SQ counters of k_split on the bench batch (two rocprofv3 --pmc runs, --kernel-trace only beside them)
: ${GRAFT_REPO_ROOT:?}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3splitpmc}
mkdir -p $O
CMD="python3 tools/_r3_split.py ecoli30x_simlord_lordec 10001"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p3 -- $CMD > $O/p3.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d $O/p4 -- $CMD > $O/p4.log 2>&1 || exit 2
python3 tools/_pmc_summary.py $O > $O/summary.json
python3 - $O/summary.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k in d:
    if k.startswith("k_split"):
        v=d[k]; n=v["SQ_WAVES"]["launches"]
        print(k, "launches", n, {c: "%.4g"%(x["total"]/n) for c,x in v.items()})
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
