#!/bin/bash
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
# GPU-box helper: SQ counters of the serial bench (per kernel family), two passes; plus FETCH/WRITE passes with $2=traffic
TAG=${1:-r2pmc}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
CMD="python3 bench.py --serial --no-cpu-baseline --steps 3 --warmup 1 ${PMC_EXTRA}"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/$TAG/p3 -- $CMD > gpurun_out/$TAG/p3.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/$TAG/p4 -- $CMD > gpurun_out/$TAG/p4.log 2>&1 || exit 2
if [ "$2" = traffic ]; then
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/$TAG/p5 -- $CMD > gpurun_out/$TAG/p5.log 2>&1 || exit 3
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/$TAG/p6 -- $CMD > gpurun_out/$TAG/p6.log 2>&1 || exit 4
fi
python3 tools/_pmc_summary.py gpurun_out/$TAG > gpurun_out/$TAG/summary.json
python3 - gpurun_out/$TAG/summary.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k in ("k_poa","k_fused_a","k_fused_b","k_stats","k_merge","k_trivial"):
    if k in d:
        v=d[k]; print(k, {c: "%.4g"%x["total"] for c,x in v.items()}, "launches", v["SQ_WAVES"]["launches"])
PY
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/$TAG -name "*counter_collection.csv" -size +20M -delete
