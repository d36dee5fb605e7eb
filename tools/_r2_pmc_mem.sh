#!/bin/bash
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
# GPU-box helper: memory-pipeline counters of the serial bench (TA / TCP / TCC), one pass per group
TAG=${1:-r2mem}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
rocprofv3 -L > gpurun_out/$TAG/counters.txt 2>&1
CMD="python3 bench.py --serial --no-cpu-baseline --steps 2 --warmup 1 ${PMC_EXTRA}"
i=0
for grp in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/$TAG/p$i -- $CMD > gpurun_out/$TAG/p$i.log 2>&1 || echo "pass $i failed" >> gpurun_out/$TAG/fail.txt
done
python3 tools/_pmc_summary.py gpurun_out/$TAG > gpurun_out/$TAG/summary.json
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/$TAG -name "*counter_collection.csv" -size +20M -delete
python3 - gpurun_out/$TAG/summary.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k in ("k_poa",):
    for c,v in d.get(k,{}).items(): print(k,c,"%.4g"%v["total"],v["launches"])
PY
