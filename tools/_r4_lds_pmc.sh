#!/bin/bash
# GPU-box helper: LDS counters of k_poa with the bank-friendly slot sizes on / off
: ${GRAFT_REPO_ROOT:?}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; export ELECTOR_BENCH_NO_FORK=1
for B in 1 0; do
  O=gpurun_out/r4lds_$B; mkdir -p $O
  ELECTOR_POA_SLOT_BANKS=$B timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/p4 -- python3 bench.py --serial --no-cpu-baseline --steps 3 --warmup 1 --profile ecoli30x_simlord_lordec > $O/p4.log 2>&1 || exit 2
  python3 tools/_pmc_summary.py $O > $O/summary.json
  python3 -c "
import json
d=json.load(open('$O/summary.json'))['k_poa']
print('banks=$B', {k:int(v['total']) for k,v in d.items()})"
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
done
