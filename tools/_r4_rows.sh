#!/bin/bash
# GPU-box helper (round 4): what the rows' way to the host costs -- copy engine on / off, contexts in flight
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4rows}; mkdir -p $O
run() { # tag, env...
  local tag=$1; shift
  env "$@" timeout -k 10 400 python bench.py --profile ${P:-ecoli30x_simlord_lordec} --batches 3 --steps 40 --no-cpu-baseline --serial-steps 2 > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
k=j['kernel_ms_per_step']
print('$tag', 'value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'pcie', j['rows_to_host']['pcie_gbs_per_gpu'], 'host', k['host_classify_and_enqueue'], 'wait', k['host_wait_for_results'])"
}
run default A=1 && run sdma0 HSA_ENABLE_SDMA=0 && run eng6 ELECTOR_BENCH_ENGINES=6 && run eng3 ELECTOR_BENCH_ENGINES=3 && run default2 A=1
