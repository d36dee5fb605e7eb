#!/bin/bash
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4rows3}; mkdir -p $O
run() { local tag=$1; shift
  env "$@" timeout -k 10 400 python bench.py --profile ${P:-yeast50x_nanosim_consent_split} --batches 3 --steps 40 --no-cpu-baseline --serial-steps 2 > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
k=j['kernel_ms_per_step']
print('$tag', 'value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'pcie', j['rows_to_host']['pcie_gbs_per_gpu'], 'host', k['host_classify_and_enqueue'], 'wait', k['host_wait_for_results'])"
}
run full A=1 && run unc ELECTOR_ROWS_UNCACHED=1 && run full_b A=1 && run unc_b ELECTOR_ROWS_UNCACHED=1
