#!/bin/bash
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4queues}; mkdir -p $O
run() { local tag=$1; shift
  env "$@" timeout -k 10 400 python bench.py --profile ${P:-yeast50x_nanosim_consent_split} --batches 3 --steps 40 --no-cpu-baseline --serial-steps 2 > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
k=j['kernel_ms_per_step']
print('$tag', 'value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'host', k['host_classify_and_enqueue'], 'wait', k['host_wait_for_results'])"
}
run q16 GPU_MAX_HW_QUEUES=16 && run q12 GPU_MAX_HW_QUEUES=12 && run q8 GPU_MAX_HW_QUEUES=8 && run q20 GPU_MAX_HW_QUEUES=20 && run q12b GPU_MAX_HW_QUEUES=12
