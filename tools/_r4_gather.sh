#!/bin/bash
# GPU-box helper (round 4): k_poa's inputs of all lists in one launch against a launch per list
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4gather}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_poa_gpu.py -x -q -m gpu 2>&1 | tail -3
run() { local tag=$1 prof=$2; shift; shift
  env "$@" timeout -k 10 400 python bench.py --profile $prof --steps 40 --no-cpu-baseline --no-configs > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
k=j['kernel_ms_per_step']
print('$tag', 'value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'k_poa', k['k_poa'], 'other', k['other'], 'checksum', j['counters_checksum'])"
}
for P in ecoli30x_simlord_lordec yeast50x_nanosim_consent_split; do
  run one_$P $P ELECTOR_GATHER_ALL=1 && run perbin_$P $P A=1 && run one2_$P $P ELECTOR_GATHER_ALL=1 && run perbin2_$P $P A=1 || exit 2
done
