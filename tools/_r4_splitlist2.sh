#!/bin/bash
# GPU-box helper (round 4): A/B of the list split with the bench's own settings (three rotating batches)
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4sl3}; mkdir -p $O
run() { # tag, profile, env...
  local tag=$1 prof=$2; shift; shift
  env "$@" timeout -k 10 400 python bench.py --profile $prof --steps 40 --no-cpu-baseline --no-configs > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
k=j['kernel_ms_per_step']
print('$tag', 'value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'k_poa', k['k_poa'], 'checksum', j['counters_checksum'])"
}
for P in yeast50x_nanosim_consent_split ecoli30x_simlord_lordec chr1_20x_ont_50kb; do
  run split_$P $P A=1 && run whole_$P $P ELECTOR_POA_SPLIT=0 && run split2_$P $P A=1 && run whole2_$P $P ELECTOR_POA_SPLIT=0 || exit 2
done
