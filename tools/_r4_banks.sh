#!/bin/bash
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4banks}; mkdir -p $O
run() { local tag=$1 prof=$2; shift; shift
  env "$@" timeout -k 10 400 python bench.py --profile $prof --batches 1 --steps 40 --no-cpu-baseline --serial-steps 6 > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
k=j['kernel_ms_per_step']
print('$tag', 'value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'k_poa', k['k_poa'], 'checksum', j['counters_checksum'])"
}
for r in 1 2; do for P in ecoli30x_simlord_lordec yeast50x_nanosim_consent_split; do run on_${P}_$r $P A=1 && run off_${P}_$r $P ELECTOR_POA_SLOT_BANKS=0 || exit 1; done; done
