#!/bin/bash
# GPU-box helper (round 4): how small a geometry class may be before it joins the next one of its lane-group size (ELECTOR_MIN_BIN, windows)
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4minbin}; mkdir -p $O
run() { # tag, profile, env...
  local tag=$1 prof=$2; shift; shift
  env "$@" timeout -k 10 400 python bench.py --profile $prof --steps 40 --no-cpu-baseline --no-configs > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
k=j['kernel_ms_per_step']
print('$tag', 'value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'k_poa', k['k_poa'], 'launches', j['roofline']['launches'], 'checksum', j['counters_checksum'])"
}
for P in ecoli30x_simlord_lordec yeast50x_nanosim_consent_split; do
  for M in 4096 16384 65536 4096 16384 65536; do run mb${M}_$P $P ELECTOR_MIN_BIN=$M || exit 2; done
done
