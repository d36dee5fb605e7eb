#!/bin/bash
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4minbin2}; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
run() { local tag=$1 prof=$2; shift; shift
  env "$@" timeout -k 10 400 python bench.py --profile $prof --steps 40 --no-cpu-baseline --no-configs > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
k=j['kernel_ms_per_step']
print('$tag', 'value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'k_poa', k['k_poa'], 'launches', j['roofline']['launches'], 'checksum', j['counters_checksum'])"
}
for P in chr1_20x_ont_50kb celegans30x_simlord_mixed; do
  for M in 16384 4096 16384 4096; do run mb${M}_$P $P ELECTOR_MIN_BIN=$M || exit 2; done
done
