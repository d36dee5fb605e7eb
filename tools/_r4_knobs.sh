#!/bin/bash
# GPU-box helper (round 4, last build): the launch-structure knobs once more, one run each between two runs of the default
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4knobs}; mkdir -p $O
run() { local tag=$1 prof=$2; shift; shift
  env "$@" timeout -k 10 400 python bench.py --profile $prof --steps 40 --no-cpu-baseline --no-configs > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
print('$tag', 'value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'])"
}
for P in yeast50x_nanosim_consent_split ecoli30x_simlord_lordec; do
  run base_$P $P A=1
  run chains3_$P $P ELECTOR_CHAINS=3
  run eng3_$P $P ELECTOR_BENCH_ENGINES=3
  run eng5_$P $P ELECTOR_BENCH_ENGINES=5
  run bygroup_$P $P ELECTOR_CHAINS_BY_GROUP=1
  run q24_$P $P GPU_MAX_HW_QUEUES=24
  run base2_$P $P A=1
done
