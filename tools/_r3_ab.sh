#!/bin/bash
# GPU-box helper: same-box A/B of an environment knob on the bench batch: _r3_ab.sh TAG "ENV=1" [profile]
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r3ab}; KNOB=$2; P=${3:-ecoli30x_simlord_lordec}
mkdir -p $O
for round in 1 2; do
  for side in base knob; do
    if [ $side = knob ]; then export $KNOB; else unset ${KNOB%%=*}; fi
    timeout -k 10 300 python bench.py --profile $P --steps ${STEPS:-40} --no-cpu-baseline --no-rows-to-host --serial-steps 4 > $O/${side}_$round.json 2> $O/${side}_$round.err || { echo FAILED $side; tail -3 $O/${side}_$round.err; exit 1; }
    python3 -c "
import json,sys
j=json.load(open('$O/${side}_$round.json'))
print('$side $round', 'value', j['value'], 'ms/step', j['ms_per_step'], 'k_poa serial', j['kernel_ms_per_step']['k_poa'], 'checksum', j['counters_checksum'])"
  done
done
