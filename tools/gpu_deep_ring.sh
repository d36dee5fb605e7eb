#!/bin/bash
# GPU-box helper: the un-overlapped step with the DEEP alignment #2's LDS ring at several depths (ELECTOR_DEEP_RING)
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-deepring}; P=${2:-yeast50x_nanosim_consent_split}; mkdir -p $O
for D in ${3:-32 128 256 512}; do
  ELECTOR_DEEP_RING=$D python bench.py --profile $P --serial --steps 12 --warmup 3 --no-cpu-baseline --no-configs --no-rows-in-hbm > $O/serial_$D.json 2> $O/serial_$D.err || { tail -5 $O/serial_$D.err; exit 1; }
  python3 -c "
import json
j=json.load(open('$O/serial_$D.json')); k=j['kernel_ms_per_step']
print('ring $D', 'other', k['other'], 'k_poa', k['k_poa'], 'wall', k['serial_step_wall'], 'checksum', j['counters_checksum'])"
done
