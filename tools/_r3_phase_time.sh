#!/bin/bash
# GPU-box helper: k_poa's time with the windows dropped after a phase (ELECTOR_DEBUG_FUSED bits 32: after staging,
# 64: after fusion #1, 128: after the alignment #2 DP, 256: after traceback #2, 0: whole kernel), un-overlapped
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r3phase}; P=${2:-ecoli30x_simlord_lordec}
mkdir -p $O
for D in 0 32 64 128 256; do
  ELECTOR_DEBUG_FUSED=$D timeout -k 10 300 python bench.py --profile $P --serial --steps 8 --warmup 2 --no-cpu-baseline > $O/d$D.json 2> $O/d$D.err || { echo FAILED $D; tail -3 $O/d$D.err; continue; }
  python3 -c "
import json
j=json.load(open('$O/d$D.json'))
print('debug $D', 'k_poa ms/step', j['kernel_ms_per_step']['k_poa'], 'launches', j['roofline']['launches'])"
done
