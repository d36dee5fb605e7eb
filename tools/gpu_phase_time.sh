#!/bin/bash
# GPU-box helper: k_poa's time with the windows dropped after a phase (ELECTOR_DEBUG_FUSED bits 32: after staging,
# 64: after alignment #1 / fusion #1, 128: after the alignment #2 DP, 256: after traceback #2, 0: whole kernel), un-overlapped,
# on a library built with -DELECTOR_POA_DEBUG=1 (ELECTOR_LIB).  Usage: gpu_phase_time.sh TAG LIB profile...
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r5phase}; LIB=${2:?debug library}; shift; shift
mkdir -p $O
for P in ${@:-ecoli30x_simlord_lordec yeast50x_nanosim_consent_split}; do
  for D in 0 32 64 128 256; do
    ELECTOR_LIB=$PWD/$LIB ELECTOR_DEBUG_FUSED=$D timeout -k 10 300 python bench.py --profile $P --serial --steps 6 --warmup 2 --no-cpu-baseline > $O/${P}_d$D.json 2> $O/${P}_d$D.err || { echo FAILED $D; tail -3 $O/${P}_d$D.err; continue; }
    python3 -c "
import json
j=json.load(open('$O/${P}_d$D.json'))
print('$P', 'debug $D', 'k_poa ms/step', j['kernel_ms_per_step']['k_poa'], 'launches', j['roofline']['launches'])"
  done
done
