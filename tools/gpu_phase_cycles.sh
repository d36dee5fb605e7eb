#!/bin/bash
# GPU-box helper: k_poa's cycles per wavefront and phase, per launch class (ELECTOR_DEBUG_FUSED=4 on a library built with
# -DELECTOR_POA_DEBUG=1): one un-overlapped batch.  Usage: gpu_phase_cycles.sh TAG LIB [profile]
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-phasecycles}; LIB=${2:?debug library}; P=${3:-yeast50x_nanosim_consent_split}; mkdir -p $O
ELECTOR_LIB=$PWD/$LIB ELECTOR_DEBUG_FUSED=${DBG:-4} ELECTOR_BENCH_NO_FORK=1 timeout -k 10 300 python bench.py --profile $P --serial --batches 1 --steps 1 --warmup 0 --serial-steps 1 --no-cpu-baseline --no-configs --no-rows-in-hbm > $O/debug_$P.json 2> $O/debug_$P.err
grep -a "k_poa waves" $O/debug_$P.err | sort | uniq | tail -24 | cut -c1-420
