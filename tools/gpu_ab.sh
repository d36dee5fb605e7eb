#!/bin/bash
# GPU-box helper: same-box A/B of library builds on the un-overlapped pass (bench.py --serial): k_poa's time per step.
# Usage: gpu_ab.sh TAG LIB_A[,LIB_B...] [profiles...]   (LIB = path under the repo, or "cur" for the built library)
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:?tag}; LIBS=${2:-cur}; shift; shift
PROFS=${@:-ecoli30x_simlord_lordec yeast50x_nanosim_consent_split}
mkdir -p $O
for P in $PROFS; do
  for rep in 1 2; do
    for L in ${LIBS//,/ }; do
      tag=$(basename $L .so)_${P}_$rep
      if [ $L = cur ]; then E="A=1"; else E="ELECTOR_LIB=$PWD/$L"; fi
      env $E timeout -k 10 300 python bench.py --serial --steps ${STEPS:-10} --no-cpu-baseline --profile $P > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; exit 2; }
      python3 -c "
import json
j=json.load(open('$O/$tag.json'))
k=j['kernel_ms_per_step']
print('$tag', 'k_poa', k['k_poa'], 'far', k.get('k_poa_far'), 'hb', round(k['alignment1_stage']+k['alignment2_stage'],3), 'other', k['other'], 'step', j['ms_per_step'], 'checksum', j['counters_checksum'])"
    done
  done
done
