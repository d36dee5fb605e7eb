#!/bin/bash
# GPU-box helper: same-box A/B of two builds of the library (ELECTOR_LIB) on the pipelined bench line
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4mergeab}; mkdir -p $O
OLD=$PWD/elector_amd/lib/${OLDLIB:-libelector_poa_oldmerge.so}
run() { local tag=$1 prof=$2; shift; shift
  env "$@" timeout -k 10 400 python bench.py --profile $prof --steps 50 --no-cpu-baseline --no-configs > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
k=j['kernel_ms_per_step']
print('$tag', 'value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'merge+counters', k['merge_and_counters'], 'checksum', j['counters_checksum'])"
}
for P in ecoli30x_simlord_lordec yeast50x_nanosim_consent_split; do
  run new_$P $P A=1 && run old_$P $P ELECTOR_LIB=$OLD && run new2_$P $P A=1 && run old2_$P $P ELECTOR_LIB=$OLD || exit 2
done
