#!/bin/bash
# GPU-box helper (round 4): parity, then A/B of the list split (tail launch with the smaller LDS slot)
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4sl}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_poa_gpu.py -x -q -m gpu --capture=sys > $O/pytest.log 2>&1 || { echo "PYTEST FAILED"; tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
run() { # tag, profile, env...
  local tag=$1 prof=$2; shift; shift
  env "$@" timeout -k 10 400 python bench.py --profile $prof --batches 1 --steps 30 --no-cpu-baseline --serial-steps 8 --no-configs > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
k=j['kernel_ms_per_step']
print('$tag', 'value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'k_poa', k['k_poa'], 'far', k.get('k_poa_far_instance'), 'a1', k['alignment1_stage'], 'a2', k['alignment2_stage'], 'other', k['other'], 'checksum', j['counters_checksum'])"
}
for P in ecoli30x_simlord_lordec yeast50x_nanosim_consent_split celegans30x_simlord_mixed; do
  run split_$P $P A=1 && run whole_$P $P ELECTOR_POA_SPLIT=0 && run split2_$P $P A=1 && run whole2_$P $P ELECTOR_POA_SPLIT=0 || exit 2
done
