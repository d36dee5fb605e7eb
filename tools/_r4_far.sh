#!/bin/bash
# GPU-box helper (round 4): parity of the far-edge instance, then A/B of it on the split / mixed / 50 kb profiles
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4far}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_poa_gpu.py -x -q -m gpu --capture=sys > $O/pytest.log 2>&1 || { echo "PYTEST FAILED"; tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
run() { # tag, profile, env...
  local tag=$1 prof=$2; shift; shift
  env "$@" timeout -k 10 400 python bench.py --profile $prof --batches 1 --steps 30 --no-cpu-baseline --serial-steps 4 > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
k=j['kernel_ms_per_step']
print('$tag', 'value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'k_poa', k['k_poa'], 'a1', k['alignment1_stage'], 'a2', k['alignment2_stage'], 'other', k['other'], 'host', k['host_classify_and_enqueue'], 'wait', k['host_wait_for_results'], 'checksum', j['counters_checksum'])"
}
for P in yeast50x_nanosim_consent_split celegans30x_simlord_mixed chr1_20x_ont_50kb ecoli30x_simlord_lordec; do
  run far_$P $P A=1 && run nofar_$P $P ELECTOR_NO_FAR=1 || exit 2
done
