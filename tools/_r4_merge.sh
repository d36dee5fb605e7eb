#!/bin/bash
# GPU-box helper (round 4): the window-parallel merge against the per-piece one: parity tests, then A/B on chr1 / E. coli / yeast -split
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4merge}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_stats_gpu.py tests/test_c1_chain.py tests/test_pipeline_gpu.py -x -q -m gpu --capture=sys > $O/pytest.log 2>&1 || { echo "PYTEST FAILED"; tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run() { local tag=$1 prof=$2; shift; shift
  env "$@" timeout -k 10 400 python bench.py --profile $prof --batches 1 --steps 30 --no-cpu-baseline --serial-steps 4 > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
k=j['kernel_ms_per_step']
print('$tag', 'value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'merge+counters', k['merge_and_counters'], 'checksum', j['counters_checksum'])"
}
for P in chr1_20x_ont_50kb ecoli30x_simlord_lordec yeast50x_nanosim_consent_split; do
  run win_$P $P A=1 && run piece_$P $P ELECTOR_MERGE_PER_PIECE=1 || exit 2
done
