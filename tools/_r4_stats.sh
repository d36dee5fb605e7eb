#!/bin/bash
# GPU-box helper (round 4): k_stats with 1024 threads per read on long reads: parity tests, then A/B on chr1
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4stats}; mkdir -p $O
for T in 256 1024; do
  ELECTOR_STATS_THREADS=$T timeout -k 10 900 python -m pytest tests/test_stats_gpu.py tests/test_c1_chain.py -x -q -m gpu --capture=sys > $O/pytest_$T.log 2>&1 || { echo "PYTEST FAILED at $T threads"; tail -40 $O/pytest_$T.log; exit 1; }
  echo "threads $T: $(tail -1 $O/pytest_$T.log)"
done
run() { local tag=$1 prof=$2; shift; shift
  env "$@" timeout -k 10 400 python bench.py --profile $prof --batches 1 --steps 30 --no-cpu-baseline --serial-steps 4 > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
k=j['kernel_ms_per_step']
print('$tag', 'value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'merge+counters', k['merge_and_counters'], 'checksum', j['counters_checksum'])"
}
run t1024_chr1 chr1_20x_ont_50kb A=1 && run t256_chr1 chr1_20x_ont_50kb ELECTOR_STATS_THREADS=256 && run t512_chr1 chr1_20x_ont_50kb ELECTOR_STATS_THREADS=512 && run auto_ecoli ecoli30x_simlord_lordec A=1 && run t1024_ecoli ecoli30x_simlord_lordec ELECTOR_STATS_THREADS=1024
