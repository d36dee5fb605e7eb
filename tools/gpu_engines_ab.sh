#!/bin/bash
# GPU-box helper: the driver's command with 2 / 3 / 4 / 5 batches in flight (ELECTOR_BENCH_ENGINES): whole region, steady state,
# completion times.  Usage: gpu_engines_ab.sh TAG [profile] [list of counts]
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-engines}; P=${2:-yeast50x_nanosim_consent_split}; mkdir -p $O
for E in ${3:-2 3 4 5}; do
  ELECTOR_BENCH_ENGINES=$E python bench.py --profile $P --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_$E.json 2> $O/bench_$E.err || { tail -5 $O/bench_$E.err; exit 1; }
  python3 -c "
import json
j=json.load(open('$O/bench_$E.json'))
print('engines $E value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'steady', j['steady_state']['ms_per_step'])
print('   completions', j['steady_state']['step_completions_ms'], 'end', round(j['ms_per_step']*j['steps'],2))"
done
