#!/bin/bash
# GPU-box helper: what the fill and the drain of the timed region are made of: the completion times of the driver's 20 steps
# (bench.py `steady_state.step_completions_ms`), and the windows the alignment #2 of the two-kernel path still owes per batch
# (ELECTOR_DEBUG_BINS: one un-overlapped batch).
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-filldrain}; P=${2:-yeast50x_nanosim_consent_split}; mkdir -p $O
python bench.py --profile $P --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "
import json
j=json.load(open('$O/bench.json'))
print('value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'steady', j['steady_state']['ms_per_step'])
print('completions', j['steady_state']['step_completions_ms'], 'end', round(j['ms_per_step']*j['steps'],2))"
ELECTOR_DEBUG_BINS=1 ELECTOR_BENCH_NO_FORK=1 python bench.py --profile $P --serial --batches 1 --steps 1 --warmup 0 --serial-steps 1 --no-cpu-baseline --no-configs --no-rows-in-hbm > $O/debug.json 2> $O/debug.err
grep -a "round 0\|handed back\|generic=" $O/debug.err | sort | uniq -c | sort -rn | head -12
