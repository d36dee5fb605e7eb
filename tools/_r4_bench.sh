#!/bin/bash
# GPU-box helper (round 4): the statistics / bench tests, then the driver's command
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4bench}; mkdir -p $O
env | grep -i "HSA\|ROC\|HIP\|GPU_" > $O/env.txt; nproc >> $O/env.txt; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())" >> $O/env.txt
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -m pytest tests/test_stats_gpu.py tests/test_configs_gpu.py -x -q -m gpu --capture=sys > $O/pytest.log 2>&1 || { echo "PYTEST FAILED"; tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
fi
time timeout -k 10 900 python bench.py --steps ${STEPS:-20} --warmup 5 > $O/bench.json 2> $O/bench.err || { echo BENCH FAILED; tail -20 $O/bench.err; exit 2; }

python3 - $O/bench.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
print(j["config"]["profile"], "value", j["value"], "ms/step", j["ms_per_step"], "rows_in_hbm", j.get("value_rows_in_hbm"), "pcie", j["rows_to_host"], "checksum", j["counters_checksum"], "setup_s", j["setup_s"])
print("   ", {k:v for k,v in j["kernel_ms_per_step"].items() if k!="note"})
print("   cpu", j.get("cpu_baseline",{}).get("value"), "parity", j.get("parity_vs_reference"))
for c in j.get("configs",[]):
    print("  cfg", c["profile"], "value", c["value"], "ms", c["ms_per_step"], "hbm", c["value_rows_in_hbm"], "parity", c.get("parity_vs_reference",{}).get("differing"), "/", c.get("parity_vs_reference",{}).get("windows"))
PY
