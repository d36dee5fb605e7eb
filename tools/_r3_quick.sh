#!/bin/bash
# GPU-box helper: bench lines of the given profiles, compact
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r3q}; shift
mkdir -p $O
for P in "$@"; do
  EXTRA=""; [ "$P" = chr1_20x_ont_50kb ] && EXTRA="--reads 2000"
  timeout -k 10 300 python bench.py --profile $P --steps ${STEPS:-30} --no-cpu-baseline $EXTRA > $O/bench_$P.json 2> $O/bench_$P.err || { echo "FAILED $P"; tail -5 $O/bench_$P.err; exit 1; }
  python3 - $O/bench_$P.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
print(j["config"]["profile"], "value", j["value"], "ms/step", j["ms_per_step"], "rows_to_host", j.get("value_rows_to_host"), "checksum", j["counters_checksum"])
print("   ", {k:v for k,v in j["kernel_ms_per_step"].items() if k!="note"})
PY
done
