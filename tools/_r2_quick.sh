#!/bin/bash
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
# GPU-box helper: serial bench lines of the given profiles (kernel ms per step), small step count
O=gpurun_out/${1:-r2q}; shift
mkdir -p $O
for P in "$@"; do
  RD=10001; [ $P = chr1_20x_ont_50kb ] && RD=2000
  python bench.py --serial --profile $P --reads $RD --steps 8 --warmup 2 --no-cpu-baseline > $O/$P.json 2> $O/$P.err || { tail -5 $O/$P.err; exit 1; }
  python - $O/$P.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["config"]["profile"], j["value"], "Mbases/s", j["ms_per_step"], "ms", {k:v for k,v in j["kernel_ms_per_step"].items() if k!="note"})
PY
done
