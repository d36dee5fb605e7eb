#!/bin/bash
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
# GPU-box helper: A/B of a debug bit of k_poa on one box, alternating (ELECTOR_DEBUG_FUSED=$2 against unset)
O=gpurun_out/${1:-r2ab2}; mkdir -p $O
for i in 1 2 3; do
for v in off on; do
  if [ $v = on ]; then export ELECTOR_DEBUG_FUSED=$2; else unset ELECTOR_DEBUG_FUSED; fi
  python bench.py --steps 60 --no-cpu-baseline > $O/$v$i.json 2> $O/$v$i.err || { tail -3 $O/$v$i.err; exit 1; }
  python - $O/$v$i.json $v$i <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], j["value"], j["ms_per_step"], j["counters_checksum"])
PY
done; done
