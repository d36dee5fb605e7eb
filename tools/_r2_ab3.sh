#!/bin/bash
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
# GPU-box helper: A/B of two builds of the library (ab_libs/old.so, ab_libs/new.so) on one box, alternating
O=gpurun_out/${1:-r2ab3}; mkdir -p $O
for i in 1 2 3; do
for v in old new; do
  cp ab_libs/$v.so elector_amd/lib/libelector_poa.so
  python bench.py --steps 60 --no-cpu-baseline > $O/$v$i.json 2> $O/$v$i.err || { tail -3 $O/$v$i.err; exit 1; }
  python - $O/$v$i.json $v$i <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], j["value"], j["ms_per_step"], j["counters_checksum"], j["kernel_ms_per_step"]["k_poa"])
PY
done; done
