#!/bin/bash
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
# GPU-box helper: default bench under a few engine-context / launch-chain settings (A/B on one box)
O=gpurun_out/${1:-r2ab}; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --steps 50 --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err || { tail -3 $O/$tag.err; return; }
  python - $O/$tag.json $tag <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], j["value"], j["ms_per_step"])
PY
}
run e4c2 ELECTOR_BENCH_ENGINES=4
run e3c2 ELECTOR_BENCH_ENGINES=3
run e5c2 ELECTOR_BENCH_ENGINES=5
run e4c3 ELECTOR_BENCH_ENGINES=4 ELECTOR_CHAINS=3
run e3c3 ELECTOR_BENCH_ENGINES=3 ELECTOR_CHAINS=3
run e6c1 ELECTOR_BENCH_ENGINES=6 ELECTOR_CHAINS=1
run e4c2b ELECTOR_BENCH_ENGINES=4
