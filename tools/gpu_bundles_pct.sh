#!/bin/bash
# GPU-box helper: bench.py --bundles under several ELECTOR_BUNDLE_GLOBAL_PCT (share of a class's blocks with their records in HBM;
# one number, or four for the LDS classes of 52 / 69 / 104 / 208 nodes).  Usage: gpu_bundles_pct.sh [TAG] ; PCTS="70 50,70,100,100" PROFS=...
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r5pct}; mkdir -p $O; export ELECTOR_BENCH_NO_FORK=1
for P in ${PROFS:-ecoli30x_simlord_lordec yeast50x_nanosim_consent_split}; do
for pct in ${PCTS:-70 30,50,100,100 100,100,70,50 100 70}; do
  ELECTOR_BUNDLE_GLOBAL_PCT=$pct timeout -k 10 200 python bench.py --bundles --profile $P --steps 10 > $O/${P}_$pct.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  python3 -c "
import json; j=json.load(open('$O/${P}_$pct.json')); p=j['pipelined']; print('$P pct $pct alone', j['value'], 'pipe', p['step_ms_without_search'], p['step_ms_with_search'], p['step_ms_with_search_from_helper_threads'], p['step_ms_with_search_queued_a_turn_later'])"
done; done
