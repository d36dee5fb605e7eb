#!/bin/bash
# GPU-box helper: a12 with a share of the blocks through the GLOBAL form
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4bundleab}; mkdir -p $O
ELECTOR_BUNDLE_GLOBAL_PCT=50 timeout -k 10 600 python -m pytest tests/test_bundles_gpu.py -x -q -m gpu 2>&1 | tail -3
ELECTOR_BUNDLE_GLOBAL_PCT=100 timeout -k 10 600 python -m pytest tests/test_bundles_gpu.py -x -q -m gpu 2>&1 | tail -3
for G in 0 30 50 70 100; do
  ELECTOR_BUNDLE_GLOBAL_PCT=$G timeout -k 10 300 python bench.py --bundles --steps 10 > $O/g$G.json 2> $O/g$G.err || { echo FAILED $G; tail -3 $O/g$G.err; exit 1; }
  python3 -c "
import json; j=json.load(open('$O/g$G.json')); print('global pct $G', j['value'], 'ms/step')"
done
