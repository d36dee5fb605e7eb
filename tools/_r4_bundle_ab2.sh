#!/bin/bash
# GPU-box helper: a12, the share of each class's blocks that goes through the GLOBAL form (52 / 69 / 104 / 208 nodes)
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4bundleab2}; mkdir -p $O
for G in 50,50,50,50 50,50,70,100 40,50,70,100 30,40,60,100 60,60,80,100 50,50,100,100 50,50,50,50; do
  ELECTOR_BUNDLE_GLOBAL_PCT=$G timeout -k 10 300 python bench.py --bundles --steps 10 > $O/g.json 2> $O/g.err || { echo FAILED $G; tail -3 $O/g.err; exit 1; }
  python3 -c "
import json; j=json.load(open('$O/g.json')); print('global pct $G', j['value'], 'ms/step')"
done
