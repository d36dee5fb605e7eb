#!/bin/bash
# GPU-box helper : end-to-end run under a few thread settings
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-e2e}; mkdir -p $O
run() { local tag=$1; shift
  env "$@" timeout -k 10 500 python bench.py --end-to-end --profile ${P:-ecoli30x_simlord_lordec} --repeat ${REPEAT:-5} --no-reference > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
n=j['without_msa_fa']
print('$tag', 'with file', j['value'], j['seconds']['getPOA (wall)'], '| without', n['value'], n['seconds']['getPOA (wall)'], n['seconds']['outputRecallPrecision (wall)'])"
}
run base A=1 && run s3 ELECTOR_SPLITTERS=3 && run s3e4 ELECTOR_SPLITTERS=3 ELECTOR_ENGINES=4 && run s4e4 ELECTOR_SPLITTERS=4 ELECTOR_ENGINES=4
