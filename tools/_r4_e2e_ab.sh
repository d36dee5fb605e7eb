#!/bin/bash
# end-to-end run under a few pipeline settings (splitter threads, engine contexts)
O=gpurun_out/${1:-r4e2eab}; mkdir -p $O
run() { # tag env...
  local tag=$1; shift
  env "$@" ELECTOR_STAGE_TRACE=1 timeout -k 10 500 python bench.py --end-to-end --profile ${PROFILE:-ecoli30x_simlord_lordec} --repeat ${REPEAT:-5} --no-reference > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
n=j['without_msa_fa']
print('$tag', 'with file', j['value'], j['seconds']['getPOA (wall)'], 'without', n['value'], n['seconds'])"
}
run base A=1 && run split3 ELECTOR_SPLITTERS=3 && run eng4 ELECTOR_ENGINES=4 && run base2 A=1
