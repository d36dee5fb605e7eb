#!/bin/bash
# end-to-end run with fewer streams side by side (is the contention between k_split and k_poa what paces it?)
O=gpurun_out/${1:-r4e2eab2}; mkdir -p $O
run() { # tag env...
  local tag=$1; shift
  env "$@" timeout -k 10 500 python bench.py --end-to-end --profile ecoli30x_simlord_lordec --repeat 5 --no-reference > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
n=j['without_msa_fa']
print('$tag', 'with file', j['value'], j['seconds']['getPOA (wall)'], 'without', n['value'], n['seconds'])"
}
run base A=1 && run split1 ELECTOR_SPLITTERS=1 && run eng2 ELECTOR_ENGINES=2 && run split1eng2 ELECTOR_SPLITTERS=1 ELECTOR_ENGINES=2
