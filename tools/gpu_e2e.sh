#!/bin/bash
# GPU-box helper: end-to-end run (three FASTA files -> getPOA -> outputRecallPrecision) under a few settings.
# Usage: gpu_e2e.sh TAG [profile] -- settings are "tag ENV=.. ENV=.." lines below
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-e2e}; P=${2:-ecoli30x_simlord_lordec}; mkdir -p $O
EXTRA=""; [ "$P" = chr1_20x_ont_50kb ] && EXTRA="--reads 6400 --repeat 2"; [ "$P" = yeast50x_nanosim_consent_split ] && EXTRA="--reads 40004 --repeat 2"
run() { local tag=$1; shift
  env "$@" timeout -k 10 500 python bench.py --end-to-end --profile $P ${EXTRA:---repeat ${REPEAT:-5}} --no-reference > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
n=j['without_msa_fa']
print('$P $tag', 'with file', j['value'], j['seconds']['getPOA (wall)'], '| without', n['value'], n['seconds']['getPOA (wall)'], n['seconds']['outputRecallPrecision (wall)'])"
}
for rep in 1 2 3; do run base$rep A=1 && run hipcopy$rep ELECTOR_ROWS_HIP_COPY=1 || exit 1; done
