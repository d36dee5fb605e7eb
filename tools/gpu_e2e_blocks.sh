#!/bin/bash
# GPU-box helper: the end-to-end run with the device splitter's grid capped (ELECTOR_SPLIT_BLOCKS): fewer long-lived workgroups
# beside the alignment kernels.  Usage: gpu_e2e_blocks.sh TAG [profile] [list]
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-e2eblocks}; P=${2:-ecoli30x_simlord_lordec}; mkdir -p $O
run() { local tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --end-to-end --profile $P --repeat ${REPEAT:-5} --no-reference > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
n=j['without_msa_fa']
print('$P $tag', 'with file', j['value'], j['seconds']['getPOA (wall)'], '| without', n['value'], n['seconds']['getPOA (wall)'])"
}
for B in ${3:-1024 512 384 256 192 1024}; do run blocks$B ELECTOR_SPLIT_BLOCKS=$B || exit 1; done
