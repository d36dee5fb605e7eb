#!/bin/bash
# GPU-box helper: the end-to-end run with the splitter's and the alignment contexts' streams on disjoint / overlapping sets
# of compute units (ELECTOR_SPLIT_CUS / ELECTOR_ENGINE_CUS = lo:hi of the 256-bit queue mask).  Usage: gpu_e2e_cus.sh TAG [profile]
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-e2ecus}; P=${2:-ecoli30x_simlord_lordec}; mkdir -p $O
run() { local tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --end-to-end --profile $P --repeat ${REPEAT:-5} --no-reference > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
n=j['without_msa_fa']
print('$P $tag', 'with file', j['value'], j['seconds']['getPOA (wall)'], '| without', n['value'], n['seconds']['getPOA (wall)'])"
}
run base1 A=1 || exit 1
run split64 ELECTOR_SPLIT_CUS=0:64 || exit 1
run split96 ELECTOR_SPLIT_CUS=0:96 || exit 1
run split64_eng ELECTOR_SPLIT_CUS=0:64 ELECTOR_ENGINE_CUS=64:256 || exit 1
run split96_eng ELECTOR_SPLIT_CUS=0:96 ELECTOR_ENGINE_CUS=96:256 || exit 1
run split128_eng ELECTOR_SPLIT_CUS=0:128 ELECTOR_ENGINE_CUS=128:256 || exit 1
run base2 A=1
