#!/bin/bash
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
# GPU-box helper: un-overlapped kernel-trace summaries (CSV) per workload + phase stamps of the fused kernels
set -o pipefail
O=gpurun_out/${1:-r2c}
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for P in ecoli30x_simlord_lordec yeast50x_nanosim_consent_split chr1_20x_ont_50kb; do
  RD=10001; [ $P = chr1_20x_ont_50kb ] && RD=2000
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$P -o serial -- python3 $R/bench.py --serial --profile $P --reads $RD --steps 5 --warmup 1 --no-cpu-baseline > $R/$O/prof_$P.json 2> $R/$O/prof_$P.err ) || exit 6
done
FB_DEBUGS=4 python tools/_fbench.py > $O/phases.txt 2>&1 || exit 7
ls -R $O | head -60
