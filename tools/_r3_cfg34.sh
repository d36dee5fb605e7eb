#!/bin/bash
# GPU-box helper: where the windows of the split / mixed profiles go (hand-back counts, phase stamps) and the
# un-overlapped per-kernel table of those profiles
: ${GRAFT_REPO_ROOT:?}
set -o pipefail
O=gpurun_out/${1:-r3cfg}
mkdir -p $O
for P in yeast50x_nanosim_consent_split celegans30x_simlord_mixed; do
  DBG=4 python tools/_r2_dbg.py $P 10001 > $O/dbg_$P.log 2>&1 || exit 1
done
export TMPDIR=/tmp
R=$PWD
for P in yeast50x_nanosim_consent_split celegans30x_simlord_mixed; do
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$P -o serial -- python3 $R/bench.py --serial --steps 5 --no-cpu-baseline --profile $P > $R/$O/prof_$P.json 2> $R/$O/prof_$P.err ) || exit 3
done
find $O -name "*kernel_trace.csv" -delete
head -c 2500 $O/dbg_yeast50x_nanosim_consent_split.log
