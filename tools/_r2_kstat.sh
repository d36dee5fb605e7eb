#!/bin/bash
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
# GPU-box helper: per-kernel times of the serial bench (rocprofv3 --stats), top lines
O=gpurun_out/${1:-r2ks}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o serial -- python3 $R/bench.py --serial --steps 10 --warmup 2 --no-cpu-baseline ${KS_EXTRA} > $R/$O/bench.json 2> $R/$O/bench.err ) || exit 1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
grep -E "k_stats|k_merge|k_trivial|k_symbolize|k_part|k_split" $O/kernel_stats.csv | cut -c1-160
find $O/prof -name "*kernel_trace.csv" -delete
