#!/bin/bash
# GPU-box helper: the device splitter's profile file (profiles/r03_k_split.txt): k_split alone and the whole call on the
# E. coli, yeast, yeast -split, C. elegans and chr1 batches, phase cycles per split() pass, parity with the host splitter
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r3splitfinal}
mkdir -p $O
: > $O/k_split.txt
for P in ecoli30x_simlord_lordec yeast50x_nanosim_consent yeast50x_nanosim_consent_split celegans30x_simlord_mixed chr1_20x_ont_50kb; do
  N=10001; [ $P = chr1_20x_ont_50kb ] && N=1559
  timeout -k 10 500 python tools/_r3_split.py $P $N > $O/$P.log 2>&1 || { echo "FAILED $P"; tail -3 $O/$P.log; exit 1; }
  grep "k_split:" $O/$P.log | tail -1 >> $O/k_split.txt
  grep "host view" $O/$P.log | tail -1 >> $O/k_split.txt
  tail -2 $O/$P.log >> $O/k_split.txt
done
cat $O/k_split.txt | cut -c1-200
