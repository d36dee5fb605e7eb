#!/bin/bash
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
# GPU-box helper: k_stats / k_merge times of the serial bench for several LDS budgets of the bit rows
for P in ecoli30x_simlord_lordec yeast50x_nanosim_consent_split; do
for BW in default 256 512 0; do
  if [ $BW = default ]; then unset ELECTOR_STATS_BITWORDS; else export ELECTOR_STATS_BITWORDS=$BW; fi
  echo "== $P bitwords $BW"
  KS_EXTRA="--profile $P" bash tools/_r2_kstat.sh r2ks_$BW | grep -E "k_stats|k_merge" | cut -d, -f1-4
done; done
