#!/bin/bash
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
# GPU-box helper: A/B of ELECTOR_CHAINS_SMALL (third chain for the small launches) at the driver's step count and at 100 steps
for K in 20 100; do
for V in 0 1500 4000 0 1500 4000; do
  ELECTOR_CHAINS_SMALL=$V python bench.py --gpus 1 --steps $K --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($K, $V, j[\"value\"], j[\"ms_per_step\"], j[\"counters_checksum\"])"
done; done
