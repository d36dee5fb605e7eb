#!/bin/bash
# stage trace of the end-to-end run, with and without msa.fa
set -e
T=${1:-r4e2etr}
mkdir -p gpurun_out/$T
ELECTOR_STAGE_TRACE=1 ELECTOR_DEBUG_HOST=1 python bench.py --end-to-end --profile ecoli30x_simlord_lordec --repeat 5 --no-reference > gpurun_out/$T/e2e.json 2> gpurun_out/$T/e2e.err
tail -c 600 gpurun_out/$T/e2e.json
