# PMC passes over bench.py (separate passes: FETCH_SIZE and WRITE_SIZE cannot share one), kernel-trace only.
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
# Usage: gpurun -- 'bash tools/_pmc_bench.sh TAG'
TAG=${1:-pmc}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
CMD="python3 bench.py --no-cpu-baseline --steps 3 --warmup 1"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/$TAG/p1 -- $CMD > gpurun_out/$TAG/p1.log 2>&1 && \
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d gpurun_out/$TAG/p2 -- $CMD > gpurun_out/$TAG/p2.log 2>&1 && \
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/$TAG/p3 -- $CMD > gpurun_out/$TAG/p3.log 2>&1
python3 tools/_pmc_summary.py gpurun_out/$TAG > gpurun_out/$TAG/summary.json && cat gpurun_out/$TAG/summary.json
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/$TAG -name "*counter_collection.csv" -size +20M -delete
