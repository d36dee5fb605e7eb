TAG=${1:-prof}
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -- python3 bench.py --no-cpu-baseline > gpurun_out/$TAG/bench_prof.log 2>&1; echo "prof rc=$?"
f=$(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/$TAG/kernel_stats.csv; head -45 gpurun_out/$TAG/kernel_stats.csv | cut -c1-200
find gpurun_out/$TAG/prof -name "*kernel_trace.csv" -delete
grep '"metric"' gpurun_out/$TAG/bench_prof.log | cut -c1-400
