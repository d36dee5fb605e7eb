# GPU-box round script: parity tests, smoke, bench, rocprof kernel stats.  Usage: gpurun -- 'bash tools/_gpu_round.sh TAG'
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/$TAG/pytest.log
tail -5 gpurun_out/$TAG/pytest.log
timeout -k 10 120 python __graft_entry__.py --smoke > gpurun_out/$TAG/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/$TAG/smoke.log
timeout -k 10 300 python bench.py > gpurun_out/$TAG/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/$TAG/bench.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -- python3 bench.py --no-cpu-baseline > gpurun_out/$TAG/bench_prof.log 2>&1; echo "prof rc=$?"
f=$(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/$TAG/kernel_stats.csv; head -12 gpurun_out/$TAG/kernel_stats.csv
find gpurun_out/$TAG/prof -name "*kernel_trace.csv" -delete
