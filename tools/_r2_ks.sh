export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out/r2ci
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2ci/prof -o ys -- python3 $R/bench.py --serial --profile yeast50x_nanosim_consent_split --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r2ci/ys.json 2> $R/gpurun_out/r2ci/ys.err )
head -16 gpurun_out/r2ci/prof/ys_kernel_stats.csv | cut -c1-130
