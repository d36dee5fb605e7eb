cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
for i in 1 2; do timeout -k 10 300 python3 tools/_hostbench.py > gpurun_out/host.log 2>&1; tail -1 gpurun_out/host.log | cut -c1-300; done
