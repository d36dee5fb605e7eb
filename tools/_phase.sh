cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
FB_DEBUGS=${FB:-4} timeout -k 10 300 python3 tools/_fbench.py > gpurun_out/phase.log 2>&1; grep -m2 "ring depth\|classes" gpurun_out/phase.log; tail -${TAILN:-22} gpurun_out/phase.log | cut -c1-330
