cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
FB_DEBUGS=4 timeout -k 10 300 python3 tests/_fbench.py > gpurun_out/phase.log 2>&1; tail -40 gpurun_out/phase.log
