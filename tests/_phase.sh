cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ELECTOR_DEBUG_BINS=1 FB_DEBUGS=4 timeout -k 10 300 python3 tests/_fbench.py > gpurun_out/phase.log 2>&1; grep -m1 "classes" gpurun_out/phase.log; tail -22 gpurun_out/phase.log | cut -c1-330
