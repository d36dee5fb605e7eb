cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
mkdir -p gpurun_out/trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace/p -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/trace/log 2>&1
f=$(find gpurun_out/trace/p -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if "elector" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last step: find last k_symbolize
idx=[i for i,r in enumerate(rows) if "k_symbolize" in r["Kernel_Name"]]
i0=idx[-1]
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    n=r["Kernel_Name"].replace("void elector::","").replace("elector::","").split("(")[0]
    print("%8.3f %8.3f  q%-3s %s" % ((int(r["Start_Timestamp"])-t0)/1e6,(int(r["End_Timestamp"])-t0)/1e6, r.get("Queue_Id","?"), n))
PY
rm -rf gpurun_out/trace/p
