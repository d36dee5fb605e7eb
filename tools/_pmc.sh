cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
export FB_DEBUGS=0
rm -rf gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc1 -- python3 tools/_fbench.py > gpurun_out/pmc1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc2 -- python3 tools/_fbench.py > gpurun_out/pmc2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc3 -- python3 tools/_fbench.py > gpurun_out/pmc3.log 2>&1
for d in pmc1 pmc2 pmc3; do
f=$(find gpurun_out/$d -name "*counter_collection.csv" | head -1)
echo "== $d $f"
python3 - "$f" <<PY
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r["Kernel_Name"].replace("void elector::","").replace("elector::","").split("(")[0]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
names=sorted({c for k in agg for c in agg[k]})
print("kernel", names)
for k in agg:
    if "fused" in k or "symbol" in k: print(k, [ "%.3g"%agg[k][c] for c in names])
PY
done
