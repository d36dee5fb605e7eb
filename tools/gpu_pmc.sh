#!/bin/bash
# GPU-box helper: the PMC passes behind `roofline.traffic` / `roofline_valu` of bench.py for one workload profile:
# two SQ counter groups, FETCH_SIZE, WRITE_SIZE -- each in a rocprofv3 run of its own (no tracing domains beside
# --kernel-trace) over `bench.py --serial`; summary -> gpurun_out/TAG/pmc_traffic_<profile>.json (copied to profiles/ by hand)
: ${GRAFT_REPO_ROOT:?}
P=${1:?profile}; TAG=${2:-r5pmc_$P}
READS=10001; [ "$P" = chr1_20x_ont_50kb ] && READS=2000
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; export ELECTOR_BENCH_NO_FORK=1
O=gpurun_out/$TAG
mkdir -p $O
CMD="python3 bench.py --serial --no-cpu-baseline --steps 3 --warmup 1 --profile $P --reads $READS"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p3 -- $CMD > $O/p3.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d $O/p4 -- $CMD > $O/p4.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p5 -- $CMD > $O/p5.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p6 -- $CMD > $O/p6.log 2>&1 || exit 4
python3 tools/pmc_summary.py $O > $O/summary.json || exit 5
python3 tools/pmc_traffic.py $O/summary.json $READS $P "${COLLECTED:-round 5}" > $O/traffic.log || exit 6
cp profiles/pmc_traffic_$P.json $O/
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
python3 -c "
import json; j=json.load(open('$O/pmc_traffic_$P.json')); print('$P', 'valu/step', j['valu_wave_insts_per_step'], 'traffic/step', j['traffic_bytes_per_step'], 'k_poa valu active', j['k_poa_valu_active_of_wave_cycles'])"
