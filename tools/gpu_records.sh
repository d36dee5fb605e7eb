#!/bin/bash
# GPU-box helper: the records behind DESIGN.md / profiles/: the driver's command, one line per BASELINE profile,
# the un-overlapped kernel tables under rocprofv3.  Usage: gpu_records.sh TAG [part]   (part: bench | lines | tables | all)
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r5records}; PART=${2:-all}; mkdir -p $O
R=$PWD
export TMPDIR=/tmp
if [ $PART = bench -o $PART = all ]; then
  python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || { echo "default bench FAILED"; tail -5 $O/bench_default.err; exit 1; }
  python3 -c "
import json
j=json.load(open('$O/bench_default.json'))
print('default', j['config']['profile'], 'value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'parity', j['parity_vs_reference']['differing'], '/', j['parity_vs_reference']['windows'], 'cpu', j['cpu_baseline']['value'])
for c in j['configs']: print('  cfg', c['profile'], c['value'], c['value_rows_in_hbm'], c['parity_vs_reference']['differing'], '/', c['parity_vs_reference']['windows'])"
fi
if [ $PART = lines -o $PART = all ]; then
  for P in ecoli30x_simlord_lordec yeast50x_nanosim_consent yeast50x_nanosim_consent_split celegans30x_simlord_mixed chr1_20x_ont_50kb; do
    timeout -k 10 400 python bench.py --profile $P --steps 50 --no-cpu-baseline > $O/bench_$P.json 2> $O/bench_$P.err || { echo FAILED $P; tail -3 $O/bench_$P.err; exit 2; }
    python3 -c "
import json
j=json.load(open('$O/bench_$P.json'))
k=j['kernel_ms_per_step']
print('$P', 'value', j['value'], 'hbm', j['value_rows_in_hbm'], 'ms/step', j['ms_per_step'], 'k_poa', k['k_poa'], 'hb', round(k['alignment1_stage']+k['alignment2_stage'],3), 'other', k['other'], 'merge', k['merge_and_counters'], 'host', k['host_classify_and_enqueue'], 'checksum', j['counters_checksum'])"
  done
  timeout -k 10 300 python bench.py --profile ecoli10x_c1 --reads 459 --steps 50 --no-cpu-baseline > $O/bench_ecoli10x_c1.json 2> $O/bench_c1.err || echo "c1 failed"
fi
if [ $PART = tables -o $PART = all ]; then
  export ELECTOR_BENCH_NO_FORK=1
  for P in ecoli30x_simlord_lordec yeast50x_nanosim_consent_split; do
    ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$P -o serial -- python3 $R/bench.py --serial --steps 20 --no-cpu-baseline --profile $P > $R/$O/serial_under_rocprof_$P.json 2> $R/$O/serial_$P.err ) || { echo "rocprof FAILED $P"; tail -5 $O/serial_$P.err; exit 3; }
    cp $(find $O/prof_$P -name "*kernel_stats.csv" | head -1) $O/serial_kernel_stats_$P.csv
    find $O/prof_$P -name "*kernel_trace.csv" -delete
    python bench.py --serial --steps 20 --no-cpu-baseline --profile $P > $O/bench_serial_$P.json 2> /dev/null
  done
fi
echo done
