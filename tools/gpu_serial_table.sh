#!/bin/bash
# GPU-box helper: the un-overlapped kernel table of one profile (rocprofv3 --kernel-trace --stats over bench.py --serial), printed per step
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-table}; P=${2:-yeast50x_nanosim_consent_split}; mkdir -p $O; R=$PWD; export TMPDIR=/tmp ELECTOR_BENCH_NO_FORK=1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o serial -- python3 $R/bench.py --serial --steps 20 --no-cpu-baseline --no-configs --no-rows-in-hbm --profile $P > $R/$O/serial.json 2> $R/$O/serial.err ) || { tail -5 $O/serial.err; exit 1; }
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/serial_kernel_stats_$P.csv
find $O/prof -name "*kernel_trace.csv" -delete
python3 - $O/serial_kernel_stats_$P.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = max(int(r['Calls']) for r in rows if 'k_classify' in r['Name'])
kp = sum(float(r['TotalDurationNs']) for r in rows if 'k_poa<' in r['Name'])
print('steps', steps, 'k_poa ms/step %.3f' % (kp / 1e6 / steps))
for r in rows[:40]:
    if 'k_poa<' in r['Name']: continue
    print('%-64s calls/step %5.1f  ms/step %7.3f  avg us %8.1f' % (r['Name'].replace('void ', '').replace('elector::', '')[:64], int(r['Calls']) / steps, float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3))
PY
