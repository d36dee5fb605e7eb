#!/bin/bash
: ${GRAFT_REPO_ROOT:?}   # these helpers run on the GPU box only (they cd and delete below that path)
# GPU-box helper: bench lines of every workload + the un-overlapped rocprof pass (kept under profiles/ by hand)
set -o pipefail
O=gpurun_out/${1:-r2b}
mkdir -p $O
python bench.py --steps 100 --warmup 5 > $O/bench_ecoli.json 2> $O/bench_ecoli.err || exit 1
tail -c 3000 $O/bench_ecoli.json
python bench.py --profile yeast50x_nanosim_consent_split --steps 40 --no-cpu-baseline > $O/bench_yeast_split.json 2> $O/bench_yeast_split.err || exit 2
python bench.py --profile yeast50x_nanosim_consent --steps 40 --no-cpu-baseline > $O/bench_yeast.json 2> $O/bench_yeast.err || exit 2
python bench.py --profile celegans30x_simlord_mixed --steps 40 --no-cpu-baseline > $O/bench_celegans.json 2> $O/bench_celegans.err || exit 3
python bench.py --profile chr1_20x_ont_50kb --reads 2000 --steps 30 --no-cpu-baseline > $O/bench_chr1.json 2> $O/bench_chr1.err || exit 4
python bench.py --serial --steps 20 --no-cpu-baseline > $O/bench_serial.json 2> $O/bench_serial.err || exit 5
export TMPDIR=/tmp
R=$PWD
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_serial -o serial -- python3 $R/bench.py --serial --steps 20 --no-cpu-baseline > $R/$O/prof_serial.json 2> $R/$O/prof_serial.err ) || exit 6
ls -R $O | head -40
