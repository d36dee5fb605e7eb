#!/bin/bash
# GPU-box helper (round 4): parity tests of the POA engine, then bench A/B of the device classification and the slot size
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r4first}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_poa_gpu.py tests/test_configs_gpu.py -x -q -m gpu --capture=sys > $O/pytest.log 2>&1 || { echo "PYTEST FAILED"; tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
run() { # tag, env...
  local tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --profile ecoli30x_simlord_lordec --batches 1 --steps 40 --no-cpu-baseline --serial-steps 4 > $O/$tag.json 2> $O/$tag.err || { echo FAILED $tag; tail -5 $O/$tag.err; return 1; }
  python3 -c "
import json
j=json.load(open('$O/$tag.json'))
k=j['kernel_ms_per_step']
print('$tag', 'value', j['value'], 'ms/step', j['ms_per_step'], 'k_poa', k['k_poa'], 'other', k['other'], 'host', k['host_classify_and_enqueue'], 'checksum', j['counters_checksum'])"
}
run dev_off A=1 && run host_off ELECTOR_BENCH_HOST_OFFSETS=1 && run slot125 ELECTOR_POA_SLOT_PCT=125 && run slot150 ELECTOR_POA_SLOT_PCT=150 && run slot200 ELECTOR_POA_SLOT_PCT=200 && run dev_off2 A=1
