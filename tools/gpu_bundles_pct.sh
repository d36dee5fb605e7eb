mkdir -p gpurun_out/r5pct; export ELECTOR_BENCH_NO_FORK=1
for P in ecoli30x_simlord_lordec yeast50x_nanosim_consent_split; do
for pct in 50 30 70 100 50; do
  ELECTOR_BUNDLE_GLOBAL_PCT=$pct timeout -k 10 200 python bench.py --bundles --profile $P --steps 10 > gpurun_out/r5pct/${P}_$pct.json 2> gpurun_out/r5pct/err.txt || { tail -3 gpurun_out/r5pct/err.txt; exit 1; }
  python3 -c "
import json; j=json.load(open('gpurun_out/r5pct/${P}_$pct.json')); p=j['pipelined']; print('$P pct $pct alone', j['value'], 'pipe', p['step_ms_without_search'], p['step_ms_with_search'], p['step_ms_with_search_from_helper_threads'])"
done; done
