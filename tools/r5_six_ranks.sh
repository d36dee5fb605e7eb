#!/bin/bash
# GPU-box helper (round 5): the N > 1 path of bench.py with SIX ranks sharing the one GPU over gloo (the box allows six
# processes on its card), weak and strong, and the strong read set on one rank for the checksum.  Rehearses the host side of
# a node-wide run: queues, threads, forked setup workers, gather helper threads, memory.
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r5six}; mkdir -p $O
export ELECTOR_BENCH_BACKEND=gloo
P=chr1_20x_ont_50kb
timeout -k 10 500 python bench.py --gpus 6 --reads 1250 --profile $P --steps 10 --warmup 3 --no-cpu-baseline > $O/weak6.json 2> $O/weak6.err || { echo "weak6 FAILED"; tail -5 $O/weak6.err; exit 1; }
timeout -k 10 300 python bench.py --gpus 1 --reads 1250 --profile $P --steps 10 --warmup 3 --no-cpu-baseline --no-configs > $O/weak1.json 2> $O/weak1.err || { echo "weak1 FAILED"; tail -5 $O/weak1.err; exit 1; }
timeout -k 10 500 python bench.py --gpus 6 --reads 1250 --profile $P --scaling strong --strong-units 6 --steps 6 --warmup 2 --no-cpu-baseline > $O/strong6.json 2> $O/strong6.err || { echo "strong6 FAILED"; tail -5 $O/strong6.err; exit 1; }
timeout -k 10 300 python bench.py --gpus 1 --reads 1250 --profile $P --scaling strong --strong-units 6 --steps 6 --warmup 2 --no-cpu-baseline --no-configs > $O/strong1.json 2> $O/strong1.err || { echo "strong1 FAILED"; tail -5 $O/strong1.err; exit 1; }
python3 - $O <<'PY'
import json, sys
o = sys.argv[1]
def load(n):
    for ln in open(o + "/" + n):
        if ln.startswith("{"):
            return json.loads(ln)
w6, w1, s6, s1 = load("weak6.json"), load("weak1.json"), load("strong6.json"), load("strong1.json")
print("weak   6 ranks: value", w6["value"], "ms/step", w6["ms_per_step"], "| 1 rank:", w1["value"], w1["ms_per_step"])
for r in w6["ranks"]["devices"]:
    print("   rank", r["rank"], "host enqueue ms", r.get("host_classify_and_enqueue_ms"), "host wait ms", r.get("host_wait_for_results_ms"), "ms/step", r.get("ms_per_step"), "setup s", r.get("setup_s"))
r1 = w1["ranks"]["devices"][0]
print("   one rank alone: host enqueue ms", r1.get("host_classify_and_enqueue_ms"), "host wait ms", r1.get("host_wait_for_results_ms"), "setup s", r1.get("setup_s"))
print("strong 6 ranks: value", s6["value"], "checksum", s6["counters_checksum"], "| 1 rank: value", s1["value"], "checksum", s1["counters_checksum"], "| equal:", s6["counters_checksum"] == s1["counters_checksum"], "pieces", s6["pieces_gathered"], s1["pieces_gathered"])
print("strong imbalance max/mean (DP cells):", s6["ranks"]["dp_cells_imbalance_max_over_mean"])
PY
