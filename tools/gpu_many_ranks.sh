#!/bin/bash
# GPU-box helper: the N > 1 path of bench.py with FIVE ranks sharing the one GPU over gloo (the box allows six processes on its card, the launcher included: five ranks
# processes on its card), weak and strong, and the strong read set on one rank for the checksum.  Rehearses the host side of
# a node-wide run: queues, threads, forked setup workers, gather helper threads, memory.
: ${GRAFT_REPO_ROOT:?}
O=gpurun_out/${1:-r5six}; mkdir -p $O
export ELECTOR_BENCH_BACKEND=gloo
# (two engine contexts per rank instead of four: five ranks x four contexts of this profile do not fit the one GPU's 288 GB;
# on a node every rank has a GPU of its own)
export ELECTOR_BENCH_ENGINES=2
P=chr1_20x_ont_50kb
timeout -k 10 500 python bench.py --gpus 5 --reads 1250 --profile $P --steps 10 --warmup 3 --no-cpu-baseline > $O/weak5.json 2> $O/weak5.err || { echo "weak5 FAILED"; tail -5 $O/weak5.err; exit 1; }
timeout -k 10 300 python bench.py --gpus 1 --reads 1250 --profile $P --steps 10 --warmup 3 --no-cpu-baseline --no-configs > $O/weak1.json 2> $O/weak1.err || { echo "weak1 FAILED"; tail -5 $O/weak1.err; exit 1; }
timeout -k 10 500 python bench.py --gpus 5 --reads 1250 --profile $P --scaling strong --strong-units 5 --steps 6 --warmup 2 --no-cpu-baseline > $O/strong5.json 2> $O/strong5.err || { echo "strong5 FAILED"; tail -5 $O/strong5.err; exit 1; }
timeout -k 10 300 python bench.py --gpus 1 --reads 1250 --profile $P --scaling strong --strong-units 5 --steps 6 --warmup 2 --no-cpu-baseline --no-configs > $O/strong1.json 2> $O/strong1.err || { echo "strong1 FAILED"; tail -5 $O/strong1.err; exit 1; }
python3 - $O <<'PY'
import json, sys
o = sys.argv[1]
def load(n):
    for ln in open(o + "/" + n):
        if ln.startswith("{"):
            return json.loads(ln)
w5, w1, s5, s1 = load("weak5.json"), load("weak1.json"), load("strong5.json"), load("strong1.json")
print("weak   5 ranks: value", w5["value"], "ms/step", w5["ms_per_step"], "| 1 rank:", w1["value"], w1["ms_per_step"])
for r in w5["ranks"]["devices"]:
    print("   rank", r["rank"], "host enqueue ms", r.get("host_classify_and_enqueue_ms"), "host wait ms", r.get("host_wait_for_results_ms"), "ms/step", r.get("ms_per_step"), "setup s", r.get("setup_s"))
r1 = w1["ranks"]["devices"][0]
print("   one rank alone: host enqueue ms", r1.get("host_classify_and_enqueue_ms"), "host wait ms", r1.get("host_wait_for_results_ms"), "setup s", r1.get("setup_s"))
print("strong 5 ranks: value", s5["value"], "checksum", s5["counters_checksum"], "| 1 rank: value", s1["value"], "checksum", s1["counters_checksum"], "| equal:", s5["counters_checksum"] == s1["counters_checksum"], "pieces", s5["pieces_gathered"], s1["pieces_gathered"])
print("strong imbalance max/mean (DP cells):", s5["ranks"]["dp_cells_imbalance_max_over_mean"])
PY
