"""bench.py as the driver runs it, at toy size: the JSON contract, and the N > 1 path rehearsed with two
ranks sharing this box's GPU over gloo (`--gpus 2` without a launcher makes bench.py start the ranks
itself; on the 8-GPU node the ranks use RCCL, one GPU each)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--reads", "300",
                        "--serial-steps", "1", "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_contract_one_gpu():
    j = _bench([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["pieces_gathered"] == 300 and j["scaling"] == "weak"
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6
    assert j["gcups_computed"] <= j["gcups_effective"]
    # the second figure: the same steps with the merged rows brought to the host (SURVEY.md 8(d)'s literal definition)
    assert j["config"]["rows_to_host"] is False and 0 < j["value_rows_to_host"] and j["rows_to_host"]["bytes_per_step_per_gpu"] > 0
    assert j["ranks"]["world"] == 1 and j["ranks"]["distinct_devices"] == 1 and len(j["ranks"]["devices"]) == 1
    # un-overlapped: the per-kernel times of a step add up to no more than the serial step's wall time
    k = j["kernel_ms_per_step"]
    assert k["k_poa"] + k["alignment1_stage"] + k["alignment2_stage"] + k["other"] + k["merge_and_counters"] <= 1.05 * k["serial_step_wall"]


def test_bench_two_ranks_share_the_gpu_over_gloo():
    one = _bench(["--profile", "yeast50x_nanosim_consent_split"])
    two = _bench(["--gpus", "2", "--profile", "yeast50x_nanosim_consent_split"], {"ELECTOR_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["config"]["parallelism"] == "shard-by-read x2"
    # weak scaling: every rank brings its own 300 reads; rank 0 gathered both ranks' counter rows
    assert two["pieces_gathered"] > 1.8 * one["pieces_gathered"] and two["pieces_gathered"] >= 600
    # what the process group saw: two ranks, ONE device here (they share it); the 8-GPU node must show N distinct ones
    assert two["ranks"]["world"] == 2 and two["ranks"]["backend"] == "gloo" and two["ranks"]["distinct_devices"] == 1
    assert sorted(r["rank"] for r in two["ranks"]["devices"]) == [0, 1]
