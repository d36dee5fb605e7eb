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
    """the driver's command at toy size: default profile (BASELINE.json configs[2], yeast `-split`), three rotating
    batches, `value` with rows and counters on the host, the `configs` array beside it"""
    j = _bench([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["scaling"] == "weak" and j["pieces_gathered"] >= 300
    assert j["config"]["profile"] == "yeast50x_nanosim_consent_split" and j["config"]["batches_rotated"] == 3
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6
    assert j["gcups_computed"] <= j["gcups_effective"]
    # `value` is SURVEY.md 8(d)'s literal figure (rows + counters on the host); the lenient one stands beside it
    assert j["config"]["rows_to_host"] is True and j["config"]["offsets_resident_in_hbm"] is True
    assert 0 < j["value"] and 0 < j["value_rows_in_hbm"] and j["rows_to_host"]["bytes_per_step_per_gpu"] > 0
    assert j["ranks"]["world"] == 1 and j["ranks"]["distinct_devices"] == 1 and len(j["ranks"]["devices"]) == 1
    # un-overlapped: the per-kernel times of a step add up to no more than the serial step's wall time
    k = j["kernel_ms_per_step"]
    assert k["k_poa"] + k["k_poa_far_instance"] + k["alignment1_stage"] + k["alignment2_stage"] + k["other"] + k["merge_and_counters"] <= 1.05 * k["serial_step_wall"]
    # the other single-GPU profiles of BASELINE.json, one entry each
    assert sorted(c["profile"] for c in j["configs"]) == ["celegans30x_simlord_mixed", "chr1_20x_ont_50kb", "ecoli30x_simlord_lordec"]
    for c in j["configs"]:
        assert c["value"] > 0 and c["value_rows_in_hbm"] > 0 and c["windows_per_step"] > 0


def test_bench_two_ranks_share_the_gpu_over_gloo():
    one = _bench(["--profile", "yeast50x_nanosim_consent_split", "--batches", "2"])
    two = _bench(["--gpus", "2", "--profile", "yeast50x_nanosim_consent_split", "--batches", "2"], {"ELECTOR_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["config"]["parallelism"] == "shard-by-read x2"
    # weak scaling: every rank brings its own 300 reads; rank 0 gathered both ranks' counter rows
    assert two["pieces_gathered"] > 1.8 * one["pieces_gathered"] and two["pieces_gathered"] >= 600
    # what the process group saw: two ranks, ONE device here (they share it); the 8-GPU node must show N distinct ones
    assert two["ranks"]["world"] == 2 and two["ranks"]["backend"] == "gloo" and two["ranks"]["distinct_devices"] == 1
    assert sorted(r["rank"] for r in two["ranks"]["devices"]) == [0, 1]


def test_bench_strong_scaling_cuts_one_read_set_by_cells():
    """--scaling strong: the same read set whatever the rank count, cut by distributed.shard_bounds over
    read_cell_estimate; one rank and two ranks (sharing the GPU over gloo) process the same reads: same totals,
    same counters checksum, and the two ranks' DP cells are balanced"""
    extra = ["--scaling", "strong", "--strong-units", "3", "--profile", "celegans30x_simlord_mixed"]
    one = _bench(extra)
    two = _bench(["--gpus", "2"] + extra, {"ELECTOR_BENCH_BACKEND": "gloo"})
    assert one["scaling"] == "strong" and two["scaling"] == "strong"
    assert one["strong"]["reads_total"] == two["strong"]["reads_total"] == 900
    assert one["pieces_gathered"] == two["pieces_gathered"] and one["counters_checksum"] == two["counters_checksum"]
    assert one["config"]["ref_bases_per_gpu"] == 2 * two["config"]["ref_bases_per_gpu"] or \
        abs(one["config"]["ref_bases_per_gpu"] - 2 * two["config"]["ref_bases_per_gpu"]) <= 2
    assert two["ranks"]["dp_cells_imbalance_max_over_mean"] < 1.10          # 450 reads per rank: a read is 0.2 % of a shard


def test_bench_two_ranks_over_rccl_on_two_devices():
    """The N > 1 path as the driver launches it: RCCL, one GPU per rank.  Rank 1's counter gathers are started from a
    helper thread, whose current device would be GPU 0 unless the pipe is told its device (the round-3 advisor's
    finding): with two devices the gather must complete and rank 0 must see both ranks' rows.  Skipped on a one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    two = _bench(["--gpus", "2", "--profile", "yeast50x_nanosim_consent_split", "--batches", "2"])
    assert two["n_gpus"] == 2 and two["ranks"]["backend"] == "nccl" and two["ranks"]["distinct_devices"] == 2
    assert two["pieces_gathered"] >= 600
