"""N > 1 path on CPU: two gloo ranks shard the reads, compute counters for their
own range and gather them to rank 0; the aggregate must equal the single-process
result (floats are computed after the gather, in read order)."""
import io
import os
import sys
import tempfile

import numpy as np
import pytest
import torch.multiprocessing as mp
from portutil import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _oracle_counter_fn(pieces):
    """Stand-in for the GPU kernel on the CPU test box: oracle counters in C-ABI layout."""
    import stats_oracle
    from elector_amd import computeStats as cs
    from test_stats_cpu import oracle_counter_array
    out = []
    for r in range(len(pieces.read_first) - 1):
        st = dict(is_extended=False, is_trimmed=False, missing=0, extended_bases=[])
        p0, p1 = int(pieces.read_first[r]), int(pieces.read_first[r + 1])
        union = set()
        for p in range(p0, p1):
            n = int(pieces.cols[p])
            if n <= 10:
                continue
            a = int(pieces.row_off[p])
            txt = pieces.rows[a:a + 3 * n].tobytes().decode()
            k, ex = stats_oracle.piece_counters(txt[:n], txt[n:2 * n], txt[2 * n:], None, [], 5, st)
            union.update(i for i, e in enumerate(ex) if e)
            if p1 - p0 > 1 and p == p1 - 1:
                k["missing_last"] = sum(1 for i in range(n) if i not in union and txt[i] != ".")
            out.append(k)
    return oracle_counter_array(pieces, out)


def _worker(rank, world, port, msa_path, result_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from elector_amd import computeStats as cs
    from elector_amd import distributed
    pieces = cs.parse_msa(msa_path, cs.getSplit(msa_path))
    allc = distributed.sharded_counters(pieces, _oracle_counter_fn)
    if rank == 0:
        np.save(result_path, allc)
    else:
        assert allc is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process(tmp_path):
    import msa_gen
    from elector_amd import computeStats as cs
    reads = msa_gen.make_reads(77, 22, 700)
    txt, _, _ = msa_gen.msa_text(reads)
    msa = str(tmp_path / "msa.fa")
    open(msa, "w").write(txt)
    res = str(tmp_path / "counters.npy")
    mp.spawn(_worker, args=(2, free_port(), msa, res), nprocs=2, join=True)
    got = np.load(res)
    pieces = cs.parse_msa(msa, cs.getSplit(msa))
    exp = _oracle_counter_fn(pieces)
    assert np.array_equal(got, exp)
    # and the float aggregation from the gathered integers equals the oracle's report
    import stats_oracle
    r, _ = stats_oracle.compute_metrics(txt, 5)
    agg = cs.aggregate(pieces, got, r["lastReadRatios"], io.StringIO())
    assert agg[0] == r["nbReads"] and agg[3] == r["precision"] and agg[4] == r["recall"]
    assert agg[12] == r["indelsubsUncorr"] and agg[13] == r["indelsubsCorr"]


def test_shard_bounds():
    from elector_amd.distributed import shard_bounds
    b = shard_bounds([10, 10, 10, 10], 2)
    assert b.tolist() == [0, 2, 4] or b.tolist() == [0, 1, 4] or b[0] == 0 and b[-1] == 4
    assert shard_bounds([], 4).tolist() == [0, 0, 0, 0, 0]
    b = shard_bounds(np.ones(101), 8)
    assert b[0] == 0 and b[-1] == 101 and np.all(np.diff(b) >= 12) and np.all(np.diff(b) <= 14)


def test_empty_shard_gathers(tmp_path):
    """Fewer reads than ranks: the rank whose shard is empty must send a [0, ES_NCOUNTERS] block."""
    import msa_gen
    from elector_amd import computeStats as cs
    reads = msa_gen.make_reads(78, 1, 700)[:1]
    txt, _, _ = msa_gen.msa_text(reads)
    msa = str(tmp_path / "msa.fa")
    open(msa, "w").write(txt)
    res = str(tmp_path / "counters.npy")
    mp.spawn(_worker, args=(2, free_port(), msa, res), nprocs=2, join=True)
    got = np.load(res)
    pieces = cs.parse_msa(msa, cs.getSplit(msa))
    assert np.array_equal(got, _oracle_counter_fn(pieces))


def test_read_cell_estimate_balances_by_cells():
    """ranks get contiguous read ranges of near-equal DP cells, not near-equal counts."""
    from elector_amd.distributed import read_cell_estimate, shard_bounds
    lr = np.array([8000] * 50 + [50000] * 10)
    w = read_cell_estimate(lr, lr, lr)
    b = shard_bounds(w, 2)
    assert b[1] > 50                       # the ten long reads weigh more than the fifty short ones
    assert abs(w[: b[1]].sum() - w[b[1]:].sum()) <= w.max()
    # a trimmed corrected read costs less than a whole one (its missing part becomes 1-letter `N` windows)
    assert read_cell_estimate(np.array([8000]), np.array([2000]), np.array([8000]))[0] < w[0]


def test_bench_refuses_world_size_mismatch():
    """`bench.py --gpus N` with another WORLD_SIZE must fail loudly, before any GPU call (no GPU here)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


def test_bench_self_launch_command(monkeypatch):
    """--gpus N without WORLD_SIZE starts N fresh rank processes through torch.distributed.run."""
    import importlib
    import subprocess
    bench = importlib.import_module("bench")
    seen = {}

    class R:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench.self_launch(bench.parse()) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_chr1_shards_of_eight_ranks_are_balanced():
    """The partitioner of DESIGN.md section 5 on BASELINE.json's 8-GPU config: one batch of chr1-like 50 kb reads
    (log-normal lengths) cut into eight contiguous ranges by shard_bounds over read_cell_estimate -- what every rank
    computes before it has split a single read -- against the DP cells the windows of those reads really have
    (Lr x Lc + max(Lr, Lc) x Lu per window of the host splitter): max / mean <= 1.05 at 2, 4 and 8 ranks."""
    from elector_amd import split, synthetic
    from elector_amd.distributed import read_cell_estimate, shard_bounds
    triples, headers, read_of = synthetic.read_pieces("chr1_20x_ont_50kb", 2000, seed=1000)
    win = split.split_reads(triples, 0.1, headers, nthreads=8)
    off = win.off
    lr, lc, lu = off[1::3] - off[0:-1:3], off[2::3] - off[1:-1:3], off[3::3] - off[2:-1:3]
    cells_piece = np.add.reduceat(lr * lc + np.maximum(lr, lc) * lu, win.read_first[:-1])
    ids = np.asarray(read_of)
    nr = int(ids.max()) + 1
    cells_read = np.bincount(ids[win.read_index], weights=cells_piece, minlength=nr)
    R, C, U = np.zeros(nr), np.zeros(nr), np.zeros(nr)
    for (r, c, u), i in zip(triples, ids):
        R[i] = len(r)
        U[i] = len(u)
        C[i] += len(c)
    w = read_cell_estimate(R, C, U)
    for world in (2, 4, 8):
        b = shard_bounds(w, world)
        act = [cells_read[b[k]:b[k + 1]].sum() for k in range(world)]
        assert max(act) / np.mean(act) <= 1.05, (world, max(act) / np.mean(act))
