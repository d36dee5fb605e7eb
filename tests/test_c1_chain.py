"""BASELINE config 1 restated (synthetic profile `ecoli10x_c1`: 459 reads, 489 corrected pieces, 4.7 Mbases)
against the WHOLE real reference chain: tests/golden/c1_chain.json holds what masterSplitter -> poa per slot ->
Donatello per slot -> the imported reference computeStats produced for these very reads in the container
(oracle/make_golden.py --c1-only; elector/__main__.py:140-141): digest and size of msa.fa, the two counters, the
19-tuple, stdout, the log text and the side files.

  * not gpu: the CPU oracle chain (host splitter -> oracle/poa_oracle.c -> host merger -> oracle/stats_oracle.py)
    reproduces every byte of it;
  * gpu: both call sites of the product -- getPOA with the DEVICE splitter, no engine shortcuts beyond the shared
    context, then outputRecallPrecision from the DEVICE counters -- reproduce every byte of it, with one rank and
    with two ranks (gloo; RCCL on the 8-GPU node).
"""
import hashlib
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "c1_chain.json")))


def c1_reads():
    from elector_amd import synthetic
    triples, headers, _ = synthetic.read_pieces(GOLD["profile"], GOLD["n_reads"], GOLD["seed"])
    assert len(triples) == GOLD["n_pieces"]
    return [(h, r, c, u) for h, (r, c, u) in zip(headers, triples)]


def write_reads(d, reads):
    for fn, k in (("ref.fa", 1), ("cor.fa", 2), ("unc.fa", 3)):
        with open(os.path.join(d, fn), "wb") as f:
            for r in reads:
                f.write(r[0] + b"\n" + r[k] + b"\n")
        assert hashlib.sha256(open(os.path.join(d, fn), "rb").read()).hexdigest() == GOLD["inputs_sha256"][fn], \
            "the generator no longer produces the reads the fixture was made from"


def check_msa(path):
    data = open(path, "rb").read()
    assert len(data) == GOLD["msa_bytes"]
    assert data.count(b">") // 3 == GOLD["msa_records"]
    assert hashlib.sha256(data).hexdigest() == GOLD["msa_sha256"]


def check_report(tup, out, log, d):
    assert json.loads(json.dumps(tup)) == GOLD["tuple"]
    assert out == GOLD["stdout"]
    assert log == GOLD["log"]
    assert open(os.path.join(d, "per_read_metrics.txt")).read() == GOLD["per_read"]
    sizes = open(os.path.join(d, "read_size_distribution.txt")).read()
    assert sizes.count("\n") == GOLD["read_size_distribution_lines"]
    assert hashlib.sha256(sizes.encode()).hexdigest() == GOLD["read_size_distribution_sha256"]


def test_oracle_chain_reproduces_the_reference_chain(tmp_path):
    """pins splitter port + poa_oracle.c + merger port + stats_oracle.py together on 83,000 windows"""
    import msa_gen
    import stats_oracle
    reads = c1_reads()
    write_reads(str(tmp_path), reads)
    txt, small, wrong = msa_gen.msa_text(reads)
    assert (small, wrong) == (GOLD["small"], GOLD["wrong"])
    (tmp_path / "msa.fa").write_text(txt)
    check_msa(str(tmp_path / "msa.fa"))
    tup, out, log, per_read, _ = stats_oracle.output_recall_precision(txt, small, wrong, 5, 0.1)
    assert json.loads(json.dumps(tup)) == GOLD["tuple"]
    assert "None\n" + out == GOLD["stdout"] and log == GOLD["log"]
    assert "score metric\n" + "".join(per_read) == GOLD["per_read"]


@pytest.mark.gpu
def test_both_call_sites_reproduce_the_reference_chain(tmp_path, capsys):
    """getPOA (device splitter, the module's own engine pool) -> outputRecallPrecision (device counters)"""
    from elector_amd import alignment, computeStats
    d = str(tmp_path)
    write_reads(d, c1_reads())
    assert os.environ.get("ELECTOR_HOST_SPLIT", "0") in ("", "0")
    small, wrong = alignment.getPOA(d + "/cor.fa", d + "/ref.fa", d + "/unc.fa", 8, d, 0.1)
    capsys.readouterr()
    assert (small, wrong) == (GOLD["small"], GOLD["wrong"])
    check_msa(d + "/msa.fa")
    assert computeStats.cached_pieces(d + "/msa.fa", {}) is not None, "the device counters did not reach call site #2"
    log = io.StringIO()
    tup = computeStats.outputRecallPrecision(d + "/cor.fa", d, log, small, wrong, 5, 0.1, "read_size_distribution.txt", {})
    check_report(tup, capsys.readouterr().out, log.getvalue(), d)
    # ... and from the text file alone (a msa.fa some other process wrote)
    alignment.MSA_CACHE.clear()
    log = io.StringIO()
    tup = computeStats.outputRecallPrecision(d + "/cor.fa", d, log, small, wrong, 5, 0.1, "read_size_distribution.txt", {})
    check_report(tup, capsys.readouterr().out, log.getvalue(), d)


@pytest.mark.gpu
def test_report_without_the_msa_file(tmp_path, capsys, monkeypatch):
    """getPOA(write_msa=False) / ELECTOR_NO_MSA=1: no msa.fa, the same report -- tuple, stdout, log and
    both side files of the pinned reference chain -- from the device counters"""
    from elector_amd import alignment, computeStats
    for mode in ("argument", "environment"):
        d = str(tmp_path / mode)
        os.makedirs(d)
        write_reads(d, c1_reads())
        if mode == "argument":
            small, wrong = alignment.getPOA(d + "/cor.fa", d + "/ref.fa", d + "/unc.fa", 8, d, 0.1, write_msa=False)
        else:
            monkeypatch.setenv("ELECTOR_NO_MSA", "1")
            small, wrong = alignment.getPOA(d + "/cor.fa", d + "/ref.fa", d + "/unc.fa", 8, d, 0.1)
            monkeypatch.delenv("ELECTOR_NO_MSA")
        capsys.readouterr()
        assert (small, wrong) == (GOLD["small"], GOLD["wrong"])
        assert not os.path.exists(d + "/msa.fa")
        log = io.StringIO()
        tup = computeStats.outputRecallPrecision(d + "/cor.fa", d, log, small, wrong, 5, 0.1, "read_size_distribution.txt", {})
        check_report(tup, capsys.readouterr().out, log.getvalue(), d)
        assert not os.path.exists(d + "/msa.fa")
    # soft clips need the file's header lines: refused, not silently ignored
    with pytest.raises(ValueError):
        computeStats.cached_pieces(d + "/msa.fa", {"read1": (3, 4)})


def _rank(rank, world, port, d, result):
    import torch.distributed as dist
    from contextlib import redirect_stdout
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"                      # the box has one GPU: the ranks share it
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from elector_amd import alignment as al, computeStats as cs
    with redirect_stdout(io.StringIO()):
        small, wrong = al.getPOA(d + "/cor.fa", d + "/ref.fa", d + "/unc.fa", 4, d, 0.1)
    if rank == 0:
        log, buf = io.StringIO(), io.StringIO()
        hit = cs.cached_pieces(d + "/msa.fa", {}) is not None
        with redirect_stdout(buf):
            tup = cs.outputRecallPrecision(d + "/cor.fa", d, log, small, wrong, 5, 0.1, "read_size_distribution.txt", {})
        json.dump({"small": small, "wrong": wrong, "hit": hit, "tuple": json.loads(json.dumps(tup)),
                   "stdout": buf.getvalue(), "log": log.getvalue()}, open(result, "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_reproduce_the_reference_chain(tmp_path):
    import torch.multiprocessing as mp
    from portutil import free_port
    d = str(tmp_path)
    write_reads(d, c1_reads())
    res = d + "/two.json"
    mp.spawn(_rank, args=(2, free_port(), d, res), nprocs=2, join=True)
    two = json.load(open(res))
    assert (two["small"], two["wrong"]) == (GOLD["small"], GOLD["wrong"]) and two["hit"]
    check_msa(d + "/msa.fa")
    assert not [f for f in os.listdir(d) if ".part" in f]
    check_report(two["tuple"], two["stdout"], two["log"], d)


@pytest.mark.gpu
def test_two_ranks_without_the_msa_file(tmp_path, monkeypatch):
    """ELECTOR_NO_MSA=1 under a process group: no msa.fa and no part files, the same report on rank 0"""
    import torch.multiprocessing as mp
    from portutil import free_port
    d = str(tmp_path)
    write_reads(d, c1_reads())
    res = d + "/two.json"
    monkeypatch.setenv("ELECTOR_NO_MSA", "1")
    mp.spawn(_rank, args=(2, free_port(), d, res), nprocs=2, join=True)
    two = json.load(open(res))
    assert (two["small"], two["wrong"]) == (GOLD["small"], GOLD["wrong"]) and two["hit"]
    assert not os.path.exists(d + "/msa.fa") and not [f for f in os.listdir(d) if ".part" in f]
    check_report(two["tuple"], two["stdout"], two["log"], d)
