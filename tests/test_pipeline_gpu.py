"""Call site #1 end to end on the GPU: elector_amd.alignment.getPOA must write the
same msa.fa bytes (and return the same counters) as the reference chain
masterSplitter -> poa -> Donatello (golden vectors made by oracle/make_golden.py
with the real binaries), and feeding that file to the computeStats mirror must
agree with the statistics oracle."""
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import stats_oracle  # noqa: E402

from elector_amd import alignment, computeStats  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden", "pipeline_golden.json")


@pytest.mark.parametrize("idx", [0, 1])
def test_getpoa_writes_reference_msa(tmp_path, engine, capsys, idx):
    case = json.load(open(GOLD))[idx]
    for k in ("ref", "cor", "unc"):
        (tmp_path / (k + ".fa")).write_text(case[k])
    small, wrong = alignment.getPOA(str(tmp_path / "cor.fa"), str(tmp_path / "ref.fa"), str(tmp_path / "unc.fa"),
                                    4, str(tmp_path), 0.1, engine=engine)
    capsys.readouterr()
    assert (small, wrong) == (case["small"], case["wrong"])
    assert (tmp_path / "msa.fa").read_text() == case["msa"]
    # soft name variant + append semantics
    alignment.getPOA(str(tmp_path / "cor.fa"), str(tmp_path / "ref.fa"), str(tmp_path / "unc.fa"),
                     1, str(tmp_path), 0.1, soft="lordec", engine=engine)
    alignment.getPOA(str(tmp_path / "cor.fa"), str(tmp_path / "ref.fa"), str(tmp_path / "unc.fa"),
                     1, str(tmp_path), 0.1, soft="lordec", engine=engine)
    capsys.readouterr()
    assert (tmp_path / "msa_lordec.fa").read_text() == case["msa"] * 2
    # call site #2 on the file just written
    computeStats._engine = engine
    log = io.StringIO()
    try:
        exp_tuple, exp_out, exp_log, _, _ = stats_oracle.output_recall_precision(case["msa"], small, wrong, 5, 0.1)
    except KeyError:
        # a header with a title breaks the reference's own header bookkeeping
        # (computeStats.py:548 KeyError); the mirror must fail the same way
        with pytest.raises(KeyError):
            computeStats.outputRecallPrecision(str(tmp_path / "cor.fa"), str(tmp_path), log, small, wrong, 5, 0.1,
                                               "sizes.txt", {})
        return
    tup = computeStats.outputRecallPrecision(str(tmp_path / "cor.fa"), str(tmp_path), log, small, wrong, 5, 0.1,
                                             "sizes.txt", {})
    out = capsys.readouterr().out
    assert tup == exp_tuple and out == "None\n" + exp_out and log.getvalue() == exp_log


def _write_reads(tmp_path, reads):
    for fn, k in (("ref.fa", 1), ("cor.fa", 2), ("unc.fa", 3)):
        with open(tmp_path / fn, "wb") as f:
            for r in reads:
                f.write(r[0] + b"\n" + r[k] + b"\n")


@pytest.mark.parametrize("name", ["slots", "batchcut"])
def test_getpoa_protocol_edges(tmp_path, engine, capsys, name):
    """a13 edges against the real chain (tests/golden/pipeline_edges.json, oracle/make_golden.py): more reads
    than a slot file holds with identical header lines inside a slot and across slot boundaries
    (Master_Splitter.cpp:366-369,435, Donatello.cpp:61-84), and more reads than one masterSplitter batch with
    identical header lines across the cut (:397-399)."""
    import hashlib
    import msa_gen
    gold = [g for g in json.load(open(os.path.join(ROOT, "tests", "golden", "pipeline_edges.json"))) if g["name"] == name][0]
    reads = msa_gen.edge_reads_slots() if name == "slots" else msa_gen.edge_reads_batchcut()
    assert len(reads) == gold["n_reads"]
    _write_reads(tmp_path, reads)
    small, wrong = alignment.getPOA(str(tmp_path / "cor.fa"), str(tmp_path / "ref.fa"), str(tmp_path / "unc.fa"),
                                    8, str(tmp_path), 0.1, engine=engine)
    capsys.readouterr()
    assert (small, wrong) == (gold["small"], gold["wrong"])
    txt = (tmp_path / "msa.fa").read_text()
    lines = txt.split("\n")
    assert sum(1 for ln in lines if ln.startswith(">")) // 3 == gold["msa_records"]
    for i, ln in gold["edge_lines"].items():
        assert lines[int(i)] == ln, (i, lines[int(i)], ln)
        assert hashlib.sha256(lines[int(i) + 1].encode()).hexdigest() == gold["edge_rows_sha256"][i], i
    assert len(txt) == gold["msa_bytes"]
    assert hashlib.sha256(txt.encode()).hexdigest() == gold["msa_sha256"]


def test_device_counters_reach_call_site_2(tmp_path, engine, capsys):
    """getPOA leaves the per-piece counters the device computed on the way (merged MSAs never left HBM) for
    outputRecallPrecision: the report from them must be the report from parsing msa.fa again, and the
    statistics oracle's."""
    import msa_gen
    from elector_amd import synthetic
    reads = msa_gen.make_reads(91, 40, 1200)              # plain, trimmed, split, extended, homopolymer-rich
    _write_reads(tmp_path, reads)
    small, wrong = alignment.getPOA(str(tmp_path / "cor.fa"), str(tmp_path / "ref.fa"), str(tmp_path / "unc.fa"),
                                    8, str(tmp_path), 0.1, engine=engine)
    capsys.readouterr()
    msa = str(tmp_path / "msa.fa")
    assert computeStats.cached_pieces(msa, {}) is not None
    computeStats._engine = engine
    log1 = io.StringIO()
    tup1 = computeStats.outputRecallPrecision(str(tmp_path / "cor.fa"), str(tmp_path), log1, small, wrong, 5, 0.1, "sizes.txt", {})
    out1 = capsys.readouterr().out
    per1 = (tmp_path / "per_read_metrics.txt").read_text()
    alignment.MSA_CACHE.clear()                            # second time: the text file is parsed
    assert computeStats.cached_pieces(msa, {}) is None
    log2 = io.StringIO()
    tup2 = computeStats.outputRecallPrecision(str(tmp_path / "cor.fa"), str(tmp_path), log2, small, wrong, 5, 0.1, "sizes.txt", {})
    out2 = capsys.readouterr().out
    assert tup1 == tup2 and out1 == out2 and log1.getvalue() == log2.getvalue()
    assert per1 == (tmp_path / "per_read_metrics.txt").read_text()
    exp_tuple, exp_out, exp_log, _, _ = stats_oracle.output_recall_precision((tmp_path / "msa.fa").read_text(), small, wrong, 5, 0.1)
    assert tup1 == exp_tuple and out1 == "None\n" + exp_out and log1.getvalue() == exp_log
    # a file that changed since getPOA wrote it is never served from the cache
    alignment.getPOA(str(tmp_path / "cor.fa"), str(tmp_path / "ref.fa"), str(tmp_path / "unc.fa"), 8, str(tmp_path), 0.1,
                     engine=engine)                        # appends: msa.fa now holds every record twice
    capsys.readouterr()
    assert computeStats.cached_pieces(msa, {}) is None


def test_getpoa_skip_mode(tmp_path, engine, capsys):
    """A window the device refuses (here: an empty corrected end piece is impossible to build from files, so a
    sequence beyond ELECTOR_MAX_SEQ in an un-anchored window): parity="raise" raises, parity="skip" leaves the
    record out and keeps the rest."""
    from elector_amd import _capi
    import numpy as np
    rng = np.random.default_rng(3)
    import synth
    good = synth.random_seq(rng, 600)
    # no k-mer is unique in a long homopolymer run: the splitter finds no anchor and hands the whole read over
    # as one window (Master_Splitter.cpp:256-261) -- which also makes it a "wrong" read (one window) replaced by
    # a dummy; so build a read with exactly two anchors around a run longer than the device limit
    run = b"A" * (_capi.ELECTOR_MAX_SEQ + 200)
    left, right = synth.random_seq(rng, 300), synth.random_seq(rng, 300)
    big = left + run + right
    reads = [(b">ok1_0", good, synth.mutate(rng, good, 0.02), synth.mutate(rng, good, 0.1)),
             (b">big_0", big, big, big),
             (b">ok2_0", good, synth.mutate(rng, good, 0.02), synth.mutate(rng, good, 0.1))]
    _write_reads(tmp_path, reads)
    with pytest.raises(_capi.ElectorError):
        alignment.getPOA(str(tmp_path / "cor.fa"), str(tmp_path / "ref.fa"), str(tmp_path / "unc.fa"), 4, str(tmp_path), 0.1,
                         engine=engine)
    capsys.readouterr()
    os.remove(tmp_path / "msa.fa")
    alignment.getPOA(str(tmp_path / "cor.fa"), str(tmp_path / "ref.fa"), str(tmp_path / "unc.fa"), 4, str(tmp_path), 0.1,
                     engine=engine, parity="skip")
    assert "1 records left out" in capsys.readouterr().out
    hdrs = [ln for ln in (tmp_path / "msa.fa").read_text().split("\n") if ln.startswith(">")]
    assert hdrs == [">ok1 "] * 3 + [">ok2 "] * 3


def _rank_getpoa(rank, world, port, d, result):
    """one rank of the multi-rank getPOA: gloo rendezvous, all ranks on this box's GPU"""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from elector_amd import alignment as al, computeStats as cs
    from contextlib import redirect_stdout
    with redirect_stdout(io.StringIO()):
        small, wrong = al.getPOA(d + "/cor.fa", d + "/ref.fa", d + "/unc.fa", 4, d + "/out2", 0.1)
    if rank == 0:
        log = io.StringIO()
        buf = io.StringIO()
        hit = cs.cached_pieces(d + "/out2/msa.fa", {}) is not None
        with redirect_stdout(buf):
            tup = cs.outputRecallPrecision(d + "/cor.fa", d + "/out2", log, small, wrong, 5, 0.1, "sizes.txt", {})
        json.dump({"small": small, "wrong": wrong, "hit": hit, "tuple": json.loads(json.dumps(tup)), "log": log.getvalue()},
                  open(result, "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_getpoa_two_ranks_equal_one(tmp_path, engine, capsys):
    """WORLD_SIZE = 2 (gloo here, RCCL on the 8-GPU node): each rank aligns its own contiguous range of reads,
    rank 0 ends up with the same msa.fa bytes, counters and report as a single process."""
    import msa_gen
    import torch.multiprocessing as mp
    reads = msa_gen.make_reads(93, 60, 900) + msa_gen.edge_reads_slots(55)
    _write_reads(tmp_path, reads)
    (tmp_path / "out1").mkdir()
    (tmp_path / "out2").mkdir()
    small, wrong = alignment.getPOA(str(tmp_path / "cor.fa"), str(tmp_path / "ref.fa"), str(tmp_path / "unc.fa"), 8,
                                    str(tmp_path / "out1"), 0.1, engine=engine)
    computeStats._engine = engine
    log = io.StringIO()
    tup = computeStats.outputRecallPrecision(str(tmp_path / "cor.fa"), str(tmp_path / "out1"), log, small, wrong, 5, 0.1,
                                             "sizes.txt", {})
    capsys.readouterr()
    res = str(tmp_path / "two.json")
    from portutil import free_port
    mp.spawn(_rank_getpoa, args=(2, free_port(), str(tmp_path), res), nprocs=2, join=True)
    two = json.load(open(res))
    assert (tmp_path / "out2" / "msa.fa").read_bytes() == (tmp_path / "out1" / "msa.fa").read_bytes()
    assert not [f for f in os.listdir(tmp_path / "out2") if ".part" in f]
    assert (two["small"], two["wrong"]) == (small, wrong) and two["hit"]
    assert two["tuple"] == json.loads(json.dumps(tup)) and two["log"] == log.getvalue()


def test_getpoa_two_ranks_into_a_used_directory(tmp_path, engine, capsys):
    """Append semantics with several ranks (Donatello.cpp:48 appends): msa.fa already holds records and a part
    file of an aborted run lies around.  The new records are appended once, the stale part is not copied, and
    call site #2 reports on the WHOLE file (the device counters only cover this run's records)."""
    import msa_gen
    import torch.multiprocessing as mp
    from portutil import free_port
    reads = msa_gen.make_reads(95, 24, 800)
    _write_reads(tmp_path, reads)
    (tmp_path / "out1").mkdir()
    (tmp_path / "out2").mkdir()
    for _ in range(2):                                     # one process, run twice: the reference's own behaviour
        small, wrong = alignment.getPOA(str(tmp_path / "cor.fa"), str(tmp_path / "ref.fa"), str(tmp_path / "unc.fa"), 8,
                                        str(tmp_path / "out1"), 0.1, engine=engine)
    computeStats._engine = engine
    log = io.StringIO()
    tup = computeStats.outputRecallPrecision(str(tmp_path / "cor.fa"), str(tmp_path / "out1"), log, small, wrong, 5, 0.1,
                                             "sizes.txt", {})
    capsys.readouterr()
    once = (tmp_path / "out1" / "msa.fa").read_bytes()[: (tmp_path / "out1" / "msa.fa").stat().st_size // 2]
    (tmp_path / "out2" / "msa.fa").write_bytes(once)
    (tmp_path / "out2" / "msa.fa.part1").write_bytes(b">stale \nacgt\n>stale \nacgt\n>stale \nacgt\n")
    res = str(tmp_path / "two.json")
    mp.spawn(_rank_getpoa, args=(2, free_port(), str(tmp_path), res), nprocs=2, join=True)
    two = json.load(open(res))
    assert (tmp_path / "out2" / "msa.fa").read_bytes() == once + once == (tmp_path / "out1" / "msa.fa").read_bytes()
    assert not [f for f in os.listdir(tmp_path / "out2") if ".part" in f]
    assert not two["hit"]                                   # the file held foreign records: no counters from the cache
    assert two["tuple"] == json.loads(json.dumps(tup)) and two["log"] == log.getvalue()


def test_getpoa_four_ranks_empty_shard_and_dominant_read(tmp_path, engine, capsys):
    """World size 4 with a read that outweighs all others together (a shard of its own, another rank's shard
    empty) and a split read at a shard boundary: same msa.fa bytes, counters and report as a single process."""
    import msa_gen
    import numpy as np
    import synth
    import torch.multiprocessing as mp
    from portutil import free_port
    rng = np.random.default_rng(97)
    big = synth.random_seq(rng, 60000)
    reads = [(b">big_0", big, synth.mutate(rng, big, 0.01), synth.mutate(rng, big, 0.12))] + msa_gen.make_reads(96, 12, 700)
    _write_reads(tmp_path, reads)
    bounds = alignment._shard(str(tmp_path / "ref.fa"), str(tmp_path / "unc.fa"), str(tmp_path / "cor.fa"), 4)
    assert any(a == b for a, b in zip(bounds, bounds[1:])) and (0, 1) in set(zip(bounds, bounds[1:]))
    (tmp_path / "out1").mkdir()
    (tmp_path / "out2").mkdir()
    small, wrong = alignment.getPOA(str(tmp_path / "cor.fa"), str(tmp_path / "ref.fa"), str(tmp_path / "unc.fa"), 8,
                                    str(tmp_path / "out1"), 0.1, engine=engine)
    computeStats._engine = engine
    log = io.StringIO()
    tup = computeStats.outputRecallPrecision(str(tmp_path / "cor.fa"), str(tmp_path / "out1"), log, small, wrong, 5, 0.1,
                                             "sizes.txt", {})
    capsys.readouterr()
    res = str(tmp_path / "four.json")
    mp.spawn(_rank_getpoa, args=(4, free_port(), str(tmp_path), res), nprocs=4, join=True)
    four = json.load(open(res))
    assert (tmp_path / "out2" / "msa.fa").read_bytes() == (tmp_path / "out1" / "msa.fa").read_bytes()
    assert (four["small"], four["wrong"]) == (small, wrong) and four["hit"]
    assert four["tuple"] == json.loads(json.dumps(tup)) and four["log"] == log.getvalue()


def test_getpoa_mixed_lengths_device_split_equals_host_split(tmp_path, engine, capsys, monkeypatch):
    """Reads from 3 kb to 40 kb in one batch (the device splitter takes them in three launches: small on-chip
    tables, medium and long partitioned tables; some corrected reads trimmed, so re-splits too): msa.fa and the
    counters must not depend on whether the windows were cut on the device or by the host splitter."""
    import hashlib
    import numpy as np
    import synth
    rng = np.random.default_rng(41)
    reads = []
    for i, n in enumerate((3000, 9000, 12400, 12600, 15000, 21000, 32000, 33000, 40000, 7000, 18000, 5000)):
        r = synth.random_seq(rng, n)
        c = synth.mutate(rng, r, 0.01)
        if i % 4 == 1:
            c = c[len(c) // 5:]
        elif i % 4 == 2:
            c = c[: 3 * len(c) // 4]
        reads.append((b">read_%d" % i, r, c, synth.mutate(rng, r, 0.13)))
    got = {}
    for mode in ("device", "host"):
        d = tmp_path / mode
        d.mkdir()
        _write_reads(d, reads)
        if mode == "host":
            monkeypatch.setenv("ELECTOR_HOST_SPLIT", "1")
        small, wrong = alignment.getPOA(str(d / "cor.fa"), str(d / "ref.fa"), str(d / "unc.fa"), 8, str(d), 0.1)
        capsys.readouterr()
        with open(d / "msa.fa", "rb") as f:
            got[mode] = (small, wrong, hashlib.sha256(f.read()).hexdigest(), os.path.getsize(d / "msa.fa"))
    monkeypatch.delenv("ELECTOR_HOST_SPLIT", raising=False)
    assert got["device"] == got["host"]
    assert got["device"][3] > 0
