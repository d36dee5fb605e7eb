"""Statistics kernel on the GPU (through the C ABI) against the Python oracle and
the golden vectors of the real reference module.  Integers: bit-exact."""
import io
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import stats_oracle  # noqa: E402

import msa_gen  # noqa: E402
from test_stats_cpu import GOLD, oracle_counter_array  # noqa: E402
from elector_amd import computeStats as cs  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,n,L", [(21, 40, 900), (22, 24, 2500), (23, 60, 400)])
def test_counters_equal_oracle(tmp_path, engine, seed, n, L):
    reads = msa_gen.make_reads(seed, n, L)
    txt, _, _ = msa_gen.msa_text(reads)
    path = tmp_path / "msa.fa"
    path.write_text(txt)
    pieces = cs.parse_msa(str(path), cs.getSplit(str(path)))
    res, oracle_pieces = stats_oracle.compute_metrics(txt, 5)
    exp = oracle_counter_array(pieces, oracle_pieces)
    got, last_mask = cs.stats_counters(pieces, None, engine)
    proc = got[:, cs.ES_PROCESSED] == 1
    assert np.array_equal(proc, exp[:, cs.ES_PROCESSED] == 1)
    assert np.array_equal(got[proc], exp[proc]), np.argwhere(got[proc] != exp[proc])[:5]
    assert cs.homopolymer_ratios(pieces, last_mask, 5) == res["lastReadRatios"]


def test_full_report_equals_reference_golden(tmp_path, engine, capsys):
    cs._engine = engine
    for i, case in enumerate(json.load(open(GOLD))):
        d = tmp_path / ("c%d" % i)
        d.mkdir()
        (d / "msa.fa").write_text(case["msa"])
        (d / "cor.fa").write_text(">x\nACGT\n")
        log = io.StringIO()
        tup = cs.outputRecallPrecision(str(d / "cor.fa"), str(d), log, case["small"], case["wrong"], 5, 0.1,
                                       "sizes.txt", {k: tuple(v) for k, v in case["clips"].items()})
        assert json.loads(json.dumps(tup)) == case["tuple"]
        assert capsys.readouterr().out == case["stdout"]
        assert log.getvalue() == case["log"]
        assert (d / "per_read_metrics.txt").read_text() == case["per_read"]


def test_stats_edge_cases(tmp_path, engine):
    """Short pieces (<= 10 columns, skipped), all-gap ends, a read whose last piece is short."""
    rows = [("r1", "acgtacgtacgtacgtacgtacgtacgtacgt", "acgtacgtacgtacgaacgtacgtacgtacgt", "acgtacgtacg.acgtacgtacgtacgtacgt"),
            ("r2", "aaa", "aaa", "aaa"),
            ("r3", "." * 30 + "acgtacgtacgtacgtacgtacgtacgt" * 3, "ggg" + "." * 27 + "acgtacgtacgtacgtacgtacgtacgt" * 3,
             "." * 30 + "acgtacgtacgtacgtacgtacgtacgt" * 3),
            ("r4", "acgtacgtacgtacgtacgtacgtacgtacgtacgt" * 3, "." * 60 + "acgtacgtacgtacgtacgtacgtacgtacgtacgtacgtacgtacgt",
             "acgtacgtacgtacgtacgtacgtacgtacgtacgt" * 3),
            ("r4", "acgtac", "acgtac", "acgtac")]
    txt = "".join(">%s \n%s\n>%s \n%s\n>%s \n%s\n" % (h, a, h, b, h, c) for h, a, b, c in rows)
    path = tmp_path / "msa.fa"
    path.write_text(txt)
    pieces = cs.parse_msa(str(path), cs.getSplit(str(path)))
    res, oracle_pieces = stats_oracle.compute_metrics(txt, 5)
    exp = oracle_counter_array(pieces, oracle_pieces)
    got, _ = cs.stats_counters(pieces, None, engine)
    proc = exp[:, cs.ES_PROCESSED] == 1
    assert np.array_equal(got[:, cs.ES_PROCESSED], exp[:, cs.ES_PROCESSED])
    assert np.array_equal(got[proc], exp[proc])


@pytest.mark.parametrize("seed,n,L,merge", [(31, 40, 900, None), (32, 16, 3000, None), (33, 40, 900, "0"), (34, 12, 4000, "0"),
                                            (35, 30, 900, "1")])
def test_device_merge_and_stats_equal_oracle(tmp_path, engine, monkeypatch, seed, n, L, merge):
    """Windows stay on the GPU from the POA kernels to the counters (elector_msa_stats_device):
    merged records, column counts and counters must equal the oracle chain's msa.fa and statistics.
    merge: ELECTOR_MERGE_PER_PIECE -- "1" a block per piece (what batches of short pieces get), "0" a lane group per
    window, the rows packed as they are written and sent to the host from where they lie (no packing pass); None: the
    library's choice."""
    import torch
    if merge is not None:
        monkeypatch.setenv("ELECTOR_MERGE_PER_PIECE", merge)
    from elector_amd import split
    reads = msa_gen.make_reads(seed, n, L)
    txt, _, _ = msa_gen.msa_text(reads)
    path = tmp_path / "msa.fa"
    path.write_text(txt)
    pieces = cs.parse_msa(str(path), cs.getSplit(str(path)))
    _, oracle_pieces = stats_oracle.compute_metrics(txt, 5)
    exp = oracle_counter_array(pieces, oracle_pieces)

    hdrs = [h for (h, _, _, _) in reads]
    win = split.split_reads([(r, c, u) for (_, r, c, u) in reads], 0.1, hdrs, nthreads=2)
    assert win.n_reads == len(pieces.cols)
    dev = torch.device("cuda", 0)
    d_bases = torch.from_numpy(win.bases).to(dev)
    d_cols = torch.empty(3 * int(win.off[-1]) + 64, dtype=torch.uint8, device=dev)
    d_ncol = torch.empty(win.n_windows, dtype=torch.int32, device=dev)
    d_status = torch.empty(win.n_windows, dtype=torch.int32, device=dev)
    engine.align_device(d_bases, win.off, d_cols, d_ncol, d_status)
    last_cap = int(pieces.cols[pieces.read_first[-2]:].sum()) + 8
    got, piece_cols, last_rows, last_mask = engine.msa_stats_device(
        win.n_windows, d_cols, d_ncol, d_status, win.read_first, pieces.read_first, None, last_cap)
    assert np.array_equal(piece_cols, pieces.cols)
    rows = engine.msa_rows_fetch(piece_cols)
    assert np.array_equal(rows, pieces.rows)
    # the same records delivered inside the queue (elector_msa_stats_enqueue_rows): to page-locked host memory (the
    # one copy on the context's copy stream) at an odd address, to device memory, and with the windows taken from DEVICE offsets
    cap = 3 * int(win.off[-1]) + 64
    pinned = torch.zeros(cap + 16, dtype=torch.uint8).pin_memory()
    d_off = torch.from_numpy(np.ascontiguousarray(win.off)).to(dev)
    engine.align_device_offsets(d_bases, d_off, win.n_windows, int(win.off[-1]), d_cols, d_ncol, d_status)
    npc = engine.msa_stats_enqueue(win.n_windows, d_cols, d_ncol, d_status, win.read_first, pieces.read_first,
                                   rows_out=pinned.data_ptr() + 3, rows_cap=cap)
    got2, piece_cols2 = engine.msa_stats_collect(npc)
    assert np.array_equal(piece_cols2, pieces.cols) and np.array_equal(got2, got)
    engine.msa_rows_wait()
    assert np.array_equal(pinned.numpy()[3:3 + len(pieces.rows)], pieces.rows)
    # ... and with the copy started when the job is collected (rounds 3-4) instead of from the stream's host function,
    # twice in a row into the other slot and a fresh destination
    for env in ("ELECTOR_ROWS_AT_COLLECT", "ELECTOR_ROWS_HIP_COPY"):
        os.environ[env] = "1"
        try:
            pinned2 = torch.zeros(cap + 16, dtype=torch.uint8).pin_memory()
            npc = engine.msa_stats_enqueue(win.n_windows, d_cols, d_ncol, d_status, win.read_first, pieces.read_first,
                                           rows_out=pinned2.data_ptr() + 1, rows_cap=cap)
            engine.msa_stats_collect(npc)
            engine.msa_rows_wait()
            assert np.array_equal(pinned2.numpy()[1:1 + len(pieces.rows)], pieces.rows), env
        finally:
            del os.environ[env]
    d_rows = torch.zeros(cap, dtype=torch.uint8, device=dev)
    npc = engine.msa_stats_enqueue(win.n_windows, d_cols, d_ncol, d_status, win.read_first, pieces.read_first,
                                   rows_out=d_rows.data_ptr(), rows_cap=cap)
    engine.msa_stats_collect(npc)
    assert np.array_equal(d_rows.cpu().numpy()[:len(pieces.rows)], pieces.rows)
    with pytest.raises(Exception):
        engine.msa_stats_enqueue(win.n_windows, d_cols, d_ncol, d_status, win.read_first, pieces.read_first,
                                 rows_out=pinned.data_ptr(), rows_cap=16)
    proc = exp[:, cs.ES_PROCESSED] == 1
    assert np.array_equal(got[:, cs.ES_PROCESSED], exp[:, cs.ES_PROCESSED])
    assert np.array_equal(got[proc], exp[proc]), np.argwhere(got[proc] != exp[proc])[:5]
    _, host_mask = cs.stats_counters(pieces, None, engine)
    assert np.array_equal(last_mask, host_mask)
    assert np.array_equal(last_rows, pieces.rows[pieces.row_off[pieces.read_first[-2]]:])


def test_two_statistics_jobs_in_flight(tmp_path, engine):
    """elector_msa_stats_enqueue/_collect: batch B is aligned and queued while batch A's job is
    still uncollected; both must come out as the oracle's counters, a third enqueue is refused."""
    import torch
    from elector_amd import split
    from elector_amd._capi import ElectorError
    dev = torch.device("cuda", 0)
    jobs = []
    for seed, n, L in [(41, 12, 700), (42, 20, 500)]:
        reads = msa_gen.make_reads(seed, n, L)
        txt, _, _ = msa_gen.msa_text(reads)
        path = tmp_path / ("msa%d.fa" % seed)
        path.write_text(txt)
        pieces = cs.parse_msa(str(path), cs.getSplit(str(path)))
        _, oracle_pieces = stats_oracle.compute_metrics(txt, 5)
        exp = oracle_counter_array(pieces, oracle_pieces)
        win = split.split_reads([(r, c, u) for (_, r, c, u) in reads], 0.1, [h for (h, _, _, _) in reads], nthreads=2)
        d_bases = torch.from_numpy(win.bases).to(dev)
        d_cols = torch.empty(3 * int(win.off[-1]) + 64, dtype=torch.uint8, device=dev)
        d_ncol = torch.empty(win.n_windows, dtype=torch.int32, device=dev)
        d_status = torch.empty(win.n_windows, dtype=torch.int32, device=dev)
        engine.align_device(d_bases, win.off, d_cols, d_ncol, d_status)
        npieces = engine.msa_stats_enqueue(win.n_windows, d_cols, d_ncol, d_status, win.read_first, pieces.read_first)
        jobs.append((npieces, exp, pieces, (d_bases, d_cols, d_ncol, d_status, win)))
    d_bases, d_cols, d_ncol, d_status, win = jobs[-1][3]
    with pytest.raises(ElectorError):
        engine.msa_stats_enqueue(win.n_windows, d_cols, d_ncol, d_status, win.read_first, jobs[-1][2].read_first)
    for npieces, exp, pieces, _ in jobs:
        got, piece_cols = engine.msa_stats_collect(npieces)
        assert np.array_equal(piece_cols, pieces.cols)
        proc = exp[:, cs.ES_PROCESSED] == 1
        assert np.array_equal(got[:, cs.ES_PROCESSED], exp[:, cs.ES_PROCESSED])
        assert np.array_equal(got[proc], exp[proc])
    with pytest.raises(ElectorError):
        engine.msa_stats_collect(jobs[-1][0])


def test_stats_pool_growth(tmp_path, engine):
    """Many runs of corrected gaps: the interval scratch outgrows the first pool size and the
    library has to run the kernel again with the worst-case pool."""
    rng = np.random.default_rng(5)
    n = 6_000_000
    ref = rng.choice(np.frombuffer(b"acgt", dtype=np.uint8), n)
    cor = ref.copy()
    for s in range(0, n - 12, 12):
        cor[s:s + 6] = ord(".")
    cor[:40] = ord(".")
    unc = ref.copy()
    txt = ">r1 \n%s\n>r1 \n%s\n>r1 \n%s\n" % (ref.tobytes().decode(), cor.tobytes().decode(), unc.tobytes().decode())
    path = tmp_path / "msa.fa"
    path.write_text(txt)
    pieces = cs.parse_msa(str(path), cs.getSplit(str(path)))
    got, _ = cs.stats_counters(pieces, None, engine)
    # all letters of cor agree with ref, the gap columns are deletions outside the masked left end
    assert got[0, cs.ES_PROCESSED] == 1 and got[0, cs.ES_SUB_C] == 0 and got[0, cs.ES_INS_C] == 0
    assert got[0, cs.ES_LEN_REF] == n and got[0, cs.ES_LEN_COR] == int((cor != ord(".")).sum())


def test_long_gap_runs_block_walk(tmp_path, engine):
    """Runs of corrected gaps longer than one lane walks (trimmed / split pieces): reference-gap streaks of
    18-22 columns inside the run (the THRESH2 edge), a run that starts the record, a run entered with a
    running reference-gap count, runs that reach the record's end."""
    rng = np.random.default_rng(9)
    acgt = np.frombuffer(b"acgt", dtype=np.uint8)

    def letters(k):
        return acgt[rng.integers(0, 4, k)].tobytes().decode()

    recs = []
    # 1: long run in the middle, reference-gap streaks of every length around THRESH2 inside it
    ref = letters(60)
    cor = ref
    for k in (3, 17, 18, 19, 20, 21, 22, 40):
        ref += letters(37) + "." * k
        cor += "." * (37 + k)
    ref += letters(300) + letters(80)
    cor += "." * 300 + ref[-80:]
    recs.append(("r1", ref, cor, ref.replace(".", "a")))
    # 2: the run starts the record, with reference gaps at its very first columns
    ref = "." * 7 + letters(400) + "." * 25 + letters(200)
    cor = "." * 500 + ref[500:]
    recs.append(("r2", ref, cor, ref.replace(".", "c")))
    # 3: the run is entered with a running count: reference gaps (under corrected letters, then gaps) just before it
    ref = letters(50) + "." * 12 + "." * 9 + letters(600)
    cor = ref[:50] + letters(4) + "." * 3 + letters(5) + "." * 9 + "." * 500 + ref[-100:]
    recs.append(("r3", ref, cor, ref.replace(".", "g")))
    # 4: run to the end of the record
    ref = letters(100) + letters(900)
    cor = ref[:100] + "." * 900
    recs.append(("r4", ref, cor, ref))
    for h, a, b, c in recs:
        assert len(a) == len(b) == len(c), h
    txt = "".join(">%s \n%s\n>%s \n%s\n>%s \n%s\n" % (h, a, h, b, h, c) for h, a, b, c in recs)
    path = tmp_path / "msa.fa"
    path.write_text(txt)
    pieces = cs.parse_msa(str(path), cs.getSplit(str(path)))
    _, oracle_pieces = stats_oracle.compute_metrics(txt, 5)
    exp = oracle_counter_array(pieces, oracle_pieces)
    got, _ = cs.stats_counters(pieces, None, engine)
    assert np.array_equal(got, exp), np.argwhere(got != exp)


def _gappy_records(seed, sizes, split_every=3):
    """Three-row records made directly (no POA): gap runs of every length in all rows, gap stretches at the
    ends, many short corrected-gap runs (more than the kernel keeps in LDS), pieces of split reads."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"acgt", dtype=np.uint8)
    dot = ord(".")

    def runs(row, n, count, lens):
        for _ in range(count):
            k = int(rng.choice(lens))
            s = int(rng.integers(0, max(1, n - k)))
            row[s:s + k] = dot

    recs, clips = [], {}
    for idx, n in enumerate(sizes):
        ref = acgt[rng.integers(0, 4, n)].copy()
        cor = ref.copy()
        unc = ref.copy()
        sub = rng.random(n) < 0.02
        cor[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
        sub = rng.random(n) < 0.12
        unc[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
        runs(ref, n, n // 40, [1, 1, 2, 3, 5, 8, 19, 20, 21, 30])
        runs(unc, n, n // 60, [1, 2, 4, 6, 25])
        runs(cor, n, n // (25 if idx % 2 else 120), [1, 2, 4, 5, 6, 7, 9, 22, 40])
        if idx % 4 == 1 and n >= 4000:
            runs(cor, n, 3, [250, 700, 1500])
        kind = idx % 5 if n >= 400 else 0
        if kind == 1:
            cor[: n // 5] = dot                                   # trimmed left
        elif kind == 2:
            cor[n - n // 4:] = dot                                # trimmed right
        elif kind == 3:
            ref[:37] = dot; unc[:41] = dot                        # corrected read extends to the left
            ref[n - 33:] = dot; unc[n - 29:] = dot
        name = "r%d" % (idx // split_every if idx % (2 * split_every) >= split_every else 1000 + idx)
        recs.append((name, ref.tobytes().decode(), cor.tobytes().decode(), unc.tobytes().decode()))
        if idx % 3 == 0:
            clips[">" + name + " "] = (int(rng.integers(0, 60)), int(rng.integers(0, 60)))
    return recs, clips


@pytest.mark.parametrize("seed,sizes", [(61, [700, 90, 3000, 12, 5000, 64, 128, 4097, 2048, 11, 1500, 333]),
                                         (62, [20000, 70000, 9000, 30000, 8000, 2500])])
def test_bit_rows_path_equals_hbm_path_and_oracle(tmp_path, engine, monkeypatch, seed, sizes):
    """k_stats on bit rows in LDS (the default) against the same kernel reading letters from HBM throughout
    (ELECTOR_STATS_BITWORDS=0, also what pieces too long for LDS get) and against the oracle; with a small LDS
    budget both paths meet inside one launch."""
    recs, clips = _gappy_records(seed, sizes)
    txt = "".join(">%s \n%s\n>%s \n%s\n>%s \n%s\n" % (h, a, h, b, h, c) for h, a, b, c in recs)
    path = tmp_path / "msa.fa"
    path.write_text(txt)
    pieces = cs.parse_msa(str(path), cs.getSplit(str(path)))
    for use_clips in (None, clips):
        _, oracle_pieces = stats_oracle.compute_metrics(txt, 5, use_clips)
        exp = oracle_counter_array(pieces, oracle_pieces)
        proc = exp[:, cs.ES_PROCESSED] == 1
        results = []
        for words in (None, "0", "40"):
            if words is None:
                monkeypatch.delenv("ELECTOR_STATS_BITWORDS", raising=False)
            else:
                monkeypatch.setenv("ELECTOR_STATS_BITWORDS", words)
            got, last_mask = cs.stats_counters(pieces, use_clips, engine)
            results.append((got, last_mask))
            assert np.array_equal(got[:, cs.ES_PROCESSED], exp[:, cs.ES_PROCESSED])
            assert np.array_equal(got[proc], exp[proc]), (words, np.argwhere(got[proc] != exp[proc])[:8])
        for got, last_mask in results[1:]:
            assert np.array_equal(got, results[0][0])
            assert np.array_equal(last_mask, results[0][1])
    monkeypatch.delenv("ELECTOR_STATS_BITWORDS", raising=False)
