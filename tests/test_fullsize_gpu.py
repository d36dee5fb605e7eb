"""Parity at a size the oracle would need minutes for, through size-independent properties: a 2,000-read
slice of the workload bench.py times (E. coli 30X SimLord-like reads, ~280k windows; the whole 10,001-read
batch and the other BASELINE.json configurations are tests/test_configs_gpu.py) goes through the device entry
points, then
  * every MSA row, with its gaps removed, is exactly the window's input sequence (a round trip that
    any misplaced, lost or duplicated letter breaks), and the column counts are consistent;
  * a second run over the same batch gives byte-identical columns (no run-to-run variation from the
    concurrent launch classes);
  * a seeded sample of windows of every size class is bit-exact against the CPU oracle."""
import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu


def ranges_gather(starts, lengths):
    """indices of the concatenation of [starts[i], starts[i] + lengths[i])"""
    total = int(lengths.sum())
    ends = np.cumsum(lengths)
    base = np.repeat(starts - (ends - lengths), lengths)
    return base + np.arange(total, dtype=np.int64)


def test_bench_batch_round_trip(engine):
    import torch
    from elector_amd import split, synthetic
    reads = synthetic.read_triples("ecoli30x_simlord_lordec", 2000, seed=4242)
    win = split.split_reads(reads, 0.1, None, nthreads=16)
    off, n = win.off, win.n_windows
    dev = torch.device("cuda", 0)
    d_bases = torch.from_numpy(win.bases).to(dev)
    d_cols = torch.zeros(3 * int(off[-1]) + 64, dtype=torch.uint8, device=dev)
    d_ncol = torch.empty(n, dtype=torch.int32, device=dev)
    d_status = torch.empty(n, dtype=torch.int32, device=dev)
    engine.align_device(d_bases, off, d_cols, d_ncol, d_status)
    engine.sync()
    cols1 = d_cols.cpu().numpy().copy()
    ncol = d_ncol.cpu().numpy().astype(np.int64)
    assert not d_status.cpu().numpy().any()

    lens = np.diff(off)
    lr, lc, lu = lens[0::3], lens[1::3], lens[2::3]
    assert (ncol >= np.maximum(np.maximum(lr, lc), lu)).all() and (ncol <= lr + lc + lu).all()
    msa = cols1[ranges_gather(3 * off[0:-1:3], 3 * ncol)].reshape(-1, 3)
    lower = np.frombuffer(bytes(win.bases).lower(), dtype=np.uint8)
    for r in range(3):
        got = msa[:, r][msa[:, r] != ord(".")]
        exp = lower[ranges_gather(off[r:-1:3], lens[r::3])]
        assert got.shape == exp.shape and np.array_equal(got, exp), "row %d does not spell its input" % r

    # idempotence
    d_cols.zero_()
    engine.align_device(d_bases, off, d_cols, d_ncol, d_status)
    engine.sync()
    used = ranges_gather(3 * off[0:-1:3], 3 * ncol)
    assert np.array_equal(d_cols.cpu().numpy()[used], cols1[used])
    assert np.array_equal(d_ncol.cpu().numpy(), ncol)

    # seeded sample against the oracle, stratified by window size
    rng = np.random.default_rng(7)
    order = np.argsort(lu, kind="stable")
    pick = np.unique(np.concatenate([order[:: max(1, n // 400)], order[-40:], rng.integers(0, n, 200)]))
    sb = b"".join(bytes(win.bases[off[3 * w]:off[3 * w + 3]]) for w in pick)
    soff = np.zeros(3 * len(pick) + 1, dtype=np.int64)
    np.cumsum(np.concatenate([lens[3 * w:3 * w + 3] for w in pick]), out=soff[1:])
    exp_rows, exp_ncol, _, _ = oracle_lib.batch(np.frombuffer(sb, dtype=np.uint8), soff)
    for k, w in enumerate(pick):
        nc = int(ncol[w])
        assert nc == exp_ncol[k]
        block = cols1[3 * off[3 * w]:3 * off[3 * w] + 3 * nc].reshape(nc, 3)
        assert tuple(bytes(block[:, r]) for r in range(3)) == exp_rows[k], int(w)
