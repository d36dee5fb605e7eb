"""Every BASELINE.json workload that can be built here, on the GPU at the size bench.py runs it
(config 1's inputs, example/*_elector.fa, are not in the reference checkout):

  config 2  E. coli 30X SimLord / LoRDEC            10,001 reads = one batch of ELECTOR's protocol
  config 3  yeast 50X NanoSim / CONSENT with -split  10,001 reads, a third of them in 2-3 pieces
  config 4  C. elegans 30X SimLord, mixed trimmed / split / extended corrected reads, 10,001 reads
  config 5  human chr1 20X ONT, 50 kb mean reads    2,000 reads (the bases of a config-2 batch), plus
            crafted reads that produce the reference's un-anchored whole-read windows
            (Master_Splitter.cpp:256-261) and windows of 10-20 kb

What is checked is in tests/config_check.py."""
import numpy as np
import pytest

import config_check
from elector_amd import synthetic

pytestmark = pytest.mark.gpu


def _run(engine, tmp_path, profile, n_reads, seed, **kw):
    triples, headers, read_of = synthetic.read_pieces(profile, n_reads, seed)
    return config_check.run_workload(engine, triples, headers, read_of, tmp_path, **kw)


def test_config2_ecoli_full_batch(engine, tmp_path):
    r = _run(engine, tmp_path, "ecoli30x_simlord_lordec", 10001, 4242)
    assert r["reads"] == 10001 and r["windows"] > 1_300_000 and r["filler"] == 0


def test_config3_yeast_consent_split(engine, tmp_path):
    r = _run(engine, tmp_path, "yeast50x_nanosim_consent_split", 10001, 4343)
    assert r["reads"] == 10001 and r["pieces"] > 13000 and r["split_reads"] > 2500
    assert r["filler"] > 0.2 * r["windows"]                 # the `N` padding of what a piece does not cover


def test_config4_celegans_mixed(engine, tmp_path):
    r = _run(engine, tmp_path, "celegans30x_simlord_mixed", 10001, 4444)
    assert r["reads"] == 10001 and r["split_reads"] > 1800 and r["filler"] > 0.15 * r["windows"] and r["small"] > 0


def _tandem(rng, period, length):
    unit = rng.integers(0, 4, size=period).astype(np.uint8)
    return np.tile(unit, length // period + 1)[:length]


def crafted_long_window_reads(seed):
    """Reads whose k-mers repeat over a long stretch, as human repeats do: no k-mer there is unique in
    the reference read, so no anchor falls inside it (Master_Splitter.cpp:200-251).
      A: repeat in the middle of an otherwise ordinary read -> one window spanning the whole repeat;
      B: the corrected read lacks the (repetitive) first half -> the splitter re-splits reference vs
         uncorrected over the missing part, finds no chain there and emits it as ONE un-anchored
         window (Master_Splitter.cpp:256-261,268-277)."""
    rng = np.random.default_rng(seed)
    A = synthetic.ACGT
    mut = synthetic.mutate_fast
    su, sc = (0.3, 0.3, 0.4), (0.3, 0.3, 0.4)
    out = []
    flank = lambda n: rng.integers(0, 4, size=n).astype(np.uint8)          # noqa: E731
    ref = np.concatenate([flank(3000), _tandem(rng, 37, 11000), flank(3000)])
    out.append((A[ref].tobytes(), A[mut(rng, ref, 0.02, sc)].tobytes(), A[mut(rng, ref, 0.12, su)].tobytes()))
    ref = np.concatenate([_tandem(rng, 53, 9000), flank(7000)])
    cor = mut(rng, ref, 0.02, sc)
    out.append((A[ref].tobytes(), A[cor[len(cor) - 6800:]].tobytes(), A[mut(rng, ref, 0.12, su)].tobytes()))
    return out


def test_config5_chr1_long_reads(engine, tmp_path):
    triples, headers, read_of = synthetic.read_pieces("chr1_20x_ont_50kb", 2000, 4545)
    assert max(len(t[0]) for t in triples) > 65520                 # reads longer than the 16-bit window limit
    first_crafted = len(triples)
    for k, t in enumerate(crafted_long_window_reads(99)):
        triples.append(t)
        headers.append(b">crafted%d_0" % k)
        read_of = np.append(read_of, read_of[-1] + 1 + k)
    # which windows the crafted reads become: probe the splitter on them alone (host only)
    from elector_amd import split
    probe = split.split_reads(triples[first_crafted:], 0.1, headers[first_crafted:], nthreads=2)
    plen = np.diff(probe.off)
    long_local = np.nonzero(plen[0::3] > 8000)[0]
    assert len(long_local) >= 2, "crafted reads did not produce long windows: %s" % plen[0::3].max()
    r = config_check.run_workload(engine, triples, headers, read_of, tmp_path, oracle_reads=3, oracle_windows=400,
                                  must_check=(), max_oracle_cells=2.0e9)
    win = r["win"]
    lens = np.diff(win.off)
    long_w = np.nonzero(lens[0::3] > 8000)[0]
    assert len(long_w) == len(long_local) and r["max_window"] > 8000
    # the long windows against the oracle (it needs ~10 s for each)
    config_check.oracle_window_sample(win, r["cols"], r["ncol"], long_w)
