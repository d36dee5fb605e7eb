"""The device splitter (elector_split_reads_device, split_dev.hip) against the host splitter, which the CPU
tests pin to the reference's masterSplitter (golden vectors, the real binary): identical windows, identical
read bookkeeping, identical small / wrong counts -- on the golden reads, on every synthetic profile (whole,
trimmed, split, extended corrected reads), and on the degenerate inputs the reference's integer quirks matter for."""
import numpy as np
import pytest

import golden_io
import synth
from elector_amd import split, synthetic

pytestmark = pytest.mark.gpu


def same(engine, reads, headers=None, thr=0.1):
    host = split.split_reads(reads, thr, headers, nthreads=4)
    dev = split.split_reads_device(engine, reads, thr, headers)
    assert (dev.n_reads, dev.n_windows) == (host.n_reads, host.n_windows)
    assert (dev.small_reads, dev.wrong_reads) == (host.small_reads, host.wrong_reads)
    assert np.array_equal(dev.read_first, host.read_first) and np.array_equal(dev.read_index, host.read_index)
    d = np.nonzero(dev.off != host.off)[0]
    assert len(d) == 0, "window offsets differ first at window %d (read %d)" % (
        d[0] // 3, int(np.searchsorted(host.read_first, d[0] // 3, side="right") - 1))
    got = dev.d_bases.numpy() if hasattr(dev.d_bases, "numpy") and not hasattr(dev.d_bases, "cpu") else dev.d_bases.cpu().numpy()
    assert np.array_equal(got, host.bases)
    return host


def test_golden_reads(engine):
    reads, wins, small, wrong = golden_io.splitter()
    h = same(engine, [r[1] for r in reads], [r[0] for r in reads])
    assert h.triples() == [w[1] for w in wins]


@pytest.mark.parametrize("profile,n", [("ecoli30x_simlord_lordec", 400), ("yeast50x_nanosim_consent_split", 400),
                                       ("celegans30x_simlord_mixed", 400), ("chr1_20x_ont_50kb", 60)])
def test_profiles(engine, profile, n):
    triples, headers, _ = synthetic.read_pieces(profile, n, 77)
    same(engine, triples, headers)
    # the device path itself took the batch (reads beyond ~60 kb included: their anchor arrays live in HBM)
    assert isinstance(split.split_reads_device(engine, triples, 0.1, headers).d_bases, split.DevBases)


def test_degenerate_reads(engine):
    rng = np.random.default_rng(5)
    r = synth.random_seq(rng, 900)
    reads = [
        (b"AC", b"AC", b"AC"),                                       # skipped (:414)
        (b"ACG", b"ACG", b"ACG"),                                    # no anchor: dummy
        (r, r[:50], synth.mutate(rng, r, 0.1)),                      # corrected too short: small read
        (r, r.lower(), synth.mutate(rng, r, 0.1)),                   # lower case: the two letter maps disagree
        (r, synth.mutate(rng, r, 0.02).replace(b"A", b"N", 7), synth.mutate(rng, r, 0.12)),
        (b"A" * 700, b"A" * 690, b"A" * 710),                        # every k-mer repeated
        (b"ACGT" * 200, b"ACGT" * 199, b"ACGT" * 201),
        (r[:14], r[:14], r[:14]), (r[:15], r[:15], r[:15]), (r[:16], r[:16], r[:16]),   # around k
        (r, b"", synth.mutate(rng, r, 0.1)), (r, r, b""),
        (r, r[300:], synth.mutate(rng, r, 0.12)), (r, r[:500], synth.mutate(rng, r, 0.12)),   # trimmed
        (r + r, r + r, r + r),                                       # every k-mer twice
    ]
    same(engine, reads)
    same(engine, reads, thr=0.6)
    same(engine, [], None)


def test_on_chip_table_limits(engine):
    """The k-mer tables in LDS take reads of up to 12,500 bases and uncorrected reads that share at most 3,200
    unique k-mers with the reference; everything else in the same batch goes through the HBM tables."""
    rng = np.random.default_rng(11)
    reads = []
    for n in (5000, 9000, 12499, 12500, 12501, 13000, 16000):
        r = synth.random_seq(rng, n)
        reads.append((r, synth.mutate(rng, r, 0.01), synth.mutate(rng, r, 0.13)))
    r = synth.random_seq(rng, 8000)
    reads.append((r, r, r))                                          # error-free: the second table overflows
    reads.append((r, synth.mutate(rng, r, 0.01), synth.mutate(rng, r, 0.02)))   # nearly error-free uncorrected read
    r = synth.random_seq(rng, 3300)
    reads.append((r, r, r))                                          # just around the fill limit
    for _ in range(30):                                              # the batch is mostly short reads: tables on chip
        r = synth.random_seq(rng, int(rng.integers(3000, 11000)))
        reads.append((r, synth.mutate(rng, r, 0.01), synth.mutate(rng, r, 0.15)))
    same(engine, reads)


@pytest.mark.parametrize("seed", [101, 202])
def test_soak(engine, seed):
    """More reads per profile than test_profiles, other seeds: the serial stretches of k_split (anchor walk by
    fixed-point rounds, chain lengths by offers, window cuts by links) see re-splits of many start / end lengths
    (minSize below and above the limits of their fast forms) and chains of every length."""
    for profile, n in (("yeast50x_nanosim_consent_split", 1200), ("celegans30x_simlord_mixed", 1200), ("ecoli10x_c1", 459)):
        triples, headers, _ = synthetic.read_pieces(profile, n, seed)
        same(engine, triples, headers)
    # corrected reads that cover only a stretch of the reference, at both ends and in the middle: re-splits with
    # minSize = 1.2 x (what is left of the corrected read), from 0 upwards
    rng = np.random.default_rng(seed)
    reads = []
    for _ in range(60):
        r = synth.random_seq(rng, int(rng.integers(1500, 9000)))
        u = synth.mutate(rng, r, 0.12)
        c = synth.mutate(rng, r, 0.01)
        a, b = sorted(int(x) for x in rng.integers(0, len(c), 2))
        cut = int(rng.integers(0, 40))
        reads.append((r, c[a:b] if b - a > 50 else c[cut:], u))
        reads.append((r, c[:max(60, len(c) - a)], u))
    same(engine, reads)


def test_long_read_tables(engine):
    """A batch of mostly long reads takes the partitioned on-chip tables (reads of up to 65,535 and of up to 122,879
    bases: two table sizes), longer reads and partitions whose second table fills up (a nearly error-free
    uncorrected read) the HBM tables -- in one launch, the reads drawn longest first."""
    rng = np.random.default_rng(23)
    reads = []
    for n in (20000, 40000, 65535, 65536, 70000, 122879, 122880, 131000):
        r = synth.random_seq(rng, n)
        reads.append((r, synth.mutate(rng, r, 0.01), synth.mutate(rng, r, 0.13)))
    r = synth.random_seq(rng, 30000)
    reads.append((r, r, r))                                          # error-free: a partition's second table overflows
    reads.append((r, synth.mutate(rng, r, 0.02)[5000:21000], synth.mutate(rng, r, 0.12)))   # re-splits at both ends
    r = synth.random_seq(rng, 90000)
    reads.append((r, synth.mutate(rng, r, 0.02)[:50000], synth.mutate(rng, r, 0.12)))
    for n in (800, 9000):                                            # a few short ones in the same batch
        r = synth.random_seq(rng, n)
        reads.append((r, synth.mutate(rng, r, 0.01), synth.mutate(rng, r, 0.15)))
    same(engine, reads)


def test_dense_resplit_gets_a_second_try(engine):
    """A corrected read that only covers the END of a 12 kb reference, an accurate uncorrected read: the re-split
    of the missing start runs with a minSize of about a dozen bases and takes an anchor every dozen bases -- more
    than the anchor arrays sized for the launch (one per 21 bases) hold.  The read is split again with room for an
    anchor per base; the batch still comes back from the device, identical to the host splitter's."""
    rng = np.random.default_rng(31)
    reads = []
    for n, a, b in ((12000, 11000, 11900), (12400, 200, 1100), (9000, 8000, 8990)):
        r = synth.random_seq(rng, n)
        reads.append((r, synth.mutate(rng, r, 0.004)[a:b], synth.mutate(rng, r, 0.01)))
    for _ in range(20):
        r = synth.random_seq(rng, int(rng.integers(3000, 9000)))
        reads.append((r, synth.mutate(rng, r, 0.01), synth.mutate(rng, r, 0.12)))
    same(engine, reads)
    assert isinstance(split.split_reads_device(engine, reads, 0.1, None).d_bases, split.DevBases)
