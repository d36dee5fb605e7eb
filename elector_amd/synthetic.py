"""Seeded synthetic long-read triples (reference, corrected, uncorrected) for
bench.py and the large property tests.  Vectorised numpy; uniform ACGT genome.

Profiles restate BASELINE.json's configs: read-length
and error models plus what the corrector does to a read's extent -- there is no
network for real genomes, simulators or correctors.  A corrector that trims or
splits (`-split`) hands ELECTOR several corrected *pieces* per read; ELECTOR then
duplicates the reference and the uncorrected read once per piece under the
headers `<name>_0`, `<name>_1`, ... (elector/readAndSortFiles.py:150-191), so a
piece is a read of its own for the splitter and the POA engine, and pieces with
one header prefix are one read again for the statistics (computeStats.py:45-56).
The splitter pads what a piece does not cover with `N` filler windows
(Master_Splitter.cpp:139-154,268-277,295-301).
"""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)

# name -> dict(mean read length, length model, uncorrected error + (sub, ins, del) shares,
#              corrected error + shares, shape mix of the corrected reads)
# shapes: full = the corrector returns the whole read; trim = one or both ends missing; split = 2-3
# pieces with uncorrected stretches between them dropped; ext = extended beyond the reference's ends.
PROFILES = {
    # configs[0]: the bundled E. coli ~10X example restated (the example's own inputs are
    # absent from the checkout): 459 reads, mean 9.5 kb, uncorrected 10.3 % with insertions : deletions :
    # substitutions = 1:1:1, corrected 0.6 %, 8 % of the reads trimmed or split -- plus what the README's log of
    # that run shows beside them (README.md:136-160): a few reads extended by the corrector and a few corrected
    # reads shorter than a tenth of their reference (the splitter's "small reads")
    "ecoli10x_c1": dict(mean=9500, sd=0.20, length="normal", eu=0.103, su=(1 / 3, 1 / 3, 1 / 3),
                        ec=0.006, sc=(1 / 3, 1 / 3, 1 / 3), shapes=dict(full=0.87, trim=0.04, split=0.04, ext=0.01, small=0.04)),
    # configs[1]: E. coli 30X SimLord PacBio 15 % error (-pi .22 -pd .08 -ps .01 shares), LoRDEC-like 1 %
    "ecoli30x_simlord_lordec": dict(mean=8000, sd=0.20, length="normal", eu=0.15, su=(0.01 / 0.31, 0.22 / 0.31, 0.08 / 0.31),
                                    ec=0.01, sc=(0.3, 0.4, 0.3), shapes=dict(full=1.0)),
    # configs[2]: yeast 50X NanoSim ONT 12 %, CONSENT-like 2 %, every corrected read returned whole
    "yeast50x_nanosim_consent": dict(mean=8000, sd=0.35, length="normal", eu=0.12, su=(0.3, 0.3, 0.4),
                                     ec=0.02, sc=(0.3, 0.3, 0.4), shapes=dict(full=1.0)),
    # configs[2] as BASELINE.json words it: CONSENT run with -split -- a third of the reads come back as
    # 2-3 pieces, a tenth with an end missing
    "yeast50x_nanosim_consent_split": dict(mean=8000, sd=0.35, length="normal", eu=0.12, su=(0.3, 0.3, 0.4),
                                           ec=0.02, sc=(0.3, 0.3, 0.4), shapes=dict(full=0.57, split=0.33, trim=0.10)),
    # configs[3]: C. elegans 30X SimLord 15 %, mixed trimmed / split / extended corrected reads
    "celegans30x_simlord_mixed": dict(mean=8000, sd=0.25, length="normal", eu=0.15, su=(0.01 / 0.31, 0.22 / 0.31, 0.08 / 0.31),
                                      ec=0.015, sc=(0.3, 0.4, 0.3), shapes=dict(full=0.40, trim=0.30, split=0.25, ext=0.05)),
    # configs[4]: human chr1 20X ONT, 50 kb mean (log-normal lengths: a fifth of the reads exceed 65 kb)
    "chr1_20x_ont_50kb": dict(mean=50000, sd=0.50, length="lognormal", eu=0.12, su=(0.3, 0.3, 0.4),
                              ec=0.02, sc=(0.3, 0.3, 0.4), shapes=dict(full=0.9, trim=0.07, split=0.03)),
}


def mutate_fast(rng, codes, err, shares):
    """codes: uint8 array of 0..3.  Per base: substitute / insert-before / delete."""
    n = len(codes)
    if err <= 0 or n == 0:
        return codes.copy()
    r = rng.random(n)
    ps, pi, pd = (err * s for s in shares)
    is_sub = r < ps
    is_ins = (r >= ps) & (r < ps + pi)
    is_del = (r >= ps + pi) & (r < ps + pi + pd)
    base = np.where(is_sub, (codes + rng.integers(1, 4, size=n).astype(np.uint8)) & 3, codes).astype(np.uint8)
    keep = ~is_del
    cnt = keep.astype(np.int64) + is_ins.astype(np.int64)
    pos = np.cumsum(cnt) - cnt
    total = int(cnt.sum())
    if total == 0:
        return codes[:1].copy()
    out = np.empty(total, dtype=np.uint8)
    ins_idx = np.nonzero(is_ins)[0]
    out[pos[ins_idx]] = rng.integers(0, 4, size=len(ins_idx)).astype(np.uint8)
    keep_idx = np.nonzero(keep)[0]
    out[pos[keep_idx] + is_ins[keep_idx]] = base[keep_idx]
    # an inserted base in front of a deleted base keeps its slot
    return out


def _length(rng, prof):
    mean, sd = prof["mean"], prof["sd"]
    if prof["length"] == "lognormal":
        sigma = sd
        return max(500, int(rng.lognormal(np.log(mean) - 0.5 * sigma * sigma, sigma)))
    return max(500, int(rng.normal(mean, sd * mean)))


def _pieces(rng, cor, shape):
    """The corrected read as the corrector would hand it out: list of code arrays."""
    n = len(cor)
    if shape == "trim":
        side = int(rng.integers(0, 3))                       # left, right, both
        a = int(n * rng.uniform(0.05, 0.45)) if side in (0, 2) else 0
        b = n - int(n * rng.uniform(0.05, 0.45)) if side in (1, 2) else n
        return [cor[a:max(b, a + 50)]]
    if shape == "split":
        k = int(rng.integers(2, 4))
        cuts = np.sort(rng.integers(n // 10, n - n // 10, size=k - 1))
        out, start = [], 0
        for c in list(cuts) + [n]:
            gap = int(rng.integers(40, 300))
            end = int(c) - (gap if c != n else 0)
            if end - start >= 60:
                out.append(cor[start:end])
            start = int(c)
        return out or [cor]
    if shape == "small":                                     # a stub of 2-8 % of the read: below SIZE_CORRECTED_READ_THRESHOLD
        m = max(12, int(n * rng.uniform(0.02, 0.08)))
        a = int(rng.integers(0, max(1, n - m)))
        return [cor[a:a + m]]
    if shape == "ext":
        left = rng.integers(0, 4, size=int(rng.integers(25, 120))).astype(np.uint8)
        right = rng.integers(0, 4, size=int(rng.integers(0, 120))).astype(np.uint8)
        return [np.concatenate([left, cor, right])]
    return [cor]


def read_pieces(profile, n_reads, seed):
    """-> (triples, headers, read_of): one (reference, corrected piece, uncorrected) ASCII triple per
    corrected piece, its header line `>read<i>_<k>` as ELECTOR's duplicateRefReads names it, and the
    index of the read each piece belongs to (non-decreasing)."""
    prof = PROFILES[profile]
    rng = np.random.default_rng(seed)
    names = list(prof["shapes"])
    probs = np.asarray([prof["shapes"][k] for k in names], dtype=np.float64)
    probs /= probs.sum()
    triples, headers, read_of = [], [], []
    for i in range(n_reads):
        L = _length(rng, prof)
        ref = rng.integers(0, 4, size=L).astype(np.uint8)
        unc = mutate_fast(rng, ref, prof["eu"], prof["su"])
        cor = mutate_fast(rng, ref, prof["ec"], prof["sc"])
        shape = names[int(rng.choice(len(names), p=probs))] if len(names) > 1 else names[0]
        r_txt, u_txt = ACGT[ref].tobytes(), ACGT[unc].tobytes()
        for k, piece in enumerate(_pieces(rng, cor, shape)):
            triples.append((r_txt, ACGT[piece].tobytes(), u_txt))
            headers.append(b">read%d_%d" % (i, k))
            read_of.append(i)
    return triples, headers, np.asarray(read_of, dtype=np.int64)


def read_triples(profile, n_reads, seed):
    """-> list of (reference, corrected, uncorrected) ASCII bytes, one per corrected piece."""
    return read_pieces(profile, n_reads, seed)[0]


def piece_groups(read_of, read_index):
    """Statistics grouping of the splitter's emitted reads (= pieces): read_of as read_pieces returns
    it, read_index = Windows.read_index (which input pieces produced output).  -> int64 boundaries
    `read_first` such that the pieces [read_first[r], read_first[r+1]) are one read for
    elector_msa_stats_* (computeStats.py:45-56 groups pieces by header)."""
    ids = np.asarray(read_of, dtype=np.int64)[np.asarray(read_index, dtype=np.int64)]
    if len(ids) == 0:
        return np.zeros(1, dtype=np.int64)
    starts = np.concatenate([[0], np.nonzero(np.diff(ids))[0] + 1, [len(ids)]])
    return starts.astype(np.int64)
