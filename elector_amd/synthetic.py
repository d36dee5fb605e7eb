"""Seeded synthetic long-read triples (reference, corrected, uncorrected) for
bench.py and the large property tests.  Vectorised numpy; uniform ACGT genome.

Profiles restate BASELINE.json's configs: read-length
and error models only -- there is no network for real genomes or simulators.
"""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)

# name -> (mean read length, sd fraction, uncorrected error, (sub, ins, del) shares,
#          corrected error, (sub, ins, del) shares)
PROFILES = {
    # configs[1]: E. coli 30X SimLord PacBio 15 % error (-pi .22 -pd .08 -ps .01 shares), LoRDEC-like 1 %
    "ecoli30x_simlord_lordec": (8000, 0.20, 0.15, (0.01 / 0.31, 0.22 / 0.31, 0.08 / 0.31), 0.01, (0.3, 0.4, 0.3)),
    # configs[2]: yeast 50X NanoSim ONT 12 %, CONSENT-like 2 %
    "yeast50x_nanosim_consent": (8000, 0.35, 0.12, (0.3, 0.3, 0.4), 0.02, (0.3, 0.3, 0.4)),
    # configs[4]: human chr1 20X ONT, 50 kb mean
    "chr1_20x_ont_50kb": (50000, 0.40, 0.12, (0.3, 0.3, 0.4), 0.02, (0.3, 0.3, 0.4)),
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


def read_triples(profile, n_reads, seed):
    """-> list of (reference, corrected, uncorrected) ASCII bytes"""
    mean, sd, eu, su, ec, scor = PROFILES[profile]
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_reads):
        L = max(500, int(rng.normal(mean, sd * mean)))
        ref = rng.integers(0, 4, size=L).astype(np.uint8)
        unc = mutate_fast(rng, ref, eu, su)
        cor = mutate_fast(rng, ref, ec, scor)
        out.append((ACGT[ref].tobytes(), ACGT[cor].tobytes(), ACGT[unc].tobytes()))
    return out
