"""Seeded synthetic inputs shared by the tests, the golden-fixture generator and
bench.py.  Pure numpy; no reference code involved.

A "window" is one (reference, corrected, uncorrected) triple of short sequences
as ELECTOR's splitter hands them to the POA engine; a "read
triple" is the three full-length reads the splitter cuts into windows.
"""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_seq(rng, n, alphabet=ACGT):
    return alphabet[rng.integers(0, len(alphabet), size=n)].tobytes()


def mutate(rng, seq, err, ratios=(0.3, 0.4, 0.3), alphabet=ACGT):
    """Apply substitutions/insertions/deletions at total rate `err`;
    ratios = (sub, ins, del)."""
    if err <= 0 or len(seq) == 0:
        return bytes(seq)
    s = np.frombuffer(seq, dtype=np.uint8)
    r = rng.random(len(s))
    psub, pins, pdel = (err * x for x in ratios)
    out = bytearray()
    for i, c in enumerate(s):
        x = r[i]
        if x < psub:
            alt = alphabet[alphabet != c]
            out.append(int(alt[rng.integers(0, len(alt))]) if len(alt) else int(c))
        elif x < psub + pins:
            out.append(int(alphabet[rng.integers(0, len(alphabet))]))
            out.append(int(c))
        elif x < psub + pins + pdel:
            continue
        else:
            out.append(int(c))
    if not out:
        out.append(int(s[0]))
    return bytes(out)


def window_triples(seed, n, lo=7, hi=120, err_unc=0.15, err_cor=0.01):
    """n windows with reference length uniform in [lo, hi]."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        ref = random_seq(rng, L)
        unc = mutate(rng, ref, err_unc)
        cor = mutate(rng, ref, err_cor)
        out.append((ref, cor, unc))
    return out


def adversarial_triples(seed, n, hi=90):
    """Tie-heavy and degenerate windows: homopolymers, 2-letter alphabets,
    single-letter corrected windows ('N' filler), 'AAA' dummies, unknown
    symbols, 'N' runs, truncated and unrelated sequences, lower case."""
    rng = np.random.default_rng(seed)
    AC = np.frombuffer(b"AC", dtype=np.uint8)
    A = np.frombuffer(b"A", dtype=np.uint8)
    out = []
    for k in range(n):
        kind = k % 12
        L = int(rng.integers(1, hi + 1))
        if kind == 0:      # 2-letter alphabet, heavy errors
            ref = random_seq(rng, L, AC); cor = mutate(rng, ref, 0.1, alphabet=AC); unc = mutate(rng, ref, 0.3, alphabet=AC)
        elif kind == 1:    # homopolymers of different lengths
            ref = b"A" * L; cor = b"A" * max(1, L + int(rng.integers(-3, 4))); unc = b"A" * max(1, L + int(rng.integers(-5, 6)))
        elif kind == 2:    # corrected is the splitter's 'N' filler
            ref = random_seq(rng, L); cor = b"N"; unc = mutate(rng, ref, 0.15)
        elif kind == 3:    # dummy triple
            ref = cor = unc = b"AAA"
        elif kind == 4:    # unknown symbols and N runs
            ref = random_seq(rng, L); b = bytearray(mutate(rng, ref, 0.05))
            for _ in range(1 + L // 10):
                b[int(rng.integers(0, len(b)))] = int(rng.choice(np.frombuffer(b"NRYKMSWXn-?]", dtype=np.uint8)))
            cor = bytes(b); unc = mutate(rng, ref, 0.15)
        elif kind == 5:    # unrelated sequences
            ref = random_seq(rng, L); cor = random_seq(rng, max(1, L // 2)); unc = random_seq(rng, L + 3)
        elif kind == 6:    # truncated corrected / uncorrected
            ref = random_seq(rng, L + 5); cor = ref[: max(1, L // 3)]; unc = mutate(rng, ref[L // 2:], 0.15)
        elif kind == 7:    # long indel in the corrected read
            ref = random_seq(rng, L + 20); cut = int(rng.integers(0, L + 1)); cor = ref[:cut] + ref[cut + 15:]; unc = mutate(rng, ref, 0.2)
        elif kind == 8:    # lower case input, mixed
            ref = random_seq(rng, L).lower(); cor = mutate(rng, ref.upper(), 0.02); unc = mutate(rng, ref.upper(), 0.15).lower()
        elif kind == 9:    # length-1 everything
            ref = random_seq(rng, 1); cor = random_seq(rng, 1); unc = random_seq(rng, 1)
        elif kind == 10:   # long insertion in corrected + tandem repeat
            unit = random_seq(rng, 3); ref = unit * (1 + L // 3); cor = ref + unit * 2; unc = mutate(rng, ref, 0.2)
        else:              # all-mismatch
            ref = b"A" * L; cor = b"C" * L; unc = b"G" * max(1, L - 1)
        out.append((ref, cor, unc))
    return out


def read_triples(seed, n, mean_len=8000, sd_frac=0.2, err_unc=0.15, err_cor=0.01,
                 ratios_unc=(0.3, 0.4, 0.3), ratios_cor=(0.3, 0.4, 0.3), min_len=200):
    """n full-length (reference, corrected, uncorrected) read triples."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        L = max(min_len, int(rng.normal(mean_len, sd_frac * mean_len)))
        ref = random_seq(rng, L)
        unc = mutate(rng, ref, err_unc, ratios_unc)
        cor = mutate(rng, ref, err_cor, ratios_cor)
        out.append((ref, cor, unc))
    return out


def pack_windows(triples):
    """-> (bases: bytes, off: int64[3n+1]) in the C-ABI layout (ref, cor, unc per window)."""
    off = np.zeros(3 * len(triples) + 1, dtype=np.int64)
    parts = []
    k = 0
    pos = 0
    for t in triples:
        for s in t:
            parts.append(s)
            pos += len(s)
            k += 1
            off[k] = pos
    return b"".join(parts), off


def write_fasta_triples(triples, prefix, header_fmt=">w{}"):
    """Write three lock-step 2-line-per-record FASTA files (as masterSplitter
    emits them): <prefix>1 = reference, <prefix>2 = uncorrected, <prefix>3 = corrected."""
    names = [prefix + "1", prefix + "2", prefix + "3"]
    with open(names[0], "wb") as fr, open(names[1], "wb") as fu, open(names[2], "wb") as fc:
        for i, (ref, cor, unc) in enumerate(triples):
            h = header_fmt.format(i).encode()
            fr.write(h + b"\n" + ref + b"\n")
            fc.write(h + b"\n" + cor + b"\n")
            fu.write(h + b"\n" + unc + b"\n")
    return names
