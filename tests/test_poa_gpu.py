"""Parity tests proper: the HIP path, called through the C ABI, against the CPU
oracle on the same seeded inputs.  Bit-exact: MSA rows (bytes), column counts
and the best scores of both alignments (integer DP)."""
import numpy as np
import pytest

import oracle_lib
import synth

pytestmark = pytest.mark.gpu


def check(engine, triples):
    bases, off = synth.pack_windows(triples)
    exp_rows, exp_ncol, exp_scores, _ = oracle_lib.batch(np.frombuffer(bases, dtype=np.uint8), off)
    got, scores = engine.align(triples, want_scores=True)
    bad = [w for w in range(len(triples)) if got[w] != exp_rows[w]]
    assert not bad, "first differing window %d of %d: %r\n got %r\n exp %r" % (
        bad[0], len(bad), triples[bad[0]], got[bad[0]], exp_rows[bad[0]])
    d = np.argwhere(scores != exp_scores)
    assert len(d) == 0, "scores differ at %s: got %s exp %s, window %r" % (
        d[:5].tolist(), scores[d[0][0]].tolist(), exp_scores[d[0][0]].tolist(), triples[d[0][0]])


def test_small_windows(engine):
    check(engine, synth.window_triples(11, 2000, 1, 62))


def test_typical_windows(engine):
    check(engine, synth.window_triples(12, 3000, 7, 120))


def test_multi_strip_windows(engine):
    check(engine, synth.window_triples(13, 600, 100, 420))


def test_adversarial_windows(engine):
    check(engine, synth.adversarial_triples(14, 3600))


def test_noisy_corrected(engine):
    # heavy corrected-read errors -> big bubbles, long predecessor distances
    check(engine, synth.window_triples(15, 800, 20, 200, err_unc=0.3, err_cor=0.25))


def test_long_windows(engine):
    check(engine, synth.window_triples(16, 12, 1500, 3000))


def deep_graph_triples(seed):
    """Windows whose graph has predecessors hundreds or thousands of nodes back: what the splitter
    emits for a trimmed / split corrected piece -- the piece's whole tail beyond the last anchor lands
    in ONE window next to a reference window of ordinary size (Master_Splitter.cpp:295-301), and the
    tail's letters, unaligned, sit between two consecutive reference letters in the graph."""
    rng = np.random.default_rng(seed)
    out = []
    for lr, tail in ((60, 300), (80, 900), (57, 2000), (120, 5000), (300, 700), (45, 33), (64, 511), (64, 512)):
        ref = synth.random_seq(rng, lr)
        unc = synth.mutate(rng, ref, 0.15)
        cor = synth.mutate(rng, ref, 0.02)
        junk = synth.random_seq(rng, tail)
        k = lr // 2
        out.append((ref, cor[:k] + junk + cor[k:], unc))      # a long insertion inside the window
        out.append((ref, cor + junk, unc))                    # the piece runs on beyond the window
        out.append((ref, junk + cor, unc))
        out.append((ref + junk, cor, unc))                    # and the mirror image: reference-only stretch
    return out


def test_deep_graphs(engine):
    """predecessor distances beyond every on-chip ring: the HBM shadow ring of k_dp2"""
    check(engine, deep_graph_triples(61))


@pytest.mark.parametrize("name", ["windows_example.tsv", "windows_synth.tsv", "windows_adversarial.tsv"])
def test_golden_vectors(engine, name):
    """HIP path against rows printed by the real reference binary (tests/golden)."""
    import golden_io
    gold = golden_io.windows(name)
    got = engine.align([g[0] for g in gold])
    for w, (t, exp) in enumerate(gold):
        assert got[w] == exp, (w, t)


GENERAL_MATRIX = ("GAP-TRUNCATION-LENGTH=3\nGAP-DECAY-LENGTH=4\nGAP-PENALTIES=9 4 1\nGAP-PENALTIES-X=7 3 2\n"
                  "  A a c g t n\nA 1 -3 -3 -3 -3 -1\na -3 4 -5 -2 -5 -1\nc -3 -5 4 -5 -2 -1\n"
                  "g -3 -2 -5 4 -5 -1\nt -3 -5 -2 -5 4 -1\nn -1 -1 -1 -1 -1 0\n")


def test_general_scoring_matrix(tmp_path):
    """Non-uniform substitution scores, decaying gap penalties, different x/y
    penalties: the table-driven kernel variant against the oracle."""
    from elector_amd import poa
    path = tmp_path / "g.mat"
    path.write_text(GENERAL_MATRIX)
    eng = poa.PoaEngine(0, poa.read_params(path))
    par = oracle_lib.read_params(path)
    try:
        for triples in (synth.window_triples(17, 1200, 1, 140), synth.adversarial_triples(18, 600),
                        synth.window_triples(19, 60, 150, 400)):
            bases, off = synth.pack_windows(triples)
            exp_rows, _, exp_scores, _ = oracle_lib.batch(np.frombuffer(bases, dtype=np.uint8), off, par)
            got, scores = eng.align(triples, want_scores=True)
            assert got == exp_rows
            assert np.array_equal(scores, exp_scores)
    finally:
        eng.close()


def test_window_status_codes(engine):
    from elector_amd import _capi
    triples = [(b"ACGT", b"ACGT", b"ACGT"), (b"ACGT", b"", b"ACGT"), (b"A" * (_capi.ELECTOR_MAX_SEQ + 1), b"ACGT", b"ACGT"),
               (b"GATTACA", b"GATACA", b"GATTTACA")]
    with pytest.raises(_capi.ElectorError):
        engine.align(triples)
    got = engine.align(triples, strict=False)
    assert got[1] is None and got[2] is None
    assert got[0] == (b"acgt", b"acgt", b"acgt")
    bases, off = synth.pack_windows([triples[3]])
    assert got[3] == oracle_lib.batch(np.frombuffer(bases, dtype=np.uint8), off)[0][0]


def test_device_offsets_entry(engine):
    """elector_poa_batch_device_offsets: windows AND offsets in HBM (status, launch class and class lists computed by
    kernels).  Same rows as the host-offsets entry on a batch that spans every path -- k_poa, its far instance, the
    two-kernel path, the generic kernels, empty and over-long windows -- and malformed offsets are refused, not aligned."""
    import torch
    from elector_amd import _capi
    triples = (synth.window_triples(31, 3000, 5, 300) + synth.adversarial_triples(32, 300) + far_edge_triples(33, 300) +
               [(b"ACGT", b"", b"ACGT"), (b"A" * (_capi.ELECTOR_MAX_SEQ + 1), b"ACGT", b"ACGT")] + synth.window_triples(34, 4, 1500, 2500))
    bases, off = synth.pack_windows(triples)
    off = np.ascontiguousarray(off, dtype=np.int64)
    n, total = len(triples), int(off[-1])
    dev = torch.device("cuda", 0)
    d_bases = torch.from_numpy(np.frombuffer(bases, dtype=np.uint8).copy()).to(dev)
    outs = []
    for entry in ("host", "device"):
        d_cols = torch.zeros(3 * total + 64, dtype=torch.uint8, device=dev)
        d_ncol = torch.zeros(n, dtype=torch.int32, device=dev)
        d_status = torch.zeros(n, dtype=torch.int32, device=dev)
        if entry == "host":
            engine.align_device(d_bases, off, d_cols, d_ncol, d_status)
        else:
            engine.align_device_offsets(d_bases, torch.from_numpy(off).to(dev), n, total, d_cols, d_ncol, d_status)
        engine.sync()
        outs.append((d_cols.cpu().numpy(), d_ncol.cpu().numpy(), d_status.cpu().numpy()))
    (c0, n0, s0), (c1, n1, s1) = outs
    assert np.array_equal(s0, s1) and s0[-6] == _capi.W_EMPTY and s0[-5] == _capi.W_TOOLONG and not s0[:-6].any() and not s0[-4:].any()
    ok = s0 == 0
    assert np.array_equal(n0[ok], n1[ok])
    for w in np.nonzero(ok)[0]:
        a = 3 * int(off[3 * w])
        assert np.array_equal(c0[a:a + 3 * n0[w]], c1[a:a + 3 * n0[w]]), int(w)
    picks = [0, 17, 3100, 3400, n - 1]
    pb, po = synth.pack_windows([triples[w] for w in picks])
    exp_rows = oracle_lib.batch(np.frombuffer(pb, dtype=np.uint8), po)[0]
    for k, w in enumerate(picks):
        a, nc = 3 * int(off[3 * w]), int(n0[w])
        blk = c1[a:a + 3 * nc].reshape(nc, 3)
        assert tuple(bytes(blk[:, r]) for r in range(3)) == exp_rows[k], w
    # malformed offsets: not starting at 0, decreasing, a total that is not the last offset
    d_cols = torch.zeros(3 * total + 64, dtype=torch.uint8, device=dev)
    d_ncol = torch.zeros(n, dtype=torch.int32, device=dev)
    d_status = torch.zeros(n, dtype=torch.int32, device=dev)
    for bad, tot in ((off + 1, total + 1), (np.concatenate([off[:10], off[9:10] - 3, off[11:]]), total), (off, total + 5)):
        with pytest.raises(_capi.ElectorError):
            engine.align_device_offsets(d_bases, torch.from_numpy(np.ascontiguousarray(bad)).to(dev), n, int(tot), d_cols, d_ncol, d_status)
    # ... and the engine is usable afterwards
    engine.align_device_offsets(d_bases, torch.from_numpy(off).to(dev), n, total, d_cols, d_ncol, d_status)
    engine.sync()
    assert np.array_equal(d_ncol.cpu().numpy()[ok], n0[ok])


def test_empty_batch(engine):
    assert engine.align([]) == []


def test_round_trip_property_large(engine):
    """Full-size property check (no oracle): removing the gaps from each MSA row
    gives back the lower-cased input sequence, rows have equal length, no column
    is all gaps."""
    from elector_amd import split, synthetic
    reads = synthetic.read_triples("ecoli30x_simlord_lordec", 300, seed=5)
    win = split.split_reads(reads, 0.1, None, nthreads=8)
    rows, row_off, ncol, status, _ = engine.align_packed(win.bases, win.off)
    assert not status.any()
    b = win.bases.tobytes().lower()
    r = rows.tobytes()
    off = win.off
    for w in range(0, win.n_windows, 7):
        a, nc = int(row_off[w]), int(ncol[w])
        r0, r1, r2 = r[a:a + nc], r[a + nc:a + 2 * nc], r[a + 2 * nc:a + 3 * nc]
        assert r0.replace(b".", b"") == b[off[3 * w]:off[3 * w + 1]]
        assert r1.replace(b".", b"") == b[off[3 * w + 1]:off[3 * w + 2]]
        assert r2.replace(b".", b"") == b[off[3 * w + 2]:off[3 * w + 3]]
        assert all(not (x == 46 and y == 46 and z == 46) for x, y, z in zip(r0, r1, r2))


@pytest.mark.parametrize("cls", list(range(17)))
def test_every_geometry_class(engine, cls, monkeypatch):
    """Each (lanes per window, rows per lane) instantiation of the fused kernels on windows of every
    size (ELECTOR_FORCE_CLASS routes all windows that fit its LDS slots to one class): the small
    classes then run many strips per window, the big ones mostly idle lanes."""
    monkeypatch.setenv("ELECTOR_FORCE_CLASS", str(cls))
    check(engine, synth.window_triples(100 + cls, 500, 5, 420) + synth.adversarial_triples(200 + cls, 300))


def test_windows_beyond_16k(engine):
    """The reference's un-anchored whole-read fallback hands poa very long triples: windows longer than
    16,384 bases take the generic kernels (moves in HBM, node distances instead of 16-bit node ids)."""
    check(engine, synth.window_triples(31, 2, 16500, 17500, err_unc=0.12, err_cor=0.01))


def test_windows_beyond_65k(engine):
    """Nothing the reference aligns is refused for its length alone (align_lpo_po2.c:254-257 takes any): a
    window with a sequence of more than 65,535 bases in each of the three places -- columns of alignment #1,
    rows of alignment #1, rows of alignment #2 (32-bit maps on the generic / tiled path).  The other two
    sequences are short so that the oracle finishes in seconds."""
    rng = np.random.default_rng(71)
    long_ref = synth.random_seq(rng, 66100)
    short = long_ref[30000:31500]
    trip = [(long_ref, synth.mutate(rng, short, 0.02), synth.mutate(rng, short, 0.12)),
            (short, synth.mutate(rng, long_ref, 0.02), synth.mutate(rng, short, 0.12)),
            (short, synth.mutate(rng, short, 0.02), synth.mutate(rng, long_ref, 0.12))]
    check(engine, trip)


def test_too_long_is_reported(engine):
    from elector_amd._capi import ELECTOR_MAX_SEQ
    big = b"A" * (ELECTOR_MAX_SEQ + 1)
    got = engine.align([(big, b"ACGT", b"ACGT"), (b"ACGT", b"ACGT", b"ACGT")], strict=False)
    assert got[0] is None and got[1] == (b"acgt", b"acgt", b"acgt")
    with pytest.raises(Exception):
        engine.align([(big, b"ACGT", b"ACGT")])


def test_tiled_long_windows(engine, monkeypatch):
    """The tiled kernels of the long-window path (one tile = 63 rows x 2048 columns, one launch per tile
    anti-diagonal) on windows of 2,100-6,500 bases -- several column blocks and up to a hundred strips --
    with the size threshold lowered so that they take it; and mixed with ordinary windows."""
    monkeypatch.setenv("ELECTOR_TILE_CELLS", "1000000")
    long_ones = synth.window_triples(41, 5, 2100, 6500, err_unc=0.12, err_cor=0.02)
    check(engine, long_ones)
    check(engine, synth.window_triples(42, 200, 20, 120) + long_ones[:2] + synth.window_triples(43, 3, 1500, 2500))


def test_tiled_general_matrix(monkeypatch):
    from elector_amd.poa import PoaEngine
    import tempfile, os
    monkeypatch.setenv("ELECTOR_TILE_CELLS", "1000000")
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.mat")
        open(path, "w").write(GENERAL_MATRIX)
        from elector_amd.poa import read_params
        eng = PoaEngine(0, read_params(path))
        try:
            triples = synth.window_triples(44, 3, 2100, 3000, err_unc=0.12, err_cor=0.02)
            bases, off = synth.pack_windows(triples)
            exp_rows, _, exp_scores, _ = oracle_lib.batch(np.frombuffer(bases, dtype=np.uint8), off, oracle_lib.read_params(path))
            got, scores = eng.align(triples, want_scores=True)
            assert got == exp_rows and np.array_equal(scores, exp_scores)
        finally:
            eng.close()


def one_indel_triples(seed):
    """Windows whose corrected sequence is the reference with ONE letter inserted or deleted -- settled
    without a dynamic program (k_trivial / trivial_graph): every position including both ends, inside runs
    of equal letters (where the gap's place is the traceback's tie rule), inserted letters equal to the left
    or the right neighbour, short windows, and the same next to windows that do need alignment #1."""
    rng = np.random.default_rng(seed)
    acgt = b"ACGT"
    out = []

    def unc_of(ref):
        return synth.mutate(rng, ref, 0.15) or ref

    for L in (2, 3, 5, 9, 31, 62, 64, 100, 130, 250):
        ref = synth.random_seq(rng, L)
        # a run of equal letters in the middle and at both ends of some windows
        if L >= 9:
            ref = ref[:L // 2] + b"AAAA" + ref[L // 2 + 4:]
        if L >= 31:
            ref = b"CCC" + ref[3:-3] + b"GGG"
        pos = range(L + 1) if L <= 64 else sorted(set(int(x) for x in rng.integers(0, L + 1, 24)) | {0, 1, L - 1, L})
        for p in pos:
            if p < L:
                out.append((ref, ref[:p] + ref[p + 1:], unc_of(ref)))                   # deletion of letter p
            for b in acgt:
                out.append((ref, ref[:p] + bytes([b]) + ref[p:], unc_of(ref)))          # insertion in front of letter p
    # neighbours that need the dynamic program, so that wavefronts mix both kinds
    mixed = []
    for k, t in enumerate(out):
        mixed.append(t)
        if k % 7 == 0:
            ref = t[0]
            mixed.append((ref, synth.mutate(rng, ref, 0.2) or ref, unc_of(ref)))
    return [t for t in mixed if len(t[1]) >= 1]


def test_one_indel_windows(engine):
    triples = one_indel_triples(77)
    assert len(triples) > 1200
    check(engine, triples)


def test_one_indel_shortcut_equals_dynamic_program(engine, monkeypatch):
    """the same windows with the shortcut switched off (ELECTOR_NO_ONEINDEL): identical rows and scores"""
    triples = one_indel_triples(78)[:1500]
    got, scores = engine.align(triples, want_scores=True)
    monkeypatch.setenv("ELECTOR_NO_ONEINDEL", "1")
    got2, scores2 = engine.align(triples, want_scores=True)
    monkeypatch.delenv("ELECTOR_NO_ONEINDEL")
    assert got == got2
    assert np.array_equal(scores, scores2)


def score_range_triples(seed):
    """Windows between the old 16-bit admission rule (10 x the three lengths < 16000) and the proven score span
    (poa_device.h: score_span): the long un-anchored windows of trimmed / split reads, with the corrected side a
    lone `N` filler (Master_Splitter.cpp:139-154,268-277) or a real sequence -- and the worst cases for the span:
    unrelated sequences (every cell near the all-gap bound), all-mismatch pairs, one side much longer."""
    rng = np.random.default_rng(seed)
    out = []
    for lr in (530, 640, 746, 900, 1100, 1400, 1560):
        ref = synth.random_seq(rng, lr)
        unc = synth.mutate(rng, ref, 0.12)
        cor = synth.mutate(rng, ref, 0.02)
        out.append((ref, b"N", unc))                                    # filler window
        out.append((ref, cor, unc))
        out.append((ref, cor, synth.random_seq(rng, lr)))               # unrelated uncorrected read
        out.append((ref, synth.random_seq(rng, lr // 2), unc))          # unrelated, shorter corrected read
        out.append((b"A" * lr, b"C" * (lr // 3), b"G" * lr))            # nothing matches anywhere
        out.append((ref, cor, unc[: lr // 5]))
        out.append((ref[: lr // 4], cor, unc))
    return out


def test_windows_up_to_the_score_span(engine):
    check(engine, score_range_triples(71))


def far_edge_triples(seed, n):
    """Graphs with an edge from more than two nodes back after fusion #1: the corrected sequence lacks two or more
    letters of the reference in one place (up to most of the window: a corrected piece that aligns at both ends of its
    window, Master_Splitter.cpp:268-301), or has two or more extra letters in one place -- k_poa<G, 8, true> keeps ONE
    such edge per window; two of them, or one beside other differences, exercise the paths around it."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        L = int(rng.choice([12, 30, 57, 64, 65, 90, 128, 200, 400]))
        L = max(8, L + int(rng.integers(-3, 4)))
        ref = synth.random_seq(rng, L)
        kind = i % 6
        cor = bytearray(ref)
        def cut(c, lo_frac=0.0):
            k = int(rng.integers(2, max(3, int(len(c) * rng.choice([0.1, 0.3, 0.8])))))
            p = int(rng.integers(1, max(2, len(c) - k - 1)))
            del c[p:p + k]
        def ins(c):
            k = int(rng.integers(2, 9))
            p = int(rng.integers(1, max(2, len(c) - 1)))
            c[p:p] = synth.random_seq(rng, k)
        if kind == 0:
            cut(cor)
        elif kind == 1:
            ins(cor)
        elif kind == 2:                       # both ends only
            a, b = int(rng.integers(3, 12)), int(rng.integers(3, 12))
            cor = bytearray(ref[:a] + ref[L - b:]) if a + b < L else bytearray(ref)
        elif kind == 3:                       # a far edge and a substitution elsewhere
            cut(cor)
            p = int(rng.integers(0, len(cor)))
            cor[p] = ord("ACGT"[("ACGT".index(chr(cor[p])) + 1) % 4]) if chr(cor[p]) in "ACGT" else cor[p]
        elif kind == 4:                       # two far edges: not for the far instance
            cut(cor)
            ins(cor)
        else:                                 # far edge at the very start / end
            k = int(rng.integers(2, 6))
            cor = bytearray(ref[:1] + ref[1 + k:]) if rng.random() < 0.5 else bytearray(ref[:L - 1 - k] + ref[L - 1:])
        if len(cor) == 0:
            cor = bytearray(ref[:1])
        out.append((ref, bytes(cor), synth.mutate(rng, ref, 0.12) or b"A"))
    return out


def test_far_edge_windows(engine):
    check(engine, far_edge_triples(91, 1800))


def test_far_edge_windows_equal_the_two_kernel_path(engine, monkeypatch):
    """the same windows with the far instance switched off (ELECTOR_NO_FAR: two-kernel path and generic kernels)"""
    triples = far_edge_triples(92, 600)
    got, scores = engine.align(triples, want_scores=True)
    monkeypatch.setenv("ELECTOR_NO_FAR", "1")
    got2, scores2 = engine.align(triples, want_scores=True)
    assert got == got2 and np.array_equal(scores, scores2)


def test_context_on_a_part_of_the_chip():
    """elector_ctx_option "cus" / "priority": a context whose streams sit on a quarter of the compute units (and one at the
    lowest priority) gives the windows the default context gives; the options are refused once the context has run"""
    from elector_amd import poa
    triples = synth.window_triples(93, 3000, 20, 160, err_unc=0.12, err_cor=0.03) + far_edge_triples(94, 300)
    ref = poa.PoaEngine(0)
    want, wscores = ref.align(triples, want_scores=True)
    for name, value in (("cus", 0 * 1000 + 64), ("priority", 1)):
        e = poa.PoaEngine(0)
        e.option(name, value)
        got, scores = e.align(triples, want_scores=True)
        assert got == want and np.array_equal(scores, wscores), name
        with pytest.raises(Exception):
            e.option(name, value)
        e.close()
    with pytest.raises(Exception):
        ref.option("cus", 64 * 1000 + 64)
    ref.close()


def uniform_matrix(match, mismatch, gaps="10 5 5"):
    letters = "A a c g t n".split()
    rows = ["  " + " ".join(letters)]
    for i, a in enumerate(letters):
        rows.append(a + " " + " ".join(str(match if i == j else mismatch) for j in range(len(letters))))
    return "GAP-TRUNCATION-LENGTH=10\nGAP-DECAY-LENGTH=5\nGAP-PENALTIES=%s\n" % gaps + "\n".join(rows) + "\n"


@pytest.mark.parametrize("mismatch,gaps", [(-7, "10 5 5"), (-9, "11 4 4"), (-3, "12 7 7")])
def test_simple_scores_off_the_extension_lattice(tmp_path, mismatch, gaps):
    """Uniform scoring whose numbers are not multiples of the extension penalty: the tight score span (poa_device.h:
    score_span) is not a bound for such a set -- a diagonal that wins by less than an extension is followed by a full
    gap opening (align_lpo_po2.c:374-407) -- so these sets must get the safe span; long un-anchored windows near the
    16-bit range, rows and scores against the oracle with the same matrix."""
    from elector_amd import poa
    path = tmp_path / "u.mat"
    path.write_text(uniform_matrix(0, mismatch, gaps))
    eng = poa.PoaEngine(0, poa.read_params(path))
    par = oracle_lib.read_params(path)
    try:
        for triples in (score_range_triples(73), synth.window_triples(74, 400, 5, 150), synth.adversarial_triples(75, 200)):
            bases, off = synth.pack_windows(triples)
            exp_rows, _, exp_scores, _ = oracle_lib.batch(np.frombuffer(bases, dtype=np.uint8), off, par)
            got, scores = eng.align(triples, want_scores=True)
            bad = [w for w in range(len(triples)) if got[w] != exp_rows[w]]
            assert not bad, "first differing window %d of %d (Lr %d Lc %d Lu %d)" % (
                bad[0], len(bad), len(triples[bad[0]][0]), len(triples[bad[0]][1]), len(triples[bad[0]][2]))
            assert np.array_equal(scores, exp_scores)
    finally:
        eng.close()


def alphabet_shortcut_triples(seed):
    """The windows alignment #1 is settled for without a dynamic program (corrected = reference, one substitution,
    one inserted or deleted letter: k_trivial / trivial_graph) over the WHOLE input alphabet of a2 (create_seq.c:121-132,
    seq_util.c:37-52,253-263): lower case, `n` / `N`, letters outside the alphabet (they all become symbol 0 and print
    as `A`, so two different unknown letters are EQUAL), and the splitter's `N` filler next to them.  The comparisons
    of the shortcut are on symbols; the rows print letters -- both must come out as the reference's."""
    rng = np.random.default_rng(seed)
    unknown = b"RYKMSWBDHV-*.x"
    out = []

    def case(s, p):
        b = bytearray(s)
        for i in range(len(b)):
            if rng.random() < p:
                b[i] = b[i] ^ 0x20 if chr(b[i]).isalpha() else b[i]
        return bytes(b)

    def unc_of(ref):
        u = synth.mutate(rng, ref.upper().replace(b"N", b"A"), 0.15) or b"A"
        return case(u, 0.3)

    for L in (2, 3, 8, 17, 40, 63, 64, 90, 140):
        for rep in range(6):
            ref = bytearray(synth.random_seq(rng, L))
            # `N` runs, lone unknown letters and a homopolymer inside the window
            if L >= 8:
                a = int(rng.integers(0, L - 3))
                ref[a:a + 3] = rng.choice([b"NNN", b"nnn", b"NnN", b"RRY", b"AAA", b"aAa"])
            if L >= 17:
                ref[int(rng.integers(0, L))] = int(rng.choice(np.frombuffer(unknown, dtype=np.uint8)))
            ref = bytes(ref)
            cor_same = case(ref, 0.5)                                        # equal up to case
            # equal up to WHICH unknown letter stands there
            cor_unk = bytes(int(rng.choice(np.frombuffer(unknown, dtype=np.uint8))) if c in unknown else c for c in ref)
            out.append((case(ref, 0.2), cor_same, unc_of(ref)))
            out.append((ref, cor_unk, unc_of(ref)))
            pos = sorted(set(int(x) for x in rng.integers(0, L + 1, 10)) | {0, L - 1, L})
            for p in pos:
                for sub in (b"N", b"n", b"R", b"a", b"C", b"-"):
                    if p < L:
                        out.append((ref, case(ref[:p] + sub + ref[p + 1:], 0.3), unc_of(ref)))      # one substitution (or none, on symbols)
                    out.append((ref, case(ref[:p] + sub + ref[p:], 0.3), unc_of(ref)))              # one inserted letter
                if p < L and L > 1:
                    out.append((case(ref, 0.3), ref[:p] + ref[p + 1:], unc_of(ref)))                # one deleted letter
    # the `N` filler as the corrected side (Master_Splitter.cpp:139-154), next to windows of the kinds above
    for L in (1, 2, 30, 64, 200):
        ref = synth.random_seq(rng, L)
        out += [(ref, b"N", unc_of(ref)), (case(ref, 0.5), b"n", unc_of(ref)), (b"N" * L, b"N", b"N" * max(1, L - 1)),
                (ref[:L // 2] + b"N" + ref[L // 2:], b"N", unc_of(ref))]
    return [t for t in out if len(t[1]) >= 1 and len(t[2]) >= 1]


def test_shortcut_windows_over_the_whole_alphabet(engine, monkeypatch):
    triples = alphabet_shortcut_triples(81)
    assert len(triples) > 4000
    check(engine, triples)
    # and the same with both shortcuts switched off: the dynamic program says the same
    got, scores = engine.align(triples, want_scores=True)
    monkeypatch.setenv("ELECTOR_NO_ONEINDEL", "1")
    monkeypatch.setenv("ELECTOR_NO_TRIVIAL", "1")
    got2, scores2 = engine.align(triples, want_scores=True)
    assert got == got2 and np.array_equal(scores, scores2)


def test_soak_slice(engine):
    """A bounded slice of the builder's parity soak (tools/parity_soak.py) inside the driver's own run: random windows of
    the four length / error mixes, most corrected sequences within one edit of the reference, rows and both scores
    bit-exact against the oracle."""
    total = 0
    for seed in (211, 212, 213):
        for (n, lo, hi, eu, ec) in ((30000, 20, 90, 0.15, 0.01), (20000, 30, 140, 0.12, 0.02), (6000, 100, 400, 0.15, 0.015),
                                    (20000, 5, 40, 0.2, 0.03)):
            triples = synth.window_triples(seed, n, lo, hi, err_unc=eu, err_cor=ec)
            check(engine, triples)
            total += len(triples)
    assert total == 228000


@pytest.mark.parametrize("cls", [-1] + list(range(17)))
def test_first_call_of_a_fresh_process(cls):
    """tests/first_call_check.py in a process of its own: k_poa's first launches, on untouched scratch memory, give the
    rows of the two-kernel path -- with the windows in their own classes (-1) and with every geometry class forced in turn
    (ELECTOR_FORCE_CLASS: all windows that fit go to that instance of k_poa)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    if cls >= 0:
        env["ELECTOR_FORCE_CLASS"] = str(cls)
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, os.path.join(here, "first_call_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "differing 0" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
