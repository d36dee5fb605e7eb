"""No-GPU checks of the product library: it loads, exports every symbol the
headers declare, parses scoring parameters, and REFUSES to compute without a
gfx950 device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import golden_io
from elector_amd import _capi, poa, split

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(elector_[a-z0-9_]+)\s*\(", txt)))


@pytest.mark.parametrize("header", ["elector_poa.h", "elector_split.h", "elector_stats.h"])
def test_every_declared_symbol_is_exported(header):
    if not os.path.exists(os.path.join(ROOT, "include", header)):
        pytest.skip(header + " not present yet")
    L = _capi.lib()
    names = declared_functions(header)
    assert names
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_default_params_match_reference_matrix():
    g = golden_io.params()
    p = poa.default_params()
    assert p.nsymbol == g["nsymbol"] and p.symbol.decode() == g["symbol"]
    assert p.max_gap_length == g["max_gap_length"]
    assert list(p.gap_penalty_x[: p.max_gap_length + 2]) == g["gap_penalty_x"]
    assert list(p.gap_penalty_y[: p.max_gap_length + 2]) == g["gap_penalty_y"]
    assert [[p.score[i][j] for j in range(p.nsymbol)] for i in range(p.nsymbol)] == g["score"]


def test_params_read_matrix_file(tmp_path):
    g = golden_io.params()
    path = tmp_path / "m.mat"
    with open(path, "w") as f:
        f.write("# comment\n\nGAP-TRUNCATION-LENGTH=10\nGAP-DECAY-LENGTH=5\nGAP-PENALTIES=10 5 5\n  ")
        f.write(" ".join(g["symbol"]) + "\n")
        for i, s in enumerate(g["symbol"]):
            f.write(s + " " + " ".join(str(x) for x in g["score"][i]) + " \n")
    p = poa.read_params(path)
    d = poa.default_params()
    assert bytes(p) == bytes(d)
    with pytest.raises(_capi.ElectorError):
        poa.read_params(tmp_path / "missing.mat")
    # decaying penalties + the "-X" directive landing in the y arrays (seq_util.c:119-123)
    with open(path, "w") as f:
        f.write("GAP-TRUNCATION-LENGTH=3\nGAP-DECAY-LENGTH=4\nGAP-PENALTIES=9 4 1\nGAP-PENALTIES-X=7 3 2\n a c\n"
                "a 1 -1\nc -1 1\n")
    p = poa.read_params(path)
    assert p.max_gap_length == 7
    assert list(p.gap_penalty_x[:9]) == [9, 4, 4, 3, 2, 2, 1, 1, 0]
    assert list(p.gap_penalty_y[:9]) == [7, 3, 3, 2, 2, 2, 2, 2, 0]


def test_no_cpu_fallback():
    L = _capi.lib()
    if L.elector_device_count() > 0:
        pytest.skip("a gfx950 device is present")
    with pytest.raises(_capi.ElectorError) as e:
        poa.PoaEngine(0)
    assert e.value.code == -2


def test_pack_windows_layout():
    tr = [(b"ACGT", b"AC", b"A"), (b"G", b"GG", b"GGG")]
    bases, off = poa.pack_windows(tr)
    assert bases.tobytes() == b"ACGTACAGGGGGG"
    assert off.tolist() == [0, 4, 6, 7, 8, 10, 13]


def test_poa_header_text():
    from elector_amd.alignment import _poa_header
    assert _poa_header(b">read7_0") == b">read7_0 untitled"
    assert _poa_header(b">read7_0 some title  ") == b">read7_0 some title  "
    assert _poa_header(b">  spaced   t") == b">spaced t"
    # Donatello then drops the last 11 characters ("_0 untitled") and adds a blank (Donatello.cpp:71-73)
    h = _poa_header(b">read7_0")
    assert h[: len(h) - 11] + b" " == b">read7 "
