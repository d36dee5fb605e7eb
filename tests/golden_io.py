"""Readers for the fixtures under tests/golden/ (written by oracle/make_golden.py
from the real reference binaries)."""
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def windows(name):
    """-> [((ref, cor, unc), (row_ref, row_cor, row_unc))]"""
    out = []
    for ln in open(os.path.join(GOLD, name), "rb").read().split(b"\n"):
        if not ln or ln.startswith(b"#"):
            continue
        f = ln.split(b"\t")
        out.append(((f[0], f[1], f[2]), (f[3], f[4], f[5])))
    return out


def bundles():
    """-> [((ref, cor, unc), [(header, row), ...])]"""
    out = []
    for ln in open(os.path.join(GOLD, "bundles.tsv"), "rb").read().split(b"\n"):
        if not ln or ln.startswith(b"#"):
            continue
        f = ln.split(b"\t")
        recs = [(f[i], f[i + 1]) for i in range(3, len(f), 2)]
        out.append(((f[0], f[1], f[2]), recs))
    return out


def bundle_expectation(recs):
    """records of one window in bundles.tsv -> (consensus rows, 'containing N seqs' counts)"""
    rows, counts = [], []
    for h, r in recs[3:]:
        assert h.startswith(b">CONSENS")
        rows.append(r)
        counts.append(int(h.split(b"containing ")[1].split()[0]))
    return rows, counts


def splitter():
    """-> reads [(header, (ref, cor, unc))], windows [(header, (ref, cor, unc))], small, wrong"""
    reads, wins, small, wrong = [], [], 0, 0
    for ln in open(os.path.join(GOLD, "splitter_reads.tsv"), "rb").read().split(b"\n"):
        if not ln or ln.startswith(b"#"):
            continue
        f = ln.split(b"\t")
        if f[0] == b"R":
            reads.append((f[1], (f[2], f[3], f[4])))
        elif f[0] == b"W":
            wins.append((f[1], (f[2], f[3], f[4])))
        elif f[0] == b"C":
            small, wrong = int(f[1]), int(f[2])
    return reads, wins, small, wrong


def merger():
    """-> (poa output lines, Donatello output lines)"""
    s, m = [], []
    for ln in open(os.path.join(GOLD, "merger.tsv"), "rb").read().split(b"\n"):
        if ln.startswith(b"S\t"):
            s.append(ln[2:])
        elif ln.startswith(b"M\t"):
            m.append(ln[2:])
    return s, m


def params():
    return json.load(open(os.path.join(GOLD, "params.json")))
