"""oracle/ref_import.py -- TEST INFRASTRUCTURE ONLY, container-only.

Imports the REAL reference module elector/computeStats.py from /root/reference
(read-only; bytecode writing disabled so nothing is created there) with a stub
`Bio` package -- the module imports Bio.SeqIO but never uses it
(computeStats.py:25).  Used to pin oracle/stats_oracle.py and to generate
tests/golden/stats_*.json.  /root/reference does not exist on the GPU box:
nothing that runs there may call this.
"""
import contextlib
import io
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def available():
    return os.path.exists(os.path.join(REFERENCE_ROOT, "elector", "computeStats.py"))


def load_compute_stats():
    sys.dont_write_bytecode = True
    if "Bio" not in sys.modules:
        bio = types.ModuleType("Bio")
        bio.SeqIO = types.ModuleType("Bio.SeqIO")
        sys.modules["Bio"] = bio
        sys.modules["Bio.SeqIO"] = bio.SeqIO
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import elector.computeStats as cs   # the reference package, not ours
    return cs


def run_reference(msa_path, corrected_fasta, out_dir, small, wrong, threshold=5, size_thr=0.1, clips=None,
                  soft=None):
    """Call the reference's outputRecallPrecision exactly as elector/__main__.py:141
    does; msa_path must be out_dir/msa.fa (or msa_<soft>.fa).  Returns (tuple, stdout, log)."""
    cs = load_compute_stats()
    log = io.StringIO()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        tup = cs.outputRecallPrecision(corrected_fasta, out_dir, log, small, wrong, threshold, size_thr,
                                       "read_size_distribution.txt" if soft is None else soft + "_read_size_distribution.txt",
                                       clips or {}, 0, 0, soft)
    return tup, buf.getvalue(), log.getvalue()
