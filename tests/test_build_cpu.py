"""The build's guard against a miscompile seen with ROCm 7.2 (elector_amd/check_spills.py): a VGPR spill store placed in
front of the `s_or_b64 exec` that closes a divergent region runs under the region's narrowed mask; k_poa may not spill at all.
And: the generated header of k_poa's loops is what the generator writes."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elector_amd import check_spills  # noqa: E402

BAD = """
_ZN7elector7k_splitILb1ELi1EEEvNS_9SplitArgsE:
\ts_and_saveexec_b64 s[2:3], vcc
\ts_cbranch_execz .LBB9_209
; %bb.207:
\tglobal_atomic_add_x2 v3, v[4:5], s[86:87]
.LBB9_209:                              ; %Flow5257
\tv_writelane_b32 v127, s60, 12
\tscratch_store_dwordx3 off, v[48:50], off offset:76 ; 12-byte Folded Spill
\ts_nop 0
\ts_or_b64 exec, exec, s[2:3]
\tv_mov_b32_e32 v4, s58
"""

# the store sits in front of a region that opens behind it (k_split<true, 1>): full mask, fine
GOOD = """
_ZN7elector7k_splitILb1ELi1EEEvNS_9SplitArgsE:
.LBB8_10:
\ts_waitcnt vmcnt(0)
\tscratch_store_dwordx4 off, v[4:7], off offset:304 ; 16-byte Folded Spill
\ts_barrier
\ts_mov_b64 s[0:1], exec
\ts_and_b64 s[2:3], s[0:1], s[2:3]
\ts_mov_b64 exec, s[2:3]
\tds_write2_b32 v107, v107, v107 offset0:11 offset1:15
\ts_or_b64 exec, exec, s[0:1]
.LBB8_11:
\tscratch_store_dword off, v1, off offset:8 ; 4-byte Folded Spill
\ts_cbranch_scc1 .LBB8_10
\ts_or_b64 exec, exec, s[4:5]
"""


def test_spill_store_under_narrowed_mask_is_flagged(tmp_path):
    p = tmp_path / "bad.s"
    p.write_text(BAD)
    found = check_spills.check(str(p))
    assert len(found) == 1 and found[0][0].startswith("_ZN7elector7k_split")


def test_spill_store_in_front_of_a_region_is_not(tmp_path):
    p = tmp_path / "good.s"
    p.write_text(GOOD)
    assert check_spills.check(str(p)) == []


def test_any_spill_in_k_poa_is_flagged(tmp_path):
    p = tmp_path / "poa.s"
    p.write_text("_ZN7elector5k_poaILi16ELi6ELb0EEEvNS_8PackArgsE:\n\tscratch_store_dword off, v1, off offset:8 ; 4-byte Folded Spill\n"
                 "\ts_endpgm\n")
    found = check_spills.check(str(p))
    assert len(found) == 1 and "k_poaI" in found[0][0]


def test_generated_header_is_up_to_date():
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_poa_engine.py"), "--check"]).returncode == 0


def test_build_runs_the_scan():
    src = open(os.path.join(ROOT, "elector_amd", "build.py")).read()
    assert "check_spills.check(" in src and "--save-temps" in src
