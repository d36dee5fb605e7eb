"""elector_amd/csrc/poa_classes.h is shared by the host and the device-side classification: its closed forms (slot tiers,
geometry classes, the class search) are checked on the host against the tables and the loop they replaced."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_class_tables_and_class_search(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "classes_check")
    subprocess.run([hipcc, "-std=c++17", "-O1", "-w", "-I", os.path.join(ROOT, "elector_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                    "-o", exe, os.path.join(ROOT, "tests", "micro", "classes_check.cpp")], check=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
