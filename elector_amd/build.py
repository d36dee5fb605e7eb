"""Build the HIP shared library in-tree (elector_amd/lib/libelector_poa.so).

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only
container as well as on the GPU box.  Usage: python -m elector_amd.build [--force]
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", f) for f in ("poa_kernels.hip", "poa_fused.hip", "poa_pack.hip", "poa_host.hip", "stats.hip", "bundle.hip", "split_dev.hip", "splitter.cpp", "io_host.cpp", "report_host.cpp")]
HDR = [os.path.join(HERE, "csrc", "poa_device.h"), os.path.join(HERE, "csrc", "ctx.h"), os.path.join(HERE, "csrc", "poa_serial.h"),
       os.path.join(ROOT, "include", "elector_poa.h"), os.path.join(ROOT, "include", "elector_stats.h"),
       os.path.join(ROOT, "include", "elector_split.h")]
OUT = os.path.join(HERE, "lib", "libelector_poa.so")
POA_BIN = os.path.join(HERE, "bin", "poa")


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return os.path.exists(POA_BIN) and all(os.path.getmtime(p) <= t for p in SRC + HDR + [os.path.join(HERE, "csrc", "poa_main.cpp")])


def build(force=False, verbose=False):
    if not force and up_to_date():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc"),
           "-o", OUT] + SRC
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    # the `poa`-compatible executable on top of the library (elector_amd/bin/poa; finds the library through its rpath)
    os.makedirs(os.path.dirname(POA_BIN), exist_ok=True)
    cmd = [hipcc_path(), "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", POA_BIN,
           os.path.join(HERE, "csrc", "poa_main.cpp"), "-L" + os.path.dirname(OUT), "-lelector_poa", "-Wl,-rpath,$ORIGIN/../lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
