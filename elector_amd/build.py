"""Build the HIP shared library in-tree (elector_amd/lib/libelector_poa.so).

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only
container as well as on the GPU box.  Every source is compiled to an object of
its own (elector_amd/lib/obj/, several at a time, only what changed) and the
objects are linked.  Usage: python -m elector_amd.build [--force]
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from elector_amd import check_spills  # noqa: E402
SRC = [os.path.join(HERE, "csrc", f) for f in ("poa_kernels.hip", "poa_fused.hip", "poa_pack.hip", "poa_classify.hip", "poa_host.hip",
                                               "stats.hip", "bundle.hip", "split_dev.hip", "splitter.cpp", "io_host.cpp",
                                               "report_host.cpp", "rows_dma.cpp")]
HDR = [os.path.join(HERE, "csrc", "poa_device.h"), os.path.join(HERE, "csrc", "ctx.h"), os.path.join(HERE, "csrc", "poa_serial.h"),
       os.path.join(HERE, "csrc", "poa_classes.h"), os.path.join(HERE, "csrc", "poa_engine_gen.h"),
       os.path.join(ROOT, "include", "elector_poa.h"), os.path.join(ROOT, "include", "elector_stats.h"),
       os.path.join(ROOT, "include", "elector_split.h")]
OUT = os.path.join(HERE, "lib", "libelector_poa.so")
OBJ = os.path.join(HERE, "lib", "obj")
POA_BIN = os.path.join(HERE, "bin", "poa")


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


FLAGS_FILE = os.path.join(OBJ, "flags.txt")


def _flags_tag():
    """what the objects were compiled with beyond the fixed flags (ELECTOR_HIPCC_FLAGS): a debug build must not be taken
    for the product build by a later plain build()"""
    return hashlib.sha256(os.environ.get("ELECTOR_HIPCC_FLAGS", "").encode()).hexdigest()[:16]


def _flags_match():
    return os.path.exists(FLAGS_FILE) and open(FLAGS_FILE).read().strip() == _flags_tag()


def up_to_date():
    if not os.path.exists(OUT) or not _flags_match():
        return False
    t = os.path.getmtime(OUT)
    return os.path.exists(POA_BIN) and all(_mtime(p) <= t for p in SRC + HDR + [os.path.join(HERE, "csrc", "poa_main.cpp")])


def regenerate():
    """k_poa's loops are generated assembly (tools/gen_poa_engine.py -> csrc/poa_engine_gen.h, committed): written again when
    the generator is there and newer than the header"""
    gen = os.path.join(ROOT, "tools", "gen_poa_engine.py")
    out = os.path.join(HERE, "csrc", "poa_engine_gen.h")
    if os.path.exists(gen) and _mtime(gen) > _mtime(out):
        subprocess.run([sys.executable, gen], check=True)


def build(force=False, verbose=False):
    regenerate()
    if not force and up_to_date():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    force = force or not _flags_match()
    hipcc = hipcc_path()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread",
             "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")]
    extra = os.environ.get("ELECTOR_HIPCC_FLAGS", "").split()
    newest_hdr = max(_mtime(h) for h in HDR)
    jobs = []
    for s in SRC:
        o = os.path.join(OBJ, os.path.splitext(os.path.basename(s))[0] + ".o")
        if force or _mtime(o) < max(_mtime(s), newest_hdr):
            jobs.append([hipcc] + flags + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    def compile_checked(cmd):
        """Device code is compiled with --save-temps and its assembly scanned for VGPR spill stores that the compiler
        put in front of the `s_or_b64 exec` closing a divergent region (tools/check_spills.py: such a store runs under
        the region's narrowed mask and the reload reads garbage -- seen with ROCm 7.2 in k_poa).  The object is only
        kept when the scan is clean."""
        src, obj = cmd[-3], cmd[-1]
        if not src.endswith(".hip"):
            return run(cmd)
        tmp = obj + ".tmp"
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        try:
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd[:-1] + [os.path.join(tmp, "out.o"), "--save-temps"], check=True, cwd=tmp)
            found = []
            for f in os.listdir(tmp):
                if f.endswith(".s") and "amdgcn" in f:
                    found += check_spills.check(os.path.join(tmp, f))
            if found:
                msg = ("%s: %d VGPR spill store(s) under a narrowed EXEC mask or in a kernel that must not spill (%s); change the "
                       "kernel's register pressure (see elector_amd/check_spills.py; ELECTOR_SKIP_SPILL_CHECK=1 builds anyway)"
                       % (os.path.basename(src), len(found), ", ".join(sorted({k for k, _, _ in found}))))
                if os.environ.get("ELECTOR_SKIP_SPILL_CHECK", "") not in ("", "0"):
                    sys.stderr.write("warning: " + msg + "\n")
                else:
                    raise RuntimeError(msg)
            os.replace(os.path.join(tmp, "out.o"), obj)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=max(1, min(6, (os.cpu_count() or 2) - 1))) as pool:
            list(pool.map(compile_checked, jobs))
    with open(FLAGS_FILE, "w") as f:
        f.write(_flags_tag() + "\n")
    objs = [os.path.join(OBJ, os.path.splitext(os.path.basename(s))[0] + ".o") for s in SRC]
    # (libhsa-runtime64: rows_dma.cpp asks the HSA runtime, which HIP sits on, for the DMA engine)
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", OUT] + objs + ["-lhsa-runtime64"])
    # the `poa`-compatible executable on top of the library (elector_amd/bin/poa; finds the library through its rpath)
    os.makedirs(os.path.dirname(POA_BIN), exist_ok=True)
    run([hipcc, "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", POA_BIN,
         os.path.join(HERE, "csrc", "poa_main.cpp"), "-L" + os.path.dirname(OUT), "-lelector_poa", "-Wl,-rpath,$ORIGIN/../lib"])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
