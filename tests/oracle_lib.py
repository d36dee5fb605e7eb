"""ctypes access to oracle/libpoa_oracle.so (TEST INFRASTRUCTURE: the CPU
restatement of the reference path; never imported by the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libpoa_oracle.so")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")

_lib = None
PARAMS_BYTES = 1 << 18   # >= sizeof(po_params)


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ORACLE_DIR, "poa_oracle.c")):
            build()
        L = C.CDLL(LIB)
        L.po_run_files.argtypes = [C.c_char_p] * 5 + [C.c_int]
        L.po_read_matrix.argtypes = [C.c_char_p, C.c_void_p]
        L.po_write_matrix.argtypes = [C.c_char_p, C.c_void_p]
        L.po_default_params.argtypes = [C.c_void_p]
        L.po_batch.restype = C.c_int64
        L.po_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_void_p, C.c_void_p, C.c_void_p]
        L.po_batch_bundles.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def default_params():
    buf = C.create_string_buffer(PARAMS_BYTES)
    lib().po_default_params(buf)
    return buf


def read_params(path):
    buf = C.create_string_buffer(PARAMS_BYTES)
    n = lib().po_read_matrix(str(path).encode(), buf)
    if n <= 0:
        raise RuntimeError("oracle: cannot read matrix %s (%d)" % (path, n))
    return buf


def write_matrix(path, params=None):
    params = params or default_params()
    if lib().po_write_matrix(str(path).encode(), params):
        raise RuntimeError("oracle: cannot write %s" % path)
    return path


def batch(bases, off, params=None):
    """-> rows list [(ref,cor,unc)], ncol int32[n], scores int32[n,2], cells"""
    params = params or default_params()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    n = (len(off) - 1) // 3
    cap = 3 * int(off[-1]) + 64
    rows = np.empty(cap, dtype=np.uint8)
    row_off = np.zeros(n + 1, dtype=np.int64)
    ncol = np.zeros(n, dtype=np.int32)
    scores = np.zeros((n, 2), dtype=np.int32)
    cells = lib().po_batch(params, n, bases.ctypes.data, off.ctypes.data, rows.ctypes.data, cap,
                           row_off.ctypes.data, ncol.ctypes.data, scores.ctypes.data)
    if cells < 0:
        raise RuntimeError("oracle batch failed")
    buf = rows.tobytes()
    out = []
    for w in range(n):
        a, nc = int(row_off[w]), int(ncol[w])
        out.append((buf[a:a + nc], buf[a + nc:a + 2 * nc], buf[a + 2 * nc:a + 3 * nc]))
    return out, ncol, scores, int(cells)


def batch_bundles(bases, off, minimum_fraction=0.9, params=None):
    """a12 -> list of (consensus rows [bytes], counts [int], bundle ids (ref, cor, unc)) per window"""
    params = params or default_params()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.int64)
    n = (len(off) - 1) // 3
    cap = 4 * int(off[-1]) + 64
    cons = np.zeros(cap, dtype=np.uint8)
    cons_off = np.zeros(n + 1, dtype=np.int64)
    info = np.zeros((n, 8), dtype=np.int32)
    rc = lib().po_batch_bundles(params, n, bases.ctypes.data, off.ctypes.data, float(minimum_fraction),
                                cons.ctypes.data, cap, cons_off.ctypes.data, info.ctypes.data)
    if rc:
        raise RuntimeError("oracle bundles failed (%d)" % rc)
    buf = cons.tobytes()
    out = []
    for w in range(n):
        k, nc, a = int(info[w, 0]), int(info[w, 7]), int(cons_off[w])
        out.append(([buf[a + i * nc:a + (i + 1) * nc] for i in range(k)],
                    [int(x) for x in info[w, 1:1 + k]], tuple(int(x) for x in info[w, 4:7])))
    return out


def run_files(matrix, ref_fa, cor_fa, unc_fa, out_path, with_bundles=False):
    return lib().po_run_files(str(matrix).encode(), str(ref_fa).encode(), str(cor_fa).encode(),
                              str(unc_fa).encode(), str(out_path).encode(), 1 if with_bundles else 0)


def have_reference_binaries():
    return all(os.path.exists(os.path.join(REF_DIR, b)) for b in ("poa", "masterSplitter", "Donatello", "poa_hb"))
