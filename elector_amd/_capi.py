"""ctypes binding of the C ABI declared in include/elector_poa.h.

This is the same stub a maintainer would add to the reference to call the
library instead of spawning `bin/poa`.  Loading fails
loudly when the HIP library has not been built: there is no fallback.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# ELECTOR_LIB: another build of the same library (same-box A/B measurements of two builds; never a different ABI)
LIB_PATH = os.environ.get("ELECTOR_LIB") or os.path.join(HERE, "lib", "libelector_poa.so")

ELECTOR_MAX_SYMBOL = 32
ELECTOR_MAX_GAPTAB = 64
ELECTOR_MAX_SEQ = 524000
ES_NCOUNTERS = 25          # include/elector_stats.h

E_IO = -6
E_WINDOW = -7
E_LIMIT = -8
W_OK, W_EMPTY, W_TOOLONG, W_INTERNAL = 0, 1, 2, 3


class ElectorParams(C.Structure):
    _fields_ = [
        ("nsymbol", C.c_int32),
        ("symbol", C.c_char * (ELECTOR_MAX_SYMBOL + 4)),
        ("score", (C.c_int32 * ELECTOR_MAX_SYMBOL) * ELECTOR_MAX_SYMBOL),
        ("max_gap_length", C.c_int32),
        ("gap_penalty_x", C.c_int32 * ELECTOR_MAX_GAPTAB),
        ("gap_penalty_y", C.c_int32 * ELECTOR_MAX_GAPTAB),
    ]


EXPORTS = [
    "elector_version", "elector_strerror", "elector_device_count",
    "elector_params_default", "elector_params_read",
    "elector_ctx_create", "elector_ctx_destroy", "elector_ctx_last_error",
    "elector_poa_batch", "elector_poa_batch_device", "elector_poa_batch_device_offsets", "elector_ctx_sync",
    "elector_ctx_timing_enable", "elector_ctx_timing_read", "elector_ctx_timing_reset",
    "elector_ctx_last_po_sizes", "elector_ctx_option", "elector_ctx_keep_graph", "elector_poa_bundles", "elector_poa_bundles_enqueue",
    "elector_stats_batch", "elector_msa_stats_device", "elector_msa_stats_enqueue", "elector_msa_stats_enqueue_rows", "elector_msa_rows_wait", "elector_msa_stats_collect",
    "elector_msa_rows_fetch", "elector_homopolymer_pairs",
    "elector_split_reads", "elector_windows_free", "elector_merge_windows", "elector_msa_free",
    "elector_split_reads_device", "elector_windows_dev_free", "elector_ctx_copy", "elector_ctx_copy_to_host",
    "elector_reads_open", "elector_reads_set_device", "elector_reads_next", "elector_reads_close", "elector_reads_scan", "elector_reads_index_free",
    "elector_msa_format", "elector_msa_records_write", "elector_msa_records_pwrite",
    "elector_report_aggregate", "elector_report_free", "elector_read_size_lines", "elector_write_count_lines",
]

_lib = None


class ElectorError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = lib().elector_strerror(code).decode()
        super().__init__("elector: %s (%d)%s" % (msg, code, (": " + detail) if detail else ""))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "HIP library %s is missing: run `python -m elector_amd.build` "
            "(this package has no CPU fallback)" % LIB_PATH)
    # One HIP runtime per process: PyTorch ships its own libamdhip64, and a process that loads the
    # system one first leaves torch without devices.  Let torch (when installed) load its copy first;
    # the library then binds to that same runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.elector_version.restype = C.c_char_p
    L.elector_strerror.restype = C.c_char_p
    L.elector_strerror.argtypes = [C.c_int]
    L.elector_device_count.restype = C.c_int
    L.elector_params_default.argtypes = [C.POINTER(ElectorParams)]
    L.elector_params_default.restype = None
    L.elector_params_read.argtypes = [C.c_char_p, C.POINTER(ElectorParams)]
    L.elector_ctx_create.argtypes = [C.c_int, C.POINTER(ElectorParams), C.POINTER(vp)]
    L.elector_ctx_destroy.argtypes = [vp]
    L.elector_ctx_destroy.restype = None
    L.elector_ctx_last_error.argtypes = [vp]
    L.elector_ctx_last_error.restype = C.c_char_p
    L.elector_poa_batch.argtypes = [vp, i64, vp, vp, vp, i64, vp, vp, vp, vp]
    L.elector_poa_batch_device.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
    L.elector_poa_batch_device_offsets.argtypes = [vp, i64, vp, vp, i64, vp, vp, vp, vp]
    L.elector_ctx_sync.argtypes = [vp]
    L.elector_ctx_timing_enable.argtypes = [vp, C.c_int]
    L.elector_ctx_timing_read.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(i64)]
    L.elector_ctx_timing_reset.argtypes = [vp]
    L.elector_ctx_last_po_sizes.argtypes = [vp, i64, vp]
    L.elector_ctx_option.argtypes = [vp, C.c_char_p, i64]
    L.elector_ctx_keep_graph.argtypes = [vp, C.c_int]
    L.elector_poa_bundles.argtypes = [vp, i64, C.c_float, vp, i64, vp, vp]
    L.elector_poa_bundles_enqueue.argtypes = [vp, i64, C.c_float]
    L.elector_msa_stats_device.argtypes = [vp, i64, vp, vp, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, i64]
    L.elector_msa_rows_fetch.argtypes = [vp, i64, vp, vp]
    L.elector_msa_stats_enqueue.argtypes = [vp, i64, vp, vp, vp, i64, vp, i64, vp, vp]
    L.elector_msa_stats_enqueue_rows.argtypes = [vp, i64, vp, vp, vp, i64, vp, i64, vp, vp, vp, i64]
    L.elector_msa_rows_wait.argtypes = [vp]
    L.elector_msa_stats_collect.argtypes = [vp, i64, vp, vp, vp, vp, i64]
    _lib = L
    return L
