"""ctypes bindings of the C-ABI in include/csvgpu.h (plumbing only — no compute happens in Python).

The product path fails loudly when the HIP library is missing or no GPU is usable: there is no
CPU fallback behind these bindings (the oracle under oracle/ is test infrastructure and is never
imported from this package).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CSVGPU_LIB: another build of the same library (tests: lib/libcsvgpu_testhooks.so, which also exports the allocation-failure hook)
LIB_PATH = os.environ.get("CSVGPU_LIB") or os.path.join(_HERE, "lib", "libcsvgpu.so")
ABI_VERSION = 3          # include/csvgpu.h CSVGPU_ABI_VERSION

CSV_OK, CSV_EINVAL, CSV_ENODEV, CSV_ENOMEM, CSV_EHIP, CSV_ECAPACITY = 0, -1, -2, -3, -4, -5
STATUS_NAMES = {0: "CSV_OK", -1: "CSV_EINVAL", -2: "CSV_ENODEV", -3: "CSV_ENOMEM", -4: "CSV_EHIP", -5: "CSV_ECAPACITY"}

K_CIGAR_SCAN, K_DEPTH, K_SORT, K_DBSCAN, K_DBSCAN1D, K_WINDOW, K_VITERBI, K_MISC, K_SPLIT_ORDER, K_COUNT = range(10)
KERNEL_NAMES = ["cigar_scan", "depth", "sort", "dbscan", "dbscan1d", "window", "viterbi", "misc", "split_order"]

SIG_DTYPE = np.dtype([("start", "<u4"), ("end", "<u4"), ("read", "<u4"), ("qpos_kind", "<u4")])
KIND_INS, KIND_DEL, KIND_CLIP = 0, 1, 2


class CsvError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {msg}")
        self.status = status


class csv_reads(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_cigar", C.c_uint64), ("pos", C.c_void_p), ("flag", C.c_void_p),
                ("mapq", C.c_void_p), ("tid", C.c_void_p), ("cigar_off", C.c_void_p), ("cigar", C.c_void_p)]


class csv_hmm(C.Structure):
    _fields_ = [("A", C.c_double * 36), ("pi", C.c_double * 6), ("B1_mean", C.c_double * 6), ("B1_sd", C.c_double * 6),
                ("B1_uf", C.c_double), ("B2_mean", C.c_double * 5), ("B2_sd", C.c_double * 5), ("B2_uf", C.c_double)]


class csv_chr_result(C.Structure):
    _fields_ = [("n_sig", C.c_uint64), ("n_del", C.c_uint64), ("n_ins", C.c_uint64), ("depth_sum", C.c_uint64),
                ("depth_nonzero", C.c_uint32), ("min_pts", C.c_int32), ("mean_cov", C.c_double),
                ("sig_del", C.c_void_p), ("sig_ins", C.c_void_p), ("label_del", C.c_void_p), ("label_ins", C.c_void_p),
                ("depth", C.c_void_p), ("ref_end", C.c_void_p), ("q_start", C.c_void_p), ("q_end", C.c_void_p)]


# every symbol include/csvgpu.h declares: name -> (restype, argtypes)
_P = C.c_void_p
ABI = {
    "csvgpu_create": (_P, [C.c_int, _P]),
    "csvgpu_create_background": (_P, [C.c_int]),
    "csvgpu_destroy": (None, [_P]),
    "csvgpu_abi_version": (C.c_int, []),
    "csvgpu_last_error": (C.c_char_p, [_P]),
    "csvgpu_synchronize": (C.c_int, [_P]),
    "csvgpu_timing_enable": (C.c_int, [_P, C.c_int]),
    "csvgpu_timing_reset": (C.c_int, [_P]),
    "csvgpu_timing_get": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "csvgpu_cigar_scan": (C.c_int, [_P, C.POINTER(csv_reads), C.c_uint32, C.c_uint32, C.c_uint8, _P, C.POINTER(C.c_uint64)]),
    "csvgpu_aln_intervals": (C.c_int, [_P, C.POINTER(csv_reads), _P, _P, _P]),
    "csvgpu_depth": (C.c_int, [_P, C.POINTER(csv_reads), C.c_uint32, _P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "csvgpu_dbscan_iv": (C.c_int, [_P, _P, _P, C.c_uint64, C.c_double, C.c_int32, _P]),
    "csvgpu_dbscan_iv_batch": (C.c_int, [_P, _P, _P, _P, C.c_uint64, C.c_double, C.c_int32, _P]),
    "csvgpu_dbscan_1d": (C.c_int, [_P, _P, _P, C.c_uint64, C.c_double, C.c_int32, _P]),
    "csvgpu_window_log2": (C.c_int, [_P, _P, C.c_uint32, _P, _P, _P, _P, C.c_uint64, C.c_double, _P, _P, _P]),
    "csvgpu_viterbi": (C.c_int, [_P, C.POINTER(csv_hmm), _P, _P, _P, _P, C.c_uint64, _P, _P]),
    "csvgpu_shard_upload": (_P, [_P, C.POINTER(csv_reads), C.c_uint32]),
    "csvgpu_shard_wrap_dev": (_P, [_P, C.POINTER(csv_reads), C.c_uint32]),
    "csvgpu_shard_free": (None, [_P, _P]),
    "csvgpu_chr_pipeline_dev": (C.c_int, [_P, _P, C.c_uint32, C.c_uint8, C.c_double, C.c_double, C.POINTER(csv_chr_result)]),
    "csvgpu_chr_pipeline_fetch": (C.c_int, [_P, _P, C.c_uint32, C.c_uint8, C.c_double, C.c_double, C.POINTER(csv_chr_result), _P, _P, C.c_uint64]),
    "csvgpu_chr_job_begin": (_P, [_P, _P, C.c_uint32, C.c_uint8, C.c_double]),
    "csvgpu_chr_job_cluster": (C.c_int, [_P, _P, C.c_double, _P, _P, C.c_uint64]),
    "csvgpu_chr_job_end": (C.c_int, [_P, _P, C.POINTER(csv_chr_result)]),
    "csvgpu_chr_job_abort": (C.c_int, [_P, _P]),
    "csvgpu_gate_create": (_P, []),
    "csvgpu_gate_open": (C.c_int, [_P, C.c_int]),
    "csvgpu_gate_destroy": (None, [_P]),
    "csvgpu_set_gate": (C.c_int, [_P, _P]),
    "csvgpu_host_alloc": (_P, [_P, C.c_size_t]),
    "csvgpu_host_free": (None, [_P, _P]),
    "csvgpu_aln_intervals_resident": (C.c_int, [_P, _P, _P, _P, _P]),
    "csvgpu_aln_intervals_gather_resident": (C.c_int, [_P, _P, _P, C.c_uint64, _P, _P, _P]),
    "csvgpu_aln_intervals_gather_batch": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P]),
    "csvgpu_shard_set_qname_hash": (C.c_int, [_P, _P, _P]),
    "csvgpu_split_order": (C.c_int, [_P, C.c_int, _P, C.c_uint8, _P, C.c_uint64, _P, C.c_uint64, _P]),
    "csvgpu_split_order_begin": (C.c_int, [_P, C.c_int, _P, C.c_uint8]),
    "csvgpu_split_order_begin_self": (C.c_int, [_P, C.c_int, _P, C.c_uint8]),
    "csvgpu_split_order_finish": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64, _P]),
    "csvgpu_window_log2_resident": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_uint64, C.c_double, _P, _P, _P]),
    "csvgpu_window_log2_resident_many": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "csvgpu_chr_fetch": (C.c_int, [_P, _P, C.POINTER(csv_chr_result), _P, _P]),
    "csvgpu_depth_lookup_resident": (C.c_int, [_P, _P, _P, C.c_uint64, _P]),
    "csvgpu_download": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "csvgpu_dbscan_iv_dev": (C.c_int, [_P, _P, _P, C.c_uint64, C.c_double, C.c_int32, _P]),
    "csvgpu_dbscan_1d_dev": (C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint64, C.c_uint32, C.c_double, C.c_int32, _P]),
    "csvgpu_window_log2_dev": (C.c_int, [_P, _P, C.c_uint32, _P, _P, _P, _P, C.c_uint64, C.c_uint64, C.c_double, _P, _P, _P]),
    "csvgpu_viterbi_dev": (C.c_int, [_P, C.POINTER(csv_hmm), _P, _P, _P, _P, C.c_uint64, C.c_uint64, _P, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load libcsvgpu.so (built in-tree by __graft_entry__.build()). Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"HIP extension not built: {LIB_PATH} is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the product path)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch, which must be loud
        fn.restype = res
        fn.argtypes = args
    got = lib.csvgpu_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} has ABI version {got}, this package expects {ABI_VERSION}: rebuild (__graft_entry__.build())")
    if hasattr(lib, "csvgpu_test_fail_next_alloc"):          # only in the test build (libcsvgpu_testhooks.so via CSVGPU_LIB)
        lib.csvgpu_test_fail_next_alloc.restype = None
        lib.csvgpu_test_fail_next_alloc.argtypes = [C.c_int]
    _lib = lib
    return lib


def ptr(a):
    """Host pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def make_hmm(A, pi, B1_mean, B1_sd, B1_uf, B2_mean, B2_sd, B2_uf) -> csv_hmm:
    h = csv_hmm()
    h.A[:] = list(np.asarray(A, dtype=np.float64).reshape(36))
    h.pi[:] = list(np.asarray(pi, dtype=np.float64))
    h.B1_mean[:] = list(np.asarray(B1_mean, dtype=np.float64))
    h.B1_sd[:] = list(np.asarray(B1_sd, dtype=np.float64))
    h.B1_uf = float(B1_uf)
    h.B2_mean[:] = list(np.asarray(B2_mean, dtype=np.float64))
    h.B2_sd[:] = list(np.asarray(B2_sd, dtype=np.float64))
    h.B2_uf = float(B2_uf)
    return h
