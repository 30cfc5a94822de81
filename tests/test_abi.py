"""The C-ABI library loads on a machine without a GPU, exports every symbol include/csvgpu.h declares,
and refuses to create a context when there is no device (no silent CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "csvgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef CSV_TEST_HOOKS.*?#endif", "", text, flags=re.S)          # the test build's hook is not part of the product ABI
    return sorted(set(re.findall(r"\b(csvgpu_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from contextsv_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/csvgpu.h but not exported"
    assert sorted(_lib.ABI) == names, "ctypes table and header disagree"
    assert lib.csvgpu_abi_version() == _lib.ABI_VERSION == 3
    assert not hasattr(lib, "csvgpu_test_fail_next_alloc"), "the allocation-failure hook must not be in the product library"


def test_struct_layouts_match_header():
    from contextsv_amd import _lib
    assert C.sizeof(_lib.csv_reads) == 64 and C.sizeof(_lib.csv_hmm) == 8 * (36 + 6 + 6 + 6 + 1 + 5 + 5 + 1)
    assert _lib.SIG_DTYPE.itemsize == 16 and C.sizeof(_lib.csv_chr_result) == 8 * 4 + 4 + 4 + 8 + 8 * 8


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import contextsv_amd as cs
    with pytest.raises(cs.CsvError) as ei:
        cs.Context(0)
    assert ei.value.status == cs._lib.CSV_ENODEV
    from contextsv_amd import host
    with pytest.raises(RuntimeError):       # the host mirror's DBSCAN has no context -> loud failure, no CPU path
        host.merge_svs(host.make_calls([1, 2], [100, 101], [0, 0]), 0.1, 2, False)


def test_product_package_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "contextsv_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hpp", ".hip")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "libcsvoracle" not in txt and "oracle_lib" not in txt and "csv_oracle" not in txt, os.path.join(dp, f)


def test_testhooks_build_exports_the_same_abi_plus_the_hook():
    """libcsvgpu_testhooks.so (csvgpu.hip with -DCSV_TEST_HOOKS; loaded only by tests/test_gpu_job_errors.py) = the product ABI + the hook."""
    from contextsv_amd import _lib
    lib = C.CDLL(os.path.join(ROOT, "contextsv_amd", "lib", "libcsvgpu_testhooks.so"))
    for n in _declared():
        assert hasattr(lib, n), n
    assert hasattr(lib, "csvgpu_test_fail_next_alloc")
    lib.csvgpu_abi_version.restype = C.c_int
    assert lib.csvgpu_abi_version() == _lib.ABI_VERSION
