"""The committed sanitizer / fuzz recipes of the host code ("ASan/UBSan build of host code"; CPU only — the GPU pool takes no
sanitizer runs). `make -C contextsv_amd/csrc asan tsan` builds:
  _obj/fuzz_io_asan               tools/fuzz/fuzz_io.cpp: DEFLATE / BAM / VCF mutation fuzzers under AddressSanitizer + UBSan
  _obj/fuzz_sort_select_asan      tools/fuzz/fuzz_sort_select.cpp: sort_select.h against std::sort on random inputs
  _obj/libcontextsv_host_asan.so  the whole host mirror under the same sanitizers
  _obj/tsan_pool                  tools/fuzz/tsan_pool.cpp: the thread pools and the threaded BGZF reader / writer under ThreadSanitizer
Each gets a short run here (a few hundred inputs per fuzzer; `fuzz_io_asan <mode> <iterations> <seed>` runs longer ones by hand), and the
host-side CPU tests run once more against the sanitized library (LD_PRELOAD of libasan + libstdc++, the interposition order ASan needs
inside a Python process)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "contextsv_amd", "csrc")
OBJ = os.path.join(CSRC, "_obj")


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-s", "-j4", "-C", CSRC, "asan", "tsan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return OBJ


def _run(cmd, env=None, timeout=300):
    e = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1")
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
    assert r.returncode == 0, (cmd, r.stdout[-3000:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("mode,iters", [("inflate", 400), ("bam", 150), ("vcf", 300)])
def test_fuzz_drivers_under_asan(built, mode, iters):
    out = _run([os.path.join(built, "fuzz_io_asan"), mode, str(iters), "20261005"])
    assert out.startswith(mode + ":")


def test_sort_select_against_std_sort_under_asan(built):
    assert "cases equal to std::sort" in _run([os.path.join(built, "fuzz_sort_select_asan"), "400", "20261005"])


def test_thread_pools_under_tsan(built):
    assert "tsan_pool: ok" in _run([os.path.join(built, "tsan_pool")])


def test_host_cpu_tests_against_the_asan_build(built):
    def lib(name):
        return subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    env = {"LD_PRELOAD": lib("libasan.so") + " " + lib("libstdc++.so"), "CONTEXTSV_HOST_LIB": os.path.join(built, "libcontextsv_host_asan.so")}
    tests = ["tests/test_bam_io.py", "tests/test_snp_io.py", "tests/test_fast_inflate.py", "tests/test_vcf_writer.py", "tests/test_merge_host.py", "tests/test_umap_order.py",
             "tests/test_sort_select.py", "tests/test_alt_sequence.py"]
    out = _run([sys.executable, "-m", "pytest", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + tests, env=env, timeout=600)
    assert " passed" in out and " failed" not in out, out[-2000:]
