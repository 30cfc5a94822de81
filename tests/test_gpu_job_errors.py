"""Error paths of the per-chromosome job (include/csvgpu.h csvgpu_chr_job_*): a shard whose signature count outgrows its buffer
(the scan is re-run into a larger one), the same with the larger allocation failing (injected: the shard must keep a usable
buffer + capacity pair and the next job on it must succeed), the limit of open jobs per context, and csvgpu_chr_job_abort keeping
the failure's own message."""
import ctypes as C

import numpy as np
import pytest

import contextsv_amd as cs
from contextsv_amd import Reads, _lib

pytestmark = pytest.mark.gpu
M, I, D = 0, 1, 2


def _many_signature_shard(n_reads=100_000):
    # three 60-base deletions per read: 3e5 signatures > the shard's initial room for max(2^18, 2 n_reads)
    cig = np.tile(np.array([(20 << 4) | M, (60 << 4) | D, (20 << 4) | M, (60 << 4) | D, (20 << 4) | M, (60 << 4) | D, (20 << 4) | M], np.uint32), n_reads)
    off = np.arange(n_reads + 1, dtype=np.uint64) * 7
    pos = (np.arange(n_reads, dtype=np.int32) * 7) % 900_000
    pos.sort()
    return Reads(pos, np.zeros(n_reads, np.uint16), np.full(n_reads, 60, np.uint8), off, cig), 1_000_000


_INJECT = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import numpy as np
import contextsv_amd as cs
from contextsv_amd import _lib
import oracle_lib
from test_gpu_job_errors import _many_signature_shard
oracle = oracle_lib.load_oracle()
reads, depth_len = _many_signature_shard()
exp = oracle.cigar_scan(reads, depth_len)
assert len(exp) == 3 * reads.n_reads
with cs.Context(0) as ctx:
    sh = ctx.upload(reads, depth_len)
    ctx.lib.csvgpu_test_fail_next_alloc(1)
    try:
        sh.pipeline(eps=0.1, min_pts_pct=0.1)
        raise SystemExit("the injected allocation failure did not surface")
    except cs.CsvError as e:
        assert e.status == _lib.CSV_ENOMEM and "signature buffer" in str(e), str(e)
    # the shard still has its old buffer and capacity: the next job grows it for real and gives the oracle's signatures
    res = sh.pipeline(eps=0.1, min_pts_pct=0.1)
    assert res.n_sig == len(exp) and res.n_del == len(exp)
    out = sh.fetch(res)
    for f in ("start", "end", "read", "qpos_kind"):
        assert np.array_equal(out["sig_del"][f], exp[f]), f
    res2 = sh.pipeline(eps=0.1, min_pts_pct=0.1)                      # and once more with the grown buffer (no retry inside)
    assert res2.n_sig == len(exp)
    ctx.lib.csvgpu_test_fail_next_alloc(0)
    sh.free()
print("ok")
"""


def test_signature_buffer_growth(ctx, oracle):
    reads, depth_len = _many_signature_shard()
    exp = oracle.cigar_scan(reads, depth_len)
    sh = ctx.upload(reads, depth_len)
    try:
        res = sh.pipeline(eps=0.1, min_pts_pct=0.1)                       # the scan outgrows the shard's first buffer and is re-run into a larger one
        assert res.n_sig == len(exp) == 3 * reads.n_reads and res.n_del == len(exp)
        out = sh.fetch(res)
        for f in ("start", "end", "read", "qpos_kind"):
            assert np.array_equal(out["sig_del"][f], exp[f]), f
    finally:
        sh.free()


def test_injected_allocation_failure_in_the_testhooks_build():
    """The allocation-failure hook lives in libcsvgpu_testhooks.so only (csvgpu.hip with -DCSV_TEST_HOOKS): a child process loads that build
    (CSVGPU_LIB) and runs the scenario — the larger signature buffer cannot be had, the shard keeps a usable buffer + capacity pair, the
    next job succeeds with the oracle's signatures."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CSVGPU_LIB=os.path.join(root, "contextsv_amd", "lib", "libcsvgpu_testhooks.so"))
    r = subprocess.run([sys.executable, "-c", _INJECT.format(root=root, tests=os.path.join(root, "tests"))], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_open_job_limit_and_abort(ctx):
    reads = Reads.from_cigar_lists([10, 20], [0, 0], [60, 60], [[(M, 100), (D, 60), (M, 100)], [(M, 100)]])
    sh = ctx.upload(reads, 10_000)
    lib = ctx.lib
    jobs = []
    try:
        for _ in range(16):                                                # CSV_MAX_JOBS
            j = lib.csvgpu_chr_job_begin(ctx.h, sh.h, 50, 20, 0.1)
            assert j
            jobs.append(j)
        assert not lib.csvgpu_chr_job_begin(ctx.h, sh.h, 50, 20, 0.1)
        assert b"CSV_MAX_JOBS" in lib.csvgpu_last_error(ctx.h)
        # a failed cluster call (eps outside [0, 1)) then abort: the message of the failure survives the clean-up
        rc = lib.csvgpu_chr_job_cluster(ctx.h, jobs[-1], 1.5, None, None, 0)
        assert rc == _lib.CSV_EINVAL
        assert lib.csvgpu_chr_job_abort(ctx.h, jobs.pop()) == 0
        assert b"eps must be in [0,1)" in lib.csvgpu_last_error(ctx.h)
        j = lib.csvgpu_chr_job_begin(ctx.h, sh.h, 50, 20, 0.1)              # the aborted job's slot is free again
        assert j
        jobs.append(j)
    finally:
        for j in jobs:
            lib.csvgpu_chr_job_abort(ctx.h, j)
        sh.free()
    res = None
    sh = ctx.upload(reads, 10_000)
    try:
        res = sh.pipeline(eps=0.1, min_pts_pct=0.1)
        assert res.n_sig == 1
    finally:
        sh.free()
