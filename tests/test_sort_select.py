"""sort_select.h must return exactly the element libstdc++'s (unstable) std::sort leaves at a slot."""
import numpy as np
import pytest

from contextsv_amd import host


def _patterns(rng, n):
    yield "few_values", rng.integers(0, 4, n)
    yield "clip_lengths", rng.integers(50, 301, n)           # the noise bucket of the CIGAR path
    yield "random", rng.integers(0, 1 << 30, n)
    yield "all_equal", np.full(n, 7)
    yield "ascending", np.arange(n)
    yield "descending", np.arange(n)[::-1]
    yield "organ_pipe", np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]])
    yield "sawtooth", np.arange(n) % 17
    yield "median3_killer_like", np.concatenate([np.arange(0, n, 2), np.arange(1, n, 2)[::-1]])[:n]


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 18, 33, 100, 1000, 4097])
def test_select_equals_std_sort_every_pattern(n):
    rng = np.random.default_rng(n)
    for name, keys in _patterns(rng, n):
        keys = np.ascontiguousarray(keys, np.uint32)
        slots = range(n) if n <= 100 else sorted(set([0, 1, n // 10, n // 5, n // 2, n - 2, n - 1] + rng.integers(0, n, 25).tolist()))
        for p in slots:
            a, b = host.sort_select_check(keys, p)
            assert a == b, (name, n, p)


def test_select_large_noise_bucket_slot():
    rng = np.random.default_rng(1)
    for n in (60_000, 200_003):
        keys = rng.integers(50, 301, n).astype(np.uint32)
        top = max(1, int(n * 0.2))
        a, b = host.sort_select_check(keys, top // 2)
        assert a == b


@pytest.mark.parametrize("n", [17, 64, 65, 127, 128, 129, 191, 192, 193, 257, 1000, 1025, 4097, 20_000])
def test_block_partition_equals_the_library_loop(n):
    """The branch-free block-wise partition (large ranges) makes the same swaps as libstdc++'s __unguarded_partition: same cut, same arrangement."""
    rng = np.random.default_rng(n + 1)
    for name, keys in _patterns(rng, n):
        assert host.partition_check(np.ascontiguousarray(keys, np.uint32)), (name, n)
    for _ in range(20):
        assert host.partition_check(rng.integers(0, int(rng.integers(1, 50)), n).astype(np.uint32))
