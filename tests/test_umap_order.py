"""host/umap_order.h replays what libstdc++'s _Hashtable does to its node list, so that the split-read pass gets the iteration
order of the reference's unordered_map<std::string, PrimaryAlignment> (src/sv_caller.cpp:216, :224) without building the map.
Here it is checked against the real container: same surviving keys in the same order, same bucket count, for key sets that cross
many rehashes, contain duplicates, collide in buckets, and have most keys erased afterwards (as the reference erases every primary
without a supplementary record, :183-202). CPU only."""
import numpy as np
import pytest

from contextsv_amd import host


def _check(keys, erase=None):
    a, b, (ba, bb) = host.umap_order_check(keys, erase)
    assert ba == bb
    assert len(a) == len(b)
    assert np.array_equal(a, b)
    return len(a)


def test_empty_and_tiny():
    assert _check([]) == 0
    assert _check(["x"]) == 1
    assert _check(["x", "x", "x"]) == 1
    assert _check(["a", "b"], [1, 0]) == 1


@pytest.mark.parametrize("seed", range(12))
def test_random_names(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([5, 13, 14, 29, 30, 59, 200, 1000, 5000, 40000]))
    style = seed % 3
    if style == 0:
        keys = ["r%d" % int(x) for x in rng.integers(0, max(2, n // 2), n)]          # many duplicates
    elif style == 1:
        keys = ["read/%x/ccs" % int(x) for x in rng.integers(0, 1 << 60, n)]
    else:
        keys = ["r7_%d" % i for i in range(n)]                                        # the benchmark's naming, insertion in id order
    erase = (rng.random(n) < rng.choice([0.0, 0.5, 0.99])).astype(np.uint8)
    _check(keys, erase)


def test_every_size_up_to_a_few_rehashes():
    keys = ["k%d" % i for i in range(700)]
    for n in list(range(0, 70)) + [126, 127, 128, 256, 257, 258, 540, 541, 542, 700]:
        _check(keys[:n])


def test_large_with_almost_everything_erased():
    n = 300_000
    keys = ["r1_%d" % i for i in range(n)]
    erase = np.ones(n, np.uint8)
    erase[:: 97] = 0
    assert _check(keys, erase) == len(range(0, n, 97))
