"""host/par.h: the host mirror's worker pool (parallel_for: per-item sections of the copy-number, split-read and merge passes) and its
persistent task threads (lane drivers, merge workers, the two chains of a run's second half). CPU only."""
import pytest

from contextsv_amd import host


@pytest.mark.parametrize("n_items,threads", [(0, 0), (1, 0), (7, 1), (1000, 0), (1000, 3), (100000, 0)])
def test_pool_selftest(n_items, threads):
    assert host.load().csvhost_par_selftest(n_items, threads) == 0, host.load().csvhost_last_error()
