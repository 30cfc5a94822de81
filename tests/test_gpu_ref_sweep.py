"""GPU interval / 1-D DBSCAN against the REFERENCE's own dbscan.cpp / dbscan1d.cpp (compiled unmodified into oracle/_ref) on
inputs the golden vectors do not have: zero-length and identical intervals, heavy ties, huge next to tiny, nested, coordinates at
the top of the domain, sorted and caller order, every eps / min_pts corner; 1-D points over the whole int32 range."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_interval_dbscan_vs_reference_code(ctx, ref):
    from sweep_inputs import interval_sweep
    for it, s, e, eps, min_pts in interval_sweep():
        assert np.array_equal(ctx.dbscan_iv(s, e, eps, min_pts), ref.dbscan_iv(s, e, eps, min_pts)), (it, len(s), eps, min_pts)


def test_dbscan1d_vs_reference_code(ctx, ref):
    rng = np.random.default_rng(8)
    for it in range(120):
        n = int(rng.choice([1, 2, 5, 64, 65, 300, 512, 513, 1500]))
        p = [rng.integers(-2**31, 2**31 - 1, n), rng.integers(0, 3000, n), rng.choice(rng.integers(0, 10**6, 5), n) + rng.integers(-120, 121, n)][it % 3]
        p = p.astype(np.int32)
        eps = float(rng.choice([0.0, 1.0, 100.0, 100.5, 1e9]))
        min_pts = int(rng.choice([1, 2, 5, 6]))
        got = ctx.dbscan_1d(p, np.array([0, n], np.uint64), eps, min_pts)
        assert np.array_equal(got, ref.dbscan_1d(p, eps, min_pts)), (it, n, eps, min_pts)


def test_interval_dbscan_batch_against_reference(ctx, ref):
    """csvgpu_dbscan_iv_batch (one workgroup per set, all pairs in LDS, caller order) against the reference's own dbscan.cpp set by
    set: unsorted starts, duplicates, zero-length and nested intervals, every min_pts the path uses, set sizes on both sides of the
    2048-point limit of the LDS kernel (larger sets take the windowed path), empty sets in between."""
    rng = np.random.default_rng(77)
    sets = []
    for n in [0, 1, 2, 3, 17, 64, 65, 300, 1000, 2047, 2048, 2049, 3000, 0, 5]:
        centres = rng.integers(1000, 200_000, max(1, n // 7 + 1))
        c = centres[rng.integers(0, len(centres), n)]
        length = rng.choice([0, 1, 49, 50, 300, 2000, 40_000], n, p=[.02, .02, .06, .3, .3, .2, .1])
        jit = rng.integers(-8, 9, n)
        s = np.maximum(1, c + jit).astype(np.uint32)
        e = (s + np.maximum(0, length + rng.integers(-3, 4, n))).astype(np.uint32)
        sets.append((s, e))
    off = np.zeros(len(sets) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s, _ in sets])
    S = np.concatenate([s for s, _ in sets]); E = np.concatenate([e for _, e in sets])
    for eps, min_pts in ((0.1, 2), (0.1, 3), (0.3, 5), (0.0, 1)):
        got = ctx.dbscan_iv_batch(S, E, off, eps, min_pts)
        for k, (s, e) in enumerate(sets):
            exp = ref.dbscan_iv(s, e, eps, min_pts)
            assert np.array_equal(got[int(off[k]): int(off[k + 1])], exp), (k, len(s), eps, min_pts)


def test_interval_dbscan_batch_sorted_equals_all_pairs(ctx):
    """The small-set kernel meets every neighbour pair once in start order; CSV_DBSCAN_SMALL_BRUTE=1 selects the all-pairs kernel it replaced
    (itself pinned against the reference above). Same labels on nested piles, duplicates, zero-length intervals, and on coordinates beyond
    2^31 (the predicate compares as int, like the reference's arithmetic: the start order has to be the signed one)."""
    import os
    rng = np.random.default_rng(123)
    sets = []
    for it in range(60):
        n = int(rng.choice([1, 2, 7, 63, 64, 65, 500, 1500, 2048]))
        kind = it % 4
        if kind == 0:                                   # clustered calls around a few centres
            c = rng.choice(rng.integers(1000, 5_000_000, max(1, n // 9 + 1)), n)
            s = np.maximum(1, c + rng.integers(-30, 31, n)).astype(np.uint32)
            e = (s + rng.choice([0, 1, 50, 51, 300, 5000, 200_000], n)).astype(np.uint32)
        elif kind == 1:                                 # one long interval over many short ones
            s = rng.integers(10_000, 60_000, n).astype(np.uint32); e = (s + rng.integers(1, 400, n)).astype(np.uint32)
            s[0] = 9_000; e[0] = 70_000
        elif kind == 2:                                 # heavy duplicates
            s = rng.choice(rng.integers(1, 1000, 5), n).astype(np.uint32); e = (s + rng.choice([100, 100, 101, 0], n)).astype(np.uint32)
        else:                                           # around 2^31 and near 2^32
            base = int(rng.choice([2**31 - 500, 2**32 - 100_000]))
            s = base + rng.integers(0, 900, n).astype(np.int64); e = s + rng.integers(0, 600, n).astype(np.int64)
            s = (s % (1 << 32)).astype(np.uint32); e = (e % (1 << 32)).astype(np.uint32)
        sets.append((s, e))
    off = np.zeros(len(sets) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s, _ in sets])
    S = np.concatenate([s for s, _ in sets]); E = np.concatenate([e for _, e in sets])
    for eps, min_pts in ((0.1, 2), (0.25, 4), (0.0, 1), (0.999, 2)):
        got = ctx.dbscan_iv_batch(S, E, off, eps, min_pts)
        os.environ["CSV_DBSCAN_SMALL_BRUTE"] = "1"
        try:
            exp = ctx.dbscan_iv_batch(S, E, off, eps, min_pts)
        finally:
            del os.environ["CSV_DBSCAN_SMALL_BRUTE"]
        bad = np.flatnonzero(got != exp)
        assert len(bad) == 0, (eps, min_pts, int(np.searchsorted(off, bad[0], side="right") - 1), bad[:5])
