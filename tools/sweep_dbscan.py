"""One-off sweep: GPU interval / 1-D DBSCAN against the REFERENCE's own dbscan.cpp / dbscan1d.cpp (oracle/_ref) on odd inputs:
zero-length and equal intervals, heavy ties, wide and narrow mixes, sorted and unsorted order, every eps / min_pts corner."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import contextsv_amd as cs
import oracle_lib
ref = oracle_lib.load_ref()
ctx = cs.Context(0)
rng = np.random.default_rng(7)
bad = 0
for it in range(400):
    n = int(rng.choice([1, 2, 3, 17, 64, 65, 257, 600, 2000]))
    kind = int(rng.integers(0, 5))
    if kind == 0:      # clustered
        c = rng.integers(0, 2_000_000, max(n // 20, 1)); s = rng.choice(c, n) + rng.integers(-6, 7, n); L = rng.choice([50, 51, 300, 5000], n) + rng.integers(-3, 4, n)
    elif kind == 1:    # many exact duplicates and zero lengths
        s = rng.choice(rng.integers(0, 10_000, 12), n); L = rng.choice([0, 0, 1, 60, 61], n)
    elif kind == 2:    # huge next to tiny
        s = rng.integers(0, 100_000, n); L = rng.choice([1, 2, 90_000, 100_000, 1_000_000], n)
    elif kind == 3:    # nested
        s = 1000 + rng.integers(0, 50, n) * 10; L = 2000 - 2 * (s - 1000) + rng.integers(0, 3, n)
    else:              # coordinates near the top of the domain
        s = 2**31 - 3_000_000 + rng.integers(0, 1_000_000, n); L = rng.integers(0, 1_000_000, n)
    s = np.maximum(s, 0).astype(np.uint32); e = (s + np.maximum(L, 0)).astype(np.uint32)
    if rng.random() < 0.5:
        o = np.argsort(s, kind='stable'); s, e = s[o], e[o]
    eps = float(rng.choice([0.0, 0.05, 0.1, 0.3, 0.5, 0.9, 0.999]))
    mp = int(rng.choice([1, 2, 3, 5, 6, 50]))
    g = ctx.dbscan_iv(s, e, eps, mp); r = ref.dbscan_iv(s, e, eps, mp)
    if not np.array_equal(g, r):
        bad += 1
        print('IV FAIL', it, n, kind, eps, mp, int((g != r).sum()))
for it in range(300):
    n = int(rng.choice([1, 2, 5, 64, 65, 300, 512, 513, 1500]))
    p = rng.choice([rng.integers(-2**31, 2**31 - 1, n), rng.integers(0, 3000, n), rng.choice(rng.integers(0, 10**6, 5), n) + rng.integers(-120, 121, n)])
    eps = float(rng.choice([0.0, 1.0, 100.0, 100.5, 1e9]))
    mp = int(rng.choice([1, 2, 5, 6]))
    seg = np.array([0, n], np.uint64)
    g = ctx.dbscan_1d(p.astype(np.int32), seg, eps, mp); r = ref.dbscan_1d(p.astype(np.int32), eps, mp)
    if not np.array_equal(g, r):
        bad += 1
        print('1D FAIL', it, n, eps, mp, int((g != r).sum()))
print('done, failures:', bad)
