"""Interval sets the golden vectors do not have (shared by the GPU sweep against the reference's own dbscan.cpp and by the pinning of the
windowed CPU labeller): zero-length and identical intervals, heavy ties, huge next to tiny, nested, coordinates at the top of the domain,
sorted and caller order, every eps / min_pts corner."""
import numpy as np


def interval_sweep(n_cases=160, seed=7):
    rng = np.random.default_rng(seed)
    for it in range(n_cases):
        n = int(rng.choice([1, 2, 3, 17, 64, 65, 257, 600, 1500]))
        kind = it % 5
        if kind == 0:      # clustered around shared loci
            c = rng.integers(0, 2_000_000, max(n // 20, 1))
            s = rng.choice(c, n) + rng.integers(-6, 7, n); L = rng.choice([50, 51, 300, 5000], n) + rng.integers(-3, 4, n)
        elif kind == 1:    # exact duplicates and zero lengths
            s = rng.choice(rng.integers(0, 10_000, 12), n); L = rng.choice([0, 0, 1, 60, 61], n)
        elif kind == 2:    # huge next to tiny
            s = rng.integers(0, 100_000, n); L = rng.choice([1, 2, 90_000, 100_000, 1_000_000], n)
        elif kind == 3:    # nested
            s = 1000 + rng.integers(0, 50, n) * 10; L = 2000 - 2 * (s - 1000) + rng.integers(0, 3, n)
        else:              # top of the coordinate domain (< 2^31)
            s = 2**31 - 3_000_000 + rng.integers(0, 1_000_000, n); L = rng.integers(0, 1_000_000, n)
        s = np.maximum(s, 0).astype(np.uint32); e = (s + np.maximum(L, 0)).astype(np.uint32)
        if it % 2:
            o = np.argsort(s, kind="stable"); s, e = s[o], e[o]
        eps = float(rng.choice([0.0, 0.05, 0.1, 0.3, 0.5, 0.9, 0.999]))
        min_pts = int(rng.choice([1, 2, 3, 5, 6, 50]))
        yield it, s, e, eps, min_pts
