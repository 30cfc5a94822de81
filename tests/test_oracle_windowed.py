"""Pins oracle/dbscan_window.cpp — the order-free, windowed CPU labeller used where the O(n^2) walk is too slow (chr1's 3e5 INS
signatures) — against the REFERENCE's own src/dbscan.cpp: the committed golden fits with eps < 1 (tests/golden/dbscan_iv.json, generated from
oracle/_ref by tests/golden/make_golden.py), and, where oracle/_ref is built, the 160 sweep inputs (degenerate, nested, tied, top-of-domain
intervals) and fresh random sets live; and against the literal port (orc_dbscan_iv) at a few thousand points."""
import json
import os

import numpy as np
import pytest

import synth_small as ss
from sweep_inputs import interval_sweep

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_windowed_labeller_on_the_golden_fits(oracle):
    cases = json.load(open(os.path.join(G, "dbscan_iv.json")))["cases"]
    assert len(cases) > 100
    n_checked = 0
    for c in cases:
        if not (0.0 <= c["eps"] < 1.0):
            continue
        s, e = np.asarray(c["start"], np.uint32), np.asarray(c["end"], np.uint32)
        assert oracle.dbscan_iv_windowed(s, e, c["eps"], c["min_pts"]).tolist() == c["labels"], (c["seed"], c["eps"], c["min_pts"])
        n_checked += 1
    assert n_checked > 100


def test_windowed_labeller_vs_reference_code_live(oracle, ref):
    for it, s, e, eps, min_pts in interval_sweep():
        assert np.array_equal(oracle.dbscan_iv_windowed(s, e, eps, min_pts), ref.dbscan_iv(s, e, eps, min_pts)), (it, len(s), eps, min_pts)
    rng = np.random.default_rng(99)
    for it in range(200):
        n = int(rng.integers(0, 400))
        s, e = ss.random_intervals(9000 + it, n, span=int(rng.choice([2000, 50_000, 2_000_000])), sort=bool(it % 3 == 0), zero_len=bool(it % 5 == 0))
        eps, mp = float(rng.choice([0.0, 0.1, 0.3, 0.5, 0.9])), int(rng.choice([1, 2, 3, 5, 6]))
        assert np.array_equal(oracle.dbscan_iv_windowed(s, e, eps, mp), ref.dbscan_iv(s, e, eps, mp)), (it, n, eps, mp)


@pytest.mark.parametrize("n,eps,min_pts,kw", [(4000, 0.1, 3, {}), (6000, 0.1, 2, dict(sort=True)), (5000, 0.3, 5, dict(zero_len=True)), (3000, 0.5, 6, dict(clustered=False, span=30000))])
def test_windowed_labeller_vs_the_literal_port(oracle, n, eps, min_pts, kw):
    s, e = ss.random_intervals(31 + n, n, **kw)
    assert np.array_equal(oracle.dbscan_iv_windowed(s, e, eps, min_pts), oracle.dbscan_iv(s, e, eps, min_pts))
    with pytest.raises(ValueError):
        oracle.dbscan_iv_windowed(s, e, 1.0, min_pts)
