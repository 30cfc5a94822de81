"""The CPU oracle (oracle/csv_oracle.c) against the golden vectors generated from the reference's own
dbscan.cpp / dbscan1d.cpp / kc.cpp (tests/golden/make_golden.py), and — where oracle/_ref is built —
against the reference code live on fresh random inputs. This is what pins the oracle."""
import json
import os

import numpy as np
import pytest

import synth_small as ss

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def test_oracle_dbscan_iv_golden(oracle):
    cases = _load("dbscan_iv.json")["cases"]
    assert len(cases) > 100
    for c in cases:
        s, e = np.asarray(c["start"], np.uint32), np.asarray(c["end"], np.uint32)
        assert oracle.dbscan_iv(s, e, c["eps"], c["min_pts"]).tolist() == c["labels"], (c["seed"], c["eps"], c["min_pts"])


def test_oracle_dbscan_1d_golden(oracle):
    for c in _load("dbscan_1d.json")["cases"]:
        p = np.asarray(c["points"], np.int32)
        lab = oracle.dbscan_1d(p, c["eps"], c["min_pts"])
        assert lab.tolist() == c["labels"]
        assert oracle.largest_cluster(p, lab).tolist() == c["largest"]


def test_oracle_kc_normal_golden(oracle):
    for v in _load("kc_normal.json")["values"]:
        assert oracle.lib.orc_pdf_normal(v["x"], v["mu"], v["sigma"]) == v["pdf"]      # bit-identical doubles
        assert oracle.lib.orc_cdf_normal(v["x"], v["mu"], v["sigma"]) == v["cdf"]


def test_oracle_vs_reference_live(oracle, ref):
    rng = np.random.default_rng(2024)
    for it in range(300):
        n = int(rng.integers(0, 120))
        s, e = ss.random_intervals(5000 + it, n, span=int(rng.choice([2000, 50_000, 2_000_000])), sort=bool(it % 3 == 0), zero_len=bool(it % 7 == 0))
        eps, mp = float(rng.choice([0.0, 0.1, 0.3, 0.5, 0.9])), int(rng.choice([1, 2, 3, 5, 6]))
        assert np.array_equal(oracle.dbscan_iv(s, e, eps, mp), ref.dbscan_iv(s, e, eps, mp))
        p = rng.integers(-300, 3000, n).astype(np.int32)
        eps1 = float(rng.choice([0.0, 10.0, 100.0, 99.5]))
        lab = oracle.dbscan_1d(p, eps1, mp)
        assert np.array_equal(lab, ref.dbscan_1d(p, eps1, mp))
        assert np.array_equal(oracle.largest_cluster(p, lab), ref.largest(p, eps1, mp))
    for x, mu, sd in rng.normal(0, 1, (200, 3)):
        sd = abs(sd) + 0.01
        assert oracle.lib.orc_pdf_normal(x, mu, sd) == ref.lib.ref_pdf_normal(x, mu, sd)
        assert oracle.lib.orc_cdf_normal(x, mu, sd) == ref.lib.ref_cdf_normal(x, mu, sd)
