#!/usr/bin/env python3
"""Generates tests/golden/*.json from the REFERENCE's own translation units (oracle/_ref/libcsvref.so =
/root/reference/src/{dbscan,dbscan1d,kc}.cpp compiled unmodified by oracle/Makefile). Run in the build
container only (needs /root/reference); the JSON files are committed and are what the GPU box and the
CPU suite check against. Inputs are seeded; expected outputs are whatever the reference code returns.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib  # noqa: E402
import synth_small as ss  # noqa: E402


def main():
    ref = oracle_lib.load_ref()
    assert ref is not None, "reference build missing"
    cases = []
    k = 0
    for n in (1, 2, 7, 40, 200, 800, 2000):
        for eps in (0.1, 0.3):
            for min_pts in (1, 2, 3, 5, 6):
                if n >= 800 and min_pts in (1, 6):
                    continue
                for sort in (False, True):
                    k += 1
                    s, e = ss.random_intervals(1000 + k, n, sort=sort, zero_len=(k % 5 == 0), span=400_000 if n <= 200 else 2_000_000)
                    cases.append({"seed": 1000 + k, "eps": eps, "min_pts": min_pts, "start": s.tolist(), "end": e.tolist(),
                                  "labels": ref.dbscan_iv(s, e, eps, min_pts).tolist()})
    json.dump({"source": "reference src/dbscan.cpp DBSCAN::fit via oracle/_ref", "cases": cases},
              open(os.path.join(HERE, "dbscan_iv.json"), "w"))

    rng = np.random.default_rng(77)
    cases = []
    for k in range(120):
        n = int(rng.choice([0, 1, 4, 5, 6, 12, 40, 150, 600]))
        base = int(rng.integers(-500, 100000))
        p = (base + rng.choice([0, 0, 0, 250, 4000], n) + rng.integers(-110, 111, n)).astype(np.int32)
        eps = float(rng.choice([100.0, 100.0, 10.0, 0.0, 99.5]))
        mp = int(rng.choice([5, 5, 1, 2, 3]))
        cases.append({"eps": eps, "min_pts": mp, "points": p.tolist(), "labels": ref.dbscan_1d(p, eps, mp).tolist(),
                      "largest": ref.largest(p, eps, mp).tolist()})
    json.dump({"source": "reference src/dbscan1d.cpp DBSCAN1D::fit / getLargestCluster via oracle/_ref", "cases": cases},
              open(os.path.join(HERE, "dbscan_1d.json"), "w"))

    vals = []
    for x, mu, sd in [(0.0, 0.5, 0.044416), (0.0, 0.5, 0.057305), (0.3, 0.0, 1.0), (-1.2, 0.0, 1.0), (2.5, 0.0, 1.0), (0.0, 0.0, 0.163877),
                      (0.1, 0.333333, 0.166946), (-3.7, -3.739099, 2.564467), (100.0, 100.0, 0.163877), (0.5, 0.25, 0.157236), (1e-9, 0.0, 0.155241)]:
        vals.append({"x": x, "mu": mu, "sigma": sd, "pdf": ref.lib.ref_pdf_normal(x, mu, sd), "cdf": ref.lib.ref_cdf_normal(x, mu, sd)})
    json.dump({"source": "reference src/kc.cpp pdf_normal / cdf_normal via oracle/_ref (17 significant digits)", "values": vals},
              open(os.path.join(HERE, "kc_normal.json"), "w"))
    print("golden vectors written:", [f for f in os.listdir(HERE) if f.endswith(".json")])


if __name__ == "__main__":
    main()
