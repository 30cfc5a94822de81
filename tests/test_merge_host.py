"""Host logic of the clustering seam (C++ host mirror, no GPU needed): addSVCall order, the representative
choice of mergeSVs given labels, mergeDuplicateSVs — against the oracle restatement (oracle/merge_oracle.cpp)
and the known answer recorded in SURVEY.md §8a row a5."""
import numpy as np
import pytest

import oracle_lib
import synth_small as ss
from contextsv_amd import host

DEL, DUP, INV, INS, BND, UNKNOWN = 0, 1, 2, 3, 4, -1


def _orc_calls(c):
    o = np.zeros(len(c), oracle_lib.CALL_DTYPE)
    for f in ("start", "end", "sv_type", "cluster_size", "hmm_likelihood", "id"):
        o[f] = c[f]
    return o


def _host_merge_all_types(oracle, calls, eps, min_pts, keep_noise):
    """mergeSVs' type loop with labels from the oracle DBSCAN (the GPU-free way to drive the host code)."""
    if len(calls) < 2:
        return calls
    out = []
    for t in (DEL, DUP, INV, INS, BND):
        tc = calls[calls["sv_type"] == t]
        if len(tc) < 2:
            out.append(tc)
            continue
        labels = oracle.dbscan_iv(tc["start"], tc["end"], eps, min_pts)
        out.append(host.merge_type_with_labels(tc, labels, keep_noise))
    return np.concatenate(out) if out else calls[:0]


def _same(a, b):
    assert len(a) == len(b)
    for f in ("start", "end", "sv_type", "cluster_size", "id"):
        assert np.array_equal(a[f], b[f]), f
    assert np.array_equal(a["hmm_likelihood"], b["hmm_likelihood"])


def test_survey_known_answer(oracle):
    """SURVEY §8a a5 [verified against the reference]: 4 mutually distant DELs + 1 INS + 1 UNKNOWN, min_pts=3:
    keep_noise=false -> one DEL 5000-5300 cluster_size=4 + the INS; keep_noise=true -> 4 DELs + the INS; UNKNOWN gone."""
    calls = host.make_calls([1000, 5000, 9000, 13000, 20000, 30000], [1100, 5300, 9200, 13150, 20500, 30400],
                            [DEL, DEL, DEL, DEL, INS, UNKNOWN])
    for merge in (lambda kn: _host_merge_all_types(oracle, calls, 0.1, 3, kn), lambda kn: oracle.merge_svs(_orc_calls(calls), 0.1, 3, kn)):
        m = merge(False)
        assert [(int(x["start"]), int(x["end"]), int(x["sv_type"]), int(x["cluster_size"])) for x in m] == [(5000, 5300, DEL, 4), (20000, 20500, INS, 0)]
        m = merge(True)
        assert [(int(x["start"]), int(x["sv_type"])) for x in m] == [(1000, DEL), (5000, DEL), (9000, DEL), (13000, DEL), (20000, INS)]


@pytest.mark.parametrize("seed", range(12))
def test_host_merge_matches_oracle(oracle, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([0, 1, 2, 30, 400, 3000]))
    s, e = ss.random_intervals(900 + seed, n, sort=bool(seed % 2), span=300_000)
    types = rng.choice([DEL, DEL, INS, INS, DUP, INV, BND, UNKNOWN, 5, 6], n)
    lh = np.where(rng.random(n) < (0.3 if seed % 3 == 0 else 0.0), rng.normal(-50, 10, n), 0.0)
    cs = rng.integers(0, 6, n) if seed % 3 == 0 else np.zeros(n, int)
    calls = host.make_calls(s, e, types, cs, lh)
    for eps, mp, kn in ((0.1, 3, False), (0.1, 2, True), (0.3, 5, False)):
        _same(_host_merge_all_types(oracle, calls, eps, mp, kn), oracle.merge_svs(_orc_calls(calls), eps, mp, kn))


def test_many_length_ties_follow_std_sort(oracle):
    """Clusters whose members tie on length: the representative is whatever libstdc++'s unstable std::sort leaves at
    the slot, on both sides (same library call on the same input order)."""
    rng = np.random.default_rng(5)
    n = 5000
    s = np.sort(rng.integers(1, 2_000_000, n)).astype(np.uint32)
    e = (s + rng.choice([100, 100, 100, 101, 250], n)).astype(np.uint32)
    calls = host.make_calls(s, e, np.full(n, INS))
    _same(_host_merge_all_types(oracle, calls, 0.1, 3, False), oracle.merge_svs(_orc_calls(calls), 0.1, 3, False))


def test_add_sv_call_order():
    # ascending (start,end); equal keys in reverse insertion order; start > end rejected (sv_object.cpp:22-33)
    calls = host.make_calls([50, 10, 50, 10, 70, 50], [60, 20, 60, 15, 65, 55], [DEL] * 6)
    assert host.add_sv_calls_order(calls).tolist() == [3, 1, 5, 2, 0]


def test_merge_duplicates_matches_oracle(oracle):
    rng = np.random.default_rng(9)
    n = 500
    s = rng.integers(1, 60, n).astype(np.uint32)
    e = (s + rng.integers(1, 4, n)).astype(np.uint32)
    calls = host.make_calls(s, e, rng.choice([DEL, INS, INV, UNKNOWN], n), rng.integers(1, 9, n))
    a, b = host.merge_duplicates(calls), oracle.merge_duplicates(_orc_calls(calls))
    _same(a, b)
