"""Hand-built known answers for the rows whose reference code cannot be built here (sv_caller.cpp,
cnv_caller.cpp, khmm.cpp need htslib): one case per quirk listed in SURVEY.md §8a, worked out by hand
from the reference source. The same cases run against the GPU in test_gpu_kat.py."""
import math

import numpy as np

from contextsv_amd import Reads, make_hmm
from hmm_params import WGS_HMM
from kat_cases import KAT_SCAN, KAT_DEPTH, kat_reads

M, I, D, N, S, H, P, EQ, X = range(9)


def test_scan_known_answers(oracle):
    for name, case in KAT_SCAN.items():
        reads = kat_reads(case)
        sig = oracle.cigar_scan(reads, case["depth_len"], case.get("min_oplen", 50), case.get("min_mapq", 20))
        got = [(int(s["start"]), int(s["end"]), int(s["read"]), int(s["qpos_kind"] >> 2), int(s["qpos_kind"] & 3)) for s in sig]
        assert got == case["expect"], name
        if "intervals" in case:
            re_, qs, qe = oracle.aln_intervals(reads)
            assert list(zip(re_.tolist(), qs.tolist(), qe.tolist())) == case["intervals"], name


def test_depth_known_answers(oracle):
    for name, case in KAT_DEPTH.items():
        reads = kat_reads(case)
        d, s, nz = oracle.depth(reads, case["depth_len"])
        assert d.tolist() == case["depth"], name
        assert (s, nz) == (sum(case["depth"]), sum(1 for x in case["depth"] if x > 0)), name


def test_window_log2_known_answer(oracle):
    depth = np.zeros(200, np.uint32)
    depth[10:110] = 30
    depth[60:80] = 0
    # region 10..109, 20 windows of 5 positions: windows 10 and 11 ... cover 60..69: zero depth -> 1e-9 floor
    l2, ws, we = oracle.window_log2(depth, 10, 109, 20, 30.0)
    assert ws.tolist() == [10 + 5 * i for i in range(20)] and we.tolist() == [15 + 5 * i for i in range(20)]
    for i in range(20):
        lo = 10 + 5 * i
        if 60 <= lo < 80:
            assert l2[i] == math.log2((1e-9 / 5) / 30.0)
        else:
            assert l2[i] == 0.0
    # region running off the end of the depth array: positions >= depth_len are not counted
    l2, ws, we = oracle.window_log2(depth, 190, 229, 20, 30.0)
    assert l2[:5].tolist() == [math.log2((1e-9 / 2) / 30.0)] * 5 and l2[5:].tolist() == [0.0] * 15


def test_viterbi_single_observation_formula(oracle):
    """T = 1, no BAF: loglik = max_i log(pi_i) + log(uf + (1-uf) * N(o; mean_i, sd_i)) with kc.cpp's PI."""
    hmm = make_hmm(**WGS_HMM)
    PI = 3.141592653579893
    for o in (0.0, -0.7, 0.4, -5.0, 3.0):
        oc = min(max(o, WGS_HMM["B1_mean"][0]), WGS_HMM["B1_mean"][5])
        cand = []
        for i in range(6):
            mu, sd = WGS_HMM["B1_mean"][i], WGS_HMM["B1_sd"][i]
            pdf = math.exp(-(oc - mu) * (oc - mu) / (2 * sd * sd)) / (sd * math.sqrt(2 * PI))
            cand.append(math.log(WGS_HMM["pi"][i]) + math.log(0.01 + (1 - 0.01) * pdf))
        st, ll = oracle.viterbi(hmm, [o], [-1.0], [0.5], np.array([0, 1], np.uint64))
        assert abs(ll[0] - max(cand)) < 1e-12 and st[0] == int(np.argmax(cand)) + 1


def test_viterbi_stays_in_state(oracle):
    """Long runs of clean observations decode to the obvious state (sanity of the recursion / backtrack)."""
    hmm = make_hmm(**WGS_HMM)
    T = 50
    for o, b, expect in ((0.0, 0.5, 3), (-0.73, 0.0, 2), (0.4, 0.33, 5), (-3.7, -1.0, 1)):
        o1 = np.full(T, o); o2 = np.full(T, b); pfb = np.full(T, 0.5)
        st, ll = oracle.viterbi(hmm, o1, o2, pfb, np.array([0, T], np.uint64))
        assert (st == expect).all(), (o, b, st)
    st, ll = oracle.viterbi(hmm, [], [], [], np.array([0, 0], np.uint64))
    assert ll[0] == -1e11 and len(st) == 0
