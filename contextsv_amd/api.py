"""Host-side mirror of the reference's interface for the clustering / HMM seams, with the same
names, argument meaning and error behaviour, so the parity tests read like tests of the reference:

    DBSCAN(eps, minPts).fit(sv_calls); .getClusters()                    include/dbscan.h:11-33
    DBSCAN1D(eps, minPts).fit(points); .getClusters(); .getLargestCluster(points)   include/dbscan1d.h:11-32
    ReadCHMM(filename) -> CHMM;  testVit_CHMM(hmm, T, O1, O2, pfb) -> (states, loglik)   include/khmm.h:14-49

Every fit()/testVit_CHMM call goes through the C-ABI to the HIP kernels; nothing is computed here
except getLargestCluster's bookkeeping over the returned labels (dbscan1d.cpp:72-90 — a std::map walk).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .context import Context

_default_ctx: Context | None = None


def default_context() -> Context:
    """Lazily created Context on device 0 (raises CsvError(CSV_ENODEV) without a GPU)."""
    global _default_ctx
    if _default_ctx is None or _default_ctx.h is None:
        _default_ctx = Context(0)
    return _default_ctx


class DBSCAN:
    def __init__(self, epsilon: float, minPts: int, ctx: Context | None = None):
        self.epsilon, self.minPts = float(epsilon), int(minPts)
        self._ctx = ctx
        self._clusters = np.zeros(0, np.int32)

    def fit(self, sv_calls):
        """sv_calls: sequence of objects/tuples with (start, end) or a (n,2) array."""
        a = np.asarray([(c.start, c.end) if hasattr(c, "start") else (c[0], c[1]) for c in sv_calls], dtype=np.uint32).reshape(-1, 2)
        ctx = self._ctx or default_context()
        self._clusters = ctx.dbscan_iv(a[:, 0].copy(), a[:, 1].copy(), self.epsilon, self.minPts)

    def getClusters(self):
        return self._clusters


class DBSCAN1D:
    def __init__(self, epsilon: float, minPts: int, ctx: Context | None = None):
        self.epsilon, self.minPts = float(epsilon), int(minPts)
        self._ctx = ctx
        self._clusters = np.zeros(0, np.int32)

    def fit(self, points):
        pts = np.asarray(points, dtype=np.int32)
        ctx = self._ctx or default_context()
        self._clusters = ctx.dbscan_1d(pts, np.array([0, len(pts)], np.uint64), self.epsilon, self.minPts) if len(pts) else np.zeros(0, np.int32)

    def getClusters(self):
        return self._clusters

    def getLargestCluster(self, points):
        """dbscan1d.cpp:72-90: members (index order) of the largest cluster; ties -> lowest id; none -> []."""
        return largest_cluster(np.asarray(points, dtype=np.int32), self._clusters)


def largest_cluster(points: np.ndarray, clusters: np.ndarray) -> np.ndarray:
    if len(clusters) == 0 or clusters.max(initial=-1) < 0:
        return np.zeros(0, np.int32)
    ids, counts = np.unique(clusters[clusters >= 0], return_counts=True)   # ascending ids
    best = ids[np.argmax(counts)]                                         # first maximum = lowest id
    return points[clusters == best]


@dataclass
class CHMM:
    """Fields ReadCHMM fills (khmm.h:14-32)."""
    N: int = 0
    M: int = 0
    A: np.ndarray = field(default_factory=lambda: np.zeros((6, 6)))
    B: np.ndarray = field(default_factory=lambda: np.zeros((6, 6)))
    pi: np.ndarray = field(default_factory=lambda: np.zeros(6))
    B1_mean: np.ndarray = field(default_factory=lambda: np.zeros(6))
    B1_sd: np.ndarray = field(default_factory=lambda: np.zeros(6))
    B1_uf: float = 0.0
    B2_mean: np.ndarray = field(default_factory=lambda: np.zeros(5))
    B2_sd: np.ndarray = field(default_factory=lambda: np.zeros(5))
    B2_uf: float = 0.0

    def c_struct(self) -> _lib.csv_hmm:
        if self.N != 6:
            raise ValueError("the device Viterbi is a fixed 6-state DP (N must be 6)")
        return _lib.make_hmm(self.A, self.pi, self.B1_mean, self.B1_sd, self.B1_uf, self.B2_mean, self.B2_sd, self.B2_uf)


def ReadCHMM(filename: str) -> CHMM:
    """Parser with the reference's grammar and stopping point (khmm.cpp:395-553): M=, N=, A:, B:, pi:,
    B1_mean:, B1_sd:, B1_uf:, B2_mean: (5 values), B2_sd: (5 values), B2_uf:; anything after is ignored.
    On a malformed file the reference logs and returns an empty CHMM(); so does this."""
    try:
        with open(filename) as f:
            toks = f.read().split()
    except OSError:
        return CHMM()
    it = iter(toks)

    def expect(tag):
        return next(it, None) == tag

    def nums(k):
        return np.array([float(next(it)) for _ in range(k)], dtype=np.float64)

    try:
        h = CHMM()
        t = next(it)
        if not t.startswith("M="):
            return CHMM()
        h.M = int(t[2:])
        t = next(it)
        if not t.startswith("N="):
            return CHMM()
        h.N = int(t[2:])
        if not expect("A:"):
            return CHMM()
        h.A = nums(h.N * h.N).reshape(h.N, h.N)
        if not expect("B:"):
            return CHMM()
        h.B = nums(h.N * h.M).reshape(h.N, h.M)
        if not expect("pi:"):
            return CHMM()
        h.pi = nums(h.N)
        if not expect("B1_mean:"):
            return CHMM()
        h.B1_mean = nums(h.N)
        if not expect("B1_sd:"):
            return CHMM()
        h.B1_sd = nums(h.N)
        if not expect("B1_uf:"):
            return CHMM()
        h.B1_uf = float(next(it))
        if not expect("B2_mean:"):
            return CHMM()
        h.B2_mean = nums(5)
        if not expect("B2_sd:"):
            return CHMM()
        h.B2_sd = nums(5)
        if not expect("B2_uf:"):
            return CHMM()
        h.B2_uf = float(next(it))
        return h
    except (StopIteration, ValueError):
        return CHMM()


def testVit_CHMM(hmm: CHMM, T: int, O1, O2, pfb, ctx: Context | None = None):
    """-> (states 1..6 of length T, max final log-likelihood) like khmm.cpp:28-56."""
    ctx = ctx or default_context()
    o1 = np.asarray(O1, dtype=np.float64)[:T]
    o2 = np.asarray(O2, dtype=np.float64)[:T]
    p = np.asarray(pfb, dtype=np.float64)[:T]
    states, ll = ctx.viterbi(hmm.c_struct(), o1, o2, p, np.array([0, T], np.uint64))
    return states, float(ll[0])


testVit_CHMM.__test__ = False   # keep pytest from collecting the reference-named function as a test
