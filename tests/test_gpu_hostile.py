"""Inputs no aligner writes but the seams must survive with the oracle's answers: zero-length ops, reads hanging over or lying
beyond the contig end, positions that wrap in uint32, empty CIGARs, every record filtered, a one-position contig, unsorted shards."""
import numpy as np
import pytest

from contextsv_amd import Reads

pytestmark = pytest.mark.gpu
M, I, D, N, S, H, P, EQ, X = range(9)


def _hostile(seed, n_reads, depth_len, sorted_pos=True):
    rng = np.random.default_rng(seed)
    pos = rng.integers(-1, depth_len + 300, n_reads)
    pos[rng.random(n_reads) < 0.02] = 2**31 - 20_000_000 - rng.integers(0, 50)  # far beyond any contig; every coordinate stays below 2^31 (the seams' documented domain)
    if sorted_pos:
        pos.sort()
    flag = rng.choice([0, 0, 0, 16, 4, 256, 512, 1024, 2048, 2064], n_reads)
    mapq = rng.choice([0, 19, 20, 60, 255], n_reads)
    cig = []
    for r in range(n_reads):
        k = int(rng.choice([0, 1, 2, 5, 40, 300, 700]))
        ops = []
        for _ in range(k):
            op = int(rng.choice([M, M, M, I, D, N, S, H, P, EQ, X]))
            ln = int(rng.choice([0, 1, 3, 49, 50, 51, 200, 5000]))
            ops.append((op, ln))
        cig.append(ops)
    return Reads.from_cigar_lists(pos, flag.astype(np.uint16), mapq.astype(np.uint8), cig)


def _same_sigs(a, b):
    assert len(a) == len(b)
    for f in ("start", "end", "read", "qpos_kind"):
        assert np.array_equal(a[f], b[f]), f


@pytest.mark.parametrize("seed,n_reads,depth_len,sorted_pos", [(1, 400, 3000, True), (2, 400, 3000, False), (3, 64, 1, True),
                                                              (4, 900, 70_000, True), (5, 30, 17, False)])
def test_seams_on_hostile_shards(ctx, oracle, seed, n_reads, depth_len, sorted_pos):
    reads = _hostile(seed, n_reads, depth_len, sorted_pos)
    for min_oplen, min_mapq in ((50, 20), (1, 0)):
        _same_sigs(ctx.cigar_scan(reads, depth_len, min_oplen, min_mapq), oracle.cigar_scan(reads, depth_len, min_oplen, min_mapq))
    for g, o in zip(ctx.aln_intervals(reads), oracle.aln_intervals(reads)):
        assert np.array_equal(g, o)
    d, s, nz = ctx.depth(reads, depth_len)
    od, os_, onz = oracle.depth(reads, depth_len)
    assert np.array_equal(d, od) and (s, nz) == (os_, onz)
    sh = ctx.upload(reads, depth_len)
    try:
        for pct in (0.1, 0.0):
            res = sh.pipeline(eps=0.1, min_pts_pct=pct)
            out = sh.fetch(res, want_depth=True)
            sig = oracle.cigar_scan(reads, depth_len)
            kind = sig["qpos_kind"] & 3
            _same_sigs(out["sig_del"], sig[kind == 1]); _same_sigs(out["sig_ins"], sig[kind != 1])
            assert np.array_equal(out["depth"], od) and (res.depth_sum, res.depth_nonzero) == (os_, onz)
            mean = os_ / onz if onz else 0.0
            min_pts = int(np.ceil(mean * pct)) if pct > 0 else 5
            assert res.min_pts == min_pts
            if min_pts >= 1:
                dels, inss = sig[kind == 1], sig[kind != 1]
                assert np.array_equal(out["label_del"], oracle.dbscan_iv(dels["start"], dels["end"], 0.1, min_pts))
                assert np.array_equal(out["label_ins"], oracle.dbscan_iv(inss["start"], inss["end"], 0.1, min_pts))
    finally:
        sh.free()


def test_reads_that_span_hundreds_of_depth_tiles(ctx, oracle):
    """The scan records, per 16 Ki-position depth tile, the first and last read that reaches it, 64 tiles per wave step: reads with
    megabase reference skips and deletions (more than 64 tiles each), reads running off the contig's end, a read covering the whole
    contig, short reads in between — depth map, sums and the pipeline's calls against the oracle."""
    depth_len = 5_000_000
    rng = np.random.default_rng(5)
    pos, cig = [], []
    for r in range(300):
        p = int(rng.integers(0, depth_len - 2000))
        ops = [(M, int(rng.integers(30, 400)))]
        kind = r % 6
        if kind == 0:
            ops += [(N, int(rng.integers(1_100_000, 3_000_000))), (M, 500), (D, 70), (M, 300)]          # > 64 tiles
        elif kind == 1:
            ops += [(D, int(rng.integers(200_000, 1_500_000))), (M, 200), (I, 60), (M, 100)]
        elif kind == 2:
            p = depth_len - int(rng.integers(1, 3000)); ops += [(D, 90), (M, 5000)]                       # hangs over the end
        else:
            ops += [(I, 55), (M, 800), (D, 51), (M, 900), (S, 120)]
        pos.append(p); cig.append(ops)
    pos.append(0); cig.append([(M, 10), (N, depth_len - 100), (M, 500)])                                  # covers every tile
    order = np.argsort(np.asarray(pos), kind="stable")
    reads = Reads.from_cigar_lists(np.asarray(pos)[order], np.zeros(len(pos), np.uint16), np.full(len(pos), 60, np.uint8), [cig[i] for i in order])
    od, os_, onz = oracle.depth(reads, depth_len)
    d, s, nz = ctx.depth(reads, depth_len)
    assert np.array_equal(d, od) and (s, nz) == (os_, onz)
    sh = ctx.upload(reads, depth_len)
    try:
        for _ in range(2):
            res = sh.pipeline(eps=0.1, min_pts_pct=0.1)
            out = sh.fetch(res, want_depth=True)
            assert np.array_equal(out["depth"], od) and (res.depth_sum, res.depth_nonzero) == (os_, onz)
            sig = oracle.cigar_scan(reads, depth_len)
            kind = sig["qpos_kind"] & 3
            _same_sigs(out["sig_del"], sig[kind == 1]); _same_sigs(out["sig_ins"], sig[kind != 1])
    finally:
        sh.free()


def test_window_and_viterbi_on_extreme_values(ctx, oracle):
    """Depth maps of zeros / 2^32-1, windows past the map, one-position regions, mean coverage 1e-300; observations of +-1e300,
    BAF and population frequencies exactly 0 and 1: the kernels give the oracle's numbers (NaN and inf included)."""
    from contextsv_amd import make_hmm
    from hmm_params import WGS_HMM, WGS_TEST_HMM
    rng = np.random.default_rng(11)
    for it in range(16):
        L = int(rng.choice([1, 2, 50, 1000, 200_000]))
        depth = [rng.poisson(30, L), np.zeros(L, np.int64), rng.integers(0, 2**31, L), np.full(L, 2**32 - 1, np.int64)][it % 4].astype(np.uint32)
        k = 10
        rs = rng.integers(0, L + 50, k).astype(np.uint32)
        re = (rs + rng.choice([0, 1, 5, 19, 20, 21, 1000, 10**6], k)).astype(np.uint32)
        ss = rng.choice([1, 2, 5, 20, 137], k).astype(np.int32)
        mean = float(rng.choice([29.7, 1e-300, 1.0, 1e6]))
        l2, ws, we, off = ctx.window_log2(depth, rs, re, ss, mean)
        for r in range(k):
            o_l2, o_ws, o_we = oracle.window_log2(depth, int(rs[r]), int(re[r]), int(ss[r]), mean)
            a, b = int(off[r]), int(off[r + 1])
            assert np.array_equal(ws[a:b], o_ws) and np.array_equal(we[a:b], o_we)
            np.testing.assert_allclose(l2[a:b], o_l2, rtol=0, atol=1e-6, equal_nan=True)
    for params in (WGS_HMM, WGS_TEST_HMM):
        hmm = make_hmm(**params)
        o1s, o2s, pfs, off = [], [], [], [0]
        for j in range(60):
            T = int(rng.choice([0, 1, 2, 3, 20, 200, 1001]))
            o1 = [rng.normal(0, 0.4, T), rng.choice([-50.0, -9.966, 0.0, 5.0, 50.0, 1e300, -1e300], T), np.zeros(T)][j % 3]
            o1s.append(o1)
            o2s.append(rng.choice([-1.0, 0.0, 1.0, 0.5, 1e-12, 1 - 1e-12, 0.3333], T))
            pfs.append(rng.choice([0.0, 1.0, 0.5, 0.01, 0.99, 1e-9], T))
            off.append(off[-1] + T)
        o1, o2, pf = np.concatenate(o1s), np.concatenate(o2s), np.concatenate(pfs)
        st, ll = ctx.viterbi(hmm, o1, o2, pf, np.asarray(off, np.uint64))
        ost, oll = oracle.viterbi(hmm, o1, o2, pf, np.asarray(off, np.uint64))
        assert np.array_equal(st, ost)
        np.testing.assert_allclose(ll, oll, rtol=0, atol=1e-6, equal_nan=True)


def test_offsets_that_leave_the_word_array_are_refused(ctx):
    """The kernels index the CIGAR words with cigar_off: a table that is not monotone, or that points past the array, must be
    turned away on the host (CSV_EINVAL) — never reach the device."""
    from contextsv_amd import CsvError
    reads = _hostile(11, 50, 3000)
    for breakage in ("swap", "dip", "beyond"):
        off = reads.cigar_off.copy()
        if breakage == "swap":
            k = int(np.argmax(np.diff(off.astype(np.int64)) > 0))
            off[k], off[k + 1] = off[k + 1], off[k]
        elif breakage == "dip":
            off[1:-1] += np.uint64(1 << 20)       # the last entry now lies below its predecessor
        else:
            off[1:] += np.uint64(1 << 20)         # monotone, but past the end of the word array
        bad = Reads(reads.pos, reads.flag, reads.mapq, reads.cigar_off, reads.cigar)
        bad.cigar_off = off                       # bypass the wrapper's own length check: this is the C-ABI's test
        for call in (lambda: ctx.cigar_scan(bad, 3000), lambda: ctx.aln_intervals(bad), lambda: ctx.depth(bad, 3000), lambda: ctx.upload(bad, 3000)):
            with pytest.raises(CsvError):
                call()
