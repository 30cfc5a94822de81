"""GPU parity tests proper: every seam of include/csvgpu.h through the C-ABI against the CPU oracle on the
same seeded inputs. Integer / index results must be bit-identical; the Viterbi log-likelihood and the
window log2 ratios are fp64 and compared at 1e-6 (BASELINE.json north_star), paths exactly."""
import numpy as np
import pytest

import synth_small as ss
from contextsv_amd import Reads, ReadCHMM, DBSCAN, DBSCAN1D, testVit_CHMM, make_hmm
from hmm_params import WGS_HMM, WGS_TEST_HMM

pytestmark = pytest.mark.gpu


def _same_sigs(a, b):
    assert len(a) == len(b)
    for f in ("start", "end", "read", "qpos_kind"):
        assert np.array_equal(a[f], b[f]), f


@pytest.mark.parametrize("seed,kw", [
    (1, {}), (2, dict(n_reads=50, mean_ops=400)), (3, dict(big_frac=0.3)), (4, dict(clip_end=True)),
    (5, dict(n_reads=1, mean_ops=3)), (6, dict(n_reads=700, mean_ops=20, chr_len=50_000)),
    (7, dict(sorted_pos=False)), (8, dict(n_reads=40, mean_ops=3000)),
])
def test_cigar_scan_matches_oracle(ctx, oracle, seed, kw):
    reads, depth_len = ss.random_shard(seed, **kw)
    _same_sigs(ctx.cigar_scan(reads, depth_len), oracle.cigar_scan(reads, depth_len))


def test_cigar_scan_thresholds(ctx, oracle):
    reads, depth_len = ss.random_shard(11, big_frac=0.2)
    for min_oplen, min_mapq in ((50, 20), (1, 0), (30, 61), (100000, 0)):
        _same_sigs(ctx.cigar_scan(reads, depth_len, min_oplen, min_mapq), oracle.cigar_scan(reads, depth_len, min_oplen, min_mapq))


def test_cigar_scan_empty_and_capacity(ctx):
    import contextsv_amd as cs
    empty = Reads(np.zeros(0, np.int32), np.zeros(0, np.uint16), np.zeros(0, np.uint8), np.zeros(1, np.uint64), np.zeros(0, np.uint32))
    assert len(ctx.cigar_scan(empty, 1000)) == 0
    reads, depth_len = ss.random_shard(3, big_frac=0.3)
    with pytest.raises(cs.CsvError) as ei:
        ctx.cigar_scan(reads, depth_len, capacity=2)
    assert ei.value.status == cs._lib.CSV_ECAPACITY


@pytest.mark.parametrize("seed,kw", [(1, {}), (4, dict(clip_end=True)), (7, dict(sorted_pos=False)), (8, dict(n_reads=40, mean_ops=3000))])
def test_aln_intervals_match_oracle(ctx, oracle, seed, kw):
    reads, _ = ss.random_shard(seed, **kw)
    for g, o in zip(ctx.aln_intervals(reads), oracle.aln_intervals(reads)):
        assert np.array_equal(g, o)


@pytest.mark.parametrize("seed,kw", [
    (1, {}), (2, dict(n_reads=50, mean_ops=400)), (4, dict(clip_end=True)), (6, dict(n_reads=700, mean_ops=20, chr_len=50_000)),
    (7, dict(sorted_pos=False)), (9, dict(n_reads=2000, mean_ops=30, chr_len=40_000)),
])
def test_depth_matches_oracle(ctx, oracle, seed, kw):
    reads, depth_len = ss.random_shard(seed, **kw)
    d, s, nz = ctx.depth(reads, depth_len)
    od, os_, onz = oracle.depth(reads, depth_len)
    assert np.array_equal(d, od)
    assert (s, nz) == (os_, onz)
    # truncated contig: bases past depth_len are dropped (cnv_caller.cpp:511-515)
    short = depth_len // 2
    d, s, nz = ctx.depth(reads, short)
    od, os_, onz = oracle.depth(reads, short)
    assert np.array_equal(d, od) and (s, nz) == (os_, onz)
    _, s2, nz2 = ctx.depth(reads, short, want_array=False)
    assert (s2, nz2) == (os_, onz)


@pytest.mark.parametrize("n,eps,min_pts,kw", [
    (0, 0.1, 3, {}), (1, 0.1, 1, {}), (2, 0.1, 2, {}), (300, 0.1, 3, {}), (300, 0.3, 5, dict(sort=True)),
    (2000, 0.1, 2, {}), (2000, 0.0, 2, dict(sort=True)), (1500, 0.5, 6, dict(zero_len=True)),
    (1200, 0.9, 3, dict(clustered=False, span=20000)), (3000, 0.1, 1, dict(sort=True)),
])
def test_dbscan_iv_matches_oracle(ctx, oracle, n, eps, min_pts, kw):
    s, e = ss.random_intervals(100 + n, n, **kw)
    assert np.array_equal(ctx.dbscan_iv(s, e, eps, min_pts), oracle.dbscan_iv(s, e, eps, min_pts))


def test_dbscan_class_mirror(ctx, oracle):
    s, e = ss.random_intervals(5, 400)
    db = DBSCAN(0.1, 3, ctx=ctx)
    db.fit(list(zip(s.tolist(), e.tolist())))
    assert np.array_equal(db.getClusters(), oracle.dbscan_iv(s, e, 0.1, 3))


def test_dbscan_rejects_bad_args(ctx):
    import contextsv_amd as cs
    s, e = ss.random_intervals(5, 10)
    for eps, mp in ((1.0, 3), (-0.1, 3), (0.1, 0)):
        with pytest.raises(cs.CsvError) as ei:
            ctx.dbscan_iv(s, e, eps, mp)
        assert ei.value.status == cs._lib.CSV_EINVAL


def test_dbscan_1d_batched_matches_oracle(ctx, oracle):
    rng = np.random.default_rng(42)
    segs, pts = [0], []
    for k in range(400):
        n = int(rng.choice([0, 1, 3, 5, 8, 30, 64, 65, 200, 511, 512]))
        base = rng.integers(-1000, 1_000_000)
        p = base + rng.choice([0, 0, 0, 300, 5000], n) + rng.integers(-120, 121, n)
        pts.extend(p.tolist()); segs.append(len(pts))
    pts = np.asarray(pts, np.int32); segs = np.asarray(segs, np.uint64)
    for eps, mp in ((100.0, 5), (10.0, 2), (0.0, 1), (99.5, 3)):
        lab = ctx.dbscan_1d(pts, segs, eps, mp)
        for k in range(len(segs) - 1):
            a, b = int(segs[k]), int(segs[k + 1])
            assert np.array_equal(lab[a:b], oracle.dbscan_1d(pts[a:b], eps, mp)), (k, eps, mp)


def test_dbscan_1d_large_segment_and_class(ctx, oracle):
    rng = np.random.default_rng(7)
    big = (rng.integers(0, 50, 3000) * 97 + rng.integers(-60, 61, 3000)).astype(np.int32)
    small = rng.integers(0, 500, 20).astype(np.int32)
    pts = np.concatenate([small, big, small]); segs = np.array([0, 20, 3020, 3040], np.uint64)
    lab = ctx.dbscan_1d(pts, segs, 100.0, 5)
    for k in range(3):
        a, b = int(segs[k]), int(segs[k + 1])
        assert np.array_equal(lab[a:b], oracle.dbscan_1d(pts[a:b], 100.0, 5))
    db = DBSCAN1D(100, 5, ctx=ctx)
    db.fit(small)
    assert np.array_equal(db.getLargestCluster(small), oracle.largest_cluster(small, oracle.dbscan_1d(small, 100.0, 5)))
    db.fit([])
    assert len(db.getLargestCluster([])) == 0


def test_window_log2_matches_oracle(ctx, oracle):
    rng = np.random.default_rng(3)
    depth = rng.poisson(30, 300_000).astype(np.uint32)
    depth[50_000:60_000] = 0
    rs = np.array([1000, 40_000, 52_000, 299_000, 100, 7, 120_000], np.uint32)
    re = np.array([21_000, 140_000, 58_000, 305_000, 110, 7, 120_019], np.uint32)   # one runs past depth_len, one step < 1
    ssz = np.array([20, 137, 20, 20, 20, 20, 20], np.int32)
    l2, ws, we, off = ctx.window_log2(depth, rs, re, ssz, 29.7)
    for r in range(len(rs)):
        o_l2, o_ws, o_we = oracle.window_log2(depth, int(rs[r]), int(re[r]), int(ssz[r]), 29.7)
        a, b = int(off[r]), int(off[r + 1])
        assert np.array_equal(ws[a:b], o_ws) and np.array_equal(we[a:b], o_we)
        np.testing.assert_allclose(l2[a:b], o_l2, rtol=0, atol=1e-6)


def _obs(rng, T, mode):
    o1 = rng.normal(0, 0.4, T)
    if mode == "del": o1 -= 0.8
    if mode == "dup": o1 += 0.45
    o2 = np.where(rng.random(T) < 0.5, -1.0, rng.choice([0.0, 1.0, 0.5, 0.33, 0.25, 0.75], T) + rng.normal(0, 0.03, T) * (rng.random(T) < 0.7))
    o2 = np.where((o2 != -1) & (o2 < 0), 0.0, o2); o2 = np.where(o2 > 1, 1.0, o2)
    pfb = np.where(o2 == -1, 0.5, rng.choice([0.0, 0.5, 0.1, 0.93], T))
    return o1, o2, pfb


@pytest.mark.parametrize("params", [WGS_HMM, WGS_TEST_HMM])
def test_viterbi_matches_oracle(ctx, oracle, params):
    hmm = make_hmm(**params)
    rng = np.random.default_rng(5)
    o1s, o2s, pfbs, off = [], [], [], [0]
    for T, mode in [(1, "n"), (2, "del"), (20, "n"), (20, "del"), (20, "dup"), (200, "dup"), (0, "n"), (1000, "del"), (37, "n")] * 3:
        a, b, c = _obs(rng, T, mode)
        o1s.append(a); o2s.append(b); pfbs.append(c); off.append(off[-1] + T)
    o1, o2, pfb = np.concatenate(o1s), np.concatenate(o2s), np.concatenate(pfbs)
    st, ll = ctx.viterbi(hmm, o1, o2, pfb, np.asarray(off, np.uint64))
    ost, oll = oracle.viterbi(hmm, o1, o2, pfb, np.asarray(off, np.uint64))
    assert np.array_equal(st, ost)                      # identical paths
    np.testing.assert_allclose(ll, oll, rtol=0, atol=1e-6)   # tolerance stated by north_star


def test_testVit_CHMM_mirror(ctx, oracle, tmp_path):
    from hmm_params import write_hmm_file
    path = write_hmm_file(tmp_path / "wgs.hmm", WGS_HMM)
    hmm = ReadCHMM(str(path))
    assert hmm.N == 6
    o1 = [0.0, -0.1, 0.2, -3.0, -2.5, -3.1]; o2 = [-1, 0.5, 0.48, -1, 0.0, 1.0]; pfb = [0.5, 0.4, 0.6, 0.5, 0.3, 0.3]
    st, ll = testVit_CHMM(hmm, 6, o1, o2, pfb, ctx=ctx)
    ost, oll = oracle.viterbi(hmm.c_struct(), o1, o2, pfb, np.array([0, 6], np.uint64))
    assert np.array_equal(st, ost) and abs(ll - oll[0]) < 1e-6


def test_chr_pipeline_matches_oracle(ctx, oracle):
    reads, depth_len = ss.random_shard(21, n_reads=1500, mean_ops=40, big_frac=0.15, chr_len=120_000)
    sh = ctx.upload(reads, depth_len)
    try:
        for eps, pct in ((0.1, 0.1), (0.3, 0.0)):
            res = sh.pipeline(eps=eps, min_pts_pct=pct)
            out = sh.fetch(res, want_depth=True)
            sig = oracle.cigar_scan(reads, depth_len)
            od, os_, onz = oracle.depth(reads, depth_len)
            assert np.array_equal(out["depth"], od) and (res.depth_sum, res.depth_nonzero) == (os_, onz)
            mean = os_ / onz
            min_pts = int(np.ceil(mean * pct)) if pct > 0 else 5
            assert res.min_pts == min_pts and res.mean_cov == mean
            kind = sig["qpos_kind"] & 3
            dels, inss = sig[kind == 1], sig[kind != 1]
            _same_sigs(out["sig_del"], dels); _same_sigs(out["sig_ins"], inss)
            assert np.array_equal(out["label_del"], oracle.dbscan_iv(dels["start"], dels["end"], eps, min_pts))
            assert np.array_equal(out["label_ins"], oracle.dbscan_iv(inss["start"], inss["end"], eps, min_pts))
            for g, o in zip((out["ref_end"], out["q_start"], out["q_end"]), oracle.aln_intervals(reads)):
                assert np.array_equal(g, o)
    finally:
        sh.free()


def _pileup(n_reads, n_distinct_pos):
    """Many reads carrying the same events at the same loci: every signature start is shared by hundreds or thousands of
    records, which is what decides between the ordering pass's bucket path and its radix fallback."""
    M, I, D, S = 0, 1, 2, 4
    pos = np.sort(np.arange(n_reads) % n_distinct_pos * 7 + 1000)
    cig = [[(S, 60 + r % 3), (M, 500 - (p - 1000)), (D, 80 + (r % 2)), (M, 300), (I, 70), (M, 200 + r % 5)] for r, p in enumerate(pos)]
    reads = Reads.from_cigar_lists(pos, np.zeros(n_reads, np.uint16), np.full(n_reads, 60, np.uint8), cig)
    return reads, 20_001


@pytest.mark.parametrize("n_reads,n_pos", [(1500, 1), (1500, 40), (5000, 1), (5000, 3)])
def test_ordering_with_heavy_ties(ctx, oracle, n_reads, n_pos):
    """(1500, *): one bucket holds up to 1500 records (several 64-record chunks per wave); (5000, *): above BK_LOCAL_MAX, the
    LSD radix fallback orders them. Both must give the reference's vector order (end ascending, then reverse insertion)."""
    reads, depth_len = _pileup(n_reads, n_pos)
    _same_sigs(ctx.cigar_scan(reads, depth_len), oracle.cigar_scan(reads, depth_len))
    sh = ctx.upload(reads, depth_len)
    try:
        res = sh.pipeline(eps=0.1, min_pts_pct=0.1)
        out = sh.fetch(res)
        sig = oracle.cigar_scan(reads, depth_len)
        kind = sig["qpos_kind"] & 3
        _same_sigs(out["sig_del"], sig[kind == 1]); _same_sigs(out["sig_ins"], sig[kind != 1])
        assert res.n_sig == 3 * n_reads
    finally:
        sh.free()


def test_pipeline_regrows_signature_buffer(ctx, oracle):
    """More signatures than the shard's initial buffer (max(2^18, 2 x reads)): the pipeline re-runs the scan into a larger one —
    with the depth pass already queued behind the first attempt — and the results are still the oracle's."""
    M, D = 0, 2
    n_reads, per = 4500, 64
    pos = np.sort(np.random.default_rng(3).integers(0, 150_000, n_reads))
    cig = [[op for k in range(per) for op in ((M, 90 + (r + k) % 7), (D, 50 + (r * 7 + k) % 40))] + [(M, 50)] for r in range(n_reads)]
    reads = Reads.from_cigar_lists(pos, np.zeros(n_reads, np.uint16), np.full(n_reads, 60, np.uint8), cig)
    depth_len = 170_000
    sh = ctx.upload(reads, depth_len)
    try:
        res = sh.pipeline(eps=0.1, min_pts_pct=0.1)
        assert res.n_sig == n_reads * per > (1 << 18)
        out = sh.fetch(res, want_depth=True)
        sig = oracle.cigar_scan(reads, depth_len)
        _same_sigs(out["sig_del"], sig)
        od, os_, onz = oracle.depth(reads, depth_len)
        assert np.array_equal(out["depth"], od) and (res.depth_sum, res.depth_nonzero) == (os_, onz)
        res2 = sh.pipeline(eps=0.1, min_pts_pct=0.1)             # second run: buffer already large enough
        assert (res2.n_sig, res2.depth_sum, res2.min_pts) == (res.n_sig, res.depth_sum, res.min_pts)
    finally:
        sh.free()
