"""The forms of the scan and of the depth walk (a wave per read; groups of 16 or 8 lanes per read; a lane per read: scan.hip, depth.hip) on the same
shards, each against the oracle: signatures, alignment intervals, depth map, sums. The form is normally chosen per shard from the mean
CIGAR words per read; CSV_SCAN_FORM forces it (read when a shard is created / a host-pointer entry point is called). Both instances of
every kernel are covered: host-pointer entry points stage an UNPADDED copy (bounds-checked loads), resident shards are padded."""
import os

import numpy as np
import pytest

import synth_small as ss
from contextsv_amd import Reads, host

pytestmark = pytest.mark.gpu

M, I, D, N, S, H, P, EQ, X = range(9)


@pytest.fixture(params=[0, 1, 2, 3], ids=["wave", "rows16", "rows8", "lanes"])
def form(request):
    old = os.environ.get("CSV_SCAN_FORM")
    os.environ["CSV_SCAN_FORM"] = str(request.param)
    yield request.param
    if old is None:
        del os.environ["CSV_SCAN_FORM"]
    else:
        os.environ["CSV_SCAN_FORM"] = old


def _check(ctx, oracle, reads, depth_len, min_oplen=50, min_mapq=20):
    sig = oracle.cigar_scan(reads, depth_len, min_oplen, min_mapq)
    got = ctx.cigar_scan(reads, depth_len, min_oplen, min_mapq)
    assert len(got) == len(sig)
    for f in ("start", "end", "read", "qpos_kind"):
        assert np.array_equal(got[f], sig[f]), f
    for g, o in zip(ctx.aln_intervals(reads), oracle.aln_intervals(reads)):
        assert np.array_equal(g, o)
    od, osum, onz = oracle.depth(reads, depth_len)
    d, s, nz = ctx.depth(reads, depth_len)
    assert np.array_equal(d, od) and (s, nz) == (osum, onz)
    # resident (padded) shard through the per-chromosome pipeline
    sh = ctx.upload(reads, depth_len)
    try:
        res = sh.pipeline(eps=0.1, min_pts_pct=0.1, min_oplen=min_oplen, min_mapq=min_mapq)
        out = sh.fetch(res, want_depth=True)
        kind = sig["qpos_kind"] & 3
        for g, e in ((out["sig_del"], sig[kind == 1]), (out["sig_ins"], sig[kind != 1])):
            assert len(g) == len(e)
            for f in ("start", "end", "read", "qpos_kind"):
                assert np.array_equal(g[f], e[f]), f
        assert np.array_equal(out["depth"], od) and (res.depth_sum, res.depth_nonzero) == (osum, onz)
        for g, o in zip((out["ref_end"], out["q_start"], out["q_end"]), oracle.aln_intervals(reads)):
            assert np.array_equal(g, o)
    finally:
        sh.free()


@pytest.mark.parametrize("seed,kw", [
    (1, {}), (2, dict(n_reads=50, mean_ops=400)), (3, dict(big_frac=0.3)), (4, dict(clip_end=True)),
    (5, dict(n_reads=1, mean_ops=3)), (6, dict(n_reads=700, mean_ops=20, chr_len=50_000)),
    (7, dict(sorted_pos=False)), (8, dict(n_reads=40, mean_ops=3000)), (9, dict(n_reads=2000, mean_ops=30, chr_len=40_000)),
    (10, dict(n_reads=900, mean_ops=8, chr_len=30_000, big_frac=0.2)),
])
def test_forms_on_random_shards(ctx, oracle, form, seed, kw):
    reads, depth_len = ss.random_shard(seed, **kw)
    _check(ctx, oracle, reads, depth_len)


def test_forms_window_edges(ctx, oracle, form):
    """Reads whose word counts and first-word offsets sit on every edge of the 64-word windows, the 4-word load alignment and the 64-word
    checkpoint grid: 0, 1, 3, 4, 5, 59..69, 127..131 words; big ops in the first and last word of a window; reads without any query op."""
    rng = np.random.default_rng(99)
    pos, flag, mapq, cig = [], [], [], []
    p = 100
    counts = [0, 1, 2, 3, 4, 5, 7, 8, 9, 59, 60, 61, 62, 63, 64, 65, 66, 67, 68, 69, 127, 128, 129, 130, 131, 191, 192, 193, 255, 256, 257, 300]
    for rep in range(6):
        for n in counts:
            ops = []
            for k in range(n):
                if k % 2 == 0:
                    ops.append((M, int(rng.integers(1, 30))))
                else:
                    big = rng.random() < 0.15
                    op = int(rng.choice([I, D, D, N, S if k in (1, n - 1) else I]))
                    ops.append((op, int(rng.integers(50, 400)) if big else int(rng.integers(1, 40))))
            if n >= 3 and rep == 1:
                ops[0] = (S, 77); ops[-1] = (S, 66)                # clips at both ends (first / last word of the read)
            if n >= 2 and rep == 2:
                ops[0] = (H, 10); ops[1] = (D, 60)                  # a read that opens with a deletion: query_start found late
            if rep == 3 and n in (3, 64):
                ops = [(D, 70)] * n                                 # no query op at all: q_start stays 0
            pos.append(p); flag.append(0 if rep != 4 else 0x10); mapq.append(60); cig.append(ops)
            p += int(rng.integers(0, 40))
        # filler reads shift the next repetition's first-word offsets through all residues mod 4 and mod 64
        for f in range(rep + 1):
            pos.append(p); flag.append(0); mapq.append(60); cig.append([(M, 20)] * (1 + 2 * f)); p += 3
    reads = Reads.from_cigar_lists(pos, flag, mapq, cig)
    _check(ctx, oracle, reads, 60_000)
    _check(ctx, oracle, reads, 2_000)          # most reads beyond the contig: clips skipped by the reference's `continue`, depth clipped


@pytest.mark.parametrize("tech,chr_len,depth", [(1, 1_500_000, 60.0), (0, 1_000_000, 20.0)])
def test_forms_on_generated_shards(ctx, oracle, form, tech, chr_len, depth):
    syn = host.SynthShard(seed=0x5EED0000 + 177 + tech, chr_len=chr_len, depth=depth, tech=tech, threads=4)
    try:
        _check(ctx, oracle, syn.reads, syn.depth_len)
    finally:
        syn.free()


def test_form_is_chosen_from_the_read_length(ctx, oracle):
    """Default choice (no override): a HiFi-shaped shard and an ONT-shaped shard both equal the oracle (whatever form each took)."""
    os.environ.pop("CSV_SCAN_FORM", None)
    for tech, depth in ((1, 40.0), (0, 15.0)):
        syn = host.SynthShard(seed=0x5EED0000 + 277 + tech, chr_len=800_000, depth=depth, tech=tech, threads=4)
        try:
            _check(ctx, oracle, syn.reads, syn.depth_len)
        finally:
            syn.free()
