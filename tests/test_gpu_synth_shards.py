"""Shards from the benchmark's own generator (ONT with a 5 % indel rate and soft clips, HiFi with 40-op CIGARs — several
reads per 1 KiB chunk, the boundary-heavy case of the scan's chunk ring), small enough for the oracle: signatures, alignment intervals,
depth map, sums, min_pts and both label sets of the device pipeline against it."""
import numpy as np
import pytest

from contextsv_amd import host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tech,chr_len,depth", [(0, 3_000_000, 20.0), (1, 3_000_000, 30.0), (1, 40_000, 60.0)])
def test_generated_shards_against_the_oracle(ctx, oracle, tech, chr_len, depth):
    syn = host.SynthShard(seed=0x5EED0000 + 77 + tech, chr_len=chr_len, depth=depth, tech=tech, threads=4)
    reads, depth_len = syn.reads, syn.depth_len
    sh = ctx.upload(reads, depth_len)
    try:
        sig = oracle.cigar_scan(reads, depth_len)
        od, osum, onz = oracle.depth(reads, depth_len)
        for g, o in zip(ctx.aln_intervals(reads), oracle.aln_intervals(reads)):
            assert np.array_equal(g, o)
        for _ in range(2):
            res = sh.pipeline(eps=0.1, min_pts_pct=0.1)
            out = sh.fetch(res, want_depth=True)
            kind = sig["qpos_kind"] & 3
            for got, exp in ((out["sig_del"], sig[kind == 1]), (out["sig_ins"], sig[kind != 1])):
                assert len(got) == len(exp)
                for f in ("start", "end", "read", "qpos_kind"):
                    assert np.array_equal(got[f], exp[f]), f
            assert np.array_equal(out["depth"], od) and (res.depth_sum, res.depth_nonzero) == (osum, onz)
            min_pts = int(np.ceil(osum / onz * 0.1)) if onz else 0
            assert res.min_pts == min_pts
            if min_pts >= 1:
                dels, inss = sig[kind == 1], sig[kind != 1]
                assert np.array_equal(out["label_del"], oracle.dbscan_iv(dels["start"], dels["end"], 0.1, min_pts))
                assert np.array_equal(out["label_ins"], oracle.dbscan_iv(inss["start"], inss["end"], 0.1, min_pts))
    finally:
        sh.free()
        syn.free()
