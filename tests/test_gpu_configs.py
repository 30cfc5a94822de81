"""The remaining BASELINE.json configurations at their per-GPU size:
  configs[2]  chr1, 30x synthetic ONT (6.3e5 reads, 7.1e8 CIGAR ops = 2.8 GB, 3.4e5 signatures, 15 k depth tiles) + the copy-number
              pass over its >= 2 kb calls: scan / intervals / depth / sum / non-zero / min_pts bit-exact against the oracle, DEL labels
              exact against the oracle's O(n^2) DBSCAN, INS labels by the seam + alphabet properties, copy-number predictions against
              oracle/cnv_oracle.cpp;
  configs[4]  a chr22-sized contig of 60x synthetic HiFi (the per-GPU unit of the whole-genome HiFi run: many short CIGARs, several
              reads per 1 KiB chunk): everything exact against the oracle, including the merged call set."""
import numpy as np
import pytest

import oracle_lib
from contextsv_amd import host, make_hmm
from hmm_params import WGS_HMM

pytestmark = pytest.mark.gpu
CHR1, CHR22 = 248956422, 50818468


def _same_sigs(a, b):
    assert a.tobytes() == b.tobytes()


def _oracle_merge(oracle, dels, inss, lab_del, lab_ins, min_pts):
    sig = np.concatenate([dels, inss])
    oc = np.zeros(len(sig), oracle_lib.CALL_DTYPE)
    oc["start"], oc["end"], oc["sv_type"], oc["id"] = sig["start"], sig["end"], np.where((sig["qpos_kind"] & 3) == 1, 0, 3), np.arange(len(sig))
    om = oracle.merge_svs(oc, 0.1, min_pts, False, label_fn=lambda s, e, eps, mp: lab_del if len(s) == len(dels) else lab_ins)
    return sig, om


def test_chr1_size_shard_with_copy_number_pass(ctx, oracle):
    host.set_context(ctx)
    syn = host.SynthShard(0x5EED0000 + 3000 + 1, CHR1, 30.0, 0, 16)
    reads = syn.reads
    assert reads.n_reads > 600_000 and reads.n_cigar > 700_000_000
    sh = ctx.upload(reads, syn.depth_len)
    try:
        res = sh.pipeline(eps=0.1, min_pts_pct=0.1)
        assert res.n_sig > (1 << 18)                                      # outgrows the shard's initial signature buffer
        out = sh.fetch(res, want_depth=True)
        sig = oracle.cigar_scan(reads, syn.depth_len)
        kind = sig["qpos_kind"] & 3
        _same_sigs(out["sig_del"], sig[kind == 1])
        _same_sigs(out["sig_ins"], sig[kind != 1])
        for g, o in zip((out["ref_end"], out["q_start"], out["q_end"]), oracle.aln_intervals(reads)):
            assert np.array_equal(g, o)
        d, s, nz = oracle.depth(reads, syn.depth_len)
        assert np.array_equal(out["depth"], d) and (res.depth_sum, res.depth_nonzero) == (s, nz)
        assert res.mean_cov == s / nz and res.min_pts == int(np.ceil(s / nz * 0.1))
        dels, inss = out["sig_del"], out["sig_ins"]
        assert np.array_equal(out["label_del"], oracle.dbscan_iv(dels["start"], dels["end"], 0.1, res.min_pts))
        # the INS set (3e5 signatures: the O(n^2) walk would take an hour) against the windowed CPU labeller, itself pinned against the
        # reference's own dbscan.cpp (oracle/dbscan_window.cpp, tests/test_oracle_windowed.py); the DEL set meets both CPU forms
        lab = out["label_ins"]
        assert len(lab) > 150_000
        assert np.array_equal(lab, oracle.dbscan_iv_windowed(inss["start"], inss["end"], 0.1, res.min_pts))
        assert np.array_equal(out["label_del"], oracle.dbscan_iv_windowed(dels["start"], dels["end"], 0.1, res.min_pts))
        assert np.array_equal(ctx.dbscan_iv(inss["start"], inss["end"], 0.1, res.min_pts), lab)       # the seam on caller-order input, same labels
        # merged calls, then the copy-number pass over the calls of >= 2 kb against the oracle's (same depth map, same SNPs)
        calls, tags, st = host.process_resident_chromosome(ctx, sh, 0.1, 0.1)
        _, om = _oracle_merge(oracle, dels, inss, out["label_del"], lab, res.min_pts)
        assert len(calls) == len(om) > 1500
        for f in ("start", "end", "sv_type", "cluster_size"):
            assert np.array_equal(calls[f], om[f]), f
        rng = np.random.default_rng(1)
        n_snp = CHR1 // 1000
        pos = np.sort(rng.choice(np.arange(1000, CHR1 - 1000, dtype=np.int64), n_snp, replace=False)).astype(np.uint32)
        snps = {"pos": pos, "baf": np.where(rng.random(n_snp) < 0.66, 0.45 + 0.1 * rng.random(n_snp), 1.0), "pfb": np.zeros(n_snp), "has_pfb": np.zeros(n_snp, np.uint8)}
        hmm = make_hmm(**WGS_HMM)
        full = host.make_calls(calls["start"], calls["end"], calls["sv_type"], calls["cluster_size"])
        full["aln_flags"] = calls["aln_flags"]
        got = host.cn_prediction(ctx, sh, full, hmm, res.mean_cov, snps, split=False)
        exp = oracle.cn_prediction(d, full, hmm, res.mean_cov, snps, split=False)
        assert int(((full["end"] - full["start"]) >= 2000).sum()) > 300
        for f in ("start", "end", "sv_type", "cluster_size", "aln_flags", "genotype", "cn_state"):
            assert np.array_equal(got[f], exp[f]), f
        np.testing.assert_allclose(got["hmm_likelihood"], exp["hmm_likelihood"], rtol=0, atol=1e-6)
        assert (got["cn_state"] != 0).sum() > 100
    finally:
        sh.free()
        syn.free()


def test_hifi_60x_chr22_size_shard(ctx, oracle):
    host.set_context(ctx)
    syn = host.SynthShard(0x5EED0000 + 4000 + 22, CHR22, 60.0, 1, 16)
    reads = syn.reads
    assert reads.n_reads > 150_000 and reads.n_cigar / reads.n_reads < 80         # short CIGARs: several reads per 256-word chunk
    sh = ctx.upload(reads, syn.depth_len)
    try:
        res = sh.pipeline(eps=0.1, min_pts_pct=0.1)
        out = sh.fetch(res, want_depth=True)
        sig = oracle.cigar_scan(reads, syn.depth_len)
        kind = sig["qpos_kind"] & 3
        _same_sigs(out["sig_del"], sig[kind == 1])
        _same_sigs(out["sig_ins"], sig[kind != 1])
        for g, o in zip((out["ref_end"], out["q_start"], out["q_end"]), oracle.aln_intervals(reads)):
            assert np.array_equal(g, o)
        d, s, nz = oracle.depth(reads, syn.depth_len)
        assert np.array_equal(out["depth"], d) and (res.depth_sum, res.depth_nonzero) == (s, nz)
        assert res.min_pts == int(np.ceil(s / nz * 0.1)) == 6
        dels, inss = out["sig_del"], out["sig_ins"]
        assert np.array_equal(out["label_del"], oracle.dbscan_iv(dels["start"], dels["end"], 0.1, res.min_pts))
        assert np.array_equal(out["label_ins"], oracle.dbscan_iv(inss["start"], inss["end"], 0.1, res.min_pts))
        calls, tags, st = host.process_resident_chromosome(ctx, sh, 0.1, 0.1)
        _, om = _oracle_merge(oracle, dels, inss, out["label_del"], out["label_ins"], res.min_pts)
        assert len(calls) == len(om) > 100
        for f in ("start", "end", "sv_type", "cluster_size"):
            assert np.array_equal(calls[f], om[f]), f
    finally:
        sh.free()
        syn.free()
