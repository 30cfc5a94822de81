"""Copy-number pass (§8a rows a10, a12, a13 together): CNVCaller host mirror over the window + Viterbi kernels on a
resident depth map, against the oracle restatement on the same depth array. Observation ORDER (libstdc++ unordered_map
iteration), state votes, SVCall update rules and the split-read duplicate rule must match exactly; log2 ratios and
log-likelihoods within 1e-6."""
import numpy as np
import pytest

from contextsv_amd import Reads, host, make_hmm
from hmm_params import WGS_HMM

pytestmark = pytest.mark.gpu
DEL, DUP, INV, INS, BND, UNKNOWN = 0, 1, 2, 3, 4, -1
CHR_LEN = 400_000


@pytest.fixture(scope="module")
def cnv_setup(ctx):
    rng = np.random.default_rng(12)
    # ~30x of 5 kb reads, with a 15x stretch (heterozygous loss), a 60x stretch (gain) and a zero-coverage hole
    dens = np.full(CHR_LEN, 30.0)
    dens[60_000:110_000] = 15.0
    dens[200_000:260_000] = 60.0
    dens[300_000:306_000] = 0.0
    starts = []
    for p in range(0, CHR_LEN - 5000, 50):
        starts += [p + int(rng.integers(0, 50))] * int(rng.poisson(dens[p] * 50 / 5000))
    starts = np.sort(np.asarray(starts))
    reads = Reads.from_cigar_lists(starts, np.zeros(len(starts), int), np.full(len(starts), 60), [[(0, 5000)]] * len(starts))
    sh = ctx.upload(reads, CHR_LEN + 1)
    res = sh.pipeline(eps=0.1, min_pts_pct=0.1)
    depth = sh.fetch(res, want_depth=True)["depth"]
    n_snp = 380
    pos = np.sort(rng.choice(np.arange(1000, CHR_LEN - 1000), n_snp, replace=False)).astype(np.uint32)
    baf = np.where(rng.random(n_snp) < 0.6, 0.5 + rng.normal(0, 0.05, n_snp), rng.choice([0.0, 1.0, 0.33, 0.67], n_snp))
    baf = np.clip(baf, 0.0, 1.0)
    snps = {"pos": pos, "baf": baf, "pfb": rng.uniform(0.02, 0.98, n_snp), "has_pfb": (rng.random(n_snp) < 0.3).astype(np.uint8)}
    yield sh, res, depth, snps
    sh.free()


def _same_obs(a, b):
    assert np.array_equal(a["pos"], b["pos"]) and np.array_equal(a["is_snp"], b["is_snp"])      # identical order
    assert np.array_equal(a["baf"], b["baf"]) and np.array_equal(a["pfb"], b["pfb"])
    np.testing.assert_allclose(a["log2_cov"], b["log2_cov"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("start,end,ss", [(70_000, 100_000, 20), (1, 399_999, 20), (205_000, 215_000, 5), (301_000, 305_000, 20),
                                           (150_000, 150_009, 20), (50_000, 52_500, 64), (399_000, 400_600, 20)])
def test_query_snp_region_matches_oracle(ctx, oracle, cnv_setup, start, end, ss):
    sh, res, depth, snps = cnv_setup
    _same_obs(host.query_snp_region(ctx, sh, start, end, res.mean_cov, ss, snps), oracle.query_snp_region(depth, start, end, res.mean_cov, ss, snps))


def _calls(rng, n, types):
    s = rng.integers(1000, CHR_LEN - 60_000, n).astype(np.uint32)
    ln = rng.choice([300, 1500, 2500, 8000, 30_000, 55_000], n)
    c = host.make_calls(s, (s + ln).astype(np.uint32), rng.choice(types, n), rng.integers(2, 30, n))
    c["aln_flags"] = rng.choice([1, 2, 4, 8, 16], n)
    return c


def _same_calls(a, b):
    assert len(a) == len(b)
    for f in ("start", "end", "sv_type", "cluster_size", "id", "aln_flags", "genotype", "cn_state", "aln_offset"):
        assert np.array_equal(a[f], b[f]), f
    np.testing.assert_allclose(a["hmm_likelihood"], b["hmm_likelihood"], rtol=0, atol=1e-6)


def test_cigar_cn_prediction_matches_oracle(ctx, oracle, cnv_setup):
    sh, res, depth, snps = cnv_setup
    hmm = make_hmm(**WGS_HMM)
    calls = _calls(np.random.default_rng(3), 120, [DEL, INS, DUP, INV])
    # fixed candidates over the engineered coverage stretches so that every branch of the update rule is taken
    calls[:4]["start"] = [62_000, 205_000, 62_000, 120_000]; calls[:4]["end"] = [105_000, 255_000, 105_000, 160_000]
    calls[:4]["sv_type"] = [DEL, INS, INS, DEL]
    got = host.cn_prediction(ctx, sh, calls, hmm, res.mean_cov, snps, split=False)
    exp = oracle.cn_prediction(depth, calls, hmm, res.mean_cov, snps, split=False)
    _same_calls(got, exp)
    assert (got["cn_state"] != 0).any() and got[0]["sv_type"] == DEL and got[0]["cn_state"] in (1, 2)      # the 15x stretch is called a loss
    assert got[1]["sv_type"] == DUP and got[1]["cn_state"] in (5, 6)                                    # INS over the 60x stretch becomes DUP
    assert got[2]["sv_type"] == INS and got[2]["cn_state"] == 0                                         # INS cannot become DEL (isValidCopyNumberUpdate)
    short = (calls["end"] - calls["start"]) < 2000
    assert (got[short]["cn_state"] == 0).all()                                                           # below --min-cnv: untouched


def test_split_cn_prediction_matches_oracle(ctx, oracle, cnv_setup):
    sh, res, depth, snps = cnv_setup
    hmm = make_hmm(**WGS_HMM)
    calls = _calls(np.random.default_rng(4), 80, [UNKNOWN, INV, INS, DEL, DUP])
    calls[:5]["start"] = [62_000, 205_000, 63_000, 206_000, 64_000]; calls[:5]["end"] = [105_000, 255_000, 104_000, 254_000, 103_000]
    calls[:5]["sv_type"] = [UNKNOWN, UNKNOWN, INV, DEL, DUP]       # -> DEL, DUP, INV+HMM, extra DUP call, extra DEL call
    order = np.lexsort((calls["end"], calls["start"]))             # the split list is kept sorted by addSVCall
    calls = calls[order]
    got = host.cn_prediction(ctx, sh, calls, hmm, res.mean_cov, snps, split=True)
    exp = oracle.cn_prediction(depth, calls, hmm, res.mean_cov, snps, split=True)
    _same_calls(got, exp)
    assert len(got) > len(calls)                                    # conflicting predictions were added as new calls
