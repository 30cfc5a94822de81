"""SVCaller::runResident — one step of the whole-genome benchmark: every contig resident in HBM, the CIGAR pass of all contigs
through the lanes, then the passes of SVCaller::run in the reference's order (src/sv_caller.cpp:804-945) — against (i) the
upload-per-contig run() that test_gpu_e2e.py checks field for field against the oracle, and (ii) the oracle chain itself on the
benchmark's own generator (shards with supplementary records, SNPs, calls >= 2 kb for the copy-number pass)."""
import numpy as np
import pytest

import contextsv_amd as cs
from contextsv_amd import host, make_hmm
from hmm_params import WGS_HMM
from oracle_chain import oracle_run
from test_gpu_e2e import _build

pytestmark = pytest.mark.gpu
FIELDS = ("start", "end", "sv_type", "cluster_size", "aln_flags", "genotype", "cn_state", "aln_offset")


def _same(a, b, exact_lh=True):
    assert len(a) == len(b)
    for f in FIELDS:
        assert np.array_equal(a[f], b[f]), f
    if exact_lh:
        assert np.array_equal(a["hmm_likelihood"], b["hmm_likelihood"])
    else:
        np.testing.assert_allclose(a["hmm_likelihood"], b["hmm_likelihood"], rtol=0, atol=1e-6)


def test_resident_run_equals_uploading_run(ctx):
    contigs = _build(11)
    hmm = make_hmm(**WGS_HMM)
    host.set_context(ctx)
    ref_calls, ref_tid = host.run(ctx, contigs, hmm, eps=0.1, min_pts_pct=0.1)
    g = host.Genome()
    for t, c in enumerate(contigs):
        g.add(ctx, "contig%d" % t, t, c["reads"], c["depth_len"], c["qname_id"], c["snps"], name_style=0)
    got, tid, st, per = g.run(ctx, hmm)
    assert np.array_equal(tid, ref_tid)
    _same(got, ref_calls)
    assert st.n_reads == sum(c["reads"].n_reads for c in contigs) and st.n_final_calls == len(got) and st.n_cigar_cn_regions > 0 and st.n_split_calls > 0
    # again (nothing was consumed), with one host thread and the qname map's order replayed on the host instead of the device,
    # and through three lanes behind a gate
    got2, tid2, _, _ = g.run(ctx, hmm, host_threads=1, host_split_order=True)
    assert np.array_equal(tid2, ref_tid)
    _same(got2, ref_calls)
    lanes = [cs.Context(0) for _ in range(3)]
    gate = cs.Gate()
    try:
        for c in lanes:
            c.set_gate(gate)
        got3, tid3, st3, per3 = g.run(ctx, hmm, lanes=lanes)
        assert np.array_equal(tid3, ref_tid)
        _same(got3, ref_calls)
        assert [p.n_signatures for p in per3] == [p.n_signatures for p in per]
        # With lanes, the CIGAR copy-number predictions of the contigs that are merged when the split pass's first half is done are made
        # during the CIGAR pass, in place (SVCaller::runResident). Which contigs those are depends on timing: none of them, all of them
        # (the task waits for every merge), and whatever it happens to be must give the same calls.
        import os
        for env in ({"CSV_NO_EARLY_CN": "1"}, {"CSV_EARLY_CN_WAIT_ALL": "1"}):
            os.environ.update(env)
            try:
                got4, tid4, st4, _ = g.run(ctx, hmm, lanes=lanes)
            finally:
                for k in env:
                    del os.environ[k]
            assert np.array_equal(tid4, ref_tid) and st4.n_cigar_cn_regions == st3.n_cigar_cn_regions
            _same(got4, ref_calls)
    finally:
        for c in lanes:
            c.set_gate(None)
            c.close()
        gate.close()
    g.free()


@pytest.mark.parametrize("tech,depth", [(0, 30.0), (1, 60.0)])
def test_resident_run_on_generated_shards_equals_oracle_chain(ctx, oracle, tech, depth):
    """Three contigs from the benchmark's generator (ONT 30x / HiFi 60x, with primary + supplementary pairs and generated SNPs),
    staged and run resident, against the oracle's pieces composed in the reference's order."""
    hmm = make_hmm(**WGS_HMM)
    host.set_context(ctx)
    lens = [1_500_000, 900_000, 1_200_000]
    g = host.Genome()
    contigs = []
    for t, L in enumerate(lens):
        syn = host.SynthShard(0xC0FFEE + 17 * t + tech, L, depth, tech, 4, sv_per_bp=1.0 / 30000.0)
        r = syn.reads
        reads = cs.Reads(r.pos.copy(), r.flag.copy(), r.mapq.copy(), r.cigar_off.copy(), r.cigar.copy())
        qid = syn.qname_id.astype(np.uint32) + np.uint32(t << 24)                   # "r<id>" names must not collide across contigs
        rng = np.random.default_rng(t)
        n_snp = L // 1000
        pos = np.sort(rng.choice(np.arange(1000, L - 1000), n_snp, replace=False)).astype(np.uint32)
        snps = {"pos": pos, "baf": np.where(rng.random(n_snp) < 0.66, 0.45 + 0.1 * rng.random(n_snp), 1.0), "pfb": np.zeros(n_snp), "has_pfb": np.zeros(n_snp, np.uint8)}
        contigs.append({"reads": reads, "depth_len": syn.depth_len, "qname_id": qid, "snps": snps})
        g.add(ctx, "contig%d" % t, t, reads, syn.depth_len, qid, snps, name_style=0)
        syn.free()
    got, tid, st, per = g.run(ctx, hmm)
    exp, exp_tid, depths, means = oracle_run(oracle, contigs, hmm)
    assert np.array_equal(tid, exp_tid)
    _same(got, exp, exact_lh=False)
    assert len(got) > 10 and st.n_cigar_cn_regions > 0
    if tech == 0:
        assert st.n_split_calls > 0
    g.free()


def _many_small(ctx, n_contigs=14, tech=0, depth=20.0):
    g = host.Genome()
    for t in range(n_contigs):
        L = 300_000 + 37_000 * (t % 5)
        syn = host.SynthShard(0xBEEF00 + 31 * t + tech, L, depth, tech, 2, sv_per_bp=1.0 / 20000.0)
        r = syn.reads
        reads = cs.Reads(r.pos.copy(), r.flag.copy(), r.mapq.copy(), r.cigar_off.copy(), r.cigar.copy())
        qid = syn.qname_id.astype(np.uint32) + np.uint32(t << 24)
        rng = np.random.default_rng(100 + t)
        n_snp = L // 1000
        pos = np.sort(rng.choice(np.arange(1000, L - 1000), n_snp, replace=False)).astype(np.uint32)
        snps = {"pos": pos, "baf": np.where(rng.random(n_snp) < 0.66, 0.45 + 0.1 * rng.random(n_snp), 1.0), "pfb": np.zeros(n_snp), "has_pfb": np.zeros(n_snp, np.uint8)}
        g.add(ctx, "contig%d" % t, t, reads, syn.depth_len, qid, snps, name_style=0)
        syn.free()
    return g


def test_early_batches_inside_the_pass_give_the_same_calls(ctx):
    """The production path of the 24-contig step: with eight or more contigs still to come, the copy-number predictions, the split-read
    chain and the merges of the contigs merged so far are made in batches INSIDE the CIGAR pass (SVCaller::runResident). Fourteen small
    contigs through three lanes: the default run (batch sizes depend on timing), batches of three down to the last contig
    (CSV_EARLY_SMALL_BATCHES), one batch of everything (CSV_EARLY_CN_WAIT_ALL), no early batch at all (CSV_NO_EARLY_CN), no split overlap,
    and the run without lanes must give the same records."""
    import os
    hmm = make_hmm(**WGS_HMM)
    host.set_context(ctx)
    g = _many_small(ctx)
    lanes = [cs.Context(0) for _ in range(3)]
    gate = cs.Gate()
    try:
        for c in lanes:
            c.set_gate(gate)
        ref, ref_tid, st0, _ = g.run(ctx, hmm)                                     # no lanes: nothing early
        assert len(ref) > 20 and st0.n_cigar_cn_regions > 0 and st0.n_split_calls > 0
        for env, kw in (({}, {}), ({"CSV_EARLY_SMALL_BATCHES": "1"}, {}), ({"CSV_EARLY_CN_WAIT_ALL": "1"}, {}), ({"CSV_NO_EARLY_CN": "1"}, {}),
                        ({"CSV_NO_EARLY_SPLIT": "1", "CSV_EARLY_SMALL_BATCHES": "1"}, {}), ({}, {"overlap_split": False}),
                        # the split order in two calls with the caller's supplementary hashes instead of queued whole (csvgpu_split_order_begin_self);
                        # the three-launch radix passes instead of the onesweep ones; a late split-read first half joined behind the pass
                        ({"CSV_SPLIT_NO_SELF": "1"}, {}), ({"CSV_SORT_ONESWEEP": "0"}, {}), ({"CSV_NO_LATE_JOIN": "1", "CSV_SPLIT_NO_SELF": "1"}, {}),
                        # the split-read first half made to outlast the CIGAR pass: joined in front of the split chain / behind the pass
                        ({"CSV_TEST_PREPARE_DELAY_MS": "40"}, {}), ({"CSV_TEST_PREPARE_DELAY_MS": "40", "CSV_NO_LATE_JOIN": "1"}, {}),
                        # without / with the split chain of all contigs beside the pass (what a run that takes no early batch does by default)
                        ({"CSV_NO_SPLIT_BESIDE_PASS": "1"}, {}), ({"CSV_NO_SPLIT_BESIDE_PASS": "1", "CSV_TEST_PREPARE_DELAY_MS": "40"}, {}),
                        ({"CSV_NO_EARLY_CN": "1", "CSV_NO_SPLIT_BESIDE_PASS": "1"}, {})):
            os.environ.update(env)
            try:
                for _ in range(2):
                    got, tid, st, _ = g.run(ctx, hmm, lanes=lanes, **kw)
                    assert np.array_equal(tid, ref_tid), env
                    _same(got, ref)
                    assert st.n_cigar_cn_regions == st0.n_cigar_cn_regions and st.n_split_calls == st0.n_split_calls
            finally:
                for k in env:
                    del os.environ[k]
    finally:
        for c in lanes:
            c.set_gate(None)
            c.close()
        gate.close()
        g.free()


def test_the_callers_context_cannot_be_a_lane(ctx):
    """runResident works on the caller's context from another thread while the lanes run the CIGAR pass: a context handed in as a lane too
    (or twice) would be driven by two threads — refused."""
    hmm = make_hmm(**WGS_HMM)
    host.set_context(ctx)
    g = _many_small(ctx, n_contigs=2)
    other = cs.Context(0)
    try:
        with pytest.raises(Exception):
            g.run(ctx, hmm, lanes=[ctx, other])
        with pytest.raises(Exception):
            g.run(ctx, hmm, lanes=[other, other])
        got, tid, _, _ = g.run(ctx, hmm, lanes=[other])                            # (still usable afterwards)
        assert len(tid) == len(got)
    finally:
        other.close()
        g.free()


def test_three_largest_contigs_at_full_size_batched_equals_per_contig(ctx):
    """The genome-wide batching of the benchmark's step at SIZE: chr1 + chr2 + chr3 of the 30x ONT genome (1.6e6 reads, 7.5 GB of CIGAR
    words) resident together and run through three lanes — one Viterbi / DBSCAN1D / small-set DBSCAN launch over all contigs, the split
    order of all contigs in one chain, early batches under timing — must give, contig by contig, exactly the records of each contig run
    ALONE on one context (the per-contig path whose scan / depth / labels / merged calls meet the oracle at this size in
    tests/test_gpu_configs.py)."""
    import hashlib
    GRCH38 = [248956422, 242193529, 198295559]
    hmm = make_hmm(**WGS_HMM)
    host.set_context(ctx)
    lanes = [cs.Context(0) for _ in range(3)]
    gate = cs.Gate()
    g = host.Genome()
    singles = []
    try:
        for c in lanes:
            c.set_gate(gate)
        for k, L in enumerate(GRCH38):
            syn = host.SynthShard(0x5EED0000 + 3000 + k + 1, L, 30.0, 0, 16)
            g.add_synth(ctx, "chr%d" % (k + 1), k, syn, snp_seed=0x5EED0000 + 3000 + k + 1, with_snps=True)
            one = host.Genome()
            one.add_synth(ctx, "chr%d" % (k + 1), k, syn, snp_seed=0x5EED0000 + 3000 + k + 1, with_snps=True)
            syn.free()
            calls1, tid1, st1, _ = one.run(ctx, hmm, capacity=1 << 18)
            singles.append((np.ascontiguousarray(calls1).tobytes(), len(calls1), st1.n_cigar_cn_regions, st1.n_split_calls))
            one.free()
        for rep in range(2):
            calls, tid, st, per = g.run(ctx, hmm, lanes=lanes, capacity=1 << 18)
            assert st.n_reads > 1_500_000 and len(calls) > 5000
            for k in range(3):
                mine = np.ascontiguousarray(calls[tid == k])
                assert len(mine) == singles[k][1], (rep, k)
                assert hashlib.sha256(mine.tobytes()).hexdigest() == hashlib.sha256(singles[k][0]).hexdigest(), (rep, k)
            assert st.n_cigar_cn_regions == sum(s[2] for s in singles) and st.n_split_calls == sum(s[3] for s in singles)
    finally:
        for c in lanes:
            c.set_gate(None)
            c.close()
        gate.close()
        g.free()
