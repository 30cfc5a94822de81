"""End to end over the whole hot path (SVCaller::run pass ordering, sv_caller.cpp:747-946) on three synthetic contigs:
CIGAR scan + depth + ordering + DBSCAN + mergeSVs -> CIGAR copy-number predictions -> split-read signatures -> their
copy-number predictions -> mergeSVs(0.1, 2, keep_noise) -> concatenation -> final mergeSVs — product (GPU kernels + C++ host
mirror) against the same chain composed from the oracle's pieces in Python."""
import numpy as np
import pytest

import oracle_lib
from contextsv_amd import Reads, host, make_hmm
from hmm_params import WGS_HMM
from test_gpu_split import _make_split_shard

pytestmark = pytest.mark.gpu
M, I, D, N, S, H = 0, 1, 2, 3, 4, 5
CONTIG_LEN = 3_000_000


def _cigar_reads(rng, tid, qid0):
    """~12x of 8 kb reads over the first 1.2 Mb with small indels, clips and shared SV events (some >= 2 kb for the CNV pass)."""
    events = []
    for _ in range(14):
        events.append((int(rng.integers(20_000, 1_150_000)), rng.choice(["D", "I"]), int(rng.choice([60, 150, 400, 2500, 6000])), bool(rng.random() < 0.4)))
    events.sort()
    recs = []
    n = int(12 * 1_200_000 / 8000)
    for r in range(n):
        p = int(rng.integers(0, 1_190_000))
        ln = int(rng.integers(5000, 11000))
        hap = int(rng.integers(0, 2))
        ops, ref, end = [], p, p + ln
        if rng.random() < 0.3:
            ops.append((S, int(rng.integers(10, 300))))
        evs = [e for e in events if p + 200 < e[0] < end - 200 and (e[3] or hap == 0)]
        cur = ref
        for (loc, kind, size, _hom) in evs:
            loc += int(rng.integers(-4, 5))
            if loc <= cur + 10:
                continue
            # small indels inside the match run
            while cur + 400 < loc:
                step = int(rng.integers(100, 400))
                ops.append((M, step)); cur += step
                if rng.random() < 0.5:
                    ops.append((I, int(rng.integers(1, 4))))
                else:
                    d = int(rng.integers(1, 4)); ops.append((D, d)); cur += d
            ops.append((M, loc - cur)); cur = loc
            sz = max(50, int(size * (1 + rng.uniform(-0.02, 0.02))))
            if kind == "D":
                ops.append((D, sz)); cur += sz
            else:
                ops.append((I, sz))
        if end > cur:
            ops.append((M, end - cur))
        if rng.random() < 0.3:
            ops.append((S, int(rng.integers(10, 300))))
        fl = 0x10 if rng.random() < 0.5 else 0
        recs.append((tid, p, fl, 60 if rng.random() > 0.05 else 7, ops, qid0 + r))
    return recs, qid0 + n


def _build(seed):
    rng = np.random.default_rng(seed)
    reads_s, tid_s, qn_s, n_contigs = _make_split_shard(seed, n_events=40)
    # unpack the split shard back into record tuples
    recs = []
    for i in range(reads_s.n_reads):
        a, b = int(reads_s.cigar_off[i]), int(reads_s.cigar_off[i + 1])
        ops = [(int(w & 15), int(w >> 4)) for w in reads_s.cigar[a:b]]
        recs.append((int(tid_s[i]), int(reads_s.pos[i]), int(reads_s.flag[i]), int(reads_s.mapq[i]), ops, int(qn_s[i])))
    qid = int(qn_s.max()) + 1
    for t in range(n_contigs):
        extra, qid = _cigar_reads(rng, t, qid)
        recs += extra
    recs.sort(key=lambda r: (r[0], r[1]))
    contigs = []
    for t in range(n_contigs):
        rt = [r for r in recs if r[0] == t]
        reads = Reads.from_cigar_lists([r[1] for r in rt], [r[2] for r in rt], [r[3] for r in rt], [r[4] for r in rt])
        n_snp = 900
        pos = np.sort(rng.choice(np.arange(1000, 1_250_000), n_snp, replace=False)).astype(np.uint32)
        snps = {"pos": pos, "baf": np.clip(np.where(rng.random(n_snp) < 0.6, 0.5 + rng.normal(0, 0.05, n_snp), rng.choice([0.0, 1.0], n_snp)), 0, 1),
                "pfb": rng.uniform(0.05, 0.95, n_snp), "has_pfb": (rng.random(n_snp) < 0.2).astype(np.uint8)}
        contigs.append({"reads": reads, "depth_len": CONTIG_LEN + 1, "qname_id": np.array([r[5] for r in rt], np.uint32), "snps": snps})
    return contigs


from oracle_chain import oracle_run


def _write_genome(path, n_contigs, rng):
    with open(path, "wb") as f:
        for t in range(n_contigs):
            f.write(b">contig%d synthetic\n" % t)
            seq = np.frombuffer(b"ACGTNRacgt", np.uint8)[rng.choice(10, CONTIG_LEN, p=[.24, .24, .24, .24, .01, .01, .005, .005, .005, .005])]
            f.write(b"\n".join(seq[i:i + 60].tobytes() for i in range(0, CONTIG_LEN, 60)) + b"\n")


def test_whole_path_end_to_end(ctx, oracle, tmp_path):
    contigs = _build(7)
    hmm = make_hmm(**WGS_HMM)
    fasta = str(tmp_path / "genome.fa")
    _write_genome(fasta, len(contigs), np.random.default_rng(1))
    gaps = str(tmp_path / "gaps.bed")
    with open(gaps, "w") as f:
        f.write("contig0\t100000\t400000\ncontig2\t0\t50000\n")
    genome = host.ReferenceGenome(fasta)
    got, got_tid, got_alts = host.run(ctx, contigs, hmm, eps=0.1, min_pts_pct=0.1, genome=genome, vcf_dir=str(tmp_path), gap_path=gaps,
                                      file_date="20250926", want_alts=True, save_cnv=True)

    # ---- the same chain from the oracle's pieces --------------------------------------------------
    exp, exp_tid, depths, means = oracle_run(oracle, contigs, hmm)

    assert len(got) == len(exp) and len(got) > 20
    assert np.array_equal(got_tid, exp_tid)
    for f in ("start", "end", "sv_type", "cluster_size", "aln_flags", "genotype", "cn_state", "aln_offset"):
        assert np.array_equal(got[f], exp[f]), f
    np.testing.assert_allclose(got["hmm_likelihood"], exp["hmm_likelihood"], rtol=0, atol=1e-6)
    assert (got["cn_state"] != 0).any() and (got["aln_flags"] & (1 << 8)).any()       # the HMM did update calls
    assert ((got["aln_flags"] & (1 << 3)) != 0).any() or ((got["aln_flags"] & (1 << 4)) != 0).any()   # split evidence survived the merges

    # ---- the VCF the run wrote (SUPPORT / DP gathered from the resident depth maps) against the oracle's writer fed with the
    # oracle's depth maps; contig order in the file is the unordered_map's, so compare per contig
    items = [("contig%d" % t, got[got_tid == t], [a for a, tt in zip(got_alts, got_tid) if tt == t], depths[t]) for t in range(len(contigs))]
    rc, counts = oracle.save_vcf(str(tmp_path / "oracle.vcf"), fasta, items, gap_path=gaps, file_date="20250926")
    assert rc == 0 and counts[0] > 20 and counts[2] > 0
    got_lines = (tmp_path / "output.vcf").read_text().split("\n")
    exp_lines = (tmp_path / "oracle.vcf").read_text().split("\n")
    assert [l for l in got_lines if l.startswith("#")] == [l for l in exp_lines if l.startswith("#")]
    for t in range(len(contigs)):
        pre = "contig%d\t" % t
        assert [l for l in got_lines if l.startswith(pre)] == [l for l in exp_lines if l.startswith(pre)]
    assert len(got_lines) == len(exp_lines)
    assert any("\tAssemblyGap\t" in l for l in got_lines) and any("SVTYPE=DEL" in l for l in got_lines) and any("SVTYPE=INS" in l for l in got_lines)

    # ---- --save-cnv: CNVCalls.json (cnv_caller.cpp:243-284, :811-974): one record per split-read region of >= 30 kb with a predicted
    # copy-number change, arrays = the observation vectors of the region and of its two flanking half-length windows
    import json
    text = (tmp_path / "CNVCalls.json").read_text()
    assert text.endswith("}\n]")
    if text == "}\n]":
        pytest.skip("no >= 30 kb copy-number change among the split-read regions of this data set")
    records = json.loads(text)
    assert len(records) >= 1
    for rec in records:
        t = int(rec["chromosome"][len("contig"):])
        start, end = rec["start"], rec["end"]
        assert rec["size"] == end - start + 1 and end - start >= 30000 and rec["sv_type"] in ("DEL", "DUP", "LOH")
        half = (end - start) // 2
        last = len(depths[t]) - 1
        windows = {"sv": (start, end), "before_sv": (max(1, start - half), max(1, start - 1)), "after_sv": (min(last, end + 1), min(last, end + half))}
        for name, (a, b) in windows.items():
            got_w = rec[name]
            if name != "sv" and not a < b:
                assert got_w["positions"] == []
                continue
            exp_w = oracle.query_snp_region(depths[t], a, b, means[t], 20, contigs[t]["snps"])
            assert got_w["positions"] == exp_w["pos"].tolist()
            assert got_w["is_snp"] == exp_w["is_snp"].astype(int).tolist()
            keep = exp_w["is_snp"]
            np.testing.assert_allclose(got_w["b_allele_freq"], np.where(keep, exp_w["baf"], 0.0), rtol=2e-5, atol=1e-12)
            np.testing.assert_allclose(got_w["population_freq"], np.where(keep, exp_w["pfb"], 0.0), rtol=2e-5, atol=1e-12)
            np.testing.assert_allclose(got_w["log2_ratio"], exp_w["log2_cov"], rtol=2e-5, atol=1e-12)
        sv = oracle.query_snp_region(depths[t], start, end, means[t], 20, contigs[t]["snps"])
        st, ll = oracle.viterbi(hmm, sv["log2_cov"], sv["baf"], sv["pfb"], np.array([0, len(sv["pos"])], np.uint64))
        assert rec["sv"]["states"] == st.tolist()
        assert abs(rec["likelihood"] - ll[0]) <= 2e-5 * abs(ll[0]) + 1e-9
