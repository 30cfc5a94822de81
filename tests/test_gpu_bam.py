"""§8f-2 end to end: the same three-contig data set once as in-memory arrays (SVCaller::run) and once as a coordinate-sorted BAM
+ BAI decoded by the host mirror's reader (SVCaller::runBam) must give the same calls and the same VCF; and a synthetic shard
written to BAM and read back is the shard."""
import numpy as np
import pytest

from contextsv_amd import Reads, host, make_hmm
from hmm_params import WGS_HMM
from test_gpu_e2e import CONTIG_LEN, _build, _write_genome

pytestmark = pytest.mark.gpu


def test_run_from_bam_equals_run_from_arrays(ctx, tmp_path):
    contigs = _build(7)
    # SNPs as a VCF for runBam and as the same numbers in arrays for run: BAF = AD[1] / (AD[0] + AD[1]); no population file
    rng = np.random.default_rng(3)
    vcf = ["##fileformat=VCFv4.2", '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">', '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="d">',
           '##FORMAT=<ID=AD,Number=R,Type=Integer,Description="a">', "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS"]
    for t, c in enumerate(contigs):
        pos = c["snps"]["pos"]
        dp = rng.integers(20, 60, len(pos))
        a1 = np.where(rng.random(len(pos)) < 0.6, dp // 2 + rng.integers(-3, 4, len(pos)), rng.choice([0, 1], len(pos)) * dp)
        c["snps"] = {"pos": pos, "baf": a1 / dp, "pfb": np.zeros(len(pos)), "has_pfb": np.zeros(len(pos), np.uint8)}
        vcf += ["contig%d\t%d\t.\tA\tG\t60\tPASS\t.\tGT:DP:AD\t0/1:%d:%d,%d" % (t, p, d, d - a, a) for p, d, a in zip(pos, dp, a1)]
        vcf.append("contig%d\t%d\t.\tA\tGT\t60\tPASS\t.\tGT:DP:AD\t0/1:30:15,15" % (t, int(pos[-1]) + 5))      # an indel: ignored
    snp_vcf = str(tmp_path / "snps.vcf")
    open(snp_vcf, "w").write("\n".join(vcf) + "\n")
    hmm = make_hmm(**WGS_HMM)
    fasta = str(tmp_path / "genome.fa")
    _write_genome(fasta, len(contigs), np.random.default_rng(1))
    genome = host.ReferenceGenome(fasta)
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    want, want_tid = host.run(ctx, contigs, hmm, genome=genome, vcf_dir=str(tmp_path / "a"), file_date="20250926")

    # all contigs into one BAM (records are already position-sorted per contig)
    cat = lambda f, dt: np.ascontiguousarray(np.concatenate([getattr(c["reads"], f) for c in contigs]), dt)
    base = np.concatenate([[0], np.cumsum([c["reads"].n_cigar for c in contigs])]).astype(np.uint64)
    coff = np.concatenate([c["reads"].cigar_off[:-1] + base[i] for i, c in enumerate(contigs)] + [base[-1:]]).astype(np.uint64)
    reads = Reads(cat("pos", np.int32), cat("flag", np.uint16), cat("mapq", np.uint8), coff, cat("cigar", np.uint32))
    tid = np.concatenate([np.full(c["reads"].n_reads, t, np.int32) for t, c in enumerate(contigs)])
    qnames = ["r%d" % q for c in contigs for q in c["qname_id"]]
    bam = str(tmp_path / "reads.bam")
    host.write_bam(bam, ["contig%d" % t for t in range(len(contigs))], [CONTIG_LEN] * len(contigs), tid, reads, qnames, level=1, threads=8)

    got, got_tid, st = host.run_bam(ctx, bam, hmm, threads=8, genome=genome, vcf_dir=str(tmp_path / "b"), file_date="20250926", snp_vcf=snp_vcf)
    assert st["n_contigs"] == len(contigs) and st["n_reads"] == reads.n_reads and st["n_cigar"] == reads.n_cigar
    assert len(got) == len(want) > 20
    assert np.array_equal(got_tid, want_tid)
    assert got.tobytes() == want.tobytes()
    a = sorted((tmp_path / "a" / "output.vcf").read_text().split("\n"))
    b = sorted((tmp_path / "b" / "output.vcf").read_text().split("\n"))
    assert a == b and len(a) > 40

    # --chr: one contig through the index gives that contig's CIGAR calls; split-read evidence needs the other contigs' records
    one, one_tid, st1 = host.run_bam(ctx, bam, hmm, chromosomes=["contig1"], threads=4, split_svs=False, snp_vcf=snp_vcf)
    ref1, _ = host.run(ctx, contigs[1:2], hmm)      # contig named contig0 there; positions and types are what matter
    assert st1["n_contigs"] == 1 and (one_tid == 1).all()
    cig_only = ref1[(ref1["aln_flags"] & 0b111) != 0]
    sel = one[(one["aln_flags"] & 0b111) != 0]
    assert len(sel) > 5 and np.array_equal(np.sort(sel["start"]), np.sort(cig_only["start"]))


def test_synthetic_shard_round_trips_through_bam(ctx, tmp_path):
    syn = host.SynthShard(seed=5, chr_len=3_000_000, depth=8.0, tech=0, threads=4, with_seq=True)
    try:
        path = str(tmp_path / "synth.bam")
        nbytes = syn.write_bam(path, "chrS", level=1, threads=8)
        assert nbytes > 0
        bam = host.BamFile(path)
        assert bam.names == ["chrS"] and bam.lens == [syn.depth_len - 1]
        sh = bam.read_contig("chrS", want_seq=True, want_qnames=True, threads=8)
        r, w = sh["reads"], syn.reads
        for f in ("pos", "flag", "mapq", "cigar_off", "cigar"):
            assert np.array_equal(getattr(r, f), getattr(w, f)), f
        assert sh["qnames"][:3] == ["r0", "r1", "r2"] and len(sh["qnames"]) == w.n_reads
        assert len(sh["seq"]) > 0
        # and the device path sees the same shard either way
        a = ctx.upload(w, syn.depth_len); b = ctx.upload(r, syn.depth_len)
        try:
            ra, rb = a.pipeline(), b.pipeline()
            assert (ra.n_sig, ra.n_del, ra.depth_sum, ra.min_pts) == (rb.n_sig, rb.n_del, rb.depth_sum, rb.min_pts) and ra.n_sig > 100
        finally:
            a.free(); b.free()
    finally:
        syn.free()
