"""§8f-3: SNP / gnomAD VCF ingestion of the host mirror (one pass per file, per-contig tables) against the oracle's per-region
restatement of readSNPAlleleFrequencies, on synthetic VCFs that hit every filter and quirk. CPU only."""
import os

import numpy as np
import pytest

import bam_py
from contextsv_amd import host

HEADER = """##fileformat=VCFv4.2
##FILTER=<ID=LowQual,Description="Low quality">
##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">
##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Read depth">
##FORMAT=<ID=AD,Number=R,Type=Integer,Description="Allelic depths">
##INFO=<ID=DP,Number=1,Type=Integer,Description="Depth">
##contig=<ID=chr1,length=1000000>
##contig=<ID=chr2,length=1000000>
#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE
"""
GNOMAD_HEADER = """##fileformat=VCFv4.2
##INFO=<ID=AF,Number=A,Type=Float,Description="AF">
##INFO=<ID=AF_nfe,Number=A,Type=Float,Description="AF nfe">
##INFO=<ID=AF_int,Number=A,Type=Integer,Description="wrongly typed">
##INFO=<ID=AC,Number=A,Type=Integer,Description="AC">
#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO
"""


def snp_lines(rng, chrom, n, span):
    """Mostly good SNP records plus one of each rejected / odd kind, position-sorted, with duplicates."""
    pos = np.sort(rng.integers(1, span, n))
    out = []
    for k, p in enumerate(pos):
        ref, alt = rng.choice(list("ACGT"), 2, replace=False)
        qual, flt, fmt = "%.1f" % rng.uniform(31, 90), "PASS", "GT:DP:AD"
        dp = int(rng.integers(11, 80))
        a0 = int(rng.integers(0, dp + 1))
        smp = "0/1:%d:%d,%d" % (dp, a0, dp - a0)
        kind = k % 23
        if kind == 1: alt = alt + "T"                       # insertion: not a SNP
        elif kind == 2: ref = ref + "G"                     # deletion
        elif kind == 3: alt = "*"                           # spanning deletion allele
        elif kind == 4: qual = "."                          # QUAL missing
        elif kind == 5: qual = "30"                         # not > 30
        elif kind == 6: smp = "0/1:10:5,5"                  # DP not > 10
        elif kind == 7: flt = "LowQual"
        elif kind == 8: flt = "."                           # no filter applied counts as PASS
        elif kind == 9: fmt, smp = "GT:DP", "0/1:%d" % dp   # no AD
        elif kind == 10: smp = "0/1:%d:7" % dp              # AD with a single value
        elif kind == 11: alt = "%s,%s" % (alt, "<*>")       # multi-allelic with the gVCF symbolic allele: still a SNP
        elif kind == 12: smp = "0/1:.:5,9"                  # DP missing
        elif kind == 13: smp = "0/1:%d:12,." % dp           # second AD value missing: the reference divides by the missing mark
        elif kind == 14: fmt, smp = "GT:AD:DP", "0/1:%d,%d:%d" % (a0, dp - a0, dp)   # key order is free
        elif kind == 15: smp = "0/1:%d" % dp                # AD declared in FORMAT, dropped from the sample column
        elif kind == 16: alt = "."                          # monomorphic site: only REF is tested
        elif kind == 17: qual = "30.0001"
        elif kind == 18: flt = "LowQual;PASS"
        elif kind == 19: smp = "0/1:%d:0,0" % dp            # 0/0 -> NaN BAF
        out.append((int(p), "%s\t%d\t.\t%s\t%s\t%s\t%s\tDP=%d\t%s\t%s" % (chrom, p, ref, alt, qual, flt, dp, fmt, smp)))
        if kind == 20:                                      # the same position twice: the later BAF wins in the map
            out.append((int(p), "%s\t%d\t.\t%s\t%s\t55\tPASS\tDP=40\tGT:DP:AD\t0/1:40:10,30" % (chrom, p, ref, alt)))
    return out


def gnomad_lines(rng, chrom, snp_positions, span):
    """Records on and off the sample's SNP positions; AF values inside, outside and on the (0.01, 0.99) bounds, missing, absent."""
    on = rng.choice(snp_positions, size=len(snp_positions) // 2, replace=False)
    off = rng.integers(1, span, len(snp_positions) // 2)
    out = []
    for k, p in enumerate(np.sort(np.concatenate([on, off]))):
        ref, alt = rng.choice(list("ACGT"), 2, replace=False)
        af = ["0.5", "0.25", "0.001", "0.995", "0.01", "0.99", ".", "1e-1", "0.3,0.4"][k % 9]
        info = "AC=3;AF=%s;AF_nfe=%s;AF_int=1" % (af, af)
        if k % 11 == 3: info = "AC=3;AF=0.2"                 # no AF_nfe
        if k % 11 == 5: alt = alt + "A"                      # indel at a SNP position
        if k % 11 == 7: info = "XAF_nfe=0.5;AC=1"            # a key that only ends like the wanted one
        out.append((int(p), "%s\t%d\trs%d\t%s\t%s\t.\tPASS\t%s" % (chrom, p, k, ref, alt, info)))
    return out


def write_vcf(path, header, lines, compress):
    text = (header + "\n".join(l for _, l in lines) + "\n").encode()
    if not compress:
        open(path, "wb").write(text)
        return
    with open(path, "wb") as f:
        for o in range(0, len(text), 3000):                  # small blocks: lines straddle them
            f.write(bam_py.bgzf_block(text[o:o + 3000]))
        f.write(bam_py.EOF_BLOCK)
    open(path + ".tbi", "wb").write(b"TBI\1")                # presence is what the mirror checks; the data are streamed


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("snp")
    rng = np.random.default_rng(17)
    snps = snp_lines(rng, "chr1", 900, 400_000) + snp_lines(rng, "chr2", 500, 300_000)
    write_vcf(str(d / "sample.vcf"), HEADER, snps, False)
    write_vcf(str(d / "sample.vcf.gz"), HEADER, snps, True)
    g = {}
    for c in ("chr1", "chr2"):
        p = np.array([q for q, l in snps if l.startswith(c + "\t")])
        lines = gnomad_lines(rng, c, p, 400_000)
        (d / "gnomad_txt").mkdir(exist_ok=True)
        write_vcf(str(d / "gnomad_txt" / ("g.%s.vcf" % c)), GNOMAD_HEADER, lines, False)          # path contains "chr": contig keeps its prefix
        write_vcf(str(d / "gnomad_txt" / ("g.%s.vcf.gz" % c)), GNOMAD_HEADER, lines, True)
        g[c] = str(d / "gnomad_txt" / ("g.%s.vcf" % c))
    return d, g


@pytest.mark.parametrize("compressed", [False, True])
@pytest.mark.parametrize("eth", ["nfe", ""])
def test_regions_match_oracle(files, oracle, compressed, eth):
    d, g = files
    ext = ".gz" if compressed else ""
    snp = host.SNPFile(str(d / ("sample.vcf" + ext)), threads=3)
    assert snp.records_kept > 700
    rng = np.random.default_rng(5)
    key = "AF_" + eth if eth else "AF"
    n_pfb = n_nan = n_dup = 0
    for chrom, span in (("chr1", 400_000), ("chr2", 300_000)):
        regions = [(1, span), (span // 2, span // 2), (span + 10, span + 500), (500, 100)]
        for _ in range(60):
            a = int(rng.integers(1, span))
            regions.append((a, a + int(rng.choice([0, 50, 2000, 20_000, 150_000]))))
        for a, b in regions:
            pos, baf, pfb = snp.query(chrom, a, b, pfb_vcf=g[chrom] + ext, ethnicity=eth, threads=3)
            epos, ebaf, epfb = oracle.read_snp_af(str(d / "sample.vcf"), g[chrom], chrom, chrom, a, b, key)
            assert np.array_equal(pos, epos), (chrom, a, b)
            assert np.array_equal(baf, ebaf, equal_nan=True)
            assert (pfb is None) == (epfb is None)
            if pfb is not None:
                assert pfb[0] == epfb[0] and (pfb[1] == epfb[1] or (np.isnan(pfb[1]) and np.isnan(epfb[1])))
                n_pfb += 1
                n_nan += int(np.isnan(pfb[1]))
            n_dup += int(len(pos) != len(set(pos.tolist())))
    assert n_pfb > 50 and n_dup > 10
    # a contig the file does not have, and no gnomAD file at all
    pos, baf, pfb = snp.query("chrX", 1, 10_000_000)
    assert len(pos) == 0 and pfb is None
    pos, baf, pfb = snp.query("chr1", 1, 400_000)           # tables were already built with the gnomAD file: hits stay attached
    assert len(pos) > 300


def test_without_population_file_and_wrong_key_type(files, oracle):
    d, g = files
    snp = host.SNPFile(str(d / "sample.vcf"))
    pos, baf, pfb = snp.query("chr2", 1, 300_000)
    epos, ebaf, epfb = oracle.read_snp_af(str(d / "sample.vcf"), None, "chr2", "chr2", 1, 300_000, "AF")
    assert np.array_equal(pos, epos) and np.array_equal(baf, ebaf, equal_nan=True) and pfb is None and epfb is None
    snp2 = host.SNPFile(str(d / "sample.vcf"))
    pos, baf, pfb = snp2.query("chr1", 1, 400_000, pfb_vcf=g["chr1"], ethnicity="int")      # AF_int is declared Integer: no float values come back
    _, _, epfb = oracle.read_snp_af(str(d / "sample.vcf"), g["chr1"], "chr1", "chr1", 1, 400_000, "AF_int")
    assert pfb is None and epfb is None and len(pos) > 300
    snp3 = host.SNPFile(str(d / "sample.vcf"))
    assert snp3.query("chr1", 1, 400_000, pfb_vcf=g["chr1"], ethnicity="zzz")[2] is None    # key absent from the header


def test_known_answers(tmp_path):
    """Hand-made records: which ones survive, the BAF arithmetic, the first-hit population frequency."""
    lines = [
        (100, "c\t100\t.\tA\tG\t50\tPASS\t.\tGT:DP:AD\t0/1:20:5,15"),          # kept, BAF 0.75
        (150, "c\t150\t.\tA\tG\t50\tq10\t.\tGT:DP:AD\t0/1:20:5,15"),           # filtered
        (200, "c\t200\t.\tC\tT\t31\t.\t.\tGT:DP:AD\t1/1:11:0,11"),             # kept, BAF 1
        (300, "c\t300\t.\tC\tCT\t99\tPASS\t.\tGT:DP:AD\t0/1:40:20,20"),        # indel
        (400, "c\t400\t.\tG\tA\t99\tPASS\t.\tGT:DP:AD\t0/1:40:30,10\t0/1:5:1,1"),   # two samples: the first counts; kept, BAF 0.25
    ]
    write_vcf(str(tmp_path / "s.vcf"), HEADER.replace("\tSAMPLE\n", "\tS1\tS2\n"), lines, False)
    glines = [
        (100, "c\t100\t.\tA\tG\t.\t.\tAF=0.005"),        # out of range: skipped, the search goes on
        (200, "c\t200\t.\tC\tT\t.\t.\tAF=0.4"),          # first acceptable hit
        (400, "c\t400\t.\tG\tA\t.\t.\tAF=0.6"),          # never reached
    ]
    write_vcf(str(tmp_path / "g.vcf"), GNOMAD_HEADER, glines, False)
    snp = host.SNPFile(str(tmp_path / "s.vcf"))
    pos, baf, pfb = snp.query("c", 1, 1000, pfb_vcf=str(tmp_path / "g.vcf"))
    assert pos.tolist() == [100, 200, 400] and baf.tolist() == [0.75, 1.0, 0.25] and pfb == (200, float(np.float32(0.4)))
    pos, baf, pfb = snp.query("c", 300, 1000, pfb_vcf=str(tmp_path / "g.vcf"))
    assert pos.tolist() == [400] and pfb == (400, float(np.float32(0.6)))
    pos, baf, pfb = snp.query("c", 1, 150, pfb_vcf=str(tmp_path / "g.vcf"))
    assert pos.tolist() == [100] and pfb is None


def test_pfb_table_and_contig_naming(tmp_path):
    a, b = tmp_path / "gnomad.1.vcf.gz", tmp_path / "gnomad.X.vcf.gz"
    a.write_bytes(b""); b.write_bytes(b"")
    tab = tmp_path / "pfb.txt"
    tab.write_text("# comment\n1=%s\nX=%s\r\nbroken line\n" % (a, b))
    assert host.pfb_path(str(tab), "chr1") == str(a) and host.pfb_path(str(tab), "1") == str(a)
    assert host.pfb_path(str(tab), "chrX") == str(b) and host.pfb_path(str(tab), "chr7") == ""
    bad = tmp_path / "bad.txt"
    bad.write_text("1=%s\n" % (tmp_path / "missing.vcf.gz"))
    with pytest.raises(RuntimeError, match="does not exist"):
        host.pfb_path(str(bad), "chr1")
    with pytest.raises(RuntimeError, match="does not exist"):
        host.pfb_path(str(tmp_path / "none.txt"), "chr1")
    # the gnomAD contig name follows whether the FILE PATH contains "chr" (cnv_caller.cpp:624-640)
    assert host.gnomad_contig("chr3", "/data/gnomad.3.vcf.gz") == "3"
    assert host.gnomad_contig("3", "/data/gnomad.3.vcf.gz") == "3"
    assert host.gnomad_contig("3", "/data/gnomad.chr3.vcf.gz") == "chr3"
    assert host.gnomad_contig("chr3", "/data/gnomad.chr3.vcf.gz") == "chr3"


def test_open_errors(files, tmp_path):
    d, _ = files
    with pytest.raises(RuntimeError, match="empty"):
        host.SNPFile("")
    with pytest.raises(RuntimeError, match="cannot open"):
        host.SNPFile(str(tmp_path / "missing.vcf.gz"))
    noidx = tmp_path / "noidx.vcf.gz"
    noidx.write_bytes(open(d / "sample.vcf.gz", "rb").read())
    with pytest.raises(RuntimeError, match="index"):
        host.SNPFile(str(noidx))
    import gzip
    plain_gz = tmp_path / "plain.vcf.gz"
    plain_gz.write_bytes(gzip.compress(open(d / "sample.vcf", "rb").read()))
    with pytest.raises(RuntimeError, match="not BGZF"):
        host.SNPFile(str(plain_gz))
