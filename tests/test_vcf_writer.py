"""§8f-1: reference-genome lookup and VCF writer of the host mirror against the oracle restatement
(oracle/vcf_oracle.cpp) and against the one output record the reference's own test quotes
(tests/test_general.py:124 of the reference). CPU only: depth values come from host arrays here; the
device gather behind ShardDepthSource is covered in test_gpu_vcf.py."""
import os

import numpy as np
import pytest

from contextsv_amd import host
from contextsv_amd.host import CALL_DTYPE

DATE = "20250926"
# sv_types.h evidence bit positions
SPLIT, HMM = 1 << 3, 1 << 8


def write_fasta(path, rng):
    """A FASTA with everything the parser treats specially: residues in front of the first header (they run into the
    first contig), descriptions, lower case and IUPAC codes, an empty line, a duplicated contig name (last one wins in the
    sequence table, both stay in the list), ragged line lengths, no final newline."""
    def seq(n):
        return "".join(rng.choice(list("ACGTacgtNRYKMSWBDHVn"), n, p=[.2, .2, .2, .2, .02, .02, .02, .02, .03] + [.009] * 10 + [.0]))
    contigs = {}
    lines = ["ACGT"]                                  # in front of any header
    order = [("chrB", "second contig  desc"), ("chrA", ""), ("chr10", "x"), ("chr2", ""), ("chrA", "again")]
    pending = "ACGT"
    for name, desc in order:
        lines.append(">" + name + (" " + desc if desc else ""))
        s = seq(int(rng.integers(300, 900)))
        w = int(rng.integers(40, 80))
        chunks = [s[i:i + w] for i in range(0, len(s), w)]
        if name == "chr10":
            chunks.insert(2, "")                     # blank line inside a record
        lines += chunks
        contigs[name] = pending + s
        pending = ""
    with open(path, "w") as f:
        f.write("\n".join(lines))                    # no trailing newline
    return contigs


@pytest.fixture(scope="module")
def fasta(tmp_path_factory):
    rng = np.random.default_rng(11)
    path = str(tmp_path_factory.mktemp("fa") / "genome.fa")
    return path, write_fasta(path, rng)


def test_fasta_matches_oracle_and_python(fasta, oracle):
    path, contigs = fasta
    g = host.ReferenceGenome(path)
    hdr, names = oracle.fasta_describe(path)
    assert g.getContigHeader() == hdr
    assert g.getChromosomes() == names == sorted(n.encode() for n in ["chrB", "chrA", "chr10", "chr2", "chrA"])
    assert hdr.decode().split("\n") == [f"##contig=<ID={n},length={len(contigs[n])}>" for n in sorted(contigs)]
    rng = np.random.default_rng(5)
    for name, s in contigs.items():
        assert g.getChromosomeLength(name) == len(s)
        L = len(s)
        cases = [(1, L), (1, 1), (L, L), (1, L + 1), (L + 1, L + 1), (0, 5), (5, 4), (0, 0), (7, 7)]
        cases += [tuple(int(x) for x in rng.integers(0, L + 3, 2)) for _ in range(60)]
        for a, b in cases:
            got = g.query(name, a, b)
            assert got == oracle.fasta_query(path, name, a, b), (name, a, b)
            if 1 <= a <= b <= L:
                assert got == s[a - 1:b].encode()
            elif a != 0:
                assert got == b""
    assert g.getChromosomeLength("nope") == 0
    assert oracle.fasta_query(path, "nope", 1, 2) is None
    with pytest.raises(KeyError):
        g.query("nope", 1, 2)
    s = contigs["chr2"]
    assert g.compare("chr2", 3, 12, s[2:12].encode(), 1.0)
    assert not g.compare("chr2", 3, 12, (("A" if s[2] != "A" else "C") + s[3:12]).encode(), 1.0)
    assert g.compare("chr2", 3, 12, (("A" if s[2] != "A" else "C") + s[3:12]).encode(), 0.9)
    assert not g.compare("chr2", 3, 3, s[2:3].encode(), 0.0)       # pos_start >= pos_end -> false (fasta_query.cpp:114)


def test_fasta_open_errors(tmp_path):
    with pytest.raises(RuntimeError):
        host.ReferenceGenome(str(tmp_path / "missing.fa"))
    with pytest.raises(RuntimeError):
        host.ReferenceGenome("")
    empty = tmp_path / "empty.fa"
    empty.write_text("")
    g = host.ReferenceGenome(str(empty))
    assert g.getChromosomes() == [] and g.getContigHeader() == b""


def random_calls(rng, n, length):
    c = np.zeros(n, CALL_DTYPE)
    c["sv_type"] = rng.choice([-1, 0, 0, 0, 1, 2, 3, 3, 3, 4, 5], n)
    c["start"] = rng.integers(0, length + 5, n)
    span = np.where(rng.random(n) < 0.3, 0, rng.integers(0, 120, n))
    c["end"] = c["start"] + span
    c["cluster_size"] = rng.integers(0, 50, n)
    c["hmm_likelihood"] = np.where(rng.random(n) < 0.3, 0.0, -rng.random(n) * 10 ** rng.integers(0, 6, n))
    c["id"] = np.arange(n)
    c["aln_flags"] = rng.integers(0, 1 << 10, n)
    c["genotype"] = rng.integers(0, 4, n)
    c["cn_state"] = rng.integers(0, 7, n)
    c["aln_offset"] = rng.integers(-300, 300, n)
    alts = []
    for t in c["sv_type"]:
        if t == 3:
            alts.append(b"<INS>" if rng.random() < 0.4 else "".join(rng.choice(list("ACGT"), int(rng.integers(1, 60)))).encode())
        else:
            alts.append({-1: b".", 0: b"<DEL>", 1: b"<DUP>", 2: b"<INV>", 4: b"<BND>", 5: b"."}[int(t)])
    return c, alts


def write_gaps(path, contigs, rng):
    rows = ["# comment", "", "chrA\tnot_a_number\t5", "chrZ\t1\t100"]
    for name, s in contigs.items():
        for _ in range(6):
            a = int(rng.integers(0, len(s)))
            rows.append(f"{name}\t{a}\t{a + int(rng.integers(1, 200))}\tgap")
    rows.append("chr2\t10")                          # too few fields
    with open(path, "w") as f:
        f.write("\n".join(rows) + "\n")


@pytest.mark.parametrize("with_gaps", [False, True])
def test_vcf_matches_oracle(fasta, oracle, tmp_path, with_gaps):
    path, contigs = fasta
    rng = np.random.default_rng(3 + with_gaps)
    g = host.ReferenceGenome(path)
    items = []
    for name in ["chr2", "chrA", "chrB", "chr10"]:
        L = len(contigs[name])
        calls, alts = random_calls(rng, 400, L)
        # depth map one longer than the contig, as the caller builds it (sv_caller.cpp:795-800); chrB's is short so that some
        # positions fall off the end (getReadDepth's out_of_range branch -> 0)
        depth = rng.integers(0, 90, (L + 1) if name != "chrB" else L // 2).astype(np.uint32)
        items.append((name, calls, alts, depth))
    items.append(("chrEmpty", np.zeros(0, CALL_DTYPE), [], None))           # no records: its missing depth map is never touched
    gap = None
    if with_gaps:
        gap = str(tmp_path / "gaps.bed")
        write_gaps(gap, contigs, rng)
    out_dir = tmp_path / "host"
    out_dir.mkdir()
    counts = host.save_vcf(str(out_dir), g, items, gap_path=gap, file_date=DATE)
    want_path = str(tmp_path / "oracle.vcf")
    rc, want_counts = oracle.save_vcf(want_path, path, items, gap_path=gap, file_date=DATE)
    assert rc == 0
    got = (out_dir / "output.vcf").read_bytes()
    want = open(want_path, "rb").read()
    assert got == want
    assert counts == want_counts
    assert counts[0] > 1000 and counts[1] > 100
    assert (counts[2] > 0) == with_gaps
    body = [l for l in got.decode().split("\n") if l and not l.startswith("#")]
    assert any("\tAssemblyGap\t" in l for l in body) == with_gaps
    assert any(";LOH\t" in l for l in body) and any("SUPPORT=0;" in l for l in body)

    # the unordered_map entry point writes the same records (contig order is the map's)
    out2 = tmp_path / "host_map"
    out2.mkdir()
    assert host.save_vcf(str(out2), g, items, gap_path=gap, file_date=DATE, map_order=True) == counts
    got2 = (out2 / "output.vcf").read_bytes().decode().split("\n")
    assert sorted(got2) == sorted(got.decode().split("\n"))


def test_vcf_record_rules(fasta, tmp_path):
    """Hand-checked records: DEL takes the preceding base, INS moves to it, first-position INS is dropped, a DEL past the
    contig end becomes symbolic, IUPAC codes in REF become N."""
    path, contigs = fasta
    s = contigs["chr2"]
    g = host.ReferenceGenome(path)
    c = np.zeros(6, CALL_DTYPE)
    c["genotype"] = [1, 2, 0, 3, 1, 1]
    c["cn_state"] = [1, 5, 4, 0, 2, 3]
    c["sv_type"] = [0, 3, 3, 0, 0, 3]
    c["start"] = [101, 201, 1, len(s) - 3, 1, 301]
    c["end"] = [150, 201, 1, len(s) + 10, 4, 330]
    c["cluster_size"] = [7, 3, 1, 2, 9, 4]
    c["hmm_likelihood"] = [-12.5, 0.0, 0.0, -1e-7, -3.25, -0.5]
    c["aln_flags"] = [1 << 1, 1 << 0, 1 << 0, SPLIT, 1 << 1 | HMM, 1 << 2]
    c["aln_offset"] = [0, 0, 0, -4, 0, 12]
    alts = [b"<DEL>", b"ACGTTT", b"AAA", b"<DEL>", b"<DEL>", b"<INS>"]
    depth = np.arange(len(s) + 1, dtype=np.uint32) % 97
    out = tmp_path / "o"
    out.mkdir()
    counts = host.save_vcf(str(out), g, [("chr2", c, alts, depth)], file_date=DATE)
    assert counts == (6, 0, 0)
    lines = [l.split("\t") for l in (out / "output.vcf").read_text().split("\n") if l and not l.startswith("#")]
    assert len(lines) == 5                                                   # the first-position insertion is dropped after being counted

    def fix(x):
        return "".join("N" if ch in "RYKMSWBDHVrykmswbdhv" else ch for ch in x)
    m = "ContextSV v1.0.0"
    assert lines[0] == ["chr2", "100", ".", fix(s[99:150]), s[99], ".", "PASS",
                        f"END=150;SVTYPE=DEL;SVLEN=-50;SVMETHOD={m};ALN=CIGARDEL;HMM=-12.500000;SUPPORT={100 % 97};CLUSTER=7;ALNOFFSET=0;CN=1", "GT:DP", f"0/1:{100 % 97}"]
    assert lines[1] == ["chr2", "200", ".", fix(s[199]), s[199] + "ACGTTT", ".", "PASS",
                        f"END=200;SVTYPE=INS;SVLEN=1;SVMETHOD={m};ALN=CIGARINS;HMM=0.000000;SUPPORT={200 % 97};CLUSTER=3;ALNOFFSET=0;CN=5", "GT:DP", f"1/1:{200 % 97}"]
    p = len(s) - 4
    assert lines[2] == ["chr2", str(p), ".", "N", "<DEL>", ".", "PASS",
                        f"END={len(s) + 10};SVTYPE=DEL;SVLEN=-14;SVMETHOD={m};ALN=SPLIT;HMM=-0.000000;SUPPORT={p % 97};CLUSTER=2;ALNOFFSET=-4;CN=0", "GT:DP", f"./.:{p % 97}"]
    assert lines[3] == ["chr2", "1", ".", fix(s[0:4]), s[0], ".", "PASS",
                        f"END=4;SVTYPE=DEL;SVLEN=-4;SVMETHOD={m};ALN=CIGARDEL,HMM;HMM=-3.250000;SUPPORT=1;CLUSTER=9;ALNOFFSET=0;CN=2", "GT:DP", "0/1:1"]
    assert lines[4] == ["chr2", "300", ".", fix(s[299]), "<INS>", ".", "PASS",
                        f"END=300;SVTYPE=INS;SVLEN=30;SVMETHOD={m};ALN=CIGARCLIP;HMM=-0.500000;SUPPORT={300 % 97};CLUSTER=4;ALNOFFSET=12;CN=3", "GT:DP", f"0/1:{300 % 97}"]


def test_reference_test_record(fasta, oracle, tmp_path):
    """The record the reference's own test documents as its output (tests/test_general.py:124):
      chr3 61149366 . N <DUP> . PASS END=61925600;SVTYPE=DUP;SVLEN=776235;SVMETHOD=…;ALN=SPLIT,HMM;HMM=-2533.541937;
      SUPPORT=63;CLUSTER=23;ALNOFFSET=0;CN=6  GT:DP  1/1:63
    (SVMETHOD there carries a `git describe` suffix of an older build; the current source prints "ContextSV v1.0.0",
    sv_caller.cpp:1163.) Both the writer and the oracle must produce it from the corresponding SVCall."""
    path, _ = fasta
    g = host.ReferenceGenome(path)
    c = np.zeros(1, CALL_DTYPE)
    c["start"], c["end"], c["sv_type"], c["cluster_size"] = 61149366, 61925600, 1, 23
    c["hmm_likelihood"], c["aln_flags"], c["genotype"], c["cn_state"] = -2533.541937, SPLIT | HMM, 2, 6
    depth = np.zeros(61149367, np.uint32)
    depth[61149366] = 63
    out = tmp_path / "o"
    out.mkdir()
    host.save_vcf(str(out), g, [("chr3", c, [b"<DUP>"], depth)], file_date=DATE)
    want = ("chr3\t61149366\t.\tN\t<DUP>\t.\tPASS\tEND=61925600;SVTYPE=DUP;SVLEN=776235;SVMETHOD=ContextSV v1.0.0;ALN=SPLIT,HMM;"
            "HMM=-2533.541937;SUPPORT=63;CLUSTER=23;ALNOFFSET=0;CN=6\tGT:DP\t1/1:63")
    body = [l for l in (out / "output.vcf").read_text().split("\n") if l and not l.startswith("#")]
    assert body == [want]
    rc, _ = oracle.save_vcf(str(tmp_path / "orc.vcf"), path, [("chr3", c, [b"<DUP>"], depth)], file_date=DATE)
    assert rc == 0
    assert [l for l in open(tmp_path / "orc.vcf").read().split("\n") if l and not l.startswith("#")] == [want]


def test_vcf_header_and_errors(fasta, tmp_path):
    path, contigs = fasta
    g = host.ReferenceGenome(path)
    out = tmp_path / "o"
    out.mkdir()
    host.save_vcf(str(out), g, [], file_date=None)
    head = (out / "output.vcf").read_text().split("\n")
    assert head[0] == "##fileformat=VCFv4.2"
    assert head[1].startswith("##fileDate=") and len(head[1]) == len("##fileDate=") + 8 and head[1][11:].isdigit()
    assert head[2] == "##source=ContextSV v1.0.0"
    assert head[3] == "##reference=" + path
    assert head[4:4 + len(contigs)] == [f"##contig=<ID={n},length={len(contigs[n])}>" for n in sorted(contigs)]
    assert head[-2] == "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE" and head[-1] == ""
    assert sum(l.startswith("##INFO=") for l in head) == 11 and sum(l.startswith("##FILTER=") for l in head) == 3
    assert sum(l.startswith("##FORMAT=") for l in head) == 2

    c = np.zeros(1, CALL_DTYPE)
    c["start"], c["end"], c["sv_type"], c["cn_state"] = 10, 20, 1, 0
    with pytest.raises(RuntimeError):                                        # record on a contig without a depth map
        host.save_vcf(str(out), g, [("chr2", c, [b"<DUP>"], None), ("chrA", c, [b"<DUP>"], np.zeros(30, np.uint32))], file_date=DATE)
    c["sv_type"] = 0
    with pytest.raises(RuntimeError):                                        # deletion on a contig the genome does not have
        host.save_vcf(str(out), g, [("chrQ", c, [b"<DEL>"], np.zeros(30, np.uint32))], file_date=DATE)
    c["cn_state"] = 9
    with pytest.raises(RuntimeError):                                        # copy-number state outside the table
        host.save_vcf(str(out), g, [("chr2", c, [b"<DEL>"], np.zeros(30, np.uint32))], file_date=DATE)
    with pytest.raises(RuntimeError):                                        # unreadable gap file: the reference returns early
        host.save_vcf(str(out), g, [], gap_path=str(tmp_path / "no.bed"), file_date=DATE, map_order=True)
    with pytest.raises(RuntimeError):
        host.save_vcf(str(tmp_path / "no_such_dir"), g, [], file_date=DATE, map_order=True)
