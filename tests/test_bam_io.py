"""§8f-2: BGZF/BAM/BAI reader and writer of the host mirror (no htslib) against an independent pure-Python implementation of
the SAM specification (tests/bam_py.py), in both directions, plus corrupt-input behaviour. CPU only."""
import os
import struct

import numpy as np
import pytest

import bam_py
from contextsv_amd import Reads, host

M, I, D, N, S, H, P, EQ, X = range(9)
REFS = [("chr1", 400_000), ("chr2", 250_000), ("chrEmpty", 1000), ("chr3", 90_000)]


def make_records(seed, n_per=(300, 200, 0, 120), long_cigars=True):
    """Coordinate-sorted records over REFS: every CIGAR op, unmapped-but-placed reads, odd and even sequence lengths, auxiliary
    fields of every type in front of a CG tag, one CIGAR of more than 65535 operations, unplaced reads at the end."""
    rng = np.random.default_rng(seed)
    recs = []
    qid = 0
    for tid, n in enumerate(n_per):
        L = REFS[tid][1]
        starts = np.sort(rng.integers(0, L - 30_000, n))
        for k, p in enumerate(starts):
            n_ops = int(rng.integers(1, 60))
            cig = []
            if rng.random() < 0.3:
                cig.append((int(rng.integers(1, 300)) << 4) | (H if rng.random() < 0.3 else S))
            for _ in range(n_ops):
                cig.append((int(rng.integers(1, 400)) << 4) | int(rng.choice([M, M, M, EQ, X])))
                cig.append((int(rng.integers(1, 80)) << 4) | int(rng.choice([I, D, N, P])))
            cig.append((int(rng.integers(1, 400)) << 4) | M)
            flag = int(rng.choice([0, 16, 256, 2048, 2064, 1024, 4, 512]))
            if flag & 4:
                cig = [] if rng.random() < 0.5 else cig
            force_cg = bool(long_cigars and k % 37 == 5 and not (flag & 4))
            if long_cigars and tid == 0 and k == 11:
                cig = [((1 + (j % 3)) << 4) | (M if j % 2 == 0 else (I if j % 4 == 1 else D)) for j in range(70_001)]
                flag = 0
            l_seq = int(rng.integers(0, 40)) if not force_cg else sum(w >> 4 for w in cig if (w & 15) in (M, I, S, EQ, X)) % 50 + 1
            seq = bytes(rng.integers(0, 256, (l_seq + 1) // 2, dtype=np.uint8)) if l_seq else b""
            if l_seq % 2:
                seq = seq[:-1] + bytes([seq[-1] & 0xF0])
            aux = b""
            if k % 5 == 0:
                aux = (b"NMi" + struct.pack("<i", 7) + b"XAA" + b"q" + b"XcC" + b"\x05" + b"XsS" + struct.pack("<H", 9) + b"Xff" + struct.pack("<f", 1.5) + b"Xdd" + struct.pack("<d", 2.5) +
                       b"MDZ" + b"10A5^AC6\0" + b"XHH" + b"1AE3\0" + b"XBB" + b"s" + struct.pack("<I", 3) + struct.pack("<3h", 1, -2, 3))
            qname = b"read%d/%d" % (qid, tid)
            qid += 1
            recs.append({"tid": tid, "pos": int(p), "mapq": int(rng.integers(0, 61)), "flag": flag, "qname": qname, "cigar": cig, "seq": seq,
                         "l_seq": l_seq, "aux": aux, "force_cg": force_cg})
    for k in range(7):                                  # unplaced reads at the end of the file
        recs.append({"tid": -1, "pos": -1, "mapq": 0, "flag": 4, "qname": b"unplaced%d" % k, "cigar": [], "seq": b"\x12\x48", "l_seq": 4, "aux": b"",
                     "force_cg": False})
    return recs


def encode(recs):
    out = []
    for r in recs:
        b = bam_py.encode_record(r["tid"], r["pos"], r["mapq"], r["flag"], r["qname"], r["cigar"], r["seq"], r["l_seq"], r["aux"], r["force_cg"])
        out.append((b, r["tid"], r["pos"], bam_py.end_pos(r["pos"], r["flag"], r["cigar"]), r["flag"]))
    return out


def check_shard(sh, recs, tid, want_seq=True, want_qnames=True):
    exp = [r for r in recs if r["tid"] == tid]
    rd = sh["reads"]
    assert sh["tid"] == tid and sh["name"] == REFS[tid][0] and sh["target_len"] == REFS[tid][1]
    assert rd.n_reads == len(exp)
    assert np.array_equal(rd.pos, [r["pos"] for r in exp])
    assert np.array_equal(rd.flag, [r["flag"] for r in exp])
    assert np.array_equal(rd.mapq, [r["mapq"] for r in exp])
    assert np.array_equal(rd.cigar_off, np.concatenate([[0], np.cumsum([len(r["cigar"]) for r in exp])]))
    assert np.array_equal(rd.cigar, np.concatenate([np.asarray(r["cigar"], np.uint32) for r in exp]))
    if want_qnames:
        assert sh["qnames"] == [r["qname"].decode() for r in exp]
    if want_seq:
        assert np.array_equal(sh["seq_off"], np.concatenate([[0], np.cumsum([len(r["seq"]) for r in exp])]))
        assert sh["seq"].tobytes() == b"".join(r["seq"] for r in exp)


@pytest.fixture(scope="module")
def py_bam(tmp_path_factory):
    recs = make_records(1)
    path = str(tmp_path_factory.mktemp("bam") / "py.bam")
    # 700-byte blocks: almost every record straddles blocks, and the header spans several; a few empty blocks mid-file
    bam_py.write_bam(path, [r[0] for r in REFS], [r[1] for r in REFS], encode(recs), text=b"@HD\tVN:1.6\tSO:coordinate\n", block_payload=700,
                     stray_eof_blocks=True)
    return path, recs


@pytest.mark.parametrize("threads,window", [(1, 1), (3, 5), (8, 0)])
def test_reader_on_python_written_bam(py_bam, threads, window):
    path, recs = py_bam
    bam = host.BamFile(path)
    assert bam.names == [r[0] for r in REFS] and bam.lens == [r[1] for r in REFS]
    assert bam.text == "@HD\tVN:1.6\tSO:coordinate\n"
    shards, unplaced = bam.read_all(want_seq=True, want_qnames=True, threads=threads, window_blocks=window)
    assert unplaced == 7
    assert [s["tid"] for s in shards] == [0, 1, 3]                    # the empty contig yields no shard
    for s in shards:
        check_shard(s, recs, s["tid"])
    for tid in (3, 0, 1):                                              # any order, through the index
        check_shard(bam.read_contig(REFS[tid][0], want_seq=True, want_qnames=True, threads=threads, window_blocks=window), recs, tid)
    empty = bam.read_contig("chrEmpty")
    assert empty["reads"].n_reads == 0 and empty["tid"] == 2
    assert max(len(r["cigar"]) for r in recs) > 65535                  # the CG-tag CIGAR was really exercised
    with pytest.raises(RuntimeError, match="unknown contig"):
        bam.read_contig("chrNope")
    lean = bam.read_contig("chr2")
    check_shard(lean, recs, 1, want_seq=False, want_qnames=False)


def test_writer_against_python_parser_and_index(tmp_path):
    recs = make_records(2, long_cigars=True)
    placed = [r for r in recs]
    n = len(placed)
    reads = Reads.from_cigar_lists([r["pos"] for r in placed], [r["flag"] for r in placed], [r["mapq"] for r in placed],
                                   [[(w & 15, w >> 4) for w in r["cigar"]] for r in placed])
    seq_off = np.concatenate([[0], np.cumsum([len(r["seq"]) for r in placed])]).astype(np.uint64)
    seq = np.frombuffer(b"".join(r["seq"] for r in placed), np.uint8)
    path = str(tmp_path / "cpp.bam")
    host.write_bam(path, [r[0] for r in REFS], [r[1] for r in REFS], [r["tid"] for r in placed], reads, [r["qname"].decode() for r in placed],
                   seq_off, seq, [r["l_seq"] for r in placed], text="@HD\tVN:1.6\tSO:coordinate\n", level=6, threads=3)
    text, refs, got = bam_py.parse_bam(path)
    assert text == b"@HD\tVN:1.6\tSO:coordinate\n" and refs == REFS
    assert len(got) == n
    for g, r in zip(got, placed):
        for k in ("tid", "pos", "mapq", "flag", "l_seq", "qname", "cigar", "seq"):
            assert g[k] == r[k], k
        assert g["bin"] == bam_py.reg2bin(r["pos"], bam_py.end_pos(r["pos"], r["flag"], r["cigar"]))

    # the index: recompute the virtual offsets from the file itself, build the BAI with the Python builder, compare bytes
    raw = open(path, "rb").read()
    blocks, o = [], 0
    while o < len(raw):
        bsize = struct.unpack_from("<H", raw, o + 16)[0] + 1
        isize = struct.unpack_from("<I", raw, o + bsize - 4)[0]
        blocks.append((o, isize))
        o += bsize
    ustart = np.concatenate([[0], np.cumsum([b[1] for b in blocks])])

    def voff(u):
        k = min(int(np.searchsorted(ustart, u, side="right") - 1), len(blocks) - 1)     # the very end resolves to the EOF marker block
        return (blocks[k][0] << 16) | (u - int(ustart[k]))
    head_len = 12 + len(text) + sum(8 + len(nm) + 1 for nm, _ in REFS)
    entries, u = [], head_len
    for r in placed:
        ln = len(bam_py.encode_record(r["tid"], r["pos"], r["mapq"], r["flag"], r["qname"], r["cigar"], r["seq"], r["l_seq"], b"", False))
        entries.append((r["tid"], r["pos"], bam_py.end_pos(r["pos"], r["flag"], r["cigar"]), r["flag"], voff(u), voff(u + ln)))
        u += ln
    assert u == ustart[-1]
    bam_py.write_bai(str(tmp_path / "py.bai"), len(REFS), entries)
    assert open(path + ".bai", "rb").read() == open(tmp_path / "py.bai", "rb").read()
    idx, no_coor = bam_py.parse_bai(path + ".bai")
    assert no_coor == 7 and 37450 in idx[0]["bins"] and idx[2]["bins"] == {}

    # and the reader on the writer's file
    bam = host.BamFile(path)
    shards, unplaced = bam.read_all(want_seq=True, want_qnames=True, threads=4)
    assert unplaced == 7 and [s["tid"] for s in shards] == [0, 1, 3]
    for s in shards:
        check_shard(s, recs, s["tid"])
        check_shard(bam.read_contig(s["name"], want_seq=True, want_qnames=True), recs, s["tid"])


def test_reg2bin_and_endpos_conventions():
    # SAMv1 §5.3 worked values and bam_endpos' "at least one base" rule
    assert bam_py.reg2bin(0, 1) == 4681 and bam_py.reg2bin(0, 1 << 14) == 4681 and bam_py.reg2bin(0, (1 << 14) + 1) == 585
    assert bam_py.reg2bin((1 << 26) - 1, (1 << 26) + 1) == 0
    assert bam_py.end_pos(100, 0, [(5 << 4) | I]) == 101 and bam_py.end_pos(100, 4, [(50 << 4) | M]) == 101
    assert bam_py.end_pos(100, 0, [(50 << 4) | M, (3 << 4) | D, (2 << 4) | N, (4 << 4) | EQ, (1 << 4) | X, (9 << 4) | S]) == 160


def test_corrupt_and_missing_inputs(py_bam, tmp_path):
    path, _ = py_bam
    raw = open(path, "rb").read()
    with pytest.raises(RuntimeError, match="cannot open"):
        host.BamFile(str(tmp_path / "nope.bam"))
    (tmp_path / "empty.bam").write_bytes(b"")
    with pytest.raises(RuntimeError, match="empty"):
        host.BamFile(str(tmp_path / "empty.bam"))
    (tmp_path / "text.bam").write_bytes(b"@HD\tVN:1.6\n" * 10)
    with pytest.raises(RuntimeError, match="not a BGZF"):
        host.BamFile(str(tmp_path / "text.bam"))
    (tmp_path / "gz.bam").write_bytes(bam_py.bgzf_block(b"SAM\1" + b"\0" * 20) + bam_py.EOF_BLOCK)
    with pytest.raises(RuntimeError, match="bad magic"):
        host.BamFile(str(tmp_path / "gz.bam"))
    # no index next to the file
    (tmp_path / "noidx.bam").write_bytes(raw)
    with pytest.raises(RuntimeError, match="index"):
        host.BamFile(str(tmp_path / "noidx.bam"))
    plain = host.BamFile(str(tmp_path / "noidx.bam"), load_index=False)
    shards, _ = plain.read_all()
    assert len(shards) == 3
    with pytest.raises(RuntimeError, match="no index"):
        plain.read_contig("chr1")
    # flipped payload byte in the middle of the file: CRC or inflate error, never silent
    bad = bytearray(raw)
    bad[len(bad) // 2] ^= 0x5A
    (tmp_path / "crc.bam").write_bytes(bytes(bad))
    with pytest.raises(RuntimeError, match="BGZF|BAM"):
        host.BamFile(str(tmp_path / "crc.bam"), load_index=False).read_all()
    # file cut in the middle of a block / of a record
    (tmp_path / "cut.bam").write_bytes(raw[: len(raw) // 2])
    with pytest.raises(RuntimeError, match="truncated"):
        host.BamFile(str(tmp_path / "cut.bam"), load_index=False).read_all()
    # <name>.bai next to <name>.bam is found as well as <name>.bam.bai
    (tmp_path / "alt.bam").write_bytes(raw)
    (tmp_path / "alt.bai").write_bytes(open(path + ".bai", "rb").read())
    assert host.BamFile(str(tmp_path / "alt.bam")).read_contig("chr3")["reads"].n_reads == 120
