"""Independent pure-Python BGZF / BAM / BAI writer and parser for the tests (SAMv1 §4.1, §4.2, §5.2, §5.3; Python's zlib
and gzip modules). htslib is not in this image and the reference ships no BAM, so the C++ reader and writer of the host
mirror are checked in both directions against this second implementation of the specification (parity unpinned by
reference fixtures)."""
import gzip
import struct
import zlib

EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def bgzf_block(data: bytes, level=6) -> bytes:
    assert len(data) <= 0xff00
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    total = 18 + len(comp) + 8
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", total - 1) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def ref_len(cigar):
    return sum(w >> 4 for w in cigar if (w & 15) in (0, 2, 3, 7, 8))


def end_pos(pos, flag, cigar):
    rl = 0 if (flag & 4) else ref_len(cigar)
    return pos + (rl if rl > 0 else 1)


def encode_record(tid, pos, mapq, flag, qname: bytes, cigar, seq4: bytes = b"", l_seq=0, aux: bytes = b"", force_cg=False):
    """One BAM alignment record (with its block_size prefix). CIGARs over 65535 ops (or force_cg) go to a CG:B,I tag."""
    end = end_pos(pos, flag, cigar)
    long = len(cigar) > 65535 or force_cg
    if long:
        cig_field = struct.pack("<II", (l_seq << 4) | 4, (ref_len(cigar) << 4) | 3)
        aux = aux + b"CGBI" + struct.pack("<I", len(cigar)) + struct.pack("<%dI" % len(cigar), *cigar)
        n_cig = 2
    else:
        cig_field = struct.pack("<%dI" % len(cigar), *cigar)
        n_cig = len(cigar)
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(qname) + 1, mapq, reg2bin(pos, end), n_cig, flag, l_seq, -1, -1, 0)
    body += qname + b"\0" + cig_field + seq4 + b"\xff" * l_seq + aux
    return struct.pack("<I", len(body)) + body


def write_bam(path, ref_names, ref_lens, records, text=b"", block_payload=0xff00, level=6, with_index=True, stray_eof_blocks=False):
    """records: iterable of already-encoded record bytes plus their (tid, beg, end, flag) for the index:
    [(bytes, tid, beg, end, flag)], coordinate-sorted. Cuts the stream into BGZF blocks of `block_payload` bytes regardless of
    record boundaries (records straddle blocks). Returns the list of (virtual offset start, virtual offset end) per record."""
    head = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(ref_names))
    for n, l in zip(ref_names, ref_lens):
        nb = n.encode() + b"\0"
        head += struct.pack("<I", len(nb)) + nb + struct.pack("<I", l)
    stream = bytearray(head)
    spans = []
    for rec, *_ in records:
        spans.append((len(stream), len(stream) + len(rec)))
        stream += rec
    # blocks
    blocks, coffs, coff = [], [], 0
    for o in range(0, len(stream), block_payload):
        b = bgzf_block(bytes(stream[o:o + block_payload]), level)
        coffs.append(coff)
        blocks.append(b)
        coff += len(b)
        if stray_eof_blocks and (o // block_payload) % 3 == 1:       # empty blocks in the middle of the file are legal
            blocks.append(EOF_BLOCK)
            coff += len(EOF_BLOCK)
    coffs.append(coff)                                                # the EOF block

    def voff(u):
        return (coffs[u // block_payload] << 16) | (u % block_payload)
    with open(path, "wb") as f:
        f.write(b"".join(blocks) + EOF_BLOCK)
    v = [(voff(a), voff(b)) for a, b in spans]
    if with_index:
        write_bai(path + ".bai", len(ref_names), [(r[1], r[2], r[3], r[4], v[i][0], v[i][1]) for i, r in enumerate(records)])
    return v


def write_bai(path, n_ref, entries):
    """entries: (tid, beg, end, flag, v0, v1) in file order."""
    refs = [{"bins": {}, "lin": [], "last": None, "b": None, "e": 0, "m": 0, "u": 0} for _ in range(n_ref)]
    no_coor = 0
    for tid, beg, end, flag, v0, v1 in entries:
        if tid < 0:
            no_coor += 1
            continue
        r = refs[tid]
        b = reg2bin(beg, end)
        ch = r["bins"].setdefault(b, [])
        if r["last"] == b and ch:
            ch[-1][1] = v1
        else:
            ch.append([v0, v1])
        r["last"] = b
        w0, w1 = max(beg, 0) >> 14, max(end - 1, 0) >> 14
        while len(r["lin"]) <= w1:
            r["lin"].append(None)
        for w in range(w0, w1 + 1):
            if r["lin"][w] is None:
                r["lin"][w] = v0
        r["b"] = v0 if r["b"] is None else min(r["b"], v0)
        r["e"] = max(r["e"], v1)
        if flag & 4:
            r["u"] += 1
        else:
            r["m"] += 1
    out = b"BAI\1" + struct.pack("<I", n_ref)
    for r in refs:
        bins = sorted(r["bins"].items())
        out += struct.pack("<I", len(bins) + (1 if bins else 0))
        for b, ch in bins:
            out += struct.pack("<II", b, len(ch)) + b"".join(struct.pack("<QQ", c[0], c[1]) for c in ch)
        if bins:
            out += struct.pack("<IIQQQQ", 37450, 2, r["b"], r["e"], r["m"], r["u"])
        lin = r["lin"]
        for w in range(len(lin) - 1, -1, -1):
            if lin[w] is None:
                lin[w] = lin[w + 1] if w + 1 < len(lin) else 0
        out += struct.pack("<I", len(lin)) + b"".join(struct.pack("<Q", x) for x in lin)
    out += struct.pack("<Q", no_coor)
    with open(path, "wb") as f:
        f.write(out)


def parse_bam(path):
    """-> (text, [(name, len)], [record dict]) using gzip (multi-member) for the BGZF layer."""
    with open(path, "rb") as f:
        raw = f.read()
    assert raw.endswith(EOF_BLOCK)
    data = gzip.decompress(raw)
    assert data[:4] == b"BAM\1"
    l_text, = struct.unpack_from("<I", data, 4)
    text = data[8:8 + l_text]
    o = 8 + l_text
    n_ref, = struct.unpack_from("<I", data, o)
    o += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<I", data, o)
        name = data[o + 4:o + 4 + l_name - 1].decode()
        l_ref, = struct.unpack_from("<I", data, o + 4 + l_name)
        refs.append((name, l_ref))
        o += 8 + l_name
    recs = []
    while o < len(data):
        bs, = struct.unpack_from("<I", data, o)
        tid, pos, l_name, mapq, bin_, n_cig, flag, l_seq, _, _, _ = struct.unpack_from("<iiBBHHHiiii", data, o + 4)
        p = o + 4 + 32
        qname = data[p:p + l_name - 1]
        p += l_name
        cigar = list(struct.unpack_from("<%dI" % n_cig, data, p))
        p += 4 * n_cig
        seq = data[p:p + (l_seq + 1) // 2]
        p += (l_seq + 1) // 2 + l_seq
        aux = data[p:o + 4 + bs]
        if aux[:4] == b"CGBI":
            cnt, = struct.unpack_from("<I", aux, 4)
            cigar = list(struct.unpack_from("<%dI" % cnt, aux, 8))
        recs.append({"tid": tid, "pos": pos, "mapq": mapq, "bin": bin_, "flag": flag, "l_seq": l_seq, "qname": qname, "cigar": cigar, "seq": seq})
        o += 4 + bs
    return text, refs, recs


def parse_bai(path):
    with open(path, "rb") as f:
        d = f.read()
    assert d[:4] == b"BAI\1"
    n_ref, = struct.unpack_from("<I", d, 4)
    o = 8
    refs = []
    for _ in range(n_ref):
        n_bin, = struct.unpack_from("<I", d, o)
        o += 4
        bins = {}
        for _ in range(n_bin):
            b, nc = struct.unpack_from("<II", d, o)
            o += 8
            bins[b] = [struct.unpack_from("<QQ", d, o + 16 * i) for i in range(nc)]
            o += 16 * nc
        n_intv, = struct.unpack_from("<I", d, o)
        lin = list(struct.unpack_from("<%dQ" % n_intv, d, o + 4))
        o += 4 + 8 * n_intv
        refs.append({"bins": bins, "lin": lin})
    no_coor = struct.unpack_from("<Q", d, o)[0] if o + 8 <= len(d) else None
    return refs, no_coor
