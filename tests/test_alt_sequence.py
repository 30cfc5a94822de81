"""§8a row a2, quirk 5 (src/sv_caller.cpp:572-590, :607-625): an insertion or soft clip of exactly 50 bases carries its inserted
sequence as the ALT allele — cut from the record's 4-bit packed sequence at the op's query offset, with the IUPAC ambiguity codes
R Y K M S W B D H V turned into N; longer ops get "<INS>", deletions "<DEL>". Known answers derived here from the SAM
specification's code table ("=ACMGRSVTWYHKDBN", high nibble first), independent of the product code. CPU only (the cut is host
code: SVCaller::toSVCall); the query offsets themselves come from the scan kernel and are checked in test_gpu_alt_sequence.py."""
import numpy as np

from contextsv_amd import _lib, host

NT16 = "=ACMGRSVTWYHKDBN"
AMBIG = set("RYKMSWBDHV")


def pack(bases: str) -> np.ndarray:
    codes = [NT16.index(b) for b in bases]
    if len(codes) & 1:
        codes.append(0)
    return np.array([(codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2)], np.uint8)


def expected_alt(bases: str, qpos: int, n: int) -> str:
    return "".join("N" if b in AMBIG else b for b in bases[qpos: qpos + n])


def sig(start, length, read, qpos, kind):
    s = np.zeros(1, _lib.SIG_DTYPE)
    s["start"], s["end"], s["read"], s["qpos_kind"] = start, start + length - 1, read, (qpos << 2) | kind
    return s


def test_fifty_base_alt_even_and_odd_offsets_all_codes():
    rng = np.random.default_rng(3)
    # read 0: every code appears; read 1: odd length; read 2: no sequence stored ("*", l_seq = 0)
    r0 = "".join(NT16[i % 16] for i in range(16)) * 8                      # 128 bases, all 16 codes
    r1 = "".join(rng.choice(list("ACGTNRYKM"), 101))
    seqs = [pack(r0), pack(r1), np.zeros(0, np.uint8)]
    off = np.zeros(4, np.uint64)
    off[1:] = np.cumsum([len(x) for x in seqs])
    seq = np.concatenate(seqs)
    cases = [(0, 0, 0), (0, 1, 0), (0, 37, 2), (0, 78, 0), (1, 0, 2), (1, 51, 0), (1, 33, 0)]
    for read, qpos, kind in cases:
        bases = (r0, r1)[read]
        got = host.sig_alts(sig(1000, 50, read, qpos, kind), off, seq)
        assert got == [expected_alt(bases, qpos, 50)], (read, qpos, kind)
        assert len(got[0]) == 50
    # the all-codes read: '=' survives, the ten ambiguity codes become N, A C G T N stay
    assert host.sig_alts(sig(5, 50, 0, 0, 0), off, seq)[0][:16] == "=ACNGNNNTNNNNNNN"
    # 51 bases and more: symbolic; deletions: <DEL> whatever the length
    assert host.sig_alts(sig(1000, 51, 0, 3, 0), off, seq) == ["<INS>"]
    assert host.sig_alts(sig(1000, 50000, 1, 0, 2), off, seq) == ["<INS>"]
    assert host.sig_alts(sig(1000, 50, 0, 3, 1), off, seq) == ["<DEL>"]
    # a record without a stored sequence, an op that runs past the stored bases, and no sequences at all: fifty N
    assert host.sig_alts(sig(1000, 50, 2, 0, 0), off, seq) == ["N" * 50]
    assert host.sig_alts(sig(1000, 50, 1, 60, 0), off, seq) == ["N" * 50]
    assert host.sig_alts(sig(1000, 50, 0, 0, 0)) == ["N" * 50]
