"""§8a row a2, quirk 5 end to end: the scan kernel's query offset (odd, even, and after a soft clip that the reference skips
TOGETHER with its cursor update, src/sv_caller.cpp:602-604) selects the 50 bases that SVCaller::processChromosome cuts from the
SeqStore as the ALT allele (:572-590, :607-625). One signature per shard, so mergeSVs passes the call through untouched (:49-51).
Expected strings are derived here from the CIGARs by hand."""
import numpy as np
import pytest

from contextsv_amd import Reads, host
from test_alt_sequence import AMBIG, NT16, pack

pytestmark = pytest.mark.gpu
M, I, D, N, S, H = 0, 1, 2, 3, 4, 5


def _bases(seed, n):
    rng = np.random.default_rng(seed)
    return "".join(rng.choice(list(NT16), n))


def _alt(b):
    return "".join("N" if c in AMBIG else c for c in b)


CASES = [
    # (pos, cigar, depth_len, expected (start, end, query offset of the 50 bases))
    ("odd offset behind a short clip", 100, [(S, 7), (M, 20), (I, 50), (M, 30)], 5000, (121, 170, 27)),
    ("even offset", 100, [(M, 20), (I, 50), (M, 30)], 5000, (121, 170, 20)),
    ("end clip of fifty bases", 100, [(M, 20), (I, 4), (M, 10), (S, 50)], 5000, (131, 180, 34)),
    ("leading clip of fifty bases", 100, [(S, 50), (M, 40)], 5000, (101, 150, 0)),
    # the read lies beyond the contig: its 60-base clip is skipped WITH its cursor update, so the insertion behind it is cut at
    # query offset 10, not 70 (the reference's `continue`)
    ("behind a skipped clip", 1200, [(S, 60), (M, 10), (I, 50), (M, 10)], 1000, (1211, 1260, 10)),
    ("deletion and hard clip in front", 300, [(H, 9), (M, 11), (D, 5), (M, 6), (I, 50), (M, 9)], 5000, (323, 372, 17)),
]


@pytest.mark.parametrize("name,pos,cigar,depth_len,exp", CASES, ids=[c[0] for c in CASES])
def test_fifty_base_alt_through_process_chromosome(ctx, name, pos, cigar, depth_len, exp):
    host.set_context(ctx)
    qlen = sum(l for op, l in cigar if op in (M, I, S))
    bases = _bases(len(name), qlen)
    reads = Reads.from_cigar_lists([pos], [0], [60], [cigar])
    seq = pack(bases)
    seq_off = np.array([0, len(seq)], np.uint64)
    sh = ctx.upload(reads, depth_len)
    try:
        calls = host.process_resident_chromosome_alts(ctx, sh, 0.1, 0.1, seq_off, seq)
        start, end, q = exp
        assert calls == [(start, end, _alt(bases[q: q + 50]))]
        # without sequences the ALT is fifty N; a 51-base op is symbolic
        assert host.process_resident_chromosome_alts(ctx, sh, 0.1, 0.1) == [(start, end, "N" * 50)]
    finally:
        sh.free()
    longer = [(op, l + 1 if l == 50 else l) for op, l in cigar]
    sh = ctx.upload(Reads.from_cigar_lists([pos], [0], [60], [longer]), depth_len)
    try:
        calls = host.process_resident_chromosome_alts(ctx, sh, 0.1, 0.1, np.array([0, len(seq)], np.uint64), seq)
        assert len(calls) == 1 and calls[0][2] == "<INS>" and calls[0][1] - calls[0][0] == 50
    finally:
        sh.free()
