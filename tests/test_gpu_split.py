"""Split-read signatures (§8a row a8): the host mirror of findSplitSVSignatures fed by the scan kernel's alignment intervals
and ONE batched DBSCAN1D launch per contig, against the literal oracle restatement (sequential fits, recursive interval
tree) fed by the oracle's own intervals. Calls must be identical in content and order."""
import numpy as np
import pytest

from contextsv_amd import Reads, host

pytestmark = pytest.mark.gpu
M, I, D, N, S, H = 0, 1, 2, 3, 4, 5


def _make_split_shard(seed, n_events=60, n_contigs=3, contig_len=3_000_000):
    rng = np.random.default_rng(seed)
    recs = []      # (tid, pos, flag, mapq, cigar, qname_id)
    qid = 0
    for ev in range(n_events):
        tid = int(rng.integers(0, n_contigs))
        L = int(rng.integers(50_000, contig_len - 1_200_000))
        kind = rng.choice(["del", "ins", "inv", "dup", "xchr", "far"])
        span = int(np.exp(rng.uniform(np.log(2500), np.log(400_000))))
        k = int(rng.choice([2, 4, 6, 9, 14, 25]))
        for _ in range(k):
            a = int(rng.integers(3000, 15000)); b = int(rng.integers(3000, 15000))
            j1, j2 = int(rng.integers(-6, 7)), int(rng.integers(-6, 7))
            rev = bool(rng.random() < 0.5)
            f = 0x10 if rev else 0
            mq = 60 if rng.random() > 0.08 else int(rng.integers(0, 20))
            lead = int(rng.choice([0, 0, 30, 120]))
            if kind in ("del", "far"):        # primary ends at L, supplementary resumes `span` later, same strand
                recs.append((tid, L - a + j1, f, mq, [(S, lead), (M, a), (S, b)] if lead else [(M, a), (S, b)], qid))
                recs.append((tid, L + span + j2, f | 0x800, mq, [(H, lead + a), (M, b)], qid))
            elif kind == "ins":                # both pieces abut on the reference, `span` extra query bases in between
                recs.append((tid, L - a + j1, f, mq, [(M, a), (S, span + b)], qid))
                recs.append((tid, L + j2, f | 0x800, mq, [(S, a + span), (M, b)], qid))
            elif kind == "dup":                # supplementary maps upstream of the primary's end
                recs.append((tid, L - a + j1, f, mq, [(M, a), (S, b)], qid))
                recs.append((tid, max(1, L - span + j2), f | 0x800, mq, [(S, a), (M, b)], qid))
            elif kind == "inv":                # opposite strand supplementary
                recs.append((tid, L - a + j1, f, mq, [(M, a), (S, b)], qid))
                recs.append((tid, L + span + j2, (f ^ 0x10) | 0x800, mq, [(S, a), (M, b)], qid))
            else:                              # supplementary on another contig
                recs.append((tid, L - a + j1, f, mq, [(M, a), (S, b)], qid))
                recs.append(((tid + 1) % n_contigs, L + j2, f | 0x800, mq, [(S, a), (M, b)], qid))
            if rng.random() < 0.1:             # a second supplementary piece
                recs.append((tid, L + 2 * span + j2, f | 0x800, mq, [(S, a + b // 2), (M, b // 2)], qid))
            qid += 1
    for _ in range(400):                        # background: primaries without supplementary, secondaries, duplicates
        tid = int(rng.integers(0, n_contigs))
        fl = int(rng.choice([0, 0, 0x10, 0x100, 0x400, 0x200, 0x4]))
        recs.append((tid, int(rng.integers(1000, contig_len - 20000)), fl, 60, [(M, int(rng.integers(2000, 15000)))], qid))
        qid += 1
    recs.sort(key=lambda r: (r[0], r[1]))
    reads = Reads.from_cigar_lists([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs], [r[4] for r in recs])
    tid = np.array([r[0] for r in recs], np.int32)
    qn = np.array([r[5] for r in recs], np.uint32)
    return reads, tid, qn, n_contigs


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_split_signatures_match_oracle(ctx, oracle, seed):
    reads, tid, qn, n_contigs = _make_split_shard(seed)
    g_end, g_qs, g_qe = ctx.aln_intervals(reads)                 # scan kernel
    o_end, o_qs, o_qe = oracle.aln_intervals(reads)
    assert np.array_equal(g_end, o_end) and np.array_equal(g_qs, o_qs) and np.array_equal(g_qe, o_qe)
    got = host.split_signatures(ctx, tid, reads.pos, reads.flag, reads.mapq, g_end, g_qs, g_qe, qn, n_contigs)
    exp = oracle.split_signatures(tid, reads.pos, reads.flag, reads.mapq, o_end, o_qs, o_qe, qn)
    assert len(got) == len(exp) and len(got) > 5
    assert got.tobytes() == exp.tobytes()
    types = set(got["sv_type"].tolist())
    assert -1 in types and (3 in types or 2 in types)          # UNKNOWN dummies plus INS and/or INV calls were produced


def test_split_signatures_empty(ctx):
    z = np.zeros(0, np.int32)
    assert len(host.split_signatures(ctx, z, z, z.astype(np.uint16), z.astype(np.uint8), z, z, z, z.astype(np.uint32), 2)) == 0
