"""BASELINE.json configs[1] at full size (chr22, 30x synthetic ONT: ~1.3e5 reads, ~1.45e8 CIGAR ops) through the
device pipeline and the C++ host mirror, checked against (1) the oracle where it finishes in seconds (scan,
depth, the DEL clustering), (2) an independent vectorised numpy restatement of the reference cursor rules
for every signature, and (3) size-independent properties of the clustering and of the merged call set."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHR22 = 50818468
REF_OPS = np.array([1, 0, 1, 1, 0, 0, 0, 1, 1] + [0] * 7, bool)
QRY_OPS = np.array([1, 1, 0, 0, 1, 0, 0, 1, 1] + [0] * 7, bool)


@pytest.fixture(scope="module")
def chr22(ctx):
    from contextsv_amd import host
    host.set_context(ctx)
    syn = host.SynthShard(0x5EED0000 + 1000 + 22, CHR22, 30.0, 0, 8)
    sh = ctx.upload(syn.reads, syn.depth_len)
    res = sh.pipeline(eps=0.1, min_pts_pct=0.1)
    out = sh.fetch(res, want_depth=True)
    yield syn, sh, res, out
    sh.free()
    syn.free()


def test_fullsize_signatures_numpy(chr22):
    syn, sh, res, out = chr22
    r = syn.reads
    op, ln = (r.cigar & 15).astype(np.int64), (r.cigar >> 4).astype(np.int64)
    counts = np.diff(r.cigar_off.astype(np.int64))
    rid = np.repeat(np.arange(r.n_reads), counts)
    c0 = r.cigar_off[:-1].astype(np.int64)
    ref_c = np.concatenate([[0], np.cumsum(np.where(REF_OPS[op], ln, 0))])
    qry_c = np.concatenate([[0], np.cumsum(np.where(QRY_OPS[op], ln, 0))])
    pos_at = r.pos.astype(np.int64)[rid] + (ref_c[:-1] - ref_c[c0][rid])          # reference `pos` before the op
    q_at = qry_c[:-1] - qry_c[c0][rid]
    passes = ((r.flag & (0x100 | 0x4 | 0x400 | 0x200 | 0x800)) == 0) & (r.mapq >= 20)
    emit = passes[rid] & (ln >= 50) & ((op == 1) | (op == 2) | ((op == 4) & (pos_at + 1 < syn.depth_len)))
    # soft clips skipped at the contig end also skip their query-cursor update (sv_caller.cpp:602-604)
    skipped = passes[rid] & (op == 4) & (ln >= 50) & (pos_at + 1 >= syn.depth_len)
    assert skipped.any()                      # the shard does exercise the quirk
    sk_c = np.concatenate([[0], np.cumsum(np.where(skipped, ln, 0))])
    q_at = q_at - (sk_c[:-1] - sk_c[c0][rid])
    start = pos_at[emit] + 1
    end = start + ln[emit] - 1
    kind = np.where(op[emit] == 1, 0, np.where(op[emit] == 2, 1, 2))
    exp = np.stack([start, end, rid[emit], q_at[emit], kind], 1)
    got_all = np.concatenate([out["sig_del"], out["sig_ins"]])
    got = np.stack([got_all["start"], got_all["end"], got_all["read"], got_all["qpos_kind"] >> 2, got_all["qpos_kind"] & 3], 1).astype(np.int64)
    assert res.n_sig == len(exp) == len(got)
    assert np.array_equal(exp[np.lexsort(exp.T[::-1])], got[np.lexsort(got.T[::-1])])
    # vector order: ascending (start,end) inside each type
    for part in (out["sig_del"], out["sig_ins"]):
        key = part["start"].astype(np.int64) << 32 | part["end"]
        assert (np.diff(key) >= 0).all()


def test_fullsize_scan_depth_oracle(chr22, oracle):
    syn, sh, res, out = chr22
    sig = oracle.cigar_scan(syn.reads, syn.depth_len)
    kind = sig["qpos_kind"] & 3
    for name, sel in (("sig_del", kind == 1), ("sig_ins", kind != 1)):
        assert out[name].tobytes() == sig[sel].tobytes()          # bit-identical records in identical order
    d, s, nz = oracle.depth(syn.reads, syn.depth_len)
    assert np.array_equal(out["depth"], d) and (res.depth_sum, res.depth_nonzero) == (s, nz)
    assert res.mean_cov == s / nz and res.min_pts == int(np.ceil(s / nz * 0.1))
    for g, o in zip((out["ref_end"], out["q_start"], out["q_end"]), oracle.aln_intervals(syn.reads)):
        assert np.array_equal(g, o)
    # depth conservation: sum of depth == number of M/=/X bases of the counted reads that fall inside the contig
    assert res.depth_sum == int(out["depth"].sum(dtype=np.uint64))


def test_fullsize_dbscan_and_merge(chr22, oracle, ctx):
    from contextsv_amd import host
    syn, sh, res, out = chr22
    dels, inss = out["sig_del"], out["sig_ins"]
    # DEL set is small enough for the O(n^2) oracle
    assert np.array_equal(out["label_del"], oracle.dbscan_iv(dels["start"], dels["end"], 0.1, res.min_pts))
    lab = out["label_ins"]
    # label alphabet: noise (-2) or 0..k-1 with every id in use
    k = int(lab.max()) + 1
    assert set(np.unique(lab).tolist()) <= set(range(k)) | {-2} and len(np.unique(lab[lab >= 0])) == k
    # windowed clustering must equal the seam called on the same intervals in caller order (same kernels, other entry)
    assert np.array_equal(ctx.dbscan_iv(inss["start"], inss["end"], 0.1, res.min_pts), lab)
    # every member of a cluster overlaps its cluster's hull; two members of one cluster are chained by eps-neighbours:
    # check the chain property on a sample of clusters with the exact metric
    rng = np.random.default_rng(0)
    for c in rng.choice(k, min(k, 40), replace=False):
        m = inss[lab == c]
        s, e = m["start"].astype(np.int64), m["end"].astype(np.int64)
        ov = np.maximum(0, np.minimum(e[:, None], e[None, :]) - np.maximum(s[:, None], s[None, :]))
        ln = (e - s).astype(np.float64)
        nb = (1.0 - np.minimum(ov / ln[:, None], ov / ln[None, :])) <= 0.1
        reach = np.eye(len(m), dtype=bool)
        for _ in range(len(m)):
            new = (reach.astype(np.int32) @ nb.astype(np.int32)) > 0
            if (new == reach).all():
                break
            reach = new
        assert reach.all()
    # full seam: oracle INS labels on the whole set (a few seconds of O(n^2)) and the merged call set
    olab = oracle.dbscan_iv(inss["start"], inss["end"], 0.1, res.min_pts)
    assert np.array_equal(lab, olab)
    calls, tags, st = host.process_resident_chromosome(ctx, sh, 0.1, 0.1)
    sig = np.concatenate([dels, inss])
    oc = np.zeros(len(sig), __import__("oracle_lib").CALL_DTYPE)
    oc["start"], oc["end"], oc["sv_type"], oc["id"] = sig["start"], sig["end"], np.where((sig["qpos_kind"] & 3) == 1, 0, 3), np.arange(len(sig))
    labels_by_type = {0: out["label_del"], 3: lab}
    om = oracle.merge_svs(oc, 0.1, res.min_pts, False, label_fn=lambda s, e, eps, mp: labels_by_type[0] if len(s) == len(dels) else labels_by_type[3])
    assert len(calls) == len(om) == st.n_calls
    for f in ("start", "end", "sv_type", "cluster_size"):
        assert np.array_equal(calls[f], om[f]), f
    # evidence flags of the merged calls come from the chosen member
    kinds = sig["qpos_kind"][om["id"]] & 3
    assert np.array_equal(calls["aln_flags"], np.where(kinds == 0, 1, np.where(kinds == 1, 2, 4)))


def test_pipelined_driver_equals_sequential(chr22, ctx):
    """SVCaller::processResidentChromosomesPipelined (merge thread overlapping the next device chain) returns the same
    merged calls and statistics as the strictly sequential path, step after step."""
    from contextsv_amd import host
    syn, sh, res, out = chr22
    calls, tags, st = host.process_resident_chromosome(ctx, sh, 0.1, 0.1)
    pc, pt, pst, ms, tot = host.process_resident_pipelined(ctx, sh, 5, 0.1, 0.1)
    assert calls.tobytes() == pc.tobytes() and tags.tobytes() == pt.tobytes()
    assert tot == 5 * len(calls) and (pst.n_signatures, pst.min_pts, pst.depth_sum) == (st.n_signatures, st.min_pts, st.depth_sum)
