"""SVCaller::run's pass order (src/sv_caller.cpp:747-946) composed from the oracle's pieces in Python: the expected result of a whole
run over in-memory contigs. Test infrastructure (used by test_gpu_e2e.py and test_gpu_genome.py)."""
import numpy as np

import oracle_lib
from contextsv_amd import host


def full_calls(start, end, sv_type, cluster, flags, aln_offset=None):
    c = host.make_calls(start, end, sv_type, cluster)
    c["aln_flags"] = flags
    if aln_offset is not None:
        c["aln_offset"] = aln_offset
    return c


def orc_merge(oracle, full, eps, min_pts, keep_noise):
    small = np.zeros(len(full), oracle_lib.CALL_DTYPE)
    for f in ("start", "end", "sv_type", "cluster_size", "hmm_likelihood"):
        small[f] = full[f]
    small["id"] = np.arange(len(full))
    m = oracle.merge_svs(small, eps, min_pts, keep_noise)
    out = full[m["id"]].copy()
    out["cluster_size"] = m["cluster_size"]
    return out




def oracle_run(oracle, contigs, hmm, eps=0.1, min_pts_pct=0.1, split=True, cigar_cn=True):
    """contigs: list of {reads, depth_len, qname_id, snps}; query name of a record = "r<qname_id>".
    -> (calls[host.CALL_DTYPE], contig index per call, depth maps, mean coverages)"""
    cigar_calls, depths, means = [], [], []
    for c in contigs:
        r = c["reads"]
        sig = oracle.cigar_scan(r, c["depth_len"])
        depth, s, nz = oracle.depth(r, c["depth_len"])
        mean = s / nz if nz else 0.0
        min_pts = int(np.ceil(mean * min_pts_pct)) if min_pts_pct > 0 else 5
        kind = sig["qpos_kind"] & 3
        full = full_calls(sig["start"], sig["end"], np.where(kind == 1, 0, 3), 0, np.where(kind == 0, 1, np.where(kind == 1, 2, 4)))
        full = orc_merge(oracle, full, eps, min_pts, False)
        if cigar_cn:
            full = oracle.cn_prediction(depth, full, hmm, mean, c["snps"], split=False)
        cigar_calls.append(full); depths.append(depth); means.append(mean)
    tid = np.concatenate([np.full(c["reads"].n_reads, t, np.int32) for t, c in enumerate(contigs)])
    cat = lambda f: np.concatenate([getattr(c["reads"], f) for c in contigs])
    exp, exp_tid = [], []
    sp = None
    if split:
        iv = [oracle.aln_intervals(c["reads"]) for c in contigs]
        sp = oracle.split_signatures(tid, cat("pos"), cat("flag"), cat("mapq"), np.concatenate([x[0] for x in iv]), np.concatenate([x[1] for x in iv]),
                                     np.concatenate([x[2] for x in iv]), np.concatenate([c["qname_id"] for c in contigs]))
    for t, c in enumerate(contigs):
        whole = cigar_calls[t]
        if split:
            st = sp[sp["tid"] == t]
            split_full = full_calls(st["start"], st["end"], st["sv_type"], st["cluster_size"], st["aln_flags"], st["aln_offset"])
            if len(split_full):
                split_full = oracle.cn_prediction(depths[t], split_full, hmm, means[t], c["snps"], split=True)
                split_full = orc_merge(oracle, split_full, 0.1, 2, True)
            whole = np.concatenate([cigar_calls[t], split_full])
        whole = orc_merge(oracle, whole, 0.1, 2, True)
        exp.append(whole); exp_tid.append(np.full(len(whole), t, np.int32))
    return np.concatenate(exp), np.concatenate(exp_tid), depths, means
