"""The hand-built known answers and the reference-generated golden vectors, through the C-ABI on the GPU."""
import json
import os

import numpy as np
import pytest

from kat_cases import KAT_DEPTH, KAT_SCAN, kat_reads

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_gpu_scan_known_answers(ctx):
    for name, case in KAT_SCAN.items():
        reads = kat_reads(case)
        sig = ctx.cigar_scan(reads, case["depth_len"], case.get("min_oplen", 50), case.get("min_mapq", 20), capacity=64)
        got = [(int(s["start"]), int(s["end"]), int(s["read"]), int(s["qpos_kind"] >> 2), int(s["qpos_kind"] & 3)) for s in sig]
        assert got == case["expect"], name
        if "intervals" in case:
            re_, qs, qe = ctx.aln_intervals(reads)
            assert list(zip(re_.tolist(), qs.tolist(), qe.tolist())) == case["intervals"], name


def test_gpu_depth_known_answers(ctx):
    for name, case in KAT_DEPTH.items():
        reads = kat_reads(case)
        d, s, nz = ctx.depth(reads, case["depth_len"])
        assert d.tolist() == case["depth"], name
        assert (s, nz) == (sum(case["depth"]), sum(1 for x in case["depth"] if x > 0)), name


def test_gpu_dbscan_iv_golden(ctx):
    with open(os.path.join(G, "dbscan_iv.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        s, e = np.asarray(c["start"], np.uint32), np.asarray(c["end"], np.uint32)
        assert ctx.dbscan_iv(s, e, c["eps"], c["min_pts"]).tolist() == c["labels"], (c["seed"], c["eps"], c["min_pts"])


def test_gpu_dbscan_1d_golden(ctx):
    from contextsv_amd.api import largest_cluster
    with open(os.path.join(G, "dbscan_1d.json")) as f:
        cases = json.load(f)["cases"]
    # one batched call per (eps, min_pts) group — the way the split-read path uses the kernel
    groups = {}
    for c in cases:
        groups.setdefault((c["eps"], c["min_pts"]), []).append(c)
    for (eps, mp), cs in groups.items():
        off = np.zeros(len(cs) + 1, np.uint64)
        off[1:] = np.cumsum([len(c["points"]) for c in cs])
        pts = np.concatenate([np.asarray(c["points"], np.int32) for c in cs]) if off[-1] else np.zeros(0, np.int32)
        lab = ctx.dbscan_1d(pts, off, eps, mp)
        for k, c in enumerate(cs):
            a, b = int(off[k]), int(off[k + 1])
            assert lab[a:b].tolist() == c["labels"]
            assert largest_cluster(pts[a:b], lab[a:b]).tolist() == c["largest"]
