"""§8f-4: csvgpu_split_order — the iteration order of the reference's per-chromosome `unordered_map<std::string, PrimaryAlignment>`
(src/sv_caller.cpp:137-172 fill, :183-202 erase, :216 / :224 iterate) computed on the device as a chain of per-epoch sorts —
against a REAL std::unordered_map<std::string,int> filled and erased the same way (host.umap_order_check's first output).
Contig sizes sit on both sides of every rehash threshold up to 4e5 nodes; filtered records, supplementary records, names without a
supplementary record and supplementary hashes that belong to no primary are mixed in."""
import numpy as np
import pytest

from contextsv_amd import Reads, host

pytestmark = pytest.mark.gpu


def _contig(rng, tid, n):
    flag = rng.choice([0, 16, 0x800, 0x810, 0x100, 0x400, 0x200, 0x4], n, p=[.42, .42, .03, .03, .03, .03, .02, .02]).astype(np.uint16)
    mapq = rng.choice([60, 60, 60, 20, 19, 0], n).astype(np.uint8)
    names = ["r%d_%d" % (tid, i) for i in range(n)]
    pos = np.sort(rng.integers(0, 10_000_000, n)).astype(np.int32)
    reads = Reads.from_cigar_lists(pos, flag, mapq, [[(0, 100)]] * n) if n < 3000 else \
        Reads(pos, flag, mapq, np.arange(n + 1, dtype=np.uint64), np.full(n, (100 << 4), np.uint32))
    return reads, names


@pytest.fixture(params=[(None, None), (0, None), (1, None), (2, None), (3, None), (0, 0), (2, 0)],
                ids=["auto", "chain", "tail1", "tail2", "tail3", "chain-nosmall", "tail2-nosmall"])
def tail(request):
    """CSV_SPLIT_TAIL: how many of every contig's last epochs are ordered for the survivors only (splitorder.hip); None = the library's own
    choice from the number of nodes per supplementary record, 0 = the full chain of sorts. CSV_SPLIT_SMALL=0: the first epochs through the
    chain's sorts as well instead of the one-launch LDS kernel."""
    import os
    old = {k: os.environ.pop(k, None) for k in ("CSV_SPLIT_TAIL", "CSV_SPLIT_SMALL")}
    for k, v in zip(("CSV_SPLIT_TAIL", "CSV_SPLIT_SMALL"), request.param):
        if v is not None:
            os.environ[k] = str(v)
    yield request.param
    for k, v in old.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


@pytest.mark.parametrize("supp_frac", [0.1, 0.01, 0.6])
@pytest.mark.parametrize("sizes", [[0, 1, 2, 12, 13, 14, 28, 29, 30, 58, 59, 60, 126, 127, 128, 257, 541, 542, 1109, 2357, 2358, 5087, 5088],
                                   [10273, 10274, 20753, 42043, 42044, 85229, 85230, 3],
                                   [172933, 172934, 351061, 351062, 400_000]])
def test_split_order_against_a_real_unordered_map(ctx, sizes, supp_frac, tail):
    if supp_frac != 0.1 and len(sizes) != 8:
        pytest.skip("one size set per extra supplementary fraction")
    rng = np.random.default_rng(len(sizes))
    contigs = [_contig(rng, t, n) for t, n in enumerate(sizes)]
    shards = []
    supp_names = []
    per = []
    try:
        for t, (reads, names) in enumerate(contigs):
            h = host.string_hashes(names)
            sh = ctx.upload(reads, 10_000_001)
            sh.set_qname_hash(h)
            shards.append(sh)
            ok = ((reads.flag & (0x100 | 0x4 | 0x400 | 0x200)) == 0) & (reads.mapq >= 20)
            prim = np.flatnonzero(ok & ((reads.flag & 0x800) == 0))
            # supplementary names: a fraction of the primaries' names (+ a few names that belong to no primary)
            has_supp = rng.random(len(prim)) < supp_frac
            per.append((names, prim, has_supp))
            supp_names += [names[i] for i in prim[has_supp]] + ["ghost%d_%d" % (t, k) for k in range(3)]
        supp_hash = np.unique(host.string_hashes(supp_names))
        got = ctx.split_order(shards, 20, supp_hash)
        two = ctx.split_order_two_calls(shards, 20, supp_hash)             # _begin before the supplementary records are known, _finish (with a retry)
        assert all(np.array_equal(a, b) for a, b in zip(got, two))
        for t, (names, prim, has_supp) in enumerate(per):
            keys = [names[i] for i in prim]
            order_real, order_emu, buckets = host.umap_order_check(keys, (~has_supp).astype(np.uint8))
            assert np.array_equal(order_real, order_emu)
            exp = prim[order_real].astype(np.uint32)
            assert np.array_equal(got[t], exp), (t, len(names), len(exp))
    finally:
        for sh in shards:
            sh.free()


def test_split_order_argument_checks(ctx):
    reads = Reads.from_cigar_lists([1, 2], [0, 0], [60, 60], [[(0, 10)], [(0, 10)]])
    sh = ctx.upload(reads, 100)
    try:
        with pytest.raises(Exception):                      # no query-name hashes attached
            ctx.split_order([sh], 20, np.array([1, 2], np.uint64))
        sh.set_qname_hash(np.array([5, 9], np.uint64))
        with pytest.raises(Exception):                      # supplementary hashes not sorted
            ctx.split_order([sh], 20, np.array([9, 5], np.uint64))
        assert [x.tolist() for x in ctx.split_order([sh], 20, np.array([], np.uint64))] == [[]]
        got = ctx.split_order([sh], 20, np.array([5, 9], np.uint64))
        assert sorted(got[0].tolist()) == [0, 1]
    finally:
        sh.free()


@pytest.mark.parametrize("supp_to", ["same", "any", "none"])
@pytest.mark.parametrize("sizes", [[0, 1, 2, 13, 14, 59, 60, 541, 542, 2357, 5087, 5088], [10273, 20753, 42044, 85229, 3], [172934, 351061, 400_000]])
def test_split_order_self_takes_the_supplementary_hashes_from_the_shards(ctx, sizes, supp_to, tail):
    """csvgpu_split_order_begin_self: the call holds every contig of the run, so the supplementary records' name hashes are the shards' own
    (flag 0x800 set, same filter as sv_caller.cpp:145) and everything is queued by _begin. Expected = csvgpu_split_order given those hashes
    (itself checked against the real container above) and, per contig, the real container again. Supplementary records carry the name of a
    primary on the same contig, on any contig, or of nobody; some fail the mapq / flag filter and must not count."""
    if supp_to != "same" and len(sizes) != 5:
        pytest.skip("one size set per extra case")
    rng = np.random.default_rng(len(sizes) + 7)
    contigs = [_contig(rng, t, n) for t, n in enumerate(sizes)]
    hashes = [host.string_hashes(names) if len(names) else np.zeros(0, np.uint64) for _, names in contigs]
    ok = [((r.flag & (0x100 | 0x4 | 0x400 | 0x200)) == 0) & (r.mapq >= 20) for r, _ in contigs]
    prim = [np.flatnonzero(o & ((r.flag & 0x800) == 0)) for o, (r, _) in zip(ok, contigs)]
    all_prim_hashes = np.concatenate([h[p] for h, p in zip(hashes, prim)]) if sizes else np.zeros(0, np.uint64)
    shards = []
    try:
        supp_set = []
        for t, (reads, names) in enumerate(contigs):
            h = hashes[t].copy()
            supp = np.flatnonzero((reads.flag & 0x800) != 0)                      # (also the ones that fail the filter)
            for i in supp:
                if supp_to == "none" or rng.random() < 0.2:
                    h[i] = np.uint64(rng.integers(1, 1 << 62))                    # a name nobody else has
                elif supp_to == "same" and len(prim[t]):
                    h[i] = hashes[t][rng.choice(prim[t])]
                elif len(all_prim_hashes):
                    h[i] = rng.choice(all_prim_hashes)
            supp_set.append(h[supp][ok[t][supp]])
            sh = ctx.upload(reads, 10_000_001)
            sh.set_qname_hash(h)
            shards.append(sh)
        supp_hash = np.unique(np.concatenate(supp_set)) if supp_set else np.zeros(0, np.uint64)
        exp = ctx.split_order(shards, 20, supp_hash)
        got = ctx.split_order_self(shards, 20)
        assert all(np.array_equal(a, b) for a, b in zip(got, exp)), [(len(a), len(b)) for a, b in zip(got, exp)]
        for t, (reads, names) in enumerate(contigs):
            keys = [names[i] for i in prim[t]]
            has = np.isin(hashes[t][prim[t]], supp_hash)
            order_real, order_emu, buckets = host.umap_order_check(keys, (~has).astype(np.uint8))
            assert np.array_equal(got[t], prim[t][order_real].astype(np.uint32)), (t, len(names))
    finally:
        for sh in shards:
            sh.free()


def test_split_order_self_with_more_supplementary_records_than_primaries(ctx):
    """More supplementary records than nodes (the sort's workspace is sized for the larger of the two), several of them per name."""
    rng = np.random.default_rng(5)
    shards, exp_sets = [], []
    try:
        for t, n in enumerate([3000, 40, 7000]):
            flag = rng.choice([0, 16, 0x800, 0x810], n, p=[.05, .05, .45, .45]).astype(np.uint16)
            mapq = np.full(n, 60, np.uint8)
            pos = np.sort(rng.integers(0, 1_000_000, n)).astype(np.int32)
            reads = Reads(pos, flag, mapq, np.arange(n + 1, dtype=np.uint64), np.full(n, (100 << 4), np.uint32))
            names = ["q%d_%d" % (t, i) for i in range(n)]
            h = host.string_hashes(names)
            prim = np.flatnonzero((flag & 0x800) == 0)
            supp = np.flatnonzero((flag & 0x800) != 0)
            if len(prim):
                h[supp] = h[rng.choice(prim[: max(1, len(prim) // 2)], len(supp))]       # every supplementary record carries a primary's name, many share one
            sh = ctx.upload(reads, 1_000_001)
            sh.set_qname_hash(h)
            shards.append(sh)
            exp_sets.append((h, prim, supp))
        supp_hash = np.unique(np.concatenate([h[s] for h, _, s in exp_sets]))
        exp = ctx.split_order(shards, 20, supp_hash)
        got = ctx.split_order_self(shards, 20)
        assert all(np.array_equal(a, b) for a, b in zip(got, exp))
        for (h, prim, _), g in zip(exp_sets, got):
            assert sorted(g.tolist()) == sorted(prim[np.isin(h[prim], supp_hash)].tolist())
    finally:
        for sh in shards:
            sh.free()
