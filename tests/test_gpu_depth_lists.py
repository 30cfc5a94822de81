"""The depth tiles' work lists (kernels/depth.hip): depth_items_kernel builds the first 512 entries of every tile's list ahead of the tile
kernel, later batches of a deep pile are still built inside it, and wrapped (unpadded) arrays take the tile kernel's bounds-checked loads.
All of them must give the oracle's depth map."""
import numpy as np
import pytest

from contextsv_amd import Reads

pytestmark = pytest.mark.gpu
M, I, D, N, S, H, P, EQ, X = range(9)


def _pile(seed, n_reads, depth_len, sorted_pos=True, long_every=0):
    rng = np.random.default_rng(seed)
    pos = rng.integers(0, depth_len, n_reads)
    if sorted_pos:
        pos.sort()
    flag = rng.choice([0, 0, 0, 16, 256, 1024], n_reads).astype(np.uint16)
    mapq = np.full(n_reads, 60, np.uint8)
    cig = []
    for r in range(n_reads):
        k = int(rng.choice([1, 3, 30, 200, 900])) if not (long_every and r % long_every == 0) else 5000
        ops = [(S, int(rng.integers(1, 80)))] if rng.random() < 0.3 else []
        for _ in range(k):
            ops.append((int(rng.choice([M, M, M, M, I, D, D, N, EQ, X])), int(rng.choice([0, 1, 2, 7, 30, 60]))))
        cig.append(ops)
    return Reads.from_cigar_lists(pos, flag, mapq, cig)


@pytest.mark.parametrize("seed,n_reads,depth_len,sorted_pos,long_every", [
    (11, 1500, 20_000, True, 0),        # ~1500 candidates per tile: three batches, the first from the list kernel
    (12, 1500, 20_000, False, 0),       # the same through the unsorted path (ranges from the prefix maximum, reads through `ord`)
    (13, 700, 40_000, True, 50),        # reads of 5000 ops across all three tiles: walks of twenty chunks, trimmed at the tile edge
    (14, 513, 16_384, True, 0),         # one entry more than a batch
])
def test_deep_piles_and_long_walks(ctx, oracle, seed, n_reads, depth_len, sorted_pos, long_every):
    reads = _pile(seed, n_reads, depth_len, sorted_pos, long_every)
    od, os_, onz = oracle.depth(reads, depth_len)
    d, s, nz = ctx.depth(reads, depth_len)                      # arena copy of the arrays: no padding, every tile builds its own list
    assert np.array_equal(d, od) and (s, nz) == (os_, onz)
    sh = ctx.upload(reads, depth_len)                            # padded upload + list kernel
    try:
        res = sh.pipeline(eps=0.1, min_pts_pct=0.1)
        out = sh.fetch(res, want_depth=True)
        assert np.array_equal(out["depth"], od) and (res.depth_sum, res.depth_nonzero) == (os_, onz)
    finally:
        sh.free()
