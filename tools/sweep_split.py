"""One-off sweep: split-read signatures (device intervals + batched DBSCAN1D + host grouping) against the oracle, many seeds and
denser / odder groupings than the suite's."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import contextsv_amd as cs
from contextsv_amd import host
import oracle_lib
from test_gpu_split import _make_split_shard
orc = oracle_lib.load_oracle()
ctx = cs.Context(0)
host.set_context(ctx)
bad = 0
for seed in range(10, 50):
    rng = np.random.default_rng(seed)
    reads, tid, qn, nc = _make_split_shard(seed, n_events=int(rng.choice([1, 5, 60, 150])), n_contigs=int(rng.choice([1, 2, 3, 5])),
                                           contig_len=int(rng.choice([1_500_000, 3_000_000])))
    if seed % 3 == 0:                       # collapse query names: many supplementaries per name
        qn = (qn % max(1, int(qn.max()) // 7 + 1)).astype(np.uint32)
    g = ctx.aln_intervals(reads); o = orc.aln_intervals(reads)
    got = host.split_signatures(ctx, tid, reads.pos, reads.flag, reads.mapq, g[0], g[1], g[2], qn, nc)
    exp = orc.split_signatures(tid, reads.pos, reads.flag, reads.mapq, o[0], o[1], o[2], qn)
    if got.tobytes() != exp.tobytes():
        bad += 1
        print('SPLIT FAIL', seed, len(got), len(exp))
print('done, failures:', bad)
