"""How long does the split-read pass take at whole-genome scale? (device alignment intervals + batched DBSCAN1D + host grouping)"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import contextsv_amd as cs
from contextsv_amd import host
from test_gpu_split import _make_split_shard
ctx = cs.Context(0)
host.set_context(ctx)
for n_events in (2000, 20000):
    t0 = time.perf_counter()
    reads, tid, qn, nc = _make_split_shard(1, n_events=n_events, n_contigs=24, contig_len=100_000_000)
    t1 = time.perf_counter()
    g = ctx.aln_intervals(reads)
    t2 = time.perf_counter()
    got = host.split_signatures(ctx, tid, reads.pos, reads.flag, reads.mapq, g[0], g[1], g[2], qn, nc)
    t3 = time.perf_counter()
    print(n_events, 'events', reads.n_reads, 'records ->', len(got), 'calls; gen %.1fs, intervals %.3fs, split pass %.3fs' % (t1 - t0, t2 - t1, t3 - t2), flush=True)
