"""What one call of the lanes driver costs besides its steps: total time of process_resident_lanes for growing step counts on three lanes
(marginal cost per step vs the fixed part: thread start-up, pipeline fill, the last chromosome's merge), next to the one-lane driver."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import contextsv_amd as cs
from contextsv_amd import host
ctxs = [cs.Context(0) for _ in range(3)]
host.set_context(ctxs[0])
syn = host.SynthShard(0x5EED0000 + 1022, 50818468, 30.0, 0, 16)
shards = [c.upload(syn.reads, syn.depth_len) for c in ctxs]
gate = cs.Gate()
for c in ctxs: c.set_gate(gate)
host.process_resident_lanes(ctxs, shards, [3, 3, 3], 0.1, 0.1)
for steps in ([1, 0, 0], [1, 1, 0], [1, 1, 1], [2, 2, 1], [2, 2, 2], [4, 3, 3], [7, 7, 6], [17, 17, 16]):
    ts = []
    for rep in range(5):
        for c in ctxs: c.synchronize()
        t0 = time.perf_counter()
        host.process_resident_lanes(ctxs, shards, steps, 0.1, 0.1)
        for c in ctxs: c.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(steps, "total ms min %.3f med %.3f" % (min(ts), sorted(ts)[2]), "per step %.3f" % (min(ts) / sum(steps)), flush=True)
# one lane for comparison
ctxs[0].set_gate(None)
for n in (1, 2, 5, 20):
    ts = []
    for rep in range(5):
        ctxs[0].synchronize(); t0 = time.perf_counter()
        host.process_resident_pipelined(ctxs[0], shards[0], n, 0.1, 0.1)
        ctxs[0].synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("one lane", n, "total ms min %.3f" % min(ts), "per step %.3f" % (min(ts) / n), flush=True)
