"""Two chromosomes in flight on one GPU: two contexts (own streams) behind one gate must give, for every step of every lane, the
calls a single context gives — whatever the interleaving of their kernels."""
import numpy as np
import pytest

import contextsv_amd as cs
from contextsv_amd import host

pytestmark = pytest.mark.gpu


def test_lanes_give_the_single_context_result(ctx):
    syn_a = host.SynthShard(seed=11, chr_len=4_000_000, depth=12.0, tech=0, threads=4)
    syn_b = host.SynthShard(seed=12, chr_len=3_000_000, depth=20.0, tech=1, threads=4)
    ctx2 = cs.Context(0)
    gate = cs.Gate()
    shards = []
    try:
        want = []
        for syn in (syn_a, syn_b):                       # reference: each contig alone on the plain context
            sh = ctx.upload(syn.reads, syn.depth_len)
            calls, _, st = host.process_resident_chromosome(ctx, sh, 0.1, 0.1)
            want.append((calls, st.n_signatures, st.depth_sum))
            sh.free()
        ctx.set_gate(gate); ctx2.set_gate(gate)
        shards = [ctx.upload(syn_a.reads, syn_a.depth_len), ctx2.upload(syn_b.reads, syn_b.depth_len)]
        for steps in ([3, 3], [5, 2]):
            for lane in (0, 1):                          # the hook returns lane 0's last result: run with each contig in lane 0
                order = [lane, 1 - lane]
                got, st, ms, total = host.process_resident_lanes([(ctx, ctx2)[i] for i in order], [shards[i] for i in order],
                                                                 [steps[i] for i in order], 0.1, 0.1)
                exp, n_sig, depth_sum = want[lane]
                assert st.n_signatures == n_sig and st.depth_sum == depth_sum
                assert got.tobytes() == exp.tobytes() and len(got) > 10
                assert total == steps[0] * len(want[0][0]) + steps[1] * len(want[1][0])
    finally:
        ctx.set_gate(None)
        for sh in shards:
            sh.free()
        ctx2.close()
        gate.close()
        syn_a.free(); syn_b.free()
