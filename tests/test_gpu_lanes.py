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
        for c in (ctx, ctx2):                            # as bench.py runs: HIP-event timers around the scan + depth pair on the gate's stream
            c.timing_enable(2); c.timing_reset()
        for steps in ([3, 3], [5, 2]):
            for lane in (0, 1):                          # the hook returns lane 0's last result: run with each contig in lane 0
                order = [lane, 1 - lane]
                got, st, ms, total = host.process_resident_lanes([(ctx, ctx2)[i] for i in order], [shards[i] for i in order],
                                                                 [steps[i] for i in order], 0.1, 0.1)
                exp, n_sig, depth_sum = want[lane]
                assert st.n_signatures == n_sig and st.depth_sum == depth_sum
                assert got.tobytes() == exp.tobytes() and len(got) > 10
                assert total == steps[0] * len(want[0][0]) + steps[1] * len(want[1][0])
        for c, n_steps in ((ctx, 2 * (3 + 5)), (ctx2, 2 * (3 + 2))):
            t = c.timing()
            timed = (n_steps + 3) // 4                   # level 2 behind a gate: every fourth pair carries the timers
            assert t["cigar_scan"][1] == timed and t["depth"][1] == timed and t["cigar_scan"][0] > 0 and t["depth"][0] > 0
    finally:
        for c in (ctx, ctx2):
            c.timing_enable(0)
        ctx.set_gate(None)
        for sh in shards:
            sh.free()
        ctx2.close()
        gate.close()
        syn_a.free(); syn_b.free()


def test_pipelined_driver_grows_result_buffers_and_matches_single_calls(ctx):
    """More signatures than the driver's initial page-locked buffers hold (first contig of that size: CSV_ECAPACITY -> grow ->
    csvgpu_chr_fetch while the next job's scan is already queued), several steps on the same shard, against the one-shot call."""
    M, D = 0, 2
    n_reads, per = 3000, 40                                # 120 000 signatures > the initial 65 536 + slack
    pos = np.sort(np.random.default_rng(4).integers(0, 150_000, n_reads))
    cig = [[op for k in range(per) for op in ((M, 90 + (r + k) % 7), (D, 50 + (r * 7 + k) % 40))] + [(M, 50)] for r in range(n_reads)]
    reads = cs.Reads.from_cigar_lists(pos, np.zeros(n_reads, np.uint16), np.full(n_reads, 60, np.uint8), cig)
    sh = ctx.upload(reads, 170_000)
    try:
        want, _, st1 = host.process_resident_chromosome(ctx, sh, 0.1, 0.1)
        got, _, st, ms, total = host.process_resident_pipelined(ctx, sh, 5, 0.1, 0.1)
        assert st.n_signatures == st1.n_signatures == n_reads * per
        assert got.tobytes() == want.tobytes() and total == 5 * len(want) and len(want) > 0
    finally:
        sh.free()


def test_job_calls_out_of_order(ctx):
    import ctypes as C
    from contextsv_amd._lib import csv_chr_result
    syn = host.SynthShard(seed=3, chr_len=500_000, depth=5.0, tech=0, threads=2)
    sh = ctx.upload(syn.reads, syn.depth_len)
    lib = ctx.lib
    try:
        job = lib.csvgpu_chr_job_begin(ctx.h, sh.h, 50, 20, 0.1)
        assert job
        res = csv_chr_result()
        assert lib.csvgpu_chr_job_end(ctx.h, job, C.byref(res)) == cs._lib.CSV_EINVAL          # end before cluster: refused, job released
        job = lib.csvgpu_chr_job_begin(ctx.h, sh.h, 50, 20, 0.1)
        assert lib.csvgpu_chr_job_cluster(ctx.h, job, 1.5, None, None, 0) == cs._lib.CSV_EINVAL   # eps outside [0, 1)
        assert lib.csvgpu_chr_job_cluster(ctx.h, job, 0.1, None, None, 0) == 0
        assert lib.csvgpu_chr_job_cluster(ctx.h, job, 0.1, None, None, 0) == cs._lib.CSV_EINVAL   # twice
        assert lib.csvgpu_chr_job_end(ctx.h, job, C.byref(res)) == 0 and res.n_sig > 0
        ref = sh.pipeline()
        assert (ref.n_sig, ref.depth_sum, ref.min_pts) == (res.n_sig, res.depth_sum, res.min_pts)
    finally:
        sh.free()
        syn.free()


def test_wrapped_device_arrays_run_the_same_pipeline_and_bad_offsets_are_refused(ctx):
    """csvgpu_shard_wrap_dev: arrays already in HBM, no copy. Same pipeline results as the uploaded shard (the first job on a wrapped
    shard does not know yet whether it is coordinate-sorted: the other branch of the job), and the offsets are validated by a
    kernel — a table that leaves the word array never reaches the scan."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")                         # the runtime the library itself is linked against
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    held = []

    def to_device(a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), max(a.nbytes, 16)) == 0
        held.append(p)
        assert hip.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0          # hipMemcpyHostToDevice
        return p.value

    syn = host.SynthShard(seed=21, chr_len=2_000_000, depth=15.0, tech=0, threads=4)
    r = syn.reads
    up = ctx.upload(r, syn.depth_len)
    wr = None
    try:
        want = up.pipeline(); w = up.fetch(want, want_depth=True)
        d_pos, d_flag, d_mapq, d_off, d_cig = (to_device(a) for a in (r.pos, r.flag, r.mapq, r.cigar_off, r.cigar))
        wr = ctx.wrap_device_ptrs(r.n_reads, r.n_cigar, d_pos, d_flag, d_mapq, d_off, d_cig, syn.depth_len)
        for _ in range(2):                                 # second pass: sortedness known, depth pass queued behind the scan
            got = wr.pipeline(); g = wr.fetch(got, want_depth=True)
            assert (got.n_sig, got.n_del, got.depth_sum, got.depth_nonzero, got.min_pts) == (want.n_sig, want.n_del, want.depth_sum, want.depth_nonzero, want.min_pts)
            for k in w:
                assert np.array_equal(w[k], g[k]), k
        for breakage in ("dip", "beyond"):
            off = r.cigar_off.copy()
            if breakage == "dip":
                off[5] = off[7] + np.uint64(1)
            else:
                off[1:] += np.uint64(1 << 20)
            with pytest.raises(cs.CsvError):
                ctx.wrap_device_ptrs(r.n_reads, r.n_cigar, d_pos, d_flag, d_mapq, to_device(off), d_cig, syn.depth_len)
    finally:
        if wr is not None:
            wr.free()
        up.free()
        ctx.synchronize()
        for p in held:
            hip.hipFree(p)
        syn.free()


def test_three_lanes_as_the_benchmark_runs_them(ctx):
    """bench.py's default: three contexts behind one gate, the same contig resident once per lane, unequal step counts — every lane's
    last result equals the single-context result, and the per-lane call totals add up."""
    syn = host.SynthShard(seed=31, chr_len=2_500_000, depth=25.0, tech=0, threads=4)
    ctxs = [ctx, cs.Context(0), cs.Context(0)]
    gate = cs.Gate()
    shards = []
    try:
        sh = ctx.upload(syn.reads, syn.depth_len)
        want, _, st1 = host.process_resident_chromosome(ctx, sh, 0.1, 0.1)
        sh.free()
        for c in ctxs:
            c.set_gate(gate)
            c.timing_enable(2); c.timing_reset()
        shards = [c.upload(syn.reads, syn.depth_len) for c in ctxs]
        for steps in ([4, 4, 4], [7, 1, 3], [1, 5, 2]):
            got, st, ms, total = host.process_resident_lanes(ctxs, shards, steps, 0.1, 0.1)
            assert st.n_signatures == st1.n_signatures and st.depth_sum == st1.depth_sum
            assert got.tobytes() == want.tobytes() and total == sum(steps) * len(want) and len(want) > 10
        assert [c.timing()["depth"][1] for c in ctxs] == [(n + 3) // 4 for n in (4 + 7 + 1, 4 + 1 + 5, 4 + 3 + 2)]      # every fourth pair is timed
    finally:
        for c in ctxs:
            c.timing_enable(0)
            c.set_gate(None)
        for s_ in shards:
            s_.free()
        for c in ctxs[1:]:
            c.close()
        gate.close()
        syn.free()
