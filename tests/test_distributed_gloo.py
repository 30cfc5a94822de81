"""N>1 path on CPU: world_size-2 gloo run of the shard assignment + final gather (the path's only
exchange step; RCCL on the GPU box, gloo here)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from contextsv_amd import host, parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_calls(shard_id, n):
    rng = np.random.default_rng(shard_id)
    s = np.sort(rng.integers(1, 1_000_000, n)).astype(np.uint32)
    c = host.make_calls(s, s + rng.integers(50, 5000, n).astype(np.uint32), rng.choice([0, 3], n), rng.integers(2, 40, n))
    c["id"] = shard_id * 100000 + np.arange(n)
    return c


def _worker(rank, world, port, weights, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = parallel.assign_shards(weights, world)[rank]
    per_shard = {sid: _fake_calls(sid, 10 + 7 * sid) for sid in mine}
    got = parallel.gather_calls(per_shard, cap=4096, dist=dist)
    if rank == 0:
        ok = sorted(got) == list(range(len(weights)))
        for sid in got:
            ref = _fake_calls(sid, 10 + 7 * sid)
            ok = ok and got[sid].tobytes() == ref.tobytes()
        q.put(ok)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_assign_shards_balances():
    chr_len = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622,
               133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
    a = parallel.assign_shards(chr_len, 8)
    assert sorted(i for x in a for i in x) == list(range(24))
    loads = [sum(chr_len[i] for i in x) for x in a]
    assert max(loads) / (sum(chr_len) / 8) < 1.05          # within 5 % of perfect balance
    assert parallel.assign_shards(chr_len, 1) == [list(range(24))]


def test_pack_roundtrip():
    per = {3: _fake_calls(3, 5), 11: _fake_calls(11, 0), 7: _fake_calls(7, 40)}
    out = parallel.unpack_calls(parallel.pack_calls(per, 128))
    assert sorted(out) == [3, 7, 11] and all(out[k].tobytes() == per[k].tobytes() for k in per)


def test_gather_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    weights = [5, 3, 9, 1, 4]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, weights, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_cpp_shard_assignment_equals_the_python_one():
    """SVCaller::assignShards (what a C++ caller of the host mirror uses) makes the partition parallel.assign_shards makes: random weights with
    ties, more ranks than shards, the benchmark's own contig lengths."""
    from contextsv_amd import host, parallel
    rng = np.random.default_rng(5)
    cases = [list(rng.integers(1, 50, n).astype(float)) for n in (1, 2, 7, 24, 40)] + [[3.0] * 9, [248956422, 242193529, 198295559, 190214555, 181538259, 170805979,
             159345973, 145138636, 138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
             46709983, 50818468, 156040895, 57227415]]
    for w in cases:
        for world in (1, 2, 3, 4, 8, 30):
            assert host.assign_shards(w, world) == parallel.assign_shards(w, world)
