"""Multi-GPU layout of the hot path: chromosomes (shards) are independent (the reference already runs
them as independent pool tasks, sv_caller.cpp:827-863), so they are bin-packed over the ranks and the
only exchange step is the final gather of the merged call records to rank 0 — RCCL when the process
group is NCCL (one rank per GPU over xGMI), gloo on CPU for the tests. No data-path collective."""
from __future__ import annotations

import numpy as np

from .host import CALL_DTYPE

REC_WORDS = CALL_DTYPE.itemsize // 4      # one merged call = 12 int32 words


def assign_shards(weights, world: int) -> list[list[int]]:
    """Longest-processing-time bin packing of shard weights (e.g. read counts) over `world` ranks.
    Deterministic; returns the shard indices of every rank (each list ascending)."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += weights[i]
    return [sorted(x) for x in out]


def pack_calls(per_shard: dict[int, np.ndarray], cap: int) -> np.ndarray:
    """Fixed-size int32 buffer: [n_shards, (shard id, n_calls)*, records...] padded to `cap` records."""
    n_rec = sum(len(v) for v in per_shard.values())
    if n_rec > cap:
        raise ValueError(f"gather capacity {cap} < {n_rec} merged calls")
    head = [len(per_shard)]
    body = []
    for sid in sorted(per_shard):
        head += [sid, len(per_shard[sid])]
        body.append(np.ascontiguousarray(per_shard[sid], CALL_DTYPE))
    buf = np.zeros(1 + 2 * 64 + cap * REC_WORDS, np.int32)
    if len(per_shard) > 64:
        raise ValueError("more than 64 shards per rank")
    buf[: len(head)] = head
    if body:
        flat = np.frombuffer(np.concatenate(body).tobytes(), dtype=np.int32)
        buf[129: 129 + len(flat)] = flat
    return buf


def unpack_calls(buf: np.ndarray) -> dict[int, np.ndarray]:
    n = int(buf[0])
    out, off = {}, 129
    for k in range(n):
        sid, cnt = int(buf[1 + 2 * k]), int(buf[2 + 2 * k])
        out[sid] = np.frombuffer(buf[off: off + cnt * REC_WORDS].tobytes(), dtype=CALL_DTYPE).copy()
        off += cnt * REC_WORDS
    return out


_gather_bufs: dict = {}            # (cap, world, device) -> page-locked staging + device tensors of the gather (kept between steps)


def gather_calls(per_shard: dict[int, np.ndarray], cap: int, dist=None, device=None) -> dict[int, np.ndarray] | None:
    """Final gather: every rank contributes its shards' merged calls; rank 0 gets {shard id: calls} for
    all shards (other ranks get None). One all_gather of a fixed-size buffer of `cap` records per rank (latency-bound: a genome's
    merged calls are ~1.4 MB in all) — size `cap` from the counts of a warm-up step, not from a worst case: every rank receives
    world x cap records and rank 0 copies them to the host. On a GPU the buffers are staged through page-locked memory kept between calls."""
    import torch
    buf = pack_calls(per_shard, cap)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return unpack_calls(buf)
    world = dist.get_world_size()
    if device is not None and torch.device(device).type == "cuda":
        key = (cap, world, str(device))
        st = _gather_bufs.get(key)
        if st is None:
            st = {"send_h": torch.empty(len(buf), dtype=torch.int32).pin_memory(),
                  "send_d": torch.empty(len(buf), dtype=torch.int32, device=device),
                  "recv_d": torch.empty(world * len(buf), dtype=torch.int32, device=device),
                  "recv_h": torch.empty(world * len(buf), dtype=torch.int32).pin_memory() if dist.get_rank() == 0 else None}
            _gather_bufs.clear()                       # (one size at a time: a new capacity replaces the old buffers)
            _gather_bufs[key] = st
        st["send_h"].numpy()[:] = buf
        st["send_d"].copy_(st["send_h"], non_blocking=True)
        dist.all_gather_into_tensor(st["recv_d"], st["send_d"])
        if dist.get_rank() == 0:
            st["recv_h"].copy_(st["recv_d"], non_blocking=True)
        torch.cuda.current_stream(device).synchronize()       # (every rank: the staging buffers are reused by the next call)
        if dist.get_rank() != 0:
            return None
        host = st["recv_h"].numpy().reshape(world, -1)
    else:
        t = torch.from_numpy(buf)
        out = torch.empty(world * t.numel(), dtype=torch.int32)
        dist.all_gather_into_tensor(out, t)
        if dist.get_rank() != 0:
            return None
        host = out.numpy().reshape(world, -1)
    merged = {}
    for r in range(host.shape[0]):
        merged.update(unpack_calls(host[r]))
    return merged
