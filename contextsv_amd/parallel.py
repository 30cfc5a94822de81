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


def gather_calls(per_shard: dict[int, np.ndarray], cap: int, dist=None, device=None) -> dict[int, np.ndarray] | None:
    """Final gather: every rank contributes its shards' merged calls; rank 0 gets {shard id: calls} for
    all shards (other ranks get None). One all_gather of a fixed-size buffer (latency-bound: ~1 MB genome-wide)."""
    import torch
    buf = pack_calls(per_shard, cap)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return unpack_calls(buf)
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    out = torch.empty(dist.get_world_size() * t.numel(), dtype=torch.int32, device=t.device)
    dist.all_gather_into_tensor(out, t)
    if dist.get_rank() != 0:
        return None
    host = out.cpu().numpy().reshape(dist.get_world_size(), -1)
    merged = {}
    for r in range(host.shape[0]):
        merged.update(unpack_calls(host[r]))
    return merged
