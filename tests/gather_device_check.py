"""Body of tests/test_gpu_bench_sharded.py::test_gather_on_the_gpu_stages_through_pinned_buffers (run as a script, torch first)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.cuda.set_device(0)
from contextsv_amd import parallel  # noqa: E402
from contextsv_amd.host import CALL_DTYPE  # noqa: E402


class FakeGroup:
    def is_initialized(self): return True
    def get_world_size(self): return 3
    def get_rank(self): return 0
    def all_gather_into_tensor(self, out, t):
        assert out.is_cuda and t.is_cuda and out.numel() == 3 * t.numel()
        out.view(3, -1)[:] = t


rng = np.random.default_rng(3)
dev = torch.device("cuda", 0)
for cap in (4096, 1024, 1024):
    per = {}
    for sid in (2, 5, 11):
        a = np.zeros(int(rng.integers(0, 300)), CALL_DTYPE)
        a["start"] = rng.integers(0, 1 << 30, len(a)); a["end"] = a["start"] + 5; a["cluster_size"] = sid
        per[sid] = a
    got = parallel.gather_calls(per, cap=cap, dist=FakeGroup(), device=dev)
    assert sorted(got) == [2, 5, 11] and all(np.array_equal(got[k], per[k]) for k in per)
    assert len(parallel._gather_bufs) == 1
try:
    parallel.gather_calls({1: np.zeros(2000, CALL_DTYPE)}, cap=1024, dist=FakeGroup(), device=dev)
    raise SystemExit("capacity overflow not reported")
except ValueError:
    pass
print("gather ok")
