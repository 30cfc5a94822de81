"""bench.py's N > 1 path rehearsed on ONE card (CSV_BENCH_REHEARSE=1: gloo instead of RCCL, the ranks share the card): the 24 contigs
of a scaled-down genome are bin-packed over two and over four ranks (at most four processes on the card), every rank steps through its own contigs, the merged calls are gathered to
rank 0 — and rank 0 asserts that the gathered call set is byte-identical to the set a single rank computes for the whole genome
(--verify-against-single). Also checks the JSON line's contract keys at N = 1 and N = 2."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


COMMON = ["--steps", "2", "--warmup", "1", "--scale", "0.004", "--gen-threads", "4", "--no-from-file"]


def test_bench_single_rank_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *COMMON, "--cpu-sample-frac", "0.5"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["unit"] == "reads/s" and j["value"] > 0 and j["config"]["contigs"] == 24
    assert "whole genome" in j["config"]["workload"] and j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1
    assert j["roofline"]["launches_per_step"] == 24
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] == 24 and j["cpu_baseline"]["value"] > 0
    assert j["chr1_cnv"]["value"] > 0 and j["chr22_cigar_path"]["value"] > 0
    st = j["stage_ms_per_step_rank0"]
    assert st["ms_cigar"] > 0 and st["ms_total"] >= st["ms_cigar"]


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_ranks_sharded_equals_single(ranks):
    env = dict(os.environ, CSV_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), *COMMON, "--no-legs", "--no-cpu-baseline", "--verify-against-single"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    j = _line(r.stdout)
    assert j["n_gpus"] == ranks and j["scaling"] == "strong" and j["verify"]["sharded_equals_single"] is True and j["verify"]["calls"] > 50
    assert 16 // ranks <= len(j["config"]["contigs_on_rank0"]) <= 32 // ranks


def test_gather_on_the_gpu_stages_through_pinned_buffers():
    """parallel.gather_calls with device tensors (the RCCL path of the N > 1 bench; two ranks cannot share a card under RCCL, so the process
    group is a stand-in that copies this rank's buffer into every slot): page-locked staging kept between calls, a smaller capacity
    replaces the buffers, the records come back unchanged. In a process of its own, torch first — as in bench.py: torch finds no device
    once another copy of the HIP runtime (the product library's) is up in the process."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gather_device_check.py")], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "gather ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("tech,depth", [("ont", "30"), ("hifi", "60")])
def test_repeated_steps_give_one_call_set(tech, depth):
    """tools/soak_digest.py on a scaled-down genome: 24 contigs, three lanes, 40 steps — the early batches, the split chain's portions beside the
    pass and the radix passes' look-back depend on timing; the SHA-256 of every step's call records must be one value (at full scale: 150
    steps each, profiles/r03e/soak.txt)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_digest.py"), "--tech", tech, "--depth", depth, "--scale", "0.004", "--steps", "40"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["distinct_digests"] == 1 and j["steps"] == 40 and j["calls"] > 20
