"""from-file throughput of one chr22-sized contig against the number of inflate threads (the host's CPU share decides): tools/inflate_scaling.py 8 16 32 64"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import contextsv_amd as cs
from contextsv_amd import host
from hmm_params import WGS_HMM
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
try:
    print("cgroup cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip(), flush=True)
except Exception as e:
    print("no cpu.max", e)
ctx = cs.Context(0); host.set_context(ctx)
hmm = cs.make_hmm(**WGS_HMM)
syn = host.SynthShard(0x5EED0000 + 1022, 50818468, 30.0, 0, 32)
with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
    bam = os.path.join(d, "s.bam")
    for wt in (16, 64):
        t0 = time.perf_counter(); nbytes = syn.write_bam(bam, "chr22", level=1, threads=wt); print("write", wt, "threads", round(time.perf_counter() - t0, 2), "s", nbytes, flush=True)
    for th in [int(x) for x in sys.argv[1:]] or [8, 16, 32, 64]:
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            calls, _, bs = host.run_bam(ctx, bam, hmm, chromosomes=["chr22"], threads=th, split_svs=True, cigar_cn=True)
            best = min(best, time.perf_counter() - t0)
        print("threads", th, "run", round(best, 4), "s decode wait", round(bs["ms_decode"] * 1e-3, 4), "reads/s", int(bs["n_reads"] / best), flush=True)
