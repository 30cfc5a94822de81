"""Determinism soak: the same resident genome stepped N times, the SHA-256 of every step's (tid, call records) must be ONE value — the
early batches, the split chain's portions and the look-back of the radix passes all depend on timing, the calls must not.
usage (GPU box, repo root): python tools/soak_digest.py [--tech ont|hifi] [--depth D] [--contigs K] [--lanes L] [--steps N] [--scale S]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench
import contextsv_amd as cs
from contextsv_amd import host
from hmm_params import WGS_HMM

ap = argparse.ArgumentParser()
ap.add_argument("--tech", default="ont"); ap.add_argument("--depth", type=float, default=30.0); ap.add_argument("--contigs", type=int, default=24)
ap.add_argument("--lanes", type=int, default=3); ap.add_argument("--steps", type=int, default=100); ap.add_argument("--scale", type=float, default=1.0)
a = ap.parse_args()
tech = 0 if a.tech == "ont" else 1
cfg = 3 if tech == 0 else 4
gate = cs.Gate(0) if a.lanes > 1 else None
ctx = cs.Context(0)
host.set_context(ctx)
lanes = [cs.Context(0) for _ in range(a.lanes)] if a.lanes > 1 else []
for c in lanes:
    c.set_gate(gate)
g = host.Genome()
t0 = time.perf_counter()
for k in range(a.contigs):
    syn = host.SynthShard(bench.seed_of(cfg, k), max(200_000, int(bench.GRCH38[k] * a.scale)), a.depth, tech, 16)
    g.add_synth(ctx, bench.NAMES[k], k, syn, snp_seed=bench.seed_of(cfg, k), with_snps=True)
    syn.free()
hmm = cs.make_hmm(**WGS_HMM)
seen = {}
t1 = time.perf_counter()
for s in range(a.steps):
    calls, tid, st, per = g.run(ctx, hmm, lanes=lanes, capacity=1 << 18, copy=False)
    d = bench.call_digest(tid, calls)
    seen[d] = seen.get(d, 0) + 1
    if (s + 1) % 25 == 0:
        print("step", s + 1, "distinct digests", len(seen), flush=True)
el = time.perf_counter() - t1
print(json.dumps({"tech": a.tech, "contigs": a.contigs, "lanes": a.lanes, "steps": a.steps, "ms_per_step": el * 1e3 / a.steps, "stage_s": round(t1 - t0, 1),
                  "distinct_digests": len(seen), "digests": {k[:16]: v for k, v in seen.items()}, "calls": int(len(calls))}))
sys.exit(0 if len(seen) == 1 else 1)
