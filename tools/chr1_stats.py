"""chr1 alone (a rank of eight): per-contig device-chain and host-merge times of the step. usage: python tools/chr1_stats.py [--lanes 3]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import contextsv_amd as cs
from contextsv_amd import host
from hmm_params import WGS_HMM
ap = argparse.ArgumentParser(); ap.add_argument("--lanes", type=int, default=3); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--len", type=int, default=248956422)
a = ap.parse_args()
ctx = cs.Context(0); host.set_context(ctx)
gate = cs.Gate(0) if a.lanes > 1 else None
lanes = [cs.Context(0) for _ in range(a.lanes)] if a.lanes > 1 else []
for c in lanes: c.set_gate(gate)
hmm = cs.make_hmm(**WGS_HMM)
g = host.Genome()
syn = host.SynthShard(0x5EED0000 + 3001, a.len, 30.0, 0, 32)
g.add_synth(ctx, "chr1", 0, syn, snp_seed=0x5EED0000 + 3001, with_snps=True); syn.free()
for _ in range(3): g.run(ctx, hmm, lanes=lanes)
t0 = time.perf_counter(); dev = mer = 0.0
for _ in range(a.steps):
    calls, tid, st, per = g.run(ctx, hmm, lanes=lanes)
    dev += per[0].ms_device; mer += per[0].ms_host_merge
el = (time.perf_counter() - t0) / a.steps * 1e3
print(f"chr1 alone, lanes {a.lanes}: {el:.2f} ms/step; device chain {dev / a.steps:.2f} ms, host merge {mer / a.steps:.2f} ms; cigar {st.ms_cigar:.2f} cn {st.ms_cigar_cn:.2f} split {st.ms_split:.2f} split_cn {st.ms_split_cn:.2f} merge_final {st.ms_merge_final:.2f} prepare {st.ms_split_prepare:.2f}")
