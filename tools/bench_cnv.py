#!/usr/bin/env python3
"""Secondary-kernel measurement (BASELINE.json configs[2] flavour): window log2 + batched Viterbi for the copy-number
pass over many SV candidates on a resident depth map. Prints one JSON line. Not the headline bench (bench.py)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "tests"))
import contextsv_amd as cs
from contextsv_amd import host
from hmm_params import WGS_HMM


def main():
    n_sv = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    chr_len = 50_000_000
    rng = np.random.default_rng(0)
    ctx = cs.Context(0)
    host.set_context(ctx)
    syn = host.SynthShard(0x5EED0000 + 2000 + 1, chr_len, 30.0, 0, 8)
    sh = ctx.upload(syn.reads, syn.depth_len)
    res = sh.pipeline()
    hmm = cs.make_hmm(**WGS_HMM)
    # SNPs: 1 per kb
    n_snp = chr_len // 1000
    pos = np.sort(rng.choice(np.arange(1000, chr_len - 1000), n_snp, replace=False)).astype(np.uint32)
    snps = {"pos": pos, "baf": np.clip(np.where(rng.random(n_snp) < 0.6, 0.5 + rng.normal(0, 0.05, n_snp), 1.0), 0, 1),
            "pfb": np.zeros(n_snp), "has_pfb": np.zeros(n_snp, np.uint8)}
    s = rng.integers(1000, chr_len - 1_100_000, n_sv).astype(np.uint32)
    ln = np.exp(rng.uniform(np.log(2000), np.log(100_000), n_sv)).astype(np.int64)
    calls = host.make_calls(s, (s + ln).astype(np.uint32), rng.choice([0, 3], n_sv))
    host.cn_prediction(ctx, sh, calls, hmm, res.mean_cov, snps, split=False)      # warm-up
    ctx.timing_enable(True); ctx.timing_reset()
    t0 = time.perf_counter()
    out = host.cn_prediction(ctx, sh, calls, hmm, res.mean_cov, snps, split=False)
    wall = time.perf_counter() - t0
    tm = ctx.timing()
    n_obs = int(np.maximum(20, (ln // 1000) + 1).sum())      # ~ observations (>= sample_size, ~1 SNP / kb)
    print(json.dumps({"n_sv": n_sv, "approx_observations": n_obs, "wall_ms": wall * 1e3, "window_kernel_ms": tm["window"][0],
                      "viterbi_kernels_ms": tm["viterbi"][0], "sv_per_s_wall": n_sv / wall,
                      "window_bases_summed": int(ln.sum()), "window_GBps": float(ln.sum() * 4 / (tm["window"][0] * 1e-3) / 1e9) if tm["window"][0] else None,
                      "called": int((out["cn_state"] != 0).sum())}))
    sh.free(); syn.free(); ctx.close()


if __name__ == "__main__":
    main()
