"""The two bandwidth-shaped kernels with the device to themselves: one synthetic contig resident in HBM, the per-chromosome device
pipeline (csvgpu_chr_pipeline_dev: scan + depth + ordering + DBSCAN) N times with event timers around every kernel group, and a
digest of everything the pair writes (signatures as a sorted set, ref_end / q_start / q_end, the depth map, its sum and non-zero count,
the labels) — the A/B tool for kernel experiments: a changed kernel must print the same digest as the committed build
(tools/probes/kernel_probe_digests.json; --record rewrites it).

    python tools/kernel_probe.py [--contig 22|1] [--tech ont|hifi] [--depth 30] [--steps 20] [--record]
"""
import argparse, hashlib, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GRCH38 = {1: 248956422, 22: 50818468, 21: 46709983}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contig", type=int, default=22)
    ap.add_argument("--tech", default="ont")
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--record", action="store_true")
    args = ap.parse_args()
    import contextsv_amd as cs
    from contextsv_amd import host
    tech = 0 if args.tech == "ont" else 1
    L = int(GRCH38[args.contig] * args.scale)
    syn = host.SynthShard(0x5EED0000 + 1000 * (3 if tech == 0 else 4) + args.contig, L, args.depth, tech, min(32, os.cpu_count() or 8))
    reads, depth_len = syn.reads, syn.depth_len
    n_reads, n_cigar = reads.n_reads, reads.n_cigar
    ctx = cs.Context(0)
    sh = ctx.upload(reads, depth_len)
    ctx.synchronize()
    syn.free()
    res = sh.pipeline()
    ctx.synchronize()
    out = sh.fetch(res, want_depth=True)
    dig = {}
    for k in ("sig_del", "sig_ins"):
        a = out[k]
        o = np.lexsort((a["qpos_kind"], a["read"], a["end"], a["start"])) if len(a) else np.zeros(0, np.int64)
        dig[k + "_set"] = hashlib.sha256(np.ascontiguousarray(a[o]).tobytes()).hexdigest()[:16]
        dig[k + "_order"] = hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
    for k in ("label_del", "label_ins", "ref_end", "q_start", "q_end", "depth"):
        dig[k] = hashlib.sha256(np.ascontiguousarray(out[k]).tobytes()).hexdigest()[:16]
    dig["scalars"] = [int(res.n_sig), int(res.n_del), int(res.n_ins), int(res.depth_sum), int(res.depth_nonzero), int(res.min_pts)]
    del out

    for _ in range(3):
        sh.pipeline()
    ctx.timing_enable(1); ctx.timing_reset(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sh.pipeline()
    ctx.synchronize()
    el = time.perf_counter() - t0
    tm = {k: ms / n for k, (ms, n) in ctx.timing().items() if n}
    b_scan = 4.0 * n_cigar + 23.0 * n_reads + 16.0 * res.n_sig
    b_depth = 4.0 * n_cigar + 4.0 * depth_len
    key = f"chr{args.contig}x{args.scale:g}_{args.tech}_{args.depth:g}"
    gpath = os.path.join(ROOT, "tools", "probes", "kernel_probe_digests.json")
    gold = json.load(open(gpath)) if os.path.exists(gpath) else {}
    same = None
    if args.record:
        gold[key] = dig
        json.dump(gold, open(gpath, "w"), indent=1, sort_keys=True)
    elif key in gold:
        same = {k: gold[key].get(k) == v for k, v in dig.items()}
    line = {"workload": key, "reads": int(n_reads), "cigar_words": int(n_cigar), "signatures": int(res.n_sig),
            "ms_per_pipeline": round(el / args.steps * 1e3, 4), "kernel_ms": {k: round(v, 4) for k, v in tm.items()},
            "scan_GBps": round(b_scan / tm["cigar_scan"] / 1e6, 1), "scan_frac": round(b_scan / tm["cigar_scan"] / 1e6 / 8000, 4),
            "depth_GBps": round(b_depth / tm["depth"] / 1e6, 1), "depth_frac": round(b_depth / tm["depth"] / 1e6 / 8000, 4),
            "digest_equal": (all(same.values()) if same is not None else None),
            "digest_diff": ([k for k, v in same.items() if not v] if same else None)}
    print(json.dumps(line), flush=True)
    sh.free(); ctx.close()
    if same is not None and not all(same.values()):
        sys.exit(1)


if __name__ == "__main__":
    main()
