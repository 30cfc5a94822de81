#!/usr/bin/env python3
"""Whole-genome-sized from-file run (synthetic): the 24 primary GRCh38 contig lengths at --depth x ONT, each staged as its own
coordinate-sorted BGZF BAM + BAI (generator of SURVEY.md §8d, seed = 0x5EED0000 + 1000 * config + contig index), then every contig
through SVCaller::runBam — BGZF inflate + BAM decode on the host, upload, CIGAR scan, depth, ordering, DBSCAN, mergeSVs, CIGAR
copy-number pass (no SNP file: every window gets the dummy observation), split-read pass + its copy-number pass, final merges — one after
the other on one GPU.
Prints one JSON line. Staging (generation + BAM writing) is reported but is not part of the run time."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
          114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=float, default=30.0)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--scale", type=float, default=1.0, help="scale every contig length (quick runs)")
    ap.add_argument("--one-bam", action="store_true", help="all contigs in ONE BAM and one runBam call: contig i + 1 is decoded while contig i is on "
                    "the device, every shard stays resident until the copy-number pass at the end")
    ap.add_argument("--no-split", action="store_true", help="skip the split-read pass (round 1's configuration)")
    args = ap.parse_args()
    import contextsv_amd as cs
    from contextsv_amd import host
    from hmm_params import WGS_HMM
    ctx = cs.Context(0)
    host.set_context(ctx)
    hmm = cs.make_hmm(**WGS_HMM)
    tot = {"reads": 0, "cigar_ops": 0, "bam_bytes": 0, "calls": 0, "stage_s": 0.0, "run_s": 0.0, "decode_wait_s": 0.0}
    per = []
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        if args.one_bam:
            n = min(args.contigs, 24)
            lens = [int(GRCH38[k] * args.scale) for k in range(n)]
            bam = os.path.join(d, "genome.bam")
            t0 = time.perf_counter()
            w = host.SynthBamWriter(bam, NAMES[:n], lens, level=1, threads=args.threads)
            for k in range(n):
                syn = host.SynthShard(0x5EED0000 + 1000 * 1 + k + 1, lens[k], args.depth, 0, args.threads)
                w.append(syn, k)
                tot["reads"] += syn.reads.n_reads; tot["cigar_ops"] += syn.reads.n_cigar
                syn.free()
                print(NAMES[k], "staged", file=sys.stderr, flush=True)
            w.close()
            t1 = time.perf_counter()
            calls, tids, bs = host.run_bam(ctx, bam, hmm, threads=args.threads, split_svs=not args.no_split, cigar_cn=True, capacity=1 << 22)
            t2 = time.perf_counter()
            tot["bam_bytes"] = os.path.getsize(bam); tot["calls"] = len(calls)
            tot["stage_s"] = t1 - t0; tot["run_s"] = t2 - t1; tot["decode_wait_s"] = bs["ms_decode"] * 1e-3
            per = [{"contig": NAMES[k], "calls": int((tids == k).sum())} for k in range(n)]
        for k in range(0 if args.one_bam else min(args.contigs, 24)):
            length = int(GRCH38[k] * args.scale)
            t0 = time.perf_counter()
            syn = host.SynthShard(0x5EED0000 + 1000 * 1 + k + 1, length, args.depth, 0, args.threads)
            bam = os.path.join(d, NAMES[k] + ".bam")
            nbytes = syn.write_bam(bam, NAMES[k], level=1, threads=args.threads)
            n_reads, n_ops = syn.reads.n_reads, syn.reads.n_cigar
            syn.free()
            t1 = time.perf_counter()
            calls, _, bs = host.run_bam(ctx, bam, hmm, chromosomes=[NAMES[k]], threads=args.threads, split_svs=not args.no_split, cigar_cn=True)
            t2 = time.perf_counter()
            os.remove(bam); os.remove(bam + ".bai")
            tot["reads"] += n_reads; tot["cigar_ops"] += n_ops; tot["bam_bytes"] += nbytes; tot["calls"] += len(calls)
            tot["stage_s"] += t1 - t0; tot["run_s"] += t2 - t1; tot["decode_wait_s"] += bs["ms_decode"] * 1e-3
            per.append({"contig": NAMES[k], "reads": int(n_reads), "run_s": round(t2 - t1, 3), "calls": int(len(calls))})
            print(NAMES[k], n_reads, "reads", "%.2f s staged, %.3f s run, %d calls" % (t1 - t0, t2 - t1, len(calls)), file=sys.stderr, flush=True)
    out = {"workload": ("GRCh38 primary contig lengths x %.2f, %gx synthetic ONT, runBam (CIGAR + depth + DBSCAN + mergeSVs + CIGAR CN pass" % (args.scale, args.depth))
                       + ("" if args.no_split else " + split-read pass + split CN pass") + " + final merges)",
           "layout": "one BAM, one runBam call" if args.one_bam else "one BAM per contig, one runBam call each",
           "n_contigs": len(per), "inflate_threads": args.threads, **{k: (round(v, 3) if isinstance(v, float) else int(v)) for k, v in tot.items()},
           "reads_per_s_from_file": tot["reads"] / tot["run_s"], "per_contig": per}
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
