"""Small seeded shard generators for the parity tests (numpy; sizes the oracle finishes in seconds)."""
import numpy as np

from contextsv_amd import Reads

M, I, D, N, S, H, P, EQ, X = range(9)


def random_shard(seed, n_reads=300, chr_len=200_000, mean_ops=60, big_frac=0.05, sorted_pos=True,
                 all_ops=True, with_flags=True, clip_end=False):
    """Reads with every CIGAR op, SV-sized I/D/S events clustered at shared loci, assorted flags/mapq."""
    rng = np.random.default_rng(seed)
    loci = rng.integers(1000, chr_len - 5000, 12)             # shared SV loci -> real clusters
    loci_len = rng.integers(50, 3000, 12)
    pos, flag, mapq, cig = [], [], [], []
    starts = rng.integers(0, max(chr_len - 20000, 1), n_reads)
    if sorted_pos:
        starts.sort()
    for r in range(n_reads):
        p = int(starts[r])
        ops = []
        n_ops = max(1, int(rng.poisson(mean_ops)))
        if rng.random() < 0.3:
            ops.append((H, int(rng.integers(1, 200))))
        if rng.random() < 0.4:
            ops.append((S, int(rng.integers(1, 400))))
        ref = p
        for k in range(n_ops):
            mlen = int(rng.integers(1, 400))
            ops.append(((M if not all_ops else int(rng.choice([M, M, M, EQ, X]))), mlen))
            ref += mlen
            u = rng.random()
            if u < big_frac:
                # SV-sized event, snapped to a shared locus when one is near
                near = np.nonzero(np.abs(loci - ref) < 800)[0]
                ln = int(loci_len[near[0]] + rng.integers(-3, 4)) if len(near) else int(rng.integers(40, 2000))
                ln = max(ln, 1)
                op = int(rng.choice([I, D]))
                ops.append((op, ln))
                if op == D:
                    ref += ln
            elif u < 0.6:
                op = int(rng.choice([I, D, D, N, P] if all_ops else [I, D]))
                ln = int(rng.integers(1, 60)) if op != N else int(rng.integers(1, 3000))
                ops.append((op, ln))
                if op in (D, N):
                    ref += ln
        if rng.random() < 0.4:
            ops.append((S, int(rng.integers(1, 400))))
        if rng.random() < 0.2:
            ops.append((H, int(rng.integers(1, 100))))
        f = 0
        if with_flags:
            u = rng.random()
            if u < 0.03: f |= 0x100
            elif u < 0.05: f |= 0x400
            elif u < 0.07: f |= 0x200
            elif u < 0.09: f |= 0x4
            elif u < 0.15: f |= 0x800
            if rng.random() < 0.5: f |= 0x10
        q = 60 if rng.random() > 0.1 else int(rng.integers(0, 30))
        pos.append(p); flag.append(f); mapq.append(q); cig.append(ops)
    if clip_end:
        # reads whose soft clip sits at/after the contig end: exercises the `continue` quirk (sv_caller.cpp:602-604)
        for extra in range(6):
            pos.append(chr_len - 300)
            flag.append(0); mapq.append(60)
            cig.append([(S, 80), (M, 300 + extra), (S, 70 + extra), (I, 55), (M, 10), (S, 90)])
        if sorted_pos:
            o = np.argsort(np.asarray(pos), kind="stable")
            pos = [pos[i] for i in o]; flag = [flag[i] for i in o]; mapq = [mapq[i] for i in o]; cig = [cig[i] for i in o]
    return Reads.from_cigar_lists(pos, flag, mapq, cig), chr_len + 1


def random_intervals(seed, n, span=2_000_000, clustered=True, sort=False, zero_len=False):
    rng = np.random.default_rng(seed)
    if clustered:
        n_loci = max(1, n // 15)
        ls = rng.integers(1, span, n_loci)
        ll = np.exp(rng.uniform(np.log(50), np.log(10000), n_loci)).astype(np.int64)
        which = rng.integers(0, n_loci, n)
        s = ls[which] + rng.integers(-5, 6, n)
        ln = (ll[which] * (1 + rng.uniform(-0.02, 0.02, n))).astype(np.int64)
        noise = rng.random(n) < 0.25
        s[noise] = rng.integers(1, span, noise.sum())
        ln[noise] = rng.integers(50, 5000, noise.sum())
    else:
        s = rng.integers(1, span, n)
        ln = rng.integers(1, 2000, n)
    if zero_len:
        ln[rng.random(n) < 0.05] = 0
    s = np.maximum(s, 1).astype(np.uint32)
    e = (s + ln).astype(np.uint32)
    if sort:
        o = np.lexsort((e, s)); s, e = s[o], e[o]
    return s, e
