"""Timeline of a rocprofv3 --kernel-trace run: how much of the wall time between the first and last kernel has a big kernel
(scan / depth) running, how much only small kernels, how much nothing. usage: python tools/trace_gaps.py DIR [skip_fraction]"""
import csv, glob, os, sys
rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "")))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * skip          # drop warm-up
rows = [r for r in rows if r[0] >= t_lo]
big = [r for r in rows if "cigar_scan" in r[2] or "depth_tile" in r[2]]
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
span = rows[-1][1] - rows[0][0]
n_steps = sum(1 for r in rows if "cigar_scan" in r[2])
print("span %.3f ms, %d scans -> %.4f ms/step" % (span / 1e6, n_steps, span / 1e6 / max(n_steps, 1)))
print("any kernel running  %.1f %%" % (100.0 * union([(r[0], r[1]) for r in rows]) / span))
print("big kernel running  %.1f %%" % (100.0 * union([(r[0], r[1]) for r in big]) / span))
# gaps between consecutive big kernels
big.sort()
gaps = [big[i + 1][0] - max(b[1] for b in big[: i + 1][-4:]) for i in range(len(big) - 1)]
gaps = [g for g in gaps if g > 0]
if gaps:
    gaps.sort()
    print("gaps between big kernels: n=%d median %.1f us, p90 %.1f us, sum %.3f ms" % (len(gaps), gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, sum(gaps) / 1e6))
by = {}
for s, e, k, q in rows:
    a = by.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-44s n=%5d  mean %.1f us  total %.3f ms" % (k[:44], n, t / n / 1e3, t / 1e6))
# a short excerpt of the timeline
t0 = rows[len(rows) // 2][0]
for s, e, k, q in rows[len(rows) // 2: len(rows) // 2 + 40]:
    print("   +%8.1f us  %7.1f us  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, k[:50]))
