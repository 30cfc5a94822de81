"""One whole-genome step under rocprofv3 --kernel-trace: inside the CIGAR pass (first scan start .. last depth end of the LAST step), how
much of the time has a big kernel (scan / depth) on the device, where the gaps are and what ran in them. usage: python tools/trace_step.py DIR [contigs=24]"""
import csv, glob, os, sys
rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("csv::", ""), r.get("Queue_Id", "")))
rows.sort()
n_contigs = int(sys.argv[2]) if len(sys.argv) > 2 else 24
scans = [r for r in rows if "cigar_scan" in r[2]]
depths = [r for r in rows if "depth_tile" in r[2]]
last_scans, last_depths = scans[-n_contigs:], depths[-n_contigs:]
t0, t1 = last_scans[0][0], max(d[1] for d in last_depths)
big = sorted(last_scans + last_depths)
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
span = t1 - t0
print("CIGAR pass on the device: %.3f ms; a big kernel running %.3f ms (%.1f %%); scan %.3f ms + depth %.3f ms summed" % (
    span / 1e6, union([(b[0], b[1]) for b in big]) / 1e6, 100.0 * union([(b[0], b[1]) for b in big]) / span,
    sum(b[1] - b[0] for b in last_scans) / 1e6, sum(b[1] - b[0] for b in last_depths) / 1e6))
gaps = []
end = big[0][1]
for b in big[1:]:
    if b[0] > end: gaps.append((b[0] - end, end, b[0], b[2]))
    end = max(end, b[1])
gaps.sort(reverse=True)
print("gaps between big kernels: n=%d, sum %.3f ms" % (len(gaps), sum(g[0] for g in gaps) / 1e6))
for g, a, b, nxt in gaps[:12]:
    inside = [r for r in rows if r[1] > a and r[0] < b and "cigar_scan" not in r[2] and "depth_tile" not in r[2]]
    names = {}
    for r in inside: names[r[2]] = names.get(r[2], 0) + 1
    print("   %7.1f us at +%8.1f us before %-20s | during it: %s" % (g / 1e3, (a - t0) / 1e3, nxt[:20], ", ".join("%s x%d" % kv for kv in sorted(names.items(), key=lambda kv: -kv[1])[:4])))
others = [r for r in rows if t0 <= r[0] <= t1 and "cigar_scan" not in r[2] and "depth_tile" not in r[2]]
by = {}
for s, e, k, q in others:
    a = by.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
print("other kernels inside the pass:")
for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print("   %-36s n=%4d  total %.3f ms" % (k[:36], n, t / 1e6))
# the step as a whole: from the first scan to the first scan of ... (last step only: to the last kernel)
print("after the pass: last kernel ends %.3f ms after the last depth" % ((rows[-1][1] - t1) / 1e6))
# timeline around the largest gap
if gaps:
    g, a, b, nxt = gaps[0]
    print("timeline around the largest gap (%.1f us):" % (g / 1e3))
    for s, e, k, q in rows:
        if e > a - 400e3 and s < b + 200e3:
            print("   %+9.1f us .. %+9.1f us  q%-3s %s" % ((s - a) / 1e3, (e - a) / 1e3, q, k[:44]))
