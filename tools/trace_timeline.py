"""Timeline of the last step of a rocprofv3 --kernel-trace run: every kernel with its start offset, duration and queue, and the idle time in
front of it. usage: python tools/trace_timeline.py DIR [n_last_scans=1]"""
import csv, glob, os, sys
rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("csv::", "").replace("void ", ""), r.get("Queue_Id", "")))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
scans = [i for i, r in enumerate(rows) if "cigar_scan" in r[2]]
first = scans[-n]
# the step starts a little before its first scan (the split pass's ordering kernels may already run): go back to the previous idle gap > 200 us
i0 = first
while i0 > 0 and rows[i0][0] - max(r[1] for r in rows[max(0, i0 - 8):i0]) < 200_000: i0 -= 1
t0 = rows[i0][0]
busy_end = t0
for s, e, k, q in rows[i0:]:
    gap = s - busy_end
    print("+%9.1f us  %8.1f us  q%-3s %-44s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, k[:44], ("   <- %.0f us idle before" % (gap / 1e3)) if gap > 20_000 else ""))
    busy_end = max(busy_end, e)
print("step on the device: %.3f ms" % ((busy_end - t0) / 1e6))
