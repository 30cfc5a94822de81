import csv, glob, sys
rows=[]
for path in glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("csv::","").replace("void ",""), r.get("Queue_Id","")))
rows.sort()
scans=[i for i,r in enumerate(rows) if 'cigar_scan' in r[2]]
first=scans[-24]
t0=rows[first][0]
gq=rows[first][3]
big=[r for r in rows[first:] if ('cigar_scan' in r[2] or 'depth_tile' in r[2])]
tend=max(r[1] for r in big)
print("gate queue", gq, "pass %.2f ms"%((tend-t0)/1e6))
byq={}
for s,e,k,q in rows[first:]:
    if s>tend: break
    a=byq.setdefault(q,{}); a[k]=a.get(k,0)+1
for q in sorted(byq): print(" q",q, sorted(byq[q].items(), key=lambda kv:-kv[1])[:6])
last_end=t0; gaps=0
for s,e,k,q in rows[first:]:
    if s>tend: break
    if not ('cigar_scan' in k or 'depth_tile' in k or 'depth_items' in k): continue
    gap=s-last_end
    if gap>30e3:
        gaps+=gap
        inside={}
        for s2,e2,k2,q2 in rows[first:]:
            if e2>last_end and s2<s and not ('cigar_scan' in k2 or 'depth_tile' in k2): inside[(q2,k2[:18])]=inside.get((q2,k2[:18]),0)+1
        print("gap %.0f us at +%.1f ms: %s"%(gap/1e3,(last_end-t0)/1e6, sorted(inside.items(), key=lambda kv:-kv[1])[:5]))
    last_end=max(last_end,e)
print("gaps total %.2f ms"%(gaps/1e6))
