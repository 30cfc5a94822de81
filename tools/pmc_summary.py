"""Mean counter value per kernel launch from rocprofv3 --pmc runs (counter_collection CSVs).
usage: python tools/pmc_summary.py OUT.json COUNTER=dir [COUNTER=dir ...]"""
import csv
import glob
import json
import os
import sys


def summarise(directory, counter):
    acc = {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = {}
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            key = (row.get("Dispatch_Id"), row.get("Kernel_Name", "").split("(")[0])
            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])     # summed over XCDs / instances
        for (_, name), v in per_dispatch.items():
            a = acc.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += v
    return {k: {"launches": n, "mean_KB": s / n} for k, (n, s) in acc.items()}


if __name__ == "__main__":
    out = {}
    for spec in sys.argv[2:]:
        counter, directory = spec.split("=", 1)
        out[counter] = summarise(directory, counter)
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    for c, d in out.items():
        for k, v in sorted(d.items(), key=lambda kv: -kv[1]["mean_KB"])[:6]:
            print(c, k, v)
