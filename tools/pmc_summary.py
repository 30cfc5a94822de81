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


def summarise_all(directory):
    """every counter found under `directory`: {kernel: {counter: mean per launch}} for the kernels that matter"""
    acc = {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for row in csv.DictReader(open(path)):
            key = (row.get("Dispatch_Id"), row.get("Kernel_Name", "").split("(")[0], row.get("Counter_Name"))
            per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
        for (_, name, counter), v in per.items():
            a = acc.setdefault(name, {}).setdefault(counter, [0, 0.0])
            a[0] += 1
            a[1] += v
    return {k: {c: s / n for c, (n, s) in d.items()} for k, d in acc.items()}


if __name__ == "__main__":
    if sys.argv[2] == "--all":
        out = summarise_all(sys.argv[3])
        json.dump(out, open(sys.argv[1], "w"), indent=1)
        for k, d in out.items():
            if "scan" in k or "depth_tile" in k:
                print(k)
                for c, v in sorted(d.items()):
                    print("   %-32s %.4g" % (c, v))
        sys.exit(0)
    out = {}
    for spec in sys.argv[2:]:
        counter, directory = spec.split("=", 1)
        out[counter] = summarise(directory, counter)
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    for c, d in out.items():
        for k, v in sorted(d.items(), key=lambda kv: -kv[1]["mean_KB"])[:6]:
            print(c, k, v)
