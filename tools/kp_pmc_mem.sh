#!/bin/bash
# Memory-side counters of the probe's kernels (separate rocprofv3 --pmc passes, no trace flags): read requests and their summed
# occupancy at the L2 <-> fabric interface (level / requests = average outstanding time in cycles, by Little's law).
# usage: tools/kp_pmc_mem.sh OUTDIR [kernel_probe args]
OUT=$(realpath -m "$1"); shift; mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1 || true
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_LEVEL_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_REQ_sum TCC_READ_sum TCC_TAG_STALL_sum"; do
    i=$((i + 1))
    timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d "$OUT/m$i" -- python3 "$REPO/tools/kernel_probe.py" --steps 5 "$@" > "$OUT/m$i.log" 2>&1 || { echo "pass $i ($grp) failed"; tail -n 2 "$OUT/m$i.log"; }
    echo "pass $i done"
done
cd "$REPO"
python3 tools/pmc_summary.py "$OUT/mem_summary.json" --all "$OUT" > /dev/null 2>&1
python3 - "$OUT/mem_summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "depth_tile" in k or "cigar_scan" in k:
        print(k)
        for c in sorted(v): print(f"   {c:36s} {v[c]:18.1f}")
PY
grep -i "RDREQ\|READ_REQ_LAT" "$OUT/avail.txt" | head -12
