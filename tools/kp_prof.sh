#!/bin/bash
# rocprofv3 kernel stats of one kernel_probe workload: tools/kp_prof.sh <out dir> [kernel_probe args]
R=$PWD; out=$R/${1:-gpurun_out/kp/prof}; shift
mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$R/tools/kernel_probe.py" "$@" > "$out/log.txt" 2>&1
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.2f} min_us {float(r["MinNs"])/1e3:9.2f} pct {r["Percentage"]}')
PY
