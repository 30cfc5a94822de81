#!/bin/bash
# rocprofv3 kernel stats of a bench.py run: tools/kstats.sh OUTDIR [bench.py args]
OUT=$(realpath -m "$1"); shift; mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-legs --no-cpu-baseline "$@" > "$OUT/stats.log" 2>&1 || { echo "stats pass failed"; tail -n 5 "$OUT/stats.log"; exit 1; }
cd "$REPO"
cp "$(ls $OUT/stats/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv"
rm -rf "$OUT/stats"
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms (6 steps)", tot / 1e6)
for r in rows[:40]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.2f} tot_ms {float(r["TotalDurationNs"])/1e6:8.2f} pct {r["Percentage"]}')
PY
