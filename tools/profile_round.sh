#!/bin/bash
# The per-round evidence under profiles/: kernel stats of the default bench run, HBM traffic of the two big kernels (separate
# --pmc passes, never combined with traces) and the bench line of the same build.
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh gpurun_out/r01x   -> copy the summaries to profiles/r01x/
set -e
OUT=$(realpath -m "$1"); mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 "$REPO/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -n 5 "$OUT/bench.err"; exit 1; }
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" --no-cpu-baseline --no-from-file --no-two-lanes > "$OUT/stats.log" 2>&1 || { echo "stats pass failed"; tail -n 5 "$OUT/stats.log"; exit 1; }
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python3 "$REPO/bench.py" --lanes 1 --no-two-lanes --no-cpu-baseline --no-from-file --steps 4 --warmup 1 > "$OUT/pmc_$c.log" 2>&1 || { echo "pmc $c failed"; tail -n 5 "$OUT/pmc_$c.log"; exit 1; }
    echo "pmc $c done"
done
cd "$REPO"
cp "$(ls $OUT/stats/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv"
python3 tools/pmc_summary.py "$OUT/pmc_summary.json" FETCH_SIZE="$OUT/pmc_FETCH_SIZE" WRITE_SIZE="$OUT/pmc_WRITE_SIZE" > "$OUT/pmc_summary.txt"
python3 - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
d = json.load(open(out + "/pmc_summary.json"))
def traffic(k):
    f = d["FETCH_SIZE"].get(k, {}).get("mean_KB", 0.0); w = d["WRITE_SIZE"].get(k, {}).get("mean_KB", 0.0)
    return (2 * f + w) * 1024
t = {"depth": traffic("csv::depth_tile_kernel"), "cigar_scan": traffic("csv::cigar_scan_kernel"),
     "_note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes (pmc_summary.json of the same round); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads)"}
json.dump(t, open(out + "/pmc_traffic.json", "w"), indent=1)
print(t)
PY
head -n 12 "$OUT/kernel_stats.csv"
