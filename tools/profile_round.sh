#!/bin/bash
# The per-round evidence under profiles/: the bench line of the default run (whole-genome step + legs + CPU baseline), kernel stats of the
# same workload, HBM traffic of the two big kernels (separate --pmc passes, never combined with traces).
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh gpurun_out/r03x [hifi]  -> copy the summaries to profiles/r03x/
# (second argument "hifi": the 60x HiFi genome — BASELINE configs[4] on one GPU — instead of the 30x ONT one; the bench line is then the
# headline-only line of that workload)
set -e
OUT=$(realpath -m "$1"); mkdir -p "$OUT"
REPO=$(pwd)
WL=""; KEY="wgs"; LEGS=""
if [ "$2" = "hifi" ]; then WL="--tech hifi --depth 60"; KEY="hifi_wgs"; LEGS="--no-legs --no-cpu-baseline"; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 "$REPO/bench.py" --steps 20 --warmup 5 $WL $LEGS > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -n 5 "$OUT/bench.err"; exit 1; }
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-legs --no-cpu-baseline $WL > "$OUT/stats.log" 2>&1 || { echo "stats pass failed"; tail -n 5 "$OUT/stats.log"; exit 1; }
echo "stats done"
PMC_STEPS=2; PMC_WARM=1
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python3 "$REPO/bench.py" --lanes 1 --no-legs --no-cpu-baseline --steps $PMC_STEPS --warmup $PMC_WARM $WL > "$OUT/pmc_$c.log" 2>&1 || { echo "pmc $c failed"; tail -n 5 "$OUT/pmc_$c.log"; exit 1; }
    echo "pmc $c done"
done
cd "$REPO"
cp "$(ls $OUT/stats/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv"
python3 tools/pmc_summary.py "$OUT/pmc_summary.json" FETCH_SIZE="$OUT/pmc_FETCH_SIZE" WRITE_SIZE="$OUT/pmc_WRITE_SIZE" > "$OUT/pmc_summary.txt"
python3 - "$OUT" $((PMC_STEPS + PMC_WARM)) $KEY <<'PY'
import json, sys
out, n_steps, key = sys.argv[1], int(sys.argv[2]), sys.argv[3]
d = json.load(open(out + "/pmc_summary.json"))
def per_step(*parts):          # every kernel whose name holds one of `parts` (template instances carry their arguments in the name)
    tot = 0.0
    for c, sign in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        for k, v in d[c].items():
            if any(p in k for p in parts): tot += sign * v.get("mean_KB", 0.0) * v.get("launches", 0)
    return tot * 1024 / n_steps
t = {key: {"depth": per_step("depth_tile_kernel", "depth_items_kernel"), "depth_tiles_only": per_step("depth_tile_kernel"), "cigar_scan": per_step("cigar_scan_kernel", "cigar_scan_rows_kernel", "cigar_scan_lanes_kernel"),
             "source": "HBM bytes per whole-genome step (24 launches; depth = the tile kernel + the work-list kernel in front of it) = sum over launches of (2*FETCH_SIZE + WRITE_SIZE)*1024 / steps, from separate rocprofv3 --pmc passes "
                       "of `bench.py --lanes 1 --steps 2 --warmup 1` (" + out.split("/")[-1] + "/pmc_summary.json); FETCH_SIZE doubled per MI355X_MICROARCH.md "
                       "(gfx950 reports half the bytes of wide coalesced reads); read from this committed file by bench.py, not measured in the run"}}
json.dump(t, open(out + "/pmc_traffic.json", "w"), indent=1)
print(t)
PY
head -n 14 "$OUT/kernel_stats.csv" | cut -c1-160
rm -rf "$OUT/stats" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"
