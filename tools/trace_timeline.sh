#!/bin/bash
# device timeline of the last step of a short bench run: tools/trace_timeline.sh OUTDIR [bench args]
OUT=$(realpath -m "$1"); shift; mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps 3 --warmup 2 --no-legs --no-cpu-baseline "$@" > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -n 5 "$OUT/trace.log"; exit 1; }
cd "$REPO"
python3 tools/trace_timeline.py "$OUT/trace" > "$OUT/timeline.txt" 2>&1
rm -rf "$OUT/trace"
tail -n 3 "$OUT/timeline.txt"
