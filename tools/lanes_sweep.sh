#!/bin/bash
# step time against the number of lanes: tools/lanes_sweep.sh OUT "3 4 6" [bench args]
out=$1; lanes=$2; shift 2
mkdir -p "$(dirname "$out")"
for l in $lanes; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-legs --steps 8 --warmup 2 --lanes $l "$@" 2> "$out.err" | tee -a "$out.json" | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('lanes $l', round(d['ms_per_step'],2), (d.get('host_cpu') or {}).get('pinned_to'), {k: round(v,2) for k,v in d['stage_ms_per_step_rank0'].items() if v > 0.3}, {k: round(v,2) for k,v in d['kernel_ms_per_step_rank0'].items() if v > 0.5})"
done
