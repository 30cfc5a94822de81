#!/bin/bash
# does the cgroup's CPU quota throttle the step? cpu.stat before / after a short bench run, for a few host pool sizes
for ht in "" 12 8 6; do
  a=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' ')
  if [ -n "$ht" ]; then export CSV_HOST_THREADS=$ht; fi
  python bench.py --no-cpu-baseline --no-legs --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('host_threads', '${ht:-default}', 'ms_per_step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms_per_step_rank0'].items() if v > 0.3})"
  b=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' ')
  echo "   cpu.stat before: $a"; echo "   cpu.stat after:  $b"
done
