#!/bin/bash
# the three forms of the scan / depth walk on the HiFi probe workloads (and ONT chr22 as the control), digests checked
out=${1:-gpurun_out/r03/forms}
mkdir -p "$(dirname "$out")"
rc=0
for f in 0 1 3; do
  for w in "--contig 22 --tech hifi --depth 60" "--contig 1 --tech hifi --depth 60"; do
    CSV_SCAN_FORM=$f python tools/kernel_probe.py $w 2>> "$out.err" | tee -a "$out.json" | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('form $f', d['workload'], 'scan', d['kernel_ms']['cigar_scan'], d['scan_frac'], 'depth', d['kernel_ms']['depth'], d['depth_frac'], 'pipe', d['ms_per_pipeline'], 'eq', d['digest_equal'], d['digest_diff'])"
    rc=$((rc + ${PIPESTATUS[0]}))
  done
done
for f in 0 1; do
  CSV_SCAN_FORM=$f python tools/kernel_probe.py --contig 22 2>> "$out.err" | tee -a "$out.json" | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('form $f', d['workload'], 'scan', d['kernel_ms']['cigar_scan'], d['scan_frac'], 'depth', d['kernel_ms']['depth'], d['depth_frac'], 'pipe', d['ms_per_pipeline'], 'eq', d['digest_equal'], d['digest_diff'])"
done
exit $rc
