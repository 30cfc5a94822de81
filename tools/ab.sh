#!/bin/bash
# Same-box A/B of library builds: tools/ab.sh <rounds> <variant> [<variant> ...]; a variant is a directory _variants/<name>/ holding
# libcsvgpu.so + libcontextsv_host.so (copy contextsv_amd/lib/*.so there after a build). Runs the probe workloads round-robin.
rounds=$1; shift
mkdir -p gpurun_out/ab
cp contextsv_amd/lib/libcsvgpu.so /tmp/keep_csvgpu.so; cp contextsv_amd/lib/libcontextsv_host.so /tmp/keep_host.so
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp _variants/$v/*.so contextsv_amd/lib/
    for w in "--contig 1" "--contig 22" "--contig 22 --tech hifi --depth 60"; do
      python tools/kernel_probe.py $w 2> gpurun_out/ab/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$v', d['workload'], 'scan', d['kernel_ms']['cigar_scan'], 'depth', d['kernel_ms']['depth'], 'pipe', d['ms_per_pipeline'], 'eq', d['digest_equal'])"
    done
  done
done | tee gpurun_out/ab/result.txt
cp /tmp/keep_csvgpu.so contextsv_amd/lib/libcsvgpu.so; cp /tmp/keep_host.so contextsv_amd/lib/libcontextsv_host.so
