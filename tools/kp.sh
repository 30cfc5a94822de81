#!/bin/bash
# kernel A/B on the GPU box: the three probe workloads, digests checked against tools/probes/kernel_probe_digests.json
out=${1:-gpurun_out/kp/run}
mkdir -p "$(dirname "$out")"
python tools/kernel_probe.py > "$out.22.json" 2> "$out.err"; a=$?
python tools/kernel_probe.py --contig 1 > "$out.1.json" 2>> "$out.err"; b=$?
python tools/kernel_probe.py --tech hifi --depth 60 > "$out.22h.json" 2>> "$out.err"; c=$?
cat "$out.22.json" "$out.1.json" "$out.22h.json"
exit $((a + b + c))
