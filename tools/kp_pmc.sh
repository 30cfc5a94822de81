#!/bin/bash
# Shader-engine counters of the probe's kernels (one rocprofv3 --pmc pass per counter group): tools/kp_pmc.sh OUTDIR [kernel_probe args]
OUT=$(realpath -m "$1"); shift; mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr"; do
    i=$((i + 1))
    timeout -k 10 280 rocprofv3 --pmc $grp --output-format csv -d "$OUT/p$i" -- python3 "$REPO/tools/kernel_probe.py" --steps 5 "$@" > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -n 5 "$OUT/p$i.log"; }
    echo "pass $i done"
done
cd "$REPO"
python3 tools/pmc_summary.py "$OUT/sq_summary.json" --all "$OUT"
python3 - "$OUT/sq_summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "depth_tile" in k or "cigar_scan" in k or "depth_items" in k:
        print(k)
        for c in sorted(v): print(f"   {c:32s} {v[c]:16.1f}")
PY
