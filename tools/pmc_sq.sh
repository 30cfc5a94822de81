#!/bin/bash
# Shader-engine counters of the hot kernels, one rocprofv3 --pmc pass per counter group (never combined with traces).
# usage (on the GPU box, from the repo root): bash tools/pmc_sq.sh OUTDIR
set -e
OUT=$(realpath "$1"); mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --lanes 1 --no-two-lanes --no-cpu-baseline --no-from-file --steps 20 --warmup 2"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr"; do
    i=$((i + 1))
    timeout -k 10 280 rocprofv3 --pmc $grp --output-format csv -d "$OUT/p$i" -- python3 $ARGS > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -n 5 "$OUT/p$i.log"; }
    echo "pass $i done"
done
cd "$REPO"
python3 tools/pmc_summary.py "$OUT/sq_summary.json" --all "$OUT"
