#!/bin/bash
# Extra PMC passes (texture addresser / vector L1) for one command, run on the GPU box from the repo root:
#   bash tools/pmc_extra.sh <outdir> <script.py> [args...]
set -u
OUT=$1; shift
SCRIPT=$1; shift
mkdir -p "$OUT"
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
ARGS=("$@")
run() {
  local name=$1; shift
  timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- python3 "$ROOT/$SCRIPT" "${ARGS[@]}" > "$ROOT/$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU || exit 1
run ta TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE || exit 1
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum || exit 1
