#!/bin/bash
# PMC passes for one command (run on the GPU box from the repo root):
#   bash profiles/run_pmc.sh <outdir> <script.py> [args...]
# One rocprofv3 run per counter set (the tool does not split passes); --pmc is never combined
# with traces.  python3 is the profiled program itself (no env/bash hop after `--`).
set -u
OUT=$1; shift
SCRIPT=$1; shift
mkdir -p "$OUT"
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- python3 "$ROOT/$SCRIPT" "${ARGS[@]}" > "$ROOT/$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
ARGS=("$@")
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES || exit 1
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_LDS || exit 1
run sq3 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD || exit 1
run fetch FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
run write WRITE_SIZE GRBM_COUNT || exit 1
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 1
run ta TA_TA_BUSY_sum TA_BUSY_avr || exit 1
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum || exit 1
cd "$ROOT" && python3 profiles/summarize_pmc.py "$OUT" > "$OUT/summary.json" && cat "$OUT/summary.json"
