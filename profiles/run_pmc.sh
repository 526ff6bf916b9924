#!/bin/bash
# PMC passes for the C3 bench (run on the GPU box from the repo root):  bash profiles/run_pmc.sh <outdir>
# One rocprofv3 run per counter set (the tool does not split passes); --pmc never combined with traces.
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- python3 "$ROOT/bench.py" --steps 2 --warmup 0 --no-cpu-baseline > "$ROOT/$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES || exit 1
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_LDS || exit 1
run fetch FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
run write WRITE_SIZE GRBM_COUNT || exit 1
