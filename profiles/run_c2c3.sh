#!/bin/bash
# C2 and C3 alone (kernel trace + PMC passes), for a build that changed only the small-scene kernels:  bash profiles/run_c2c3.sh <tag>
set -u
TAG=${1:-r03a}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c3" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-extras > "$OUT/trace_c3.log" 2>&1 || exit 1
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c2" -- python3 "$ROOT/tools/bench_primary.py" --rounds 10 --cpu 0 > "$OUT/trace_c2.log" 2>&1 || exit 1
cd "$ROOT"
bash profiles/run_pmc.sh gpurun_out/$TAG/pmc_c3 bench.py --steps 2 --warmup 0 --no-extras > "$OUT/pmc_c3.log" 2>&1 || exit 1
bash profiles/run_pmc.sh gpurun_out/$TAG/pmc_c2 tools/bench_primary.py --rounds 2 --cpu 0 > "$OUT/pmc_c2.log" 2>&1 || exit 1
echo done
