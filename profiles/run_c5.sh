#!/bin/bash
# C5 alone (kernel trace + PMC passes), for a build that changed only the hierarchy kernel:  bash profiles/run_c5.sh <tag>
set -u
TAG=${1:-r02c}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c5" -- python3 "$ROOT/tools/bench_spheres.py" --steps 2 --cpu-spheres 0 > "$OUT/trace_c5.log" 2>&1 || exit 1
cd "$ROOT"
bash profiles/run_pmc.sh gpurun_out/$TAG/pmc_c5 tools/bench_spheres.py --spp 16 --steps 1 --cpu-spheres 0 > "$OUT/pmc_c5.log" 2>&1 || exit 1
echo done
