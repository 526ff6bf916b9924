#!/bin/bash
# Round profile set (run on the GPU box from the repo root):  bash profiles/run_all.sh <tag>
# Kernel-trace summaries for the four measured configs, then PMC passes for every config.
# Results under gpurun_out/<tag>/; profiles/collect.py copies the summaries into profiles/.
set -u
TAG=${1:-r03c}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
trace() { # name script args...
  local name=$1; shift
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$name" -- python3 "$@" > "$OUT/trace_$name.log" 2>&1
  local rc=$?
  echo "trace $name rc=$rc"
  return $rc
}
trace c3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-extras || exit 1
trace c2 "$ROOT/tools/bench_primary.py" --rounds 10 --cpu 0 || exit 1
trace c4 "$ROOT/tools/bench_photon.py" --cpu-queries 0 --rounds 5 --only global || exit 1   # k_gather_knn row = the global-map gather alone
trace c4c "$ROOT/tools/bench_photon.py" --cpu-queries 0 --rounds 5 --only caustic || exit 1
trace c4pm "$ROOT/tools/bench_photon.py" --cpu-queries 0 --rounds 1 || exit 1                 # incl. the photon-mapped render
trace c5 "$ROOT/tools/bench_spheres.py" --steps 2 --cpu-spheres 0 || exit 1
cd "$ROOT"
bash profiles/run_pmc.sh gpurun_out/$TAG/pmc_c3 bench.py --steps 2 --warmup 0 --no-extras > "$OUT/pmc_c3.log" 2>&1 || exit 1
bash profiles/run_pmc.sh gpurun_out/$TAG/pmc_c4 tools/bench_photon.py --cpu-queries 0 --rounds 2 --only global > "$OUT/pmc_c4.log" 2>&1 || exit 1
bash profiles/run_pmc.sh gpurun_out/$TAG/pmc_c2 tools/bench_primary.py --rounds 2 --cpu 0 > "$OUT/pmc_c2.log" 2>&1 || exit 1
bash profiles/run_pmc.sh gpurun_out/$TAG/pmc_c5 tools/bench_spheres.py --spp 16 --steps 1 --cpu-spheres 0 > "$OUT/pmc_c5.log" 2>&1 || exit 1
echo done
