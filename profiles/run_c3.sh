set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r02b; cd /tmp && export TMPDIR=/tmp
mkdir -p "$OUT"; rm -rf "$OUT/trace_c3" "$OUT/pmc_c3"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c3" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-extras > "$OUT/trace_c3.log" 2>&1 || exit 1
cd "$ROOT"
bash profiles/run_pmc.sh gpurun_out/r02b/pmc_c3 bench.py --steps 2 --warmup 0 --no-extras > "$OUT/pmc_c3.log" 2>&1 || exit 1
echo done
