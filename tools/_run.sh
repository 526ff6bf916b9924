set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/gpu_tests.log; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/bench_c3.json
