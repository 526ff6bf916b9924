set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
timeout -k 10 200 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -q --timeout 120 --timeout-method thread 2>&1 | tail -1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 500 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
