set -e
mkdir -p gpurun_out
rm -f gpurun_out/c5s_*.json
run() { # name, env...
  n=$1; shift
  env "$@" timeout -k 10 120 python tools/bench_spheres.py --spp 256 --steps 3 --cpu-spheres 0 $EXTRA > gpurun_out/c5s_$n.json
}
timeout -k 10 300 python -u -m pytest tests/test_gpu_bvh.py -x -q --timeout 120 --timeout-method thread > gpurun_out/bvh_tests.log 2>&1 || { tail -30 gpurun_out/bvh_tests.log; exit 1; }
run sm32 GRT_BVH_SHADE_MIN=32
run sm40 GRT_BVH_SHADE_MIN=40
run sm48 GRT_BVH_SHADE_MIN=48
run sm56 GRT_BVH_SHADE_MIN=56
run sm32ir8 GRT_BVH_SHADE_MIN=32 GRT_BVH_INNER_RUN=8
run sm32ir16 GRT_BVH_SHADE_MIN=32 GRT_BVH_INNER_RUN=16
run sm40ir8 GRT_BVH_SHADE_MIN=40 GRT_BVH_INNER_RUN=8
run sm32o6 GRT_BVH_SHADE_MIN=32 GRT_BVH_OCC=6
run sm32o8 GRT_BVH_SHADE_MIN=32 GRT_BVH_OCC=8
run sm32o5 GRT_BVH_SHADE_MIN=32 GRT_BVH_OCC=5
