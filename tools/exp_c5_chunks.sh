#!/bin/bash
# C5 frame time and the schedule's lane counts against the chunk length (samples per wave item).
set -e
mkdir -p gpurun_out
for c in 8 16 32 64 128 256; do
  timeout -k 10 120 python tools/bench_spheres.py --spp 256 --steps 2 --cpu-spheres 0 --chunk-spp $c > gpurun_out/c5_chunk_$c.json
done
