#!/bin/bash
# Tuning sweep for the photon gather (run on the GPU box): waves per workgroup, queries per wave, counters.
set -e
for w in 4 8 16; do for q in 8 16 32; do
  echo "== GRT_GATHER_WAVES=$w GRT_GATHER_QPW=$q"
  GRT_GATHER_WAVES=$w GRT_GATHER_QPW=$q timeout -k 10 120 python tools/bench_photon.py --cpu-queries 0 --rounds 5 | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print({k:(round(v['ms'],3)) for k,v in d['gather'].items()})"
done; done
echo "== stats (default)"
GRT_GATHER_STATS=1 timeout -k 10 120 python tools/bench_photon.py --cpu-queries 0 --rounds 1 2>&1 | grep "gather stats" | sort | uniq
