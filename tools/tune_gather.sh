#!/bin/bash
# Tuning sweep of the carried-pool gather (run on the GPU box): margin of the pooled ball, queries per wave.
for m in 0.15 0.25 0.35; do for q in 24 48; do
  echo "== margin $m qpw $q"
  GRT_GATHER_MARGIN=$m GRT_GATHER_QPW=$q timeout -k 10 120 python tools/bench_photon.py --cpu-queries 0 --rounds 3 --only global | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['gather'])" || exit 1
  GRT_GATHER_MARGIN=$m GRT_GATHER_QPW=$q GRT_GATHER_STATS=1 timeout -k 10 120 python tools/bench_photon.py --cpu-queries 0 --rounds 1 --only global 2>&1 >/dev/null | grep "gather stats" | head -1
done; done
echo "== round-1 kernel"; GRT_GATHER_CARRY=0 timeout -k 10 120 python tools/bench_photon.py --cpu-queries 0 --rounds 3 --only global | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['gather'])"
