#!/bin/bash
# Tuning sweep of k_gather_knn (run on the GPU box): loose-bound threshold x queries per wave; times and per-query step counts.
for q in ${QPWS:-24}; do for l in ${LOOSES:-1.6 2.5 4}; do
  echo "== qpw $q loose $l"
  GRT_GATHER_QPW=$q GRT_GATHER_LOOSE=$l GRT_GATHER_STATS=1 timeout -k 10 120 python tools/bench_photon.py --cpu-queries 0 --rounds 5 --render-spp 0 --only global 2>/tmp/g.err \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:round(v['ms'],4) for k,v in d['gather'].items()})" || exit 1
  grep "gather stats" /tmp/g.err | tail -1
done; done
