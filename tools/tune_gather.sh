#!/bin/bash
# Tuning sweep of k_gather_knn (run on the GPU box): queries per wave, waves per workgroup.
for w in 16 8; do for q in 8 16 24 32; do
  echo "== waves $w qpw $q"
  GRT_GATHER_WAVES=$w GRT_GATHER_QPW=$q timeout -k 10 120 python tools/bench_photon.py --cpu-queries 0 --rounds 5 --render-spp 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:round(v['ms'],4) for k,v in d['gather'].items()})" || exit 1
done; done
