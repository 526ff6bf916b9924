#!/usr/bin/env python3
"""How long does ONE rank's share of the C3 frame take (tile sharding over G ranks), per chunk size?  Strong scaling
is only as good as shard 0's time against frame_time / G.   python tools/exp_shard_time.py [--gpus 8]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("go-raytracer_amd")
shard = importlib.import_module("go-raytracer_amd.shard")
ap = argparse.ArgumentParser()
ap.add_argument("--gpus", default="1,2,4,8")
ap.add_argument("--chunks", default="0,16,32,64,125,250")
a = ap.parse_args()
scene = pkg.scenes.cornell(1920, 1080)
for G in [int(g) for g in a.gpus.split(",")]:
    for c in [int(x) for x in a.chunks.split(",")]:
        worst = 0.0
        for r in sorted({0, G - 1, G // 2}):
            ren = shard.ShardRenderer(scene, 1000, r, G, seed=1, chunk_spp=c)
            ts = []
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ren.render()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            worst = max(worst, sorted(ts)[1])
        print(f"G={G} chunk_spp={c or 'auto'}: slowest of ranks 0, G/2, G-1: {worst:.3f} ms  x G = {worst * G:.2f} ms", flush=True)
