#!/usr/bin/env python3
"""Where the hierarchy overtakes the reference's linear object loop (scene.go:173): random
spheres + boxes inside the Cornell room, 960x540, 64 spp, linear scan vs GRT_ACCEL_BVH.
    python tools/bench_crossover.py [--sizes 16,32,64,...]
Prints one JSON line per size."""
import argparse
import importlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("go-raytracer_amd")
T = pkg.tracer
ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="16,32,64,128,256,512,1024,4096")
ap.add_argument("--spp", type=int, default=64)
a = ap.parse_args()
W, H = 960, 540
torch.cuda.init()
for n in [int(x) for x in a.sizes.split(",")]:
    objs = pkg.scenes.random_objects(n // 2, n - n // 2 - 6, seed=11, diffuse=False)
    row = {"objects": len(objs)}
    for name, accel in (("linear", pkg._capi.GRT_ACCEL_LINEAR), ("bvh", pkg._capi.GRT_ACCEL_BVH)):
        sc = T.Scene(W, H, pkg.scenes.cornell_camera(W / H), objs, accel=accel)
        sc.RenderAccum(4, seed=1)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            f = sc.RenderAccum(a.spp, seed=1)
            best = min(best, time.perf_counter() - t0)
        row[name + "_ms"] = round(best * 1e3, 2)
        row[name + "_mean"] = float(f.mean())
        sc.close()
    print(json.dumps(row), flush=True)
