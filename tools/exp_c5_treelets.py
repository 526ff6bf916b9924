#!/usr/bin/env python3
"""Experiment (library built with -DGRT_COUNT_DEPTH): node visits of the C5 walk by depth class.  If 3-level subtrees were
packed into one cache line each, a ray would fetch one line per visited subtree ROOT: the visits at depth = r (mod 3) for the
alignment r that puts the deepest inner level last.
    python tools/exp_c5_treelets.py [--spp 16]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("go-raytracer_amd")
shard = importlib.import_module("go-raytracer_amd.shard")
ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=16)
a = ap.parse_args()
ren = shard.SampleShardRenderer(pkg.scenes.spheres(1000000, 1920, 1080), a.spp, 0, 1, seed=1)
ren.opts.flags = 6
ren.ray_count.zero_()
ren.render()
torch.cuda.synchronize()
c = ren.ray_count.tolist()
rays = c[0]
print(f"rays {rays}, inner nodes per ray {c[1] / rays:.2f}, objects per ray {c[2] / rays:.2f}")
print("visits per ray at depth = 0, 1, 2 (mod 3):", [round(x / rays, 2) for x in c[11:14]], " at depth >= 17:", round(c[14] / rays, 2))
