#!/usr/bin/env python3
"""Experiment: where two hierarchy walks disagree on the full C5 scene, which one does the oracle's linear scan side with?
    python tools/exp_c5_truth.py [--spp 16] [--variants 0,11]"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("go-raytracer_amd")
shard = importlib.import_module("go-raytracer_amd.shard")
from oracle import binding  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--spheres", type=int, default=1000000)
ap.add_argument("--variants", default="0,11")
a = ap.parse_args()
W, H = 1920, 1080
scene = pkg.scenes.spheres(a.spheres, W, H)
frames = []
for v in [int(x) for x in a.variants.split(",")]:
    ren = shard.SampleShardRenderer(scene, a.spp, 0, 1, seed=1)
    ren.opts.flags = v
    ren.render()
    torch.cuda.synchronize()
    frames.append(ren.rgb_sum.view(H, W, 3).cpu().numpy().copy())
    del ren
d = (frames[0] != frames[1]).any(axis=-1)
ys, xs = np.nonzero(d)
print("pixels that differ:", len(xs))
osc = binding.OracleScene(scene.flat)
rgb, nr = osc.TraceSamples(np.stack([xs, ys], 1), a.spp, seed=1)
for i, (x, y) in enumerate(zip(xs, ys)):
    print((int(x), int(y)), "first", frames[0][y, x], "second", frames[1][y, x], "oracle", rgb[i].sum(0), "oracle rays", int(nr[i].sum()))
