#!/usr/bin/env python3
"""A/B of the hierarchy render kernels at C5 (opts.flags variants): frames must be bit-identical, times differ.
    python tools/ab_c5.py [--variants 0,8] [--spp 256] [--spheres 1000000]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("go-raytracer_amd")
shard = importlib.import_module("go-raytracer_amd.shard")
ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0,8")
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--spheres", type=int, default=1000000)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
scene = pkg.scenes.spheres(a.spheres, a.width, a.height)
ren = shard.SampleShardRenderer(scene, a.spp, 0, 1, seed=1)
ref = None
for v in [int(x) for x in a.variants.split(",")]:
    ren.opts.flags = v
    times = []
    for r in range(a.rounds):
        ren.ray_count.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ren.render()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    c = ren.ray_count.tolist()
    if ref is None:
        ref = ren.rgb_sum.clone()
    same = bool((ren.rgb_sum == ref).all())
    times.sort()
    print(f"variant {v}: {times[len(times)//2]:.2f} ms (min {times[0]:.2f}), rays {c[0]}, counters {c[1:11]}, frame == first variant's: {same}, "
          f"max |diff| {(ren.rgb_sum - ref).abs().max().item():.3e}")
