#!/usr/bin/env python3
"""Experiment: how many lane-levels of k_pathtrace<8> hold a live path (the end-of-chunk drain)?
Needs a library built with -DGRT_COUNT_LEVELS (make -C go-raytracer_amd/csrc FLAGS="... -DGRT_COUNT_LEVELS"):
ray_count[1] then receives 64 x the trips of every wave's loop.  C3: 92.7 %."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("go-raytracer_amd")
shard = importlib.import_module("go-raytracer_amd.shard")
scene = pkg.scenes.cornell(1920, 1080)
ren = shard.SampleShardRenderer(scene, 1000, 0, 1, seed=1)
ren.ray_count.zero_()
ren.render()
torch.cuda.synchronize()
c = ren.ray_count.tolist()
print("rays", c[0], "lane-levels", c[1], "alive fraction", c[0] / max(1, c[1]))
