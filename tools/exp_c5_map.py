#!/usr/bin/env python3
"""Experiment: where do the idle lanes of k_pathtrace_bvh come from?  Needs a library built with -DGRT_COUNT_MAP
(make -C go-raytracer_amd/csrc FLAGS="... -DGRT_COUNT_MAP"): the counting build (opts.flags = 6) then writes every
pixel's inner-node steps, object tests and rays into the frame.  Prints, per 8x8 tile (= one wave), how the lanes'
work compares with the busiest lane's — what a wave that keeps lane = pixel can reach at best."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("go-raytracer_amd")
shard = importlib.import_module("go-raytracer_amd.shard")
W, H = 1920, 1080
scene = pkg.scenes.spheres(1000000, W, H)
for spp, chunk in ((256, 256), (8, 8)):
    ren = shard.SampleShardRenderer(scene, spp, 0, 1, seed=1, chunk_spp=chunk)
    ren.opts.flags = 6
    ren.ray_count.zero_()
    ren.render()
    torch.cuda.synchronize()
    m = ren.rgb_sum.cpu().numpy().reshape(H, W, 3)
    steps = m[..., 0] + m[..., 1] + m[..., 2]            # rounds in which the lane had work of its own
    t = steps.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(H // 8, W // 8, 64)
    live = t.max(-1) > 0
    mean, mx = t.mean(-1)[live], t.max(-1)[live]
    print(f"spp {spp} chunk {chunk}: tiles with work {live.sum()} of {live.size}; sum(mean)/sum(max) = {mean.sum() / mx.sum():.3f}; "
          f"median tile mean/max = {np.median(mean / mx):.3f}; rays/sample/pixel min {m[..., 2][m[..., 2] > 0].min() / spp:.2f} "
          f"mean {m[..., 2][m[..., 2] > 0].mean() / spp:.2f} max {m[..., 2].max() / spp:.2f}")
    if spp == 256:
        np.savez_compressed("gpurun_out/c5_map.npz", inner=m[..., 0].astype(np.float32), objs=m[..., 1].astype(np.float32),
                            rays=m[..., 2].astype(np.float32))
