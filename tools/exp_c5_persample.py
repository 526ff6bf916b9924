#!/usr/bin/env python3
"""Experiment: per-sample values of chosen C5 pixels from the queue kernels (one frame per sample) against
grt_trace_samples' per-lane walk (gpurun_out/px_cull.npz)."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("go-raytracer_amd")
shard = importlib.import_module("go-raytracer_amd.shard")
PIXELS = [(1123, 152), (706, 164), (1309, 165), (763, 171), (1017, 178), (1158, 186), (1124, 229), (1086, 245), (814, 270),
          (747, 283), (1162, 290), (886, 297), (815, 394), (637, 416), (1106, 418), (1246, 508)]
scene = pkg.scenes.spheres(1000000, 1920, 1080)
truth, nr = scene.TraceSamples(np.array(PIXELS, np.int32), 16, seed=1)
for s in ([int(sys.argv[2])] if len(sys.argv) > 2 else range(16)):
    ren = shard.SampleShardRenderer(scene, 16, s, 16, seed=1)
    for v in [int(x) for x in sys.argv[1].split(",")]:
        ren.opts.flags = v
        print('--- variant', v, flush=True)
        ren.render()
        torch.cuda.synchronize()
        f = ren.rgb_sum.cpu().numpy()
        for i, (x, y) in enumerate(PIXELS):
            if not np.allclose(f[y, x], truth[i, s], rtol=1e-5, atol=1e-6):
                print(f"sample {s} variant {v} pixel {(x, y)}: kernel {f[y, x]} per-lane walk {truth[i, s]} ({nr[i, s]} rays)", flush=True)
    del ren
