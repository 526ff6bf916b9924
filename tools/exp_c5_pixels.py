#!/usr/bin/env python3
"""Experiment: per-sample radiance and ray counts of chosen C5 pixels through grt_trace_samples (the per-lane walk of the
binary hierarchy).  With a library built with -DGRT_NOCULL the walk visits every object: the FP32 linear scan.
    python tools/exp_c5_pixels.py OUT.npz"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("go-raytracer_amd")
PIXELS = [(1123, 152), (706, 164), (1309, 165), (763, 171), (1017, 178), (1158, 186), (1124, 229), (1086, 245), (814, 270),
          (747, 283), (1162, 290), (886, 297), (815, 394), (637, 416), (1106, 418), (1246, 508)]
scene = pkg.scenes.spheres(1000000, 1920, 1080)
rgb, nr = scene.TraceSamples(np.array(PIXELS, np.int32), 16, seed=1)
np.savez(sys.argv[1], rgb=rgb, nr=nr)
for p, c, n in zip(PIXELS, rgb.sum(1), nr.sum(1)):
    print(p, c, int(n))
