#!/usr/bin/env python3
"""Experiment: cost per ray of the specular branches.  Same geometry, spheres Lambert vs mirror+glass."""
import importlib, os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("go-raytracer_amd")
shard = importlib.import_module("go-raytracer_amd.shard")
T = pkg.tracer
W, H, S = 1920, 1080, 200
for name in ("cornell8", "lambert_spheres", "mirror_only", "glass_only"):
    objs = pkg.scenes.cornell_objects()
    white = T.LambertMaterial((0.73, 0.73, 0.73))
    if name == "lambert_spheres":
        objs[6].Mat = white; objs[7].Mat = white
    elif name == "mirror_only":
        objs[7].Mat = white
    elif name == "glass_only":
        objs[6].Mat = white
    scene = T.NewScene(W, H, pkg.scenes.cornell_camera(W / H), objs)
    ren = shard.ShardRenderer(scene, S)
    ts = []
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ren.ray_count.zero_()
        e0.record(); ren.render(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    rays = int(ren.ray_count.item())
    print(f"{name:16s} {ts[1]:7.2f} ms  rays {rays/1e9:.3f} G  {ts[1]*1e6/rays*1e3:.2f} ps/ray  rays/sample {rays/(W*H*S):.3f}")
