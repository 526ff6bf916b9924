#!/usr/bin/env python3
"""Interleaved A/B timing of k_pathtrace variants in ONE process (opts.flags bits 0-3).
    python tools/ab_pathtrace.py [--variants 0,1,2] [--rounds 5] [--spp 1000] [--chunks 0,..]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("go-raytracer_amd")
shard = importlib.import_module("go-raytracer_amd.shard")

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0,1,2")
ap.add_argument("--chunks", default="0")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--spp", type=int, default=1000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
scene = pkg.scenes.cornell(a.width, a.height)
rens = {}
for v in [int(x) for x in a.variants.split(",")]:
    for c in [int(x) for x in a.chunks.split(",")]:
        r = shard.ShardRenderer.__new__(shard.ShardRenderer)
        # build by hand so that the variant flag is set before the warm-up launch
        tmp = shard.ShardRenderer(scene, a.spp, chunk_spp=c) if False else None
        ren = shard.ShardRenderer(scene, 1, chunk_spp=c)  # cheap staging launch
        ren.samples = a.spp
        ren.opts.flags = v
        import ctypes as C
        nbytes = int(ren.lib.grt_accum_workspace_bytes(scene.handle, C.byref(ren.opts), 0, 1, a.spp))
        ren.workspace = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=ren.device)
        ren.workspace_bytes = nbytes
        rens[(v, c)] = ren
times = {k: [] for k in rens}
ref = None
for rnd in range(a.rounds + 1):
    for k, ren in rens.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ren.ray_count.zero_()
        e0.record()
        ren.render()
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            times[k].append(e0.elapsed_time(e1))
        if ref is None:
            ref = ren.rgb_sum.clone()
        elif rnd == 1:
            d = (ren.rgb_sum - ref).abs().max().item()
            print(f"variant {k}: max |diff| vs first = {d:.3e}, rays {int(ren.ray_count.item())}")
n = a.width * a.height * a.spp
for k, t in times.items():
    t.sort()
    print(f"variant {k[0]} chunk {k[1]}: median {t[len(t)//2]:.2f} ms  min {t[0]:.2f} ms  -> {n / t[len(t)//2] / 1e6:.2f} Gsamples/s")
