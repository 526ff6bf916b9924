#!/usr/bin/env python3
"""A/B of two builds of libgrt_hip.so on one box: C3 frame (cornell8 1920x1080, 256 spp) and the C5 16-spp frame, each build
in a child process of its own.
    python tools/ab_lib.py go-raytracer_amd/csrc/libgrt_hip.so /path/to/other.so [...]"""
import subprocess
import sys

CHILD = r"""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
capi = importlib.import_module("go-raytracer_amd._capi")
capi.LIB_PATH = sys.argv[1]
pkg = importlib.import_module("go-raytracer_amd")
shard = importlib.import_module("go-raytracer_amd.shard")
def timed(ren, rounds):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ren.render(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]
ren = shard.ShardRenderer(pkg.scenes.cornell(1920, 1080), 256, 0, 1, seed=1)
print("C3 median %.3f ms, min %.3f ms" % timed(ren, 9), "checksum", float(ren.rgb_sum.double().sum()))
del ren
ren = shard.SampleShardRenderer(pkg.scenes.spheres(1000000, 1920, 1080), 16, 0, 1, seed=1)
print("C5/16 median %.3f ms, min %.3f ms" % timed(ren, 5), "checksum", float(ren.rgb_sum.double().sum()))
"""
for lib in sys.argv[1:]:
    print("==", lib, flush=True)
    r = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
    print(r.stdout + "".join(l for l in r.stderr.splitlines(True) if "amdgpu.ids" not in l), flush=True)
    if r.returncode != 0:
        sys.exit(r.returncode)
