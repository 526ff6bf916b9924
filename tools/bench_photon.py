#!/usr/bin/env python3
"""Config C4: photon emission + kd-tree build + k-nearest radiance gather at 1920x1080.
    python tools/bench_photon.py [--photons 1000000] [--k 100] [--rounds 5] [--cpu-queries 20000]
Prints one JSON line.  Roofline: algorithmic bytes per query = k * 28 + 36."""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("go-raytracer_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--photons", type=int, default=1000000)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--cpu-queries", type=int, default=20000)
ap.add_argument("--render-spp", type=int, default=16, help="photon-mapped trace() (scene.go:308-320) at this spp; 0 = skip")
ap.add_argument("--only", choices=("all", "global", "caustic"), default="all",
                help="gather launches of one map only (and no photon-mapped render): under rocprofv3 the k_gather_knn row of "
                     "kernel_stats.csv is then that launch by itself")
a = ap.parse_args()
W, H = a.width, a.height
lib = pkg._capi.load()
scene = pkg.scenes.cornell(W, H, "cornell8d")

pkg.tracer.PhotonMap(4096).build(scene, caustics=False, seed=2).close()   # loads the code objects (first-call cost, ~0.15 s)
t0 = time.perf_counter()
gmap = pkg.tracer.PhotonMap(a.photons).build(scene, caustics=False, seed=1)
t_global = time.perf_counter() - t0
t0 = time.perf_counter()
cmap = pkg.tracer.PhotonMap(a.photons // 2).build(scene, caustics=True, seed=1)
t_caustic = time.perf_counter() - t0

# one query per primary hit on a diffuse wall (objects 1..5), position rebuilt from t
ids, t, nrm, _ = scene.PrimaryHits()
hit = (ids >= 1) & (ids <= 5)
ys, xs = np.nonzero(hit)
c = scene.flat[0].cam
cam = np.array(c.origin)
u, v = (xs + 0.5) / W, (ys + 0.5) / H
D = np.array(c.lower_left)[None] + np.array(c.horizontal)[None] * u[:, None] + np.array(c.vertical)[None] * v[:, None] - cam[None]
P = (cam[None] + D * t[hit].astype(np.float64)[:, None]).astype(np.float32)
N = nrm[hit].astype(np.float32)
nq = P.shape[0]
dP, dN = torch.from_numpy(P).cuda(), torch.from_numpy(N).cuda()
out = torch.empty_like(dP)
ws = torch.empty(max(16, int(lib.grt_photon_gather_workspace_bytes(nq))), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
res = {}
if a.only != "all":
    a.render_spp = 0
for name, pm, radius in (("global_r0", gmap, 0.0), ("caustic_r1", cmap, 1.0)):
    if a.only != "all" and not name.startswith(a.only):
        continue
    times = []
    for r in range(a.rounds + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pkg._capi.check(lib.grt_photon_gather_dev(pm.handle, dP.data_ptr(), dN.data_ptr(), nq, radius, a.k, out.data_ptr(),
                                                  ws.data_ptr(), stream))
        e1.record()
        torch.cuda.synchronize()
        if r:
            times.append(e0.elapsed_time(e1))
    times.sort()
    ms = times[len(times) // 2]
    bytes_q = a.k * 28 + 36
    res[name] = {"ms": ms, "queries_per_s": nq / ms * 1e3, "mean_irradiance": out.mean(0).tolist()}
    if radius == 0.0:   # SURVEY.md §8d's convention prices k photons per query; with a radius most queries accept far fewer
        res[name].update({"algorithmic_GBps": nq * bytes_q / ms / 1e6, "frac_of_8TBps": nq * bytes_q / ms / 1e6 / 8000.0})

cpu = None
if a.cpu_queries > 0:
    from oracle import binding as ob
    gp, gw, gt = gmap.read()
    o = ob.OraclePhotonMap.load(gp.astype(np.float64), gw.astype(np.float64), gt)
    sel = np.random.RandomState(0).choice(nq, min(a.cpu_queries, nq), replace=False)
    threads = importlib.import_module("bench").host_cpu_share()[0] + 1   # NumCPU()+1 workers (scene.go:106)
    t0 = time.perf_counter()
    o.gather(P[sel].astype(np.float64), N[sel].astype(np.float64), 0.0, a.k, nthreads=threads)
    dt = time.perf_counter() - t0
    cpu = {"queries_per_s": len(sel) / dt, "cores": threads, "kind": "port", "sample": f"{len(sel)} of the {nq} queries, k={a.k}, {dt:.2f} s"}
    # mapPhotons + ScalePhotonPower (scene.go:140-165, photon_mapping.go:165-209) on the CPU: single-threaded as in the
    # reference (one goroutine, one rnd), 10^5 stored photons
    nb = 100000
    t0 = time.perf_counter()
    om = ob.OraclePhotonMap.build(ob.OracleScene(scene.flat), nb, caustics=False, seed=1)
    dtb = time.perf_counter() - t0
    cpu["build"] = {"stored_photons_per_s": om.count() / dtb, "cores": 1, "sample": f"{om.count()} stored photons ({om.emitted()} emitted) + balanced tree, {dtb:.2f} s"}

pm = None
if a.render_spp > 0:
    o = scene.opts(seed=1)
    nbytes = int(lib.grt_render_pm_workspace_bytes(scene.handle, C.byref(o), 0, 1, a.render_spp))
    wsp = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    frame = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    tt = []
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pkg._capi.check(lib.grt_render_accum_pm_dev(scene.handle, C.byref(o), gmap.handle, cmap.handle, 0, 1, a.render_spp,
                                                    frame.data_ptr(), wsp.data_ptr(), nbytes, None, stream))
        e1.record()
        torch.cuda.synchronize()
        tt.append(e0.elapsed_time(e1))
    tt.sort()
    pm = {"spp": a.render_spp, "ms": tt[1], "Msamples_per_s": W * H * a.render_spp / tt[1] / 1e3,
          "mean_radiance": (frame.mean((0, 1)) / a.render_spp).tolist(),
          "note": "path trace + one gather request per path that ends on a diffuse surface x 2 maps (k=100; radius 0 / 1) + accumulate"}

print(json.dumps({
    "workload": f"C4: cornell8d, {a.photons} global + {a.photons // 2} caustic photons, k={a.k}, {nq} primary-hit queries at {W}x{H}",
    "emission": {"global_s": t_global, "global_emitted": gmap.emitted(), "global_stored": gmap.count(),
                 "caustic_s": t_caustic, "caustic_emitted": cmap.emitted(), "caustic_stored": cmap.count(),
                 "global_stored_per_s": gmap.count() / t_global, "caustic_emitted_per_s": cmap.emitted() / t_caustic,
                 "note": "wall time of grt_photon_build (synchronous): emission + scan/scatter batches, then the device tree build "
                         "(one radix sort per level); photons never leave the device"},
    "gather": res, "photon_mapped_render": pm, "cpu_baseline": cpu, "build": {"src_sha16": pkg._capi.source_hash()}}))
