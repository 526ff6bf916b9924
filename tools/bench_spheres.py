#!/usr/bin/env python3
"""Config C5: synthetic N-sphere scene (SURVEY.md §8d `spheres1M`) at 1920x1080, 256 spp, through
the bounding-volume hierarchy.  `--gpus N` (N > 1: the script starts its N ranks itself, or runs under
torch.distributed.run) shards the SAMPLE range over the ranks and sums the FP32 frames with one RCCL reduce.
    python tools/bench_spheres.py [--spheres 1000000] [--spp 256] [--steps 3] [--gpus N]
Prints one JSON line (rank 0)."""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--spheres", type=int, default=1000000)
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--cpu-spheres", type=int, default=10000)
ap.add_argument("--chunk-spp", type=int, default=0, help="samples per wave item (0 = the library's choice)")
ap.add_argument("--as-boxes", action="store_true", help="every sphere exported as its bounding AABB (SURVEY.md §8d)")
a = ap.parse_args()
_rc = importlib.import_module("go-raytracer_amd.launch").self_launch(__file__, sys.argv[1:], a.gpus)   # before any GPU call
if _rc is not None:
    raise SystemExit(_rc)

import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

pkg = importlib.import_module("go-raytracer_amd")
shard = importlib.import_module("go-raytracer_amd.shard")
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
if world != a.gpus:
    raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
dev = torch.device("cuda", torch.cuda.current_device())
grouped = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ   # one rank under a launcher still forms its group (RCCL path as with N)
if grouped:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=dev)
W, H = a.width, a.height
t0 = time.perf_counter()
scene = pkg.scenes.spheres(a.spheres, W, H, as_boxes=a.as_boxes)
t_create = time.perf_counter() - t0
t0 = time.perf_counter()
ren = shard.SampleShardRenderer(scene, a.spp, rank, world, seed=1, chunk_spp=a.chunk_spp, device=dev)   # stages the scene: BVH build + upload
t_stage = time.perf_counter() - t0
torch.cuda.synchronize()
times = []
for k in range(a.steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    w0 = time.perf_counter()
    e0.record()
    ren.render()
    e1.record()
    frame = ren.reduce(0)
    torch.cuda.synchronize()
    times.append((time.perf_counter() - w0, e0.elapsed_time(e1)))
wall = sorted(t[0] for t in times)[len(times) // 2]
kms = sorted(t[1] for t in times)[len(times) // 2]
rays = float(ren.ray_count[0].item()) / a.steps
# executed tests: one more frame through the counting build of the kernel
ren.ray_count.zero_()
ren.opts.flags = 6
ren.render()
torch.cuda.synchronize()
c_rays, c_inner, c_objs = (float(v) for v in ren.ray_count[:3].tolist())
c_sched = [float(v) for v in ren.ray_count[3:11].tolist()]
ren.opts.flags = 0
if grouped:
    st = torch.tensor([wall, rays], dtype=torch.float64, device=dev)
    mx = st.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    sm = st.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    wall, rays = mx[0].item(), sm[1].item()
if rank == 0:
    out = {"workload": f"C5: {a.spheres} {'sphere-bounding boxes' if a.as_boxes else 'spheres'} + light box + floor, {W}x{H}, {a.spp} spp, depth cap 6, seed 1, BVH",
           "n_gpus": world, "sharding": "sample range per rank, RCCL reduce(sum) of the FP32 frame" if world > 1 else "none",
           "scene_create_s": t_create, "stage_s_incl_bvh_build_and_first_frame": t_stage,
           "frame_s": wall, "kernel_ms_rank0": kms, "Msamples_per_s": W * H * a.spp / wall / 1e6,
           "rays_per_s": rays / wall, "rays_per_sample": rays / (W * H * a.spp),
           "mean_radiance": frame.mean((0, 1)).div(a.spp).tolist()}
    if c_rays > 0:
        # per ray, from device counters: inner nodes visited (32 B, two box tests of 25 flop) and objects tested
        # (32 B record, 27 flop for a sphere); rates use this run's rays/s
        rps = rays / wall
        fo = 25 if a.as_boxes else 27
        out["executed_tests"] = {"inner_nodes_per_ray": c_inner / c_rays, "objects_per_ray": c_objs / c_rays,
                                 "flop_per_ray": (50 * c_inner + fo * c_objs) / c_rays,
                                 "bytes_per_ray": 32 * (c_inner + c_objs) / c_rays,
                                 "executed_TFLOPs": (50 * c_inner + fo * c_objs) / c_rays * rps / 1e12,
                                 "fetched_GBps": 32 * (c_inner + c_objs) / c_rays * rps / 1e9,
                                 "schedule": {"wave_steps_inner_leaf_shade": c_sched[0:3],
                                              "lanes_per_step_inner_leaf_shade": [c_sched[3 + k] / max(1.0, c_sched[k]) for k in range(3)],
                                              "lanes_with_work_per_round": c_sched[7] / max(1.0, c_sched[6])},
                                 "note": "rank 0's sample range, counting build (opts.flags = 6); the reference would run "
                                         f"{a.spheres + 2} object tests per ray (scene.go:173)"}
    if a.cpu_spheres > 0 and world == 1:
        from oracle import binding as ob
        cw, ch, cspp = 320, 180, 8
        threads = importlib.import_module("bench").host_cpu_share()[0] + 1   # NumCPU()+1 workers (scene.go:106)
        o = ob.OracleScene(pkg.scenes.spheres_flat(a.cpu_spheres, cw, ch))
        t0 = time.perf_counter()
        o.RenderAccum(cspp, seed=1, nthreads=threads)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"Msamples_per_s": cw * ch * cspp / dt / 1e6, "cores": threads, "kind": "port",
                               "sample": f"{a.cpu_spheres} spheres (linear scan as scene.go:173), {cw}x{ch}, {cspp} spp, {dt:.1f} s",
                               "extrapolated_to_scene_Msamples_per_s": cw * ch * cspp / dt / 1e6 * a.cpu_spheres / a.spheres,
                               "note": "the reference's cost is linear in the object count; extrapolation is x N_cpu/N"}
    out["build"] = {"src_sha16": pkg._capi.source_hash()}
    out["process_group"] = {"backend": dist.get_backend(), "world": dist.get_world_size()} if grouped else None
    print(json.dumps(out))
if grouped:
    dist.barrier()
    dist.destroy_process_group()
