"""grt_render_accum_noalloc_dev — the entry point that allocates nothing (include/grt.h): against
grt_render_accum_dev, through the C ABI, bit for bit, on two calls in a row, the second with a smaller input.
Collected last (test_zz_*): the library scratch keeps the previous call's contents on purpose."""
import ctypes as C
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def both(pkg, scene, samples, shard, n_shards, chunk_spp, seed=3):
    """(frame, rays) of grt_render_accum_dev with a caller workspace and of the no-alloc entry point."""
    import torch

    lib = pkg._capi.load()
    o = scene.opts(seed=seed, chunk_spp=chunk_spp)
    rows = int(lib.grt_shard_rows(scene.H, shard, n_shards))
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for noalloc in (False, True):
        frame = torch.full((rows, scene.W, 3), float("nan"), dtype=torch.float32, device="cuda")
        rays = torch.zeros(1, dtype=torch.int64, device="cuda")
        if noalloc:
            free0 = torch.cuda.mem_get_info()[0]
            pkg._capi.check(lib.grt_render_accum_noalloc_dev(scene.handle, C.byref(o), shard, n_shards, samples, frame.data_ptr(),
                                                             rays.data_ptr(), st))
            free1 = torch.cuda.mem_get_info()[0]
            assert free1 == free0, "device memory was allocated or freed during the call"
        else:
            nbytes = int(lib.grt_accum_workspace_bytes(scene.handle, C.byref(o), shard, n_shards, samples))
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
            pkg._capi.check(lib.grt_render_accum_dev(scene.handle, C.byref(o), shard, n_shards, samples, frame.data_ptr(), ws.data_ptr(),
                                                     nbytes, rays.data_ptr(), st))
        torch.cuda.synchronize()
        out.append((frame.clone(), int(rays.item())))
    return out


def test_no_alloc_entry_point_equals_the_workspace_one(pkg):
    import torch

    lib = pkg._capi.load()
    scene = pkg.scenes.cornell(320, 180)
    fresh = pkg.scenes.cornell(64, 64)                      # never rendered: not staged on the device
    o = fresh.opts(seed=1)
    buf = torch.zeros((64, 64, 3), dtype=torch.float32, device="cuda")
    assert lib.grt_render_accum_noalloc_dev(fresh.handle, C.byref(o), 0, 1, 4, buf.data_ptr(), None, None) == -1
    assert b"not staged" in lib.grt_last_error()
    # first call: whole frame, 64 spp in 8 chunks (5.5 MB of slabs in the library scratch)
    (a, ra), (b, rb) = both(pkg, scene, 64, 0, 1, 8)
    assert torch.isfinite(a).all() and a.max() > 0
    assert (a == b).all() and ra == rb
    # second call, smaller: shard 1 of 3, 24 spp in 3 chunks — the scratch still holds the first call's slabs
    (c, rc_), (d, rd) = both(pkg, scene, 24, 1, 3, 8)
    assert torch.isfinite(c).all() and c.max() > 0
    assert (c == d).all() and rc_ == rd
    # a hierarchy scene and one chunk (no scratch used at all)
    sph = pkg.scenes.spheres(5000, 160, 90)
    (e, re_), (f, rf) = both(pkg, sph, 4, 0, 1, 4)
    assert (e == f).all() and re_ == rf and e.max() > 0
    # automatic chunk size (chunk_spp = 0) agrees with the workspace entry point when its choice fits
    (g, _), (h, _) = both(pkg, scene, 64, 0, 1, 0)
    assert (g == h).all()


def test_sizes_beyond_the_scratch(pkg):
    """1920x1080 in 8 chunks = 199 MB of slabs > 64 MiB: an explicit chunk size is refused with a clear error,
    the automatic one is coarsened to fit (2 chunks) and renders the same image up to FP32 summation order."""
    import torch

    lib = pkg._capi.load()
    scene = pkg.scenes.cornell(1920, 1080)
    frame = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda")
    rays = torch.zeros(1, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    o = scene.opts(seed=1, chunk_spp=2)
    nbytes = int(lib.grt_accum_workspace_bytes(scene.handle, C.byref(o), 0, 1, 16))
    assert nbytes > 64 << 20
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    pkg._capi.check(lib.grt_render_accum_dev(scene.handle, C.byref(o), 0, 1, 16, frame.data_ptr(), ws.data_ptr(), nbytes, rays.data_ptr(), st))
    torch.cuda.synchronize()
    want = frame.clone()
    assert lib.grt_render_accum_noalloc_dev(scene.handle, C.byref(o), 0, 1, 16, frame.data_ptr(), rays.data_ptr(), st) == -4
    msg = lib.grt_last_error()
    assert b"chunk_spp" in msg and b"scratch" in msg
    o.chunk_spp = 0
    frame.zero_()
    pkg._capi.check(lib.grt_render_accum_noalloc_dev(scene.handle, C.byref(o), 0, 1, 16, frame.data_ptr(), rays.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.allclose(frame, want, rtol=1e-5, atol=1e-5) and frame.max() > 0


def test_no_alloc_entry_point_under_graph_capture(pkg):
    """include/grt.h names HIP graph capture as a caller: the call is captured on a side stream (an allocation, a
    synchronisation or a legacy-stream operation inside it would fail the capture), replayed twice, and the replayed
    frame equals the eager one bit for bit."""
    import torch

    lib = pkg._capi.load()
    scene = pkg.scenes.cornell(320, 180)
    (want, rays_want), _ = both(pkg, scene, 48, 0, 1, 8)            # eager; also stages the scene on the device
    o = scene.opts(seed=3, chunk_spp=8)
    frame = torch.zeros((180, 320, 3), dtype=torch.float32, device="cuda")
    rays = torch.zeros(1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    rc = None
    with torch.cuda.graph(graph, stream=side):
        rc = lib.grt_render_accum_noalloc_dev(scene.handle, C.byref(o), 0, 1, 48, frame.data_ptr(), rays.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.grt_last_error()
    torch.cuda.synchronize()
    assert float(frame.abs().max()) == 0.0                         # captured, not run
    for n in (1, 2):
        frame.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert (frame == want).all()
        assert int(rays.item()) == n * rays_want
