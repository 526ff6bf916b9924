"""The N>1 path on CPU: world_size-2 (and 3) gloo groups gather interleaved row bands into the
frame a single process produces.  Band pixels come from the oracle here (no GPU in this
test); the sharding/gather code under test is the product's go-raytracer_amd/shard.py."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

W, H, SPP = 40, 52, 2  # 52 rows = 6 full bands + one 4-row band: ragged on purpose


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(target, world, n_results):
    """Start `world` ranks, collect n_results queue items; a rank that dies before it reports (import error,
    failed assert, rendezvous trouble) fails the test instead of blocking it."""
    import queue as _q

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = []
    try:
        while len(out) < n_results:
            try:
                out.append(q.get(timeout=2.0))
            except _q.Empty:
                dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
                assert not dead, f"a rank exited with {dead} before reporting"
                assert any(p.is_alive() for p in procs) or not q.empty(), "all ranks ended without reporting"
        for p in procs:
            p.join(120)
            assert p.exitcode == 0, p.exitcode
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("go-raytracer_amd")
    shard = importlib.import_module("go-raytracer_amd.shard")
    from oracle import binding as ob

    s = ob.OracleScene(pkg.scenes.cornell_flat(W, H))
    rows = shard.band_rows(H, rank, world)
    local = torch.zeros((shard.max_shard_rows(H, world), W, 3), dtype=torch.float32)
    for i, y in enumerate(rows):  # render only the rows this rank owns
        local[i] = torch.from_numpy(s.RenderAccum(SPP, seed=5, row_begin=y, row_end=y + 1, nthreads=1)[y].astype(np.float32))
    frame = shard.gather_frame(local, H, rank, world, dst=0)
    if rank == 0:
        q.put(frame.numpy())
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_frame_equals_single_process(pkg, ob, world):
    frame, = run_ranks(_worker, world, 1)
    full = ob.OracleScene(pkg.scenes.cornell_flat(W, H)).RenderAccum(SPP, seed=5).astype(np.float32)
    assert frame.shape == (H, W, 3) and (frame == full).all()  # exact: result independent of G


def test_band_rows_match_library(pkg):
    shard = importlib.import_module("go-raytracer_amd.shard")
    lib = pkg._capi.load()
    for h in (1, 7, 8, 9, 52, 1080, 2160):
        for g in (1, 2, 3, 4, 8):
            owned = [shard.band_rows(h, r, g) for r in range(g)]
            assert sorted(sum(owned, [])) == list(range(h))           # a partition of the rows
            for r in range(g):
                assert len(owned[r]) == lib.grt_shard_rows(h, r, g)
            assert shard.max_shard_rows(h, g) == max(len(o) for o in owned)
    assert shard.band_rows(52, 1, 2)[:9] == [8, 9, 10, 11, 12, 13, 14, 15, 24]


def _sample_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("go-raytracer_amd")
    shard = importlib.import_module("go-raytracer_amd.shard")
    from oracle import binding as ob

    w, h, spp = 12, 10, 7
    s = ob.OracleScene(pkg.scenes.cornell_flat(w, h))
    xy = np.stack(np.meshgrid(np.arange(w), np.arange(h)), -1).reshape(-1, 2)
    rgb, _ = s.TraceSamples(xy, spp, seed=5)               # [pixel, sample, 3]: sample s is keyed by its absolute index
    first, count = shard.sample_range(spp, rank, world)
    ren = object.__new__(shard.SampleShardRenderer)        # the reduce step only (render() needs a GPU)
    ren.world = world
    ren.rgb_sum = torch.from_numpy(rgb[:, first:first + count].sum(1).reshape(h, w, 3).astype(np.float32))
    frame = ren.reduce(0)
    if rank == 0:
        q.put((frame.numpy(), rgb.sum(1).reshape(h, w, 3)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sample_range_reduce_equals_single_process(world):
    """Sample-range sharding (config C5): the ranks' partial frames add up to the whole frame."""
    shard = importlib.import_module("go-raytracer_amd.shard")
    for spp in (1, 7, 256):
        for g in (1, 2, 3, 8):
            parts = [shard.sample_range(spp, r, g) for r in range(g)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == spp
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(g - 1))
    (frame, full), = run_ranks(_sample_worker, world, 1)
    np.testing.assert_allclose(frame, full, rtol=1e-6, atol=1e-7)


def _slice_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = importlib.import_module("go-raytracer_amd.shard")
    n_slice = 16
    g = torch.Generator().manual_seed(100 + rank)
    count = [5, 0, 16][rank % 3]
    pos4 = torch.rand((n_slice, 4), generator=g)
    pow4 = torch.rand((n_slice, 4), generator=g)
    offs = torch.sort(torch.randperm(n_slice, generator=g)[:count]).values.to(torch.int32)
    pow4[:count, 3] = offs.view(torch.float32)                     # emission offsets inside the slice, as int bits
    pos, pw, off = shard.gather_slices(pos4, pow4, count, rank, world)
    q.put((rank, pos.numpy(), off.numpy(), pos4[:count].numpy(), offs.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_photon_slices_gather_in_emission_order(world):
    """Photon-build sharding: the ranks' stored photons concatenate in rank order with global
    emission offsets rank * slice + local offset, identically on every rank."""
    res = sorted(run_ranks(_slice_worker, world, world), key=lambda t: t[0])
    want_pos = np.concatenate([r[3] for r in res])
    want_off = np.concatenate([r[4].astype(np.int64) + r[0] * 16 for r in res])
    for r in res:
        assert (r[1] == want_pos).all() and (r[2] == want_off).all()
    assert (np.diff(want_off) > 0).all()                            # emission order
