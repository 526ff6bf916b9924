"""Tile sharding of one frame over ranks (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference's unit of work is an image row handed to a goroutine and re-ordered by a
collector (scene.go:107-132).  Here rows are dealt in interleaved bands of GRT_BAND_ROWS rows:
band b belongs to rank b mod G.  Pixels are independent (scene.go:89-97 holds no cross-pixel
state) and the counter RNG is keyed by the absolute pixel index, so the assembled frame is
bit-identical for every G.  The only exchange is one gather of the finished bands to rank 0
(W*H*3 float32 / G per rank); there is no collective inside the render.

torch is plumbing here (device memory, streams, process groups); the pixels come from
libgrt_hip.so through the C ABI.
"""
import ctypes as C

import torch
import torch.distributed as dist

from ._capi import GRT_BAND_ROWS, check, load


def band_rows(height, shard, n_shards):
    """Global row indices owned by `shard`, ascending (the order they are packed in)."""
    rows = []
    n_bands = (height + GRT_BAND_ROWS - 1) // GRT_BAND_ROWS
    for b in range(shard, n_bands, n_shards):
        rows.extend(range(b * GRT_BAND_ROWS, min(height, (b + 1) * GRT_BAND_ROWS)))
    return rows


def max_shard_rows(height, n_shards):
    return len(band_rows(height, 0, n_shards))  # shard 0 always owns the most rows


def gather_frame(local, height, rank, world, dst=0, group=None):
    """local: [max_shard_rows, W, C] tensor holding this rank's packed bands (tail rows unused).
    Returns the assembled [H, W, C] frame on rank `dst`, None elsewhere."""
    if world == 1:
        parts = [local]
    else:
        parts = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
        dist.gather(local, parts, dst=dst, group=group)
        if rank != dst:
            return None
    frame = torch.empty((height,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        rows = band_rows(height, r, world)
        if rows:
            idx = torch.as_tensor(rows, device=local.device)
            frame.index_copy_(0, idx, parts[r][: len(rows)])
    return frame


class ShardRenderer:
    """One rank's share of a frame, device-resident: buffers are allocated once, render()
    only enqueues kernels on the current torch stream (no allocation, no sync)."""

    def __init__(self, scene, samples, rank=0, world=1, seed=1, chunk_spp=0, device=None):
        self.lib = load()
        self.scene, self.samples, self.rank, self.world = scene, int(samples), rank, world
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.opts = scene.opts(seed=seed, chunk_spp=chunk_spp)
        self.rows = int(self.lib.grt_shard_rows(scene.H, rank, world))
        self.max_rows = max_shard_rows(scene.H, world)
        self.rgb_sum = torch.zeros((self.max_rows, scene.W, 3), dtype=torch.float32, device=self.device)
        nbytes = int(self.lib.grt_accum_workspace_bytes(scene.handle, C.byref(self.opts), rank, world, self.samples))
        self.workspace = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.device)
        self.workspace_bytes = nbytes
        self.ray_count = torch.zeros(1, dtype=torch.int64, device=self.device)
        # stage the scene on this device now so that render() never allocates
        self.render()
        torch.cuda.synchronize(self.device)
        self.ray_count.zero_()

    def render(self):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.grt_render_accum_dev(self.scene.handle, C.byref(self.opts), self.rank, self.world,
                                            self.samples, self.rgb_sum.data_ptr(), self.workspace.data_ptr(),
                                            self.workspace_bytes, self.ray_count.data_ptr(), stream))

    def gather(self, dst=0):
        return gather_frame(self.rgb_sum, self.scene.H, self.rank, self.world, dst)

    def tonemap(self, frame_sum):
        """[H, W, 3] float32 sums -> [H, W, 4] uint8 B,G,R,x on the device (scene.go:98-99,67-74)."""
        h, w = frame_sum.shape[:2]
        out = torch.zeros((h, w, 4), dtype=torch.uint8, device=frame_sum.device)
        stream = torch.cuda.current_stream(frame_sum.device).cuda_stream
        check(self.lib.grt_tonemap_dev(frame_sum.data_ptr(), w, h, self.samples, out.data_ptr(), 4 * w, stream))
        return out


def sample_range(samples, rank, world):
    """Samples [first, first + count) of every pixel rendered by `rank`."""
    first = samples * rank // world
    return first, samples * (rank + 1) // world - first


class SampleShardRenderer:
    """Sample-range sharding: every rank renders the whole frame for its share of the sample
    indices (the scene is replicated: the mode for scenes that stress memory, config C5) and the
    FP32 sums are added with one RCCL reduce to rank 0."""

    def __init__(self, scene, samples, rank=0, world=1, seed=1, chunk_spp=0, device=None):
        self.lib = load()
        self.scene, self.samples, self.rank, self.world = scene, int(samples), rank, world
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.opts = scene.opts(seed=seed, chunk_spp=chunk_spp)
        self.first, self.count = sample_range(self.samples, rank, world)
        self.rgb_sum = torch.zeros((scene.H, scene.W, 3), dtype=torch.float32, device=self.device)
        nbytes = int(self.lib.grt_accum_workspace_bytes(scene.handle, C.byref(self.opts), 0, 1, max(1, self.count)))
        self.workspace = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.device)
        self.workspace_bytes = nbytes
        self.ray_count = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.render()  # stages the scene on this device so that later calls never allocate
        torch.cuda.synchronize(self.device)
        self.ray_count.zero_()

    def render(self):
        if self.count == 0:
            self.rgb_sum.zero_()
            return
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.grt_render_accum_samples_dev(self.scene.handle, C.byref(self.opts), self.first, self.count,
                                                    self.rgb_sum.data_ptr(), self.workspace.data_ptr(),
                                                    self.workspace_bytes, self.ray_count.data_ptr(), stream))

    def reduce(self, dst=0):
        """Sum of the ranks' frames on rank dst ([H, W, 3] float32), this rank's own buffer elsewhere."""
        if self.world > 1:
            dist.reduce(self.rgb_sum, dst=dst, op=dist.ReduceOp.SUM)
        return self.rgb_sum
