"""Tile sharding of one frame over ranks (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference's unit of work is an image row handed to a goroutine and re-ordered by a
collector (scene.go:107-132).  Here rows are dealt in interleaved bands of GRT_BAND_ROWS rows:
band b belongs to rank b mod G.  Pixels are independent (scene.go:89-97 holds no cross-pixel
state) and the counter RNG is keyed by the absolute pixel index, so the assembled frame is
bit-identical for every G at a given chunk size (the automatic chunk size follows the shard's tile
count: frames of different G then agree to FP32 rounding).  The only exchange is one gather of the finished bands to rank 0
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


def frame_row_source(height, world, max_rows):
    """src[y] = position of global row y in the stacked per-rank buffers ([world * max_rows] rows)."""
    src = [0] * height
    for r in range(world):
        for k, y in enumerate(band_rows(height, r, world)):
            src[y] = r * max_rows + k
    return src


def gather_frame(local, height, rank, world, dst=0, group=None, cache=None):
    """local: [max_shard_rows, W, C] tensor holding this rank's packed bands (tail rows unused).
    Returns the assembled [H, W, C] frame on rank `dst`, None elsewhere.  `cache` (a dict the caller
    keeps) holds the receive buffer and the row permutation between calls, so a repeated gather
    allocates nothing and assembles the frame with one index_select."""
    cache = {} if cache is None else cache
    if rank == dst and "src" not in cache:
        cache["src"] = torch.as_tensor(frame_row_source(height, world, local.shape[0]), device=local.device)
        cache["stack"] = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        cache["frame"] = torch.empty((height,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if world == 1 and not dist.is_initialized():
        cache["stack"][0].copy_(local)
    else:  # also with ONE rank when a process group exists (torchrun --nproc-per-node 1): the RCCL path runs as with N ranks
        dist.gather(local, list(cache["stack"].unbind(0)) if rank == dst else None, dst=dst, group=group)
        if rank != dst:
            return None
    torch.index_select(cache["stack"].view((-1,) + tuple(local.shape[1:])), 0, cache["src"], out=cache["frame"])
    return cache["frame"]


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
        self._gather_cache = {}
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
        """The assembled frame on rank dst (a buffer this renderer owns: the next gather overwrites it)."""
        return gather_frame(self.rgb_sum, self.scene.H, self.rank, self.world, dst, cache=self._gather_cache)

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
        self.ray_count = torch.zeros(16, dtype=torch.int64, device=self.device)  # [0] rays; [1..10]: counters of variant 6
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
        if self.world > 1 or dist.is_initialized():
            dist.reduce(self.rgb_sum, dst=dst, op=dist.ReduceOp.SUM)
        return self.rgb_sum


# ---- photon-build sharding: emission index ranges per rank, all-gather ------------
def gather_slices(pos4, pow4, count, rank, world, group=None):
    """pos4, pow4: [slice, 4] float32 tensors whose first `count` rows are this rank's stored photons
    (emission order).  Returns (pos, pow, global_offset) of ALL ranks' photons concatenated in rank
    order = emission order; global_offset[i] = offset of photon i's emission inside the batch
    (rank * slice + the offset grt_photon_emit_dev left in pow4[:, 3])."""
    n_slice = pos4.shape[0]
    cnt = torch.tensor([int(count)], dtype=torch.int64, device=pos4.device)
    if world > 1 or dist.is_initialized():
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt, group=group)
        poss = [torch.empty_like(pos4) for _ in range(world)]
        pows = [torch.empty_like(pow4) for _ in range(world)]
        dist.all_gather(poss, pos4, group=group)
        dist.all_gather(pows, pow4, group=group)
    else:
        cnts, poss, pows = [cnt], [pos4], [pow4]
    out_pos, out_pow, out_off = [], [], []
    for r in range(world):
        c = int(cnts[r].item())
        out_pos.append(poss[r][:c])
        out_pow.append(pows[r][:c])
        out_off.append(pows[r][:c, 3].contiguous().view(torch.int32).to(torch.int64) + r * n_slice)
    return torch.cat(out_pos), torch.cat(out_pow), torch.cat(out_off)


def build_photon_map_sharded(scene, max_photons, caustics=False, rank=0, world=1, seed=1, device=None,
                             batch_max=1 << 22, gather=gather_slices):
    """mapPhotons for one map (scene.go:140-165) with the emission loop's iterations dealt to the
    ranks: every batch of emission indices is cut into `world` consecutive slices, rank g traces
    slice g (grt_photon_emit_dev), the stored photons are all-gathered in rank order (= emission
    order), and every rank keeps the first `max` stores and builds the identical map
    (grt_photon_load_dev).  The result equals grt_photon_build's on one GPU."""
    from .tracer import PhotonMap

    lib = load()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    opts = scene.opts(seed=seed)
    stream = torch.cuda.current_stream(device).cuda_stream
    slice_max = max(256, batch_max // world)
    ws = torch.empty(int(lib.grt_photon_emit_workspace_bytes(slice_max)), dtype=torch.uint8, device=device)
    pos4 = torch.zeros((slice_max, 4), dtype=torch.float32, device=device)
    pow4 = torch.zeros((slice_max, 4), dtype=torch.float32, device=device)
    d_cnt = torch.zeros(1, dtype=torch.int32, device=device)
    kept_pos, kept_pow = [], []
    stored, emitted = 0, 0
    for light in range(int(lib.grt_scene_light_count(scene.handle))):
        target = min(max_photons, max_photons * int(lib.grt_scene_light_share(scene.handle, light)))
        yield_, empty = 0.0, 0
        while stored < target:
            remaining = target - stored
            want = int(remaining / yield_ * 1.05 + 4096) if yield_ > 0 else max(65536, 2 * remaining)
            n_slice = min(slice_max, max(256, -(-want // world)))
            check(lib.grt_photon_emit_dev(scene.handle, C.byref(opts), int(caustics), light, emitted + rank * n_slice, n_slice,
                                          pos4.data_ptr(), pow4.data_ptr(), d_cnt.data_ptr(), ws.data_ptr(), ws.numel(), stream))
            p, w, off = gather(pos4[:n_slice], pow4[:n_slice], int(d_cnt.item()), rank, world)
            got = p.shape[0]
            if stored + got >= target:                       # Store's cap (photon_mapping.go:166): the first `target` stores
                keep = target - stored
                p, w = p[:keep], w[:keep]
                emitted += int(off[keep - 1].item()) + 1     # the loop ends with the store that fills the map
                stored = target
            else:
                emitted += n_slice * world
                stored += got
            kept_pos.append(p.clone())
            kept_pow.append(w.clone())
            yield_ = max(1e-6, got / (n_slice * world))
            empty = 0 if got else empty + 1
            if empty >= 64:
                raise RuntimeError(f"no photon stored in {emitted} emissions (no DiffuseMaterial surface reachable?)")
    all_pos = torch.cat(kept_pos).contiguous() if kept_pos else pos4[:0]
    all_pow = torch.cat(kept_pow).contiguous() if kept_pow else pow4[:0]
    h = C.c_void_p()
    torch.cuda.synchronize(device)
    check(lib.grt_photon_load_dev(C.byref(opts), all_pos.data_ptr(), all_pow.data_ptr(), stored, 1000.0 / max_photons, emitted,
                                  C.byref(h)))
    return PhotonMap(max_photons, h)
