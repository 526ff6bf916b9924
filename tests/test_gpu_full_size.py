"""BASELINE.json's configurations at their FULL sizes, checked through properties that do not need the
oracle to finish (it cannot at these sizes): determinism, independence of how the work is cut (bands,
sample ranges, chunks, scheduling thresholds, query order), monotonicity, and totals that must agree.
Each case runs in seconds on the GPU."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shard():
    return importlib.import_module("go-raytracer_amd.shard")


def test_c3_full_frame_is_independent_of_the_sharding(pkg, shard):
    """C3: cornell8, 1920x1080, 1000 spp.  The frame assembled from 3 and from 8 shards of interleaved
    8-row bands equals the single-shard frame bit for bit (RNG keyed by absolute pixel and sample, ordered
    chunk sums) when the chunk size is the same — the automatic choice follows the shard's tile count, which
    moves the chunk boundaries and with them the last bit of the FP32 sums: then the frames agree to rounding.
    Twice in a row the same frame, and the ray counts add up."""
    import torch

    W, H, S = 1920, 1080, 1000
    scene = pkg.scenes.cornell(W, H)
    one = shard.ShardRenderer(scene, S, 0, 1, seed=1, chunk_spp=125)
    one.render()
    torch.cuda.synchronize()
    whole, rays = one.gather(0).clone(), int(one.ray_count.item())
    one.ray_count.zero_()
    one.render()
    torch.cuda.synchronize()
    assert (one.gather(0) == whole).all()                                   # reproducible
    assert 1.5 < rays / (W * H * S) < 2.5 and torch.isfinite(whole).all() and (whole >= 0).all()
    for G in (3, 8):
        frame = torch.empty_like(whole)
        total = 0
        for r in range(G):
            ren = shard.ShardRenderer(scene, S, r, G, seed=1, chunk_spp=125)
            ren.render()
            torch.cuda.synchronize()
            rows = torch.as_tensor(shard.band_rows(H, r, G), device=frame.device)
            frame.index_copy_(0, rows, ren.rgb_sum[: len(rows)])
            total += int(ren.ray_count.item())
        assert (frame == whole).all(), G
        assert total == rays
    auto = shard.ShardRenderer(scene, S, 1, 8, seed=1)                       # automatic chunk size: 16 spp at 8 shards
    auto.render()
    torch.cuda.synchronize()
    rows = torch.as_tensor(shard.band_rows(H, 1, 8), device=whole.device)
    assert torch.allclose(auto.rgb_sum[: len(rows)], whole[rows], rtol=2e-5, atol=1e-4)
    # the tone-mapped bytes of the full frame: B,G,R in [0,255], the 4th byte untouched
    px = one.tonemap(whole)
    assert px.shape == (H, W, 4) and int(px[..., 3].max()) == 0 and int(px[..., :3].max()) == 255


def test_c5_full_scene_is_independent_of_scheduling_and_sample_split(pkg, shard, monkeypatch):
    """C5: 10^6 spheres, 1920x1080 (8 spp here).  The frame does not depend on the scheduling thresholds of
    k_pathtrace_bvh (bit-identical), the counting build traces the same rays, and the frames of two sample
    ranges add up to the whole (FP32 addition order aside)."""
    import torch

    W, H, S = 1920, 1080, 8
    scene = pkg.scenes.spheres(1000000, W, H)
    ren = shard.SampleShardRenderer(scene, S, 0, 1, seed=1)
    ren.render()
    torch.cuda.synchronize()
    whole, rays = ren.rgb_sum.clone(), int(ren.ray_count[0])
    assert whole.max() > 0 and torch.isfinite(whole).all()
    for env in ({"GRT_BVH_SHADE_MIN": "64"}, {"GRT_BVH_INNER_RUN": "1", "GRT_BVH_INNER_WEIGHT": "4"}, {"GRT_BVH_OCC": "6"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ren.ray_count.zero_()
        ren.render()
        torch.cuda.synchronize()
        for k in env:
            monkeypatch.delenv(k)
        assert (ren.rgb_sum == whole).all(), env
        assert int(ren.ray_count[0]) == rays
    ren.ray_count.zero_()
    ren.opts.flags = 6                                                       # counting build
    ren.render()
    torch.cuda.synchronize()
    ren.opts.flags = 0
    c = ren.ray_count.tolist()
    assert c[0] == rays and (ren.rgb_sum == whole).all()
    assert 20 * rays < c[1] < 200 * rays and rays < c[2] < 10 * rays         # ~59 inner nodes, ~2 objects per ray, not 10^6
    total = torch.zeros_like(whole)
    for first, count in ((0, 3), (3, 5)):
        part = shard.SampleShardRenderer(scene, S, 0, 1, seed=1)
        part.first, part.count = first, count
        part.render()
        torch.cuda.synchronize()
        total += part.rgb_sum
    assert torch.allclose(total, whole, rtol=1e-5, atol=1e-6)


def test_c5_every_kernel_computes_the_same_path(pkg, shard):
    """One value per expression: the per-lane walk of grt_trace_samples, the wave-queue kernel and the
    lane = pixel kernel are three instantiations of the same inline functions and must return the same radiance for
    every (pixel, sample).  The C5 scene amplifies an ulp within two bounces; the pixels below are the 16 of 2 073 600
    on which the kernels disagreed while the backend was free to fuse a*b - c*d differently per instantiation
    (hipcc's default -ffp-contract=fast), plus a band of ordinary ones."""
    import torch

    W, H, S = 1920, 1080, 16
    pixels = [(1123, 152), (706, 164), (1309, 165), (763, 171), (1017, 178), (1158, 186), (1124, 229), (1086, 245), (814, 270),
              (747, 283), (1162, 290), (886, 297), (815, 394), (637, 416), (1106, 418), (1246, 508)]
    pixels += [(x, 300) for x in range(0, W, 16)]
    scene = pkg.scenes.spheres(1000000, W, H)
    per_sample, n_rays = scene.TraceSamples(np.array(pixels, np.int32), S, seed=1)
    want = per_sample.astype(np.float64).sum(axis=1)
    xs, ys = [p[0] for p in pixels], [p[1] for p in pixels]
    for flags in (0, 8):
        ren = shard.SampleShardRenderer(scene, S, 0, 1, seed=1)
        ren.opts.flags = flags
        ren.ray_count.zero_()
        ren.render()
        torch.cuda.synchronize()
        got = ren.rgb_sum.cpu().numpy()[ys, xs].astype(np.float64)
        # same paths: the sums differ only by their own rounding (64-bit fixed point / FP32 in sample order);
        # a path that went another way moves a pixel by a light's worth (>= 0.4)
        assert np.abs(got - want).max() < 1e-4 * max(1.0, want.max()), flags
    # and sample by sample: one frame per sample from the queue kernel
    for s in (1, 7, 12):
        one = shard.SampleShardRenderer(scene, S, s, S, seed=1)
        one.render()
        torch.cuda.synchronize()
        got = one.rgb_sum.cpu().numpy()[ys, xs]
        assert np.allclose(got, per_sample[:, s], rtol=2e-6, atol=1e-7), s


def test_c5_full_scene_at_its_baseline_sample_count(pkg, shard, monkeypatch):
    """C5 at BASELINE's 256 spp (the sample count the fixed-point pixel sums are sized for): the production frame
    (wave queue, 64-bit fixed-point sums) is bit-identical under another scheduling setting, equals the lane = pixel
    kernel (opts.flags = 8: FP32 sums in sample order) to FP32 rounding — a fixed-point overflow or a lost path would
    show as a gross difference —, and the two kernels trace the same number of rays."""
    import torch

    W, H, S = 1920, 1080, 256
    scene = pkg.scenes.spheres(1000000, W, H)
    ren = shard.SampleShardRenderer(scene, S, 0, 1, seed=1)
    ren.render()
    torch.cuda.synchronize()
    whole, rays = ren.rgb_sum.clone(), int(ren.ray_count[0])
    assert torch.isfinite(whole).all() and (whole >= 0).all() and whole.max() > 0
    assert 1.5 < rays / (W * H * S) < 2.5
    monkeypatch.setenv("GRT_BVH_SHADE_MIN", "48")
    monkeypatch.setenv("GRT_BVH_INNER_RUN", "2")
    ren.ray_count.zero_()
    ren.render()
    torch.cuda.synchronize()
    monkeypatch.delenv("GRT_BVH_SHADE_MIN")
    monkeypatch.delenv("GRT_BVH_INNER_RUN")
    assert (ren.rgb_sum == whole).all() and int(ren.ray_count[0]) == rays
    ren.ray_count.zero_()
    ren.opts.flags = 8
    ren.render()
    torch.cuda.synchronize()
    ren.opts.flags = 0
    assert int(ren.ray_count[0]) == rays
    scale = float(whole.max())
    diff = (ren.rgb_sum - whole).abs()
    print(f"C5 256 spp: queue vs lane=pixel kernel, max abs diff {float(diff.max()):.3e} at frame max {scale:.3e}")
    assert torch.allclose(ren.rgb_sum, whole, rtol=5e-5, atol=5e-6 * scale)


def test_c4_full_gather_is_independent_of_the_query_order(pkg):
    """C4: 10^6 photons, one query per primary diffuse hit at 1920x1080 (~9*10^5), k = 100.  The estimate of a
    query does not depend on which queries precede it (the carried bound only prunes), k = 50 and k = 100
    estimate the same irradiance, the k-th distance grows with k, and every photon asked for is stored."""
    import torch

    W, H = 1920, 1080
    scene = pkg.scenes.cornell(W, H, "cornell8d")
    gmap = pkg.tracer.PhotonMap(1000000).build(scene, caustics=False, seed=1)
    assert gmap.count() == 1000000 and gmap.emitted() > gmap.count()
    ids, t, nrm, _ = scene.PrimaryHits()
    hit = (ids >= 1) & (ids <= 5)
    ys, xs = np.nonzero(hit)
    c = scene.flat[0].cam
    cam = np.array(c.origin)
    u, v = (xs + 0.5) / W, (ys + 0.5) / H
    D = np.array(c.lower_left)[None] + np.array(c.horizontal)[None] * u[:, None] + np.array(c.vertical)[None] * v[:, None] - cam[None]
    P = (cam[None] + D * t[hit].astype(np.float64)[:, None]).astype(np.float32)
    N = nrm[hit].astype(np.float32)
    assert 8e5 < len(P) < 1e6
    a = gmap.IrradianceEst(P, N, 0.0, 100)
    assert np.isfinite(a).all() and (a >= 0).all() and (a.sum(1) > 0).mean() > 0.95
    perm = np.random.RandomState(1).permutation(len(P))                      # image order destroyed: every bound is loose
    b = gmap.IrradianceEst(P[perm], N[perm], 0.0, 100)
    close = np.isclose(a[perm], b, rtol=1e-4, atol=1e-9).all(1)              # same photon set; FP32 sums in another order
    assert close.mean() > 0.9999, close.mean()
    half = gmap.IrradianceEst(P[::7], N[::7], 0.0, 50)
    full = a[::7]
    # k = 50 against k = 100 on the same surface: both estimate the same irradiance (within Monte Carlo spread)
    m50, m100 = half.mean(0), full.mean(0)
    assert np.allclose(m50, m100, rtol=0.05), (m50, m100)
    # the squared k-th distance the gather reports (its workspace output)
    lib = pkg._capi.load()
    dP, dN = torch.from_numpy(P[::7].copy()).cuda(), torch.from_numpy(N[::7].copy()).cuda()
    out = torch.empty_like(dP)
    ws = torch.empty(int(lib.grt_photon_gather_workspace_bytes(len(dP))), dtype=torch.uint8, device="cuda")
    rk = {}
    for k in (50, 100):
        pkg._capi.check(lib.grt_photon_gather_dev(gmap.handle, dP.data_ptr(), dN.data_ptr(), len(dP), 0.0, k, out.data_ptr(),
                                                  ws.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        rk[k] = ws.view(torch.float32)[: len(dP)].clone()
    assert (rk[50] <= rk[100]).all() and (rk[50] > 0).all()                  # the k-th distance grows with k


def test_c4_full_sharded_build_equals_the_single_build(pkg, shard):
    """C4 at 10^6 photons: the photon map built from emission slices (grt_photon_emit_dev + all-gather +
    grt_photon_load_dev, here with one rank) holds the photons of grt_photon_build, bit for bit and in the
    same order, and reports the same emission count."""
    scene = pkg.scenes.cornell(64, 64, "cornell8d")
    ref = pkg.tracer.PhotonMap(1000000).build(scene, caustics=False, seed=1)
    one = shard.build_photon_map_sharded(scene, 1000000, False, 0, 1, seed=1)
    assert one.count() == ref.count() == 1000000 and one.emitted() == ref.emitted()
    for a, b in zip(one.read(), ref.read()):
        assert (a == b).all()


def test_c4_ten_million_photons_knn_is_exact(pkg):
    """C4's larger size: 10^7 photons (tree depth 18).  The squared k-th distance the gather reports for
    256 queries equals a brute-force k-th smallest over ALL photons (torch on the GPU)."""
    import torch

    scene = pkg.scenes.cornell(64, 64, "cornell8d")
    gmap = pkg.tracer.PhotonMap(10000000).build(scene, caustics=False, seed=1)
    assert gmap.count() == 10000000
    pos, pw, tp = gmap.read()
    P = torch.from_numpy(pos).cuda()
    rs = np.random.RandomState(4)
    sel = rs.choice(len(pos), 256, replace=False)
    Q = pos[sel] + rs.normal(scale=0.05, size=(256, 3)).astype(np.float32)   # near the surfaces, off the photons
    Nq = np.tile(np.array([0.0, 0.0, 0.0], np.float32), (256, 1))
    lib = pkg._capi.load()
    dQ, dN = torch.from_numpy(Q).cuda(), torch.from_numpy(Nq).cuda()
    out = torch.empty_like(dQ)
    ws = torch.empty(int(lib.grt_photon_gather_workspace_bytes(256)), dtype=torch.uint8, device="cuda")
    pkg._capi.check(lib.grt_photon_gather_dev(gmap.handle, dQ.data_ptr(), dN.data_ptr(), 256, 0.0, 100, out.data_ptr(), ws.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    rk2 = ws.view(torch.float32)[:256].cpu().numpy()
    ref = np.empty(256, np.float32)
    for i in range(0, 256, 32):                                              # 32 queries x 10^7 distances at a time
        d = P[None, :, :] - dQ[i:i + 32, None, :]
        d2 = torch.addcmul(torch.addcmul(d[..., 0] * d[..., 0], d[..., 1], d[..., 1]), d[..., 2], d[..., 2])
        ref[i:i + 32] = torch.topk(d2, 100, dim=1, largest=False).values[:, -1].cpu().numpy()
    assert np.allclose(rk2, ref, rtol=1e-5), np.abs(rk2 / ref - 1).max()


def test_c5_million_spheres_hits_equal_a_brute_force_scan(pkg):
    """C5's hierarchy at its full size (10^6 spheres, 20 levels, 16-bit grid): the object a pixel-centre
    ray hits equals what the reference's loop over ALL objects finds (scene.go:169-181, evaluated in float64
    with torch on the GPU for 1024 pixels inside the cloud)."""
    import torch

    W, H = 1920, 1080
    flat = pkg.scenes.spheres_flat(1000000, W, H)
    scene = pkg.tracer.Scene.from_flat(flat)
    ids, t, _, _ = scene.PrimaryHits(0, 1)
    prims = np.frombuffer(flat[1], dtype=pkg.tracer.PRIM_DTYPE)[: flat[2]]
    c = flat[0].cam
    rs = np.random.RandomState(8)
    ys, xs = rs.randint(300, 780, 1024), rs.randint(560, 1360, 1024)
    f64 = dict(dtype=torch.float64, device="cuda")
    O = torch.tensor(list(c.origin), **f64)
    u, v = torch.tensor((xs + 0.5) / W, **f64), torch.tensor((ys + 0.5) / H, **f64)
    D = (torch.tensor(list(c.lower_left), **f64)[None] + torch.tensor(list(c.horizontal), **f64)[None] * u[:, None]
         + torch.tensor(list(c.vertical), **f64)[None] * v[:, None] - O[None])
    sph = prims["kind"] == pkg.tracer.GRT_KIND_SPHERE
    C = torch.tensor(prims["a"][sph], **f64)
    R = torch.tensor(prims["radius"][sph], **f64)
    sph_index = torch.tensor(np.nonzero(sph)[0], device="cuda")
    bias, best_t, best_i = float(flat[0].bias), torch.full((1024,), float("inf"), **f64), torch.full((1024,), -1, dtype=torch.int64, device="cuda")
    for k, (lo, hi) in enumerate(zip(prims["a"][~sph], prims["b"][~sph])):   # the two boxes come first in Scene.Objects
        lo, hi = torch.tensor(lo, **f64), torch.tensor(hi, **f64)
        n, f = (lo[None] - O[None]) / D, (hi[None] - O[None]) / D
        t0, t1 = torch.minimum(n, f).max(1).values, torch.maximum(n, f).min(1).values
        ok = (torch.minimum(best_t, t1) > torch.clamp(t0, min=bias)) & (t0 > 0)
        best_t = torch.where(ok, t0, best_t)
        best_i = torch.where(ok, torch.full_like(best_i, int(np.nonzero(~sph)[0][k])), best_i)
    a = (D * D).sum(1)
    for i in range(0, 1024, 64):                                             # 64 rays x 10^6 spheres at a time
        oc = O[None, None, :] - C[None, :, :]
        hb = (oc * D[i:i + 64, None, :]).sum(2)
        cc = (oc * oc).sum(2) - (R * R)[None, :]
        disc = hb * hb - a[i:i + 64, None] * cc
        sq = torch.sqrt(torch.clamp(disc, min=0))
        t1, t2 = (-hb - sq) / a[i:i + 64, None], (-hb + sq) / a[i:i + 64, None]
        inf = torch.full_like(t1, float("inf"))
        th = torch.where((disc >= 0) & (t1 > bias), t1, torch.where((disc >= 0) & (t2 > bias), t2, inf))
        m = th.min(1)
        closer = m.values < best_t[i:i + 64]                                 # strictly closer: later objects never win ties
        best_t[i:i + 64] = torch.where(closer, m.values, best_t[i:i + 64])
        best_i[i:i + 64] = torch.where(closer, sph_index[m.indices], best_i[i:i + 64])
    ref_i, ref_t = best_i.cpu().numpy(), best_t.cpu().numpy()
    got_i, got_t = ids[ys, xs], t[ys, xs]
    assert (ref_i >= 2).mean() > 0.5                                         # most of these rays hit a sphere
    assert (got_i == ref_i).all(), (got_i != ref_i).sum()                   # float64 re-test of uncertain lanes (grt_parity.h)
    hit = ref_i >= 0
    assert (np.abs(got_t[hit] - ref_t[hit]) / ref_t[hit]).max() < 1e-5
    fp32_i = scene.PrimaryHits(0, 1, mode=1)[0][ys, xs]                      # the renderer's FP32 walk alone
    assert (fp32_i == ref_i).mean() > 0.998


def test_c4_full_frame_photon_mapped_render_with_dark_photons_is_the_plain_render(pkg):
    """Photon-mapped trace() (scene.go:308-320) at 1920x1080 x 16 spp: with maps whose photons carry no
    power the photon term is zero, so the frame must be the plain path-traced frame (another grouping of
    the FP32 sums: batches of 8 samples instead of chunks); with real maps it can only add light."""
    W, H, S = 1920, 1080, 16
    scene = pkg.scenes.cornell(W, H, "cornell8d")
    plain = scene.RenderAccum(S, seed=1)
    rs = np.random.RandomState(2)
    pos = rs.uniform(0, 555, (5000, 3)).astype(np.float32)
    dark = pkg.tracer.PhotonMap.load(pos, np.zeros((5000, 3), np.float32), rs.randint(0, 256, (5000, 2)).astype(np.uint8))
    scene.globalPmap, scene.causticPmap = dark, dark
    a = scene.RenderAccumPhotonMapped(S, seed=1)
    assert np.allclose(a, plain, rtol=1e-5, atol=1e-5)
    scene.globalPmap = pkg.tracer.PhotonMap(1000000).build(scene, caustics=False, seed=1)
    scene.causticPmap = pkg.tracer.PhotonMap(500000).build(scene, caustics=True, seed=1)
    b = scene.RenderAccumPhotonMapped(S, seed=1)
    gain = float(b.mean() / plain.mean())
    assert (b >= plain - 1e-4).all() and 1.0005 < gain < 1.5, gain           # the maps add light, a little (powers scale with 1000 / max)
