"""Scene.intersect through the bounding-volume hierarchy (csrc/grt_bvh.h, config C5) against the
linear scan the reference performs (scene.go:169-181): same object ids, same images."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def both(pkg, w, h, objs):
    cam = pkg.scenes.cornell_camera(w / h)
    lin = pkg.tracer.Scene(w, h, cam, objs, accel=pkg._capi.GRT_ACCEL_LINEAR)
    bvh = pkg.tracer.Scene(w, h, cam, objs, accel=pkg._capi.GRT_ACCEL_BVH)
    return lin, bvh


@pytest.mark.parametrize("n_s,n_b,seed", [(0, 0, 1), (30, 30, 2), (300, 0, 3), (0, 300, 4), (600, 600, 5)])
def test_primary_ids_equal_the_linear_scan(pkg, ob, n_s, n_b, seed):
    objs = pkg.scenes.random_objects(n_s, n_b, seed, diffuse=False)
    lin, bvh = both(pkg, 512, 512, objs)
    for jitter in (0, 1):
        li, lt, ln, _ = lin.PrimaryHits(jitter, seed, mode=1)      # FP32 only: the arithmetic the hierarchy shares
        bi, bt, bn, _ = bvh.PrimaryHits(jitter, seed, mode=1)      # the renderer's walk
        assert (li == bi).all(), f"{(li != bi).sum()} ids differ from the FP32 linear scan"
        assert (lt == bt).all() and (ln == bn).all()
        exact_vs_oracle(ob, bvh, jitter, seed)


def exact_vs_oracle(ob, scene, jitter, seed, also_fp64_walk=True):
    """The north-star bar on the hierarchy path (grt_parity.h): hit/miss and object id bit-exact against the
    float64 reference loop (scene.go:169-181), t and normals within 1e-5.  mode 2 = every lane through the
    float64 walk: the same answer."""
    rid, rt, rn = ob.OracleScene(scene.flat).PrimaryHits(jitter, seed)
    hit = rid >= 0
    for mode in (0, 2) if also_fp64_walk else (0,):
        bi, bt, bn, fb = scene.PrimaryHits(jitter, seed, mode=mode)
        assert (bi == rid).all(), f"mode {mode}: {(bi != rid).sum()} ids differ from the float64 oracle"
        assert (np.abs(bt[hit] - rt[hit]) / rt[hit]).max() < 1e-5
        assert np.abs(bn - rn).max() < 1e-5
        assert (bt[~hit] == 0).all()
    return fb


def test_paths_equal_the_linear_scan(pkg):
    """Same RNG, same per-object arithmetic, order-free nearest hit: every sample's radiance and
    path length equal the linear kernel's."""
    objs = pkg.scenes.random_objects(150, 150, 9, diffuse=False)
    lin, bvh = both(pkg, 96, 96, objs)
    xy = np.stack(np.meshgrid(np.arange(0, 96, 3), np.arange(0, 96, 3)), -1).reshape(-1, 2)
    lr, ln = lin.TraceSamples(xy, 16, seed=4)
    br, bn = bvh.TraceSamples(xy, 16, seed=4)
    same = (ln == bn) & (lr == br).all(-1)
    assert same.mean() > 0.999, same.mean()
    a = lin.RenderAccum(32, seed=2)
    b = bvh.RenderAccum(32, seed=2)
    assert np.abs(a - b).max() <= 1e-3 * a.max() or (a == b).mean() > 0.999


def test_deferred_shading_equals_lockstep_levels(pkg, monkeypatch):
    """k_pathtrace_bvh lets lanes whose walk has ended shade while the others keep walking; the
    frame must not depend on when a lane is shaded.  Against itself with every level in lockstep
    (threshold 64) and against k_pathtrace<-1> (variant 5: the small scenes' state machine)."""
    scene = pkg.scenes.spheres(20000, 256, 144)
    a = scene.RenderAccum(24, seed=3)
    monkeypatch.setenv("GRT_BVH_SHADE_MIN", "64")
    b = scene.RenderAccum(24, seed=3)
    monkeypatch.setenv("GRT_BVH_SHADE_MIN", "7")
    c = scene.RenderAccum(24, seed=3)
    monkeypatch.delenv("GRT_BVH_SHADE_MIN")
    d = scene.RenderAccum(24, seed=3, variant=5)
    assert a.max() > 0
    assert (a == b).all() and (a == c).all()
    # variant 5 adds a pixel's samples in FP32, in sample order; the production kernel adds them in fixed point (exact, any order)
    assert np.allclose(a, d, rtol=2e-6, atol=1e-9) and (a == d).mean() > 0.999, (np.abs(a - d).max(), (a == d).mean())
    e = scene.RenderAccum(24, seed=3, variant=8)     # lane = pixel loops (round 1's kernel): the same paths, FP32 sums
    assert np.allclose(a, e, rtol=2e-6, atol=1e-9) and (a == e).mean() > 0.999


def test_counting_build_counts_the_same_rays(pkg):
    """opts.flags = 6: the profiling build of the hierarchy kernel adds inner-node and
    object-test counters; frame and ray count equal the production kernel's."""
    import importlib
    import torch
    shard = importlib.import_module("go-raytracer_amd.shard")
    scene = pkg.scenes.spheres(20000, 192, 108)
    ren = shard.SampleShardRenderer(scene, 8, 0, 1, seed=2)
    ren.render()
    torch.cuda.synchronize()
    frame, rays = ren.rgb_sum.clone(), int(ren.ray_count[0])
    ren.ray_count.zero_()
    ren.opts.flags = 6
    ren.render()
    torch.cuda.synchronize()
    c = ren.ray_count.tolist()
    assert c[0] == rays and (ren.rgb_sum == frame).all()
    assert c[1] > c[0] and 0 < c[2] < 20000 * c[0]                           # > 1 inner node per ray; far fewer tests than a scan


def test_tiny_and_degenerate_scenes(pkg, ob):
    m = pkg.tracer.LambertMaterial((0.5, 0.5, 0.5))
    for objs in ([pkg.tracer.NewSphere((278, 278, 300), 100, m)],
                 [pkg.tracer.NewAABB((100, 100, 100), (400, 400, 400), m)],
                 pkg.scenes.cornell_objects()[:3],
                 [pkg.tracer.NewSphere((278, 278, 300), 100, m)] * 9):      # coincident objects: lowest index wins
        lin, bvh = both(pkg, 64, 64, objs)
        li, lt, ln, _ = lin.PrimaryHits(0, 1, mode=1)
        bi, bt, bn, _ = bvh.PrimaryHits(0, 1, mode=1)
        assert (li == bi).all() and (lt == bt).all()
        assert bi.max() <= 0 or len(objs) != 9                               # nine copies: id 0 everywhere
        exact_vs_oracle(ob, bvh, 0, 1)                                       # tied distances: float64 decides by index


def test_small_scene_seen_from_far_away(pkg, ob):
    """The rounding error of FP32 slab arithmetic grows with the distance of the eye, not with the size of
    the scene: grid cells and box padding must cover it, or the hierarchy culls what the linear scan hits."""
    T, C = pkg.tracer, pkg._capi
    import random
    rnd = random.Random(3)
    m = T.LambertMaterial((0.5, 0.5, 0.5))
    objs = [T.NewSphere((rnd.uniform(-5, 5), rnd.uniform(-5, 5), rnd.uniform(-5, 5)), rnd.uniform(0.3, 1.2), m) for _ in range(40)]
    objs += [T.NewAABB((x, -6, -1), (x + 0.7, -5.5, 1), m) for x in range(-5, 5)]
    for dist, fov in ((2.0e4, 0.05), (3.0e5, 0.0035)):
        cam = T.NewCamera((3.0, 2.0, -dist), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), fov, 1.0)
        lin = T.Scene(128, 128, cam, objs, accel=C.GRT_ACCEL_LINEAR)
        bvh = T.Scene(128, 128, cam, objs, accel=C.GRT_ACCEL_BVH)
        li, lt, _, _ = lin.PrimaryHits(1, 2, mode=1)
        bi, bt, _, _ = bvh.PrimaryHits(1, 2, mode=1)
        assert (li >= 0).mean() > 0.1
        assert (li == bi).all() and (lt == bt).all()
        exact_vs_oracle(ob, bvh, 1, 2)


def test_origin_inside_and_entry_within_bias(pkg, ob):
    """aabb.go:44 returns the entry distance: a box the ray starts inside is no hit, a box
    entered within the bias wins over everything behind it.  Camera inside / just in front of boxes."""
    T = pkg.tracer
    m = T.LambertMaterial((0.5, 0.5, 0.5))
    eye = (278.0, 273.0, -800.0)
    objs = [T.NewAABB((0, 0, 100), (555, 555, 200), m),
            T.NewAABB((200, 200, -900), (350, 350, -700), m),                # contains the eye
            T.NewAABB((0, 0, -799.9995), (555, 555, -700), m),               # entered at t ~ 0.0005 < bias
            T.NewSphere((278, 278, 400), 150, m)] + pkg.scenes.random_objects(40, 40, 3, diffuse=False)[6:]
    lin, bvh = both(pkg, 128, 128, objs)
    li, lt, _, _ = lin.PrimaryHits(1, 2, mode=1)
    bi, bt, _, _ = bvh.PrimaryHits(1, 2, mode=1)
    assert (li == bi).all() and (lt == bt).all()
    assert (bi == 2).mean() > 0.5                                             # the quirk box wins where it is entered
    exact_vs_oracle(ob, bvh, 1, 2)


def test_c5_spheres_scene_at_test_size(pkg, ob):
    """C5 generator at 6000 spheres (above the linear-scan limit, so AUTO picks the hierarchy):
    ids against the float64 oracle's linear scan, image mean against the oracle's."""
    scene = pkg.scenes.spheres(6000, 160, 90)
    ref = ob.OracleScene(scene.flat)
    for jitter in (0, 1):
        fb = exact_vs_oracle(ob, scene, jitter, 1)
        assert 0 < fb <= 160 * 90
    ids, _, _, _ = scene.PrimaryHits(0, 1, mode=1)                           # FP32 alone: silhouette pixels differ
    rid, _, _ = ref.PrimaryHits(0, 1)
    assert (ids != rid).mean() < 5e-4
    a = scene.RenderAccum(16, seed=1).astype(np.float64) / 16
    b = ref.RenderAccum(16, seed=1) / 16
    rel = np.abs(a.mean((0, 1)) - b.mean((0, 1))) / b.mean((0, 1))
    assert (rel < 0.03).all(), rel


def test_auto_takes_the_hierarchy_above_208_objects(pkg, ob):
    """GRT_ACCEL_AUTO: linear scan up to 208 objects, hierarchy above with or without plain-diffuse objects.
    The parity kernel keeps the linear scan + float64 re-test for such scenes (ids exact against the oracle);
    the renderer's frame equals the forced-hierarchy frame bit for bit."""
    C = pkg._capi
    cam = pkg.scenes.cornell_camera(128 / 96)
    objs = pkg.scenes.random_objects(120, 120, 21, diffuse=False)
    auto = pkg.tracer.Scene(128, 96, cam, objs)
    forced = pkg.tracer.Scene(128, 96, cam, objs, accel=C.GRT_ACCEL_BVH)
    lin = pkg.tracer.Scene(128, 96, cam, objs, accel=C.GRT_ACCEL_LINEAR)
    a, f, l = auto.RenderAccum(8, seed=5), forced.RenderAccum(8, seed=5), lin.RenderAccum(8, seed=5)
    assert (a == f).all()
    assert np.allclose(a, l, rtol=2e-6, atol=1e-9) and (a == l).mean() > 0.99     # FP32 sums in sample order against exact fixed-point sums
    ids, t, _, _ = auto.PrimaryHits(0, 1)
    rid, rt, _ = ob.OracleScene(auto.flat).PrimaryHits(0, 1)
    assert (ids == rid).all()
    small = pkg.tracer.Scene(128, 96, cam, objs[:90])                       # <= 208 objects: linear (LDS copies of the objects)
    assert (small.RenderAccum(4, seed=5) == pkg.tracer.Scene(128, 96, cam, objs[:90], accel=C.GRT_ACCEL_LINEAR).RenderAccum(4, seed=5)).all()
    withd = pkg.scenes.random_objects(120, 120, 21, diffuse=True)           # plain diffuse objects: hierarchy all the same
    pkg.tracer.Scene(64, 48, cam, withd).RenderAccum(2, seed=1)
    with pytest.raises(C.GrtError, match="linear scan takes at most"):
        pkg.scenes.spheres(5000, 64, 36, accel=C.GRT_ACCEL_LINEAR)


def test_plain_diffuse_objects_shadow_rays_through_the_hierarchy(pkg, ob):
    """The shadow loop of trace() (scene.go:336-344) is order-dependent as written; direct_light_bvh
    evaluates its order-free form with any-hit queries.  Every sample against the linear loop, the image
    against the float64 oracle.  Lights first, in the middle and last in Scene.Objects (the cap of a
    non-emissive object depends on the emissive objects BELOW it in index order), and two lights."""
    C, T = pkg._capi, pkg.tracer
    cam = pkg.scenes.cornell_camera(96 / 64)
    base = pkg.scenes.random_objects(70, 70, 13, diffuse=True)            # light box = object 0
    light, rest = base[0], base[1:]
    second = T.NewSphere((300, 300, 250), 25, T.LightMaterial(T.NewColor(0.9, 0.8, 0.6), 6))
    for objs in (base, rest[:60] + [light] + rest[60:], rest + [light], [light] + rest[:40] + [second] + rest[40:]):
        lin = T.Scene(96, 64, cam, objs, accel=C.GRT_ACCEL_LINEAR)
        bvh = T.Scene(96, 64, cam, objs, accel=C.GRT_ACCEL_BVH)
        xy = np.stack(np.meshgrid(np.arange(0, 96, 2), np.arange(0, 64, 2)), -1).reshape(-1, 2)
        lr, ln = lin.TraceSamples(xy, 8, seed=4)
        br, bn = bvh.TraceSamples(xy, 8, seed=4)
        same = (ln == bn) & (lr == br).all(-1)
        assert same.mean() > 0.999, same.mean()
        a, b = lin.RenderAccum(16, seed=2), bvh.RenderAccum(16, seed=2)     # k_pathtrace<0> vs k_pathtrace_bvh<.., DIFFUSE>
        # the hierarchy kernel adds a pixel's samples in fixed point (exact), the others in FP32: equal to FP32 rounding
        assert np.isclose(a, b, rtol=2e-6, atol=1e-9).mean() > 0.999 and a.max() > 0
        assert np.isclose(b, bvh.RenderAccum(16, seed=2, variant=5), rtol=2e-6, atol=1e-9).mean() > 0.999  # lockstep kernel, same branch
    o = ob.OracleScene(bvh.flat).RenderAccum(16, seed=2) / 16
    g = b.astype(np.float64) / 16
    rel = np.abs(g.mean((0, 1)) - o.mean((0, 1))) / o.mean((0, 1))
    assert (rel < 0.03).all(), rel                                        # FP32 vs float64 paths part ways at silhouettes (as at C5)


def test_photon_pass_on_a_hierarchy_scene(pkg):
    """tracePhotons (scene.go:354-424) through the hierarchy: the caustic filter (:359-363) is an any-hit
    query over the transparent objects, intersect() the order-free walk.  Same photons as the linear scan."""
    C, T = pkg._capi, pkg.tracer
    objs = pkg.scenes.cornell_objects("cornell8d") + pkg.scenes.random_objects(60, 60, 17, diffuse=True)[6:]
    cam = pkg.scenes.cornell_camera(1.0)
    lin = T.Scene(64, 64, cam, objs, accel=C.GRT_ACCEL_LINEAR)
    bvh = T.Scene(64, 64, cam, objs, accel=C.GRT_ACCEL_BVH)
    for caustics, n in ((False, 8000), (True, 1500)):
        a = T.PhotonMap(n).build(lin, caustics=caustics, seed=5)
        b = T.PhotonMap(n).build(bvh, caustics=caustics, seed=5)
        assert a.count() == b.count() == n
        assert abs(a.emitted() - b.emitted()) <= max(2, 1e-3 * a.emitted())
        (ap, aw, at), (bp, bw, bt) = a.read(), b.read()
        same = (ap == bp).all(1) & (aw == bw).all(1) & (at == bt).all(1)
        assert same.mean() > 0.99, same.mean()
    # and the photon-mapped trace() on the hierarchy scene (scene.go:308-320)
    lin.globalPmap, lin.causticPmap = T.PhotonMap(8000).build(lin, seed=5), T.PhotonMap(1500).build(lin, caustics=True, seed=5)
    bvh.globalPmap, bvh.causticPmap = lin.globalPmap, lin.causticPmap
    fa, fb = lin.RenderAccumPhotonMapped(8, seed=1), bvh.RenderAccumPhotonMapped(8, seed=1)
    assert np.isclose(fa, fb, rtol=1e-4, atol=1e-6).mean() > 0.999 and fa.max() > 0


def test_wave_queue_on_a_frame_with_partial_tiles(pkg):
    """k_pathtrace_bvh hands the (pixel, sample) pairs of a tile x chunk to whichever lane is free and adds a path's
    radiance, in fixed point, to the sum its pixel's lane holds.  On a frame whose edge tiles are partly outside
    (101 x 67) the pairs of outside pixels are born finished: same intersect() count as the lane = pixel loops
    (variant 8) and as the linear scan, same frame to FP32 rounding, and bit-identical frames for every chunk length
    whose chunks add up in the same order (one chunk) and for every scheduling threshold."""
    import importlib
    import torch
    shard = importlib.import_module("go-raytracer_amd.shard")
    C = pkg._capi
    W, H, S = 101, 67, 12
    cam = pkg.scenes.cornell_camera(W / H)
    objs = pkg.scenes.random_objects(150, 150, 5, diffuse=False)
    bvh = pkg.tracer.Scene(W, H, cam, objs, accel=C.GRT_ACCEL_BVH)
    lin = pkg.tracer.Scene(W, H, cam, objs, accel=C.GRT_ACCEL_LINEAR)
    lib = C.load()
    q = bvh.RenderAccum(S, seed=6, chunk_spp=S)
    rays_q = int(lib.grt_last_ray_count())
    p = bvh.RenderAccum(S, seed=6, chunk_spp=S, variant=8)
    rays_p = int(lib.grt_last_ray_count())
    l = lin.RenderAccum(S, seed=6, chunk_spp=S)
    rays_l = int(lib.grt_last_ray_count())
    assert q.max() > 0 and rays_q == rays_p
    assert abs(rays_q - rays_l) <= 2e-3 * rays_l                         # FP32 walks and the linear loop part ways at a few silhouettes
    assert np.allclose(q, p, rtol=2e-6, atol=1e-9)
    assert np.isclose(q, l, rtol=2e-6, atol=1e-9).mean() > 0.998
    assert (bvh.RenderAccum(S, seed=6, chunk_spp=S) == q).all()          # reproducible
    parts = bvh.RenderAccum(S, seed=6, chunk_spp=5)                      # chunks 5 + 5 + 2: slabs added in FP32
    assert np.allclose(parts, q, rtol=2e-6, atol=1e-9)
    ren = shard.SampleShardRenderer(bvh, S, 0, 1, seed=6, chunk_spp=S)   # device-resident path, same numbers
    ren.render()
    torch.cuda.synchronize()
    assert (ren.rgb_sum.cpu().numpy() == q).all() and int(ren.ray_count[0]) == rays_q


def test_fixed_point_sums_follow_the_scene_brightness(pkg):
    """k_pathtrace_bvh adds a pixel's radiance in 64-bit fixed point whose scale comes from a bound on what one path can
    return (grt_scene_create: brightest source x largest throughput).  Very bright and very dim lights, colours above 1
    (throughput grows along a path) and a NormalMaterial object (negative radiance, scene.go:275): the frame equals the
    linear scan's FP32 sums to FP32 rounding in every case — nothing overflows, nothing is rounded away."""
    T, C = pkg.tracer, pkg._capi
    cam = pkg.scenes.cornell_camera(96 / 64)
    rest = pkg.scenes.random_objects(40, 40, 3, diffuse=False)[1:]
    for emit, colour in ((1e7, (0.73, 0.73, 0.73)), (1e-6, (0.73, 0.73, 0.73)), (15.0, (2.5, 1.5, 3.0))):
        light = T.NewAABB((213, 548, 227), (343, 548.1, 332), T.LightMaterial(T.NewColor(1, 0.9, 0.8), emit))
        bright = T.NewSphere((278, 200, 278), 70, T.LambertMaterial(colour))
        normal = T.NewSphere((150, 100, 150), 60, T.NormalMaterial())
        objs = [light, bright, normal] + rest
        bvh = T.Scene(96, 64, cam, objs, accel=C.GRT_ACCEL_BVH)
        lin = T.Scene(96, 64, cam, objs, accel=C.GRT_ACCEL_LINEAR)
        a, b = bvh.RenderAccum(16, seed=9, chunk_spp=16), lin.RenderAccum(16, seed=9, chunk_spp=16)
        assert np.isfinite(a).all() and np.abs(a).max() > 0 and a.min() < 0          # the NormalMaterial sphere shows negative sums
        close = np.isclose(a, b, rtol=3e-6, atol=1e-6 * np.abs(b).max())
        assert close.mean() > 0.998, (emit, colour, close.mean())
    # a scene far below 1 with no NormalMaterial in it: the unit follows the scene (it stopped at 2^-40 and
    # truncated towards zero, 9e-13 absolute, which a 1e-9 light shows as a relative bias)
    for emit in (1e-9, 3e-20):
        light = T.NewAABB((213, 548, 227), (343, 548.1, 332), T.LightMaterial(T.NewColor(1, 0.9, 0.8), emit))
        objs = [light, T.NewSphere((278, 200, 278), 70, T.LambertMaterial((0.73, 0.73, 0.73)))] + rest
        bvh = T.Scene(96, 64, cam, objs, accel=C.GRT_ACCEL_BVH)
        lin = T.Scene(96, 64, cam, objs, accel=C.GRT_ACCEL_LINEAR)
        a, b = bvh.RenderAccum(16, seed=9, chunk_spp=16), lin.RenderAccum(16, seed=9, chunk_spp=16)
        assert b.max() > 0 and np.isfinite(a).all()
        lit = b > 1e-3 * b.max()
        rel = np.abs(a[lit] - b[lit]) / b[lit]
        assert (rel < 1e-5).mean() > 0.998, (emit, rel.max())
        assert abs(a.sum() - b.sum()) / b.sum() < 1e-5            # no systematic bias
