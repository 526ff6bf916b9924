"""Photon pass, CPU side: the float64 restatement against hand-derived values.
PARITY UNPINNED for this path (the reference never runs it and its kd-tree is gonum's, absent):
these tests pin the restatement to the formulas in the source text, nothing more."""
import ctypes as C
import math

import numpy as np


def quant(ob, d):
    out = (C.c_uint8 * 2)()
    ob.load().grto_photon_quantize(ob.d3(d), out)
    return tuple(out)


def test_store_direction_quantisation(ob):
    """photon_mapping.go:177-191: theta = int(acos(z)*256/pi) capped at 255; phi = int(atan2(y,x)*256/2pi),
    negative -> +256."""
    assert quant(ob, (0, 0, 1)) == (0, 0)
    assert quant(ob, (0, 0, -1)) == (255, 0)          # acos(-1)*256/pi = 256 -> capped
    assert quant(ob, (1, 0, 0)) == (128, 0)
    assert quant(ob, (0, 1, 0)) == (128, 64)
    assert quant(ob, (-1, 1e-9, 0)) == (128, 127)     # just below pi -> 127.99 truncates
    assert quant(ob, (0, -1, 0)) == (128, 192)        # -64 + 256
    assert quant(ob, (1, -1e-9, 0)) == (128, 0)       # int(-1e-10...) truncates toward zero: phi = 0
    s = 1 / math.sqrt(3)
    assert quant(ob, (s, s, s)) == (int(math.acos(s) * 256 / math.pi), 32)


def test_photon_dir_lut(ob):
    """photon_mapping.go:105-122: dir = (sin t cos p, sin t sin p, cos t), t = theta*pi/256, p = phi*2pi/256."""
    d = (C.c_double * 3)()
    ob.load().grto_photon_dir(64, 32, d)
    t, p = 64 * math.pi / 256, 32 * 2 * math.pi / 256
    np.testing.assert_allclose(tuple(d), (math.sin(t) * math.cos(p), math.sin(t) * math.sin(p), math.cos(t)), atol=1e-15)


def grid_map(ob, n=12, power=(0.5, 0.25, 0.125), theta_phi=(255, 0)):
    """n x n photons on the plane y=0 at integer (x, z), all travelling straight down (-z in the
    LUT frame is theta=255: dir ~ (0,0,-1))."""
    xs, zs = np.meshgrid(np.arange(n, dtype=np.float64), np.arange(n, dtype=np.float64))
    pos = np.stack([xs.ravel(), np.zeros(n * n), zs.ravel()], 1)
    pw = np.tile(np.array(power), (n * n, 1))
    tp = np.tile(np.array(theta_phi, np.uint8), (n * n, 1))
    return ob.OraclePhotonMap.load(pos, pw, tp), pos, pw, tp


def test_irradiance_estimate_by_hand(ob):
    m, pos, pw, tp = grid_map(ob)
    q = np.array([[5.0, 0.0, 5.0]])
    up = np.array([[0.0, 0.0, 1.0]])   # photon dir.z = cos(255*pi/256) < 0 -> dir.normal < 0: accepted
    # k = 9: the centre, 4 at d2=1, 4 at d2=2 -> sum = 9*power, r2 = 2 -> / (pi*2)
    e = m.gather(q, up, 0.0, 9)[0]
    np.testing.assert_allclose(e, 9 * np.array([0.5, 0.25, 0.125]) / (math.pi * 2), rtol=1e-14)
    # k = 5: ties at the k-th distance (d2=1) are all kept -> 5 photons; found = 5 < 8 -> UNSCALED sum
    e = m.gather(q, up, 0.0, 5)[0]
    np.testing.assert_allclose(e, 5 * np.array([0.5, 0.25, 0.125]), rtol=1e-14)
    # k = 2: the k-th distance is 1, all four photons at d2 = 1 tie -> 5 photons again
    np.testing.assert_allclose(m.gather(q, up, 0.0, 2)[0], e, rtol=1e-14)
    # radius filter (photon_mapping.go:141): k = 25 nearest, only d2 < 1.5^2 pass -> 9 photons, r2 = 2
    e = m.gather(q, up, 1.5, 25)[0]
    np.testing.assert_allclose(e, 9 * np.array([0.5, 0.25, 0.125]) / (math.pi * 2), rtol=1e-14)
    # direction test (photon_mapping.go:144): normal pointing along the photons' travel -> nothing
    assert (m.gather(q, -up, 0.0, 9)[0] == 0).all()
    # fewer photons than k: all of them
    e = m.gather(q, up, 0.0, 500)[0]
    d2max = 6**2 + 6**2  # the far corner (11, 11)
    np.testing.assert_allclose(e, 144 * np.array([0.5, 0.25, 0.125]) / (math.pi * d2max), rtol=1e-14)


def test_tree_equals_brute_force(ob):
    rs = np.random.RandomState(3)
    pos = rs.uniform(0, 555, (20000, 3))
    pos[:, 1] = np.where(rs.rand(20000) < 0.5, 0.1, pos[:, 1])      # half of them on a plane
    pw = rs.uniform(0, 1, (20000, 3))
    tp = rs.randint(0, 256, (20000, 2)).astype(np.uint8)
    m = ob.OraclePhotonMap.load(pos, pw, tp)
    q = rs.uniform(0, 555, (300, 3))
    n = rs.normal(size=(300, 3))
    for radius, k in ((0.0, 100), (40.0, 50), (0.0, 1)):
        a, b = m.gather(q, n, radius, k), m.gather(q, n, radius, k, brute=True)
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=0)


def test_map_photons_semantics(pkg, ob):
    """scene.go:140-165, 354-424 on the diffuse-wall Cornell box."""
    s = ob.OracleScene(pkg.scenes.cornell_flat(32, 32, "cornell8d"))
    g = ob.OraclePhotonMap.build(s, 5000, False, seed=2)
    assert g.count() == 5000 and g.emitted() > 5000      # some photons die (light, depth cap)
    pos, pw, tp = g.read()
    # stored only on diffuse surfaces: on a wall plane of the box, never on the spheres' interior
    on_wall = ((np.abs(pos - 0.1) < 1e-6) | (np.abs(pos - 555) < 1e-6) | (np.abs(pos[:, [1]] - 0.1) < 1e-6)).any(1)
    assert on_wall.mean() > 0.999
    # power: (15,15,15) * f / (1 - rrp) at the first hit, scaled by 1000/max (scene.go:154,163,416)
    white = 15 * 0.73 / (1 - 0.73) * 1000 / 5000
    assert np.isclose(pw, white, rtol=1e-12).all(1).mean() > 0.3   # first-bounce stores on white walls
    assert (pw > 0).all()
    # Lambert walls can never store (scene.go:395-396): the loop is cut off by max_emit here
    s2 = ob.OracleScene(pkg.scenes.cornell_flat(32, 32, "cornell8"))
    assert ob.OraclePhotonMap.build(s2, 100, False, seed=2, max_emit=20000).count() == 0
    # caustic map: the first hit must be the glass sphere (scene.go:359-363,375), power starts at 1
    c = ob.OraclePhotonMap.build(s, 500, True, seed=2)
    assert c.count() == 500 and c.emitted() > 10 * 500
    cpos, cpw, _ = c.read()
    first_white = 1 * 0.73 / (1 - 0.73) * 1000 / 500  # refracted twice (power unchanged), absorbed on the white floor
    assert np.isclose(cpw, first_white, rtol=1e-12).all(1).mean() > 0.5


def test_irradiance_view_is_the_gather_at_the_first_non_specular_hit(pkg, ob):
    """Oracle wiring of Scene.irradiance (scene.go:184-232) and of trace()'s debugging block (:261-271): on a scene
    without mirrors or glass, one sample per pixel shows IrradianceEst(p, n, 0, k) at the primary hit — lights and
    Lambert walls included, k = 100 for irradiance(), 50 for the debugging block; a miss is black."""
    W, H = 24, 18
    objs = pkg.scenes.cornell_objects("cornell8d")[:6] + pkg.scenes.cornell_objects()[3:4]   # a Lambert copy of the back wall
    flat = pkg.tracer.flatten(W, H, pkg.scenes.cornell_camera(W / H), objs)
    s = ob.OracleScene(flat)
    rs = np.random.RandomState(3)
    n = 4000
    pos = rs.uniform(0, 555, (n, 3))
    pos[: n // 2, 1] = 0.1                                                   # half of them on the floor
    power = rs.uniform(0.0, 0.02, (n, 3))
    tp = rs.randint(0, 256, (n, 2)).astype(np.uint8)
    m = ob.OraclePhotonMap.load(pos, power, tp)
    ids, t, nrm = s.PrimaryHits(1, 7)                                        # the jitter of sample 0
    c = flat[0].cam
    ys, xs = np.mgrid[0:H, 0:W]
    ju = np.vectorize(lambda x, y: ob.load().grto_rng_u01(7, y * W + x, 0, 0))(xs, ys)
    jv = np.vectorize(lambda x, y: ob.load().grto_rng_u01(7, y * W + x, 0, 1))(xs, ys)
    u, v = (xs + ju) / W, (ys + jv) / H
    D = (np.array(c.lower_left)[None, None] + np.array(c.horizontal)[None, None] * u[..., None]
         + np.array(c.vertical)[None, None] * v[..., None] - np.array(c.origin)[None, None])
    P = np.array(c.origin)[None, None] + D * t[..., None]
    hit = ids >= 0
    assert hit.any() and (~hit).any() and (ids == 0).any()                   # the light is among the hits
    for mode, k in ((1, 100), (2, 50)):
        want = np.zeros((H, W, 3))
        want[hit] = m.gather(P[hit], nrm[hit], 0.0, k)
        s.SetView(m, mode)
        got = s.RenderAccum(1, seed=7)
        s.SetView(None, 0)
        assert np.allclose(got, want, rtol=1e-12, atol=0) and got[hit].max() > 0
