"""Known-answer tests that pin the float64 oracle to the reference's formulas.

The reference has no tests; every expected value below is derived by hand
from the cited source lines, not from running any implementation.
"""
import ctypes as C
import math

import numpy as np
import pytest

INF, NAN = float("inf"), float("nan")
MAXF = 1.7976931348623157e308


def sphere(pkg, c, r):
    p = pkg._capi.PrimRec()
    p.kind, p.material, p.radius = 0, 0, r
    p.a[:] = c
    return p


def box(pkg, lo, hi):
    p = pkg._capi.PrimRec()
    p.kind, p.material = 1, 0
    p.a[:] = lo
    p.b[:] = hi
    return p


def hit(ob, prim, o, d, tmin=0.001, tmax=MAXF):
    return ob.load().grto_prim_hit(C.byref(prim), ob.d3(o), ob.d3(d), tmin, tmax)


def surf(ob, prim, p):
    out = (C.c_double * 3)()
    ob.load().grto_prim_surface(C.byref(prim), ob.d3(p), out)
    return tuple(out)


# ---- Sphere.Hit, sphere.go:23-46 ---------------------------------------------------
def test_sphere_hit_roots(pkg, ob):
    s = sphere(pkg, (0, 0, 0), 1)
    assert hit(ob, s, (0, 0, -5), (0, 0, 1)) == 4.0          # near root
    # unnormalised direction: a=4, halfB=-10, c=24, disc=4 -> (10-2)/4
    assert hit(ob, s, (0, 0, -5), (0, 0, 2)) == 2.0
    assert hit(ob, s, (0, 0, 0), (0, 0, 1)) == 1.0           # inside: near root -1 < tMin, far root
    assert hit(ob, s, (0, 2, -5), (0, 0, 1)) == -1.0         # disc < 0
    assert hit(ob, s, (0, 0, -5), (0, 0, 1), tmax=3.0) == -1.0   # 4 and 6 both >= tMax
    assert hit(ob, s, (0, 0, -5), (0, 0, 1), tmax=4.0) == -1.0   # strict t < tMax (sphere.go:37)
    assert hit(ob, s, (0, 0, -5), (0, 0, 1), tmin=4.0) == 6.0    # strict t > tMin -> far root


def test_sphere_surface_scales_by_radius(pkg, ob):
    # sphere.go:57: (p - C).Scale(R).Unit(): a negative radius flips the normal
    assert surf(ob, sphere(pkg, (1, 2, 3), 2), (1, 2, 5)) == (0.0, 0.0, 1.0)
    assert surf(ob, sphere(pkg, (1, 2, 3), -2), (1, 2, 5)) == (-0.0, -0.0, -1.0)


# ---- AABB.Hit, aabb.go:25-45 ---------------------------------------------------------
def test_aabb_hit_entry_distance(pkg, ob):
    b = box(pkg, (-1, -1, -1), (1, 1, 1))
    assert hit(ob, b, (0, 0, -5), (0, 0, 1)) == 4.0
    assert hit(ob, b, (0, 0, -5), (0, 0, 2)) == 2.0
    assert hit(ob, b, (0, 3, -5), (0, 0, 1)) == -1.0
    # origin inside: returns the NEGATIVE entry distance, which intersect() rejects (ht > 0)
    assert hit(ob, box(pkg, (-2, -2, -2), (2, 2, 2)), (0, 0, 0), (0, 0, 1)) == -2.0
    # entry below tMin but exit above it: aabb.go:44 returns t0, not max(tMin, t0)
    t = hit(ob, b, (0, 0, -1.0005), (0, 0, 1))
    assert 0 < t < 0.001 and abs(t - 0.0005) < 1e-12
    # interval closed by tMax: ltMax = 3 <= ltMin = 4
    assert hit(ob, b, (0, 0, -5), (0, 0, 1), tmax=3.0) == -1.0
    assert hit(ob, b, (0, 0, -5), (0, 0, 1), tmax=4.0) == -1.0  # ltMax <= ltMin is inclusive


def test_aabb_hit_go_nan_rules(pkg, ob):
    """0/0 on a slab: Go's math.Min/Max let +-Inf beat NaN, else propagate NaN (vec3.go:104-111)."""
    b = box(pkg, (-1, -1, -1), (1, 1, 1))
    # O.x == max.x, D.x == 0: n.x=-inf, f.x=NaN -> Min=-inf, Max=NaN -> t1=NaN, `t1 < ltMax`
    # false, interval stays open -> returns t0 = 4: a HIT
    assert hit(ob, b, (1, 0, -5), (0, 0, 1)) == 4.0
    # O.x == min.x: n.x=NaN, f.x=+inf -> Min=NaN -> t0=NaN -> returned NaN, rejected by ht > 0
    assert math.isnan(hit(ob, b, (-1, 0, -5), (0, 0, 1)))
    # axis-parallel ray strictly inside the slab: -inf/+inf, ordinary hit
    assert hit(ob, b, (0.5, 0, -5), (0, 0, 1)) == 4.0
    # axis-parallel ray outside the slab: n.x=f.x=+inf -> t0=+inf -> ltMax <= ltMin
    assert hit(ob, b, (2, 0, -5), (0, 0, 1)) == -1.0


def test_aabb_surface_cascade(pkg, ob):
    b = box(pkg, (0, 0, 0), (10, 20, 30))
    assert surf(ob, b, (0, 5, 5)) == (-1, 0, 0)
    assert surf(ob, b, (10, 5, 5)) == (1, 0, 0)
    assert surf(ob, b, (5, 0, 5)) == (0, -1, 0)
    assert surf(ob, b, (5, 20, 5)) == (0, 1, 0)
    assert surf(ob, b, (5, 5, 0)) == (0, 0, -1)
    assert surf(ob, b, (5, 5, 30)) == (0, 0, 1)
    assert surf(ob, b, (5, 5, 5)) == (0, 0, 0)               # no case matches (aabb.go:57-70)
    assert surf(ob, b, (0, 0, 0)) == (-1, 0, 0)              # X is tested first on corners
    assert surf(ob, b, (5, 20, 30)) == (0, 1, 0)             # Y before Z on edges
    assert surf(ob, b, (0.00009, 5, 5)) == (-1, 0, 0)        # eps = 1e-4 (aabb.go:56)
    assert surf(ob, b, (0.00011, 5, 5)) == (0, 0, 0)


# ---- Scene.intersect, scene.go:169-181 -------------------------------------------------
def test_intersect_order_and_ties(pkg, ob):
    T = pkg.tracer
    m = T.LambertMaterial((1, 1, 1))
    objs = [T.NewAABB((-1, -1, 4), (1, 1, 6), m), T.NewAABB((-1, -1, 4), (1, 1, 6), m),
            T.NewSphere((0, 0, 2), 1, m), T.NewSphere((0, 0, -10), 1, m)]
    s = ob.OracleScene(T.flatten(8, 8, pkg.scenes.cornell_camera(), objs))
    assert s.Intersect((0, 0, -5), (0, 0, 1)) == (2, 6.0)     # sphere at z=1 is nearest
    assert s.Intersect((0, 3, -5), (0, 0, 1))[0] == -1
    s2 = ob.OracleScene(T.flatten(8, 8, pkg.scenes.cornell_camera(), objs[:2]))
    assert s2.Intersect((0, 0, -5), (0, 0, 1)) == (0, 9.0)    # exact tie: lowest index wins
    # bias: a hit at t <= 0.001 on a sphere is skipped (tMin, scene.go:14,170)
    s3 = ob.OracleScene(T.flatten(8, 8, pkg.scenes.cornell_camera(), [T.NewSphere((0, 0, 0), 1, m)]))
    i, t = s3.Intersect((0, 0, -1), (0, 0, 1))
    assert (i, t) == (0, 2.0)


# ---- Reflect / Refract / SampleHemisphereNormal, vec3.go:129-159,202-218 -------------
def _call3(fn, *args):
    out = (C.c_double * 3)()
    r = fn(*args, out)
    return r, np.array(out)


def test_reflect(ob):
    s = 1 / math.sqrt(2)
    _, r = _call3(ob.load().grto_reflect, ob.d3((s, -s, 0)), ob.d3((0, 1, 0)))
    np.testing.assert_allclose(r, (s, s, 0), atol=1e-15)


def test_refract(ob):
    f = ob.load().grto_refract
    ratio = 1 / 1.5
    # normal incidence: cosi=1, r0=((1-r)/(1+r))^2=0.04; xi=0.5 -> refracts straight through
    ok, r = _call3(f, ob.d3((0, -1, 0)), ob.d3((0, 1, 0)), ratio, 0.5)
    assert ok == 1
    np.testing.assert_allclose(r, (0, -1, 0), atol=1e-15)
    # Schlick roulette: r0 = 0.04 > xi = 0.01 -> reflect (returns false, vec3.go:152-153)
    ok, _ = _call3(f, ob.d3((0, -1, 0)), ob.d3((0, 1, 0)), ratio, 0.01)
    assert ok == 0
    # from inside (v.n >= 0): ratio inverted to 1.5; 60 deg -> 1.5*sin > 1 -> TIR for any xi
    v = (math.sin(math.radians(60)), math.cos(math.radians(60)), 0)
    ok, _ = _call3(f, ob.d3(v), ob.d3((0, 1, 0)), ratio, 0.999999)
    assert ok == 0
    # Snell at 30 deg from outside: sin(t) = sin(30)/1.5
    v = (math.sin(math.radians(30)), -math.cos(math.radians(30)), 0)
    ok, r = _call3(f, ob.d3(v), ob.d3((0, 1, 0)), ratio, 0.999999)
    assert ok == 1
    st = 0.5 / 1.5
    np.testing.assert_allclose(r, (st, -math.sqrt(1 - st * st), 0), atol=1e-12)


def test_sample_hemisphere_normal(ob):
    f = ob.load().grto_sample_hemisphere_normal
    # n=(0,1,0): |n.x| <= 0.1 -> u = ((1,0,0) x n)^ = (0,0,1), v = n x u = (1,0,0)
    np.testing.assert_allclose(_call3(f, ob.d3((0, 1, 0)), 0.0, 0.0)[1], (0, 1, 0), atol=1e-15)
    np.testing.assert_allclose(_call3(f, ob.d3((0, 1, 0)), 0.0, 1.0)[1], (0, 0, 1), atol=1e-15)
    np.testing.assert_allclose(_call3(f, ob.d3((0, 1, 0)), 0.25, 1.0)[1], (1, 0, 0), atol=1e-15)
    # n=(1,0,0): |n.x| > 0.1 -> u = ((0,1,0) x n)^ = (0,0,-1), v = n x u = (0,1,0)
    np.testing.assert_allclose(_call3(f, ob.d3((1, 0, 0)), 0.0, 1.0)[1], (0, 0, -1), atol=1e-15)
    np.testing.assert_allclose(_call3(f, ob.d3((1, 0, 0)), 0.25, 1.0)[1], (0, 1, 0), atol=1e-15)
    # cosine lobe: r2 = 0.75 -> cos(theta) = sqrt(1 - r2) = 0.5
    d = _call3(f, ob.d3((0, 0, 1)), 0.1, 0.75)[1]
    assert abs(d[2] - 0.5) < 1e-15 and abs(np.linalg.norm(d) - 1) < 1e-15


# ---- Camera, camera.go:16-39 -----------------------------------------------------------
def test_camera_kat(pkg, ob):
    """Default camera at aspect 1: values computed in float64 in SURVEY.md §8d."""
    cam = pkg._capi.Camera()
    ob.load().grto_camera_new(ob.d3(pkg.scenes.EYE), ob.d3(pkg.scenes.LOOKAT), ob.d3(pkg.scenes.VUP), 40.0, 1.0,
                              C.byref(cam))
    assert list(cam.origin) == [278.0, 273.0, -800.0]
    assert list(cam.lower_left) == [278.3639702342662, 273.3702052190618, -799.0022914116781]
    assert [abs(v) for v in cam.horizontal] == [0.7279404685324047, 0.0, 0.0] and cam.horizontal[0] < 0
    assert cam.vertical[1] == -0.727926286837004 and cam.vertical[2] == 0.00454385946839578
    # the host mirror (tracer.NewCamera) performs the same float64 operations
    cam2 = pkg.scenes.cornell_camera(1.0)
    for f in ("origin", "lower_left", "horizontal", "vertical"):
        assert list(getattr(cam, f)) == list(getattr(cam2, f))
    # Camera.Ray is NOT normalised: the centre ray has length ~1
    o, d = (C.c_double * 3)(), (C.c_double * 3)()
    ob.load().grto_camera_ray(C.byref(cam), 0.5, 0.5, o, d)
    assert abs(math.sqrt(sum(x * x for x in d)) - 1.0) < 1e-12 and d[2] > 0.99


# ---- WriteColor + tone ops, scene.go:67-74,98-99 ---------------------------------------
def test_write_color(ob):
    bgr = (C.c_uint8 * 3)()
    ob.load().grto_write_color(ob.d3((0.25, 1.0, 4.0)), bgr)
    # sqrt -> (0.5, 1, 2) -> clamp (0.5, 1, 1) -> uint8(255.99 c) = (127, 255, 255); order B,G,R
    assert tuple(bgr) == (255, 255, 127)
    ob.load().grto_write_color(ob.d3((0.0, 1e-12, 0.0000153)), bgr)
    assert tuple(bgr) == (1, 0, 0)  # 255.99*sqrt(1.53e-5) = 1.0013 -> 1


def test_render_layout_bgrx(pkg, ob):
    """pixels[y*pitch + x*bpp + {0,1,2}] = B,G,R; 4th byte untouched (scene.go:71-73,85,128)."""
    flat = pkg.scenes.cornell_flat(16, 16, "cornell8n")
    s = ob.OracleScene(flat)
    pitch = 4 * 16 + 8
    px = np.full(16 * pitch, 0xAB, np.uint8)
    s.Render(px, pitch, 1, seed=3)
    img = px.reshape(16, pitch)
    assert (img[:, 3:64:4] == 0xAB).all() and (img[:, 64:] == 0xAB).all()
    assert (img[:, 0:64:4] != 0xAB).any()
