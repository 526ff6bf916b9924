"""Scene literals used by tests and bench.py (data, not code, from cmd/raytracer/main.go:32-59).

cornell8   the README "final scene": 6 AABB walls/light + mirror sphere + glass sphere
cornell8d  same geometry with DiffuseMaterial walls (material.go:20) so photons can be stored
           (with Lambert walls the photon loops never terminate)
cornell8n  every object NormalMaterial (material.go:15): RNG-free shading given fixed jitter
spheresN   config C5: N spheres, centres uniform in [0,555]^3 and radii uniform in
           [0.5,1.5] from a counter RNG (numpy Philox, seed 1), material by index mod 4 ->
           {Lambert white, Lambert red, mirror, glass 1.53}, plus the cornell8 light box and a floor
"""
from . import tracer as T

EYE, LOOKAT, VUP, VFOV = (278.0, 273.0, -800.0), (278.0, 278.0, 1.0), (0.0, 1.0, 0.0), 40.0


def cornell_objects(variant="cornell8"):
    red = T.NewColor(0.65, 0.05, 0.05)
    green = T.NewColor(0.12, 0.45, 0.15)
    white = T.NewColor(0.73, 0.73, 0.73)
    wall = T.DiffuseMaterial if variant == "cornell8d" else T.LambertMaterial
    mat_red, mat_green, mat_white = wall(red), wall(green), wall(white)
    mat_light = T.LightMaterial(T.NewColor(0.2, 0.2, 0.2), 10)
    mat_glass = T.DielectricMaterial(1.53)
    mat_mirror = T.MetalicMaterial(T.NewColor(1, 1, 1), 1, 0)
    if variant == "cornell8n":
        n = T.NormalMaterial()
        mat_red = mat_green = mat_white = mat_light = mat_glass = mat_mirror = n
    return [
        T.NewAABB((113, 548, 127), (443, 548.1, 432), mat_light),
        T.NewAABB((0, 0, 0), (555, 0.1, 555), mat_white),      # floor
        T.NewAABB((0, 555, 0), (555, 555.1, 555), mat_white),  # ceiling
        T.NewAABB((0, 0, 555), (555, 555, 555.1), mat_white),  # back wall
        T.NewAABB((555, 0, 0), (555.1, 555, 555), mat_red),    # left wall
        T.NewAABB((0, 0, 0), (0.1, 555, 555), mat_green),      # right wall
        T.NewSphere((278 + 110, 90, 227 + 120), 90, mat_mirror),
        T.NewSphere((278 - 110, 90, 227 - 40), 90, mat_glass),
    ]


def cornell_camera(aspect=1.0):
    return T.NewCamera(EYE, LOOKAT, VUP, VFOV, aspect)


def cornell_flat(width, height, variant="cornell8"):
    """(desc, prims, n_prims, mats, n_mats) without touching any library."""
    return T.flatten(width, height, cornell_camera(width / height), cornell_objects(variant))


def cornell(width, height, variant="cornell8"):
    """tracer.Scene on the GPU; aspect = W/H as in SURVEY.md §8d (main.go hard-codes 1)."""
    return T.NewScene(width, height, cornell_camera(width / height), cornell_objects(variant))


def random_objects(n_spheres, n_boxes, seed=7, diffuse=True):
    """Mixed spheres/boxes inside the Cornell room for intersection parity tests.
    diffuse=False leaves out the plain DiffuseMaterial (no shadow rays: the lean hierarchy kernel)."""
    import random

    rnd = random.Random(seed)
    mats = [T.LambertMaterial((0.73, 0.73, 0.73)), T.LambertMaterial((0.65, 0.05, 0.05)),
            T.MetalicMaterial((0.9, 0.9, 0.9), 0.9, 0.2), T.DielectricMaterial(1.53),
            T.DiffuseMaterial((0.3, 0.5, 0.7)) if diffuse else T.MetalicMaterial((0.3, 0.5, 0.7), 0.8, 0.0)]
    objs = cornell_objects()[:6]
    for i in range(n_spheres):
        r = rnd.uniform(10, 60)
        c = (rnd.uniform(r, 555 - r), rnd.uniform(r, 555 - r), rnd.uniform(r, 555 - r))
        objs.append(T.NewSphere(c, r, mats[i % len(mats)]))
    for i in range(n_boxes):
        lo = (rnd.uniform(10, 450), rnd.uniform(10, 450), rnd.uniform(10, 450))
        sz = (rnd.uniform(5, 90), rnd.uniform(5, 90), rnd.uniform(5, 90))
        objs.append(T.NewAABB(lo, tuple(a + b for a, b in zip(lo, sz)), mats[(i + 2) % len(mats)]))
    return objs


def spheres_flat(n_spheres, width, height, seed=1, accel=0, as_boxes=False):
    """(desc, prims, n_prims, mats, n_mats) of the C5 scene.  Objects 0 and 1 are the cornell8
    light box and floor (main.go:41-42), objects 2.. the spheres; as_boxes=True exports every sphere
    as its bounding AABB instead (the "AABB-bounded" variant)."""
    import numpy as np

    g = np.random.Generator(np.random.Philox(seed))
    c = g.uniform(0.0, 555.0, (n_spheres, 3))
    r = g.uniform(0.5, 1.5, n_spheres)
    mats = [T.LightMaterial(T.NewColor(0.2, 0.2, 0.2), 10), T.LambertMaterial(T.NewColor(0.73, 0.73, 0.73)),
            T.LambertMaterial(T.NewColor(0.65, 0.05, 0.05)), T.MetalicMaterial(T.NewColor(1, 1, 1), 1, 0),
            T.DielectricMaterial(1.53)]
    a = np.zeros(n_spheres + 2, T.PRIM_DTYPE)
    a["kind"][0:2] = T.GRT_KIND_AABB
    a["a"][0], a["b"][0], a["material"][0] = (113, 548, 127), (443, 548.1, 432), 0
    a["a"][1], a["b"][1], a["material"][1] = (0, 0, 0), (555, 0.1, 555), 1
    if as_boxes:
        a["kind"][2:] = T.GRT_KIND_AABB
        a["a"][2:] = c - r[:, None]
        a["b"][2:] = c + r[:, None]
    else:
        a["kind"][2:] = T.GRT_KIND_SPHERE
        a["a"][2:] = c
        a["radius"][2:] = r
    a["material"][2:] = 1 + (np.arange(n_spheres) % 4)
    return T.flatten_arrays(width, height, cornell_camera(width / height), a, mats, accel=accel)


def spheres(n_spheres, width, height, seed=1, accel=0, as_boxes=False):
    return T.Scene.from_flat(spheres_flat(n_spheres, width, height, seed, accel, as_boxes))
