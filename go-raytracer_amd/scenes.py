"""Scene literals used by tests and bench.py (data, not code, from cmd/raytracer/main.go:32-59).

cornell8   the README "final scene": 6 AABB walls/light + mirror sphere + glass sphere
cornell8d  same geometry with DiffuseMaterial walls (material.go:20) so photons can be stored
           (with Lambert walls the photon loops never terminate)
cornell8n  every object NormalMaterial (material.go:15): RNG-free shading given fixed jitter
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


def random_objects(n_spheres, n_boxes, seed=7):
    """Mixed spheres/boxes inside the Cornell room for intersection parity tests."""
    import random

    rnd = random.Random(seed)
    mats = [T.LambertMaterial((0.73, 0.73, 0.73)), T.LambertMaterial((0.65, 0.05, 0.05)),
            T.MetalicMaterial((0.9, 0.9, 0.9), 0.9, 0.2), T.DielectricMaterial(1.53),
            T.DiffuseMaterial((0.3, 0.5, 0.7))]
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
