"""Host-side mirror of the reference's `pkg/tracer` interface, backed by libgrt_hip.so.

Same names, argument meaning and behaviour as the Go package so parity tests read like code
written against the reference:

    NewCamera(eye, lookat, vup, vfov, aspect)            camera.go:16
    NewSphere(center, radius, mat) / NewAABB(min, max, mat)   sphere.go:17 / aabb.go:18
    NormalMaterial / DiffuseMaterial / LambertMaterial / MetalicMaterial /
    DielectricMaterial / LightMaterial                   material.go:15-41
    NewScene(width, height, cam, objects, ...)           scene.go:35
    Scene.Render(pixels, pitch, samples)                 scene.go:79

The tested native host (C++) lives in host/tracer.hpp; this module is the ctypes route used
by tests/ and bench.py.  Nothing here computes pixels: every call goes through the C ABI.
"""
import ctypes as C
import math
import time

import numpy as np

from . import _capi
from ._capi import (GRT_KIND_AABB, GRT_KIND_SPHERE, Camera, MaterialRec, Opts, PrimRec, SceneDesc,
                    check, load)

BIAS = 0.001  # scene.go:14
MAX_DEPTH = 6  # scene.go:62


# ---- materials (material.go:4-41) ---------------------------------------------------
class Material:
    __slots__ = ("Color", "RefrIndex", "Reflectivity", "Roughness", "Emittance", "Transparent",
                 "Lambert", "Normal")

    def __init__(self, color, refr_index, reflectivity, roughness, emittance, transparent, lambert,
                 normal):
        self.Color = tuple(float(c) for c in color)
        self.RefrIndex = float(refr_index)
        self.Reflectivity = float(reflectivity)
        self.Roughness = float(roughness)
        self.Emittance = float(emittance)
        self.Transparent = bool(transparent)
        self.Lambert = bool(lambert)
        self.Normal = bool(normal)

    def key(self):
        return tuple(getattr(self, s) for s in self.__slots__)


def NewColor(r, g, b):  # color.go:15
    return (float(r), float(g), float(b))


def NormalMaterial():  # material.go:15
    return Material((0, 0, 0), 1, 0, 0, 0, False, False, True)


def DiffuseMaterial(color):  # material.go:20
    return Material(color, 1, 0, 0, 0, False, False, False)


def LambertMaterial(albedo):  # material.go:25
    return Material(albedo, 1, 0, 0, 0, False, True, False)


def MetalicMaterial(albedo, reflectivity, roughness):  # material.go:30
    return Material(albedo, 1, reflectivity, roughness, 0, False, False, False)


def DielectricMaterial(index):  # material.go:35
    return Material((0, 0, 0), index, 0, 0, 0, True, False, False)


def LightMaterial(intensity, emittance):  # material.go:39
    return Material(intensity, 1, 0, 0, emittance, False, False, False)


# ---- primitives (sphere.go:10-19, aabb.go:10-21) ------------------------------------
class Sphere:
    def __init__(self, center, radius, mat):
        self.Center = tuple(float(c) for c in center)
        self.Radius = float(radius)
        self.Mat = mat

    def Material(self):
        return self.Mat

    def Pos(self):
        return self.Center


class AABB:
    def __init__(self, mn, mx, mat):
        self.MinBound = tuple(float(c) for c in mn)
        self.MaxBound = tuple(float(c) for c in mx)
        self.Center = tuple((a + b) * 0.5 for a, b in zip(self.MaxBound, self.MinBound))  # aabb.go:19
        self.Mat = mat

    def Material(self):
        return self.Mat

    def Pos(self):
        return self.Center


def NewSphere(center, radius, mat):
    return Sphere(center, radius, mat)


def NewAABB(mn, mx, mat):
    return AABB(mn, mx, mat)


# ---- camera (camera.go:16-30) -------------------------------------------------------
def _sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def _scale(a, k):
    return (a[0] * k, a[1] * k, a[2] * k)


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def _unit(a):
    k = 1.0 / math.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])
    return _scale(a, k)


def NewCamera(eye, lookat, vup, vfov, aspect):
    """Float64, operation order of camera.go:16-30 (host-side, once per scene)."""
    theta = vfov * math.pi / 180
    half_h = math.tan(theta / 2)
    half_w = aspect * half_h
    eye = tuple(float(c) for c in eye)
    w = _unit(_sub(eye, lookat))
    u = _unit(_cross(vup, w))
    v = _unit(_cross(u, w))
    cam = Camera()
    cam.origin[:] = eye
    cam.lower_left[:] = _sub(_sub(_sub(eye, _scale(u, half_w)), _scale(v, half_h)), w)
    cam.horizontal[:] = _scale(u, 2 * half_w)
    cam.vertical[:] = _scale(v, 2 * half_h)
    return cam


# ---- flattening: Scene.Objects []Hitable -> POD arrays ------------------------------
def flatten(width, height, cam, objects, max_depth=MAX_DEPTH, bias=BIAS, accel=0):
    """The type switch the cgo shim performs: one grt_prim per object in
    Scene.Objects order (the index is the object id), materials de-duplicated by value."""
    mats, mat_index = [], {}
    prims = (PrimRec * max(1, len(objects)))()
    for i, o in enumerate(objects):
        m = o.Material()
        k = m.key()
        if k not in mat_index:
            mat_index[k] = len(mats)
            mats.append(m)
        p = prims[i]
        p.material = mat_index[k]
        if isinstance(o, Sphere):
            p.kind = GRT_KIND_SPHERE
            p.a[:] = o.Center
            p.radius = o.Radius
        elif isinstance(o, AABB):
            p.kind = GRT_KIND_AABB
            p.a[:] = o.MinBound
            p.b[:] = o.MaxBound
        else:  # e.g. tracer.List (list.go:6): no flat form
            raise TypeError(f"object {i}: unsupported Hitable {type(o).__name__}")
    mrec = (MaterialRec * max(1, len(mats)))()
    for i, m in enumerate(mats):
        r = mrec[i]
        r.color[:] = m.Color
        r.refr_index, r.reflectivity, r.roughness = m.RefrIndex, m.Reflectivity, m.Roughness
        r.emittance = m.Emittance
        r.transparent, r.lambert, r.normal = int(m.Transparent), int(m.Lambert), int(m.Normal)
    desc = SceneDesc()
    desc.width, desc.height, desc.max_depth, desc.bias = width, height, max_depth, bias
    desc.accel = accel
    desc.cam = cam
    return desc, prims, len(objects), mrec, len(mats)


PRIM_DTYPE = np.dtype([("kind", "<i4"), ("material", "<i4"), ("a", "<f8", 3), ("b", "<f8", 3), ("radius", "<f8")])
assert PRIM_DTYPE.itemsize == C.sizeof(PrimRec)


def flatten_arrays(width, height, cam, prim_array, materials, max_depth=MAX_DEPTH, bias=BIAS, accel=0):
    """The same POD arrays from a numpy record array (PRIM_DTYPE) and a list of Material: the bulk
    route for scenes of millions of objects (config C5), where one Python object per sphere is
    too slow.  prim_array[i] is object i of Scene.Objects."""
    prim_array = np.ascontiguousarray(prim_array, PRIM_DTYPE)
    prims = (PrimRec * max(1, prim_array.shape[0])).from_buffer(prim_array) if prim_array.shape[0] else (PrimRec * 1)()
    mrec = (MaterialRec * max(1, len(materials)))()
    for i, m in enumerate(materials):
        r = mrec[i]
        r.color[:] = m.Color
        r.refr_index, r.reflectivity, r.roughness = m.RefrIndex, m.Reflectivity, m.Roughness
        r.emittance = m.Emittance
        r.transparent, r.lambert, r.normal = int(m.Transparent), int(m.Lambert), int(m.Normal)
    desc = SceneDesc()
    desc.width, desc.height, desc.max_depth, desc.bias = width, height, max_depth, bias
    desc.accel = accel
    desc.cam = cam
    return desc, prims, int(prim_array.shape[0]), mrec, len(materials)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Scene:
    """tracer.Scene (scene.go:17-27) staged on the GPU behind a grt_scene handle."""

    def __init__(self, width, height, cam, objects, globalPmap=None, causticPmap=None,
                 max_depth=MAX_DEPTH, bias=BIAS, accel=0, flat=None):
        self.W, self.H, self.Cam = int(width), int(height), cam
        self.Objects = list(objects) if objects is not None else None
        self.globalPmap, self.causticPmap = globalPmap, causticPmap
        self.maxDepth = max_depth
        self.flat = flat if flat is not None else flatten(self.W, self.H, cam, self.Objects, max_depth, bias, accel)
        self._lib = load()
        h = C.c_void_p()
        desc, prims, n_prims, mats, n_mats = self.flat
        check(self._lib.grt_scene_create(C.byref(desc), prims, n_prims, mats, n_mats, C.byref(h)))
        self.handle = h

    @classmethod
    def from_flat(cls, flat):
        """Scene from the POD arrays of flatten() / flatten_arrays()."""
        desc = flat[0]
        return cls(desc.width, desc.height, desc.cam, None, max_depth=desc.max_depth, bias=desc.bias, flat=flat)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.grt_scene_destroy(self.handle)
            self.handle = None

    __del__ = close

    @staticmethod
    def opts(seed=None, first_device=0, n_devices=1, chunk_spp=0, variant=0):
        """variant: kernel variant for A/B timing and cross-checks (grt_opts.flags bits 0-3; 0 = default)."""
        o = Opts()
        # the reference seeds with time.Now().Unix() (scene.go:114)
        o.seed = int(time.time()) if seed is None else int(seed)
        o.first_device, o.n_devices, o.chunk_spp, o.flags = first_device, n_devices, chunk_spp, int(variant) & 15
        return o

    # scene.go:79
    def Render(self, pixels, pitch, samples, seed=None, **kw):
        """pixels: writable buffer of H*pitch bytes (numpy uint8 / bytearray); B,G,R at
        pixels[y*pitch + x*bpp + {0,1,2}], bpp = pitch // W."""
        buf = np.frombuffer(pixels, dtype=np.uint8) if not isinstance(pixels, np.ndarray) else pixels
        if buf.size < self.H * pitch:
            raise ValueError("pixels buffer smaller than H*pitch")
        o = self.opts(seed, **kw)
        check(self._lib.grt_render(self.handle, C.byref(o), _ptr(buf), pitch, samples))

    def RenderAccum(self, samples, seed=None, **kw):
        """Linear RGB sums, float32 [H, W, 3] (what scene.go:96 accumulates)."""
        out = np.empty((self.H, self.W, 3), np.float32)
        o = self.opts(seed, **kw)
        check(self._lib.grt_render_accum(self.handle, C.byref(o), _ptr(out), samples))
        return out

    def RenderPhotonMapped(self, pixels, pitch, samples, seed=None):
        """Render with the photon block of trace() enabled (scene.go:308-320): needs
        self.globalPmap and self.causticPmap built (scene.go:22-23, mapPhotons scene.go:140-165)."""
        buf = np.frombuffer(pixels, dtype=np.uint8) if not isinstance(pixels, np.ndarray) else pixels
        if buf.size < self.H * pitch:
            raise ValueError("pixels buffer smaller than H*pitch")
        o = self.opts(seed)
        check(self._lib.grt_render_pm(self.handle, C.byref(o), self.globalPmap.handle, self.causticPmap.handle, _ptr(buf),
                                      pitch, samples))

    def RenderAccumPhotonMapped(self, samples, seed=None):
        out = np.empty((self.H, self.W, 3), np.float32)
        o = self.opts(seed)
        check(self._lib.grt_render_accum_pm(self.handle, C.byref(o), self.globalPmap.handle, self.causticPmap.handle,
                                            _ptr(out), samples))
        return out

    def RenderAccumPhotonView(self, pmap, samples, seed=None, mode=_capi.GRT_PMVIEW_IRRADIANCE):
        """Render through Scene.irradiance (scene.go:184-232) or, mode GRT_PMVIEW_FIRST_HIT, trace()'s debugging
        block (scene.go:261-271): linear RGB sums [H, W, 3]."""
        out = np.empty((self.H, self.W, 3), np.float32)
        o = self.opts(seed)
        check(self._lib.grt_render_accum_pmview(self.handle, C.byref(o), pmap.handle, mode, _ptr(out), samples))
        return out

    def RenderPhotonView(self, pmap, pixels, pitch, samples, seed=None, mode=_capi.GRT_PMVIEW_IRRADIANCE):
        buf = np.frombuffer(pixels, dtype=np.uint8) if not isinstance(pixels, np.ndarray) else pixels
        if buf.size < self.H * pitch:
            raise ValueError("pixels buffer smaller than H*pitch")
        o = self.opts(seed)
        check(self._lib.grt_render_pmview(self.handle, C.byref(o), pmap.handle, mode, _ptr(buf), pitch, samples))

    def TraceSamples(self, pixel_xy, samples, seed=None):
        xy = np.ascontiguousarray(pixel_xy, np.int32).reshape(-1, 2)
        rgb = np.empty((xy.shape[0], samples, 3), np.float32)
        nr = np.empty((xy.shape[0], samples), np.int32)
        o = self.opts(seed)
        check(self._lib.grt_trace_samples(self.handle, C.byref(o), _ptr(xy), xy.shape[0], samples,
                                          _ptr(rgb), _ptr(nr)))
        return rgb, nr

    def PrimaryHits(self, jitter_mode=_capi.GRT_JITTER_CENTER, seed=1, mode=0):
        """mode: 0 = FP32 with FP64 re-test of ambiguous lanes, 1 = FP32 only (the renderer's own
        intersect), 2 = FP64 only; on linear-scan and hierarchy scenes alike."""
        n = self.W * self.H
        ids = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        nrm = np.empty((n, 3), np.float32)
        fb = C.c_uint64(0)
        o = self.opts(seed)
        o.flags = (int(mode) & 3) << 8
        check(self._lib.grt_primary_hits(self.handle, C.byref(o), jitter_mode, _ptr(ids), _ptr(t),
                                         _ptr(nrm), C.byref(fb)))
        return ids.reshape(self.H, self.W), t.reshape(self.H, self.W), nrm.reshape(self.H, self.W, 3), fb.value

    def last_ray_count(self):
        return int(self._lib.grt_last_ray_count())

    def last_kernel_ms(self):
        return float(self._lib.grt_last_kernel_ms())


def NewScene(width, height, cam, objects, globalPmap=None, causticPmap=None):  # scene.go:35
    return Scene(width, height, cam, objects, globalPmap, causticPmap)


# ---- photon map (photon_mapping.go:84-209, scene.go:140-165) -------------------------
class PhotonMap:
    """tracer.PhotonMap: maxPhotons on the host, storage + kd-tree on the GPU (grt_pmap)."""

    def __init__(self, maxPhotons=0, handle=None):
        self.maxPhotons = int(maxPhotons)
        self.handle = handle
        self._lib = load()

    def close(self):
        if getattr(self, "handle", None):
            self._lib.grt_photon_free(self.handle)
            self.handle = None

    __del__ = close

    def build(self, scene, caustics=False, seed=1):
        """mapPhotons for this map (scene.go:152-160) + ScalePhotonPower(1000/max) (:163-164)."""
        self.close()
        h = C.c_void_p()
        o = Scene.opts(seed)
        check(self._lib.grt_photon_build(scene.handle, C.byref(o), self.maxPhotons, int(caustics), C.byref(h)))
        self.handle = h
        return self

    @classmethod
    def load(cls, pos, power, theta_phi):
        pos = np.ascontiguousarray(pos, np.float32)
        power = np.ascontiguousarray(power, np.float32)
        tp = np.ascontiguousarray(theta_phi, np.uint8)
        h = C.c_void_p()
        o = Scene.opts(1)
        check(load().grt_photon_load(C.byref(o), _ptr(pos), _ptr(power), _ptr(tp), pos.shape[0], C.byref(h)))
        return cls(pos.shape[0], h)

    def count(self):
        return int(self._lib.grt_photon_count(self.handle))

    def emitted(self):
        return int(self._lib.grt_photon_emitted(self.handle))

    def device(self):
        """HIP ordinal of the device the map lives on."""
        return int(self._lib.grt_photon_device(self.handle))

    def read(self):
        n = self.count()
        pos, power, tp = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32), np.empty((n, 2), np.uint8)
        check(self._lib.grt_photon_read(self.handle, _ptr(pos), _ptr(power), _ptr(tp)))
        return pos, power, tp

    def IrradianceEst(self, pos, normal, radius, nphotons):
        """photon_mapping.go:125 for a batch: pos, normal [n, 3] -> irradiance [n, 3]."""
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        nrm = np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
        out = np.empty_like(pos)
        check(self._lib.grt_photon_gather(self.handle, _ptr(pos), _ptr(nrm), pos.shape[0], float(radius),
                                          int(nphotons), _ptr(out)))
        return out


def NewPhotonMap(maxPhotons):  # photon_mapping.go:97
    return PhotonMap(maxPhotons)
