"""ctypes mirror of include/grt.h (POD records + prototypes of libgrt_hip.so).

The records are the drop-in boundary's: grt_camera = camera.go:10-13, grt_material =
material.go:4-13, grt_prim = sphere.go:10-14 / aabb.go:10-15, grt_scene_desc =
scene.go:14,17-27 (reference paths relative to the upstream repo).

Loading fails loudly when the HIP library is missing: there is no CPU fallback in the
product path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgrt_hip.so")



def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources of libgrt_hip.so: names the build a profile was taken on
    (bench.py prints it; profiles/traffic.json records it, and bench.py drops carried-over counters that belong to
    another build)."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(_HERE, "csrc", "*.h")) + glob.glob(os.path.join(_HERE, "csrc", "*.hip"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


GRT_KIND_SPHERE = 0
GRT_KIND_AABB = 1
GRT_JITTER_CENTER = 0
GRT_JITTER_RNG = 1
GRT_BAND_ROWS = 8
GRT_ACCEL_AUTO = 0
GRT_ACCEL_BVH = 1
GRT_ACCEL_LINEAR = 2
GRT_PMVIEW_IRRADIANCE = 0
GRT_PMVIEW_FIRST_HIT = 1


class Camera(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("lower_left", C.c_double * 3),
                ("horizontal", C.c_double * 3), ("vertical", C.c_double * 3)]


class MaterialRec(C.Structure):
    _fields_ = [("color", C.c_double * 3), ("refr_index", C.c_double),
                ("reflectivity", C.c_double), ("roughness", C.c_double),
                ("emittance", C.c_double), ("transparent", C.c_uint8), ("lambert", C.c_uint8),
                ("normal", C.c_uint8), ("_pad", C.c_uint8 * 5)]


class PrimRec(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("a", C.c_double * 3),
                ("b", C.c_double * 3), ("radius", C.c_double)]


class SceneDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_depth", C.c_int32),
                ("accel", C.c_int32), ("bias", C.c_double), ("cam", Camera)]


class Opts(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_device", C.c_int32), ("n_devices", C.c_int32),
                ("chunk_spp", C.c_int32), ("flags", C.c_int32)]


assert C.sizeof(Camera) == 96 and C.sizeof(MaterialRec) == 64
assert C.sizeof(PrimRec) == 64 and C.sizeof(SceneDesc) == 120 and C.sizeof(Opts) == 24

P = C.POINTER
_vp = C.c_void_p

# name -> (restype, argtypes).  Every symbol include/grt.h declares is listed here;
# tests/test_capi_symbols.py checks the header, this table and the built library agree.
PROTOTYPES = {
    "grt_scene_create": (C.c_int, [P(SceneDesc), P(PrimRec), C.c_int32, P(MaterialRec), C.c_int32,
                                   P(_vp)]),
    "grt_scene_destroy": (None, [_vp]),
    "grt_render": (C.c_int, [_vp, P(Opts), _vp, C.c_int32, C.c_int32]),
    "grt_render_accum": (C.c_int, [_vp, P(Opts), _vp, C.c_int32]),
    "grt_render_accum_dev": (C.c_int, [_vp, P(Opts), C.c_int32, C.c_int32, C.c_int32, _vp, _vp,
                                       C.c_uint64, _vp, _vp]),
    "grt_render_accum_samples_dev": (C.c_int, [_vp, P(Opts), C.c_int32, C.c_int32, _vp, _vp, C.c_uint64, _vp, _vp]),
    "grt_render_accum_noalloc_dev": (C.c_int, [_vp, P(Opts), C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "grt_shard_rows": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "grt_accum_workspace_bytes": (C.c_uint64, [_vp, P(Opts), C.c_int32, C.c_int32, C.c_int32]),
    "grt_tonemap_dev": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_int32, _vp]),
    "grt_trace_samples": (C.c_int, [_vp, P(Opts), _vp, C.c_int32, C.c_int32, _vp, _vp]),
    "grt_primary_hits": (C.c_int, [_vp, P(Opts), C.c_int32, _vp, _vp, _vp, P(C.c_uint64)]),
    "grt_primary_sweep_dev": (C.c_int, [_vp, P(Opts), C.c_int32, _vp, _vp]),
    "grt_photon_build": (C.c_int, [_vp, P(Opts), C.c_int32, C.c_int32, P(_vp)]),
    "grt_photon_count": (C.c_int32, [_vp]),
    "grt_photon_device": (C.c_int32, [_vp]),
    "grt_photon_read": (C.c_int, [_vp, _vp, _vp, _vp]),
    "grt_photon_load": (C.c_int, [P(Opts), _vp, _vp, _vp, C.c_int32, P(_vp)]),
    "grt_photon_emit_dev": (C.c_int, [_vp, P(Opts), C.c_int32, C.c_int32, C.c_uint64, C.c_int32, _vp, _vp, _vp, _vp, C.c_uint64, _vp]),
    "grt_photon_emit_workspace_bytes": (C.c_uint64, [C.c_int32]),
    "grt_photon_load_dev": (C.c_int, [P(Opts), _vp, _vp, C.c_int32, C.c_float, C.c_uint64, P(_vp)]),
    "grt_scene_light_count": (C.c_int32, [_vp]),
    "grt_scene_light_share": (C.c_int32, [_vp, C.c_int32]),
    "grt_photon_gather": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_float, C.c_int32, _vp]),
    "grt_photon_gather_dev": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_float, C.c_int32, _vp, _vp, _vp]),
    "grt_photon_gather_workspace_bytes": (C.c_uint64, [C.c_int32]),
    "grt_photon_emitted": (C.c_uint64, [_vp]),
    "grt_photon_free": (None, [_vp]),
    "grt_render_pm": (C.c_int, [_vp, P(Opts), _vp, _vp, _vp, C.c_int32, C.c_int32]),
    "grt_render_accum_pm": (C.c_int, [_vp, P(Opts), _vp, _vp, _vp, C.c_int32]),
    "grt_render_accum_pm_dev": (C.c_int, [_vp, P(Opts), _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, C.c_uint64, _vp, _vp]),
    "grt_render_pm_workspace_bytes": (C.c_uint64, [_vp, P(Opts), C.c_int32, C.c_int32, C.c_int32]),
    "grt_render_pmview": (C.c_int, [_vp, P(Opts), _vp, C.c_int32, _vp, C.c_int32, C.c_int32]),
    "grt_render_accum_pmview": (C.c_int, [_vp, P(Opts), _vp, C.c_int32, _vp, C.c_int32]),
    "grt_rng_u01": (C.c_double, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]),
    "grt_last_ray_count": (C.c_uint64, []),
    "grt_last_kernel_ms": (C.c_double, []),
    "grt_device_count": (C.c_int, []),
    "grt_last_error": (C.c_char_p, []),
    "grt_version": (C.c_char_p, []),
}


def bind(lib, prototypes, prefix_from="grt_", prefix_to="grt_"):
    """Attach restype/argtypes; raises AttributeError naming any missing symbol."""
    for name, (res, args) in prototypes.items():
        sym = prefix_to + name[len(prefix_from):]
        fn = getattr(lib, sym)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def load():
    """Load libgrt_hip.so (built by __graft_entry__.build()).  No fallback."""
    global _lib
    if _lib is None:
        # A process must hold ONE HIP runtime.  torch bundles its own libamdhip64 / libhsa-runtime64;
        # if /opt/rocm's copy is loaded first (through libgrt_hip.so) and torch's afterwards, the second
        # HSA runtime finds no GPU ("No HIP GPUs are available").  Loaded after torch, libgrt_hip.so
        # binds to the runtime already in the process (same SONAME).  shard.py and bench.py use torch
        # for device memory and process groups, so it goes first whenever it is installed.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _lib = bind(C.CDLL(LIB_PATH), PROTOTYPES)
    return _lib


class GrtError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = load().grt_last_error()
        raise GrtError(f"grt error {rc}: {msg.decode() if msg else ''}")
