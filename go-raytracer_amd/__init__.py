"""go-raytracer_amd — MI355X (gfx950) implementation of go-raytracer's per-pixel rendering
hot path behind a C ABI (include/grt.h).  The directory name is not a Python identifier:
import it with importlib.import_module("go-raytracer_amd").

    csrc/      HIP kernels + the C ABI  -> csrc/libgrt_hip.so
    host/      C++ mirror of pkg/tracer + the cmd/raytracer driver (tested native host)
    go/        the cgo shim a maintainer adds to pkg/tracer (not buildable here: no Go)
    tracer.py  ctypes mirror of the pkg/tracer interface for tests and bench.py
    scenes.py  scene literals (cmd/raytracer/main.go:32-59 as data)
    shard.py   tile sharding over ranks (torch.distributed / RCCL)
"""
from . import _capi, scenes, tracer  # noqa: F401
