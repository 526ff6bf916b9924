"""The C-ABI library loads and exports exactly what include/grt.h declares (no GPU needed)."""
import os
import re
import subprocess

from conftest import ROOT


def header_functions():
    src = open(os.path.join(ROOT, "include", "grt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(grt_[a-z0-9_]+)\s*\(", src)))


def test_header_table_and_library_agree(pkg):
    declared = header_functions()
    assert len(declared) >= 20
    assert sorted(pkg._capi.PROTOTYPES) == declared
    lib = pkg._capi.load()  # binds every prototype; AttributeError names a missing symbol
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg._capi.LIB_PATH], text=True)
    exported = sorted(set(re.findall(r"\bT (grt_[a-z0-9_]+)$", out, flags=re.M)))
    assert exported == declared, "exports differ from include/grt.h"
    assert lib.grt_version().startswith(b"grt-hip")


def test_record_sizes_match_header(pkg):
    """Compile a tiny C program against grt.h and compare sizeof() with the ctypes mirror."""
    import ctypes as C
    import tempfile

    prog = r'''
#include <stdio.h>
#include "grt.h"
int main(void){printf("%zu %zu %zu %zu %zu\n", sizeof(grt_camera), sizeof(grt_material),
  sizeof(grt_prim), sizeof(grt_scene_desc), sizeof(grt_opts)); return 0;}'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o",
                               os.path.join(d, "s")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "s")], text=True).split()]
    cap = pkg._capi
    assert sizes == [C.sizeof(cap.Camera), C.sizeof(cap.MaterialRec), C.sizeof(cap.PrimRec),
                     C.sizeof(cap.SceneDesc), C.sizeof(cap.Opts)]


def test_scene_create_validates_without_gpu(pkg):
    """Host logic only: grt_scene_create / destroy and argument checks touch no device."""
    import ctypes as C

    lib = pkg._capi.load()
    desc, prims, n_prims, mats, n_mats = pkg.scenes.cornell_flat(32, 32)
    h = C.c_void_p()
    assert lib.grt_scene_create(C.byref(desc), prims, n_prims, mats, n_mats, C.byref(h)) == 0
    assert lib.grt_shard_rows(1080, 0, 8) + lib.grt_shard_rows(1080, 7, 8) == 136 + 128
    assert sum(lib.grt_shard_rows(1081, r, 3) for r in range(3)) == 1081
    o = pkg.tracer.Scene.opts(seed=1)
    assert lib.grt_render(h, C.byref(o), None, 128, 1) == -1           # NULL pixels
    assert b"NULL" in lib.grt_last_error()
    buf = (C.c_uint8 * 16)()
    assert lib.grt_render(h, C.byref(o), buf, 32 * 2, 1) == -1         # bpp < 3
    assert b"bytes per pixel" in lib.grt_last_error()
    lib.grt_scene_destroy(h)
    prims[0].material = 99
    assert lib.grt_scene_create(C.byref(desc), prims, n_prims, mats, n_mats, C.byref(h)) == -1
    assert b"material" in lib.grt_last_error()
    prims[0].material = 0
    prims[0].kind = 7
    assert lib.grt_scene_create(C.byref(desc), prims, n_prims, mats, n_mats, C.byref(h)) == -1


def test_flatten_rejects_unknown_hitable(pkg):
    import pytest

    class List:  # tracer.List (list.go:6) has no flat form
        def Material(self):
            return pkg.tracer.NormalMaterial()

    with pytest.raises(TypeError):
        pkg.tracer.flatten(8, 8, pkg.scenes.cornell_camera(), [List()])
