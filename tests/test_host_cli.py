"""The native host (C++ mirror of pkg/tracer + the cmd/raytracer driver) over the C ABI."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

from conftest import ROOT

CLI = os.path.join(ROOT, "go-raytracer_amd", "host", "grt_cli")


def test_cli_usage_mirrors_reference_flags():
    out = subprocess.run([CLI, "-h"], capture_output=True, text=True)
    assert out.returncode == 0
    for flag in ("-w int", "-s int", "-p int", "-o string"):  # main.go:21-24
        assert flag in out.stderr
    assert subprocess.run([CLI, "-bogus"], capture_output=True).returncode == 2


def test_cli_fails_loudly_without_gpu(pkg):
    if pkg._capi.load().grt_device_count() > 0:
        pytest.skip("a GPU is visible")
    out = subprocess.run([CLI, "-w", "16", "-s", "1"], capture_output=True, text=True)
    assert out.returncode == 1 and "no HIP device" in out.stderr  # no CPU fallback


@pytest.mark.gpu
def test_cli_png_equals_library_render(pkg, tmp_path):
    """grt_cli -w 96 -s 8 -seed 5 -o x.png writes the same bytes Scene.Render produces
    through ctypes (same library, same seed), as an RGBA PNG with A = 255 (image.go:19-24)."""
    out = subprocess.run([CLI, "-w", "96", "-s", "8", "-seed", "5", "-o", "cli.png"], cwd=tmp_path,
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "Started rendering (8 samples)" in out.stderr and "Rendering took" in out.stderr
    img = np.asarray(Image.open(tmp_path / "output" / "cli.png"))
    assert img.shape == (96, 96, 4) and (img[..., 3] == 255).all()
    scene = pkg.scenes.cornell(96, 96)
    px = np.zeros(96 * 96 * 4, np.uint8)
    scene.Render(px, 4 * 96, 8, seed=5)
    bgr = px.reshape(96, 96, 4)
    assert (img[..., 0] == bgr[..., 2]).all() and (img[..., 1] == bgr[..., 1]).all() and (img[..., 2] == bgr[..., 0]).all()
