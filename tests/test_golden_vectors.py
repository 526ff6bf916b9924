"""Committed oracle vectors (tests/golden/oracle_*.npz, made by tests/golden/make_golden.py): the oracle must
still produce them (CPU), and the HIP path must match them to the parity bars without the oracle in the loop (GPU)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

W, H = 96, 54


def load(variant):
    return np.load(os.path.join(GOLDEN, f"oracle_{variant}.npz"))


@pytest.mark.parametrize("variant", ["cornell8", "cornell8d"])
def test_oracle_reproduces_its_vectors(pkg, ob, variant):
    g = load(variant)
    s = ob.OracleScene(pkg.scenes.cornell_flat(W, H, variant))
    for jitter in (0, 1):
        ids, t, n = s.PrimaryHits(jitter, 3)
        assert (ids == g[f"ids_j{jitter}"]).all() and (t == g[f"t_j{jitter}"]).all() and (n == g[f"n_j{jitter}"]).all()
    rgb, nr = s.TraceSamples(g["xy"], 6, seed=9)
    assert (nr == g["sample_rays"]).all() and np.allclose(rgb, g["sample_rgb"], rtol=1e-12, atol=0)
    assert np.allclose(s.RenderAccum(8, seed=4), g["sum_8spp"], rtol=1e-12, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["cornell8", "cornell8d"])
def test_hip_matches_the_committed_vectors(pkg, variant):
    """ids bit-exact, t and normals 1e-5 (north star); per-sample radiance with the shared RNG; image sums."""
    g = load(variant)
    scene = pkg.scenes.cornell(W, H, variant)
    for jitter in (0, 1):
        ids, t, n, _ = scene.PrimaryHits(jitter, 3)
        rid, rt, rn = g[f"ids_j{jitter}"], g[f"t_j{jitter}"], g[f"n_j{jitter}"]
        assert (ids == rid).all()
        hit = rid >= 0
        assert (np.abs(t[hit] - rt[hit]) / rt[hit]).max() < 1e-5 and np.abs(n - rn).max() < 1e-5
    rgb, nr = scene.TraceSamples(g["xy"], 6, seed=9)
    same = (nr == g["sample_rays"]) & np.isclose(rgb, g["sample_rgb"], rtol=2e-3, atol=1e-5).all(-1)
    assert same.mean() > 0.999, same.mean()
    a = scene.RenderAccum(8, seed=4).astype(np.float64)
    rel = np.abs(a.mean((0, 1)) - g["sum_8spp"].mean((0, 1))) / g["sum_8spp"].mean((0, 1))
    assert rel.max() < 5e-3
