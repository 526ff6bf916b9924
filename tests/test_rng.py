"""The counter RNG is stated twice (oracle C, product header): both must name the same uniforms."""
import numpy as np


def test_rng_streams_agree(pkg, ob):
    lib, olib = pkg._capi.load(), ob.load()
    rs = np.random.RandomState(0)
    for _ in range(2000):
        seed = int(rs.randint(0, 2**63 - 1))
        pix, smp, dim = int(rs.randint(0, 2**31)), int(rs.randint(0, 100000)), int(rs.randint(0, 32))
        a, b = lib.grt_rng_u01(seed, pix, smp, dim), olib.grto_rng_u01(seed, pix, smp, dim)
        assert a == b and 0.0 <= a < 1.0
        assert (a * 2**24) == int(a * 2**24)  # 24 bits: exact in float32 and float64


def test_rng_is_uniform_and_decorrelated(pkg):
    lib = pkg._capi.load()
    u = np.array([[lib.grt_rng_u01(1, p, s, d) for d in range(4)] for p in range(40) for s in range(50)])
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    c = np.corrcoef(u.T)
    assert np.abs(c - np.eye(4)).max() < 0.08
    # neighbouring pixels / samples do not repeat each other's draws
    assert len(np.unique(u[:, 0])) > 0.99 * len(u)
