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
        assert (a * 2**16) == int(a * 2**16)  # 16 bits per uniform (one hash per pair): exact in float32 and float64


def test_rng_is_uniform_and_decorrelated(pkg):
    lib = pkg._capi.load()
    u = np.array([[lib.grt_rng_u01(1, p, s, d) for d in range(4)] for p in range(40) for s in range(50)])
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    c = np.corrcoef(u.T)
    assert np.abs(c - np.eye(4)).max() < 0.08
    # neighbouring pixels / samples do not repeat each other's draws: a pair of uniforms is one 32-bit hash (a single
    # 16-bit uniform repeats among 2000 draws by the birthday bound: 2000^2 / 2^17 = 30 expected)
    assert len(np.unique(u[:, :2], axis=0)) == len(u) and len(np.unique(u[:, 2:], axis=0)) == len(u)
    assert len(np.unique(u[:, 0])) > 0.975 * len(u)
    # the two halves of a hash are as good as two hashes: joint uniformity of (u0, u1) on a 8 x 8 grid
    cells = np.histogram2d(u[:, 0], u[:, 1], bins=8, range=[[0, 1], [0, 1]])[0]
    assert cells.min() > 10 and cells.max() < 60          # 31 expected per cell


def _mix32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def test_path_keys_do_not_collide():
    """The two-word path key (csrc/grt_rng.h): pairs (pixel, sample) whose first words collide — with one 32-bit
    word a third of a 1080p x 1000 spp frame's paths did — differ in the second word.  4 M random pairs (pixel keys are
    a bijection of the pixel index, so random 32-bit values stand for them): ~2000 first-word collisions expected."""
    rs = np.random.RandomState(11)
    n = 1 << 22
    pk = rs.randint(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    smp = rs.randint(0, 1000, n).astype(np.uint32)
    with np.errstate(over="ignore"):
        a = _mix32(pk + smp * np.uint32(0x85EBCA77) + np.uint32(0xC2B2AE3D))
        b = pk + smp * np.uint32(0x27D4EB2E) + np.uint32(0x165667B1)     # the second word is not hashed
    order = np.argsort(a, kind="stable")
    a_s, b_s, pk_s, s_s = a[order], b[order], pk[order], smp[order]
    same_a = a_s[1:] == a_s[:-1]
    distinct_pair = (pk_s[1:] != pk_s[:-1]) | (s_s[1:] != s_s[:-1])
    coll = same_a & distinct_pair
    assert coll.sum() > 500                                  # the one-word key does collide at this scale
    assert not (coll & (b_s[1:] == b_s[:-1])).any()          # the second word tells every such pair apart


def test_draws_of_colliding_first_words_are_unrelated(ob):
    """Two paths with the same first key word and different second words draw unrelated uniforms."""
    lib = ob.load()
    # pixel keys are not reachable through the API; use two samples of one pixel and many dims instead:
    u = np.array([[lib.grto_rng_u01(5, 1234, s, d) for d in range(64)] for s in range(64)])
    assert abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.01
    c = np.corrcoef(u)                                       # sample streams against each other
    assert np.abs(c - np.eye(64)).max() < 0.6
