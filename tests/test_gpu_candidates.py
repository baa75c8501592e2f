"""GPU loop-closure candidate search (pgo_generate_candidates) against the CPU generator of
posegraph-ceres_amd/datasets.py (itself replayed against the reference's Edge_Candidates_index.txt, test_datasets.py).
Index data: bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_kitti00_candidates_match_the_reference_file(pkg, ds):
    """The GPU search against the ids of the reference's own config/Edge_Candidates_index.txt (stored in kitti00.npz
    as keys / offsets / flat ids + the sha256 of the file text): bit-exact, and the formatted text hashes to the file."""
    import hashlib
    k = np.load(os.path.join(GOLD, "kitti00.npz"))
    xyz = k["origin"][:, :3]
    got = pkg.generate_candidates(xyz, 6.0, 100)
    assert sorted(kk for kk in got if got[kk]) == [int(x) for x in k["cand_keys"]]
    for i, kk in enumerate(k["cand_keys"]):
        assert got[int(kk)] == k["cand_flat"][k["cand_offsets"][i]:k["cand_offsets"][i + 1]].tolist(), "line %d" % kk
    assert hashlib.sha256(ds.format_candidates(got).encode()).hexdigest() == str(k["cand_sha256"])
    assert got == ds.generate_candidates(xyz, 6.0, 100)       # and the CPU generator agrees
    assert sum(len(v) - 1 for v in got.values()) > 10000     # KITTI 00 revisits: the lists are not trivial


@pytest.mark.parametrize("n,radius,gap,seed", [(1, 6.0, 100, 0), (2, 6.0, 100, 1), (101, 6.0, 100, 2), (102, 6.0, 100, 3),
                                               (700, 2.5, 7, 4), (3000, 1.0, 0, 5), (5000, 3.0, 100, 6)])
def test_random_walk_candidates(pkg, ds, n, radius, gap, seed):
    rng = np.random.default_rng(seed)
    xyz = np.cumsum(rng.normal(scale=0.6, size=(n, 3)), axis=0)
    # plant exact ties on the threshold: integer lattice points at distance == radius are representable for radius 1 / 3
    if n >= 3000:
        xyz = np.round(xyz)
    ref = ds.generate_candidates(xyz, radius, gap)
    got = pkg.generate_candidates(xyz, radius, gap)
    assert got == ref


def test_empty_and_capacity_errors(pkg):
    assert pkg.generate_candidates(np.zeros((0, 3))) == {}
    import ctypes as C
    p = np.zeros((300, 3), dtype=np.float32)
    rp = np.zeros(301, dtype=np.int64)
    idx = np.zeros(4, dtype=np.int32)
    rc = pkg.lib().pgo_generate_candidates(p.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(300), C.c_float(6.0), C.c_int(100),
                                           rp.ctypes.data_as(C.POINTER(C.c_longlong)), idx.ctypes.data_as(C.POINTER(C.c_int)),
                                           C.c_longlong(4), None)
    assert rc == pkg.ERR_INVALID_ARGUMENT


def test_candidates_property(pkg, ds):
    """hypothesis: random small clouds on a coarse lattice (many exact ties with the threshold), random radius / gap."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=40, deadline=None)
    @given(st.integers(0, 400), st.integers(0, 2 ** 31 - 1), st.sampled_from([1.0, 2.0, 3.0, 5.0]), st.integers(0, 120))
    def check(n, seed, radius, gap):
        rng = np.random.default_rng(seed)
        xyz = rng.integers(-4, 5, size=(n, 3)).astype(np.float64)
        assert pkg.generate_candidates(xyz, radius, gap) == ds.generate_candidates(xyz, radius, gap)

    check()
