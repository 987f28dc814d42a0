"""distCUDA2 (simple-knn).  CPU: the oracle against brute force.  GPU (-m gpu): the HIP
implementation against the oracle, through the drop-in `simple_knn._C.distCUDA2` and the C ABI."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import knn_oracle as KO


def _cloud(P, seed, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.uniform(-3, 5, size=(P, 3)).astype(np.float32)
    if kind == "clustered":            # SfM-like: dense clusters + outliers, all coordinates positive (min stays at the origin seed)
        c = rng.uniform(1, 20, size=(8, 3))
        p = c[rng.integers(0, 8, P)] + rng.normal(size=(P, 3)) * rng.uniform(0.01, 1.0, size=(P, 1))
        return np.abs(p).astype(np.float32)
    if kind == "duplicates":
        base = rng.uniform(-1, 1, size=(max(P // 3, 1), 3)).astype(np.float32)
        return base[rng.integers(0, base.shape[0], P)]
    if kind == "plane":
        p = rng.uniform(-2, 2, size=(P, 3)).astype(np.float32); p[:, 2] = 0.5
        return p
    raise ValueError(kind)


@pytest.mark.parametrize("P,kind", [(1, "uniform"), (3, "uniform"), (4, "uniform"), (300, "uniform"), (500, "clustered"), (200, "duplicates")])
def test_oracle_matches_brute_force(P, kind):
    pts = _cloud(P, P, kind)
    a, b = KO.dist2_knn3(pts), KO.dist2_knn3_bruteforce(pts)
    np.testing.assert_array_equal(a, b)
    if P < 3:
        assert np.all(np.isinf(a))                 # FLT_MAX + FLT_MAX overflows, as in the reference
    if P == 3:
        assert np.all(a > 1e38) and np.all(np.isfinite(a))      # (d0 + d1 + FLT_MAX) / 3


@pytest.mark.gpu
@pytest.mark.parametrize("P,kind", [(1, "uniform"), (3, "uniform"), (4, "uniform"), (63, "uniform"), (1025, "uniform"), (5000, "clustered"),
                                    (3000, "duplicates"), (4096, "plane"), (200_000, "clustered")])
def test_distCUDA2_matches_oracle(P, kind):
    import torch
    from simple_knn._C import distCUDA2
    pts = _cloud(P, P + 7, kind)
    got = distCUDA2(torch.tensor(pts, device="cuda:0")).cpu().numpy()
    ref = KO.dist2_knn3(pts)
    assert got.shape == (P,) and got.dtype == np.float32
    if P < 4:
        np.testing.assert_array_equal(got, ref)     # inf for P < 3, (d0 + d1 + FLT_MAX) / 3 for P == 3
        return
    # same neighbours, same float32 distance expression: equal up to the rounding of the three-term sum
    np.testing.assert_allclose(got, ref, rtol=3e-7, atol=0)
    # property at any size: translating the cloud leaves the result unchanged up to the rounding of the differences
    if P >= 1000:
        got2 = distCUDA2(torch.tensor(pts + np.float32(0.5), device="cuda:0")).cpu().numpy()
        np.testing.assert_allclose(got2, got, rtol=2e-3, atol=1e-9)


@pytest.mark.gpu
def test_knn_argument_checks():
    import ctypes as C
    import torch
    from gs_localization_amd import _lib
    lib = _lib.load()
    null_cb = C.cast(None, _lib.RESIZE_FN)
    assert lib.gsr_dist2_knn3(0, None, None, null_cb, None, None) == 0
    assert lib.gsr_dist2_knn3(5, None, None, null_cb, None, None) < 0
    assert lib.gsr_knn_bytes(1000) > 1000 * 32
    with pytest.raises(RuntimeError):
        from simple_knn._C import distCUDA2
        distCUDA2(torch.zeros(4, 3))
