"""training-step loss epilogue (L1 + SSIM + pseudo-depth Pearson) and densification
statistics.  CPU: the oracle against vectors produced by the reference's own loss_utils.py.  GPU (-m gpu): the HIP
kernels against the oracle and the golden vectors, through the C ABI."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import train_loss_oracle as TO

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "train_loss_vectors.npz"))


def rel_l1(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).sum() / max(np.abs(b).sum(), 1e-300)


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_oracle_matches_reference_loss_utils(name):
    import torch
    img, gt, lam = GOLD[f"{name}_img"], GOLD[f"{name}_gt"], float(GOLD[f"{name}_lambda"])
    np.testing.assert_array_equal(TO.window_1d().numpy(), GOLD["window_1d"])
    for dtype, tol in ((torch.float32, 2e-6), (torch.float64, 2e-5)):      # fp64 differs from the fp32 reference by fp32 rounding
        r = TO.training_loss(img, gt, lam, dtype=dtype)
        assert abs(r["Ll1"] - float(GOLD[f"{name}_Ll1"])) <= tol * abs(float(GOLD[f"{name}_Ll1"]))
        assert abs(r["ssim"] - float(GOLD[f"{name}_ssim"])) <= tol
        assert abs(r["loss"] - float(GOLD[f"{name}_loss"])) <= tol
        assert rel_l1(r["dL_dimage"], GOLD[f"{name}_grad"]) <= (1e-5 if dtype == torch.float32 else 2e-4)


def test_oracle_pearson_properties():
    rng = np.random.default_rng(0)
    d = rng.uniform(0.5, 5.0, size=(20, 30)).astype(np.float32)
    img = rng.uniform(size=(3, 20, 30)).astype(np.float32)
    # pseudo depth perfectly anti-correlated with depth: the first candidate is 1 - rho(-m, d) = 1 - rho(d - 5, d) = 0
    r = TO.training_loss(img, img, 0.2, depth=d, pseudo_depth=(5.0 - d), depth_weight=0.1)
    assert abs(r["pseudo"]) < 1e-9 and abs(r["loss"]) < 1e-9
    # invariance of rho under affine maps of the prediction, and the gradient sums to zero (rho is shift invariant)
    m = rng.uniform(1.0, 50.0, size=(20, 30)).astype(np.float32)
    r1 = TO.training_loss(img, img, 0.2, depth=d, pseudo_depth=m)
    r2 = TO.training_loss(img, img, 0.2, depth=3.0 * d + 1.0, pseudo_depth=m)
    assert abs(r1["pseudo"] - r2["pseudo"]) < 1e-9
    assert abs(r1["dL_ddepth"].sum()) < 1e-12


def test_oracle_densification_stats():
    rng = np.random.default_rng(1)
    P = 50
    radii = rng.integers(-1, 30, size=P).astype(np.int32)
    g = rng.normal(size=(P, 3)).astype(np.float32)
    mr, acc, den = rng.uniform(0, 20, P).astype(np.float32), rng.uniform(0, 1, P).astype(np.float32), rng.integers(0, 5, P).astype(np.float32)
    r, a, d = TO.densification_stats(radii, g, mr, acc, den)
    vis = radii > 0
    assert np.array_equal(r[~vis], mr[~vis]) and np.array_equal(a[~vis], acc[~vis]) and np.array_equal(d[~vis], den[~vis])
    assert np.all(r[vis] >= radii[vis]) and np.all(d[vis] == den[vis] + 1)
    np.testing.assert_allclose(a[vis] - acc[vis], np.hypot(g[vis, 0], g[vis, 1]), rtol=1e-5)


# ------------------------------------------------------------------------------------------------ GPU
def _gpu_loss(img, gt, lam, depth=None, pseudo=None, w=0.1):
    import torch
    from gs_localization_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    _, H, W = img.shape
    t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
    x, g = t(img), t(gt)
    d = t(depth) if depth is not None else None
    m = t(pseudo) if pseudo is not None else None
    gi = torch.full((3, H, W), float("nan"), device=dev)
    gd = torch.full((H, W), float("nan"), device=dev) if d is not None else None
    out = torch.zeros(4, device=dev)
    keep = {}

    def resize(ctx, n):
        keep["t"] = torch.empty(n, dtype=torch.uint8, device=dev)
        return keep["t"].data_ptr()
    cb = _lib.RESIZE_FN(resize)
    p = lambda q: C.c_void_p(q.data_ptr()) if q is not None else None
    assert lib.gsr_training_loss_bytes(W, H) == 36 * W * H + 256
    _lib.check(lib.gsr_training_loss(W, H, p(x), p(g), lam, p(d), p(m), w, p(gi), p(gd), p(out), cb, None, None))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    return dict(loss=float(o[0]), Ll1=float(o[1]), ssim=float(o[2]), pseudo=float(o[3]), dL_dimage=gi.cpu().numpy(),
                dL_ddepth=gd.cpu().numpy() if gd is not None else None)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_kernels_match_reference_vectors_and_oracle(name):
    img, gt, lam = GOLD[f"{name}_img"], GOLD[f"{name}_gt"], float(GOLD[f"{name}_lambda"])
    r = _gpu_loss(img, gt, lam)
    assert abs(r["Ll1"] - float(GOLD[f"{name}_Ll1"])) <= 2e-6 * abs(float(GOLD[f"{name}_Ll1"]))
    assert abs(r["ssim"] - float(GOLD[f"{name}_ssim"])) <= 2e-6
    assert abs(r["loss"] - float(GOLD[f"{name}_loss"])) <= 2e-6
    assert r["pseudo"] == 0.0
    o = TO.training_loss(img, gt, lam)                       # float64
    # both fp32 evaluations (the reference's and ours) sit within fp32 rounding of the float64 gradient
    assert rel_l1(r["dL_dimage"], o["dL_dimage"]) <= 2e-4
    assert rel_l1(r["dL_dimage"], GOLD[f"{name}_grad"]) <= 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(16, 16), (23, 37), (480, 640)])
def test_kernels_with_depth_term(H, W):
    rng = np.random.default_rng(H * W)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    gt = np.clip(0.5 + 0.4 * np.sin(xx / 9.0)[None] * np.cos(yy / 6.0)[None] + 0.05 * rng.normal(size=(3, H, W)), 0, 1).astype(np.float32)
    img = np.clip(gt + 0.08 * rng.normal(size=(3, H, W)), 0, 1).astype(np.float32)
    depth = (2.0 + np.sin(xx / 20.0) + 0.1 * rng.normal(size=(H, W))).astype(np.float32)
    pseudo = (100.0 / depth + rng.normal(size=(H, W))).astype(np.float32)        # a disparity-like prediction
    r = _gpu_loss(img, gt, 0.2, depth, pseudo, 0.1)
    o = TO.training_loss(img, gt, 0.2, depth, pseudo, 0.1)
    assert abs(r["loss"] - o["loss"]) <= 5e-6 and abs(r["pseudo"] - o["pseudo"]) <= 5e-6
    assert abs(r["ssim"] - o["ssim"]) <= 5e-6 and abs(r["Ll1"] - o["Ll1"]) <= 1e-6
    assert rel_l1(r["dL_dimage"], o["dL_dimage"]) <= 2e-4
    assert rel_l1(r["dL_ddepth"], o["dL_ddepth"]) <= 1e-4
    # size-independent properties: the Pearson gradient is orthogonal to constants; identical images give ssim 1, L1 0
    assert abs(float(r["dL_ddepth"].astype(np.float64).sum())) <= 1e-6 * float(np.abs(r["dL_ddepth"]).sum())
    s = _gpu_loss(gt, gt, 0.2)
    assert s["Ll1"] == 0.0 and abs(s["ssim"] - 1.0) <= 1e-6 and abs(s["loss"]) <= 1e-6


@pytest.mark.gpu
def test_densification_stats_kernel():
    import torch
    from gs_localization_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    P = 100_003
    radii = rng.integers(-1, 40, size=P).astype(np.int32)
    g = rng.normal(size=(P, 3)).astype(np.float32)
    mr, acc, den = rng.uniform(0, 20, P).astype(np.float32), rng.uniform(0, 1, P).astype(np.float32), rng.integers(0, 5, P).astype(np.float32)
    t = lambda a: torch.tensor(a, device=dev)
    tr, tg, tm, ta, td = t(radii), t(g), t(mr), t(acc), t(den)
    p = lambda q: C.c_void_p(q.data_ptr())
    _lib.check(lib.gsr_densification_stats(P, p(tr), p(tg), p(tm), p(ta), p(td), None))
    torch.cuda.synchronize()
    r, a, d = TO.densification_stats(radii, g, mr, acc, den)
    np.testing.assert_array_equal(tm.cpu().numpy(), r)
    np.testing.assert_array_equal(td.cpu().numpy(), d)
    np.testing.assert_allclose(ta.cpu().numpy(), a, rtol=3e-7)
    assert lib.gsr_densification_stats(0, None, None, None, None, None, None) == 0


@pytest.mark.gpu
def test_python_wrapper_autograd_and_densification_wrapper():
    import torch
    from gs_localization_amd import train_epilogue as TE
    dev = torch.device("cuda:0")
    name = "a"
    img, gt, lam = GOLD[f"{name}_img"], GOLD[f"{name}_gt"], float(GOLD[f"{name}_lambda"])
    x = torch.tensor(img, device=dev, requires_grad=True)
    rng = np.random.default_rng(5)
    d = torch.tensor(rng.uniform(1, 4, size=(1,) + img.shape[1:]).astype(np.float32), device=dev, requires_grad=True)
    m = torch.tensor(rng.uniform(10, 90, size=img.shape[1:]).astype(np.float32), device=dev)
    loss = TE.training_loss(x, torch.tensor(gt, device=dev), lam, d, m, 0.1)
    (2.0 * loss).backward()                                           # upstream gradient is honoured
    o = TO.training_loss(img, gt, lam, d.detach().cpu().numpy()[0], m.cpu().numpy(), 0.1)
    assert abs(float(loss.detach()) - o["loss"]) <= 5e-6
    assert rel_l1(x.grad.cpu().numpy(), 2.0 * o["dL_dimage"]) <= 2e-4
    assert d.grad.shape == d.shape and rel_l1(d.grad.cpu().numpy()[0], 2.0 * o["dL_ddepth"]) <= 1e-4
    with pytest.raises(RuntimeError):
        TE.training_loss(torch.zeros(3, 4, 4), torch.zeros(3, 4, 4))
    P = 1000
    radii = torch.tensor(rng.integers(-1, 9, size=P).astype(np.int32), device=dev)
    g = torch.tensor(rng.normal(size=(P, 3)).astype(np.float32), device=dev)
    mr, acc, den = torch.zeros(P, device=dev), torch.zeros(P, 1, device=dev), torch.zeros(P, 1, device=dev)
    TE.add_densification_stats(radii, g, mr, acc, den)
    torch.cuda.synchronize()
    vis = (radii > 0).cpu().numpy()
    assert np.array_equal(den.cpu().numpy()[:, 0], vis.astype(np.float32))
    np.testing.assert_allclose(acc.cpu().numpy()[vis, 0], np.hypot(*g.cpu().numpy()[vis, :2].T), rtol=3e-7)
