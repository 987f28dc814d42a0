"""CPU: the oracle's hand-written backward (restating backward.cu) and the pose gradient of the
un-vendored package (B) against float64 autograd (oracle/autograd_ref.py).  Finite differences are
not usable here (hard thresholds), autograd with frozen decisions is."""
import numpy as np
import pytest
import torch

from gs_localization_amd import scenes as S
from oracle import oracle as O, autograd_ref as AG
from tests.util import rel_l1

W2C = S.se3_exp([0.05, -0.03, 0.1, 0.02, -0.04, 0.03])


def _run(sc, w2c, pose, precomp=False, seed=0, tol=1e-5):
    view, proj, proj_raw, campos = S.camera_matrices(sc, w2c)
    kw = dict(sh_degree=sc.sh_degree, want_n_touched=pose)
    f0 = O.forward(sc.means3D, sc.opacities, view, proj, campos, sc.W, sc.H, sc.tanfovx, sc.tanfovy, sc.bg,
                   shs=sc.shs, scales=sc.scales, rotations=sc.rotations, **kw)
    st0 = f0.state()
    if precomp:
        f = O.forward(sc.means3D, sc.opacities, view, proj, campos, sc.W, sc.H, sc.tanfovx, sc.tanfovy, sc.bg,
                      colors_precomp=st0["rgb"], cov3D_precomp=st0["cov3D"], **kw)
    else:
        f = f0
    rng = np.random.default_rng(seed)
    gc = rng.normal(size=(3, sc.H, sc.W)).astype(np.float32)
    gd = rng.normal(size=(1, sc.H, sc.W)).astype(np.float32)
    g = O.backward(f, gc, gd, np.zeros((1, sc.H, sc.W), np.float32), pose_mode=pose)
    t = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    m, o = t(sc.means3D), t(sc.opacities)
    leaves = dict(means3D=m, opacities=o)
    kwargs = {}
    if precomp:
        leaves["colors_precomp"], leaves["cov3Ds_precomp"] = t(st0["rgb"]), t(st0["cov3D"])
        kwargs = dict(colors_precomp=leaves["colors_precomp"], cov3D_precomp=leaves["cov3Ds_precomp"])
    else:
        leaves["sh"], leaves["scales"], leaves["rotations"] = t(sc.shs), t(sc.scales), t(sc.rotations)
        kwargs = dict(shs=leaves["sh"], scales=leaves["scales"], rotations=leaves["rotations"])
    tau = torch.zeros(6, dtype=torch.float64, requires_grad=True) if pose else None
    col, dep, alp, aux = AG.render_autograd(f.state(), f.radii, m, o, torch.tensor(np.asarray(w2c, np.float64)),
                                            torch.tensor(proj_raw.T.astype(np.float64)), sc.W, sc.H, sc.tanfovx,
                                            sc.tanfovy, torch.tensor(sc.bg.astype(np.float64)), sh_degree=sc.sh_degree,
                                            tau=tau, depth_to_mean=pose, **kwargs)
    # forward agreement and no flipped threshold decisions between fp32 and fp64
    assert np.abs(col.detach().numpy() - f.color).max() < 2e-5
    assert np.abs(alp.detach().numpy() - f.alpha[0]).max() < 2e-5
    if pose:
        assert np.array_equal(aux["n_touched"], f.n_touched)
    L = (col * torch.tensor(gc.astype(np.float64))).sum() + (dep * torch.tensor(gd[0].astype(np.float64))).sum()
    L.backward()
    for k, leaf in leaves.items():
        assert rel_l1(g[k], leaf.grad.numpy().reshape(g[k].shape)) < tol, k
    if pose:
        assert rel_l1(g["tau"], tau.grad.numpy()) < tol
    return f


@pytest.mark.parametrize("pose", [False, True])
def test_backward_matches_autograd_sh3(pose):
    sc = S.small(P=300, W=48, H=32, sh_degree=3, seed=3)
    f = _run(sc, W2C, pose)
    assert f.num_rendered > 300


def test_backward_matches_autograd_dense_termination():
    """large opaque splats: exercises alpha cap 0.99, T<1e-4 stop and the n_contrib walk"""
    sc = S.small(P=500, W=32, H=32, sh_degree=1, seed=8, scale_med=0.25)
    sc.opacities[:] = np.clip(sc.opacities * 1.5, 0, 0.999)
    # the reference rebuilds T by repeated fp32 division by (1 - alpha) with alpha up to 0.99
    # (backward.cu:516): ~1e-5 relative error per division is inherent to the algorithm in fp32
    f = _run(sc, W2C, True, tol=1e-3)
    nc = f.state()["n_contrib"]
    ranges = f.state()["ranges"]
    assert (nc.max() < (ranges[:, 1] - ranges[:, 0]).max())      # some pixel stopped before the end of its list


def test_pose_gradient_offcentre_principal_point_white_bg():
    sc = S.small(P=200, W=40, H=24, sh_degree=1, seed=5)
    sc.bg[:] = 1.0
    sc.cx, sc.cy = 17.0, 14.5
    _run(sc, W2C, True)


def test_precomputed_mode_backward():
    sc = S.small(P=250, W=48, H=32, sh_degree=2, seed=12)
    _run(sc, W2C, False, precomp=True)
    _run(sc, W2C, True, precomp=True)


def test_backward_is_linear_in_upstream_gradients():
    sc = S.small(P=300, W=48, H=32, sh_degree=2, seed=4)
    view, proj, _, campos = S.camera_matrices(sc, W2C)
    f = O.forward(sc.means3D, sc.opacities, view, proj, campos, sc.W, sc.H, sc.tanfovx, sc.tanfovy, sc.bg,
                  sh_degree=2, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    rng = np.random.default_rng(0)
    mk = lambda: [rng.normal(size=s).astype(np.float32) for s in ((3, sc.H, sc.W), (1, sc.H, sc.W), (1, sc.H, sc.W))]
    a, b = mk(), mk()
    ga, gb = O.backward(f, *a, pose_mode=True), O.backward(f, *b, pose_mode=True)
    gs = O.backward(f, *[x + y for x, y in zip(a, b)], pose_mode=True)
    for k in gs:
        assert rel_l1(gs[k], ga[k] + gb[k]) < 1e-5, k
