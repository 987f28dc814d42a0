"""Shared helpers for the parity tests: run the same inputs through the oracle (CPU) and through
the product path (Python packages -> C ABI -> HIP kernels) and compare."""
import numpy as np

from gs_localization_amd import scenes as S


def rel_l1(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).sum() / max(np.abs(b).sum(), 1e-30))


def row_errors(a, b, eps=0.01):
    """Per-ROW deviation of a [P, ...] tensor `a` from the reference `b`: |a - b|_1 / (|b|_1 + eps x mean row |b|_1) for every row.
    An aggregate rel_l1 over a million rows hides a few hundred wrong ones; this does not.  Returns (worst, share of rows above
    1e-3, index of the worst row).  (eps x the mean row norm in the denominator: a row whose reference is a thousand times
    smaller than the typical one is held to an absolute, not a relative, error.)"""
    a = np.asarray(a, np.float64).reshape(np.shape(a)[0], -1)
    b = np.asarray(b, np.float64).reshape(a.shape)
    num = np.abs(a - b).sum(1)
    nb = np.abs(b).sum(1)
    den = nb + eps * max(float(nb.mean()), 1e-300)
    r = num / den
    i = int(r.argmax()) if r.size else 0
    return (float(r[i]) if r.size else 0.0), (float((r > 1e-3).mean()) if r.size else 0.0), i


def scene_inputs(sc, w2c=None):
    view, proj, proj_raw, campos = S.camera_matrices(sc, w2c)
    return dict(view=view, proj=proj, proj_raw=proj_raw, campos=campos)


def oracle_run(sc, cam, grads=None, pose=False, colors_precomp=None, cov3D_precomp=None):
    from oracle import oracle as O
    kw = dict(sh_degree=sc.sh_degree, want_n_touched=pose)
    if colors_precomp is None:
        kw["shs"] = sc.shs
    else:
        kw["colors_precomp"] = colors_precomp
    if cov3D_precomp is None:
        kw["scales"], kw["rotations"] = sc.scales, sc.rotations
    else:
        kw["cov3D_precomp"] = cov3D_precomp
    f = O.forward(sc.means3D, sc.opacities, cam["view"], cam["proj"], cam["campos"], sc.W, sc.H, sc.tanfovx, sc.tanfovy,
                  sc.bg, **kw)
    g = None
    if grads is not None:
        g = O.backward(f, grads[0], grads[1], grads[2], pose_mode=pose)
    return f, g


def hip_run(sc, cam, grads=None, pose=False, colors_precomp=None, cov3D_precomp=None, device="cuda:0", debug=False):
    """Product path.  Returns (outputs dict of numpy, grads dict of numpy or None)."""
    import torch
    t = lambda a, rg=True: torch.tensor(np.asarray(a, np.float32), device=device, requires_grad=rg)
    means3D, opac = t(sc.means3D), t(sc.opacities)
    means2D = torch.zeros_like(means3D, requires_grad=True)
    shs = colors = scales = rots = cov = None
    if colors_precomp is None:
        shs = t(sc.shs)
    else:
        colors = t(colors_precomp)
    if cov3D_precomp is None:
        scales, rots = t(sc.scales), t(sc.rotations)
    else:
        cov = t(cov3D_precomp)
    common = dict(image_height=sc.H, image_width=sc.W, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, bg=t(sc.bg, False),
                  scale_modifier=1.0, viewmatrix=t(cam["view"], False), projmatrix=t(cam["proj"], False),
                  sh_degree=sc.sh_degree, campos=t(cam["campos"], False), prefiltered=False, debug=debug)
    if pose:
        import diff_gaussian_rasterization_pose as pkg
        rs = pkg.GaussianRasterizationSettings(projmatrix_raw=t(cam["proj_raw"], False), **common)
        theta = torch.zeros(3, device=device, requires_grad=True)
        rho = torch.zeros(3, device=device, requires_grad=True)
        color, radii, depth, alpha, n_touched = pkg.GaussianRasterizer(rs)(
            means3D=means3D, means2D=means2D, opacities=opac, shs=shs, colors_precomp=colors, scales=scales,
            rotations=rots, cov3D_precomp=cov, theta=theta, rho=rho)
    else:
        import diff_gaussian_rasterization as pkg
        rs = pkg.GaussianRasterizationSettings(**common)
        color, radii, depth, alpha = pkg.GaussianRasterizer(rs)(
            means3D=means3D, means2D=means2D, opacities=opac, shs=shs, colors_precomp=colors, scales=scales,
            rotations=rots, cov3D_precomp=cov)
        n_touched = None
    out = dict(color=color.detach().cpu().numpy(), depth=depth.detach().cpu().numpy(), alpha=alpha.detach().cpu().numpy(),
               radii=radii.cpu().numpy(), n_touched=None if n_touched is None else n_touched.cpu().numpy())
    g = None
    if grads is not None:
        gc, gd, ga = (torch.tensor(x, device=device) for x in grads)
        loss = (color * gc).sum() + (depth * gd).sum() + (alpha * ga).sum()
        loss.backward()
        np_ = lambda x: None if x is None or x.grad is None else x.grad.detach().cpu().numpy()
        g = dict(means3D=np_(means3D), means2D=np_(means2D), opacities=np_(opac), sh=np_(shs), colors_precomp=np_(colors),
                 scales=np_(scales), rotations=np_(rots), cov3Ds_precomp=np_(cov))
        if pose:
            g["tau"] = np.concatenate([rho.grad.cpu().numpy(), theta.grad.cpu().numpy()])
    return out, g


def random_grads(sc, seed=0, with_alpha=True):
    rng = np.random.default_rng(seed)
    gc = rng.normal(size=(3, sc.H, sc.W)).astype(np.float32)
    gd = rng.normal(size=(1, sc.H, sc.W)).astype(np.float32)
    ga = rng.normal(size=(1, sc.H, sc.W)).astype(np.float32) if with_alpha else np.zeros((1, sc.H, sc.W), np.float32)
    return gc, gd, ga
