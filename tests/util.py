"""Shared helpers for the parity tests: run the same inputs through the oracle (CPU) and through
the product path (Python packages -> C ABI -> HIP kernels) and compare."""
import numpy as np

from gs_localization_amd import scenes as S


def rel_l1(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).sum() / max(np.abs(b).sum(), 1e-30))


def row_error_vector(a, b, eps=0.01):
    """Per-ROW deviation of a [P, ...] tensor `a` from the reference `b`: |a - b|_1 / (|b|_1 + eps x mean row |b|_1) for every row.
    (eps x the mean row norm in the denominator: a row whose reference is a thousand times smaller than the typical one is held to an
    absolute, not a relative, error.)"""
    a = np.asarray(a, np.float64).reshape(np.shape(a)[0], -1)
    b = np.asarray(b, np.float64).reshape(a.shape)
    nb = np.abs(b).sum(1)
    return np.abs(a - b).sum(1) / (nb + eps * max(float(nb.mean()) if nb.size else 0.0, 1e-300))


def row_errors(a, b, eps=0.01, rows=None):
    """An aggregate rel_l1 over a million rows hides a few hundred wrong ones; this does not.  Returns (worst, share of rows above
    1e-3, index of the worst row) over all rows, or over the boolean selection `rows`."""
    r = row_error_vector(a, b, eps)
    idx = np.arange(r.size)
    if rows is not None:
        r, idx = r[rows], idx[rows]
    if not r.size:
        return 0.0, 0.0, 0
    i = int(r.argmax())
    return float(r[i]), float((r > 1e-3).mean()), int(idx[i])


def rows_with_a_cause(a, refs, unstable, worst_bar, share_bar, eps=0.01):
    """Flip accounting: every row of `a` that misses the per-row bars must have a CAUSE -- a pixel in which one of
    the forward's threshold tests sits within rounding of its threshold (oracle.flip_audit's `unstable`).  The rows WITHOUT such a
    pixel are held to the standard bars.  `refs`: the evaluations of the reference's algorithm a row may agree with -- its fp32
    restatement and the same decisions evaluated in double (a row that is a difference of nearly equal colours is rounding noise in
    BOTH fp32 paths; the double evaluation says which of them is off).  Returns a dict for the report; `ok` False = the bars are missed."""
    r = None
    for b in refs:
        rb = row_error_vector(a, b, eps)
        r = rb if r is None else np.minimum(r, rb)
    stable = ~np.asarray(unstable, bool)
    rs = r[stable]
    worst = float(rs.max()) if rs.size else 0.0
    share = float((rs > 1e-3).mean()) if rs.size else 0.0
    idx = np.flatnonzero(stable)
    at = int(idx[int(rs.argmax())]) if rs.size else 0
    return dict(ok=worst <= worst_bar and share <= share_bar, worst=worst, share=share, at=at,
                worst_all=float(r.max()) if r.size else 0.0, share_all=float((r > 1e-3).mean()) if r.size else 0.0,
                excused=int(((r > 1e-3) & ~stable).sum()), unexplained=np.flatnonzero((r > 1e-3) & stable), r=r)


ROW_WORST, ROW_SHARE = 0.1, 2e-4      # the standard per-row bars: worst row, share of rows above 1e-3 (round 5: measured <= 0.035 / 2.5e-5 unsplit)


def flip_accounted_parity(f, gi, gd, grads, tau, n_touched, worst_bar=ROW_WORST, share_bar=ROW_SHARE):
    """The Gaussian-parameter gradients, dL/dtau and n_touched of a HIP backward against the oracle's on the forward `f`, with ONE
    set of bars for every scene (split tiles or not) and a cause demanded for every row and every count that misses them.
      grads: {"m3d", "sh", "opac", "scale", "rot"} -> arrays; gi / gd: dL/dimage, dL/ddepth the backward was fed.
      aggregate  <= 2e-5 per tensor, dL/dtau <= 1e-5, against the reference's algorithm in fp32 OR the same decisions evaluated in
                 double (oracle/gs_oracle_k7.inc) -- both are reported
      rows       worst <= worst_bar, share above 1e-3 <= share_bar over the rows WITHOUT a cause; a cause is a threshold test within
                 rounding of its threshold in one of the row's live pixels (oracle.flip_audit) or an ill-conditioned sum: the
                 rounding of the transmittances alone moves the row by a quarter of the bar (oracle.backward_with_conditioning)
      n_touched  an integer: every Gaussian whose count differs owns as many pixels at a T = 0.5 / alpha = 1/255 threshold
    Returns (summary string, per-tensor report, failures list: empty = pass)."""
    from oracle import oracle as O
    zero_a = np.zeros((1, f.H, f.W), np.float32)
    go, mass, reach = O.backward_with_conditioning(f, gi, gd, zero_a, pose_mode=True)
    failures = []
    O.set_backward_double(True)
    try:
        go64 = O.backward(f, gi, gd, zero_a, pose_mode=True)
    finally:
        O.set_backward_double(False)
    # (dL/dtau is a signed sum over every visible Gaussian: like the tensors it is held to the nearer of the two evaluations of the
    # reference's algorithm -- seen in tools/fuzz_split.py: 1.00004e-5 against the fp32 one on a room with 200 tiles split)
    e_tau32, e_tau64 = rel_l1(tau, go["tau"]), rel_l1(tau, go64["tau"])
    e_tau = min(e_tau32, e_tau64)
    if e_tau > 1e-5:
        failures.append(("tau", e_tau32, e_tau64))
    live = (np.abs(gi).sum(0) + np.abs(gd[0])) != 0
    near_half, flips, events, w_evt, w_all = O.flip_audit(f, live=live, weights=True)
    net = np.abs(go["opacities"]).reshape(-1).astype(np.float64)
    ill = reach >= 2.5e-4 * (net + 0.01 * max(float(net.mean()), 1e-300))
    unstable = flips | ill
    report = {}
    for k, ok in (("m3d", "means3D"), ("sh", "sh"), ("opac", "opacities"), ("scale", "scales"), ("rot", "rotations")):
        if grads.get(k) is None:
            continue
        b, b64 = go[ok], go64[ok]
        a = np.asarray(grads[k]).reshape(b.shape)
        e32, e64, d = rel_l1(a, b), rel_l1(a, b64), rel_l1(b, b64)
        if min(e32, e64) > 2e-5:
            failures.append((k, "aggregate", e32, e64, d))
        q = rows_with_a_cause(a, (b, b64), unstable, worst_bar, share_bar)
        if not q["ok"]:
            un = q["unexplained"][np.argsort(q["r"][q["unexplained"]])[::-1][:6]]
            failures.append((k, "rows", q["worst"], q["share"], [(int(i), float(q["r"][i]), float(w_evt[i]), int(w_all[i]), float(reach[i]), float(net[i])) for i in un]))
        report[k] = "vs fp32 %.2e, vs double %.2e (fp32 vs double %.2e); rows: worst %.3f (all: %.3f), > 1e-3: %.2e (all: %.2e), excused %d" % (
            e32, e64, d, q["worst"], q["worst_all"], q["share"], q["share_all"], q["excused"])
    summary = "live pixels %.3f; events %s; rows with a flip cause %d, ill-conditioned %d, visible %d; dL/dtau %.1e (fp32) / %.1e (double);" % (
        float(live.mean()), events, int(flips.sum()), int((ill & ~flips).sum()), int((f.radii > 0).sum()), e_tau32, e_tau64)
    if n_touched is not None:
        nt = np.asarray(n_touched).astype(np.int64).reshape(-1)
        dn = np.abs(nt - f.n_touched.astype(np.int64))
        bad = dn > near_half
        summary += " n_touched: %d Gaussians differ (sum %d), unexplained %d;" % (int((dn > 0).sum()), int(dn.sum()), int(bad.sum()))
        if bad.any():
            failures.append(("n_touched", int(bad.sum()), [(int(i), int(nt[i]), int(f.n_touched[i]), int(near_half[i])) for i in np.flatnonzero(bad)[:6]]))
    return summary, report, failures


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
