"""float64 torch-autograd restatement of the rasterizer -- TEST INFRASTRUCTURE ONLY.

Purpose: validate the hand-written chain rule of oracle/gs_oracle.c (which restates
cuda_rasterizer/backward.cu) and, above all, the SE(3) pose gradient dL/dtau of the
un-vendored `diff_gaussian_rasterization_pose` package, for which no reference source
exists.  Discrete decisions (tile membership from radii / rect, sort order)
are frozen from an fp32 oracle forward; hard thresholds (power>0, alpha<1/255, T<1e-4) are
re-evaluated in fp64 and returned so the caller can assert they did not flip.

Reference quirks reproduced on purpose (SURVEY.md section 8(a) "quirks"):
  * alpha = min(0.99, o*G): gradient passes through as if unclamped (backward.cu:511-512,562,578)
  * 1.3*tanfov clamp: clamped t.x/t.y are constants for the derivative (backward.cu:168-176,262-264)
  * package (A): per-Gaussian depth z_i receives no gradient (backward.cu:539-543); package (B): it does.
The grad_alpha quirk (backward.cu:545-547) is NOT an exact derivative and is therefore not
restated here: use grad_alpha = 0 when comparing against this module.
"""
import math
import numpy as np
import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def skew(x):
    z = torch.zeros((), dtype=x.dtype)
    return torch.stack([torch.stack([z, -x[2], x[1]]), torch.stack([x[2], z, -x[0]]), torch.stack([-x[1], x[0], z])])


def se3_exp_first_order_safe(tau):
    """SE3_exp of gs_localization/pipelines/tools/pose_utils.py:90-102, small-angle branch
    (:62-63, :79-80) -- tau is evaluated at 0, where that branch is the one taken."""
    rho, th = tau[:3], tau[3:]
    Wm = skew(th)
    W2 = Wm @ Wm
    I = torch.eye(3, dtype=tau.dtype)
    R = I + Wm + 0.5 * W2
    V = I + 0.5 * Wm + W2 / 6.0
    T = torch.eye(4, dtype=tau.dtype)
    T = T.clone()
    T[:3, :3] = R
    T[:3, 3] = V @ rho
    return T


def eval_sh_deg(deg, sh, d):
    """sh [P,M,3], d [P,3] unit -> [P,3].  Same polynomial as forward.cu:20-71."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = SH_C0 * sh[:, 0]
    if deg > 0:
        r = r - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            r = (r + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
                 + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                r = (r + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                     + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                     + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                     + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return r + 0.5


def render_autograd(fwd_state, radii, means3D, opacities, w2c, P_raw, W, H, tanfovx, tanfovy, bg, sh_degree=0,
                    shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                    scale_modifier=1.0, tau=None, depth_to_mean=False, se3_exp=None):
    """All tensor inputs are float64 torch tensors (leaf tensors may require grad).
    fwd_state / radii come from an fp32 oracle forward of the same inputs (frozen decisions).
    w2c: 4x4 world-to-camera; P_raw: 4x4 intrinsics projection (not transposed).
    tau: optional 6-vector [rho, theta] (zeros, requires_grad) applied as exp(tau) @ w2c.
    se3_exp: the exponential to use for that (tests/golden/make_pose_golden.py passes the reference's own SE3_exp,
    imported from pose_utils.py); default: the restatement above.
    Returns color[3,H,W], depth[H,W], alpha[H,W], aux dict."""
    dt = torch.float64
    Pn = means3D.shape[0]
    T_w2c = w2c if tau is None else (se3_exp or se3_exp_first_order_safe)(tau) @ w2c
    Rm, tv = T_w2c[:3, :3], T_w2c[:3, 3]
    p_view = means3D @ Rm.T + tv
    full = P_raw @ T_w2c
    ph = torch.cat([means3D, torch.ones(Pn, 1, dtype=dt)], 1) @ full.T
    p_w = 1.0 / (ph[:, 3] + 1e-7)
    m2x = ((ph[:, 0] * p_w + 1.0) * W - 1.0) * 0.5
    m2y = ((ph[:, 1] * p_w + 1.0) * H - 1.0) * 0.5
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)

    if cov3D_precomp is None:
        q = rotations
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        Rq = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], 1),
                          torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], 1),
                          torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1)], 1)
        S = torch.diag_embed(scale_modifier * scales)
        L = Rq @ S
        Sig = L @ L.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sig = torch.stack([torch.stack([c[:, 0], c[:, 1], c[:, 2]], 1), torch.stack([c[:, 1], c[:, 3], c[:, 4]], 1),
                           torch.stack([c[:, 2], c[:, 4], c[:, 5]], 1)], 1)

    tx, ty, tz = p_view[:, 0], p_view[:, 1], p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = tx / tz, ty / tz
    cx_out = (txtz < -limx) | (txtz > limx)
    cy_out = (tytz < -limy) | (tytz > limy)
    # clamped values enter J as constants (see module docstring)
    txc = torch.where(cx_out, (torch.clamp(txtz, -limx, limx) * tz).detach(), tx)
    tyc = torch.where(cy_out, (torch.clamp(tytz, -limy, limy) * tz).detach(), ty)
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -(fx * txc) / (tz * tz)], 1),
                     torch.stack([zero, fy / tz, -(fy * tyc) / (tz * tz)], 1)], 1)      # [P,2,3]
    SigC = Rm @ Sig @ Rm.T
    cov2 = J @ SigC @ J.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c_ = cov2[:, 1, 1] + 0.3
    det = a * c_ - b * b
    conA, conB, conC = c_ / det, -b / det, a / det

    if colors_precomp is None:
        campos = -(Rm.T @ tv)
        d = means3D - campos
        d = d / d.norm(dim=1, keepdim=True)
        col = torch.clamp_min(eval_sh_deg(sh_degree, shs, d), 0.0)
    else:
        col = colors_precomp

    z_i = p_view[:, 2] if depth_to_mean else p_view[:, 2].detach()
    opac = opacities.reshape(-1)

    # frozen tile membership from fp32 oracle (auxiliary.h:46-56 on the oracle's means2D / radii)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    m2 = fwd_state["means2D"].astype(np.float32)
    rad = np.asarray(radii)
    trunc = lambda v: np.trunc(v).astype(np.int64)
    x0 = np.clip(trunc((m2[:, 0] - rad) / np.float32(16)), 0, gx)
    y0 = np.clip(trunc((m2[:, 1] - rad) / np.float32(16)), 0, gy)
    x1 = np.clip(trunc((m2[:, 0] + rad + 15) / np.float32(16)), 0, gx)
    y1 = np.clip(trunc((m2[:, 1] + rad + 15) / np.float32(16)), 0, gy)
    depths32 = fwd_state["depths"]
    order = [i for i in np.lexsort((np.arange(Pn), depths32.view(np.uint32))) if rad[i] > 0]

    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pxf, pyf = xs.reshape(-1).to(dt), ys.reshape(-1).to(dt)
    tpx, tpy = (xs.reshape(-1) // 16).numpy(), (ys.reshape(-1) // 16).numpy()
    N = W * H
    T = torch.ones(N, dtype=dt)
    C = torch.zeros(N, 3, dtype=dt)
    D = torch.zeros(N, dtype=dt)
    done = torch.zeros(N, dtype=torch.bool)
    n_touched = np.zeros(Pn, np.int64)
    blended = []
    for i in order:
        inl = torch.from_numpy((tpx >= x0[i]) & (tpx < x1[i]) & (tpy >= y0[i]) & (tpy < y1[i]))
        if not bool(inl.any()):
            continue
        dx, dy = m2x[i] - pxf, m2y[i] - pyf
        power = -0.5 * (conA[i] * dx * dx + conC[i] * dy * dy) - conB[i] * dx * dy
        G = torch.exp(power)
        oG = opac[i] * G
        alpha = oG + (torch.clamp(oG, max=0.99) - oG).detach()
        ok = inl & (~done) & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
        test_T = T * (1 - alpha)
        stop = ok & (test_T.detach() < 1e-4)
        done = done | stop
        ok = ok & ~stop
        w = torch.where(ok, alpha * T, torch.zeros_like(T))
        C = C + w[:, None] * col[i][None, :]
        D = D + w * z_i[i]
        n_touched[i] = int((ok & (test_T.detach() > 0.5)).sum())
        T = torch.where(ok, test_T, T)
        blended.append(ok)
    color = (C + T[:, None] * bg[None, :]).T.reshape(3, H, W)
    depth = D.reshape(H, W)
    alpha_out = (1 - T).reshape(H, W)
    n_contrib_count = torch.stack(blended).sum(0).reshape(H, W) if blended else torch.zeros(H, W)
    return color, depth, alpha_out, dict(n_touched=n_touched, n_blended=n_contrib_count)
