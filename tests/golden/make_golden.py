"""Generates tests/golden/ref_python_vectors.npz by IMPORTING the reference's Python helpers.

Runs only in the development container (needs /root/reference); the resulting .npz holds
inputs + expected outputs only (data, no reference source) and is what travels to the GPU box.

Pinned reference functions (all paths relative to /root/reference):
  gaussian_splatting/utils/sh_utils.py:57-112           eval_sh        -> SH colour (forward.cu:20-71 analogue)
  gaussian_splatting/utils/general_utils.py:81-127      build_rotation, build_scaling_rotation, strip_symmetric
  gaussian_splatting/scene/gaussian_model.py:27-31      build_covariance_from_scaling_rotation (restated call chain)
  gs_localization/pipelines/tools/graphics_utils.py:38-98   getWorld2View2, getProjectionMatrix2
  gs_localization/pipelines/tools/pose_utils.py:54-102  SO3_exp, V, SE3_exp
  gs_localization/pipelines/tools/descent_utils.py:85-123   get_loss_tracking (rgb + rgbd)
"""
import importlib.util
import os
import sys
import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_python_vectors.npz")


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    # the reference hard-codes device="cuda" in a few constructors; run them on the CPU
    _zeros = torch.zeros
    torch.zeros = lambda *a, **k: _zeros(*a, **{kk: v for kk, v in k.items() if kk != "device"})
    torch.Tensor.cuda = lambda self, *a, **k: self

    sh_utils = load("gaussian_splatting/utils/sh_utils.py", "ref_sh_utils")
    gen = load("gaussian_splatting/utils/general_utils.py", "ref_general_utils")
    gfx = load("gs_localization/pipelines/tools/graphics_utils.py", "ref_pl_graphics")
    pose = load("gs_localization/pipelines/tools/pose_utils.py", "ref_pose_utils")
    desc = load("gs_localization/pipelines/tools/descent_utils.py", "ref_descent_utils")

    rng = np.random.default_rng(1234)
    out = {}
    P = 257
    means = rng.normal(0, 2, (P, 3)).astype(np.float32)
    campos = np.array([0.3, -0.2, -4.0], np.float32)
    shs = (rng.normal(0, 1, (P, 16, 3)) * np.array([1.0] + [0.3] * 15)[None, :, None]).astype(np.float32)
    out["sh_means"], out["sh_campos"], out["sh_coeffs"] = means, campos, shs
    d = torch.tensor(means) - torch.tensor(campos)
    d = d / d.norm(dim=1, keepdim=True)
    for deg in range(4):
        M = (deg + 1) ** 2
        sh_view = torch.tensor(shs[:, :M]).transpose(1, 2)       # [P,3,M] as render() builds it
        rgb = torch.clamp_min(sh_utils.eval_sh(deg, sh_view, d) + 0.5, 0.0)
        out[f"sh_rgb_deg{deg}"] = rgb.numpy()

    scales = np.exp(rng.normal(-3, 0.7, (P, 3))).astype(np.float32)
    rots = rng.normal(0, 1, (P, 4)).astype(np.float32)
    rots /= np.linalg.norm(rots, axis=1, keepdims=True)
    for mod in (1.0, 0.7):
        L = gen.build_scaling_rotation(mod * torch.tensor(scales), torch.tensor(rots))
        cov = gen.strip_symmetric(L @ L.transpose(1, 2))
        out[f"cov3d_mod{mod}"] = cov.numpy()
    out["cov_scales"], out["cov_rots"] = scales, rots

    intr = np.array([[525, 525, 320, 240, 640, 480], [744, 760, 430.5, 236.25, 852, 480], [400, 400, 260, 180, 504, 378]], np.float64)
    out["proj_intr"] = intr
    out["proj_P"] = np.stack([gfx.getProjectionMatrix2(0.01, 100.0, cx=c[2], cy=c[3], fx=c[0], fy=c[1], W=int(c[4]), H=int(c[5])).numpy() for c in intr])

    taus = np.concatenate([rng.normal(0, 0.2, (6, 6)), rng.normal(0, 1e-7, (2, 6)), np.zeros((1, 6))]).astype(np.float64)
    out["se3_tau"] = taus
    out["se3_T"] = np.stack([pose.SE3_exp(torch.tensor(t)).numpy() for t in taus])
    Rt = out["se3_T"][0]
    out["w2v_in"] = Rt
    out["w2v_out"] = gfx.getWorld2View2(torch.tensor(Rt[:3, :3]), torch.tensor(Rt[:3, 3])).numpy()

    # tracking loss (7scenes_localize_full_dslam.py:296-297,323: monocular False, alpha .99, opacity_threshold .99)
    H, W = 24, 32
    image = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    depth = rng.uniform(0.5, 4, (1, H, W)).astype(np.float32)
    opacity = rng.uniform(0.9, 1.0, (1, H, W)).astype(np.float32)
    gt = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    gt_depth = rng.uniform(0, 4, (H, W)).astype(np.float32)
    gt_depth[rng.uniform(size=(H, W)) < 0.2] = 0.0
    grad_mask = rng.uniform(size=(1, H, W)) < 0.6

    class VP:
        pass
    vp = VP()
    vp.exposure_a = torch.tensor([0.05])
    vp.exposure_b = torch.tensor([-0.02])
    vp.original_image = torch.tensor(gt)
    vp.depth = gt_depth
    vp.grad_mask = torch.tensor(grad_mask)
    for mono in (True, False):
        cfg = {"Training": {"monocular": mono, "alpha": 0.99, "opacity_threshold": 0.99}}
        loss = desc.get_loss_tracking(cfg, torch.tensor(image), torch.tensor(depth), torch.tensor(opacity), vp)
        out[f"loss_mono{int(mono)}"] = np.float64(loss.item())
    # ---- round 5: BACKWARD of the SH and covariance stages, by autograd THROUGH the reference's own functions (K9: backward.cu:20-139,
    # 278-341).  New draws come after every earlier one, so the vectors above keep their bits.
    g_col = rng.normal(0, 1, (P, 3)).astype(np.float32)
    out["shb_dL_dcolor"] = g_col
    for deg in range(4):
        M = (deg + 1) ** 2
        m_t = torch.tensor(means, dtype=torch.float64, requires_grad=True)
        sh_t = torch.tensor(shs[:, :M], dtype=torch.float64, requires_grad=True)
        dd = m_t - torch.tensor(campos, dtype=torch.float64)
        dd = dd / dd.norm(dim=1, keepdim=True)                       # render(): dir_pp / dir_pp.norm (gaussian_renderer/__init__.py:75-77)
        rgb = torch.clamp_min(sh_utils.eval_sh(deg, sh_t.transpose(1, 2), dd) + 0.5, 0.0)
        (rgb * torch.tensor(g_col, dtype=torch.float64)).sum().backward()
        out[f"shb_dL_dsh_deg{deg}"] = sh_t.grad.numpy()
        out[f"shb_dL_dmean_deg{deg}"] = (m_t.grad if m_t.grad is not None else torch.zeros_like(m_t)).numpy()      # (degree 0 does not look at the direction)
        out[f"shb_rgb64_deg{deg}"] = rgb.detach().numpy()
    g_cov = rng.normal(0, 1, (P, 6)).astype(np.float32)
    out["covb_dL_dcov"] = g_cov
    for mod in (1.0, 0.7):
        s_t = torch.tensor(scales, dtype=torch.float64, requires_grad=True)
        q_t = torch.tensor(rots, dtype=torch.float64, requires_grad=True)
        Lm = gen.build_scaling_rotation(mod * s_t, q_t)
        cov = gen.strip_symmetric(Lm @ Lm.transpose(1, 2))
        (cov * torch.tensor(g_cov, dtype=torch.float64)).sum().backward()
        out[f"covb_dL_dscale_mod{mod}"] = s_t.grad.numpy()
        out[f"covb_dL_drot_mod{mod}"] = q_t.grad.numpy()              # (build_rotation normalises q inside: the tangential part of dL/dq)
    out.update(loss_image=image, loss_depth=depth, loss_opacity=opacity, loss_gt=gt, loss_gt_depth=gt_depth,
               loss_grad_mask=grad_mask, loss_exposure=np.array([0.05, -0.02], np.float32))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: getattr(v, "shape", ()) for k, v in out.items()})


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (development container only)")
    main()
