"""Generates tests/golden/pose_loop_vectors.npz by IMPORTING the reference's own pose-refinement Python
(development container only: needs /root/reference).  The .npz holds inputs and expected outputs -- data, no source.

What is executed here is the reference's code, loaded file by file (its `tools/__init__.py` needs plyfile / the CUDA
extension and is skipped; the modules below only need each other):
  gs_localization/pipelines/tools/pose_utils.py:54-122      SO3_exp, V, SE3_exp, update_pose
  gs_localization/pipelines/tools/descent_utils.py:85-123   get_loss_tracking, _rgb, _rgbd
  gs_localization/pipelines/tools/graphics_utils.py:38-98   getWorld2View2, getProjectionMatrix2
  gs_localization/pipelines/tools/camera_utils.py:38-158    Camera (R/T/deltas/exposure, world_view_transform, ...)
  torch.optim.Adam with the four parameter groups of 7scenes_localize_full_dslam.py:33-64

Fixture groups (SURVEY.md section 8(c) fixtures 8 and 9):
  track_*   tracking loss + its autograd gradients w.r.t. image, depth and the exposure pair   -> k_tracking_loss
  traj_*    N Adam steps + update_pose from given gradient 8-vectors: R, T, exposure, converged -> k_pose_step
  cam_*     Camera.world_view_transform / full_proj_transform / camera_center for given (R, T)  -> k_pose_init
  tau_*     float64 dL/dtau of small scenes: oracle/autograd_ref.py's differentiable restatement of the rasterizer
            composed with the REFERENCE's SE3_exp as the pose perturbation                      -> K8 pose gradient
  loop_*    K iterations of the reference's gradient_decent body where render/backward are the CPU oracle and loss,
            Adam and update_pose are the reference's: pose after every iteration               -> gsr_refine
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
REF = "/root/reference"
PL = "gs_localization/pipelines/tools/"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pose_loop_vectors.npz")


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def reference_modules():
    torch.Tensor.cuda = lambda self, *a, **k: self          # get_loss_tracking_rgb calls .cuda() on the ground truth
    sys.modules.setdefault("tools", types.ModuleType("tools"))      # the package shell only; its modules are the reference's files
    gfx = load(PL + "graphics_utils.py", "tools.graphics_utils")
    desc = load(PL + "descent_utils.py", "tools.descent_utils")
    pose = load(PL + "pose_utils.py", "tools.pose_utils")
    cam = load(PL + "camera_utils.py", "tools.camera_utils")
    return gfx, desc, pose, cam


def pose_optimizer(vp):
    """the four groups of 7scenes_localize_full_dslam.py:33-64 (one lr, torch defaults)"""
    return torch.optim.Adam([{"params": [p], "lr": 0.001, "name": n} for p, n in
                             ((vp.cam_rot_delta, "rot"), (vp.cam_trans_delta, "trans"), (vp.exposure_a, "exposure_a"),
                              (vp.exposure_b, "exposure_b"))])


def main():
    gfx, desc, pose, camu = reference_modules()
    from gs_localization_amd import scenes as S
    from oracle import oracle as O, autograd_ref as AG
    rng = np.random.default_rng(20260)
    out = {}

    # ---------------------------------------------------------------- track_*: loss and gradients
    for tag, (H, W) in (("a", (37, 53)), ("b", (24, 32))):
        image = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
        depth = rng.uniform(0.5, 4, (1, H, W)).astype(np.float32)
        opacity = rng.uniform(0.93, 1.03, (1, H, W)).astype(np.float32)
        gt = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
        gt_depth = rng.uniform(0, 4, (H, W)).astype(np.float32)
        gt_depth[rng.uniform(size=(H, W)) < 0.2] = 0.0
        mask = rng.uniform(size=(1, H, W)) < 0.7
        expo = np.array([0.07, -0.03], np.float32)
        out.update({f"track_{tag}_image": image, f"track_{tag}_depth": depth, f"track_{tag}_opacity": opacity, f"track_{tag}_gt": gt,
                    f"track_{tag}_gt_depth": gt_depth, f"track_{tag}_mask": mask, f"track_{tag}_exposure": expo})
        for mono in (0, 1):
            vp = types.SimpleNamespace(exposure_a=torch.tensor(expo[:1], requires_grad=True), exposure_b=torch.tensor(expo[1:], requires_grad=True),
                                       original_image=torch.tensor(gt), depth=gt_depth, grad_mask=torch.tensor(mask))
            cfg = {"Training": {"monocular": bool(mono), "alpha": 0.99, "opacity_threshold": 0.99}}
            im, dp = torch.tensor(image, requires_grad=True), torch.tensor(depth, requires_grad=True)
            loss = desc.get_loss_tracking(cfg, im, dp, torch.tensor(opacity), vp)
            loss.backward()
            k = f"track_{tag}_mono{mono}_"
            out[k + "loss"] = np.float64(loss.item())
            out[k + "dimage"] = im.grad.numpy()
            out[k + "ddepth"] = (dp.grad if dp.grad is not None else torch.zeros_like(dp)).numpy()
            out[k + "dexposure"] = np.array([vp.exposure_a.grad.item(), vp.exposure_b.grad.item()], np.float64)

    # ---------------------------------------------------------------- traj_* and cam_*: Adam + update_pose, camera matrices
    proj = gfx.getProjectionMatrix2(znear=0.01, zfar=100.0, fx=525.0, fy=525.0, cx=320.0, cy=240.0, W=640, H=480).transpose(0, 1)
    T0 = torch.tensor(S.se3_exp([0.3, -0.2, 0.5, 0.1, -0.3, 0.2]), dtype=torch.float32)
    cam = camu.Camera(0, None, None, torch.eye(4), proj, 525.0, 525.0, 320.0, 240.0, 1.0, 1.0, 480, 640, device="cpu")
    cam.update_RT(T0[:3, :3].clone(), T0[:3, 3].clone())
    out["cam_proj_raw_T"] = proj.numpy()
    out["cam_R"], out["cam_T"] = cam.R.numpy().copy(), cam.T.numpy().copy()
    out["cam_view"], out["cam_fullproj"], out["cam_center"] = (cam.world_view_transform.numpy().copy(), cam.full_proj_transform.numpy().copy(),
                                                               cam.camera_center.numpy().copy())
    N = 16
    grads = np.stack([(rng.normal(size=8) * 10.0 ** rng.uniform(-6, 1)).astype(np.float32) for _ in range(N)])

    def walk(thr):
        cam.update_RT(T0[:3, :3].clone(), T0[:3, 3].clone())
        with torch.no_grad():
            for p_ in (cam.cam_rot_delta, cam.cam_trans_delta, cam.exposure_a, cam.exposure_b):
                p_.zero_()
        opt = pose_optimizer(cam)
        rec = dict(R=[], T=[], ex=[], conv=[], view=[], norm=[])
        for g in grads:
            cam.cam_rot_delta.grad = torch.tensor(g[0:3])
            cam.cam_trans_delta.grad = torch.tensor(g[3:6])
            cam.exposure_a.grad = torch.tensor(g[6:7])
            cam.exposure_b.grad = torch.tensor(g[7:8])
            with torch.no_grad():
                opt.step()
                rec["norm"].append(float(torch.cat([cam.cam_trans_delta, cam.cam_rot_delta]).norm()))
                rec["conv"].append(bool(pose.update_pose(cam, converged_threshold=thr)))
            rec["R"].append(cam.R.numpy().copy()); rec["T"].append(cam.T.numpy().copy())
            rec["ex"].append([cam.exposure_a.item(), cam.exposure_b.item()])
            rec["view"].append(cam.world_view_transform.numpy().copy())
        return rec
    # Adam moves every component by about lr whatever the gradient's size, so |tau| stays near 2e-3: the reference's 1e-4 is
    # never reached in 16 steps.  The recorded walk uses the median step length as the threshold so that both outcomes occur.
    thr = float(np.median(walk(1e-4)["norm"]))
    rec = walk(thr)
    Rs, Ts, ex, conv, views = rec["R"], rec["T"], rec["ex"], rec["conv"], rec["view"]
    assert any(conv) and not all(conv)
    out["traj_threshold"] = np.float64(thr)
    out.update(traj_R0=T0[:3, :3].numpy(), traj_T0=T0[:3, 3].numpy(), traj_grads=grads, traj_R=np.stack(Rs), traj_T=np.stack(Ts),
               traj_exposure=np.array(ex, np.float64), traj_converged=np.array(conv), traj_view=np.stack(views))

    # ---------------------------------------------------------------- tau_*: float64 pose gradients with the reference's SE3_exp
    W2C = S.se3_exp([0.05, -0.03, 0.1, 0.02, -0.04, 0.03])
    scenes = {"sh3": dict(P=300, W=48, H=32, sh_degree=3, seed=3),
              "offcentre_white": dict(P=200, W=40, H=24, sh_degree=1, seed=5),
              "partial_tiles": dict(P=400, W=72, H=40, sh_degree=2, seed=17)}
    for name, kw in scenes.items():
        sc = S.small(**kw)
        if name == "offcentre_white":
            sc.bg[:] = 1.0
            sc.cx, sc.cy = 17.0, 14.5
        view, projm, proj_raw, campos = S.camera_matrices(sc, W2C)
        f = O.forward(sc.means3D, sc.opacities, view, projm, campos, sc.W, sc.H, sc.tanfovx, sc.tanfovy, sc.bg, sh_degree=sc.sh_degree,
                      shs=sc.shs, scales=sc.scales, rotations=sc.rotations, want_n_touched=True)
        g_rng = np.random.default_rng(kw["seed"] + 100)
        gc = g_rng.normal(size=(3, sc.H, sc.W)).astype(np.float32)
        gd = g_rng.normal(size=(1, sc.H, sc.W)).astype(np.float32)
        t64 = lambda a: torch.tensor(np.asarray(a, np.float64))
        tau = torch.zeros(6, dtype=torch.float64, requires_grad=True)
        col, dep, alp, aux = AG.render_autograd(f.state(), f.radii, t64(sc.means3D), t64(sc.opacities), t64(W2C), t64(proj_raw.T),
                                                sc.W, sc.H, sc.tanfovx, sc.tanfovy, t64(sc.bg), sh_degree=sc.sh_degree, shs=t64(sc.shs),
                                                scales=t64(sc.scales), rotations=t64(sc.rotations), tau=tau, depth_to_mean=True,
                                                se3_exp=pose.SE3_exp)
        assert np.abs(col.detach().numpy() - f.color).max() < 2e-5
        L = (col * t64(gc)).sum() + (dep * t64(gd[0])).sum()
        L.backward()
        out[f"tau_{name}_scene"] = np.array([kw["P"], kw["W"], kw["H"], kw["sh_degree"], kw["seed"]], np.int64)
        out[f"tau_{name}_cxcy_bg"] = np.array([sc.cx, sc.cy, sc.bg[0]], np.float64)
        out[f"tau_{name}_gc"], out[f"tau_{name}_gd"] = gc, gd
        out[f"tau_{name}_expected"] = tau.grad.numpy().copy()
    out["tau_w2c"] = W2C

    # ---------------------------------------------------------------- loop_*: the reference's loop body around the CPU oracle
    sc = S.small(P=3000, W=96, H=64, sh_degree=2, seed=31, scale_med=0.05)
    K = 8
    gt_view, gt_proj, proj_raw, gt_campos = S.camera_matrices(sc, np.eye(4))
    f_gt = O.forward(sc.means3D, sc.opacities, gt_view, gt_proj, gt_campos, sc.W, sc.H, sc.tanfovx, sc.tanfovy, sc.bg, sh_degree=sc.sh_degree,
                     shs=sc.shs, scales=sc.scales, rotations=sc.rotations, want_n_touched=True)
    d_t = rng.normal(size=3); d_t *= 0.02 / np.linalg.norm(d_t)
    d_r = rng.normal(size=3); d_r *= np.radians(1.0) / np.linalg.norm(d_r)
    init = torch.tensor(S.se3_exp(np.concatenate([d_t, d_r])), dtype=torch.float32)
    projT = gfx.getProjectionMatrix2(znear=sc.znear, zfar=sc.zfar, fx=sc.fx, fy=sc.fy, cx=sc.cx, cy=sc.cy, W=sc.W, H=sc.H).transpose(0, 1)
    vp = camu.Camera(0, torch.tensor(f_gt.color), f_gt.depth[0].copy(), torch.eye(4), projT, sc.fx, sc.fy, sc.cx, sc.cy, 1.0, 1.0, sc.H, sc.W,
                     device="cpu")
    vp.original_image = torch.tensor(f_gt.color)
    vp.depth = f_gt.depth[0].copy()
    vp.grad_mask = torch.ones((1, sc.H, sc.W), dtype=torch.bool)
    vp.update_RT(init[:3, :3].clone(), init[:3, 3].clone())
    opt = pose_optimizer(vp)
    cfg = {"Training": {"monocular": False, "alpha": 0.99, "opacity_threshold": 0.99}}
    lR, lT, lloss, ltau = [], [], [], []
    for it in range(K):
        view = vp.world_view_transform.numpy().astype(np.float32)
        projm = vp.full_proj_transform.numpy().astype(np.float32)
        campos = vp.camera_center.numpy().astype(np.float32)
        f = O.forward(sc.means3D, sc.opacities, view, projm, campos, sc.W, sc.H, sc.tanfovx, sc.tanfovy, sc.bg, sh_degree=sc.sh_degree,
                      shs=sc.shs, scales=sc.scales, rotations=sc.rotations, want_n_touched=True)
        im, dp = torch.tensor(f.color, requires_grad=True), torch.tensor(f.depth, requires_grad=True)
        opt.zero_grad()
        loss = desc.get_loss_tracking(cfg, im, dp, torch.tensor(f.alpha), vp)
        loss.backward()
        g = O.backward(f, im.grad.numpy(), dp.grad.numpy(), np.zeros((1, sc.H, sc.W), np.float32), pose_mode=True)
        vp.cam_trans_delta.grad = torch.tensor(np.asarray(g["tau"][:3], np.float32))
        vp.cam_rot_delta.grad = torch.tensor(np.asarray(g["tau"][3:], np.float32))
        with torch.no_grad():
            opt.step()
            pose.update_pose(vp, converged_threshold=1e-4)
        lR.append(vp.R.numpy().copy()); lT.append(vp.T.numpy().copy()); lloss.append(loss.item()); ltau.append(np.asarray(g["tau"], np.float64))
    out.update(loop_scene=np.array([3000, 96, 64, 2, 31], np.int64), loop_scale_med=np.float64(0.05), loop_init=init.numpy(),
               loop_gt_image=f_gt.color, loop_gt_depth=f_gt.depth[0], loop_R=np.stack(lR), loop_T=np.stack(lT),
               loop_loss=np.array(lloss, np.float64), loop_tau=np.stack(ltau))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    for k, v in out.items():
        print(" ", k, getattr(v, "shape", ()), getattr(v, "dtype", type(v)))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (development container only)")
    main()
