"""Generates tests/golden/masked_loop_vectors.npz: K iterations of the reference's gradient_decent() body UNDER THE REFERENCE'S MASK
(development container only: needs /root/reference).  Data only -- inputs and expected outputs.

Everything around the rasterizer is the reference's own code, imported (tests/golden/make_pose_golden.py, make_grad_mask_golden.py):
  camera_utils.py:164-193   Camera.compute_grad_mask (image_gradient / image_gradient_mask of descent_utils.py:33-67 behind it)
  7scenes_localize_full_dslam.py:126-149   create_mask (the function's definition, compiled from the parsed script), OR-ed in (:355-360)
  descent_utils.py:85-123   get_loss_tracking;  torch.optim.Adam over the four groups (:33-64);  pose_utils.py:105-122 update_pose
render / backward are the CPU oracle (the only stand-in, as in pose_loop_vectors.npz).  Round 5's recorded loop ran with an all-ones
mask, which no localiser of the reference does; this one is what `gsr_refine` is held to in tests/test_gpu_refine.py."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
OUT = os.path.join(HERE, "masked_loop_vectors.npz")


def main():
    import make_pose_golden as MP
    import make_grad_mask_golden as MG
    from gs_localization_amd import scenes as S
    from oracle import oracle as O
    MG.reference_modules()                     # (patches the filters' device="cuda"; loads tools.descent_utils / camera_utils)
    gfx, desc, pose, camu = MP.reference_modules()
    create_mask = MG.reference_function("gs_localization/pipelines/7scenes_localize_full_dslam.py", "create_mask")
    rng = np.random.default_rng(606)
    P, W, H, deg, seed, scale_med = 6000, 128, 96, 2, 41, 0.05
    sc = S.small(P=P, W=W, H=H, sh_degree=deg, seed=seed, scale_med=scale_med)
    K = 8
    gt_view, gt_proj, proj_raw, gt_campos = S.camera_matrices(sc, np.eye(4))
    fw = lambda view, projm, campos: O.forward(sc.means3D, sc.opacities, view, projm, campos, sc.W, sc.H, sc.tanfovx, sc.tanfovy, sc.bg,
                                               sh_degree=sc.sh_degree, shs=sc.shs, scales=sc.scales, rotations=sc.rotations, want_n_touched=True)
    f_gt = fw(gt_view, gt_proj, gt_campos)
    d_t = rng.normal(size=3); d_t *= 0.02 / np.linalg.norm(d_t)
    d_r = rng.normal(size=3); d_r *= np.radians(1.0) / np.linalg.norm(d_r)
    init = torch.tensor(S.se3_exp(np.concatenate([d_t, d_r])), dtype=torch.float32)
    projT = gfx.getProjectionMatrix2(znear=sc.znear, zfar=sc.zfar, fx=sc.fx, fy=sc.fy, cx=sc.cx, cy=sc.cy, W=sc.W, H=sc.H).transpose(0, 1)
    vp = camu.Camera(0, torch.tensor(f_gt.color), f_gt.depth[0].copy(), torch.eye(4), projT, sc.fx, sc.fy, sc.cx, sc.cy, 1.0, 1.0, sc.H, sc.W, device="cpu")
    cfg = {"Training": {"monocular": False, "alpha": 0.99, "opacity_threshold": 0.99, "edge_threshold": 1.1}, "Dataset": {"type": "tum"}}
    vp.compute_grad_mask(cfg)                                                                  # 7scenes_localize_full_dslam.py:355
    mask0 = vp.grad_mask.numpy()[0].copy()
    kp = np.stack([rng.uniform(0, W - 1, 14), rng.uniform(0, H - 1, 14)], 1).astype(np.float32)
    boxes = create_mask(mkpts_lst=kp, width=W, height=H, k=10)                                 # :359
    vp.grad_mask = vp.grad_mask | torch.tensor(boxes)                                          # :360
    mask = vp.grad_mask.numpy()[0].copy()
    assert 0.3 < mask.mean() < 0.8
    vp.update_RT(init[:3, :3].clone(), init[:3, 3].clone())
    opt = MP.pose_optimizer(vp)
    lR, lT, lloss, ltau = [], [], [], []
    for it in range(K):
        view = vp.world_view_transform.numpy().astype(np.float32)
        projm = vp.full_proj_transform.numpy().astype(np.float32)
        campos = vp.camera_center.numpy().astype(np.float32)
        f = fw(view, projm, campos)
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
    out = dict(loop_scene=np.array([P, W, H, deg, seed], np.int64), loop_scale_med=np.float64(scale_med), loop_init=init.numpy(),
               loop_gt_image=f_gt.color, loop_gt_depth=f_gt.depth[0], loop_mask_bits=np.packbits(mask), loop_mask_without_boxes_bits=np.packbits(mask0),
               loop_keypoints=kp, loop_R=np.stack(lR), loop_T=np.stack(lT), loop_loss=np.array(lloss, np.float64), loop_tau=np.stack(ltau))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; mask share", float(mask.mean()), "loss", lloss[0], "->", lloss[-1])


if __name__ == "__main__":
    main()
