"""TEST INFRASTRUCTURE: drives the drop-in packages the way the reference's localisation scripts do.

The reference scripts cannot run here (hard-coded D:/ paths; h5py, cv2, munch, plyfile absent; no datasets), so the
`-m gpu` tests, the tools and bench.py's "python loop" leg need something that makes the same calls in the same order:
  one render through `diff_gaussian_rasterization_pose` with theta / rho       (tools/__init__.py:24-153)
  tracking loss, backward, Adam over four groups, SE(3) pose update            (7scenes_localize_full_dslam.py:29-93)
This module is that driver, written against the behaviour of those call sites -- not a copy of them -- and pinned to the
reference's own code by tests/test_pose_golden.py (fixtures produced by tests/golden/make_pose_golden.py, which imports
pose_utils.py / descent_utils.py / camera_utils.py from /root/reference).  Nothing in the product package imports it.
"""
import math

import numpy as np
import torch

from diff_gaussian_rasterization_pose import GaussianRasterizationSettings, GaussianRasterizer
from gs_localization_amd.pipelines import GaussianMap, FusedRefiner, pose_errors, TRACKING_CONFIG  # noqa: F401  (re-exported for the callers)


def fov_of(focal, pixels):
    return 2.0 * math.atan(0.5 * pixels / focal)


def intrinsics_projection(scene, device):
    """P^T as the reference's cameras hold it (`projection_matrix`, camera_utils.py:129-133), from the scene's intrinsics."""
    from gs_localization_amd import scenes as S
    return torch.tensor(S.camera_matrices(scene)[2], dtype=torch.float32, device=device)


def _hat(v):
    z = v.new_zeros(())
    return torch.stack([torch.stack([z, -v[2], v[1]]), torch.stack([v[2], z, -v[0]]), torch.stack([-v[1], v[0], z])])


def se3_exp(tau):
    """[rho, theta] -> 4x4 rigid transform; closed form with the first-order series below 1e-5 rad."""
    rho, theta = tau[:3], tau[3:]
    K = _hat(theta)
    K2 = K @ K
    a = torch.linalg.vector_norm(theta)
    if a < 1e-5:
        ca, cb, cc = 1.0, 0.5, 1.0 / 6.0
    else:
        ca, cb, cc = torch.sin(a) / a, (1 - torch.cos(a)) / a**2, (a - torch.sin(a)) / a**3
    I = torch.eye(3, dtype=tau.dtype, device=tau.device)
    out = torch.eye(4, dtype=tau.dtype, device=tau.device)
    out[:3, :3] = I + ca * K + cb * K2
    out[:3, 3] = (I + cb * K + cc * K2) @ rho
    return out


class QueryFrame:
    """The attributes the reference's loop reads and writes on its camera object."""

    def __init__(self, uid, proj_raw_T, scene, device, gt_w2c=None):
        self.uid, self.device = uid, device
        self.R = torch.eye(3, device=device)
        self.T = torch.zeros(3, device=device)
        gt = torch.eye(4, device=device) if gt_w2c is None else gt_w2c
        self.R_gt, self.T_gt = gt[:3, :3], gt[:3, 3]
        self.original_image = self.depth = self.grad_mask = None
        self.fx, self.fy, self.cx, self.cy = scene.fx, scene.fy, scene.cx, scene.cy
        self.image_height, self.image_width = scene.H, scene.W
        self.FoVx, self.FoVy = fov_of(scene.fx, scene.W), fov_of(scene.fy, scene.H)
        mk = lambda n: torch.nn.Parameter(torch.zeros(n, device=device))
        self.cam_rot_delta, self.cam_trans_delta, self.exposure_a, self.exposure_b = mk(3), mk(3), mk(1), mk(1)
        self.projection_matrix = proj_raw_T.to(device)

    def update_RT(self, R, t):
        self.R, self.T = R.to(self.device), t.to(self.device)

    @property
    def world_view_transform(self):          # (W2C)^T
        m = torch.eye(4, device=self.device)
        m[:3, :3] = self.R.T
        m[3, :3] = self.T
        return m

    @property
    def full_proj_transform(self):
        return self.world_view_transform @ self.projection_matrix

    @property
    def camera_center(self):
        return -(self.R.T @ self.T)


def render(frame, gmap, background, scaling_modifier=1.0):
    """One render() of the pose package as the localisation scripts issue it: settings from the frame, SH colours,
    scale / rotation covariances, theta / rho = the frame's pose deltas.  Returns the same dictionary keys."""
    xyz = gmap.get_xyz
    if xyz.shape[0] == 0:
        return None
    means2D = torch.zeros_like(xyz, requires_grad=True)
    settings = GaussianRasterizationSettings(
        image_height=int(frame.image_height), image_width=int(frame.image_width),
        tanfovx=math.tan(0.5 * frame.FoVx), tanfovy=math.tan(0.5 * frame.FoVy), bg=background, scale_modifier=scaling_modifier,
        viewmatrix=frame.world_view_transform, projmatrix=frame.full_proj_transform, projmatrix_raw=frame.projection_matrix,
        sh_degree=gmap.active_sh_degree, campos=frame.camera_center, prefiltered=False, debug=False)
    scales = gmap.get_scaling
    if scales.shape[-1] == 1:
        scales = scales.expand(-1, 3)
    image, radii, depth, opacity, n_touched = GaussianRasterizer(raster_settings=settings)(
        means3D=xyz, means2D=means2D, opacities=gmap.get_opacity, shs=gmap.get_features, colors_precomp=None, scales=scales,
        rotations=gmap.get_rotation, cov3D_precomp=None, theta=frame.cam_rot_delta, rho=frame.cam_trans_delta)
    return {"render": image, "viewspace_points": means2D, "visibility_filter": radii > 0, "radii": radii, "depth": depth,
            "opacity": opacity, "n_touched": n_touched}


def tracking_loss(config, image, depth, opacity, frame):
    """mean over 3HW of [opacity > thr] |exp(a) image + b - gt| inside grad_mask, plus (1 - alpha) x the same for depth where
    the sensor depth is valid (RGB-D configuration)."""
    tr = config["Training"]
    seen = (opacity > tr["opacity_threshold"]).reshape(depth.shape)
    gm = frame.grad_mask
    exposed = torch.exp(frame.exposure_a) * image + frame.exposure_b
    loss = (seen * (exposed * gm - frame.original_image.to(image.device) * gm).abs()).mean()
    if tr["monocular"]:
        return loss
    gd = frame.depth if torch.is_tensor(frame.depth) else torch.from_numpy(frame.depth)
    gd = gd.to(dtype=torch.float32, device=image.device)[None]
    dm = (gd > 0.01).reshape(depth.shape) * seen * gm
    return loss + (1.0 - tr.get("alpha", 0.98)) * (depth * dm - gd * dm).abs().mean()


def pose_adam(frame, lr=0.001):
    return torch.optim.Adam([{"params": [p], "lr": lr} for p in
                             (frame.cam_rot_delta, frame.cam_trans_delta, frame.exposure_a, frame.exposure_b)])


def apply_pose_delta(frame, converged_threshold=1e-4):
    """W2C <- exp([trans_delta, rot_delta]) W2C, deltas back to zero; True when the step was below the threshold."""
    tau = torch.cat([frame.cam_trans_delta, frame.cam_rot_delta]).detach()
    w2c = torch.eye(4, device=tau.device)
    w2c[:3, :3], w2c[:3, 3] = frame.R, frame.T
    new = se3_exp(tau) @ w2c
    frame.update_RT(new[:3, :3], new[:3, 3])
    small = tau.norm() < converged_threshold
    frame.cam_rot_delta.data.zero_()
    frame.cam_trans_delta.data.zero_()
    return small


def loop_iteration(frame, config, gmap, background, optimizer):
    pkg = render(frame, gmap, background)
    optimizer.zero_grad()
    tracking_loss(config, pkg["render"], pkg["depth"], pkg["opacity"], frame).backward()
    with torch.no_grad():
        optimizer.step()
        converged = apply_pose_delta(frame)
    return converged, pkg


def python_loop(frame, config, R0, T0, gmap, background, iters=50):
    """The refinement of one query frame through autograd, with the reference's per-iteration convergence test (a host sync)."""
    frame.update_RT(R0, T0)
    opt = pose_adam(frame)
    pkg = None
    for _ in range(iters):
        converged, pkg = loop_iteration(frame, config, gmap, background, opt)
        if converged:
            break
    return frame.R, frame.T, pkg


N_KEYPOINTS = 500          # boxes OR-ed into a frame's mask, as the scripts do with the detector's keypoints of score > 0.2


def frame_keypoints(W, H, uid=0, n=N_KEYPOINTS):
    """seeded stand-ins for `group['keypoints'][group['scores'][:] > 0.2]` (7scenes_localize_full_dslam.py:357-358): [n, 2] (x, y)"""
    rng = np.random.default_rng(7000 + int(uid))
    return np.stack([rng.uniform(0, W - 1, n), rng.uniform(0, H - 1, n)], 1).astype(np.float32)


def reference_mask(original_image, uid=0, config=TRACKING_CONFIG, keypoints=True):
    """The mask the reference's localisers refine under: `viewpoint.compute_grad_mask(config)` OR-ed with 10 x 10-ish boxes around the
    frame's keypoints (7scenes_localize_full_dslam.py:355-360), computed by the product's `gsr_grad_mask` (pinned bit for bit to the
    reference's own Python by tests/test_grad_mask.py)."""
    from gs_localization_amd import pipelines as PL
    H, W = int(original_image.shape[-2]), int(original_image.shape[-1])
    kp = frame_keypoints(W, H, uid) if keypoints else None
    return PL.grad_mask(original_image, config["Training"]["edge_threshold"], kp, 10)


def make_frame(scene, gmap, device, background=None, uid=0, mask="reference"):
    """A query frame whose observations (image, depth) are renders of the map at the identity pose.
    mask = "reference": the gradient mask of that image as the reference's scripts build it (Scharr gradient above 1.1 x its
    median, keypoint boxes OR-ed in) -- what every localiser of the reference refines under; "ones": every pixel (the
    secondary leg of bench.py, and what rounds 1-5 measured)."""
    bg = torch.zeros(3, device=device) if background is None else background
    fr = QueryFrame(uid, intrinsics_projection(scene, device), scene, device)
    with torch.no_grad():
        pkg = render(fr, gmap, bg)
    fr.original_image = pkg["render"].detach().clone()
    fr.depth = pkg["depth"].detach()[0].clone()
    if mask == "ones":
        fr.grad_mask = torch.ones((1, scene.H, scene.W), dtype=torch.bool, device=device)
    else:
        assert mask == "reference", mask
        fr.grad_mask = reference_mask(fr.original_image, uid)
    return fr


def perturbed_start(seed, trans=0.02, rot_deg=1.0, device="cpu"):
    """identity moved by `trans` metres and `rot_deg` degrees in seeded random directions (SURVEY.md 8(c) fixture 9)"""
    from gs_localization_amd import scenes as S
    rng = np.random.default_rng(seed)
    dt = rng.normal(size=3); dt *= trans / np.linalg.norm(dt)
    dr = rng.normal(size=3); dr *= math.radians(rot_deg) / np.linalg.norm(dr)
    return torch.tensor(S.se3_exp(np.concatenate([dt, dr])), dtype=torch.float32, device=device)
