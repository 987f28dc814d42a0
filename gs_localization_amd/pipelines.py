"""Host-side mirror of the pose-refinement loop that the metric times.

Same names / argument meaning as the reference callers (paths relative to /root/reference):
  render()             gs_localization/pipelines/tools/__init__.py:24-153
  Camera               gs_localization/pipelines/tools/camera_utils.py:38-205
  get_loss_tracking*   gs_localization/pipelines/tools/descent_utils.py:85-123
  SO3_exp/V/SE3_exp, update_pose   gs_localization/pipelines/tools/pose_utils.py:54-122
  gradient_decent      gs_localization/pipelines/7scenes_localize_full_dslam.py:29-93
  getProjectionMatrix2, getWorld2View2   gs_localization/pipelines/tools/graphics_utils.py:38-98
  pose errors          gs_localization/pipelines/7scenes_localize_full_dslam.py:368-377

The reference scripts themselves cannot run here (hard-coded D:/ paths, h5py/cv2/munch/plyfile
missing, no datasets -- SURVEY.md section 0 fact 5); this module replays their call sequence so that the
drop-in packages are exercised exactly as those scripts would exercise them.
"""
import math

import numpy as np
import torch
from torch import nn

from diff_gaussian_rasterization_pose import GaussianRasterizationSettings, GaussianRasterizer


# ------------------------------------------------------------------ graphics_utils
def getWorld2View2(R, t, translate=None, scale=1.0):
    Rt = torch.zeros((4, 4), device=R.device, dtype=R.dtype)
    Rt[:3, :3] = R
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    if translate is None and scale == 1.0:
        return Rt      # inverse(inverse(Rt)) of graphics_utils.py:38-51 with the default arguments
    C2W = torch.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + translate.to(R.device)) * scale
    return torch.linalg.inv(C2W)


def getProjectionMatrix2(znear, zfar, cx, cy, fx, fy, W, H):
    left = ((2 * cx - W) / W - 1.0) * W / 2.0
    right = ((2 * cx - W) / W + 1.0) * W / 2.0
    top = ((2 * cy - H) / H + 1.0) * H / 2.0
    bottom = ((2 * cy - H) / H - 1.0) * H / 2.0
    left = znear / fx * left
    right = znear / fx * right
    top = znear / fy * top
    bottom = znear / fy * bottom
    P = torch.zeros(4, 4)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


# ------------------------------------------------------------------ pose_utils
def skew_sym_mat(x):
    ssm = torch.zeros(3, 3, device=x.device, dtype=x.dtype)
    ssm[0, 1] = -x[2]
    ssm[0, 2] = x[1]
    ssm[1, 0] = x[2]
    ssm[1, 2] = -x[0]
    ssm[2, 0] = -x[1]
    ssm[2, 1] = x[0]
    return ssm


def SO3_exp(theta):
    W = skew_sym_mat(theta)
    W2 = W @ W
    angle = torch.norm(theta)
    I = torch.eye(3, device=theta.device, dtype=theta.dtype)
    if angle < 1e-5:
        return I + W + 0.5 * W2
    return I + (torch.sin(angle) / angle) * W + ((1 - torch.cos(angle)) / (angle**2)) * W2


def V(theta):
    I = torch.eye(3, device=theta.device, dtype=theta.dtype)
    W = skew_sym_mat(theta)
    W2 = W @ W
    angle = torch.norm(theta)
    if angle < 1e-5:
        return I + 0.5 * W + (1.0 / 6.0) * W2
    return I + W * ((1.0 - torch.cos(angle)) / (angle**2)) + W2 * ((angle - torch.sin(angle)) / (angle**3))


def SE3_exp(tau):
    rho = tau[:3]
    theta = tau[3:]
    T = torch.eye(4, device=tau.device, dtype=tau.dtype)
    T[:3, :3] = SO3_exp(theta)
    T[:3, 3] = V(theta) @ rho
    return T


def update_pose(camera, converged_threshold=1e-4):
    tau = torch.cat([camera.cam_trans_delta, camera.cam_rot_delta], axis=0)
    T_w2c = torch.eye(4, device=tau.device)
    T_w2c[0:3, 0:3] = camera.R
    T_w2c[0:3, 3] = camera.T
    new_w2c = SE3_exp(tau) @ T_w2c
    converged = tau.norm() < converged_threshold
    camera.update_RT(new_w2c[0:3, 0:3], new_w2c[0:3, 3])
    camera.cam_rot_delta.data.fill_(0)
    camera.cam_trans_delta.data.fill_(0)
    return converged


# ------------------------------------------------------------------ camera_utils
class Camera(nn.Module):
    def __init__(self, uid, color, depth, gt_T, projection_matrix, fx, fy, cx, cy, fovx, fovy, image_height,
                 image_width, device="cuda:0"):
        super().__init__()
        self.uid = uid
        self.device = device
        T = torch.eye(4, device=device)
        self.R = T[:3, :3]
        self.T = T[:3, 3]
        self.R_gt = gt_T[:3, :3]
        self.T_gt = gt_T[:3, 3]
        self.original_image = color
        self.depth = depth
        self.grad_mask = None
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.FoVx, self.FoVy = fovx, fovy
        self.image_height, self.image_width = image_height, image_width
        self.cam_rot_delta = nn.Parameter(torch.zeros(3, requires_grad=True, device=device))
        self.cam_trans_delta = nn.Parameter(torch.zeros(3, requires_grad=True, device=device))
        self.exposure_a = nn.Parameter(torch.tensor([0.0], requires_grad=True, device=device))
        self.exposure_b = nn.Parameter(torch.tensor([0.0], requires_grad=True, device=device))
        self.projection_matrix = projection_matrix.to(device=device)

    @property
    def world_view_transform(self):
        return getWorld2View2(self.R, self.T).transpose(0, 1)

    @property
    def full_proj_transform(self):
        return (self.world_view_transform.unsqueeze(0).bmm(self.projection_matrix.unsqueeze(0))).squeeze(0)

    @property
    def camera_center(self):
        # = world_view_transform.inverse()[3, :3] of camera_utils.py:156-158 for a rigid W2C
        return -(self.R.transpose(0, 1) @ self.T)

    def update_RT(self, R, t):
        self.R = R.to(device=self.device)
        self.T = t.to(device=self.device)


# ------------------------------------------------------------------ descent_utils
def get_loss_tracking(config, image, depth, opacity, viewpoint, initialization=False):
    image_ab = (torch.exp(viewpoint.exposure_a)) * image + viewpoint.exposure_b
    if config["Training"]["monocular"]:
        return get_loss_tracking_rgb(config, image_ab, depth, opacity, viewpoint)
    return get_loss_tracking_rgbd(config, image_ab, depth, opacity, viewpoint)


def get_loss_tracking_rgb(config, image, depth, opacity, viewpoint):
    gt_image = viewpoint.original_image.to(image.device)
    opacity_mask = (opacity > config["Training"]["opacity_threshold"]).view(*depth.shape)
    l1 = opacity_mask * torch.abs(image * viewpoint.grad_mask - gt_image * viewpoint.grad_mask)
    return l1.mean()


def get_loss_tracking_rgbd(config, image, depth, opacity, viewpoint, initialization=False):
    alpha = config["Training"]["alpha"] if "alpha" in config["Training"] else 0.98
    gt_depth = viewpoint.depth
    if not torch.is_tensor(gt_depth):
        gt_depth = torch.from_numpy(gt_depth)
    gt_depth = gt_depth.to(dtype=torch.float32, device=image.device)[None]
    depth_pixel_mask = (gt_depth > 0.01).view(*depth.shape)
    opacity_mask = (opacity > config["Training"]["opacity_threshold"]).view(*depth.shape)
    l1_rgb = get_loss_tracking_rgb(config, image, depth, opacity, viewpoint)
    depth_mask = depth_pixel_mask * opacity_mask * viewpoint.grad_mask
    l1_depth = torch.abs(depth * depth_mask - gt_depth * depth_mask)
    return l1_rgb + (1 - alpha) * l1_depth.mean()


# ------------------------------------------------------------------ map container
class GaussianMap:
    """Read-only localisation map exposing the getters render() uses (the subset of
    tools/gaussian_model.py:77-106).  Activations are applied once at load time -- the map never
    changes during localisation; tensors keep requires_grad=True like the
    reference's nn.Parameters (tools/gaussian_model.py:437-462)."""

    def __init__(self, xyz, features, opacity, scaling, rotation, sh_degree, requires_grad=True):
        self.max_sh_degree = sh_degree
        self.active_sh_degree = sh_degree
        self._t = [xyz, features, opacity, scaling, rotation]
        for t in self._t:
            t.requires_grad_(requires_grad)

    @staticmethod
    def from_scene(scene, device="cuda:0", requires_grad=True):
        t = lambda a: torch.tensor(np.asarray(a, np.float32), device=device)
        return GaussianMap(t(scene.means3D), t(scene.shs), t(scene.opacities), t(scene.scales), t(scene.rotations),
                           scene.sh_degree, requires_grad)

    @staticmethod
    def from_ply(path, device="cuda:0", requires_grad=True, max_sh_degree=None):
        """load_ply (tools/gaussian_model.py:377-467) straight into the device layout: the file's vertex rows are
        uploaded as stored and converted (column gather, SH layout, activations) by one HIP kernel."""
        from . import map_io
        xyz, shs, opac, scales, rots, deg = map_io.load_map_tensors(path, device, activate=True, max_sh_degree=max_sh_degree)
        return GaussianMap(xyz, shs, opac, scales, rots, deg, requires_grad)

    get_xyz = property(lambda s: s._t[0])
    get_features = property(lambda s: s._t[1])
    get_opacity = property(lambda s: s._t[2])
    get_scaling = property(lambda s: s._t[3])
    get_rotation = property(lambda s: s._t[4])


class PipelineParams:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False


# ------------------------------------------------------------------ tools/__init__.py render()
def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, mask=None):
    if pc.get_xyz.shape[0] == 0:
        return None
    screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx,
        tanfovy=tanfovy,
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        projmatrix_raw=viewpoint_camera.projection_matrix,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=False,
    )
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    means3D = pc.get_xyz
    means2D = screenspace_points
    opacity = pc.get_opacity
    if pc.get_scaling.shape[-1] == 1:
        scales = pc.get_scaling.repeat(1, 3)
    else:
        scales = pc.get_scaling
    rotations = pc.get_rotation
    shs = pc.get_features if override_color is None else None
    rendered_image, radii, depth, opacity, n_touched = rasterizer(
        means3D=means3D, means2D=means2D, shs=shs, colors_precomp=override_color, opacities=opacity, scales=scales,
        rotations=rotations, cov3D_precomp=None, theta=viewpoint_camera.cam_rot_delta,
        rho=viewpoint_camera.cam_trans_delta)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "depth": depth, "opacity": opacity, "n_touched": n_touched}


# ------------------------------------------------------------------ the timed loop
def make_pose_optimizer(viewpoint):
    """Adam groups of 7scenes_localize_full_dslam.py:33-64"""
    return torch.optim.Adam([
        {"params": [viewpoint.cam_rot_delta], "lr": 0.001, "name": "rot_{}".format(viewpoint.uid)},
        {"params": [viewpoint.cam_trans_delta], "lr": 0.001, "name": "trans_{}".format(viewpoint.uid)},
        {"params": [viewpoint.exposure_a], "lr": 0.001, "name": "exposure_a_{}".format(viewpoint.uid)},
        {"params": [viewpoint.exposure_b], "lr": 0.001, "name": "exposure_b_{}".format(viewpoint.uid)},
    ])


def refine_iteration(viewpoint, config, Model, pipeline_params, background, pose_optimizer):
    """One body of the loop at 7scenes_localize_full_dslam.py:66-91.  Returns (converged, render_pkg)."""
    render_pkg = render(viewpoint, Model, pipeline_params, background)
    image, depth, opacity = render_pkg["render"], render_pkg["depth"], render_pkg["opacity"]
    pose_optimizer.zero_grad()
    loss_tracking = get_loss_tracking(config, image, depth, opacity, viewpoint)
    loss_tracking.backward()
    with torch.no_grad():
        pose_optimizer.step()
        converged = update_pose(viewpoint, converged_threshold=1e-4)
    return converged, render_pkg


def gradient_decent(viewpoint, config, initial_R, initial_T, Model, pipeline_params, background, iters=50):
    viewpoint.update_RT(initial_R, initial_T)
    pose_optimizer = make_pose_optimizer(viewpoint)
    render_pkg = None
    for _ in range(iters):
        converged, render_pkg = refine_iteration(viewpoint, config, Model, pipeline_params, background, pose_optimizer)
        if converged:
            break
    return viewpoint.R, viewpoint.T, render_pkg


class FusedRefiner:
    """gradient_decent() with the whole loop on the device.

    Same inputs, same hyper-parameters and the same result as `gradient_decent` above -- render (B),
    tracking loss, backward, Adam step, update_pose, early exit on convergence -- but each iteration is
    the four native calls of `gsr_refine` (include/gsr.h) instead of ~130 torch launches, and autograd is
    not involved.  `gaussian_grads=True` keeps computing every Gaussian-parameter gradient like the
    reference does (its map tensors require grad, tools/gaussian_model.py:437-462) although nothing
    consumes them; `False` is the pose-only fast path."""

    def __init__(self, Model, image_height, image_width, device="cuda:0", gaussian_grads=True):
        import ctypes as C
        from . import _lib
        from .rasterizer import _Workspace, _f32c
        self._C, self._lib_mod = C, _lib
        self.lib = _lib.load()
        self.dev = torch.device(device)
        self.H, self.W = int(image_height), int(image_width)
        self.model = Model
        dev = self.dev
        self.means3D = _f32c(Model.get_xyz.detach())
        self.shs = _f32c(Model.get_features.detach())
        self.opac = _f32c(Model.get_opacity.detach())
        sc = Model.get_scaling.detach()
        self.scales = _f32c(sc.repeat(1, 3) if sc.shape[-1] == 1 else sc)
        self.rots = _f32c(Model.get_rotation.detach())
        P, M = self.means3D.shape[0], self.shs.shape[1]
        self.P, self.M = P, M
        e = lambda *s, dt=torch.float32: torch.empty(s, dtype=dt, device=dev)
        H, W = self.H, self.W
        self.color, self.depth, self.alpha = e(3, H, W), e(1, H, W), e(1, H, W)
        self.radii, self.n_touched = e(P, dt=torch.int32), e(P, dt=torch.int32)
        self.g_img, self.g_depth, self.g_alpha = e(3, H, W), e(1, H, W), e(1, H, W)
        self.g_m2d, self.g_conic, self.g_opac, self.g_col = e(P, 3), e(P, 4), e(P, 1), e(P, 3)
        if gaussian_grads:
            self.g_m3d, self.g_cov, self.g_sh, self.g_scale, self.g_rot = e(P, 3), e(P, 6), e(P, M, 3), e(P, 3), e(P, 4)
        else:
            self.g_m3d = self.g_cov = self.g_sh = self.g_scale = self.g_rot = None
        self.g_tau, self.loss_out = e(6), e(4)
        self.state = torch.zeros(_lib.POSE_STATE_FLOATS, dtype=torch.float32, device=dev)
        self.ws = [_Workspace(dev), _Workspace(dev), _Workspace(dev)]
        self._warm = C.c_int(0)          # gsr_refine_args.warm_state of this refiner's image workspace

    def refine(self, viewpoint, config, initial_R, initial_T, background, iters=50, lr=0.001, converged_threshold=1e-4,
               stop_on_converged=True, speculative=True, bound_margin=None, warm_start=False):
        C, _lib = self._C, self._lib_mod
        dev = self.dev
        # speculative=True: exact optimisation (include/gsr.h, gsr_refine_args.speculative): ~9x fewer binned
        # instances and ~7x fewer SH rows on S-1M-640.  Neutral for one frame at a time (one more host sync per
        # iteration), +14 % with 4 frames in flight per GPU (median 2075 vs 1812 it/s on MI355X).
        viewpoint.update_RT(initial_R, initial_T)
        st = torch.zeros(_lib.POSE_STATE_FLOATS, dtype=torch.float32)
        st[0:9] = viewpoint.R.detach().float().cpu().reshape(-1)
        st[9:12] = viewpoint.T.detach().float().cpu()
        st[18] = float(viewpoint.exposure_a.detach())
        st[19] = float(viewpoint.exposure_b.detach())
        self.state.copy_(st)
        proj_raw = viewpoint.projection_matrix.detach().float().contiguous().to(dev)
        gt_image = viewpoint.original_image.detach().float().contiguous().to(dev)
        mono = bool(config["Training"]["monocular"])
        gt_depth = None
        if not mono:
            gd = viewpoint.depth
            gd = torch.from_numpy(gd) if not torch.is_tensor(gd) else gd
            gt_depth = gd.to(dtype=torch.float32, device=dev).contiguous()
        mask = viewpoint.grad_mask.to(device=dev).reshape(self.H, self.W).to(torch.uint8).contiguous()
        bg = background.detach().float().contiguous().to(dev)
        alpha_cfg = config["Training"]["alpha"] if "alpha" in config["Training"] else 0.98
        stream = torch.cuda.current_stream(dev).cuda_stream
        p = lambda t: None if t is None else t.data_ptr()
        a = _lib.RefineArgs()
        a.P, a.D, a.M = self.P, int(self.model.active_sh_degree), self.M
        a.means3D, a.shs, a.opacities, a.scales, a.rotations = map(p, (self.means3D, self.shs, self.opac, self.scales, self.rots))
        a.scale_modifier = 1.0
        # warm_start=True: frame sequences -- start speculating from the depth bounds the previous refine() of this
        # refiner left behind instead of binning the first iteration with the global sorts (still verified, still exact)
        if not warm_start:
            self._warm.value = 0
        a.warm_state = C.pointer(self._warm)
        a.width, a.height = self.W, self.H
        a.tan_fovx, a.tan_fovy = math.tan(viewpoint.FoVx * 0.5), math.tan(viewpoint.FoVy * 0.5)
        a.background, a.projmatrix_raw = p(bg), p(proj_raw)
        a.gt_image, a.gt_depth, a.grad_mask = p(gt_image), p(gt_depth), p(mask)
        a.opacity_threshold = float(config["Training"]["opacity_threshold"])
        a.depth_weight = float(1 - alpha_cfg)
        a.monocular = int(mono)
        a.pose_state = p(self.state)
        a.out_color, a.out_depth, a.out_alpha, a.radii, a.n_touched = map(p, (self.color, self.depth, self.alpha, self.radii, self.n_touched))
        a.dL_dimage, a.dL_ddepth, a.dL_dalpha = map(p, (self.g_img, self.g_depth, self.g_alpha))
        a.dL_dmean2D, a.dL_dconic, a.dL_dopacity, a.dL_dcolor = map(p, (self.g_m2d, self.g_conic, self.g_opac, self.g_col))
        a.dL_dmean3D, a.dL_dcov3D, a.dL_dsh, a.dL_dscale, a.dL_drot = map(p, (self.g_m3d, self.g_cov, self.g_sh, self.g_scale, self.g_rot))
        a.dL_dtau, a.loss_out = p(self.g_tau), p(self.loss_out)
        a.geometry_buffer, a.binning_buffer, a.image_buffer = self.ws[0].fn, self.ws[1].fn, self.ws[2].fn
        a.lr, a.converged_threshold, a.max_iters = float(lr), float(converged_threshold), int(iters)
        a.stop_on_converged = int(bool(stop_on_converged))
        a.speculative = int(bool(speculative))
        # bound_margin=None: adaptive margin of the speculative depth bounds (include/gsr.h); (mul, add) fixes it
        a.bound_margin_mul, a.bound_margin_add = (0.0, 0.0) if bound_margin is None else (float(bound_margin[0]), float(bound_margin[1]))
        stats = (C.c_int * 2)()
        a.stats_out = stats
        a.stream = stream
        n_done, conv = C.c_int(0), C.c_int(0)
        with torch.cuda.device(dev):
            _lib.check(self.lib.gsr_pose_init(p(self.state), p(proj_raw), stream))
            _lib.check(self.lib.gsr_refine(C.byref(a), C.byref(n_done), C.byref(conv)))
        self._keep = (proj_raw, gt_image, gt_depth, mask, bg)        # alive until the stream has drained
        s = self.state.cpu()
        viewpoint.update_RT(s[0:9].reshape(3, 3).clone(), s[9:12].clone())
        with torch.no_grad():
            viewpoint.exposure_a.fill_(float(s[18]))
            viewpoint.exposure_b.fill_(float(s[19]))
        self.last_info = {"fallbacks": int(stats[0]), "num_rendered": int(stats[1])}
        return viewpoint.R, viewpoint.T, {"iters": n_done.value, "converged": bool(conv.value), "loss": float(s[38]),
                                          "fallbacks": int(stats[0]), "num_rendered": int(stats[1]),
                                          "render": self.color, "depth": self.depth, "opacity": self.alpha}


def pose_errors(R_gt, t_gt, R, t):
    """(translation error [m], rotation error [deg]) as 7scenes_localize_full_dslam.py:368-377"""
    R_gt, t_gt, R, t = (np.asarray(x, np.float64) for x in (R_gt, t_gt, R, t))
    trans_error = float(np.linalg.norm(-R_gt.T @ t_gt.reshape(3, 1) + R.T @ t.reshape(3, 1)))
    cos = np.clip((np.trace(np.dot(R_gt.T, R)) - 1) / 2, -1.0, 1.0)
    return trans_error, float(np.rad2deg(np.abs(np.arccos(cos))))


TRACKING_CONFIG = {"Training": {"monocular": False, "alpha": 0.99, "opacity_threshold": 0.99, "edge_threshold": 1.1}}
