"""Seeded synthetic scenes for the rasterizer hot path (numpy only).

The scene definitions and the exact draw order are those of SURVEY.md section 8(d)
(S-1M-640, S-800k-chess, S-3M-cam, S-50k-fern); there is no dataset in the
reference tree or in this image, so every test / bench input comes from here.

Camera conventions follow the reference callers:
  viewmatrix = (W2C)^T            gaussian_splatting/scene/cameras.py:53 (world_view_transform)
  projmatrix = viewmatrix @ P^T   gaussian_splatting/scene/cameras.py:55 (full_proj_transform)
  campos     = inverse(viewmatrix)[3, :3]
"""
from dataclasses import dataclass, field
import math
import numpy as np


@dataclass
class Scene:
    name: str
    W: int
    H: int
    fx: float
    fy: float
    cx: float
    cy: float
    znear: float
    zfar: float
    sh_degree: int
    means3D: np.ndarray          # [P,3] f32
    scales: np.ndarray           # [P,3] f32 (post-exp)
    rotations: np.ndarray        # [P,4] f32 (w,x,y,z normalised)
    opacities: np.ndarray        # [P,1] f32 (post-sigmoid)
    shs: np.ndarray              # [P,M,3] f32
    bg: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))

    @property
    def P(self):
        return self.means3D.shape[0]

    @property
    def tanfovx(self):
        return self.W / (2.0 * self.fx)

    @property
    def tanfovy(self):
        return self.H / (2.0 * self.fy)


def projection_matrix(znear, zfar, fx, fy, cx, cy, W, H):
    """Intrinsics-only projection P (not transposed), same entries as
    gs_localization/pipelines/tools/graphics_utils.py:77-98 (getProjectionMatrix2)."""
    left = ((2 * cx - W) / W - 1.0) * W / 2.0
    right = ((2 * cx - W) / W + 1.0) * W / 2.0
    top = ((2 * cy - H) / H + 1.0) * H / 2.0
    bottom = ((2 * cy - H) / H - 1.0) * H / 2.0
    left = znear / fx * left
    right = znear / fx * right
    top = znear / fy * top
    bottom = znear / fy * bottom
    P = np.zeros((4, 4), np.float64)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera_matrices(scene, w2c=None):
    """Returns (viewmatrix, projmatrix, projmatrix_raw, campos) as f32 arrays."""
    if w2c is None:
        w2c = np.eye(4)
    w2c = np.asarray(w2c, np.float64)
    P = projection_matrix(scene.znear, scene.zfar, scene.fx, scene.fy, scene.cx, scene.cy, scene.W, scene.H)
    view = w2c.T
    proj_raw = P.T
    proj = view @ proj_raw
    campos = np.linalg.inv(view)[3, :3]
    f = lambda a: np.ascontiguousarray(a, np.float32)
    return f(view), f(proj), f(proj_raw), f(campos)


def _draw(name, P, W, H, fx, fy, zlo, zhi, scale_med, scale_sigma, sh_degree, seed):
    rng = np.random.default_rng(seed)
    tanx = W / (2.0 * fx)
    tany = H / (2.0 * fy)
    z = rng.uniform(zlo, zhi, P)
    x = rng.uniform(-1, 1, P) * 1.2 * tanx * z
    y = rng.uniform(-1, 1, P) * 1.2 * tany * z
    scale = np.exp(rng.normal(math.log(scale_med), scale_sigma, (P, 3)))
    q = rng.normal(0, 1, (P, 4))
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    opacity = 1.0 / (1.0 + np.exp(-rng.normal(0, 2, P)))
    M = (sh_degree + 1) ** 2
    amp = np.array([1.0] + [0.1] * 15)[:M]
    sh = rng.normal(0, 1, (P, M, 3)) * amp[None, :, None]
    f = lambda a: np.ascontiguousarray(a, np.float32)
    return Scene(name=name, W=W, H=H, fx=fx, fy=fy, cx=W / 2.0, cy=H / 2.0, znear=0.01, zfar=100.0,
                 sh_degree=sh_degree, means3D=f(np.stack([x, y, z], 1)), scales=f(scale), rotations=f(q),
                 opacities=f(opacity[:, None]), shs=f(sh))


def s_1m_640(P=1_000_000, seed=0):
    """Headline scene (BASELINE.json metric): 640x480, fx=fy=525, 1 M Gaussians, SH3."""
    return _draw("S-1M-640", P, 640, 480, 525.0, 525.0, 0.5, 6.0, 0.01, 0.6, 3, seed)


def s_800k_chess(seed=0):
    return _draw("S-800k-chess", 800_000, 640, 480, 525.0, 525.0, 0.5, 6.0, 0.01, 0.6, 3, seed)


def s_3m_cam(seed=0):
    return _draw("S-3M-cam", 3_000_000, 852, 480, 744.0, 744.0, 2.0, 60.0, 0.05, 0.7, 3, seed)


def s_3m_cam_1024(seed=0):
    """S-3M-cam at the image size the reference's Cambridge script really uses (1024x576 masks,
    gs_localization/pipelines/cambridge_localize_full.py:366): same field of view, 64 x 36 = 2 304 tiles."""
    f = 744.0 * 1024 / 852
    return _draw("S-3M-cam-1024", 3_000_000, 1024, 576, f, f, 2.0, 60.0, 0.05, 0.7, 3, seed)


def s_50k_fern(seed=0):
    return _draw("S-50k-fern", 50_000, 504, 378, 400.0, 400.0, 0.5, 6.0, 0.03, 0.6, 3, seed)


def small(P=512, W=64, H=48, sh_degree=3, seed=1, scale_med=0.05, fx=None):
    """Small parity-test scene; same distributions, larger splats so that tiles fill up."""
    fx = fx if fx is not None else 0.8 * W
    return _draw(f"small-{P}-{W}x{H}", P, W, H, fx, fx, 0.5, 6.0, scale_med, 0.6, sh_degree, seed)


def se3_exp(tau):
    """tau = [rho(3), theta(3)] -> 4x4, same series as
    gs_localization/pipelines/tools/pose_utils.py:54-102 (float64 numpy)."""
    tau = np.asarray(tau, np.float64)
    rho, th = tau[:3], tau[3:]
    Wm = np.array([[0, -th[2], th[1]], [th[2], 0, -th[0]], [-th[1], th[0], 0]])
    W2 = Wm @ Wm
    a = np.linalg.norm(th)
    I = np.eye(3)
    if a < 1e-5:
        R = I + Wm + 0.5 * W2
        V = I + 0.5 * Wm + W2 / 6.0
    else:
        R = I + (math.sin(a) / a) * Wm + ((1 - math.cos(a)) / a**2) * W2
        V = I + Wm * ((1 - math.cos(a)) / a**2) + W2 * ((a - math.sin(a)) / a**3)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ rho
    return T
