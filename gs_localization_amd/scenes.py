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


# ---------------------------------------------------------------------------------------------
# Structured variants of the headline scene (round 5).  Every scene of SURVEY.md 8(d) is a uniform random cloud; a trained
# indoor map is surfaces and objects.  These keep S-1M-640's camera, size and per-splat distributions and change only WHERE the
# splats are (and, for the room, their shape), so that their rates read against the headline's.
# ---------------------------------------------------------------------------------------------
def s_1m_640_object(P=1_000_000, seed=0, vseed=11):
    """S-1M-640 with half of the map inside a cone around the optical axis a tenth of the image wide: ~60 of the 1 200 tiles
    carry most of the compositing work (a dense object in front of a sparse scene)."""
    sc = s_1m_640(P, seed)
    rng = np.random.default_rng(vseed)
    m = sc.means3D.astype(np.float64)
    sel = rng.random(P) < 0.5
    z = m[sel, 2]
    n = int(sel.sum())
    m[sel, 0] = rng.normal(0, 0.03, n) * z
    m[sel, 1] = rng.normal(0, 0.03, n) * z
    sc.means3D = np.ascontiguousarray(m, np.float32)
    sc.name = "S-1M-640-object"
    return sc


def s_1m_640_walls(P=1_000_000, seed=0, vseed=11):
    """S-1M-640 with every splat on one of two thin fronto-parallel layers (z = 3 m and 5 m, 2 cm thick): a depth bound per
    tile cannot tell coplanar splats apart."""
    sc = s_1m_640(P, seed)
    rng = np.random.default_rng(vseed)
    m = sc.means3D.astype(np.float64)
    lay = rng.random(P) < 0.5
    zz = np.where(lay, 3.0, 5.0) + rng.normal(0, 0.02, P)
    m[:, 0] *= zz / m[:, 2]
    m[:, 1] *= zz / m[:, 2]
    m[:, 2] = zz
    sc.means3D = np.ascontiguousarray(m, np.float32)
    sc.name = "S-1M-640-walls"
    return sc


def _quat_from_frame(t1, t2, n):
    """(w, x, y, z) of the rotation whose columns are (t1, t2, n), row-wise for arrays [K, 3]."""
    R = np.stack([t1, t2, n], axis=2)                       # [K, 3, 3], columns = local axes
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    q = np.empty((R.shape[0], 4))
    # numerically safe branch per row (largest of w, x, y, z)
    cand = np.stack([tr, R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]], 1)
    which = cand.argmax(1)
    for k in range(4):
        s_ = which == k
        if not s_.any():
            continue
        r = R[s_]
        if k == 0:
            s4 = np.sqrt(1.0 + r[:, 0, 0] + r[:, 1, 1] + r[:, 2, 2]) * 2
            q[s_] = np.stack([0.25 * s4, (r[:, 2, 1] - r[:, 1, 2]) / s4, (r[:, 0, 2] - r[:, 2, 0]) / s4, (r[:, 1, 0] - r[:, 0, 1]) / s4], 1)
        elif k == 1:
            s4 = np.sqrt(1.0 + r[:, 0, 0] - r[:, 1, 1] - r[:, 2, 2]) * 2
            q[s_] = np.stack([(r[:, 2, 1] - r[:, 1, 2]) / s4, 0.25 * s4, (r[:, 0, 1] + r[:, 1, 0]) / s4, (r[:, 0, 2] + r[:, 2, 0]) / s4], 1)
        elif k == 2:
            s4 = np.sqrt(1.0 - r[:, 0, 0] + r[:, 1, 1] - r[:, 2, 2]) * 2
            q[s_] = np.stack([(r[:, 0, 2] - r[:, 2, 0]) / s4, (r[:, 0, 1] + r[:, 1, 0]) / s4, 0.25 * s4, (r[:, 1, 2] + r[:, 2, 1]) / s4], 1)
        else:
            s4 = np.sqrt(1.0 - r[:, 0, 0] - r[:, 1, 1] + r[:, 2, 2]) * 2
            q[s_] = np.stack([(r[:, 1, 0] - r[:, 0, 1]) / s4, (r[:, 0, 2] + r[:, 2, 0]) / s4, (r[:, 1, 2] + r[:, 2, 1]) / s4, 0.25 * s4], 1)
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def s_room_640(P=1_000_000, seed=0):
    """S-room-640: the nearest honest stand-in for a trained indoor map (7-Scenes chess) that can be drawn without data.
    A box room 6 x 3 x 8 m around the camera (identity pose, looking down +z; the wall behind the camera is part of the map and
    never visible) with six pieces of box furniture; every splat lies ON a surface, flattened (its extent along the surface
    normal is a tenth of its tangential extents, the normal is one of its principal axes, random in-plane rotation), with 5 mm of
    positional noise along the normal; opacities are bimodal (two thirds near-opaque, one third faint), as trained maps are.
    640x480, fx = fy = 525, SH degree 3, 1 M Gaussians."""
    rng = np.random.default_rng(seed)
    W, H, fx = 640, 480, 525.0
    # surfaces: (origin, edge u, edge v, normal) -- rectangles; sampled in proportion to their area
    rects = []

    def box(lo, hi, inward):
        lo, hi = np.asarray(lo, float), np.asarray(hi, float)
        d = hi - lo
        sgn = -1.0 if inward else 1.0
        for ax in range(3):
            u, v = (ax + 1) % 3, (ax + 2) % 3
            eu, ev = np.zeros(3), np.zeros(3)
            eu[u], ev[v] = d[u], d[v]
            for side, o in ((0, lo), (1, lo + np.eye(3)[ax] * d[ax])):
                n = np.zeros(3)
                n[ax] = (1.0 if side else -1.0) * sgn
                rects.append((o.copy(), eu, ev, n))
    box((-3.0, -1.5, -2.0), (3.0, 1.5, 6.0), inward=True)                    # the room
    furniture = [((-2.6, 0.3, 2.2), (-1.2, 1.5, 3.6)), ((0.9, 0.6, 1.6), (2.3, 1.5, 2.4)), ((-0.6, 0.9, 3.0), (0.6, 1.5, 4.2)),
                 ((1.6, -0.4, 4.4), (2.9, 1.5, 5.6)), ((-2.9, -0.9, 4.8), (-1.9, 1.5, 5.9)), ((-0.4, 0.2, 1.2), (0.2, 0.9, 1.5))]
    n_room = len(rects)
    for lo, hi in furniture:
        box(lo, hi, inward=False)
    area = np.array([np.linalg.norm(np.cross(eu, ev)) for _, eu, ev, _ in rects])
    area[n_room:] *= 3.0                                                       # objects are mapped more densely than bare walls
    which = rng.choice(len(rects), size=P, p=area / area.sum())
    a, b = rng.random(P), rng.random(P)
    O = np.stack([r[0] for r in rects])[which]
    EU = np.stack([r[1] for r in rects])[which]
    EV = np.stack([r[2] for r in rects])[which]
    Nn = np.stack([r[3] for r in rects])[which]
    pos = O + a[:, None] * EU + b[:, None] * EV + rng.normal(0, 0.005, P)[:, None] * Nn
    t1 = EU / np.linalg.norm(EU, axis=1, keepdims=True)
    t2 = np.cross(Nn, t1)
    phi = rng.uniform(0, 2 * np.pi, P)
    c, s_ = np.cos(phi)[:, None], np.sin(phi)[:, None]
    u1, u2 = c * t1 + s_ * t2, -s_ * t1 + c * t2                               # in-plane axes; (u1, u2, n) is right-handed
    q = _quat_from_frame(u1, u2, Nn)
    tang = np.exp(rng.normal(math.log(0.02), 0.5, (P, 2)))
    scale = np.concatenate([tang, 0.1 * tang.mean(1, keepdims=True)], 1)
    strong = rng.random(P) < (2.0 / 3.0)
    logit = np.where(strong, rng.normal(3.0, 1.0, P), rng.normal(-2.5, 1.0, P))
    opacity = 1.0 / (1.0 + np.exp(-logit))
    amp = np.array([1.0] + [0.1] * 15)
    sh = rng.normal(0, 1, (P, 16, 3)) * amp[None, :, None]
    f = lambda x: np.ascontiguousarray(x, np.float32)
    return Scene(name="S-room-640", W=W, H=H, fx=fx, fy=fx, cx=W / 2.0, cy=H / 2.0, znear=0.01, zfar=100.0, sh_degree=3,
                 means3D=f(pos), scales=f(scale), rotations=f(q), opacities=f(opacity[:, None]), shs=f(sh))


VARIANTS = {"object": s_1m_640_object, "walls": s_1m_640_walls, "room": s_room_640}


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
