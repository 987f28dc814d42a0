"""EXPERIMENT: host time of one forward / backward through the pose package on a scene with negligible GPU work, the CPython hop
(_gsrcall) against the ctypes route, alternating in ONE process (rasterizer._gsrcall switched on and off)."""
import torch, time, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gs_localization_amd import scenes as S, rasterizer as RZ
from tests import replay as PL
dev = torch.device("cuda:0")
bg = torch.zeros(3, device=dev)
tiny = S.small(P=2000, W=64, H=48, sh_degree=3, seed=3, scale_med=0.06)
tm = PL.GaussianMap.from_scene(tiny, device=dev)
tvp = PL.make_frame(tiny, tm, dev, bg)
from diff_gaussian_rasterization_pose import GaussianRasterizationSettings as _RS, GaussianRasterizer as _RZ
rs = _RS(image_height=48, image_width=64, tanfovx=math.tan(0.5 * tvp.FoVx), tanfovy=math.tan(0.5 * tvp.FoVy), bg=bg, scale_modifier=1.0, viewmatrix=tvp.world_view_transform,
         projmatrix=tvp.full_proj_transform, projmatrix_raw=tvp.projection_matrix, sh_degree=3, campos=tvp.camera_center, prefiltered=False, debug=False)
rz = _RZ(raster_settings=rs)
m2d = torch.zeros_like(tm.get_xyz, requires_grad=True)
gi, gd = torch.ones((3, 48, 64), device=dev), torch.ones((1, 48, 64), device=dev)
ext = RZ._gsrcall
def run(n):
    fw = bw = 0.0
    for it_ in range(n + 10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o_ = rz(means3D=tm.get_xyz, means2D=m2d, opacities=tm.get_opacity, shs=tm.get_features, colors_precomp=None, scales=tm.get_scaling, rotations=tm.get_rotation,
                cov3D_precomp=None, theta=tvp.cam_rot_delta, rho=tvp.cam_trans_delta)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        torch.autograd.backward([o_[0], o_[2]], [gi, gd])
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if it_ >= 10: fw += t1 - t0; bw += t2 - t1
    return 1e6 * fw / n, 1e6 * bw / n
res = {"ext": [], "ctypes": []}
for rep in range(6):
    for name, mod in (("ext", ext), ("ctypes", None)):
        RZ._gsrcall = mod
        res[name].append(run(200))
for k, v in res.items():
    print(k, "forward us", [round(x[0], 1) for x in v], "backward us", [round(x[1], 1) for x in v])
