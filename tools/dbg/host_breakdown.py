"""EXPERIMENT: where a forward / backward through the pose package spends its HOST time on a scene with negligible GPU work:
the wrapper's own steps (monkey-patched timers around torch.empty, the C call, the autograd machinery around them)."""
import torch, time, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gs_localization_amd import scenes as S, rasterizer as RZ, _lib
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
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w
RZ._forward_impl = timed("_forward_impl", RZ._forward_impl)
RZ._backward_impl = timed("_backward_impl", RZ._backward_impl)
ext_f, ext_b = RZ._gsrcall.forward, RZ._gsrcall.backward
class _Ext:
    ABI_VERSION = RZ._gsrcall.ABI_VERSION
    forward = staticmethod(timed("C forward (launches + its blocking read)", ext_f))
    backward = staticmethod(timed("C backward (launches)", ext_b))
RZ._gsrcall = _Ext
_empty = torch.empty
def empty_timed(*a, **k):
    t0 = time.perf_counter(); r = _empty(*a, **k); acc["torch.empty (all)"] = acc.get("torch.empty (all)", 0.0) + time.perf_counter() - t0; return r
torch.empty = empty_timed
N = 300
fw = bw = 0.0
for it_ in range(N + 20):
    if it_ == 20:
        acc.clear(); fw = bw = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o_ = rz(means3D=tm.get_xyz, means2D=m2d, opacities=tm.get_opacity, shs=tm.get_features, colors_precomp=None, scales=tm.get_scaling, rotations=tm.get_rotation,
            cov3D_precomp=None, theta=tvp.cam_rot_delta, rho=tvp.cam_trans_delta)
    t1a = time.perf_counter()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    torch.autograd.backward([o_[0], o_[2]], [gi, gd])
    t2a = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    fw += t1 - t0; bw += t2 - t1
    acc["forward: return -> GPU idle"] = acc.get("forward: return -> GPU idle", 0.0) + t1 - t1a
    acc["backward: return -> GPU idle"] = acc.get("backward: return -> GPU idle", 0.0) + t2 - t2a
print("forward %.1f us, backward %.1f us per call" % (1e6 * fw / N, 1e6 * bw / N))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("   %-46s %7.1f us" % (k, 1e6 * v / N))
