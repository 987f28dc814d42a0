"""ctypes binding of oracle/gs_oracle.c -- TEST INFRASTRUCTURE ONLY.

May be imported only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package gs_localization_amd never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgs_oracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("gs_oracle.c", "gs_oracle_k7.inc")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.gso_forward.restype = C.c_void_p
        _lib.gso_r_eff.restype = C.c_long
        _lib.gso_r_eff.argtypes = [C.c_void_p]
        _lib.gso_num_rendered.argtypes = [C.c_void_p]
        _lib.gso_free.argtypes = [C.c_void_p]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, np.float32)


def set_threads(n):
    lib().gso_set_threads(int(n))


def set_accumulate_double(on):
    """Checker option: the compositing backward sums its per-Gaussian atomics in double and rounds once (see gs_oracle.c)."""
    lib().gso_set_accumulate_double(int(bool(on)))


def set_backward_double(on):
    """Checker option: the compositing backward's recurrences and sums in double, on the forward's fp32 decisions
    (gs_oracle_k7.inc) -- the yardstick on lists of many hundreds of entries."""
    lib().gso_set_backward_double(int(bool(on)))


def backward_with_conditioning(fwd, grad_color, grad_depth, grad_alpha, pose_mode=False, eps=1.6e-6):
    """backward() + the conditioning report of gso_set_condition_out: (grads, mass [P], rounding_reach [P]) -- per Gaussian the
    sum of |terms| of its opacity gradient and how far the rounding of the transmittances alone (eps = relative error of one alpha:
    four times 4e-7, as in flip_audit) can move that sum."""
    out = np.zeros((fwd.P, 2), np.float64)
    L = lib()
    L.gso_set_condition_out.argtypes = [C.c_void_p, C.c_double]
    L.gso_set_condition_out(_p(out), C.c_double(eps))
    try:
        g = backward(fwd, grad_color, grad_depth, grad_alpha, pose_mode=pose_mode)
    finally:
        L.gso_set_condition_out(None, C.c_double(0.0))
    return g, out[:, 0].copy(), out[:, 1].copy()


def flip_audit(f, tol=4.0, live=None, row_threshold=2.5e-4, weights=False):
    """gso_flip_audit on a Forward: (near_half int32 [P], unstable bool [P], counts dict).  live: optional [H, W] bool -- only
    pixels with a gradient enter a gradient row.  unstable = the share of the row's pixels-worth of contributions that a flipped
    threshold decision can move reaches `row_threshold` (a quarter of the 1e-3 at which the parity tests start counting a row:
    pixels do not weigh the same)."""
    P = f.P
    near_half = np.zeros(P, np.int32)
    w_evt = np.zeros(P, np.float32)
    w_all = np.zeros(P, np.int32)
    counts = np.zeros(4, np.int64)
    lv = None if live is None else np.ascontiguousarray(np.asarray(live).reshape(f.H, f.W), np.uint8)
    lib().gso_flip_audit(C.c_void_p(f._state), C.c_float(tol), _p(lv), _p(near_half), _p(w_evt), _p(w_all), _p(counts))
    unstable = w_evt >= row_threshold * np.maximum(w_all, 1)
    unstable &= w_evt > 0
    ev = dict(alpha_events=int(counts[0]), termination_events=int(counts[1]), half_events=int(counts[2]), pixels_with_an_event=int(counts[3]))
    return (near_half, unstable, ev, w_evt, w_all) if weights else (near_half, unstable, ev)


class Forward:
    """Result of one oracle forward; owns the opaque state needed by backward()."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __del__(self):
        if getattr(self, "_state", None):
            try:
                lib().gso_free(C.c_void_p(self._state))
            except TypeError:          # interpreter shutdown: the module globals are already gone
                pass
            self._state = None

    def state(self):
        P, W, H = self.P, self.W, self.H
        gx, gy = (W + 15) // 16, (H + 15) // 16
        R = self.num_rendered
        out = dict(depths=np.zeros(P, np.float32), means2D=np.zeros((P, 2), np.float32),
                   cov3D=np.zeros((P, 6), np.float32), conic_opacity=np.zeros((P, 4), np.float32),
                   rgb=np.zeros((P, 3), np.float32), clamped=np.zeros((P, 3), np.uint8),
                   tiles_touched=np.zeros(P, np.uint32), point_list=np.zeros(max(R, 1), np.uint32),
                   ranges=np.zeros((gx * gy, 2), np.uint32), n_contrib=np.zeros((H, W), np.uint32))
        lib().gso_get_state(C.c_void_p(self._state), *[_p(out[k]) for k in (
            "depths", "means2D", "cov3D", "conic_opacity", "rgb", "clamped", "tiles_touched", "point_list",
            "ranges", "n_contrib")])
        out["point_list"] = out["point_list"][:R]
        return out


def forward(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, bg, sh_degree=0,
            shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
            scale_modifier=1.0, want_n_touched=False):
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    shs, colors_precomp, scales, rotations, cov3D_precomp = map(_f32, (shs, colors_precomp, scales, rotations, cov3D_precomp))
    opacities = _f32(opacities).reshape(-1)
    viewmatrix, projmatrix, campos, bg = map(_f32, (viewmatrix, projmatrix, campos, bg))
    M = shs.shape[1] if shs is not None else 0
    color = np.zeros((3, H, W), np.float32)
    depth = np.zeros((1, H, W), np.float32)
    alpha = np.zeros((1, H, W), np.float32)
    radii = np.zeros(P, np.int32)
    n_touched = np.zeros(P, np.int32) if want_n_touched else None
    st = L.gso_forward(C.c_int(P), C.c_int(sh_degree), C.c_int(M), _p(bg), C.c_int(W), C.c_int(H), _p(means3D),
                       _p(shs), _p(colors_precomp), _p(opacities), _p(scales), C.c_float(scale_modifier),
                       _p(rotations), _p(cov3D_precomp), _p(viewmatrix), _p(projmatrix), _p(campos),
                       C.c_float(tanfovx), C.c_float(tanfovy), _p(color), _p(depth), _p(alpha), _p(radii),
                       _p(n_touched))
    R = L.gso_num_rendered(C.c_void_p(st))
    return Forward(_state=st, P=P, W=W, H=H, M=M, sh_degree=sh_degree, num_rendered=R, color=color, depth=depth,
                   alpha=alpha, radii=radii, n_touched=n_touched,
                   _in=dict(means3D=means3D, shs=shs, colors_precomp=colors_precomp, scales=scales,
                            rotations=rotations, cov3D_precomp=cov3D_precomp, opacities=opacities,
                            viewmatrix=viewmatrix, projmatrix=projmatrix, campos=campos, bg=bg,
                            tanfovx=tanfovx, tanfovy=tanfovy, scale_modifier=scale_modifier))


def r_eff(fwd):
    return int(lib().gso_r_eff(C.c_void_p(fwd._state)))


def backward(fwd, grad_color, grad_depth, grad_alpha, pose_mode=False):
    """Returns dict of gradients with the shapes of diff_gaussian_rasterization/__init__.py:146-156
    (+ 'tau' = [d/drho, d/dtheta] when pose_mode)."""
    L = lib()
    i = fwd._in
    P, M = fwd.P, fwd.M
    g = dict(means2D=np.zeros((P, 3), np.float32), conic=np.zeros((P, 4), np.float32),
             opacities=np.zeros((P, 1), np.float32), colors_precomp=np.zeros((P, 3), np.float32),
             means3D=np.zeros((P, 3), np.float32), cov3Ds_precomp=np.zeros((P, 6), np.float32),
             sh=np.zeros((P, M, 3), np.float32), scales=np.zeros((P, 3), np.float32),
             rotations=np.zeros((P, 4), np.float32))
    tau = np.zeros(6, np.float32)
    gc, gd, ga = _f32(grad_color), _f32(grad_depth), _f32(grad_alpha)
    L.gso_backward(C.c_void_p(fwd._state), C.c_int(fwd.sh_degree), C.c_int(M), _p(i["bg"]), _p(i["means3D"]),
                   _p(i["shs"]), _p(i["colors_precomp"]), _p(fwd.alpha), _p(i["scales"]),
                   C.c_float(i["scale_modifier"]), _p(i["rotations"]), _p(i["cov3D_precomp"]), _p(i["viewmatrix"]),
                   _p(i["projmatrix"]), _p(i["campos"]), C.c_float(i["tanfovx"]), C.c_float(i["tanfovy"]), _p(gc),
                   _p(gd), _p(ga), _p(g["means2D"]), _p(g["conic"]), _p(g["opacities"]), _p(g["colors_precomp"]),
                   _p(g["means3D"]), _p(g["cov3Ds_precomp"]), _p(g["sh"]), _p(g["scales"]), _p(g["rotations"]),
                   C.c_int(1 if pose_mode else 0), _p(tau))
    if pose_mode:
        g["tau"] = tau
    return g


def sh_stage(means3D, campos, shs, sh_degree, dL_dcolor=None):
    """K1's SH colour stage and K9's SH backward stage on their own (tests/test_oracle_pinning.py):
    returns (rgb [P,3], clamped [P,3] bool) and, given dL_dcolor, also (dL_dsh [P,M,3], dL_dmean [P,3] = the view-direction term)."""
    L = lib()
    means3D, campos, shs = _f32(means3D), _f32(campos), _f32(shs)
    P, M = shs.shape[0], shs.shape[1]
    rgb = np.zeros((P, 3), np.float32)
    clamped = np.zeros((P, 3), np.uint8)
    L.gso_sh_forward_stage(C.c_int(P), C.c_int(sh_degree), C.c_int(M), _p(means3D), _p(campos), _p(shs), _p(rgb), _p(clamped))
    if dL_dcolor is None:
        return rgb, clamped.astype(bool)
    g = _f32(dL_dcolor)
    dsh = np.zeros((P, M, 3), np.float32)
    dmean = np.zeros((P, 3), np.float32)
    L.gso_sh_backward_stage(C.c_int(P), C.c_int(sh_degree), C.c_int(M), _p(means3D), _p(campos), _p(shs), _p(clamped), _p(g), _p(dsh), _p(dmean))
    return rgb, clamped.astype(bool), dsh, dmean


def cov3d_backward_stage(scales, rotations, dL_dcov3D, scale_modifier=1.0):
    """K9's covariance stage on its own: dL/dcov3D (6-vector) -> (dL/dscale [P,3], dL/drot [P,4])."""
    L = lib()
    scales, rotations, g = _f32(scales), _f32(rotations), _f32(dL_dcov3D)
    P = scales.shape[0]
    ds = np.zeros((P, 3), np.float32)
    dq = np.zeros((P, 4), np.float32)
    L.gso_cov3d_backward_stage(C.c_int(P), _p(scales), C.c_float(scale_modifier), _p(rotations), _p(g), _p(ds), _p(dq))
    return ds, dq


def mark_visible(means3D, viewmatrix, projmatrix):
    means3D = _f32(means3D)
    out = np.zeros(means3D.shape[0], np.uint8)
    lib().gso_mark_visible(C.c_int(means3D.shape[0]), _p(means3D), _p(_f32(viewmatrix)), _p(_f32(projmatrix)), _p(out))
    return out.astype(bool)


def forward_scene(scene, w2c=None, **kw):
    from gs_localization_amd import scenes as S
    view, proj, _, campos = S.camera_matrices(scene, w2c)
    return forward(scene.means3D, scene.opacities, view, proj, campos, scene.W, scene.H, scene.tanfovx,
                   scene.tanfovy, scene.bg, sh_degree=scene.sh_degree, shs=scene.shs, scales=scene.scales,
                   rotations=scene.rotations, **kw)
