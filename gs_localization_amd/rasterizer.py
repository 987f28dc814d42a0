"""Host-side mirror of the reference's autograd boundary, on top of the C ABI (include/gsr.h).

Mirrors, name for name and argument for argument:
  diff_gaussian_rasterization/__init__.py:21-42    rasterize_gaussians
  diff_gaussian_rasterization/__init__.py:44-158   _RasterizeGaussians (forward / backward)
  diff_gaussian_rasterization/__init__.py:160-172  GaussianRasterizationSettings  (12 fields)
  diff_gaussian_rasterization/__init__.py:174-223  GaussianRasterizer (nn.Module, markVisible, forward)
and the un-vendored pose variant whose only trace is its call site
  gs_localization/pipelines/tools/__init__.py:15-18,58-72,116-141
  (13-field settings with projmatrix_raw; forward(..., theta, rho) -> 5-tuple with n_touched).

PyTorch is used for device memory, streams and autograd plumbing only; all arithmetic happens in
the HIP kernels behind gsr_forward / gsr_backward.  There is no CPU path.
"""
from typing import NamedTuple
import ctypes as C
import os
import threading
import weakref

import torch
import torch.nn as nn

from . import _lib

# The hop into the C ABI: a small CPython extension (csrc/gsrcall.c, built next to the library by build.py) that takes the 34 / 40
# arguments as positional ints / floats and calls gsr_forward_packed / gsr_backward_packed -- ctypes spends 10-15 us per call on the
# same arguments.  Absent (a diagnostic library under GSR_LIB_PATH, a tree that was not built): the ctypes route below does the same.
try:
    if os.environ.get("GSR_LIB_PATH") or os.environ.get("GSR_NO_EXT"):
        raise ImportError("diagnostic library: ctypes route")
    from . import _gsrcall
    if _gsrcall.ABI_VERSION != _lib.ABI_VERSION:
        raise ImportError("_gsrcall was built against another gsr.h")
except ImportError:
    _gsrcall = None


def cpu_deep_copy_tuple(input_tuple):
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizationSettingsPose(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    projmatrix_raw: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("gs_localization_amd: tensors must live on a HIP device (cuda:N); there is no CPU rasterizer")


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t):
    """device pointer, or NULL for the reference's 'empty tensor means absent' convention"""
    return None if (t is None or t.numel() == 0) else t.data_ptr()


class _Workspace:
    """One resizable device byte buffer (the std::function<char*(size_t)> of rasterize_points.cu:27-33).  All workspaces
    share ONE C trampoline (building a ctypes callback per forward costs more host time than the forward's launches); the
    callback context is the workspace's key in a registry -- of WEAK references: a forward's workspaces die with the call
    (their tensors live on in what autograd saved), and a registry that kept them alive would keep every forward's buffers."""
    _registry = weakref.WeakValueDictionary()
    _next_key = [1]
    _lock = threading.Lock()

    def __init__(self, device):
        self.device = device
        self.t = torch.empty(0, dtype=torch.uint8, device=device)
        self.error = None
        with _Workspace._lock:
            self.key = _Workspace._next_key[0]
            _Workspace._next_key[0] += 1
            _Workspace._registry[self.key] = self
        self.fn = _WORKSPACE_TRAMPOLINE
        self.ctx = C.c_void_p(self.key)

    def _resize(self, _ctx, nbytes):
        try:
            if self.t.numel() < int(nbytes):      # a workspace that is kept (FusedRefiner) keeps its contents and its address
                self.t = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            return self.t.data_ptr()
        except Exception as ex:      # an exception cannot cross the C frame: the library sees NULL (GSR_E_ALLOC) and the
            self.error = ex          # caller re-raises the original one (raise_pending)
            return 0

    @staticmethod
    def raise_pending(*workspaces):
        for w in workspaces:
            if w.error is not None:
                ex, w.error = w.error, None
                raise ex


def _workspace_dispatch(ctx, nbytes):
    w = _Workspace._registry.get(ctx)
    return w._resize(ctx, nbytes) if w is not None else 0


_WORKSPACE_TRAMPOLINE = _lib.RESIZE_FN(_workspace_dispatch)


class _SpecCache(threading.local):
    """Per thread: the `gsr_spec_state` of each (device, stream, image size, camera) the drop-in packages have rendered -- what lets
    the fifty render() calls of a refinement (7scenes_localize_full_dslam.py:66-91) and the repeated visits of a training view
    (train.py:71-75 draws from a fixed set of cameras) skip the complete lists.  Purely a matter of speed: every speculative
    forward is verified on the device and redone if it missed (include/gsr.h).
    camera: None for the pose package (its view matrix is rebuilt every iteration: ONE state per stream and size follows the
    refinement); for package (A) the storage address of `raster_settings.viewmatrix` -- a 3DGS `Camera` builds that tensor once
    (scene/cameras.py:53), so the address names the camera; a recycled address only costs a failed guess.
    States are dropped oldest first beyond GSR_SPEC_CACHE_MB (default 2 048) megabytes of device memory per thread."""

    def __init__(self):
        self.states = {}      # key -> (SpecState, device tensor); insertion order = age
        self.bytes = 0

    def get(self, lib, dev, stream, W, H, camera=None):
        key = (dev.index, stream, W, H, camera)
        hit = self.states.pop(key, None)
        if hit is None:
            n = int(lib.gsr_spec_state_bytes(W, H))
            budget = int(os.environ.get("GSR_SPEC_CACHE_MB", "2048")) << 20
            while self.states and (self.bytes + n > budget or len(self.states) >= 256):
                _old_state, old_buf = self.states.pop(next(iter(self.states)))
                self.bytes -= int(old_buf.numel())
            buf = torch.empty(n, dtype=torch.uint8, device=dev)
            st = _lib.SpecState()
            st.device_buffer = buf.data_ptr()
            # a camera seen for the first time starts from the bounds of the view rendered last at this size (a camera path, a new
            # tensor for the same camera): a guess like any other
            for k in reversed(self.states):
                if k[:4] == key[:4]:
                    src, src_buf = self.states[k]
                    if src.valid:
                        nb = int(lib.gsr_spec_state_bounds_bytes(W, H))
                        buf[:nb].copy_(src_buf[:nb])
                        st.width, st.height, st.valid, st.parity = src.width, src.height, 1, src.parity
                    break
            hit = (st, buf)
            self.bytes += n
        self.states[key] = hit
        return hit[0]

    def clear(self):
        self.states.clear()
        self.bytes = 0


_spec_cache = _SpecCache()

_ENV = getattr(os.environ, "_data", None)      # posix: the bytes-keyed dict behind os.environ (a plain lookup instead of a raised and caught KeyError)


def _env_has(name):
    return (name.encode() in _ENV) if _ENV is not None else (name in os.environ)


def speculation_enabled(pose_package):
    """Whether a drop-in forward carries depth bounds from one call to the next (gsr_forward_speculative).
    Default: on for the pose package (its caller renders the same frame fifty times, a few millimetres apart), off for package (A).
    GSR_SPECULATION=1 / 0 forces it on / off for both.  Switched on, package (A) keeps one state per CAMERA (round 4) -- worth it
    for a static map rendered from a fixed set of cameras or along a camera path; NOT for train.py: measured on the training
    replay (16 cameras, Adam moving every opacity between two visits of a view) 13 verified against 37 missed guesses at 0.2 M
    Gaussians, 40 against 34 at 1.5 M, and a miss costs the wasted forward on top of the complete one (rasterizer forward 0.72 ms
    against 0.50 ms without)."""
    if not _env_has("GSR_SPECULATION"):
        return bool(pose_package)
    return os.environ.get("GSR_SPECULATION") != "0"


def speculation_counters(device=None):
    """(verified, missed) speculative forwards of this thread's states -- for tests and tools."""
    v = m = 0
    for key, (st, _) in _spec_cache.states.items():
        if device is None or torch.device(device).index in (None, key[0]):
            v += st.n_speculative; m += st.n_failed
    return v, m


_size_cache = {}


def _workspace_sizes(lib, P, W, H, det=False):
    """(geometry, binning, image) bytes of a forward whose lists go into per-tile bins, or None when the binning workspace cannot
    be sized up front (gsr_binning_bytes_bins == 0)"""
    key = (P, W, H, det)
    hit = _size_cache.get(key)
    if hit is None:
        b = int(lib.gsr_binning_bytes_bins(P, W, H))
        hit = (int(lib.gsr_geometry_bytes_det(P) if det else lib.gsr_geometry_bytes(P)), b, int(lib.gsr_image_bytes(W, H))) if b > 0 else False
        if len(_size_cache) > 64:
            _size_cache.clear()
        _size_cache[key] = hit
    return hit or None


class _CallBlocks(threading.local):
    """Per host thread (the forward runs on the caller's thread, the backward on autograd's worker): the argument blocks of
    gsr_forward_packed / gsr_backward_packed and the three fixed-buffer descriptors, allocated once and updated in place.
    ctypes converts and checks every argument of a call -- 34 of them cost ~100 us, more than the forward's launches; a struct
    field assignment costs a quarter of a microsecond and the call itself carries one pointer."""

    def __init__(self):
        self.fa = _lib.ForwardArgs()
        self.ba = _lib.BackwardArgs()
        self.fb = (_lib.FixedBuffer * 3)()
        self.fa_ref = C.byref(self.fa)
        self.ba_ref = C.byref(self.ba)
        self.fb_addr = [C.addressof(self.fb[k]) for k in range(3)]


_blocks = _CallBlocks()
_F32 = torch.float32


def _forward_impl(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs, want_touched):
    lib = _lib.load()
    if not means3D.is_cuda:
        _require_gpu(means3D)
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    dev = means3D.device
    P = means3D.size(0)
    H, W = int(rs.image_height), int(rs.image_width)
    # (float32 + contiguous is the usual case: checked inline, converted only when needed)
    ok = lambda t: t.dtype is _F32 and t.is_contiguous()
    if not ok(means3D): means3D = _f32c(means3D)
    if not ok(sh): sh = _f32c(sh)
    if not ok(colors_precomp): colors_precomp = _f32c(colors_precomp)
    if not ok(opacities): opacities = _f32c(opacities)
    if not ok(scales): scales = _f32c(scales)
    if not ok(rotations): rotations = _f32c(rotations)
    if not ok(cov3Ds_precomp): cov3Ds_precomp = _f32c(cov3Ds_precomp)
    bg, view, proj, campos = rs.bg, rs.viewmatrix, rs.projmatrix, rs.campos
    if not (bg.device == dev and ok(bg)): bg = _f32c(bg.to(dev))
    if not (view.device == dev and ok(view)): view = _f32c(view.to(dev))
    if not (proj.device == dev and ok(proj)): proj = _f32c(proj.to(dev))
    if not (campos.device == dev and ok(campos)): campos = _f32c(campos.to(dev))
    images = torch.empty((5, H, W), dtype=_F32, device=dev)      # one allocation: colour | depth | opacity
    color, depth, alpha = images[0:3], images[3:4], images[4:5]
    ints = torch.empty((2 if want_touched else 1, P), dtype=torch.int32, device=dev)      # radii | n_touched
    radii = ints[0]
    n_touched = ints[1] if want_touched else None
    n_sh = sh.numel()
    M = sh.size(1) if n_sh != 0 else 0
    stream = torch.cuda.current_stream(dev).cuda_stream
    blk = _blocks
    a = blk.fa
    # (package (A): the per-camera state is keyed by the CALLER's view-matrix tensor -- `view` is a fresh temporary whenever that tensor
    # needed converting (CPU / float64 / non-contiguous), whose address would name nothing; such a call goes without speculation)
    cam_key = None if want_touched else (rs.viewmatrix.data_ptr() if view is rs.viewmatrix else -1)
    state_addr = (C.addressof(_spec_cache.get(lib, dev, stream, W, H, cam_key))
                  if (P > 0 and cam_key != -1 and speculation_enabled(want_touched)) else None)
    a.state = state_addr
    geom_t = bin_t = img_t = None
    nz = P > 0
    det = _env_has("GSR_DETERMINISTIC")
    sizes = _workspace_sizes(lib, P, W, H, det) if nz else None
    if _gsrcall is not None and sizes is not None:
        # fast route: workspaces of known size, every argument a positional int / float (None = NULL)
        bufs = [torch.empty(n, dtype=torch.uint8, device=dev) for n in sizes]
        base = images.data_ptr()
        N4 = 4 * H * W
        ip = ints.data_ptr()
        rc = _gsrcall.forward(
            state_addr, P, int(rs.sh_degree), M, bg.data_ptr(), W, H, means3D.data_ptr(), sh.data_ptr() if n_sh != 0 else None,
            colors_precomp.data_ptr() if colors_precomp.numel() != 0 else None, opacities.data_ptr(),
            scales.data_ptr() if scales.numel() != 0 else None, float(rs.scale_modifier), rotations.data_ptr() if rotations.numel() != 0 else None,
            cov3Ds_precomp.data_ptr() if cov3Ds_precomp.numel() != 0 else None, view.data_ptr(), proj.data_ptr(), campos.data_ptr(),
            float(rs.tanfovx), float(rs.tanfovy), int(bool(rs.prefiltered)), base, base + 3 * N4, base + 4 * N4, ip,
            int(bool(rs.debug)) | (2 if _env_has("GSR_SH_EAGER") else 0) | (4 if det else 0), (ip + 4 * P) if want_touched else None, stream,
            bufs[0].data_ptr(), sizes[0], bufs[1].data_ptr(), sizes[1], bufs[2].data_ptr(), sizes[2])
        if rc != _lib.E_ALLOC:
            num_rendered = rc if rc >= 0 else _lib.check(rc)
            geom_t, bin_t, img_t = bufs
            saved = (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom_t, bin_t, img_t, alpha, opacities)
            return num_rendered, color, radii, depth, alpha, n_touched, saved, (bg, view, proj, campos, det)
    a.P, a.D, a.M = P, int(rs.sh_degree), M
    a.background = bg.data_ptr()
    a.width, a.height = W, H
    a.means3D = means3D.data_ptr() if nz else None
    a.shs = sh.data_ptr() if n_sh != 0 else None
    a.colors_precomp = colors_precomp.data_ptr() if colors_precomp.numel() != 0 else None
    a.opacities = opacities.data_ptr() if nz else None
    a.scales = scales.data_ptr() if scales.numel() != 0 else None
    a.scale_modifier = float(rs.scale_modifier)
    a.rotations = rotations.data_ptr() if rotations.numel() != 0 else None
    a.cov3D_precomp = cov3Ds_precomp.data_ptr() if cov3Ds_precomp.numel() != 0 else None
    a.viewmatrix, a.projmatrix, a.cam_pos = view.data_ptr(), proj.data_ptr(), campos.data_ptr()
    a.tan_fovx, a.tan_fovy = float(rs.tanfovx), float(rs.tanfovy)
    a.prefiltered = int(bool(rs.prefiltered))
    base = images.data_ptr()
    N4 = 4 * H * W
    a.out_color, a.out_depth, a.out_alpha = base, base + 3 * N4, base + 4 * N4
    a.radii = radii.data_ptr() if nz else None
    a.debug = int(bool(rs.debug)) | (2 if _env_has("GSR_SH_EAGER") else 0) | (4 if det else 0)
    a.n_touched = (ints.data_ptr() + 4 * P) if (want_touched and nz) else None
    a.stream = stream
    # (the library selects the device that owns means3D itself; torch's current device only matters for the allocations above,
    # which name theirs explicitly)

    # The three workspaces.  Where their sizes are known before the call (per-tile bins: every image up to 2 048 tiles) they are
    # allocated here and handed over through the library's own fixed-buffer callback -- no callback into the interpreter; the
    # growing `_Workspace` route is the fallback (large images: the key array is sized from a device read-back, as the
    # reference's is; a bin that overflowed).
    rc = None
    if sizes is not None:
        bufs = [torch.empty(n, dtype=torch.uint8, device=dev) for n in sizes]
        fb = blk.fb
        for k in range(3):
            fb[k].ptr, fb[k].capacity = bufs[k].data_ptr(), sizes[k]
        ff = _lib.fixed_buffer_fn()
        a.geometry_buffer = a.binning_buffer = a.image_buffer = ff
        a.geometry_ctx, a.binning_ctx, a.image_ctx = blk.fb_addr
        rc = lib.gsr_forward_packed(blk.fa_ref)
        if rc == _lib.E_ALLOC:
            rc = None
        else:
            geom_t, bin_t, img_t = bufs
    if rc is None:
        geom, binning, img = _Workspace(dev), _Workspace(dev), _Workspace(dev)
        a.geometry_buffer, a.binning_buffer, a.image_buffer = geom.fn, binning.fn, img.fn
        a.geometry_ctx, a.binning_ctx, a.image_ctx = geom.ctx, binning.ctx, img.ctx
        rc = lib.gsr_forward_packed(blk.fa_ref)
        _Workspace.raise_pending(geom, binning, img)          # e.g. torch's out-of-memory error, not a bare GSR_E_ALLOC
        geom_t, bin_t, img_t = geom.t, binning.t, img.t
    num_rendered = rc if rc >= 0 else _lib.check(rc)
    saved = (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom_t, bin_t, img_t, alpha,
             opacities)
    consts = (bg, view, proj, campos, det)      # (det: this forward's geometry workspace has room for the deterministic backward's records)
    return num_rendered, color, radii, depth, alpha, n_touched, saved, consts


def _backward_impl(rs, num_rendered, saved, consts, grad_color, grad_depth, grad_alpha, pose_mode, need):
    lib = _lib.load()
    colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer, alpha, _ = saved
    bg, view, proj, campos, det = consts
    dev = means3D.device
    P = means3D.size(0)
    n_sh = sh.numel()
    M = sh.size(1) if n_sh != 0 else 0
    H, W = int(rs.image_height), int(rs.image_width)
    ok = lambda t: t.dtype is _F32 and t.is_contiguous()
    if not ok(grad_color): grad_color = _f32c(grad_color)
    if not ok(grad_depth): grad_depth = _f32c(grad_depth)
    if not ok(grad_alpha): grad_alpha = _f32c(grad_alpha)
    # All gradient tensors are slices of ONE allocation, the 16-byte-aligned ones first: the library then zero-fills the whole
    # block with a single memset instead of ten (host time is what the reference-style loop is short of).
    want_sh = bool(need["sh"] and M > 0)
    want_rot, want_scale = bool(need["rotations"]), bool(need["scales"])
    # floats per Gaussian, in block order: conic 4 | rot 4 | sh 3M | m2d 3 | m3d 3 | cov 6 | col 3 | scale 3 | opac 1   (+ 8 for tau)
    widths = (4, 4 if want_rot else 0, 3 * M if want_sh else 0, 3, 3, 6, 3, 3 if want_scale else 0, 1)
    total = P * sum(widths) + (8 if pose_mode else 0)
    flat = torch.empty((total,), dtype=_F32, device=dev)
    parts = flat.split([P * w for w in widths] + ([8] if pose_mode else []))
    dL_dconic = parts[0].view(P, 2, 2)
    dL_drotations = parts[1].view(P, 4) if want_rot else None
    dL_dsh = parts[2].view(P, M, 3) if want_sh else None
    dL_dmeans2D, dL_dmeans3D = parts[3].view(P, 3), parts[4].view(P, 3)
    dL_dcov3D, dL_dcolors = parts[5].view(P, 6), parts[6].view(P, 3)
    dL_dscales = parts[7].view(P, 3) if want_scale else None
    dL_dopacity = parts[8].view(P, 1)
    dL_dtau = parts[9][:6] if pose_mode else None
    if _gsrcall is not None and P > 0:
        f0 = flat.data_ptr()
        ptrs, off = [], 0
        for w in widths:
            ptrs.append((f0 + 4 * off) if w else None)
            off += P * w
        dconic, drot, dsh, dm2, dm3, dcov, dcol, dscale, dopac = ptrs
        rc = _gsrcall.backward(
            P, int(rs.sh_degree), M, int(num_rendered), bg.data_ptr(), W, H, means3D.data_ptr(), sh.data_ptr() if n_sh != 0 else None,
            colors_precomp.data_ptr() if colors_precomp.numel() != 0 else None, alpha.data_ptr(),
            scales.data_ptr() if scales.numel() != 0 else None, float(rs.scale_modifier), rotations.data_ptr() if rotations.numel() != 0 else None,
            cov3Ds_precomp.data_ptr() if cov3Ds_precomp.numel() != 0 else None, view.data_ptr(), proj.data_ptr(), campos.data_ptr(),
            float(rs.tanfovx), float(rs.tanfovy), radii.data_ptr(), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imgBuffer),
            grad_color.data_ptr(), grad_depth.data_ptr(), grad_alpha.data_ptr(), dm2, dconic, dopac, dcol, dm3, dcov, dsh, dscale, drot,
            int(bool(rs.debug)) | (4 if det else 0), 1 if pose_mode else 0, (f0 + 4 * off) if pose_mode else None,
            torch.cuda.current_stream(dev).cuda_stream)
        if rc < 0:
            _lib.check(rc)
        if dL_dsh is None and need["sh"]:
            dL_dsh = torch.empty((P, M, 3), dtype=_F32, device=dev)
        return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dtau
    b = _blocks.ba
    b.P, b.D, b.M, b.R = P, int(rs.sh_degree), M, int(num_rendered)
    b.background = bg.data_ptr()
    b.width, b.height = W, H
    nz = P > 0
    b.means3D = means3D.data_ptr() if nz else None
    b.shs = sh.data_ptr() if n_sh != 0 else None
    b.colors_precomp = colors_precomp.data_ptr() if colors_precomp.numel() != 0 else None
    b.alphas = alpha.data_ptr()
    b.scales = scales.data_ptr() if scales.numel() != 0 else None
    b.scale_modifier = float(rs.scale_modifier)
    b.rotations = rotations.data_ptr() if rotations.numel() != 0 else None
    b.cov3D_precomp = cov3Ds_precomp.data_ptr() if cov3Ds_precomp.numel() != 0 else None
    b.viewmatrix, b.projmatrix, b.campos = view.data_ptr(), proj.data_ptr(), campos.data_ptr()
    b.tan_fovx, b.tan_fovy = float(rs.tanfovx), float(rs.tanfovy)
    b.radii = radii.data_ptr() if nz else None
    b.geom_buffer, b.binning_buffer, b.img_buffer = _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imgBuffer)
    b.dL_dpix, b.dL_ddepths, b.dL_dalphas = grad_color.data_ptr(), grad_depth.data_ptr(), grad_alpha.data_ptr()
    f0 = flat.data_ptr() if total else 0
    offs, off = [], 0
    for w in widths:
        offs.append((f0 + 4 * off) if (w and nz) else None)
        off += P * w
    (b.dL_dconic, b.dL_drot, b.dL_dsh, b.dL_dmean2D, b.dL_dmean3D, b.dL_dcov3D, b.dL_dcolor, b.dL_dscale, b.dL_dopacity) = offs
    # (deterministic sums only if the FORWARD already ran with the option: it sized the accumulator records)
    b.debug = int(bool(rs.debug)) | (4 if det else 0)
    b.pose_mode = 1 if pose_mode else 0
    b.dL_dtau = (f0 + 4 * off) if pose_mode else None
    b.stream = torch.cuda.current_stream(dev).cuda_stream
    rc = lib.gsr_backward_packed(_blocks.ba_ref)
    if rc < 0:
        _lib.check(rc)
    if dL_dsh is None and need["sh"]:
        dL_dsh = torch.empty((P, M, 3), dtype=_F32, device=dev)
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dtau


def _debug_guard(rs, args, fn, dump_name, what):
    """reference behaviour: in debug mode dump the arguments on failure, then re-raise
    (diff_gaussian_rasterization/__init__.py:83-90,135-142)"""
    if not rs.debug:
        return fn()
    cpu_args = cpu_deep_copy_tuple(args)
    try:
        return fn()
    except Exception as ex:
        torch.save(cpu_args, dump_name)
        print(f"\nAn error occured in {what}. Please forward {dump_name} for debugging.")
        raise ex


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        args = (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
        num_rendered, color, radii, depth, alpha, _, saved, consts = _debug_guard(
            raster_settings, args, lambda: _forward_impl(*args, raster_settings, False), "snapshot_fw.dump", "forward")
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.consts = consts
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        rs = ctx.raster_settings
        nig = ctx.needs_input_grad
        need = dict(sh=nig[2], scales=nig[5], rotations=nig[6])
        saved = ctx.saved_tensors
        fn = lambda: _backward_impl(rs, ctx.num_rendered, saved, ctx.consts, grad_color, grad_depth, grad_alpha, False, need)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations, _) = _debug_guard(rs, saved + (grad_color, grad_depth, grad_alpha), fn, "snapshot_bw.dump", "backward")
        if saved[11].dim() == 1:
            grad_opacities = grad_opacities.reshape(-1)
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, None)


class _RasterizeGaussiansPose(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho,
                raster_settings):
        args = (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
        num_rendered, color, radii, depth, alpha, n_touched, saved, consts = _debug_guard(
            raster_settings, args, lambda: _forward_impl(*args, raster_settings, True), "snapshot_fw.dump", "forward")
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.consts = consts
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(radii, n_touched)
        return color, radii, depth, alpha, n_touched

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha, grad_n_touched):
        rs = ctx.raster_settings
        nig = ctx.needs_input_grad
        need = dict(sh=nig[2], scales=nig[5], rotations=nig[6])
        saved = ctx.saved_tensors
        fn = lambda: _backward_impl(rs, ctx.num_rendered, saved, ctx.consts, grad_color, grad_depth, grad_alpha, True, need)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations, grad_tau) = _debug_guard(rs, saved + (grad_color, grad_depth, grad_alpha), fn, "snapshot_bw.dump", "backward")
        if saved[11].dim() == 1:
            grad_opacities = grad_opacities.reshape(-1)
        grad_rho = grad_tau[:3]
        grad_theta = grad_tau[3:]
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, grad_theta, grad_rho, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings)


def rasterize_gaussians_pose(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta,
                             rho, raster_settings):
    return _RasterizeGaussiansPose.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                         cov3Ds_precomp, theta, rho, raster_settings)


def _mark_visible(positions, raster_settings):
    lib = _lib.load()
    _require_gpu(positions)
    positions = _f32c(positions)
    dev = positions.device
    P = positions.size(0)
    present = torch.empty((P,), dtype=torch.uint8, device=dev)
    view = _f32c(raster_settings.viewmatrix.to(dev))
    proj = _f32c(raster_settings.projmatrix.to(dev))
    with torch.cuda.device(dev):
        _lib.check(lib.gsr_mark_visible(P, _ptr(positions), _ptr(view), _ptr(proj), _ptr(present),
                                        torch.cuda.current_stream(dev).cuda_stream))
    return present.bool()


def _check_inputs(shs, colors_precomp, scales, rotations, cov3D_precomp):
    if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    if ((scales is None or rotations is None) and cov3D_precomp is None) or (
            (scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')


_EMPTY = torch.Tensor([])      # the reference's "absent" convention (an empty tensor); one shared instance instead of five per call


def _empty_if_none(*ts):
    return tuple(_EMPTY if t is None else t for t in ts)


class GaussianRasterizer(nn.Module):
    """Package (A): forward(...) -> (color, radii, depth, alpha)."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            return _mark_visible(positions, self.raster_settings)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        _check_inputs(shs, colors_precomp, scales, rotations, cov3D_precomp)
        shs, colors_precomp, scales, rotations, cov3D_precomp = _empty_if_none(shs, colors_precomp, scales, rotations,
                                                                               cov3D_precomp)
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)


class GaussianRasterizerPose(nn.Module):
    """Package (B): forward(..., theta, rho) -> (color, radii, depth, opacity, n_touched)."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            return _mark_visible(positions, self.raster_settings)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, theta=None, rho=None):
        _check_inputs(shs, colors_precomp, scales, rotations, cov3D_precomp)
        shs, colors_precomp, scales, rotations, cov3D_precomp = _empty_if_none(shs, colors_precomp, scales, rotations,
                                                                               cov3D_precomp)
        if theta is None:
            theta = torch.zeros(3, dtype=torch.float32, device=means3D.device)
        if rho is None:
            rho = torch.zeros(3, dtype=torch.float32, device=means3D.device)
        return rasterize_gaussians_pose(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                        cov3D_precomp, theta, rho, self.raster_settings)
