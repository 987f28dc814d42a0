"""Native pose refinement of one query frame: the device-resident map container and the host wrapper of `gsr_refine`.

What the reference does per frame in Python -- `gradient_decent()`,
gs_localization/pipelines/7scenes_localize_full_dslam.py:29-93: up to fifty times { render through the pose rasterizer,
tracking loss, backward, Adam step, update_pose, convergence test } -- runs here as ONE C-ABI call (include/gsr.h,
`gsr_refine`).  The caller's camera object is used as it is (duck-typed: R, T, exposure_a/b, projection_matrix, FoVx/FoVy,
original_image, depth, grad_mask, update_RT -- the attributes of tools/camera_utils.py:38-158), so a script written against the
reference's `Camera` can hand its viewpoint straight to `FusedRefiner.refine`.

The reference-style Python loop on the drop-in packages (what the unchanged scripts would execute) is test infrastructure
and lives in tests/replay.py.
"""
import math

import numpy as np
import torch


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


class FusedRefiner:
    """gradient_decent() with the whole loop on the device.

    Same inputs, same hyper-parameters and the same pose trajectory as the reference's loop -- render (B), tracking loss,
    backward, Adam step, update_pose, early exit on convergence -- but each iteration is a handful of kernels enqueued by
    `gsr_refine` (include/gsr.h) instead of ~130 torch launches, and autograd is not involved.  `gaussian_grads=True` keeps
    computing every Gaussian-parameter gradient like the reference does (its map tensors require grad,
    tools/gaussian_model.py:437-462) although nothing consumes them; `False` is the pose-only fast path.

    Aliasing: the R, T handed back (and `viewpoint.exposure_a/b.data`) are VIEWS of this call's device pose state -- the next
    refine() allocates a fresh state, so they stay valid, but they share storage with each other; clone before writing in place.

    One documented difference in what is RETURNED next to the pose: the reference hands back the render_pkg of its last
    loop body, i.e. the render at the pose BEFORE the last update (7scenes_localize_full_dslam.py:66-93).  `refine` does the
    same when max_iters is reached; on early convergence its images (render / depth / opacity, radii, n_touched) are those
    of one more forward at the FINAL pose -- the converged update is below 1e-4, so the two differ by less than that step."""

    def __init__(self, Model, image_height, image_width, device="cuda:0", gaussian_grads=True, colors_precomp=None, cov3D_precomp=None):
        """colors_precomp [P, 3] / cov3D_precomp [P, 6] (optional): the precomputed-input modes of render() (B)
        (tools/__init__.py:85-112, pipe.convert_SHs_python / pipe.compute_cov3D_python) -- used instead of the map's SH
        coefficients / scales and rotations; `g_col` / `g_cov` are then their gradients."""
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
        self.colors_pre = None if colors_precomp is None else _f32c(colors_precomp.detach().to(dev))
        self.cov_pre = None if cov3D_precomp is None else _f32c(cov3D_precomp.detach().to(dev))
        self.shs = _f32c(Model.get_features.detach()) if self.colors_pre is None else None
        self.opac = _f32c(Model.get_opacity.detach())
        sc = Model.get_scaling.detach()
        self.scales = _f32c(sc.repeat(1, 3) if sc.shape[-1] == 1 else sc) if self.cov_pre is None else None
        self.rots = _f32c(Model.get_rotation.detach()) if self.cov_pre is None else None
        P, M = self.means3D.shape[0], (self.shs.shape[1] if self.shs is not None else 0)
        self.P, self.M = P, M
        e = lambda *s, dt=torch.float32: torch.empty(s, dtype=dt, device=dev)
        H, W = self.H, self.W
        self.color, self.depth, self.alpha = e(3, H, W), e(1, H, W), e(1, H, W)
        self.radii, self.n_touched = e(P, dt=torch.int32), e(P, dt=torch.int32)
        self.g_img, self.g_depth, self.g_alpha = e(3, H, W), e(1, H, W), e(1, H, W)
        self.g_m2d, self.g_conic, self.g_opac, self.g_col = e(P, 3), e(P, 4), e(P, 1), e(P, 3)
        if gaussian_grads:
            self.g_m3d, self.g_cov = e(P, 3), e(P, 6)
            self.g_sh = e(P, M, 3) if M > 0 else None
            self.g_scale, self.g_rot = (e(P, 3), e(P, 4)) if self.cov_pre is None else (None, None)
        else:
            self.g_m3d = self.g_cov = self.g_sh = self.g_scale = self.g_rot = None
        self.g_tau, self.loss_out = e(6), e(4)
        self.state = torch.zeros(_lib.POSE_STATE_FLOATS, dtype=torch.float32, device=dev)
        self.ws = [_Workspace(dev), _Workspace(dev), _Workspace(dev)]
        self._warm = C.c_int(0)          # gsr_refine_args.warm_state of this refiner's image workspace
        # gsr_refine_args.carry_state: this refiner owns the gradient tensors, the workspaces and a frozen map, and nothing else
        # writes them between two refine() calls -- so the second call need not zero-fill 300 MB of gradients again
        self._carry = C.c_int(0)
        self._carry_versions = None      # torch's in-place-modification counters of those tensors when the last call returned
        self._conv_cache = {}
        self._state_host = (C.c_float * _lib.POSE_STATE_FLOATS)()
        self._state_host_np = np.frombuffer(self._state_host, dtype=np.float32)      # (a view: no copy per call)
        # gsr_refine_args: everything that does not change from call to call is filled in once
        p = lambda t: None if t is None else t.data_ptr()
        a = _lib.RefineArgs()
        a.P, a.M = self.P, self.M
        a.means3D, a.shs, a.opacities, a.scales, a.rotations = map(p, (self.means3D, self.shs, self.opac, self.scales, self.rots))
        a.colors_precomp, a.cov3D_precomp = p(self.colors_pre), p(self.cov_pre)
        a.width, a.height = self.W, self.H
        a.pose_state = p(self.state)
        a.pose_state_host = self._state_host
        a.out_color, a.out_depth, a.out_alpha, a.radii, a.n_touched = map(p, (self.color, self.depth, self.alpha, self.radii, self.n_touched))
        a.dL_dimage, a.dL_ddepth, a.dL_dalpha = map(p, (self.g_img, self.g_depth, self.g_alpha))
        a.dL_dmean2D, a.dL_dconic, a.dL_dopacity, a.dL_dcolor = map(p, (self.g_m2d, self.g_conic, self.g_opac, self.g_col))
        a.dL_dmean3D, a.dL_dcov3D, a.dL_dsh, a.dL_dscale, a.dL_drot = map(p, (self.g_m3d, self.g_cov, self.g_sh, self.g_scale, self.g_rot))
        a.dL_dtau, a.loss_out = p(self.g_tau), p(self.loss_out)
        a.geometry_buffer, a.binning_buffer, a.image_buffer = self.ws[0].fn, self.ws[1].fn, self.ws[2].fn
        a.geometry_ctx, a.binning_ctx, a.image_ctx = self.ws[0].key, self.ws[1].key, self.ws[2].key
        a.warm_state = C.pointer(self._warm)
        a.carry_state = C.pointer(self._carry)
        self._stats = (C.c_int * 4)(0, 0, 0, 0)
        a.stats_out = self._stats
        self._args = a
        # Once the three workspaces have their size (after the first call) they are handed over as FIXED buffers (include/gsr.h,
        # gsr_fixed_buffer_resize): the library's size requests then never enter the interpreter (three ctypes callbacks per call
        # otherwise); a request that no longer fits fails the call with GSR_E_ALLOC, which is repeated once with growing buffers.
        self._fb = (_lib.FixedBuffer * 3)()
        self._fb_addr = [C.addressof(self._fb[k]) for k in range(3)]

    def _tensor_versions(self):
        ts = (self.means3D, self.scales, self.rots, self.g_alpha, self.g_m2d, self.g_conic, self.g_opac, self.g_col, self.g_m3d, self.g_cov, self.g_sh, self.g_scale, self.g_rot)
        return tuple(-1 if t is None else t._version for t in ts)

    def _cached(self, slot, t, make):
        """Per-refiner cache of the per-frame constants' device-side conversions (projection matrix, background, mask ...):
        keyed by the source tensor's identity, storage address and in-place-modification counter, so a caller that hands in the
        same tensors for every frame (the reference's scripts do: one projection matrix, one background per run) pays for them
        once.  Contract: a write that bypasses torch's version counter (`t.data.copy_`, a numpy array shared with a CPU tensor)
        is not seen -- hand in a new tensor, or clear `refiner._conv_cache`."""
        if not torch.is_tensor(t):          # (a numpy depth map: no modification counter to trust)
            return make(t)
        key = (id(t), t.data_ptr(), t._version)
        hit = self._conv_cache.get(slot)
        if hit is None or hit[0] != key or hit[1] is not t:
            hit = (key, t, make(t))
            self._conv_cache[slot] = hit
        return hit[2]

    @staticmethod
    def _env_flags():
        """GSR_NO_LEAN / GSR_SH_SEPARATE / GSR_NO_BALANCE / GSR_DEBUG_TILES / GSR_DETERMINISTIC / GSR_NO_SPLIT / GSR_NO_DILATE / GSR_GRADS_EVERY_ITERATION of the environment ->
        gsr_refine_args.flags (the library itself reads no environment variable on this path)."""
        from . import _lib
        from .rasterizer import _env_has as has
        return ((_lib.REFINE_NO_LEAN if has("GSR_NO_LEAN") else 0) | (_lib.REFINE_SH_SEPARATE if has("GSR_SH_SEPARATE") else 0) |
                (_lib.REFINE_NO_BALANCE if has("GSR_NO_BALANCE") else 0) | (_lib.REFINE_LOG_REDO if has("GSR_DEBUG_TILES") else 0) |
                (_lib.REFINE_DETERMINISTIC if has("GSR_DETERMINISTIC") else 0) | (_lib.REFINE_NO_SPLIT if has("GSR_NO_SPLIT") else 0) | (_lib.REFINE_NO_DILATE if has("GSR_NO_DILATE") else 0) |
                (_lib.REFINE_GRADS_EVERY_ITERATION if has("GSR_GRADS_EVERY_ITERATION") else 0))

    def refine(self, viewpoint, config, initial_R, initial_T, background, iters=50, lr=0.001, converged_threshold=1e-4,
               stop_on_converged=True, speculative=True, bound_margin=None, warm_start=None, count_instances=False,
               scale_modifier=1.0, lean_min_P=0, flags=None):
        C, _lib = self._C, self._lib_mod
        dev = self.dev
        # speculative=True: exact optimisation (include/gsr.h, gsr_refine_args.speculative): ~9x fewer binned
        # instances and ~7x fewer SH rows on S-1M-640.
        viewpoint.update_RT(initial_R, initial_T)
        def f32d(t):          # float32, contiguous, on the device: usually it already is
            if t.dtype is torch.float32 and t.device == dev and t.is_contiguous():
                return t.detach() if t.requires_grad else t
            return t.detach().to(device=dev, dtype=torch.float32).contiguous()
        # the start pose goes to the library as the camera's own device tensors (gsr_refine_args.init_*: the pose state is
        # built by one launch inside the call; reading R, T and the exposure back to build it on the host would cost four
        # stream synchronisations, five tiny torch copies cost five launches)
        R0, T0 = f32d(viewpoint.R), f32d(viewpoint.T)
        ea0, eb0 = f32d(viewpoint.exposure_a).reshape(-1), f32d(viewpoint.exposure_b).reshape(-1)
        proj_raw = self._cached("proj", viewpoint.projection_matrix, f32d)
        gt_image = self._cached("gt", viewpoint.original_image, f32d)
        mono = bool(config["Training"]["monocular"])
        gt_depth = None
        if not mono:
            gt_depth = self._cached("gtd", viewpoint.depth, lambda gd: (torch.from_numpy(gd) if not torch.is_tensor(gd) else gd).to(dtype=torch.float32, device=dev).contiguous())

        def make_mask(m):
            m = m.to(device=dev).reshape(self.H, self.W).contiguous()
            return m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)      # (a bool tensor already is one byte per pixel)
        mask = self._cached("mask", viewpoint.grad_mask, make_mask)
        bg = self._cached("bg", background, f32d)
        alpha_cfg = config["Training"]["alpha"] if "alpha" in config["Training"] else 0.98
        stream = torch.cuda.current_stream(dev).cuda_stream
        p = lambda t: None if t is None else t.data_ptr()
        a = self._args
        a.D = int(self.model.active_sh_degree)
        a.scale_modifier = float(scale_modifier)
        # warm_start: start speculating from the depth bounds the previous refine() of this refiner left behind instead of
        # binning the first iteration completely (still verified on the device, still exact: stale bounds cost one redone
        # forward).  None = whenever such bounds exist -- consecutive query frames of a sequence see almost the same depths.
        if warm_start is False:
            self._warm.value = 0
        # (anything torch has written into them since -- fr.g_sh.zero_(), an optimiser stepping the scales -- shows in the
        # tensors' version counters and withdraws the promise; so does another scale modifier)
        if self._carry_versions != self._tensor_versions() + (float(scale_modifier),):
            self._carry.value = 0
        a.tan_fovx, a.tan_fovy = math.tan(viewpoint.FoVx * 0.5), math.tan(viewpoint.FoVy * 0.5)
        a.background, a.projmatrix_raw = p(bg), p(proj_raw)
        a.gt_image, a.gt_depth, a.grad_mask = p(gt_image), p(gt_depth), p(mask)
        a.opacity_threshold = float(config["Training"]["opacity_threshold"])
        a.depth_weight = float(1 - alpha_cfg)
        a.monocular = int(mono)
        a.init_R, a.init_T, a.init_exposure_a, a.init_exposure_b = p(R0), p(T0), p(ea0), p(eb0)
        a.lr, a.converged_threshold, a.max_iters = float(lr), float(converged_threshold), int(iters)
        a.stop_on_converged = int(bool(stop_on_converged))
        a.speculative = int(bool(speculative))
        # bound_margin=None: adaptive margin of the speculative depth bounds (include/gsr.h); (mul, add) fixes it
        a.bound_margin_mul, a.bound_margin_add = (0.0, 0.0) if bound_margin is None else (float(bound_margin[0]), float(bound_margin[1]))
        # count_instances: also report how many tile instances the LAST forward binned (info["num_rendered"]; costs a copy of
        # the tile ranges to the host and a stream synchronisation)
        stats = self._stats
        stats[0], stats[1], stats[2], stats[3] = 0, (0 if count_instances else -1), 0, 0
        a.stream = stream
        # a fresh pose state per call (the pose load of k_refine_init fills all of it): the pose handed back below is a VIEW of it, not a copy -- three
        # tiny torch kernels per call, each behind a host-side launch gap with the GPU idle (0.1 ms of a 3 ms call)
        self.state = torch.empty(_lib.POSE_STATE_FLOATS, dtype=torch.float32, device=dev)
        a.pose_state = self.state.data_ptr()
        a.flags = self._env_flags() if flags is None else int(flags)
        a.lean_min_P = int(lean_min_P)
        n_done, conv = C.c_int(0), C.c_int(0)
        ws = self.ws
        fixed = all(w.t.numel() > 0 for w in ws)
        if fixed:
            ff = _lib.fixed_buffer_fn()
            for k in range(3):
                self._fb[k].ptr, self._fb[k].capacity = ws[k].t.data_ptr(), ws[k].t.numel()
            a.geometry_buffer = a.binning_buffer = a.image_buffer = ff
            a.geometry_ctx, a.binning_ctx, a.image_ctx = self._fb_addr
        else:
            a.geometry_buffer, a.binning_buffer, a.image_buffer = ws[0].fn, ws[1].fn, ws[2].fn
            a.geometry_ctx, a.binning_ctx, a.image_ctx = ws[0].key, ws[1].key, ws[2].key
        cur = torch.cuda.current_device()
        if cur != dev.index:
            torch.cuda.set_device(dev)
        try:
            rc = self.lib.gsr_refine(C.byref(a), C.byref(n_done), C.byref(conv))
            if fixed and rc == _lib.E_ALLOC:          # a workspace has to grow: once more through the growing callbacks
                self._warm.value = 0
                self._carry.value = 0
                a.geometry_buffer, a.binning_buffer, a.image_buffer = ws[0].fn, ws[1].fn, ws[2].fn
                a.geometry_ctx, a.binning_ctx, a.image_ctx = ws[0].key, ws[1].key, ws[2].key
                rc = self.lib.gsr_refine(C.byref(a), C.byref(n_done), C.byref(conv))
            type(ws[0]).raise_pending(*ws)
            if rc < 0:
                _lib.check(rc)
        finally:
            if cur != dev.index:
                torch.cuda.set_device(cur)
        self._carry_versions = self._tensor_versions() + (float(scale_modifier),)
        self._last_args = a                                          # (gsr_debug_lean_check takes the same struct)
        self._keep = (R0, T0, ea0, eb0, proj_raw, gt_image, gt_depth, mask, bg)        # alive until the stream has drained
        # the final pose came back with the call (gsr_refine_args.pose_state_host): no second blocking read.  The camera gets
        # device tensors that are views of this call's state (no copy, no launch; host -> device uploads would each stall)
        s = self._state_host_np
        st = self.state          # (views of a tensor that does not require grad: no autograd bookkeeping to switch off)
        viewpoint.update_RT(st[0:9].view(3, 3), st[9:12])
        viewpoint.exposure_a.data = st[18:19].view(viewpoint.exposure_a.shape)
        viewpoint.exposure_b.data = st[19:20].view(viewpoint.exposure_b.shape)
        info = {"iters": n_done.value, "converged": bool(conv.value), "loss": float(s[38]),
                "fallbacks": int(stats[0]), "num_rendered": int(stats[1]), "lean_iters": int(stats[2]), "host_redos": int(stats[3]),
                # (host copies of the final pose: no device read-back for the caller's error statistics)
                "R_host": s[0:9].reshape(3, 3).copy(), "T_host": s[9:12].copy(),
                # the pose and exposure the LAST forward / backward of the call ran with (the state before the final update; pose-state
                # words 96..109): what `render` / `depth` / `opacity` below and the gradient tensors g_* belong to when max_iters was reached
                "R_last_forward_host": s[96:105].reshape(3, 3).copy(), "T_last_forward_host": s[105:108].copy(),
                "exposure_last_forward_host": s[108:110].copy(),
                "render": self.color, "depth": self.depth, "opacity": self.alpha}
        self.last_info = {k: info[k] for k in ("fallbacks", "num_rendered", "lean_iters", "host_redos")}
        return viewpoint.R, viewpoint.T, info

    def seg_stats(self):
        """gsr_debug_seg_stats on the state the last refine() left (tests, tools): (blocks with work, tiles split, largest number of
        depth ranges of a tile, blocks per launch) of its last speculative iteration."""
        C, _lib = self._C, self._lib_mod
        out = (C.c_longlong * 4)()
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.gsr_debug_seg_stats(C.byref(self._last_args), out))
        return tuple(int(x) for x in out)

    def lean_check(self):
        """gsr_debug_lean_check on the state the last refine() left (tests): (settled, candidates, binned by the exact walk,
        violations, first violating index)."""
        C, _lib = self._C, self._lib_mod
        out = (C.c_longlong * 5)()
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.gsr_debug_lean_check(C.byref(self._last_args), out))
        return tuple(int(x) for x in out)


# ------------------------------------------------------------------ the per-frame gradient mask
_GM_WORKSPACES = {}          # one scratch buffer per (device, stream): three histograms + the intensity image; grows to the largest frame seen


def _gm_workspace(dev):
    from .rasterizer import _Workspace
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)          # (calls on different streams may overlap: no shared scratch)
    ws = _GM_WORKSPACES.get(key)
    if ws is None:
        ws = _GM_WORKSPACES[key] = _Workspace(dev)
    return ws


def grad_mask(original_image, edge_threshold, keypoints=None, box_k=10, return_intensity=False):
    """`Camera.compute_grad_mask` (tools/camera_utils.py:164-193, every dataset type but "replica") and, with `keypoints`, the
    `grad_mask | create_mask(keypoints, width, height, k)` step of the scripts (7scenes_localize_full_dslam.py:126-149,355-360)
    as ONE C-ABI call (`gsr_grad_mask`): [3,H,W] float image on a HIP device -> bool [1,H,W] on that device.
    keypoints: [K,2] (x, y) -- numpy, a list or a tensor (the scripts pass group['keypoints'][scores > 0.2]).
    Raises on a CPU tensor: there is no CPU path."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    if not torch.is_tensor(original_image) or original_image.device.type != "cuda":
        raise _lib.GsrError("grad_mask: original_image must be a tensor on the HIP device (there is no CPU fallback)")
    dev = original_image.device
    img = original_image.detach()
    if img.dtype is not torch.float32 or not img.is_contiguous():
        img = img.to(torch.float32).contiguous()
    if img.dim() != 3 or img.shape[0] != 3:
        raise ValueError("grad_mask: original_image must be [3, H, W]")
    H, W = int(img.shape[1]), int(img.shape[2])
    kp, nk = None, 0
    if keypoints is not None and len(keypoints):
        kp = torch.as_tensor(np.asarray(keypoints, np.float32) if not torch.is_tensor(keypoints) else keypoints, dtype=torch.float32).reshape(-1, 2)
        kp = kp.to(dev, non_blocking=True).contiguous()
        nk = int(kp.shape[0])
    mask = torch.empty((1, H, W), dtype=torch.bool, device=dev)
    inten = torch.empty((H, W), dtype=torch.float32, device=dev) if return_intensity else None
    med = torch.empty(2, dtype=torch.float32, device=dev) if return_intensity else None
    ws = _gm_workspace(dev)
    p = lambda t: None if t is None else t.data_ptr()
    with torch.cuda.device(dev):
        rc = lib.gsr_grad_mask(W, H, p(img), float(edge_threshold), p(kp), nk, int(box_k), p(mask), p(inten), p(med), ws.fn, ws.key,
                               torch.cuda.current_stream(dev).cuda_stream)
    type(ws).raise_pending(ws)
    _lib.check(rc)
    return (mask, inten, med) if return_intensity else mask


def grad_mask_replica(original_image, edge_threshold, rows=32, cols=32):
    """the config["Dataset"]["type"] == "replica" branch of `Camera.compute_grad_mask` (tools/camera_utils.py:174-188): a FLOAT
    [1,H,W] image, the reference's quirks included (include/gsr.h, `gsr_grad_mask_replica`)."""
    from . import _lib
    lib = _lib.load()
    if not torch.is_tensor(original_image) or original_image.device.type != "cuda":
        raise _lib.GsrError("grad_mask_replica: original_image must be a tensor on the HIP device (there is no CPU fallback)")
    dev = original_image.device
    img = original_image.detach().to(torch.float32).contiguous()
    H, W = int(img.shape[1]), int(img.shape[2])
    out = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    ws = _gm_workspace(dev)
    with torch.cuda.device(dev):
        rc = lib.gsr_grad_mask_replica(W, H, img.data_ptr(), float(edge_threshold), int(rows), int(cols), out.data_ptr(), ws.fn, ws.key,
                                       torch.cuda.current_stream(dev).cuda_stream)
    type(ws).raise_pending(ws)
    _lib.check(rc)
    return out


def compute_grad_mask(viewpoint, config, keypoints=None, box_k=10):
    """Drop-in for `viewpoint.compute_grad_mask(config)` [+ the keypoint boxes]: sets and returns `viewpoint.grad_mask`."""
    thr = config["Training"]["edge_threshold"]
    if config.get("Dataset", {}).get("type") == "replica":
        viewpoint.grad_mask = grad_mask_replica(viewpoint.original_image, thr)
    else:
        viewpoint.grad_mask = grad_mask(viewpoint.original_image, thr, keypoints, box_k)
    return viewpoint.grad_mask


def pose_errors(R_gt, t_gt, R, t):
    """(translation error [m], rotation error [deg]) as 7scenes_localize_full_dslam.py:368-377"""
    R_gt, t_gt, R, t = (np.asarray(x, np.float64) for x in (R_gt, t_gt, R, t))
    trans_error = float(np.linalg.norm(-R_gt.T @ t_gt.reshape(3, 1) + R.T @ t.reshape(3, 1)))
    cos = np.clip((np.trace(np.dot(R_gt.T, R)) - 1) / 2, -1.0, 1.0)
    return trans_error, float(np.rad2deg(np.abs(np.arccos(cos))))


TRACKING_CONFIG = {"Training": {"monocular": False, "alpha": 0.99, "opacity_threshold": 0.99, "edge_threshold": 1.1}}
