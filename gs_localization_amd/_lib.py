"""ctypes loader of the C-ABI library (include/gsr.h).  No fallback: if libgsr_hip.so is missing
or cannot be loaded, every entry point raises -- the product path never runs on the CPU."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSR_LIB_PATH") or os.path.join(_HERE, "libgsr_hip.so")      # (GSR_LIB_PATH: a diagnostic build, see build.py)

RESIZE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

_vp, _i, _f = C.c_void_p, C.c_int, C.c_float

# name -> (restype, argtypes); must list every symbol include/gsr.h declares (tests check this)
SIGNATURES = {
    "gsr_forward": (_i, [RESIZE_FN, _vp, RESIZE_FN, _vp, RESIZE_FN, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp,
                         _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "gsr_spec_state_bytes": (C.c_size_t, [_i, _i]),
    "gsr_spec_state_bounds_bytes": (C.c_size_t, [_i, _i]),
    "gsr_forward_speculative": (_i, [_vp, RESIZE_FN, _vp, RESIZE_FN, _vp, RESIZE_FN, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp,
                                     _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "gsr_backward": (_i, [_i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp,
                          _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "gsr_mark_visible": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "gsr_knn_bytes": (C.c_size_t, [_i]),
    "gsr_dist2_knn3": (_i, [_i, _vp, _vp, RESIZE_FN, _vp, _vp]),
    "gsr_training_loss_bytes": (C.c_size_t, [_i, _i]),
    "gsr_training_loss": (_i, [_i, _i, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _vp, RESIZE_FN, _vp, _vp]),
    "gsr_densification_stats": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsr_map_from_ply_rows": (_i, [_i, _vp, _i, C.POINTER(C.c_int), _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsr_geometry_bytes": (C.c_size_t, [_i]),
    "gsr_geometry_bytes_det": (C.c_size_t, [_i]),
    "gsr_image_bytes": (C.c_size_t, [_i, _i]),
    "gsr_binning_bytes": (C.c_size_t, [_i]),
    "gsr_binning_bytes_bins": (C.c_size_t, [_i, _i, _i]),
    "gsr_fixed_buffer_resize": (_vp, [_vp, C.c_size_t]),
    "gsr_forward_stats": (_i, [_i, _i, _i, _vp, _vp, _vp, C.POINTER(C.c_longlong), _vp]),
    "gsr_profile_enable": (_i, [C.c_uint]),
    "gsr_profile_sampling": (_i, [C.c_uint]),
    "gsr_debug_timing": (_i, [C.POINTER(C.c_ulonglong)]),
    "gsr_debug_tile_order": (_i, [_vp, _vp, _i, _vp]),
    "gsr_profile_collect": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "gsr_profile_kernel_count": (_i, []),
    "gsr_profile_kernel_name": (C.c_char_p, [_i]),
    "gsr_last_error": (C.c_char_p, []),
    "gsr_abi_version": (_i, []),
    "gsr_device_ok": (_i, []),
}

class FixedBuffer(C.Structure):
    """mirror of `gsr_fixed_buffer` (include/gsr.h)"""
    _fields_ = [("ptr", _vp), ("capacity", C.c_size_t), ("requested", C.c_size_t)]


E_ALLOC = -3


class SpecState(C.Structure):
    """mirror of `gsr_spec_state` (include/gsr.h)"""
    _fields_ = [("device_buffer", _vp), ("width", _i), ("height", _i), ("valid", _i), ("parity", _i), ("fail_streak", _i),
                ("skip", _i), ("last_speculative", _i), ("n_speculative", _i), ("n_failed", _i)]


class ForwardArgs(C.Structure):
    """mirror of `gsr_forward_args` (include/gsr.h): one pointer across the ctypes boundary instead of 34 arguments"""
    _fields_ = [("state", _vp), ("geometry_buffer", RESIZE_FN), ("geometry_ctx", _vp), ("binning_buffer", RESIZE_FN), ("binning_ctx", _vp),
                ("image_buffer", RESIZE_FN), ("image_ctx", _vp), ("P", _i), ("D", _i), ("M", _i), ("background", _vp), ("width", _i), ("height", _i),
                ("means3D", _vp), ("shs", _vp), ("colors_precomp", _vp), ("opacities", _vp), ("scales", _vp), ("scale_modifier", _f),
                ("rotations", _vp), ("cov3D_precomp", _vp), ("viewmatrix", _vp), ("projmatrix", _vp), ("cam_pos", _vp), ("tan_fovx", _f),
                ("tan_fovy", _f), ("prefiltered", _i), ("out_color", _vp), ("out_depth", _vp), ("out_alpha", _vp), ("radii", _vp), ("debug", _i),
                ("n_touched", _vp), ("stream", _vp)]


class BackwardArgs(C.Structure):
    """mirror of `gsr_backward_args` (include/gsr.h)"""
    _fields_ = [("P", _i), ("D", _i), ("M", _i), ("R", _i), ("background", _vp), ("width", _i), ("height", _i), ("means3D", _vp), ("shs", _vp),
                ("colors_precomp", _vp), ("alphas", _vp), ("scales", _vp), ("scale_modifier", _f), ("rotations", _vp), ("cov3D_precomp", _vp),
                ("viewmatrix", _vp), ("projmatrix", _vp), ("campos", _vp), ("tan_fovx", _f), ("tan_fovy", _f), ("radii", _vp),
                ("geom_buffer", _vp), ("binning_buffer", _vp), ("img_buffer", _vp), ("dL_dpix", _vp), ("dL_ddepths", _vp), ("dL_dalphas", _vp),
                ("dL_dmean2D", _vp), ("dL_dconic", _vp), ("dL_dopacity", _vp), ("dL_dcolor", _vp), ("dL_dmean3D", _vp), ("dL_dcov3D", _vp),
                ("dL_dsh", _vp), ("dL_dscale", _vp), ("dL_drot", _vp), ("debug", _i), ("pose_mode", _i), ("dL_dtau", _vp), ("stream", _vp)]


class RefineArgs(C.Structure):
    """mirror of `gsr_refine_args` (include/gsr.h)"""
    _fields_ = [
        ("P", _i), ("D", _i), ("M", _i),
        ("means3D", _vp), ("shs", _vp), ("opacities", _vp), ("scales", _vp), ("rotations", _vp),
        ("scale_modifier", _f),
        ("width", _i), ("height", _i), ("tan_fovx", _f), ("tan_fovy", _f),
        ("background", _vp), ("projmatrix_raw", _vp),
        ("gt_image", _vp), ("gt_depth", _vp), ("grad_mask", _vp),
        ("opacity_threshold", _f), ("depth_weight", _f), ("monocular", _i),
        ("pose_state", _vp),
        ("out_color", _vp), ("out_depth", _vp), ("out_alpha", _vp), ("radii", _vp), ("n_touched", _vp),
        ("dL_dimage", _vp), ("dL_ddepth", _vp), ("dL_dalpha", _vp),
        ("dL_dmean2D", _vp), ("dL_dconic", _vp), ("dL_dopacity", _vp), ("dL_dcolor", _vp),
        ("dL_dmean3D", _vp), ("dL_dcov3D", _vp), ("dL_dsh", _vp), ("dL_dscale", _vp), ("dL_drot", _vp),
        ("dL_dtau", _vp), ("loss_out", _vp),
        ("geometry_buffer", RESIZE_FN), ("geometry_ctx", _vp),
        ("binning_buffer", RESIZE_FN), ("binning_ctx", _vp),
        ("image_buffer", RESIZE_FN), ("image_ctx", _vp),
        ("lr", _f), ("converged_threshold", _f), ("max_iters", _i), ("stop_on_converged", _i),
        ("speculative", _i), ("bound_margin_mul", _f), ("bound_margin_add", _f), ("stats_out", C.POINTER(_i)),
        ("warm_state", C.POINTER(_i)), ("carry_state", C.POINTER(_i)), ("stream", _vp),
        ("flags", C.c_uint), ("lean_min_P", _i),
        ("init_R", _vp), ("init_T", _vp), ("init_exposure_a", _vp), ("init_exposure_b", _vp),
        ("pose_state_host", C.POINTER(_f)),
        ("colors_precomp", _vp), ("cov3D_precomp", _vp),
    ]


# the GSR_ABI_VERSION these ctypes mirrors (SpecState, RefineArgs, the pose-state layout) were written for
ABI_VERSION = 5
REFINE_NO_LEAN, REFINE_SH_SEPARATE, REFINE_NO_BALANCE, REFINE_LOG_REDO, REFINE_DETERMINISTIC, REFINE_NO_SPLIT, REFINE_NO_DILATE = 1, 2, 4, 8, 16, 32, 64
REFINE_GRADS_EVERY_ITERATION = 128      # diagnostics: the Gaussian-parameter gradient rows written by every iteration instead of once per call


POSE_STATE_FLOATS = 112
SIGNATURES.update({
    "gsr_tracking_loss": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "gsr_pose_init": (_i, [_vp, _vp, _vp]),
    "gsr_pose_step": (_i, [_vp, _vp, _vp, _vp, _f, _f, _vp]),
    "gsr_refine": (_i, [C.POINTER(RefineArgs), C.POINTER(_i), C.POINTER(_i)]),
    "gsr_forward_packed": (_i, [C.POINTER(ForwardArgs)]),
    "gsr_backward_packed": (_i, [C.POINTER(BackwardArgs)]),
    "gsr_debug_lean_check": (_i, [C.POINTER(RefineArgs), C.POINTER(C.c_longlong)]),
    "gsr_debug_seg_stats": (_i, [C.POINTER(RefineArgs), C.POINTER(C.c_longlong)]),
    "gsr_debug_lam_offset": (C.c_size_t, [_i]),
    "gsr_grad_mask_bytes": (C.c_size_t, [_i, _i]),
    "gsr_grad_mask": (_i, [_i, _i, _vp, _f, _vp, _i, _i, _vp, _vp, _vp, RESIZE_FN, _vp, _vp]),
    "gsr_grad_mask_replica": (_i, [_i, _i, _vp, _f, _i, _i, _vp, RESIZE_FN, _vp, _vp]),
})

_lib = None
_fixed_fn = None


def fixed_buffer_fn():
    """`gsr_fixed_buffer_resize` as a RESIZE_FN value: a resize callback that never enters the interpreter"""
    global _fixed_fn
    if _fixed_fn is None:
        load()
        _fixed_fn = RESIZE_FN(("gsr_fixed_buffer_resize", _lib))
    return _fixed_fn


class GsrError(RuntimeError):
    pass


def load():
    """Returns the loaded library; raises GsrError if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GsrError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64.  If libgsr_hip.so is loaded
        # FIRST it binds to the system's (/opt/rocm), torch then brings its own, and with two runtimes in the process the one
        # initialised second sees no device (gsr_device_ok() == 0 although torch.cuda.is_available(); found in round 3 by running
        # build() and smoke() in one process).  Loaded after torch, the library resolves against torch's runtime -- the one
        # whose streams and device pointers it is handed anyway.
        try:
            import torch  # noqa: F401
        except ImportError:          # (a Python caller without torch: the system runtime is the only one)
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        got = lib.gsr_abi_version()
        if got != ABI_VERSION:      # a stale .so would read these structs with another layout
            raise GsrError(f"{LIB_PATH} has ABI version {got}, this package was written for {ABI_VERSION}: rebuild it "
                           "(python -c 'import __graft_entry__ as g; g.build()')")
        _lib = lib
    return _lib


def check(rc):
    if rc < 0:
        raise GsrError(f"gsr error {rc}: {load().gsr_last_error().decode()}")
    return rc
