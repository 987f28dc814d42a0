"""ctypes loader of the C-ABI library (include/gsr.h).  No fallback: if libgsr_hip.so is missing
or cannot be loaded, every entry point raises -- the product path never runs on the CPU."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgsr_hip.so")

RESIZE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

_vp, _i, _f = C.c_void_p, C.c_int, C.c_float

# name -> (restype, argtypes); must list every symbol include/gsr.h declares (tests check this)
SIGNATURES = {
    "gsr_forward": (_i, [RESIZE_FN, _vp, RESIZE_FN, _vp, RESIZE_FN, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp,
                         _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "gsr_backward": (_i, [_i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp,
                          _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "gsr_mark_visible": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "gsr_geometry_bytes": (C.c_size_t, [_i]),
    "gsr_image_bytes": (C.c_size_t, [_i, _i]),
    "gsr_binning_bytes": (C.c_size_t, [_i]),
    "gsr_forward_stats": (_i, [_i, _i, _i, _vp, _vp, _vp, C.POINTER(C.c_longlong), _vp]),
    "gsr_profile_enable": (_i, [C.c_uint]),
    "gsr_profile_collect": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "gsr_profile_kernel_count": (_i, []),
    "gsr_profile_kernel_name": (C.c_char_p, [_i]),
    "gsr_last_error": (C.c_char_p, []),
    "gsr_abi_version": (_i, []),
    "gsr_device_ok": (_i, []),
}

_lib = None


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
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc < 0:
        raise GsrError(f"gsr error {rc}: {load().gsr_last_error().decode()}")
    return rc
