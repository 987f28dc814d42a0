"""CPU: host-side logic, the Python surface of the two drop-in packages, and the C-ABI library
(loads, exports every symbol include/gsr.h declares; no compute without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "gsr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z_]+)\s*\(", hdr)) - {"gsr_resize_fn"})


def test_library_exports_every_declared_symbol():
    from gs_localization_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), n
        assert n in _lib.SIGNATURES, f"{n} missing from the ctypes signature table"
    assert lib.gsr_abi_version() == _lib.ABI_VERSION == 5
    names_k = [lib.gsr_profile_kernel_name(i).decode() for i in range(lib.gsr_profile_kernel_count())]
    assert names_k == ["preprocess_fwd", "sh_color", "tile_count", "tile_scan", "tile_emit", "render_fwd", "bwd_zero", "render_bwd",
                       "preprocess_bwd", "pose_step"]


def test_workspace_sizes():
    from gs_localization_amd import _lib
    lib = _lib.load()
    g1, g2 = lib.gsr_geometry_bytes(1000), lib.gsr_geometry_bytes(1_000_000)
    assert 0 < g1 < g2 and g2 < 180 * 1_000_000      # (~160 B per Gaussian; the deterministic option's 64-bit accumulator records only on request)
    assert g2 + 140 * 1_000_000 < lib.gsr_geometry_bytes_det(1_000_000) < 320 * 1_000_000
    assert lib.gsr_image_bytes(640, 480) >= 640 * 480 * 4 + 1200 * 8
    assert lib.gsr_binning_bytes(1_000_000) >= 12 * 1_000_000


def test_entry_points_reject_bad_arguments_without_touching_a_gpu():
    from gs_localization_amd import _lib
    lib = _lib.load()
    cb = _lib.RESIZE_FN(lambda ctx, n: 0)
    rc = lib.gsr_forward(cb, None, cb, None, cb, None, -1, 0, 0, None, 16, 16, None, None, None, None, None, 1.0, None,
                         None, None, None, None, 1.0, 1.0, 0, None, None, None, None, 0, None, None)
    assert rc == -1 and b"P >= 0" in lib.gsr_last_error()
    with pytest.raises(_lib.GsrError):
        _lib.check(rc)
    assert lib.gsr_mark_visible(-3, None, None, None, None, None) == -1


def test_package_surface_a():
    import diff_gaussian_rasterization as A
    assert A.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")
    r = A.GaussianRasterizer(raster_settings=None)
    assert isinstance(r, torch.nn.Module) and hasattr(r, "markVisible")
    z = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=z, means2D=z, opacities=z[:, :1], scales=z, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=z, means2D=z, opacities=z[:, :1], colors_precomp=z, scales=z)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=z, means2D=z, opacities=z[:, :1], colors_precomp=z, scales=z, rotations=torch.zeros(4, 4),
          cov3D_precomp=torch.zeros(4, 6))


def test_package_surface_b():
    import diff_gaussian_rasterization_pose as B
    f = B.GaussianRasterizationSettings._fields
    assert len(f) == 13 and f[8] == "projmatrix_raw" and f[:8] == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix")
    import inspect
    sig = inspect.signature(B.GaussianRasterizer.forward)
    assert list(sig.parameters)[-2:] == ["theta", "rho"]


def test_no_silent_cpu_fallback():
    """the product path must fail loudly when asked to run without a HIP device"""
    import diff_gaussian_rasterization as A
    z = torch.zeros(4, 3)
    rs = A.GaussianRasterizationSettings(image_height=16, image_width=16, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3),
                                         scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4),
                                         sh_degree=0, campos=torch.zeros(3), prefiltered=False, debug=False)
    with pytest.raises(RuntimeError, match="no CPU rasterizer"):
        A.GaussianRasterizer(rs)(means3D=z, means2D=z, opacities=z[:, :1], colors_precomp=z, scales=z,
                                 rotations=torch.zeros(4, 4))
    with pytest.raises(RuntimeError):
        A.GaussianRasterizer(rs).markVisible(z)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gs_localization_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in src.replace("no oracle", ""), os.path.join(dirpath, fn)
    for pkg in ("diff_gaussian_rasterization", "diff_gaussian_rasterization_pose"):
        assert "oracle" not in open(os.path.join(ROOT, pkg, "__init__.py")).read()


def test_product_holds_no_restated_reference_python():
    """The reference-style loop (render / loss / Adam / pose update as the scripts issue them) is test infrastructure
    (tests/replay.py); the product package offers the native loop only and never imports from tests/."""
    import gs_localization_amd.pipelines as P
    for name in ("SE3_exp", "SO3_exp", "update_pose", "get_loss_tracking", "gradient_decent", "render", "Camera", "getProjectionMatrix2"):
        assert not hasattr(P, name), name
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gs_localization_amd")):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                assert "from tests" not in src and "import tests" not in src, fn


def test_workspaces_do_not_outlive_their_forward():
    """The resize callbacks find their workspace through a registry; it must not keep a forward's buffers alive (a train.py
    run would otherwise hold every step's geometry / binning / image buffers: 0.5 GB a step at 1.5 M Gaussians)."""
    import gc
    import torch
    from gs_localization_amd import rasterizer as RZ
    w = RZ._Workspace(torch.device("cpu"))
    key = w.key
    assert RZ._workspace_dispatch(key, 64) == w.t.data_ptr() and w.t.numel() == 64
    kept = w.t                                   # what autograd saves
    del w
    gc.collect()
    assert key not in RZ._Workspace._registry
    assert RZ._workspace_dispatch(key, 64) == 0  # a stale context gets NULL, not a dangling buffer
    assert kept.numel() == 64


def test_cpython_hop_is_built_and_speaks_the_same_abi():
    """csrc/gsrcall.c: the drop-in packages' hop into the C ABI (no ctypes on the fast route).  It must be there, built against the same
    gsr.h as the ctypes mirrors, and refuse a call with the wrong number of arguments instead of reading past them."""
    import pytest
    from gs_localization_amd import rasterizer as RZ, _lib
    assert RZ._gsrcall is not None, "gs_localization_amd/_gsrcall.so missing: python gs_localization_amd/build.py"
    assert RZ._gsrcall.ABI_VERSION == _lib.ABI_VERSION and RZ._gsrcall.E_ALLOC == _lib.E_ALLOC
    with pytest.raises(TypeError):
        RZ._gsrcall.forward(0, 1, 2)
    with pytest.raises(TypeError):
        RZ._gsrcall.backward(0)
    assert isinstance(RZ._gsrcall.last_error(), str)
