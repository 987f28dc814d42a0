"""Drop-in for `diff_gaussian_rasterization_pose`, the un-vendored rasterizer with camera-pose
gradients that gs_localization/pipelines/tools/__init__.py:15-18 imports.  Settings carry the extra
`projmatrix_raw` field (tools/__init__.py:58-72); forward takes `theta`/`rho` and returns
(color, radii, depth, opacity, n_touched) (tools/__init__.py:130-141)."""
from gs_localization_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettingsPose as GaussianRasterizationSettings,
    GaussianRasterizerPose as GaussianRasterizer,
    rasterize_gaussians_pose as rasterize_gaussians,
    cpu_deep_copy_tuple,
    _RasterizeGaussiansPose as _RasterizeGaussians,
)
