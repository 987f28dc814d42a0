"""Drop-in for the reference's `diff_gaussian_rasterization` package
(gaussian_splatting/submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py),
imported by gaussian_splatting/gaussian_renderer/__init__.py:14.  Same names, same signatures,
same tensor semantics; the work is done by hand-written gfx950 kernels behind include/gsr.h."""
from gs_localization_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
    cpu_deep_copy_tuple,
    _RasterizeGaussians,
)
