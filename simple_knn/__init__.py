"""Drop-in for the reference's `simple_knn` package (gaussian_splatting/submodules/simple-knn): only `_C.distCUDA2`
is used by the reference (scene/gaussian_model.py:20,134; pipelines/tools/gaussian_model.py:18,185)."""
