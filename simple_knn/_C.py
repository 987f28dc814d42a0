"""`from simple_knn._C import distCUDA2` -- spatial.cu:15-26 on the MI355X C ABI (gsr_dist2_knn3)."""
import ctypes as C

import torch

from gs_localization_amd import _lib


def distCUDA2(points):
    """points: [P, 3] float tensor on the GPU -> [P] mean squared distance to the three nearest other points."""
    lib = _lib.load()
    if points.device.type != "cuda":
        raise RuntimeError("distCUDA2 needs a HIP tensor (there is no CPU path)")
    pts = points.detach().contiguous().float()
    P = pts.shape[0]
    means = torch.full((P,), 0.0, dtype=torch.float32, device=pts.device)
    keep = {}

    def resize(_ctx, n):
        keep["ws"] = torch.empty(n, dtype=torch.uint8, device=pts.device)
        return keep["ws"].data_ptr()
    with torch.cuda.device(pts.device):
        _lib.check(lib.gsr_dist2_knn3(P, C.c_void_p(pts.data_ptr()), C.c_void_p(means.data_ptr()), _lib.RESIZE_FN(resize), None,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.current_stream().synchronize()          # the workspace is released on return
    return means
