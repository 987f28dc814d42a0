"""Host-side mirror of the training step's loss epilogue:
gaussian_splatting/train.py:92-108 (L1 + SSIM + pseudo-depth Pearson term) and :142-145 (densification statistics).

`training_loss(image, gt_image, lambda_dssim, depth, pseudo_depth)` returns the loss as a tensor whose backward
delivers dL/dimage and dL/ddepth computed by the fused HIP kernels (`gsr_training_loss`); nothing falls back to
torch ops."""
import ctypes as C

import torch

from . import _lib


class _TrainingLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt_image, depth, pseudo_depth, lambda_dssim, depth_weight):
        lib = _lib.load()
        if image.device.type != "cuda":
            raise RuntimeError("training_loss needs HIP tensors (there is no CPU path)")
        _, H, W = image.shape
        x, g = image.detach().contiguous().float(), gt_image.detach().contiguous().float()
        d = depth.detach().contiguous().float() if depth is not None else None
        m = pseudo_depth.detach().contiguous().float() if pseudo_depth is not None else None
        gi = torch.empty_like(x)
        gd = torch.empty((H, W), dtype=torch.float32, device=x.device) if d is not None else None
        out = torch.empty(4, dtype=torch.float32, device=x.device)
        keep = {}

        def resize(_ctx, n):
            keep["ws"] = torch.empty(n, dtype=torch.uint8, device=x.device)
            return keep["ws"].data_ptr()
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        with torch.cuda.device(x.device):
            _lib.check(lib.gsr_training_loss(W, H, p(x), p(g), float(lambda_dssim), p(d), p(m), float(depth_weight), p(gi), p(gd),
                                             p(out), _lib.RESIZE_FN(resize), None,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        ctx.save_for_backward(gi, gd if gd is not None else torch.empty(0, device=x.device))
        ctx.has_depth = gd is not None
        ctx.depth_shape = tuple(depth.shape) if depth is not None else None
        ctx.terms = out            # [loss, Ll1, ssim, pseudo-depth loss]
        return out[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        gi, gd = ctx.saved_tensors
        return gi * grad_out, None, (gd.reshape(ctx.depth_shape) * grad_out if ctx.has_depth else None), None, None, None


def training_loss(image, gt_image, lambda_dssim=0.2, depth=None, pseudo_depth=None, depth_weight=0.1):
    """(1 - lambda) * l1_loss(image, gt) + lambda * (1 - ssim(image, gt)) [+ depth_weight * min(1 - pearson(-m, d),
    1 - pearson(1 / (m + 200), d))]   -- train.py:92-108."""
    return _TrainingLoss.apply(image, gt_image, depth, pseudo_depth, lambda_dssim, depth_weight)


def add_densification_stats(radii, viewspace_grad, max_radii2D, xyz_gradient_accum, denom):
    """train.py:142-145 + GaussianModel.add_densification_stats (gaussian_model.py:405-407), in place, one launch."""
    lib = _lib.load()
    P = radii.shape[0]
    p = lambda t: C.c_void_p(t.data_ptr())
    for t in (radii, viewspace_grad, max_radii2D, xyz_gradient_accum, denom):
        if t.device.type != "cuda" or not t.is_contiguous():
            raise RuntimeError("add_densification_stats needs contiguous HIP tensors")
    with torch.cuda.device(radii.device):
        _lib.check(lib.gsr_densification_stats(P, p(radii), p(viewspace_grad), p(max_radii2D), p(xyz_gradient_accum), p(denom),
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))
